/*
 * CloverVector32.h -- fp32 vector container: the INPUT of quantize and the OUTPUT of restore.
 *
 * Same class name, constructors, padding and accessors as the reference's include/CloverVector32.h
 * (:47-70 constructors, :93-148 accessors; padding to a multiple of 128 with zeroed tail from
 * CloverVector.h:86-92), written from scratch on top of clover_device.h.  The 4-bit path touches the storage
 * part only; the class's own fp32 arithmetic (dot / scaleAndAdd / threshold, :160-684) is provided on the HOST
 * (clover_fp32.h: plain loops in the reference's order, nothing on the GPU) so that code comparing 4-bit results
 * with the 32-bit ones, or running Q_IHT / Q_GD on the fp32 classes as its baseline, compiles unchanged.
 */
#ifndef CLOVER_VECTOR32_H
#define CLOVER_VECTOR32_H

#include <iomanip>
#include <sstream>
#include <string>

#include "clover_device.h"
#include "clover_fp32.h"

#define CLOVER_VECTOR_BLOCK 64
#define CLOVER_VECTOR_SIZE_PAD (CLOVER_VECTOR_BLOCK * 2)

class CloverVector32 {
protected:
    const uint64_t length;
    const uint64_t length_pad;
    mutable clover_hip::Mirror mem;

public:
    explicit CloverVector32(uint64_t s) : length(s), length_pad(clover_hip::round_up(s, CLOVER_VECTOR_SIZE_PAD))
    {
        mem.allocate(length_pad * sizeof(float));
        float *v = reinterpret_cast<float *>(mem.host_rw());
        for (uint64_t i = length; i < length_pad; i++) v[i] = 0;      /* zeroed padding (CloverVector32.h:61-63) */
    }
    /* non-owning view over caller memory holding size_pad() floats (CloverVector32.h:47-51) */
    CloverVector32(uint64_t s, float *data) : length(s), length_pad(clover_hip::round_up(s, CLOVER_VECTOR_SIZE_PAD))
    {
        mem.adopt(data, length_pad * sizeof(float));
    }
    CloverVector32(const CloverVector32 &other) : length(other.length), length_pad(other.length_pad)
    {
        mem.allocate(length_pad * sizeof(float));
        memcpy(mem.host_rw(), other.mem.host_ro(), length_pad * sizeof(float));
    }

    uint64_t size() const { return length; }
    uint64_t size_pad() const { return length_pad; }
    uint64_t getBitsLength() const { return 32; }
    uint64_t getBytes() const { return length_pad * sizeof(float); }

    float get(uint64_t i) const { return reinterpret_cast<const float *>(mem.host_ro())[i]; }
    float getAbs(uint64_t i) const { float v = get(i); return v < 0 ? -v : v; }
    void set(uint64_t i, float v) { reinterpret_cast<float *>(mem.host_rw())[i] = v; }
    /* Explicit residency (clover_device.h, -DCLOVER_HIP_EXPLICIT_SYNC): move the bytes NOW instead of at the next use.  toDevice(): upload
     * if the host copy is the newer one; toHost(): bring a device result back.  Optional in every build (the default build's page tracking
     * and all accessors synchronise by themselves); not in the reference, which has one copy. */
    void toDevice() const { (void)mem.dev_ro(); }
    void toHost() const { (void)mem.host_ro(); }
    float *getData() const { return reinterpret_cast<float *>(mem.host_ptr()); }      /* stays valid and current, see clover_device.h */

    void clear() { memset(mem.host_rw(), 0, length_pad * sizeof(float)); }
    /* from now on a view over `data` (size_pad() floats, caller-owned): CloverVector32.h:111-114 */
    void setData(float *data) { mem.adopt(data, length_pad * sizeof(float)); }

    /* ---- fp32 arithmetic on the host (clover_fp32.h; CloverVector32.h:160-684): the comparison baseline of the 4-bit path ---- */
    /* "quantize" into 32 bits is a copy, "restore" the copy back (:217-289) */
    void quantize(const CloverVector32 &other) { memcpy(mem.host_rw(), other.mem.host_ro(), length_pad * sizeof(float)); }
    void quantize_scalar(const CloverVector32 &other) { quantize(other); }
    void quantize_parallel(const CloverVector32 &other) { quantize(other); }
    void restore(CloverVector32 &other) const { memcpy(other.mem.host_rw(), mem.host_ro(), length_pad * sizeof(float)); }
    float dot(const CloverVector32 &other) const { return clover_fp32::dot_chains32(host_ro(), other.host_ro(), length_pad); }
    float dot_scalar(const CloverVector32 &other) const { return clover_fp32::dot_sequential(host_ro(), other.host_ro(), length); }
    /* the reference's team reduction has no fixed order (:458-530); this one has: the sequential method's */
    float dot_parallel(const CloverVector32 &other) const { return dot(other); }
    /* this += s * other; (other, s, result): result = this + s * other (the reference writes through a const reference, :334-345) */
    void scaleAndAdd(const CloverVector32 &other, float s) { clover_fp32::axpy_fma(host_rw(), other.host_ro(), s, host_rw(), length_pad, false); }
    void scaleAndAdd(const CloverVector32 &other, float s, const CloverVector32 &result) const
    {
        clover_fp32::axpy_fma(host_ro(), other.host_ro(), s, const_cast<CloverVector32 &>(result).host_rw(), length_pad, false);
    }
    void scaleAndAdd_parallel(const CloverVector32 &other, float s) { clover_fp32::axpy_fma(host_rw(), other.host_ro(), s, host_rw(), length_pad, true); }
    void scaleAndAdd_parallel(const CloverVector32 &other, float s, const CloverVector32 &result) const
    {
        clover_fp32::axpy_fma(host_ro(), other.host_ro(), s, const_cast<CloverVector32 &>(result).host_rw(), length_pad, true);
    }
    void scaleAndAdd_scalar(const CloverVector32 &other, float s) { clover_fp32::axpy_two_roundings(host_rw(), other.host_ro(), s, host_rw(), length_pad); }
    void scaleAndAdd_scalar(const CloverVector32 &other, float s, const CloverVector32 &result) const
    {
        clover_fp32::axpy_two_roundings(host_ro(), other.host_ro(), s, const_cast<CloverVector32 &>(result).host_rw(), length_pad);
    }
    /* keep the k largest |values| of the first size() elements, zero the others: the reference's survivors (:533-600) */
    void threshold(uint64_t k) { clover_fp32::keep_top_k(host_rw(), length, k); }
    void threshold_parallel(uint64_t k) { threshold(k); }      /* (the reference merges per-thread heaps, :602-682: same magnitudes, team-dependent ties) */

    /* test data like the reference's setRandomInteger (CloverVector32.h:697-744): integers uniform in
     * [-max, max].  Uses a splitmix64 stream, not the reference's XORShift keys. */
    void setRandomInteger(float max_value, uint64_t seed = 0x2545F4914F6CDD1Dull)
    {
        float *v = reinterpret_cast<float *>(mem.host_rw());
        const int64_t range = (int64_t)max_value;
        uint64_t z = seed;
        for (uint64_t i = 0; i < length; i++) {
            z += 0x9E3779B97F4A7C15ull;
            uint64_t r = z;
            r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
            r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
            r ^= r >> 31;
            v[i] = (float)((int64_t)(r % (uint64_t)(2 * range + 1)) - range);
        }
    }

    /* uniform floats in [lo, hi) (CloverVector32.h:746-790), same splitmix64 stream as setRandomInteger */
    void setRandomFloats(float min_value, float max_value, uint64_t seed = 0x2545F4914F6CDD1Dull)
    {
        clover_fp32::fill_uniform(reinterpret_cast<float *>(mem.host_rw()), length, min_value, max_value, seed);
    }

    std::string toString() const
    {
        std::stringstream sout;
        for (uint64_t i = 0; i < length; i++)
            sout << std::setw(10) << i << " | " << std::setw(20) << std::fixed << std::setprecision(7) << get(i) << std::endl;
        return sout.str();
    }

    /* host access for the scalar validation twins (clover_scalar.h) */
    const float *host_ro() const { return reinterpret_cast<const float *>(mem.host_ro()); }
    float *host_rw() { return reinterpret_cast<float *>(mem.host_rw()); }
    /* device access for the 4-bit containers */
    const float *device_ro() const { return reinterpret_cast<const float *>(mem.dev_ro()); }
    float *device_wo() { return reinterpret_cast<float *>(mem.dev_wo()); }
    void commit() { mem.commit(); }      /* after a launch that wrote through device_wo(): a view is written through */
};

#endif
