/*
 * CloverVector4.h -- 4-bit quantized vector, MI355X-backed.
 *
 * Drop-in for the reference's include/CloverVector4.h: same class name, constructors, method names,
 * padding and data format (64-element blocks with one fp32 absolute-max scale, element 2i in the high
 * nibble of byte i, values immediately followed by the scales in ONE allocation, :68-103), but every
 * hot method is a call into libclover_hip.so (include/clover_hip.h) instead of an AVX2 loop:
 *
 *   quantize / quantize_parallel   -> clv4_quantize   (CloverVector4.h:605-807, :809-1024)
 *   restore                        -> clv4_restore    (:1027-1093)
 *   dot                            -> clv4_dot EXACT  (:1095-1192; bit-identical fp32 order)
 *   dot_parallel                   -> clv4_dot FAST   (:1793-1907; the reference's OpenMP reduction order is
 *                                                      unspecified, FAST is deterministic and tolerance-equal)
 *   *_scalar                       -> scalar HOST loops (clover_scalar.h): the reference's validation partners of the methods
 *                                     above (:336-595), kept as an independent implementation so that "kernel == scalar twin" means
 *                                     something; no hot method calls them
 *
 * Results are bit-identical to the reference when built with -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1; the
 * default build rounds stochastically from the same XORShift stream (setRandomKeys for determinism).
 * Element accessors (get/set/getBits/setBits/getData/...) work on the host copy and synchronise lazily.
 * The "next" rows of SURVEY.md 8(f) are here too: scaleAndAdd (:1196-1478) and threshold (:1913-2060).
 */
#ifndef CLOVER_VECTOR4_H
#define CLOVER_VECTOR4_H

#include <bitset>

#include "CloverVector32.h"
#include "clover_scalar.h"

class CloverVector4 {
protected:
    const uint64_t length;
    const uint64_t length_pad;
    mutable clover_hip::Mirror mem;            /* [length_pad/2 value bytes][scales] */
    mutable clover_hip::RandomState random;
    uint64_t value_bytes;
    /* a view built from two unrelated pointers cannot be one block: keep separate mirrors for it */
    mutable clover_hip::Mirror view_scales;
    bool split_view;

    void allocate()
    {
        const uint64_t blocks = length_pad / CLOVER_VECTOR_BLOCK;
        const uint64_t blocks_pad = clover_hip::round_up(blocks, CLOVER_VECTOR_BLOCK);
        value_bytes = length_pad / 2;
        mem.allocate(value_bytes + blocks_pad * sizeof(float));
        split_view = false;
        int8_t *v = values_rw();
        float *s = reinterpret_cast<float *>(v + value_bytes);
        for (uint64_t i = length >> 1; i < value_bytes; i++) v[i] = 0;          /* zeroed value padding  */
        for (uint64_t i = length / 64; i < blocks; i++) s[i] = 1;               /* padding scales = 1.0  */
    }

    static inline int8_t nibble(int8_t byte, uint64_t pos) { return (int8_t)((int8_t)(byte << ((pos % 2) * 4)) >> 4); }

public:
    explicit CloverVector4(uint64_t s) : length(s), length_pad(clover_hip::round_up(s, CLOVER_VECTOR_SIZE_PAD)) { allocate(); }

    /* non-owning view (CloverVector4.h:114-119) */
    CloverVector4(uint64_t s, int8_t *data, float *data_scales)
        : length(s), length_pad(clover_hip::round_up(s, CLOVER_VECTOR_SIZE_PAD))
    {
        value_bytes = length_pad / 2;
        mem.adopt(data, value_bytes);
        view_scales.adopt(data_scales, (length_pad / 64) * sizeof(float));
        split_view = true;
    }

    explicit CloverVector4(const CloverVector32 &other) : length(other.size()), length_pad(other.size_pad())
    {
        allocate();
        quantize(other);
    }

    CloverVector4(const CloverVector4 &other) : length(other.length), length_pad(other.length_pad)
    {
        allocate();
        memcpy(values_rw(), other.values_ro(), value_bytes);
        memcpy(scales_rw(), other.scales_ro(), (length_pad / 64) * sizeof(float));
    }

    /* ---- support methods (CloverVector4.h:150-326) ------------------------------------------- */
    uint64_t size() const { return length; }
    uint64_t size_pad() const { return length_pad; }
    uint64_t getBitsLength() const { return 4; }
    uint64_t getBytes() const { return length_pad / 2 + (length_pad / 64) * sizeof(float); }

    /* Explicit residency (clover_device.h, -DCLOVER_HIP_EXPLICIT_SYNC): move the bytes NOW instead of at the next use.  toDevice(): upload
     * if the host copy is the newer one; toHost(): bring a device result back.  Optional in every build (the default build's page tracking
     * and all accessors synchronise by themselves); not in the reference, which has one copy. */
    void toDevice() const { (void)mem.dev_ro(); if (split_view) (void)view_scales.dev_ro(); }
    void toHost() const { (void)mem.host_ro(); if (split_view) (void)view_scales.host_ro(); }
    /* Raw pointers as in the reference (:229-237): valid for the life of the object and always current -- reads through a
     * kept pointer see the results of later device operations, writes through it reach the next one (clover_device.h). */
    int8_t *getData() const { return reinterpret_cast<int8_t *>(mem.host_ptr()); }
    float *getScales() const
    {
        if (split_view) return reinterpret_cast<float *>(view_scales.host_ptr());
        return reinterpret_cast<float *>(mem.host_ptr() + value_bytes);
    }

    int8_t getBits(uint64_t pos) const { return nibble(values_ro()[pos >> 1], pos); }
    void setBits(uint64_t pos, int8_t bits)
    {
        int8_t *v = values_rw();
        const int8_t qu = (int8_t)((bits & 0x0F) << ((1 - pos % 2) * 4));
        v[pos >> 1] = (int8_t)((v[pos >> 1] & (pos % 2 == 0 ? 0x0F : 0xF0)) | qu);
    }
    float get(uint64_t pos) const
    {
        const float scale = scales_ro()[pos >> 6] / 7.0f;
        return scale * (float)nibble(values_ro()[pos >> 1], pos);
    }
    float getAbs(uint64_t pos) const { const float v = get(pos); return v < 0 ? -v : v; }
    /* the reference uses the approximate rcpss here (CloverVector4.h:211); an exact division is used instead */
    void set(uint64_t pos, float value)
    {
        const float scale = 7.0f / scales_ro()[pos >> 6];
        setBits(pos, (int8_t)roundf(value * scale));
    }
    void clear()
    {
        memset(values_rw(), 0, value_bytes);
        float *s = scales_rw();
        for (uint64_t b = 0; b < length_pad / 64; b++) s[b] = 1.0f;
    }
    std::string toString() const
    {
        std::stringstream sout;
        for (uint64_t i = 0; i < length_pad; i++) {
            sout << std::setw(10) << i << " | " << std::setw(20) << std::fixed << std::setprecision(7) << get(i) << " | "
                 << std::setw(20) << scales_ro()[i >> 6] << " | " << std::setw(5) << (int)getBits(i) << " | "
                 << std::bitset<8>((uint8_t)values_ro()[i >> 1]) << std::endl;
        }
        return sout.str();
    }

    /* CloverRandom::setRandomKeys (CloverRandom.h:90-94): the four 64-bit lanes of each key register */
    void setRandomKeys(const uint64_t key1[4], const uint64_t key2[4]) { random.set(key1, key2); }
#ifdef CLOVER_HIP_M256_KEYS
    void setRandomKeys(__m256i key1, __m256i key2) { clover_hip::set_keys_m256(random, key1, key2); }   /* the reference's signature */
#endif
    /* avx_xorshift128plus_init(key1, key2) + setRandomKeys in one call */
    void seedRandomKeys(uint64_t key1, uint64_t key2) { random.seed(key1, key2); }

    /* ---- hot path ---------------------------------------------------------------------------- */
    void quantize(const CloverVector32 &other)
    {
        if (other.size_pad() != length_pad) {
            std::cout << "Vectors do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        const float *x = other.device_ro();
        clover_hip::check(clv4_quantize(x, length_pad, dev_values_wo(), dev_scales_wo(), clover_hip::rng_or_null(random), nullptr),
                          "CloverVector4::quantize");
        commit();
    }
    void quantize_parallel(const CloverVector32 &other) { quantize(other); }
    /* the reference's scalar twin (:452-517), on the host: the validation partner of quantize (clover_scalar.h) */
    void quantize_scalar(const CloverVector32 &other)
    {
        if (other.size_pad() != length_pad) {
            std::cout << "Vectors do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::scalar::quantize4(other.host_ro(), length_pad, values_rw(), scales_rw());
    }

    void restore(CloverVector32 &other) const
    {
        clover_hip::check(clv4_restore(dev_values_ro(), dev_scales_ro(), length_pad, other.device_wo(), nullptr), "CloverVector4::restore");
        other.commit();
    }
    void restore_scalar(CloverVector32 &other) const      /* :519-553, on the host */
    {
        clover_hip::scalar::restore4(values_ro(), scales_ro(), length_pad, other.host_rw());
    }

    /* dot(): by default the reference's summation order, bit for bit (16 sequential fma chains: latency-bound by definition, 0.38 ms at
     * n = 2^24 -- NOT faster than one host core running the same order; the floor is n/128 dependent fmas).  Under -DCLOVER_FAST (or
     * clover_hip::set_exactness(clover_hip::FAST), or the older -DCLOVER_DOT_FAST) dot() is the fast order: exact integer block sums,
     * fp32 tree order -- memory-bound, one launch, within 2e-6 * sum|terms| of the exact order, as the reference's own dot_parallel
     * differs from its dot.  The switch and its threshold() half: clover_device.h.  dot_parallel() / dot_fast(): always the fast order;
     * dot_exact(): always the reference's.  RULE OF THUMB: n >= 2^20 and no need for the last bits -> dot_parallel(). */
    float dot(const CloverVector4 &other) const { return dot_mode(other, clover_hip::dot_mode()); }
    float dot_exact(const CloverVector4 &other) const { return dot_mode(other, CLV_DOT_EXACT); }
    float dot_parallel(const CloverVector4 &other) const { return dot_mode(other, CLV_DOT_FAST); }
    float dot_fast(const CloverVector4 &other) const { return dot_mode(other, CLV_DOT_FAST); }

    float dot_scalar(const CloverVector4 &other) const
    {
        const int8_t *u = values_ro(), *v = other.values_ro();
        const float *su = scales_ro(), *sv = other.scales_ro();
        float result = 0;
        for (uint64_t b = 0; b < length_pad / 64; b++) {
            int16_t acc = 0;
            for (uint64_t i = 32 * b; i < 32 * b + 32; i++)
                acc = (int16_t)(acc + nibble(u[i], 0) * nibble(v[i], 0) + nibble(u[i], 1) * nibble(v[i], 1));
            result += (su[b] / 7.0f) * (sv[b] / 7.0f) * (float)acc;
        }
        return result;
    }

    /* ---- next rows: the other steps of the quantized IHT / GD iterations ------------------ */
    /* this = quantize(this + a * other)   (CloverVector4.h:1196-1205; _parallel :1489-1499; _scalar :336-345) */
    void scaleAndAdd(const CloverVector4 &other, float a)
    {
        same_size(other);
        const int8_t *v = other.dev_values_ro();
        const float *sv = other.dev_scales_ro();
        int8_t *u = dev_values_rw();
        float *su = dev_scales_rw();
        clover_hip::check(clv4_scale_and_add(u, su, v, sv, a, length_pad, u, su, clover_hip::rng_or_null(random), nullptr),
                          "CloverVector4::scaleAndAdd");
        commit();
    }
    /* result = quantize(this + a * other) (CloverVector4.h:1207-1220) */
    void scaleAndAdd(const CloverVector4 &other, float a, CloverVector4 &result)
    {
        same_size(other);
        same_size(result);
        clover_hip::check(clv4_scale_and_add(dev_values_ro(), dev_scales_ro(), other.dev_values_ro(), other.dev_scales_ro(), a, length_pad,
                                             result.dev_values_wo(), result.dev_scales_wo(), clover_hip::rng_or_null(random), nullptr),
                          "CloverVector4::scaleAndAdd");
        result.commit();
    }
    void scaleAndAdd_parallel(const CloverVector4 &other, float a) { scaleAndAdd(other, a); }
    void scaleAndAdd_parallel(const CloverVector4 &other, float a, CloverVector4 &result) { scaleAndAdd(other, a, result); }
    /* :336-449, on the host */
    void scaleAndAdd_scalar(const CloverVector4 &other, float a)
    {
        same_size(other);
        const int8_t *v = other.values_ro();
        const float *sv = other.scales_ro();
        int8_t *u = values_rw();
        float *su = scales_rw();
        clover_hip::scalar::scale_and_add4(u, su, v, sv, a, length_pad, u, su);
    }
    void scaleAndAdd_scalar(CloverVector4 &other, float a, CloverVector4 &result)
    {
        same_size(other);
        same_size(result);
        clover_hip::scalar::scale_and_add4(values_ro(), scales_ro(), other.values_ro(), other.scales_ro(), a, length_pad, result.values_rw(),
                                           result.scales_rw());
    }

    /* keep the k largest magnitudes, zero the rest (CloverVector4.h:1913-2060) */
    void threshold(uint64_t k)
    {
        clover_hip::check(clv4_threshold_mode(dev_values_rw(), dev_scales_ro(), length, length_pad, k, clover_hip::threshold_mode(), nullptr, nullptr), "CloverVector4::threshold");
        commit();
    }
    void threshold_parallel(uint64_t k) { threshold(k); }
    /* threshold with the caller's own heap memory (CloverVector4.h:1929-1970, 1975-2057): ALWAYS the reference's walk, whatever the
     * exactness switch says -- the caller keeps the heap, so it gets the reference's: min_heap[i] = {|value|, bits, idx} of the K
     * survivors in the reference's array order.  k <= size().  _parallel: the reference merges per-thread heaps, which keeps the same K
     * largest but orders the array differently; here it is the sequential method (the device is the parallel implementation). */
    typedef clover_hip::idx_t idx_t;
    void threshold_min_heap(idx_t *min_heap, uint64_t k)
    {
        if (k == 0 || k > length) { std::cout << "threshold_min_heap: k must lie in 1 .. size(). Exiting ..." << std::endl; exit(1); }
        clover_hip::threshold_heap_to_host(clv4_threshold_heap, dev_values_rw(), dev_scales_ro(), length, length_pad, min_heap, k,
                                           "CloverVector4::threshold_min_heap");
        commit();
        for (uint64_t i = 0; i < k; i++) min_heap[i].bits.i = getBits(min_heap[i].idx);       /* survivors keep their bits */
    }
    void threshold_min_heap_parallel(idx_t *min_heaps, uint64_t k) { threshold_min_heap(min_heaps, k); }

    /* ---- device views, used by CloverMatrix4 ------------------------------------------------------ */
    const int8_t *dev_values_ro() const { return reinterpret_cast<const int8_t *>(mem.dev_ro()); }
    const float *dev_scales_ro() const
    {
        if (split_view) return reinterpret_cast<const float *>(view_scales.dev_ro());
        return reinterpret_cast<const float *>(mem.dev_ro() + value_bytes);
    }
    int8_t *dev_values_wo() { return reinterpret_cast<int8_t *>(mem.dev_wo()); }     /* kernels overwrite every used byte */
    float *dev_scales_wo()
    {
        if (split_view) return reinterpret_cast<float *>(view_scales.dev_wo());
        return reinterpret_cast<float *>(mem.dev_wo() + value_bytes);
    }

    int8_t *dev_values_rw() { return reinterpret_cast<int8_t *>(mem.dev_rw()); }
    float *dev_scales_rw()
    {
        if (split_view) return reinterpret_cast<float *>(view_scales.dev_rw());
        return reinterpret_cast<float *>(mem.dev_rw() + value_bytes);
    }
    /* after a launch that wrote through dev_*_wo()/dev_*_rw(): a view copies the result into the caller's memory now */
    void commit()
    {
        mem.commit();
        if (split_view) view_scales.commit();
    }

private:
    void same_size(const CloverVector4 &other) const
    {
        if (other.length_pad != length_pad) {
            std::cout << "Vectors do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
    }
    int8_t *values_rw() const { return reinterpret_cast<int8_t *>(mem.host_rw()); }
    float *scales_rw() const
    {
        if (split_view) return reinterpret_cast<float *>(view_scales.host_rw());
        return reinterpret_cast<float *>(mem.host_rw() + value_bytes);
    }
    const int8_t *values_ro() const { return reinterpret_cast<const int8_t *>(mem.host_ro()); }
    const float *scales_ro() const
    {
        if (split_view) return reinterpret_cast<const float *>(view_scales.host_ro());
        return reinterpret_cast<const float *>(mem.host_ro() + value_bytes);
    }
    float dot_mode(const CloverVector4 &other, int mode) const
    {
        if (other.length_pad != length_pad) {
            std::cout << "Vectors do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::ResultSlot &slot = clover_hip::result_slot();          /* per-thread device word + pinned host word */
        clover_hip::check(clv4_dot(dev_values_ro(), dev_scales_ro(), other.dev_values_ro(), other.dev_scales_ro(), length_pad, mode,
                                   slot.device(), nullptr, nullptr), "CloverVector4::dot");
        return slot.fetch();
    }
};

#endif
