/*
 * CloverVector8.h -- 8-bit quantized vector, MI355X-backed: the operand type of the mixed-precision
 * CloverMatrix4::mvm(const CloverVector8 &, CloverVector8 &).
 *
 * Same class name, constructors, data format and element accessors as the reference's include/CloverVector8.h
 * (int8 values in natural order, one fp32 absolute-max scale per 64 elements, value = q * scale / 127, values and
 * scales in ONE allocation, :42-80), with
 *
 *   quantize / quantize_parallel / quantize_scalar  -> clv8_quantize  (CloverVector8.h:393-606, :607-833, :205-253)
 *   restore / restore_scalar                        -> clv8_restore   (:835-909, :255-266)
 *
 *   scaleAndAdd / _parallel / _scalar               -> clv8_scale_and_add (:1063-1358, :1360-1678, :311-391)
 *   threshold / threshold_parallel                  -> clv8_threshold_mode (:1680-1740; tie rule by the exactness switch, clover_device.h)
 *   dot / dot_parallel / dot_scalar                 -> clv8_dot EXACT / FAST   (:911-977, :979-1061, :268-310; round 5)
 *
 * on the device: what Q_IHT / Q_GD need when they run with a CloverMatrix4 and CloverVector8 vectors, the configuration the
 * reference publishes as "4-bit" (test/performance/02_bit04.cpp:140), and with round 5 every method of the reference's class.
 * CloverMatrix8 belongs to the 8-bit containers' own path and is not part of this backend.
 */
#ifndef CLOVER_VECTOR8_H
#define CLOVER_VECTOR8_H

#include "CloverVector32.h"
#include "clover_scalar.h"

class CloverVector8 {
protected:
    const uint64_t length;
    const uint64_t length_pad;
    mutable clover_hip::Mirror mem;            /* [length_pad value bytes][scales] */
    mutable clover_hip::RandomState random;
    uint64_t value_bytes;
    mutable clover_hip::Mirror view_scales;    /* non-owning view: two unrelated pointers */
    bool split_view;

    void allocate()
    {
        const uint64_t blocks = length_pad / CLOVER_VECTOR_BLOCK;
        const uint64_t blocks_pad = clover_hip::round_up(blocks, CLOVER_VECTOR_BLOCK);
        value_bytes = length_pad;
        mem.allocate(value_bytes + blocks_pad * sizeof(float));
        split_view = false;
        int8_t *v = values_rw();
        float *s = reinterpret_cast<float *>(v + value_bytes);
        for (uint64_t i = length; i < length_pad; i++) v[i] = 0;                /* zeroed value padding  */
        for (uint64_t i = length / 64; i < blocks; i++) s[i] = 1;               /* padding scales = 1.0  */
    }

public:
    explicit CloverVector8(uint64_t s) : length(s), length_pad(clover_hip::round_up(s, CLOVER_VECTOR_SIZE_PAD)) { allocate(); }

    /* non-owning view (CloverVector8.h:84-89) */
    CloverVector8(uint64_t s, int8_t *data_values, float *data_scales)
        : length(s), length_pad(clover_hip::round_up(s, CLOVER_VECTOR_SIZE_PAD))
    {
        value_bytes = length_pad;
        mem.adopt(data_values, value_bytes);
        view_scales.adopt(data_scales, (length_pad / 64) * sizeof(float));
        split_view = true;
    }

    explicit CloverVector8(const CloverVector32 &other) : length(other.size()), length_pad(other.size_pad())
    {
        allocate();
        quantize(other);
    }

    CloverVector8(const CloverVector8 &other) : length(other.length), length_pad(other.length_pad)
    {
        allocate();
        memcpy(values_rw(), other.values_ro(), value_bytes);
        memcpy(scales_rw(), other.scales_ro(), (length_pad / 64) * sizeof(float));
    }

    uint64_t size() const { return length; }
    uint64_t size_pad() const { return length_pad; }
    uint64_t getBitsLength() const { return 8; }
    uint64_t getBytes() const { return length_pad + (length_pad / 64) * sizeof(float); }

    /* Explicit residency (clover_device.h, -DCLOVER_HIP_EXPLICIT_SYNC): move the bytes NOW instead of at the next use.  toDevice(): upload
     * if the host copy is the newer one; toHost(): bring a device result back.  Optional in every build (the default build's page tracking
     * and all accessors synchronise by themselves); not in the reference, which has one copy. */
    void toDevice() const { (void)mem.dev_ro(); if (split_view) (void)view_scales.dev_ro(); }
    void toHost() const { (void)mem.host_ro(); if (split_view) (void)view_scales.host_ro(); }
    /* raw pointers as in the reference: valid for the life of the object and always current (clover_device.h) */
    int8_t *getData() const { return reinterpret_cast<int8_t *>(mem.host_ptr()); }
    float *getScales() const
    {
        if (split_view) return reinterpret_cast<float *>(view_scales.host_ptr());
        return reinterpret_cast<float *>(mem.host_ptr() + value_bytes);
    }

    /* CloverVector8.h:137-140 */
    float get(uint64_t i) const { return values_ro()[i] * scales_ro()[i >> 6] / 127.0f; }
    /* :141-147 */
    float getAbs(uint64_t i) const
    {
        const float v = get(i);
        return v < 0 ? -v : (v == 0 ? 0.0f : v);            /* the sign bit cleared (-0 -> +0) */
    }
    /* :149-153 */
    void set(uint64_t i, float v) const
    {
        const float rcp_scale = 127.0f / scales_ro()[i >> 6];
        values_rw()[i] = (int8_t)roundf(rcp_scale * v);
    }
    int8_t getBits(uint64_t i) const { return values_ro()[i]; }
    void setBits(uint64_t i, int8_t bits) { values_rw()[i] = bits; }
    void clear()
    {
        memset(values_rw(), 0, value_bytes);
        float *s = scales_rw();
        for (uint64_t b = 0; b < length_pad / 64; b++) s[b] = 1.0f;
    }
    std::string toString() const
    {
        std::stringstream sout;
        for (uint64_t i = 0; i < length_pad; i++)
            sout << std::setw(10) << i << " | " << std::setw(20) << std::fixed << std::setprecision(7) << get(i) << " | " << std::setw(20)
                 << scales_ro()[i >> 6] << " | " << std::setw(5) << (int)values_ro()[i] << std::endl;
        return sout.str();
    }

    void setRandomKeys(const uint64_t key1[4], const uint64_t key2[4]) { random.set(key1, key2); }
#ifdef CLOVER_HIP_M256_KEYS
    void setRandomKeys(__m256i key1, __m256i key2) { clover_hip::set_keys_m256(random, key1, key2); }   /* CloverRandom.h:90-94 */
#endif
    void seedRandomKeys(uint64_t key1, uint64_t key2) { random.seed(key1, key2); }

    void quantize(const CloverVector32 &other)
    {
        if (other.size_pad() != length_pad) {
            std::cout << "Vectors do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::check(clv8_quantize(other.device_ro(), length_pad, dev_values_wo(), dev_scales_wo(), clover_hip::rng_or_null(random),
                                        nullptr), "CloverVector8::quantize");
        commit();
    }
    void quantize_parallel(const CloverVector32 &other) { quantize(other); }
    void quantize_scalar(const CloverVector32 &other)      /* CloverVector8.h:205-253, on the host (clover_scalar.h) */
    {
        if (other.size_pad() != length_pad) {
            std::cout << "Vectors do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::scalar::quantize8(other.host_ro(), length_pad, values_rw(), scales_rw());
    }

    void restore(CloverVector32 &other) const
    {
        clover_hip::check(clv8_restore(dev_values_ro(), dev_scales_ro(), length_pad, other.device_wo(), nullptr), "CloverVector8::restore");
        other.commit();
    }
    void restore_scalar(CloverVector32 &other) const { clover_hip::scalar::restore8(values_ro(), scales_ro(), length_pad, other.host_rw()); }

    /* this = quantize(this + a * other)   (CloverVector8.h:1063-1072) */
    void scaleAndAdd(const CloverVector8 &other, float a)
    {
        same_size(other);
        const int8_t *v = other.dev_values_ro();
        const float *sv = other.dev_scales_ro();
        int8_t *u = dev_values_rw();
        float *su = dev_scales_rw();
        clover_hip::check(clv8_scale_and_add(u, su, v, sv, a, length_pad, u, su, clover_hip::rng_or_null(random), nullptr),
                          "CloverVector8::scaleAndAdd");
        commit();
    }
    /* result = quantize(this + a * other) (CloverVector8.h:1074-1087) */
    void scaleAndAdd(const CloverVector8 &other, float a, CloverVector8 &result)
    {
        same_size(other);
        same_size(result);
        clover_hip::check(clv8_scale_and_add(dev_values_ro(), dev_scales_ro(), other.dev_values_ro(), other.dev_scales_ro(), a, length_pad,
                                             result.dev_values_wo(), result.dev_scales_wo(), clover_hip::rng_or_null(random), nullptr),
                          "CloverVector8::scaleAndAdd");
        result.commit();
    }
    void scaleAndAdd_parallel(const CloverVector8 &other, float a) { scaleAndAdd(other, a); }
    void scaleAndAdd_parallel(const CloverVector8 &other, float a, CloverVector8 &result) { scaleAndAdd(other, a, result); }
    void scaleAndAdd_scalar(const CloverVector8 &other, float a)      /* CloverVector8.h:311-391, on the host */
    {
        same_size(other);
        const int8_t *v = other.values_ro();
        const float *sv = other.scales_ro();
        int8_t *u = values_rw();
        float *su = scales_rw();
        clover_hip::scalar::scale_and_add8(u, su, v, sv, a, length_pad, u, su);
    }
    void scaleAndAdd_scalar(const CloverVector8 &other, float a, CloverVector8 &result)
    {
        same_size(other);
        same_size(result);
        clover_hip::scalar::scale_and_add8(values_ro(), scales_ro(), other.values_ro(), other.scales_ro(), a, length_pad, result.values_rw(),
                                           result.scales_rw());
    }

    /* keep the k largest magnitudes, zero the rest (CloverVector8.h:1680-1740) */
    /* dot(): by default the reference's order, bit for bit -- 8 sequential fma chains of n / 64 steps (CloverVector8.h:911-977): latency-bound
     * by that definition.  Under -DCLOVER_FAST / clover_hip::set_exactness(FAST) the fast order (exact block integers, fp32 tree, one launch,
     * memory-bound).  dot_parallel(): always the fast order (the reference's own is "any order", :979-1061); dot_exact(): always the
     * reference's.  The switch: clover_device.h. */
    float dot(const CloverVector8 &other) const { return dot_mode(other, clover_hip::dot_mode()); }
    float dot_exact(const CloverVector8 &other) const { return dot_mode(other, CLV_DOT_EXACT); }
    float dot_parallel(const CloverVector8 &other) const { return dot_mode(other, CLV_DOT_FAST); }
    float dot_fast(const CloverVector8 &other) const { return dot_mode(other, CLV_DOT_FAST); }
    /* :268-310, on the host: the validation partner of dot */
    float dot_scalar(const CloverVector8 &other) const
    {
        same_size(other);
        const int8_t *u = values_ro(), *v = other.values_ro();
        const float *su = scales_ro(), *sv = other.scales_ro();
        float result = 0;
        for (uint64_t b = 0; b < length_pad / 64; b++) {
            const float scale = (su[b] / 127.0f) * (sv[b] / 127.0f);
            int32_t block = 0;
            for (uint64_t i = 64 * b; i < 64 * b + 64; i++) block += (int32_t)u[i] * (int32_t)v[i];
            result += block * scale;
        }
        return result;
    }

    void threshold(uint64_t k)
    {
        clover_hip::check(clv8_threshold_mode(dev_values_rw(), dev_scales_ro(), length, length_pad, k, clover_hip::threshold_mode(), nullptr, nullptr), "CloverVector8::threshold");
        commit();
    }
    void threshold_parallel(uint64_t k) { threshold(k); }
    /* threshold with the caller's own heap memory (CloverVector8.h:1696-1737, 1742-1824): as CloverVector4::threshold_min_heap */
    typedef clover_hip::idx_t idx_t;
    void threshold_min_heap(idx_t *min_heap, uint64_t k)
    {
        if (k == 0 || k > length) { std::cout << "threshold_min_heap: k must lie in 1 .. size(). Exiting ..." << std::endl; exit(1); }
        clover_hip::threshold_heap_to_host(clv8_threshold_heap, dev_values_rw(), dev_scales_ro(), length, length_pad, min_heap, k,
                                           "CloverVector8::threshold_min_heap");
        commit();
        for (uint64_t i = 0; i < k; i++) min_heap[i].bits.i = getBits(min_heap[i].idx);
    }
    void threshold_min_heap_parallel(idx_t *min_heaps, uint64_t k) { threshold_min_heap(min_heaps, k); }

    /* ---- device views, used by CloverMatrix4 ------------------------------------------------------ */
    const int8_t *dev_values_ro() const { return reinterpret_cast<const int8_t *>(mem.dev_ro()); }
    const float *dev_scales_ro() const
    {
        if (split_view) return reinterpret_cast<const float *>(view_scales.dev_ro());
        return reinterpret_cast<const float *>(mem.dev_ro() + value_bytes);
    }
    int8_t *dev_values_wo() { return reinterpret_cast<int8_t *>(mem.dev_wo()); }
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
    float dot_mode(const CloverVector8 &other, int mode) const
    {
        same_size(other);
        clover_hip::ResultSlot &slot = clover_hip::result_slot();          /* per-thread device word + pinned host word */
        clover_hip::check(clv8_dot(dev_values_ro(), dev_scales_ro(), other.dev_values_ro(), other.dev_scales_ro(), length_pad, mode,
                                   slot.device(), nullptr, nullptr), "CloverVector8::dot");
        return slot.fetch();
    }
    void same_size(const CloverVector8 &other) const
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
};

#endif
