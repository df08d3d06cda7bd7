/*
 * CloverMatrix4.h -- 4-bit quantized matrix, MI355X-backed.
 *
 * Drop-in for the reference's include/CloverMatrix4.h: same class name, constructor, method names and
 * data format (row-major nibbles followed by a row-major grid of fp32 scales, one per 64x64 tile;
 * rows/cols padded to multiples of 128; :77-139).  Hot methods call libclover_hip.so:
 *
 *   quantize                         -> clm4_quantize (CloverMatrix4.h:512-766)
 *   mvm / mvm_parallel / mvm_scalar  -> clm4_mvm      (:777-1083, :1681-2006, :311-392 -- all three give
 *                                                      the same result in the reference, bit for bit)
 *   gemm (new)                       -> clm4_gemm     (the reference has no GEMM; semantics in DESIGN.md)
 *
 * As in the reference, values/scales are not exposed (they are `protected` there, :73-75); the matrix
 * lives in HBM once quantized.  transpose and both mixed-precision mvm variants (4-bit x 8-bit with
 * CloverVector8 operands, 4-bit x fp32) are here as well.
 */
#ifndef CLOVER_MATRIX4_H
#define CLOVER_MATRIX4_H

#include "CloverMatrix32.h"
#include "CloverVector4.h"
#include "CloverVector8.h"

class CloverMatrix4 {
protected:
    const uint64_t rows;
    const uint64_t cols;
    mutable clover_hip::Mirror mem;            /* [rows*cols/2 value bytes][(rows/64)*(cols/64) scales] */
    mutable clover_hip::RandomState random;
    uint64_t value_bytes;

    const int8_t *dev_values() const { return reinterpret_cast<const int8_t *>(mem.dev_ro()); }
    const float *dev_scales() const { return reinterpret_cast<const float *>(mem.dev_ro() + value_bytes); }

public:
    CloverMatrix4(uint64_t h, uint64_t w)
        : rows(clover_hip::round_up(h, CLOVER_VECTOR_SIZE_PAD)), cols(clover_hip::round_up(w, CLOVER_VECTOR_SIZE_PAD))
    {
        value_bytes = rows * cols / 2;
        mem.allocate(value_bytes + (rows >> 6) * (cols >> 6) * sizeof(float));
    }

    uint64_t getRows() const { return rows; }
    uint64_t getCols() const { return cols; }
    uint64_t size() const { return rows * cols; }
    uint64_t getBitsLength() const { return 4; }
    uint64_t getBytes() const { return value_bytes + (rows >> 6) * (cols >> 6) * sizeof(float); }

    /* host views of the packed values and of the tile scales (the reference keeps them protected; exposed for interop) */
    int8_t *getData() const { return reinterpret_cast<int8_t *>(mem.host_ptr()); }
    float *getScales() const { return reinterpret_cast<float *>(mem.host_ptr() + value_bytes); }

    float get(uint64_t i, uint64_t j) const
    {
        const uint8_t *h = mem.host_ro();
        const float *s = reinterpret_cast<const float *>(h + value_bytes);
        const float scale = s[(i >> 6) * (cols >> 6) + (j >> 6)] / 7.0f;
        const uint64_t pos = i * cols + j;
        const int8_t b = (int8_t)h[pos >> 1];
        return scale * (float)(int8_t)((int8_t)(b << ((pos % 2) * 4)) >> 4);
    }

    void setRandomKeys(const uint64_t key1[4], const uint64_t key2[4]) { random.set(key1, key2); }
#ifdef CLOVER_HIP_M256_KEYS
    void setRandomKeys(__m256i key1, __m256i key2) { clover_hip::set_keys_m256(random, key1, key2); }   /* CloverRandom.h:90-94 */
#endif
    void seedRandomKeys(uint64_t key1, uint64_t key2) { random.seed(key1, key2); }

    void quantize(const CloverMatrix32 &m)
    {
        if (m.getRows() != rows || m.getCols() != cols) {
            std::cout << "Matrices do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        const float *A = m.device_ro();
        uint8_t *d = mem.dev_wo();
        clover_hip::check(clm4_quantize(A, rows, cols, reinterpret_cast<int8_t *>(d), reinterpret_cast<float *>(d + value_bytes),
                                        clover_hip::rng_or_null(random), nullptr), "CloverMatrix4::quantize");
    }
    void quantize_scalar(const CloverMatrix32 &m) { quantize(m); }

    /* CloverMatrix4.h:266-301 (the reference has only the scalar variant) */
    void restore_scalar(CloverMatrix32 &other) const
    {
        if (other.getRows() != rows || other.getCols() != cols) {
            std::cout << "Matrices do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::check(clm4_restore(dev_values(), dev_scales(), rows, cols, other.device_wo(), nullptr), "CloverMatrix4::restore");
    }
    void restore(CloverMatrix32 &other) const { restore_scalar(other); }

    void mvm(const CloverVector4 &productVector, CloverVector4 &resultVector)
    {
        if (productVector.size() != getCols()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        if (resultVector.size_pad() != getRows()) {                 /* checked by mvm_scalar in the reference (:313) */
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        const int8_t *x = productVector.dev_values_ro();
        const float *sx = productVector.dev_scales_ro();
        clover_hip::check(clm4_mvm(dev_values(), dev_scales(), rows, cols, x, sx, resultVector.dev_values_wo(),
                                   resultVector.dev_scales_wo(), clover_hip::rng_or_null(random), nullptr), "CloverMatrix4::mvm");
        resultVector.commit();
    }
    void mvm_parallel(const CloverVector4 &productVector, CloverVector4 &resultVector) { mvm(productVector, resultVector); }
    void check_fused(const CloverVector4 &x, const CloverVector4 &u, const CloverVector4 &t) const
    {
        if (x.size() != getCols() || t.size_pad() != getRows()) { std::cout << "MVM can not be performed. Exiting ..." << std::endl; exit(1); }
        if (u.size_pad() != getRows()) { std::cout << "Vectors do not have the same size. Exiting ..." << std::endl; exit(1); }
    }
    void mvm_scalar(const CloverVector4 &productVector, CloverVector4 &resultVector) { mvm(productVector, resultVector); }

    /* Not in the reference: t = this * x immediately followed by r = quantize(u + a * t), the pair of steps the IHT / GD
     * loops repeat (01_measure.h:930-931, :932-933).  One launch when rounding is deterministic; with stochastic rounding
     * the two steps draw from two objects' generators (the matrix's, then u's), as they do in the reference, and are
     * issued as the two calls.  Results are identical to mvm(x, t); u.scaleAndAdd(t, a, r) either way. */
    void mvm_scaleAndAdd(const CloverVector4 &x, const CloverVector4 &u, float a, CloverVector4 &t, CloverVector4 &r)
    {
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
        check_fused(x, u, t);
        if (r.size_pad() != getRows()) { std::cout << "Vectors do not have the same size. Exiting ..." << std::endl; exit(1); }
        clover_hip::check(clm4_mvm_scale_and_add(dev_values(), dev_scales(), rows, cols, x.dev_values_ro(), x.dev_scales_ro(),
                                                 u.dev_values_ro(), u.dev_scales_ro(), a, t.dev_values_wo(), t.dev_scales_wo(),
                                                 r.dev_values_wo(), r.dev_scales_wo(), nullptr, nullptr), "CloverMatrix4::mvm_scaleAndAdd");
        t.commit();
        r.commit();
#else
        mvm(x, t);
        const_cast<CloverVector4 &>(u).scaleAndAdd(t, a, r);
#endif
    }
    /* in place: u = quantize(u + a * (this * x)) */
    void mvm_scaleAndAdd(const CloverVector4 &x, CloverVector4 &u, float a, CloverVector4 &t)
    {
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
        check_fused(x, u, t);
        int8_t *qu = u.dev_values_rw();
        float *su = u.dev_scales_rw();
        clover_hip::check(clm4_mvm_scale_and_add(dev_values(), dev_scales(), rows, cols, x.dev_values_ro(), x.dev_scales_ro(), qu, su, a,
                                                 t.dev_values_wo(), t.dev_scales_wo(), qu, su, nullptr, nullptr),
                          "CloverMatrix4::mvm_scaleAndAdd");
        t.commit();
        u.commit();
#else
        mvm(x, t);
        u.scaleAndAdd(t, a);
#endif
    }

    /* mixed precision: 8-bit vector in, 8-bit vector out (CloverMatrix4.h:1093-1441; _parallel :2017-2387; _scalar :402-413) */
    void mvm(const CloverVector8 &productVector, CloverVector8 &resultVector)
    {
        if (productVector.size() != getCols() || resultVector.size_pad() != getRows()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::check(clm4_mvm_v8(dev_values(), dev_scales(), rows, cols, productVector.dev_values_ro(), productVector.dev_scales_ro(),
                                      resultVector.dev_values_wo(), resultVector.dev_scales_wo(), clover_hip::rng_or_null(random), nullptr),
                          "CloverMatrix4::mvm");
        resultVector.commit();
    }
    void mvm_parallel(const CloverVector8 &productVector, CloverVector8 &resultVector) { mvm(productVector, resultVector); }
    void mvm_scalar(const CloverVector8 &productVector, CloverVector8 &resultVector) { mvm(productVector, resultVector); }
    /* the same pairing as mvm_scaleAndAdd above, for 8-bit vectors */
    void mvm_scaleAndAdd(const CloverVector8 &x, const CloverVector8 &u, float a, CloverVector8 &t, CloverVector8 &r)
    {
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
        if (x.size() != getCols() || t.size_pad() != getRows()) { std::cout << "MVM can not be performed. Exiting ..." << std::endl; exit(1); }
        if (u.size_pad() != getRows() || r.size_pad() != getRows()) { std::cout << "Vectors do not have the same size. Exiting ..." << std::endl; exit(1); }
        clover_hip::check(clm4_mvm_v8_scale_and_add(dev_values(), dev_scales(), rows, cols, x.dev_values_ro(), x.dev_scales_ro(),
                                                    u.dev_values_ro(), u.dev_scales_ro(), a, t.dev_values_wo(), t.dev_scales_wo(),
                                                    r.dev_values_wo(), r.dev_scales_wo(), nullptr, nullptr), "CloverMatrix4::mvm_scaleAndAdd");
        t.commit();
        r.commit();
#else
        mvm(x, t);
        const_cast<CloverVector8 &>(u).scaleAndAdd(t, a, r);
#endif
    }
    void mvm_scaleAndAdd(const CloverVector8 &x, CloverVector8 &u, float a, CloverVector8 &t)
    {
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
        if (x.size() != getCols() || t.size_pad() != getRows()) { std::cout << "MVM can not be performed. Exiting ..." << std::endl; exit(1); }
        if (u.size_pad() != getRows()) { std::cout << "Vectors do not have the same size. Exiting ..." << std::endl; exit(1); }
        int8_t *qu = u.dev_values_rw();
        float *su = u.dev_scales_rw();
        clover_hip::check(clm4_mvm_v8_scale_and_add(dev_values(), dev_scales(), rows, cols, x.dev_values_ro(), x.dev_scales_ro(), qu, su, a,
                                                    t.dev_values_wo(), t.dev_scales_wo(), qu, su, nullptr, nullptr),
                          "CloverMatrix4::mvm_scaleAndAdd");
        t.commit();
        u.commit();
#else
        mvm(x, t);
        u.scaleAndAdd(t, a);
#endif
    }

    /* mixed precision: fp32 vector in, fp32 vector out (CloverMatrix4.h:1451-1547; _parallel :2397-2505) */
    void mvm(const CloverVector32 &productVector, CloverVector32 &resultVector)
    {
        if (productVector.size() != getCols() || resultVector.size_pad() != getRows()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::check(clm4_mvm_f32(dev_values(), dev_scales(), rows, cols, productVector.device_ro(), resultVector.device_wo(), nullptr),
                          "CloverMatrix4::mvm");
        resultVector.commit();
    }
    void mvm_parallel(const CloverVector32 &productVector, CloverVector32 &resultVector) { mvm(productVector, resultVector); }
    /* the reference's scalar variant accumulates in double (:423-432); the SIMD order is used here */
    void mvm_scalar(const CloverVector32 &productVector, CloverVector32 &resultVector) { mvm(productVector, resultVector); }

    /* other = this^T  (CloverMatrix4.h:1549-1663; _parallel :2508-2640; _scalar :435-502) */
    void transpose(CloverMatrix4 &other) const
    {
        if (other.rows != cols || other.cols != rows) {
            std::cout << "Matrix can not be transposed. Exiting ..." << std::endl;
            exit(1);
        }
        uint8_t *d = other.mem.dev_wo();
        clover_hip::check(clm4_transpose(dev_values(), dev_scales(), rows, cols, reinterpret_cast<int8_t *>(d),
                                         reinterpret_cast<float *>(d + other.value_bytes), nullptr), "CloverMatrix4::transpose");
    }
    void transpose_parallel(CloverMatrix4 &other) const { transpose(other); }
    void transpose_scalar(CloverMatrix4 &other) const { transpose(other); }
    void transpose_scalar_faster(CloverMatrix4 &other) const { transpose(other); }      /* CloverMatrix4.h:2649-2799 */

    /* C = this * B^T, fp32: this is M x K, B is N x K, C is M x N (build-defined) */
    void gemm(const CloverMatrix4 &B, CloverMatrix32 &C) const
    {
        if (B.cols != cols || C.getRows() != rows || C.getCols() != B.rows) {
            std::cout << "GEMM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::check(clm4_gemm(dev_values(), dev_scales(), rows, cols, B.dev_values(), B.dev_scales(), B.rows, C.device_wo(), nullptr),
                          "CloverMatrix4::gemm");
    }
};

#endif
