/*
 * CloverMatrix4.h -- 4-bit quantized matrix, MI355X-backed.
 *
 * Drop-in for the reference's include/CloverMatrix4.h: same class name, constructor, method names and
 * data format (row-major nibbles followed by a row-major grid of fp32 scales, one per 64x64 tile;
 * rows/cols padded to multiples of 128; :77-139).  Hot methods call libclover_hip.so:
 *
 *   quantize                         -> clm4_quantize (CloverMatrix4.h:512-766)
 *   mvm / mvm_parallel               -> clm4_mvm      (:777-1083, :1681-2006)
 *   *_scalar                         -> scalar HOST code (clover_scalar.h), the reference's validation partners: quantize_scalar
 *                                       (:178-264), restore_scalar (:266-301), mvm_scalar (:311-432; for 4-bit vectors: row views +
 *                                       dot(), as there), transpose_scalar (:435-502); in the reference mvm == mvm_parallel ==
 *                                       mvm_scalar bit for bit when rounding is disabled, and so here
 *   gemm (new)                       -> clm4_gemm     (the reference has no GEMM; semantics in DESIGN.md)
 *
 * As in the reference, values/scales are not exposed (they are `protected` there, :73-75); the matrix
 * lives in HBM once quantized.  transpose and both mixed-precision mvm variants (4-bit x 8-bit with
 * CloverVector8 operands, 4-bit x fp32) are here as well.
 */
#ifndef CLOVER_MATRIX4_H
#define CLOVER_MATRIX4_H

#include <cmath>
#include <iomanip>
#include <sstream>
#include <string>

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
    /* cacheGemmOperand(): the FP6 image the GEMM kernel streams (clm4_gemm_prepare), kept between gemm() calls while the device
     * copy of this matrix does not change (Mirror::device_version) */
    bool gemm_cache_on;
    mutable clm4_gemm_operand *gemm_image;
    mutable uint64_t gemm_image_version;

    const int8_t *dev_values() const { return reinterpret_cast<const int8_t *>(mem.dev_ro()); }
    const float *dev_scales() const { return reinterpret_cast<const float *>(mem.dev_ro() + value_bytes); }

public:
    CloverMatrix4(uint64_t h, uint64_t w)
        : rows(clover_hip::round_up(h, CLOVER_VECTOR_SIZE_PAD)), cols(clover_hip::round_up(w, CLOVER_VECTOR_SIZE_PAD)), gemm_cache_on(false),
          gemm_image(nullptr), gemm_image_version(0)
    {
        value_bytes = rows * cols / 2;
        mem.allocate(value_bytes + (rows >> 6) * (cols >> 6) * sizeof(float));
    }

    uint64_t getRows() const { return rows; }
    uint64_t getCols() const { return cols; }
    uint64_t size() const { return rows * cols; }
    uint64_t getBitsLength() const { return 4; }
    uint64_t getBytes() const { return value_bytes + (rows >> 6) * (cols >> 6) * sizeof(float); }

    /* Explicit residency (clover_device.h, -DCLOVER_HIP_EXPLICIT_SYNC): move the bytes NOW instead of at the next use.  toDevice(): upload
     * if the host copy is the newer one; toHost(): bring a device result back.  Optional in every build (the default build's page tracking
     * and all accessors synchronise by themselves); not in the reference, which has one copy. */
    void toDevice() const { (void)mem.dev_ro(); }
    void toHost() const { (void)mem.host_ro(); }
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

    /* CloverMatrix4.h:141-163: the restored elements row by row, then the grid of tile scales */
    std::string toString() const
    {
        const uint8_t *h = mem.host_ro();
        const float *s = reinterpret_cast<const float *>(h + value_bytes);
        const uint64_t v_blocks = rows >> 6, h_blocks = cols >> 6;
        std::stringstream sout;
        for (uint64_t i = 0; i < rows; i++) {
            for (uint64_t j = 0; j < cols; j++) sout << std::setw(7) << std::fixed << std::setprecision(2) << get(i, j) << " ";
            sout << ";" << std::endl;
        }
        for (uint64_t i = 0; i < v_blocks; i++) {
            for (uint64_t j = 0; j < h_blocks; j++) sout << std::setw(7) << std::fixed << std::setprecision(2) << s[i * h_blocks + j] << " ";
            sout << ";" << std::endl;
        }
        return sout.str();
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
    /* the reference's scalar twin (CloverMatrix4.h:178-264), on the host: tiles column-block outer, maximum over the 64 x 64 tile */
    void quantize_scalar(const CloverMatrix32 &m)
    {
        if (m.getRows() != rows || m.getCols() != cols) {
            std::cout << "Matrices do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        const float *u = m.host_ro();
        uint8_t *h = mem.host_rw();
        int8_t *r = reinterpret_cast<int8_t *>(h);
        float *sr = reinterpret_cast<float *>(h + value_bytes);
        const uint64_t hb = cols >> 6, vb = rows >> 6;
        for (uint64_t bj = 0; bj < hb; bj++)
            for (uint64_t bi = 0; bi < vb; bi++) {
                const uint64_t off = (bi << 6) * cols + (bj << 6);
                float mx = 0.0f;
                for (uint64_t i = 0; i < 64; i++)
                    for (uint64_t j = 0; j < 64; j++) { const float a = std::fabs(u[off + i * cols + j]); if (a > mx) mx = a; }
                if (mx == 0.0f) mx = 1.0f;                         /* the SIMD contract (:598-603); the scalar code divides by zero */
                sr[bi * hb + bj] = mx;
                const float k = 7.0f / mx;
                for (uint64_t i = 0; i < 64; i++)
                    for (uint64_t j = 0; j < 64; j += 2) {
                        const uint64_t idx = off + i * cols + j;
                        r[idx >> 1] = (int8_t)((clover_hip::scalar::quant1(u[idx], k) << 4) | (clover_hip::scalar::quant1(u[idx + 1], k) & 0xF));
                    }
            }
    }

    /* CloverMatrix4.h:266-301: the reference has only the scalar variant.  restore() is the kernel, restore_scalar() the host loop. */
    void restore(CloverMatrix32 &other) const
    {
        if (other.getRows() != rows || other.getCols() != cols) {
            std::cout << "Matrices do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        clover_hip::check(clm4_restore(dev_values(), dev_scales(), rows, cols, other.device_wo(), nullptr), "CloverMatrix4::restore");
    }
    void restore_scalar(CloverMatrix32 &other) const
    {
        if (other.getRows() != rows || other.getCols() != cols) {
            std::cout << "Matrices do not have the same size. Exiting ..." << std::endl;
            exit(1);
        }
        float *out = other.host_rw();
        for (uint64_t i = 0; i < rows; i++)
            for (uint64_t j = 0; j < cols; j++) out[i * cols + j] = get(i, j);
    }

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
    /* The reference's mvm_scalar (:311-392): every row wrapped in a non-owning CloverVector4 view over the matrix's own memory and
     * multiplied with dot() (the SIMD order, here the exact-order kernel), then 64 results at a time quantised by scalar code.  An
     * implementation independent of the mvm kernel: the validation partner of mvm / mvm_parallel. */
    void mvm_scalar(const CloverVector4 &productVector, CloverVector4 &resultVector)
    {
        if (productVector.size() != getCols() || resultVector.size_pad() != getRows()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        int8_t *vals = getData();
        float *scs = getScales();
        int8_t *r = resultVector.getData();
        float *sr = resultVector.getScales();
        const uint64_t hb = cols >> 6;
        for (uint64_t bi = 0; bi < (rows >> 6); bi++) {
            float block[64];
            for (uint64_t i = 0; i < 64; i++) {
                CloverVector4 rowVector(cols, vals + (((bi << 6) + i) * cols >> 1), scs + bi * hb);
                block[i] = rowVector.dot(productVector);
            }
            sr[bi] = clover_hip::scalar::quantize_block4(block, r + 32 * bi);
        }
    }

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
    /* The WHOLE quantized IHT / GD loop of 01_measure.h:923-946, 999-1021 with this matrix as Phi, in one call (clm4_iht): x.clear(), then
     * `iterations` times t1 = Phi x; t2 = y - t1; t3 = PhiT t2; x += mu t3; [threshold(K)].  One persistent launch with Phi and PhiT in
     * LDS when the problem qualifies (clover_hip.h), else the launch-per-step loop; same bits as the five method calls.  Deterministic
     * rounding only (each step of a stochastic loop draws from its own object's generator: CloverIHT.h keeps the calls apart there). */
    void iht_loop(CloverMatrix4 &PhiT, CloverVector4 &x, const CloverVector4 &y, CloverVector4 &t1, CloverVector4 &t2, CloverVector4 &t3,
                  uint64_t iterations, uint64_t K, float mu, bool with_threshold)
    {
        if (PhiT.getRows() != getCols() || PhiT.getCols() != getRows() || x.size_pad() != getCols() || y.size_pad() != getRows() ||
            t1.size_pad() != getRows() || t2.size_pad() != getRows() || t3.size_pad() != getCols()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        const int thr = !with_threshold ? 0 : (clover_hip::threshold_mode() == CLV_THRESHOLD_FAST ? 1 : 2);
        clover_hip::check(clm4_iht(dev_values(), dev_scales(), PhiT.dev_values(), PhiT.dev_scales(), rows, cols, x.dev_values_wo(), x.dev_scales_wo(),
                                   x.size(), y.dev_values_ro(), y.dev_scales_ro(), t1.dev_values_wo(), t1.dev_scales_wo(), t2.dev_values_wo(),
                                   t2.dev_scales_wo(), t3.dev_values_wo(), t3.dev_scales_wo(), iterations, K, mu, thr, nullptr, nullptr),
                          "CloverMatrix4::iht_loop");
        x.commit();
        if (iterations) { t1.commit(); t2.commit(); t3.commit(); }
    }
    /* the same loop with CloverVector8 vectors (clm4_iht_v8): the reference's published "4-bit" IHT / GD configuration (02_bit04.cpp:140) */
    void iht_loop(CloverMatrix4 &PhiT, CloverVector8 &x, const CloverVector8 &y, CloverVector8 &t1, CloverVector8 &t2, CloverVector8 &t3,
                  uint64_t iterations, uint64_t K, float mu, bool with_threshold)
    {
        if (PhiT.getRows() != getCols() || PhiT.getCols() != getRows() || x.size_pad() != getCols() || y.size_pad() != getRows() ||
            t1.size_pad() != getRows() || t2.size_pad() != getRows() || t3.size_pad() != getCols()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        const int thr = !with_threshold ? 0 : (clover_hip::threshold_mode() == CLV_THRESHOLD_FAST ? 1 : 2);
        clover_hip::check(clm4_iht_v8(dev_values(), dev_scales(), PhiT.dev_values(), PhiT.dev_scales(), rows, cols, x.dev_values_wo(), x.dev_scales_wo(),
                                      x.size(), y.dev_values_ro(), y.dev_scales_ro(), t1.dev_values_wo(), t1.dev_scales_wo(), t2.dev_values_wo(),
                                      t2.dev_scales_wo(), t3.dev_values_wo(), t3.dev_scales_wo(), iterations, K, mu, thr, nullptr, nullptr),
                          "CloverMatrix4::iht_loop");
        x.commit();
        if (iterations) { t1.commit(); t2.commit(); t3.commit(); }
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
    /* :402-413: double accumulation of get(i, j) * x.get(j), cast to float, then the 8-bit quantiser (the kernel one, as there) */
    void mvm_scalar(const CloverVector8 &productVector, CloverVector8 &resultVector)
    {
        if (productVector.size() != getCols() || resultVector.size_pad() != getRows()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        CloverVector32 resultVector32(rows);
        for (uint64_t i = 0; i < rows; i++) {
            double sum = 0;
            for (uint64_t j = 0; j < cols; j++) sum += (double)get(i, j) * (double)productVector.get(j);
            resultVector32.set(i, (float)sum);
        }
        resultVector.quantize(resultVector32);
    }
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
    /* :423-432: double accumulation on the host */
    void mvm_scalar(const CloverVector32 &productVector, CloverVector32 &resultVector)
    {
        if (productVector.size() != getCols() || resultVector.size_pad() != getRows()) {
            std::cout << "MVM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        for (uint64_t i = 0; i < rows; i++) {
            double sum = 0;
            for (uint64_t j = 0; j < cols; j++) sum += (double)get(i, j) * (double)productVector.get(j);
            resultVector.set(i, (float)sum);
        }
    }

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
    /* :435-502, element by element on the host */
    void transpose_scalar(CloverMatrix4 &other) const
    {
        if (other.rows != cols || other.cols != rows) {
            std::cout << "Matrix can not be transposed. Exiting ..." << std::endl;
            exit(1);
        }
        const uint8_t *h = mem.host_ro();
        uint8_t *o = other.mem.host_rw();
        const float *s = reinterpret_cast<const float *>(h + value_bytes);
        float *so = reinterpret_cast<float *>(o + other.value_bytes);
        for (uint64_t i = 0; i < rows; i++)
            for (uint64_t j = 0; j < cols; j++) {
                const uint64_t src = i * cols + j, dst = j * rows + i;
                const uint8_t nib = (src & 1) ? (h[src >> 1] & 0xF) : (h[src >> 1] >> 4);
                o[dst >> 1] = (dst & 1) ? (uint8_t)((o[dst >> 1] & 0xF0) | nib) : (uint8_t)((o[dst >> 1] & 0x0F) | (nib << 4));
            }
        for (uint64_t bi = 0; bi < (rows >> 6); bi++)
            for (uint64_t bj = 0; bj < (cols >> 6); bj++) so[bj * (rows >> 6) + bi] = s[bi * (cols >> 6) + bj];
    }
    void transpose_scalar_faster(CloverMatrix4 &other) const { transpose_scalar(other); }      /* CloverMatrix4.h:2649-2799 */

    /* C = this * B^T, fp32: this is M x K, B is N x K, C is M x N (build-defined) */
    void gemm(const CloverMatrix4 &B, CloverMatrix32 &C) const
    {
        if (B.cols != cols || C.getRows() != rows || C.getCols() != B.rows) {
            std::cout << "GEMM can not be performed. Exiting ..." << std::endl;
            exit(1);
        }
        const int8_t *qa = dev_values(), *qb = B.dev_values();            /* uploads pending host writes: versions are final after this */
        const clm4_gemm_operand *opA = gemm_operand(qa), *opB = B.gemm_operand(qb);
        if (opA || opB)
            /* the nibbles go along with the images: two cached operands of very different size hold images in different staging
             * layouts, and the call then re-codes the smaller one (clover_hip.h) */
            clover_hip::check(clm4_gemm_prepared(opA, qa, dev_scales(), rows, cols, opB, qb, B.dev_scales(), B.rows,
                                                 C.device_wo(), nullptr), "CloverMatrix4::gemm");
        else
            clover_hip::check(clm4_gemm(qa, dev_scales(), rows, cols, qb, B.dev_scales(), B.rows, C.device_wo(), nullptr), "CloverMatrix4::gemm");
    }
    /* A matrix that is multiplied many times (weights): keep the FP6 image the GEMM kernel streams between gemm() calls instead
     * of re-coding the nibbles on every call (+3/4 of the matrix's size in HBM; 8192^3: 0.41 -> 0.38 ms per call).  The image
     * follows the matrix: any change of its contents -- quantize(), a transpose into it, a write through getData() -- is seen at
     * the next gemm(), which then re-codes once.  Results are those of the uncached call, bit for bit. */
    void cacheGemmOperand(bool on = true)
    {
        gemm_cache_on = on;
        if (!on) drop_gemm_image();
    }
    bool gemmOperandCached() const { return gemm_image != nullptr && gemm_image_version == mem.device_version(); }
    ~CloverMatrix4() { drop_gemm_image(); }

private:
    void drop_gemm_image() const
    {
        if (gemm_image) clover_hip::check(clm4_gemm_release(gemm_image), "CloverMatrix4: releasing the cached GEMM operand");
        gemm_image = nullptr;
    }
    /* the cached image if caching is on (re-coded now if the matrix changed since), else NULL; q = dev_values() */
    const clm4_gemm_operand *gemm_operand(const int8_t *q) const
    {
        if (!gemm_cache_on) return nullptr;
        if (gemm_image && gemm_image_version == mem.device_version()) return gemm_image;
        drop_gemm_image();
        clover_hip::check(clm4_gemm_prepare(q, rows, cols, &gemm_image, nullptr), "CloverMatrix4::cacheGemmOperand");
        gemm_image_version = mem.device_version();
        return gemm_image;
    }
};

#endif
