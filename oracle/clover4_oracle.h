/*
 * clover4_oracle.h -- CPU restatement of Clover's 4-bit hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X backend: a plain-C, scalar restatement of the arithmetic that
 * the reference's AVX2 path performs for CloverVector4 / CloverMatrix4 (quantize, restore, dot, mvm) plus
 * the build-defined GEMM.  It is NOT part of the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (libclover_hip.so) never links or calls it.
 *
 * Pinning status: the reference cannot be built in this image (include/CloverBase.h:35-36 include ipp.h
 * and mkl.h unconditionally; neither exists here and stand-ins are not allowed), so the oracle is pinned
 * against the known-answer vectors captured from the reference's SIMD path that SURVEY.md Appendix D
 * records (KAT1..KAT3: quantize bytes + scales, dot bit patterns, restore bit patterns, matrix quantize +
 * mvm bytes + scales) -- see tests/test_oracle_kat.py.  The generator (a7) IS pinned on the reference: its
 * header include/simdxorshift128plus.h builds standalone (oracle/ref_xorshift.cpp -> oracle/_ref/), and
 * tests/golden/xorshift_ref.json (init lanes, 256 draws, 2^20-draw digest, jump) is generated from it by
 * `make -C oracle ref-fixtures` -- see tests/test_xorshift_ref.py.  What remains "parity unpinned" is the
 * noise DERIVATION from those draws and its lane maps (CloverVector4.h:690-734, :1236-1243; CloverMatrix4.h:
 * 925-932), which live inside the unbuildable container headers: restated from the source only.
 *
 * All sizes are PADDED sizes (vector: multiple of 128 elements; matrix: rows, cols multiples of 128),
 * exactly as the reference's containers hold them (include/CloverVector.h:86-92, CloverMatrix.h:48-53).
 * Packing: element 2i is the HIGH nibble of byte i, element 2i+1 the LOW nibble, two's complement
 * (include/CloverVector4.h:511-514).  Scales are the block absolute maximum (0 -> 1.0), one per 64
 * elements (vector) or per 64x64 tile (matrix, row-major tile grid) (CloverVector4.h:661-673,
 * CloverMatrix4.h:598-603).  Quantized nibbles are always in [-7,7]; -8 is outside the contract.
 */
#ifndef CLOVER4_ORACLE_H
#define CLOVER4_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* XORShift state as the reference keeps it: 4 lanes of (s0,s1) (simdxorshift128plus.h:73-92). */
typedef struct { uint64_t s0[4]; uint64_t s1[4]; } orc_rng;

/* avx_xorshift128plus_init (simdxorshift128plus.h:47-92): lane0=(k1,k2), lanes 1..3 = 2^64-step jumps. */
void     orc_rng_init(orc_rng *r, uint64_t key1, uint64_t key2);
/* One avx_xorshift128plus draw (simdxorshift128plus.h:97-109): 8 x uint32, W[2k]=lo32(lane k). */
void     orc_rng_draw(orc_rng *r, uint32_t W[8]);
/* avx_xorshift128plus_jump (simdxorshift128plus.h:115-127). */
void     orc_rng_jump(orc_rng *r);
/* count draws -> xor-fold and wrapping sum of the 64-bit lane outputs (fixture: tests/golden/xorshift_ref.json) */
void     orc_rng_digest(orc_rng *r, uint64_t count, uint64_t *xor_fold, uint64_t *sum);
/* The 64 noises of one block (two draws), indexed [group g=0..7][lane j=0..7] (CloverVector4.h:690-734). */
void     orc_rng_block_noise(orc_rng *r, float noise[8][8]);

/* CloverVector4::quantize (CloverVector4.h:605-807). rng==NULL <=> CLOVER_STOCHASTIC_ROUNDING_DISABLED. */
void     orc_v4_quantize(const float *x, uint64_t n_pad, uint8_t *q, float *s, orc_rng *rng);
/* CloverVector4::restore (CloverVector4.h:1027-1093). */
void     orc_v4_restore(const uint8_t *q, const float *s, uint64_t n_pad, float *x);
/* CloverVector4::get (CloverVector4.h:179-188). */
float    orc_v4_get(const uint8_t *q, const float *s, uint64_t pos);
/* Exact per-(block, 32-bit word) integer sums I[b*8+w] (CloverVector4.h:1140-1180). */
void     orc_v4_word_isums(const uint8_t *qu, const uint8_t *qv, uint64_t n_pad, int32_t *I);
/* CloverVector4::dot -- SIMD lane/accumulator order (CloverVector4.h:1095-1192, CloverBase.h:149-157). */
float    orc_v4_dot(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad);
/* CloverVector4::dot_scalar (CloverVector4.h:555-595) -- the reference's own tolerance partner. */
float    orc_v4_dot_scalar(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad);
/* Order-free fp64 evaluation of the same sum (tolerance anchor for dot_fast / gemm). */
double   orc_v4_dot_f64(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, uint64_t n_pad);

/* CloverMatrix4::quantize (CloverMatrix4.h:512-766). Tile order for the rng: column-block outer. */
void     orc_m4_quantize(const float *A, uint64_t rows, uint64_t cols, uint8_t *q, float *s, orc_rng *rng);
/* CloverMatrix4::get (CloverMatrix4.h:123-139). */
/* CloverMatrix4::restore_scalar (CloverMatrix4.h:266-301): A[i][j] = f32(s_tile / 7) * q */
void     orc_m4_restore(const uint8_t *q, const float *s, uint64_t rows, uint64_t cols, float *A);
float    orc_m4_get(const uint8_t *q, const float *s, uint64_t rows, uint64_t cols, uint64_t i, uint64_t j);
/* fp32 row dots of mvm before the re-quantisation (CloverMatrix4.h:804-916): d[r], r=0..rows-1. */
void     orc_m4_rowdots(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols,
                        const uint8_t *x, const float *sx, float *d);
/* CloverMatrix4::mvm(CloverVector4, CloverVector4&) (CloverMatrix4.h:777-1083). */
void     orc_m4_mvm(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols,
                    const uint8_t *x, const float *sx, uint8_t *r, float *sr, orc_rng *rng);
/* Re-quantisation of 64 row dots (mvm epilogue, CloverMatrix4.h:919-1080); noise lane map 8j+g. */
void     orc_m4_requantize64(const float d[64], uint8_t r[32], float *sr, orc_rng *rng);

/* ---- section 8(f) "next" rows ------------------------------------------------------------------------ */
/* CloverVector4::scaleAndAdd (CloverVector4.h:1196-1478): r = quantize(u + a*v), block by block:
 * val = fma((float)qv, f32(f32(sv*a)/7), (float)qu * f32(su/7)); then the quantiser.  r may alias u.
 * Noise lane map differs from quantize: group g, lane j lands on element 8j + (g^1) (:1236-1243, 1459-1466). */
void     orc_v4_scale_and_add(const uint8_t *qu, const float *su, const uint8_t *qv, const float *sv, float a,
                              uint64_t n_pad, uint8_t *r, float *sr, orc_rng *rng);
/* CloverMatrix4::transpose (CloverMatrix4.h:1549-1663; scalar :435-502): out(j,i) = in(i,j), scales too. */
void     orc_m4_transpose(const uint8_t *q, const float *s, uint64_t rows, uint64_t cols, uint8_t *qt, float *st);
/* CloverVector4::threshold(K) (CloverVector4.h:1913-1975): keeps the K largest |value| of the first n
 * elements with the reference's min-heap walk (std::make_heap + min_heapify, CloverBase.h:208-249), zeroes
 * the other nibbles; scales untouched.  Which of several EQUAL magnitudes survive depends on heap order. */
void     orc_v4_threshold(uint8_t *q, const float *s, uint64_t n, uint64_t k);

/* Mixed precision CloverMatrix4::mvm(const CloverVector32&, CloverVector32&) (CloverMatrix4.h:1451-1547):
 * fp32 vector in, fp32 result out (no re-quantisation).  Per row 4 accumulators x 8 AVX lanes = 32 sequential
 * fma chains, chain (e mod 32) takes elements e, e+32, ...: acc = fma(x[e], f32((float)q * f32(s/7)), acc);
 * then (acc1+acc2)+(acc3+acc4) per lane and the CloverBase.h:149-157 tree. */
void     orc_m4_mvm_f32(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const float *x, float *r);

/* ---- mixed precision, 4-bit matrix x 8-bit vector ---------------------------------------------
 * CloverVector8 (CloverVector8.h:35-140): int8 values in natural order + one fp32 scale per 64 elements,
 * value = q * scale / 127. */
/* CloverVector8::quantize (CloverVector8.h:393-606; scalar :205-253): block max (0 -> 1.0), k = 127/max,
 * q = sign(x) * trunc(fma(|x|, k, noise)); noise group g = element/8 (draw g>>2, byte g&3), word W[element%8] */
void     orc_v8_quantize(const float *x, uint64_t n_pad, int8_t *q, float *s, orc_rng *rng);
/* CloverVector8::restore (CloverVector8.h:835-909): x = (float)q * (scale / 127.0f) */
void     orc_v8_restore(const int8_t *q, const float *s, uint64_t n_pad, float *x);
/* CloverVector8::dot (CloverVector8.h:911-977: 8 sequential fma chains), dot_scalar (:268-310), and the order-free fp64 value */
float    orc_v8_dot(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad);
float    orc_v8_dot_scalar(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad);
double   orc_v8_dot_f64(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad);
/* fp32 row dots of CloverMatrix4::mvm(const CloverVector8 &, CloverVector8 &) (CloverMatrix4.h:1093-1243) in SIMD order:
 * 8 fp32 fma chains per row (chain L = elements 4L..4L+3 and 32+4L..32+4L+3 of every 64-block),
 * c_b = f32(f32(sA*1/7) * f32(sx*1/127)), chain += c_b * (float)I exactly-summed, then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)) */
void     orc_m4_rowdots_v8(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, float *d);
/* the whole mixed mvm: row dots, then 64 at a time re-quantised to 8 bits (CloverMatrix4.h:1246-1440);
 * noise group g, word j lands on output row 8g + j */
void     orc_m4_mvm_v8(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                       int8_t *r, float *sr, orc_rng *rng);
/* CloverVector8::scaleAndAdd (CloverVector8.h:1063-1358): r = quantize8(u + a*v) per 64-block,
 * val = fma((float)qv, f32(f32(sv*a)/127), (float)qu * f32(su/127)); noise for element e: draw e>>5, byte e&3, word (e&31)>>2 */
void     orc_v8_scale_and_add(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, float a, uint64_t n_pad,
                              int8_t *r, float *sr, orc_rng *rng);
/* CloverVector8::threshold (CloverVector8.h:1680-1740): min-heap top-K on |q * scale / 127| over the first n elements */
void     orc_v8_threshold(int8_t *q, const float *s, uint64_t n, uint64_t k);
/* mvm_scalar(CloverVector8) (CloverMatrix4.h:402-413): double accumulation of get(i,j) * x.get(j), cast to float */
void     orc_m4_rowdots_v8_f64(const uint8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, float *d);

/*
 * GEMM -- no reference function exists; build-defined semantics:
 *   A is M x K, B is N x K (both CloverMatrix4 layouts), C = A * B^T, fp32, row-major M x N.
 *   S[i][j][b] = sum over the 64 nibble products of K-block b   (exact int32)
 *   c[b]       = f32(f32(sA[(i>>6)*(K/64)+b] * (1/49)) * sB[(j>>6)*(K/64)+b])
 *   C[i][j]    = fold over b = 0,1,2,... of  C = fmaf(c[b], (float)S[i][j][b], C),  C0 = 0
 * i.e. ONE sequential fma chain per output element (the MFMA-friendly order), not dot()'s 16 chains.
 */
void     orc_m4_gemm(const uint8_t *A, const float *sA, uint64_t M, uint64_t K,
                     const uint8_t *B, const float *sB, uint64_t N, float *C);
/* Exact K-block integer sums S[(i*N + j)*(K/64) + b] for small cases. */
void     orc_m4_gemm_isums(const uint8_t *A, uint64_t M, uint64_t K, const uint8_t *B, uint64_t N, int32_t *S);

#ifdef __cplusplus
}
#endif
#endif
