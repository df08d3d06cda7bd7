// matrix4.hip -- CloverMatrix4 hot path on gfx950: quantize (64x64 tiles) and mvm (GEMV + re-quantise).
//
// HBM layout = the reference's (CloverMatrix4.h:77-93, 123-139): row-major nibbles (rows*cols/2 bytes),
// then one fp32 scale per 64x64 tile in a row-major (rows/64) x (cols/64) grid.
#include "rng_device.h"

#include <stdlib.h>
#include <string.h>

// ================================================================================================
// mvm  (CloverMatrix4.h:777-1083)
// ================================================================================================
//
// Bit-exactness fixes the arithmetic: every output row owns 16 sequential fp32 fma chains, chain
// j = (32-bit word index within the row) mod 16 -- the reference's 2 accumulators x 8 AVX lanes -- and
// a fixed add tree at the end.  Mapping used here ("chain per lane"):
//   workgroup  = one 64-row output block (256 threads), so the re-quantise epilogue is fused;
//   lane       = (row rho = tid>>2, quarter q = tid&3): it owns chains 4q..4q+3 of its row and walks
//                the row 64 B (one block pair) per step, loading its 16 B with one dwordx4;
//   x          = staged once per 65536-column chunk in LDS (32 KiB) together with the per-block factor
//                c[b] = f32(f32(sA[b] * 1/49) * sx[b]) (4 KiB), which is identical for all 64 rows;
//   v_dot8_i32_i4 yields the exact integer of each word, v_cvt + v_fma continue the chain.
// Algorithmic bytes per call: rows*cols/2 + 4*(rows/64)*(cols/64) + 0.5625*(rows + cols)  (SURVEY 8(d)).

#define MVM_CHUNK 65536u   // columns staged in LDS per pass

template <int U, bool NT>
__device__ __forceinline__ void mvm_steps(const u32x4 *__restrict__ Ap, const u32x4 *xs, const float *cs, int q,
                                          uint32_t t0, float &a0, float &a1, float &a2, float &a3)
{
    u32x4 a[U];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = NT ? __builtin_nontemporal_load(&Ap[4 * (t0 + u) + q]) : Ap[4 * (t0 + u) + q];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const u32x4 xv = xs[4 * (t0 + u) + q];
        const float c = cs[2 * (t0 + u) + (q >> 1)];
        a0 = __builtin_fmaf(c, (float)sdot8(a[u].x, xv.x, 0), a0);
        a1 = __builtin_fmaf(c, (float)sdot8(a[u].y, xv.y, 0), a1);
        a2 = __builtin_fmaf(c, (float)sdot8(a[u].z, xv.z, 0), a2);
        a3 = __builtin_fmaf(c, (float)sdot8(a[u].w, xv.w, 0), a3);
    }
}

// 8 lanes per row (lane e owns chains 2e, 2e+1 and loads 8 bytes per step): twice the waves per workgroup, for
// matrices with too few 64-row groups to fill the chip with 4-wave workgroups
template <int U, bool NT>
__device__ __forceinline__ void mvm_steps8(const u32x2 *__restrict__ Ap, const u32x2 *xs, const float *cs, int e, uint32_t t0, float &a0,
                                           float &a1)
{
    u32x2 a[U];
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = NT ? __builtin_nontemporal_load(&Ap[8 * (t0 + u) + e]) : Ap[8 * (t0 + u) + e];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const u32x2 xv = xs[8 * (t0 + u) + e];
        const float c = cs[2 * (t0 + u) + (e >> 2)];
        a0 = __builtin_fmaf(c, (float)sdot8(a[u].x, xv.x, 0), a0);
        a1 = __builtin_fmaf(c, (float)sdot8(a[u].y, xv.y, 0), a1);
    }
}

// re-quantise 64 values held one per lane of a full wave (CloverMatrix4.h:919-1080); returns this lane's nibble value,
// *scale = the block maximum.  r_words / sr may be NULL (result not stored).
__device__ __forceinline__ int requantize_wave(float d, float noise, uint32_t *r_words, float *sr, float *scale)
{
    const int lane = threadIdx.x & 63;
    float m = wave_max(__builtin_fabsf(d));
    m = fix_zero_max(m);
    const float k = 7.0f / m;
    const int qv = quant1(d, k, noise);
    if (r_words) {
        uint32_t w = ((uint32_t)qv & 0xFu) << nib_shift(lane & 7);
        w |= __shfl_xor(w, 1);
        w |= __shfl_xor(w, 2);
        w |= __shfl_xor(w, 4);
        if ((lane & 7) == 0) r_words[lane >> 3] = w;
        if (lane == 0) *sr = m;
    }
    *scale = m;
    return qv;
}

// FUSE: the scaleAndAdd that follows mvm in the IHT / GD loops (t2 = y - Phi x;  x += mu Phi' t2), done on the row
// group while it is still in the wave:  r2 = quantize(u + a * quantize(A x))  (CloverVector4.h:1196-1478).
struct MvmFuse {
    const uint32_t *qu;      // u, one 64-element block per row group
    const float *su;
    float a;
    uint32_t *r2;            // may alias qu (the in-place overload)
    float *sr2;
};

// ST: stochastic re-quantisation fused into the epilogue (CloverMatrix4.h:919-1080 with the rnd_* branch).  Row group
// rb consumes draws 2rb, 2rb+1 of the stream; the dots sit pre-transposed in the reference's block_values, so noise
// group g of AVX lane j lands on row 8j+g: row l uses group g = l&7 (draw g>>2, byte g&3) of word W[l>>3].
// With FUSE the scaleAndAdd draws follow ALL the mvm draws in the stream, as in the two separate calls: row group rb
// then uses draws 2G + 2rb, 2G + 2rb + 1 (G = number of row groups), with the scaleAndAdd lane map 8j + (g ^ 1).
template <int U, bool NT, bool ST, bool FUSE, int L = 4>
__global__ __launch_bounds__(64 * L) void k_m4_mvm64(const uint8_t *__restrict__ A, const float *__restrict__ sA,
                                                  uint64_t cols, const uint8_t *__restrict__ x, const float *__restrict__ sx,
                                                  float *__restrict__ d_out, uint32_t *r, float *sr,
                                                  uint64_t *rng_state, uint64_t seq, const uint64_t *__restrict__ pow_rows, MvmFuse fuse)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *xs = reinterpret_cast<u32x4 *>(smem);                        // MVM_CHUNK/2 bytes
    float *cs = reinterpret_cast<float *>(smem + MVM_CHUNK / 2);        // MVM_CHUNK/64 floats
    float *dsh = cs + MVM_CHUNK / 64;                                   // 64 floats
    uint64_t *rbase = reinterpret_cast<uint64_t *>(dsh + 64);           // ST: 4 lane bases, 8 raw draws (twice with FUSE)
    uint64_t *raw = rbase + 4;
    uint64_t *rbase2 = raw + 8, *raw2 = rbase2 + 4;
    if (ST) {
        const uint64_t a0 = rng_workgroup_begin(rng_state, seq, pow_rows, blockIdx.x, 1, (FUSE ? 4ull : 2ull) * gridDim.x, rbase);
        if (FUSE) {
            const uint64_t b2 = wave_pow_apply(pow_rows, a0, (uint64_t)gridDim.x + blockIdx.x, 1);
            if ((threadIdx.x & 63) == 0) rbase2[threadIdx.x >> 6] = b2;      // read in the epilogue, barriers in between
        }
    }
    // FUSE: this row group's block of u, fetched now so that the epilogue does not wait for it
    uint32_t fuse_w = 0;
    float fuse_s = 0.0f;
    if (FUSE && threadIdx.x < 64) {
        fuse_w = fuse.qu[blockIdx.x * 8 + (threadIdx.x >> 3)];
        fuse_s = fuse.su[blockIdx.x];
    }

    static_assert(L == 4 || (L == 8 && !ST), "the stochastic epilogue assumes 4 waves");
    constexpr int THREADS = 64 * L;
    const uint64_t rb = blockIdx.x;
    const int tid = threadIdx.x;
    const int q = tid & (L - 1);                    // L == 4: quarter (chains 4q..4q+3);  L == 8: eighth (chains 2q, 2q+1)
    const int rho = tid / L;
    const uint64_t row = rb * 64 + rho;
    const u32x4 *Arow = reinterpret_cast<const u32x4 *>(A + row * (cols / 2));
    const float *sArow = sA + rb * (cols / 64);

    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;

    for (uint64_t c0 = 0; c0 < cols; c0 += MVM_CHUNK) {
        const uint32_t cw = (uint32_t)((cols - c0) < MVM_CHUNK ? (cols - c0) : MVM_CHUNK);
        if (c0) __syncthreads();
        // stage x and c[b]: all loads first (one round trip), then the LDS writes.  Written with guarded, fully
        // unrolled loads instead of a runtime-trip-count loop, which hipcc turns into load-wait-store chains.
        const u32x4 *xg = reinterpret_cast<const u32x4 *>(x + c0 / 2);
        {
            constexpr int NX = MVM_CHUNK / 32 / THREADS;      // 8 x 16 B per thread (L == 4)
            constexpr int NC = MVM_CHUNK / 64 / THREADS;      // 4 factors per thread
            u32x4 xr[NX];
            float sa[NC], sv[NC];
            const uint32_t nx = cw / 32, nc = cw / 64;
#pragma unroll
            for (int k = 0; k < NX; k++) { const uint32_t i = tid + THREADS * k; xr[k] = xg[i < nx ? i : 0]; }
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const uint32_t i = tid + THREADS * k, ii = i < nc ? i : 0;
                sa[k] = sArow[c0 / 64 + ii];
                sv[k] = sx[c0 / 64 + ii];
            }
#pragma unroll
            for (int k = 0; k < NX; k++) { const uint32_t i = tid + THREADS * k; if (i < nx) xs[i] = xr[k]; }
#pragma unroll
            for (int k = 0; k < NC; k++) { const uint32_t i = tid + THREADS * k; if (i < nc) cs[i] = (sa[k] * CLV_RCP49) * sv[k]; }
        }
        __syncthreads();

        // (a register double-buffered variant -- next U loads requested before the current U are consumed -- was
        //  measured slower at every size: it costs the fourth wave per SIMD; r01 microbench variants 5-7)
        const u32x4 *Ap = Arow + c0 / 32;
        const uint32_t npairs = cw / 128;
        uint32_t t = 0;
        if constexpr (L == 4) {
            for (; t + U <= npairs; t += U) mvm_steps<U, NT>(Ap, xs, cs, q, t, a0, a1, a2, a3);
            for (; t < npairs; t++) mvm_steps<1, NT>(Ap, xs, cs, q, t, a0, a1, a2, a3);
        } else {
            const u32x2 *Ap8 = reinterpret_cast<const u32x2 *>(Ap);
            const u32x2 *xs8 = reinterpret_cast<const u32x2 *>(xs);
            for (; t + U <= npairs; t += U) mvm_steps8<U, NT>(Ap8, xs8, cs, q, t, a0, a1);
            for (; t < npairs; t++) mvm_steps8<1, NT>(Ap8, xs8, cs, q, t, a0, a1);
        }
    }

    // chain (4q+i): accumulator a = q>>1, AVX lane w = 4(q&1)+i.  Fixed tree of CloverBase.h:149-157:
    const float v0 = a0 + __shfl_xor(a0, 2);     // acc[0][w] + acc[1][w]
    const float v1 = a1 + __shfl_xor(a1, 2);
    const float v2 = a2 + __shfl_xor(a2, 2);
    const float v3 = a3 + __shfl_xor(a3, 2);
    const float x0 = v0 + __shfl_xor(v0, 1);     // v[i+4] + v[i]
    const float x1 = v1 + __shfl_xor(v1, 1);
    const float x2 = v2 + __shfl_xor(v2, 1);
    const float x3 = v3 + __shfl_xor(v3, 1);
    float dot = (x0 + x2) + (x1 + x3);
    if constexpr (L == 8) {
        // chain 2q+i: accumulator a = q>>2, AVX lane w = 2(q&3)+i.  Same tree: acc[0][w]+acc[1][w] (lanes q, q^4), then
        // v[w]+v[w+4] (q, q^2) gives x[2(q&1)+i], then (x0+x2)+(x1+x3) (q, q^1)
        const float u0 = a0 + __shfl_xor(a0, 4), u1 = a1 + __shfl_xor(a1, 4);
        const float y0 = u0 + __shfl_xor(u0, 2), y1 = u1 + __shfl_xor(u1, 2);
        const float z0 = y0 + __shfl_xor(y0, 1), z1 = y1 + __shfl_xor(y1, 1);       // x0+x2, x1+x3
        dot = z0 + z1;
    }

    if (q == 0) {
        dsh[rho] = dot;
        if (d_out) d_out[row] = dot;
    }
    if (ST && tid < 4) {
        gen_blocks(rbase[tid], 1, raw, tid);
        if (FUSE) gen_blocks(rbase2[tid], 1, raw2, tid);
    }
    __syncthreads();
    if ((r || FUSE) && tid < 64) {
        float noise = 0.0f;
        if (ST) {
            const int grp = tid & 7, j = tid >> 3;
            noise = noise_of(reinterpret_cast<const uint32_t *>(raw + (size_t)(grp >> 2) * 4)[j], grp & 3);
        }
        float m;
        const int qv = requantize_wave(dsh[tid], noise, r ? r + rb * 8 : nullptr, r ? sr + rb : nullptr, &m);
        if (FUSE) {
            const float su7 = div7(fuse_s), sv7 = div7(m * fuse.a);
            const float val = __builtin_fmaf((float)qv, sv7, (float)unpack1(fuse_w, tid & 7) * su7);
            float noise2 = 0.0f;
            if (ST) {
                const int g = (tid & 7) ^ 1, j = tid >> 3;
                noise2 = noise_of(reinterpret_cast<const uint32_t *>(raw2 + (size_t)(g >> 2) * 4)[j], g & 3);
            }
            float m2;
            requantize_wave(val, noise2, fuse.r2 + rb * 8, fuse.sr2 + rb, &m2);
        }
    }
}

// ================================================================================================
// quantize  (CloverMatrix4.h:512-766, rounding disabled)
// ================================================================================================
// k_m4_quantize_strip (the one in use): workgroup = 64 rows x 256 columns = 4 tiles side by side (256 threads, 64 floats each).  A wave-instruction reads
// one contiguous KiB of a row (lane = float4); lanes 16t..16t+15 -- one DPP row -- belong to tile t, so the tile maximum
// is a per-lane maximum over the wave's 16 rows, a row rotation reduce and a 4-wave combine in LDS.  A lane quantises
// half a dword; lane pairs swap halves between two consecutive rows so that every lane stores a whole dword and a row
// receives 128 contiguous bytes.  Columns beyond `cols` (cols is a multiple of 128, not of 256) are masked.
__global__ __launch_bounds__(256) void k_m4_quantize_strip(const float *__restrict__ A, uint64_t cols, uint32_t *__restrict__ q,
                                                           float *__restrict__ s, uint32_t strips_x, uint32_t tiles_x)
{
    __shared__ float sh[4][4];
    const uint32_t sj = blockIdx.x % strips_x;
    const uint64_t bi = blockIdx.x / strips_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t col = (uint64_t)sj * 256 + 4 * lane;
    const bool live = col < cols;
    const uint64_t row0 = bi * 64 + wave * 16;

    f32x4 v[16];
    float m = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        v[r] = live ? __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(A + (row0 + r) * cols + col)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        m = fmaxf(m, fmaxf(fmaxf(__builtin_fabsf(v[r].x), __builtin_fabsf(v[r].y)), fmaxf(__builtin_fabsf(v[r].z), __builtin_fabsf(v[r].w))));
    }
    m = row16_max(m);
    if ((lane & 15) == 0) sh[wave][lane >> 4] = m;
    __syncthreads();
    const int t = lane >> 4;
    m = fix_zero_max(fmaxf(fmaxf(sh[0][t], sh[1][t]), fmaxf(sh[2][t], sh[3][t])));
    const float k = 7.0f / m;
    if (wave == 0 && (lane & 15) == 0 && live) s[bi * tiles_x + sj * 4 + t] = m;
    const int odd = lane & 1;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const uint32_t h0 = quant_pack4(v[r], k), h1 = quant_pack4(v[r + 1], k);
        const uint32_t give = odd ? h0 : h1;
        const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)give, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
        const uint32_t word = odd ? (recv | (h1 << 16)) : (h0 | (recv << 16));
        // even lanes hold the dword of row r, odd lanes that of row r+1; dword index inside the row = (col of the pair) / 8
        if (live) __builtin_nontemporal_store(word, &q[((row0 + r + odd) * cols + (col & ~7ull)) / 8]);
    }
}

// k_m4_quantize (the first kernel of the round, kept behind CLV_M4Q_TILE=1 for A/B): workgroup = one 64x64 tile (256 threads);
// thread = (row r = tid>>3 [+32 on the second pass], octet o = tid&7) holds 8 consecutive values = one output dword.  Pass 1
// reduces the tile maximum (registers -> wave shuffle -> LDS); pass 2 quantises from the registers, nothing is re-read.
__global__ __launch_bounds__(256) void k_m4_quantize(const float *__restrict__ A, uint64_t cols, uint32_t *__restrict__ q,
                                                     float *__restrict__ s, uint32_t tiles_x)
{
    __shared__ float sh[4];
    const uint32_t bj = blockIdx.x % tiles_x;
    const uint64_t bi = blockIdx.x / tiles_x;
    const int tid = threadIdx.x;
    const int o = tid & 7;
    const int r0 = tid >> 3;

    float v[2][8];
    float m = 0.0f;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint64_t row = bi * 64 + r0 + 32 * p;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(A + row * cols + bj * 64 + o * 8);
        const f32x4 lo = __builtin_nontemporal_load(&src[0]);
        const f32x4 hi = __builtin_nontemporal_load(&src[1]);
        v[p][0] = lo.x; v[p][1] = lo.y; v[p][2] = lo.z; v[p][3] = lo.w;
        v[p][4] = hi.x; v[p][5] = hi.y; v[p][6] = hi.z; v[p][7] = hi.w;
#pragma unroll
        for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[p][e]));
    }
    m = wave_max(m);
    if ((tid & 63) == 0) sh[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    m = fix_zero_max(m);
    const float k = 7.0f / m;
    if (tid == 0) s[bi * tiles_x + bj] = m;
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint64_t row = bi * 64 + r0 + 32 * p;
        q[(row * cols + bj * 64) / 8 + o] = quant_pack8(v[p], k, nullptr);
    }
}

// ================================================================================================
// GEMM, first (VALU) version: one thread per C element, exact K-block integer via 8 x v_dot8, one fma
// chain over K-blocks.  Semantics in oracle/clover4_oracle.h; the MFMA kernel replaces this for big shapes.
// ================================================================================================
__global__ __launch_bounds__(256) void k_m4_gemm_simple(const uint8_t *__restrict__ A, const float *__restrict__ sA, uint64_t M,
                                                        uint64_t K, const uint8_t *__restrict__ B, const float *__restrict__ sB,
                                                        uint64_t N, float *__restrict__ C)
{
    const uint64_t j = (uint64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const uint64_t i = (uint64_t)blockIdx.y * 16 + (threadIdx.x >> 4);
    if (i >= M || j >= N) return;
    const uint64_t kb = K / 64;
    const u32x4 *a = reinterpret_cast<const u32x4 *>(A + i * (K / 2));
    const u32x4 *b = reinterpret_cast<const u32x4 *>(B + j * (K / 2));
    const float *sa = sA + (i >> 6) * kb;
    const float *sb = sB + (j >> 6) * kb;
    float acc = 0.0f;
    for (uint64_t blk = 0; blk < kb; blk++) {
        const int S = dot32(a[2 * blk], b[2 * blk]) + dot32(a[2 * blk + 1], b[2 * blk + 1]);
        const float c = (sa[blk] * CLV_RCP49) * sb[blk];
        acc = __builtin_fmaf(c, (float)S, acc);
    }
    C[i * N + j] = acc;
}

// ================================================================================================
// restore  (CloverMatrix4::restore_scalar, CloverMatrix4.h:266-301):  A[i][j] = f32(s_tile / 7) * q
// lane = one output float4 (two lanes share an input dword), as in k_v4_restore; the scale comes from the 64x64 tile
// ================================================================================================
template <bool NT>
__global__ __launch_bounds__(256) void k_m4_restore(const uint32_t *__restrict__ q, const float *__restrict__ s, f32x4 *__restrict__ A,
                                                    uint64_t nquads, uint64_t cols)
{
    const uint64_t f0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 256;
    const int lane = threadIdx.x & 63;
    const uint64_t tiles_x = cols / 64;
    uint32_t wd[4];
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t f = f0 + 64 * j + lane, fc = f < nquads ? f : 0;
        const uint64_t e = fc * 4, row = e / cols, col = e - row * cols;
        wd[j] = q[fc >> 1];
        sc[j] = s[(row >> 6) * tiles_x + (col >> 6)];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t f = f0 + 64 * j + lane;
        const uint32_t hw = wd[j] >> (16 * (lane & 1));         // the 4 nibbles of this half (f and lane have the same parity)
        const float k = div7(sc[j]);
        f32x4 v;
        v.x = (float)(((int)(hw << 24)) >> 28) * k;
        v.y = (float)(((int)(hw << 28)) >> 28) * k;
        v.z = (float)(((int)(hw << 16)) >> 28) * k;
        v.w = (float)(((int)(hw << 20)) >> 28) * k;
        if (f < nquads) {
            if (NT) __builtin_nontemporal_store(v, &A[f]);
            else A[f] = v;
        }
    }
}

// ================================================================================================
// C ABI
// ================================================================================================
int clm4_quantize_stochastic(const float *A, uint64_t rows, uint64_t cols, int8_t *q, float *s, uint64_t *rng, hipStream_t st);
int clm4_gemm_mfma(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N, float *C, hipStream_t st);
int clm4_gemm_fp6(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N, float *C, hipStream_t st);

#define MVM_LDS_BYTES (MVM_CHUNK / 2 + (MVM_CHUNK / 64) * sizeof(float) + 64 * sizeof(float) + 256)

static int launch_mvm(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                      float *d, int8_t *r, float *sr, uint64_t *rng, hipStream_t st, const MvmFuse *fuse = nullptr)
{
    const size_t lds = MVM_LDS_BYTES;
    const dim3 grid((unsigned)(rows / 64)), block(256);
    RngTables T = {nullptr, nullptr, nullptr};
    uint64_t seq = 0;
    if (rng) {
        int rc = clv_rng_tables(&T);
        if (rc) return rc;
        seq = clv_rng_seq_for(rng, st);
    }
    const MvmFuse no_fuse = {nullptr, nullptr, 0.0f, nullptr, nullptr};
#define MVM_LAUNCH_F(NT, ST, FUSE)                                                                                               \
    hipLaunchKernelGGL((k_m4_mvm64<8, NT, ST, FUSE>), grid, block, lds, st, (const uint8_t *)A, sA, cols, (const uint8_t *)x, sx, d, \
                       (uint32_t *)r, sr, rng, seq, T.pow_rows, FUSE ? *fuse : no_fuse)
#define MVM_LAUNCH(NT, ST)                                  \
    do {                                                    \
        if (fuse) MVM_LAUNCH_F(NT, ST, true);               \
        else MVM_LAUNCH_F(NT, ST, false);                   \
    } while (0)
    // Streaming (nt) loads win once the matrix cannot live in the 256 MiB Infinity Cache (+14 % at 2 GiB); below
    // that, default-policy loads keep it cache-resident across calls (8192^2: 7.3 vs 11.9 us) -- measured, r01.
    const bool streaming = rows * (cols / 2) > (256ull << 20);
    // few row groups (<= 2 per CU) of a matrix that streams from HBM: 8 lanes per row double the waves in flight
    // (32768^2: 92 -> 89 us); measured no better anywhere else (r01 microbench variants 7-10)
    const bool wide = streaming && !rng && rows / 64 <= 2ull * (uint64_t)clv_cu_count();
    if (wide) {
        if (fuse)
            hipLaunchKernelGGL((k_m4_mvm64<16, true, false, true, 8>), grid, dim3(512), lds, st, (const uint8_t *)A, sA, cols,
                               (const uint8_t *)x, sx, d, (uint32_t *)r, sr, rng, seq, T.pow_rows, *fuse);
        else
            hipLaunchKernelGGL((k_m4_mvm64<16, true, false, false, 8>), grid, dim3(512), lds, st, (const uint8_t *)A, sA, cols,
                               (const uint8_t *)x, sx, d, (uint32_t *)r, sr, rng, seq, T.pow_rows, no_fuse);
    } else if (streaming) {
        if (rng) MVM_LAUNCH(true, true); else MVM_LAUNCH(true, false);
    } else {
        if (rng) MVM_LAUNCH(false, true); else MVM_LAUNCH(false, false);
    }
#undef MVM_LAUNCH
#undef MVM_LAUNCH_F
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

// ---- experiments: kernel variants for tools/microbench.py.  Compiled ONLY into the bench-only probe build (-DCLV_EXPERIMENTS,
// clover_amd/build.py build_probe_library -> tools/_build/libclover_hip_probe.so); the product library exports none of this ----------
#ifdef CLV_EXPERIMENTS
__global__ __launch_bounds__(256) void k_read_bw(const u32x4 *__restrict__ p, uint64_t n16, uint32_t *__restrict__ out, int nt)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = nt ? __builtin_nontemporal_load(&p[i + u * stride]) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const u32x4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;      // keeps the loads alive
}

extern "C" int clvx_read_bw(const void *p, uint64_t bytes, int nt, int blocks_per_cu, void *out, void *stream)
{
    hipLaunchKernelGGL(k_read_bw, dim3(clv_cu_count() * blocks_per_cu), dim3(256), 0, as_stream(stream), (const u32x4 *)p, bytes / 16,
                       (uint32_t *)out, nt);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clvx_mvm_variant(int variant, const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x,
                                const float *sx, int8_t *r, float *sr, void *stream)
{
    const size_t lds = MVM_LDS_BYTES;
    hipStream_t st = as_stream(stream);
    const dim3 grid((unsigned)(rows / 64)), block(256);
#define CLVX_LAUNCH(U, NT)                                                                                                      \
    hipLaunchKernelGGL((k_m4_mvm64<U, NT, false, false>), grid, block, lds, st, (const uint8_t *)A, sA, cols, (const uint8_t *)x, sx, \
                       (float *)nullptr, (uint32_t *)r, sr, (uint64_t *)nullptr, 0ull, (const uint64_t *)nullptr,                      \
                       MvmFuse{nullptr, nullptr, 0.0f, nullptr, nullptr})
    switch (variant) {
    case 0: CLVX_LAUNCH(8, true); break;
    case 1: CLVX_LAUNCH(8, false); break;
    case 2: CLVX_LAUNCH(4, true); break;
    case 3: CLVX_LAUNCH(16, true); break;
    case 4: CLVX_LAUNCH(2, true); break;
    case 5: CLVX_LAUNCH(16, false); break;
    case 6: CLVX_LAUNCH(32, false); break;
#define CLVX_LAUNCH8(U, NT)                                                                                                       \
    hipLaunchKernelGGL((k_m4_mvm64<U, NT, false, false, 8>), grid, dim3(512), lds, st, (const uint8_t *)A, sA, cols, (const uint8_t *)x, \
                       sx, (float *)nullptr, (uint32_t *)r, sr, (uint64_t *)nullptr, 0ull, (const uint64_t *)nullptr,                 \
                       MvmFuse{nullptr, nullptr, 0.0f, nullptr, nullptr})
    case 7: CLVX_LAUNCH8(8, false); break;
    case 8: CLVX_LAUNCH8(16, false); break;
    case 9: CLVX_LAUNCH8(8, true); break;
    case 10: CLVX_LAUNCH8(16, true); break;
#undef CLVX_LAUNCH8
    default: clv_set_error("clvx_mvm_variant: unknown variant %d", variant); return CLV_ERR_INVALID;
    }
#undef CLVX_LAUNCH
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
#endif      // CLV_EXPERIMENTS

static int check_mvm_args(const char *fn, const void *A, const void *sA, uint64_t rows, uint64_t cols, const void *x, const void *sx)
{
    CLV_REQUIRE(A && sA && x && sx, "%s: null pointer", fn);
    // a whole CloverMatrix4 has rows % 128 == 0 (CloverMatrix.h:48-53); a multiple of 64 is a row shard of one -- the unit
    // mvm_parallel hands a thread (CloverMatrix4.h:1700-1705) and clm4_sharded_* / sharding.py hand a GPU
    CLV_REQUIRE(rows % 64 == 0 && cols % 128 == 0, "%s: rows=%llu must be a multiple of 64 and cols=%llu of 128", fn,
                (unsigned long long)rows, (unsigned long long)cols);
    CLV_REQUIRE(rows / 64 <= 0x7FFFFFFFull, "%s: too many rows", fn);
    return CLV_OK;
}

extern "C" int clm4_restore(const int8_t *q, const float *s, uint64_t rows, uint64_t cols, float *A, void *stream)
{
    CLV_REQUIRE(A && q && s, "clm4_restore: null pointer");
    CLV_REQUIRE(rows % 128 == 0 && cols % 128 == 0, "clm4_restore: rows=%llu cols=%llu must be multiples of 128", (unsigned long long)rows,
                (unsigned long long)cols);
    if (!rows || !cols) return CLV_OK;
    const uint64_t nquads = rows * cols / 4, waves = (nquads + 255) / 256;
    CLV_REQUIRE((waves + 3) / 4 <= 0x7FFFFFFFull, "clm4_restore: matrix too large");
    if (rows * cols * sizeof(float) > (256ull << 20))
        hipLaunchKernelGGL(k_m4_restore<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), (const uint32_t *)q, s,
                           (f32x4 *)A, nquads, cols);
    else
        hipLaunchKernelGGL(k_m4_restore<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), (const uint32_t *)q, s,
                           (f32x4 *)A, nquads, cols);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clm4_mvm(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                        int8_t *r, float *sr, uint64_t *rng_state_dev, void *stream)
{
    int rc = check_mvm_args("clm4_mvm", A, sA, rows, cols, x, sx);
    if (rc) return rc;
    CLV_REQUIRE(r && sr, "clm4_mvm: null result pointer");
    if (!rows) return CLV_OK;
    hipStream_t st = as_stream(stream);
    return launch_mvm(A, sA, rows, cols, x, sx, nullptr, r, sr, rng_state_dev, st);
}

extern "C" int clm4_mvm_scale_and_add(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                                      const int8_t *qu, const float *su, float a, int8_t *t, float *st_, int8_t *r, float *sr,
                                      uint64_t *rng_state_dev, void *stream)
{
    int rc = check_mvm_args("clm4_mvm_scale_and_add", A, sA, rows, cols, x, sx);
    if (rc) return rc;
    CLV_REQUIRE(qu && su && r && sr, "clm4_mvm_scale_and_add: null pointer");
    CLV_REQUIRE((t == nullptr) == (st_ == nullptr), "clm4_mvm_scale_and_add: t and st must both be given or both be NULL");
    CLV_REQUIRE((const void *)r != (const void *)x && (const void *)sr != (const void *)sx,
                "clm4_mvm_scale_and_add: the result must not alias the vector being multiplied");
    if (!rows) return CLV_OK;
    const MvmFuse fuse = {(const uint32_t *)qu, su, a, (uint32_t *)r, sr};
    return launch_mvm(A, sA, rows, cols, x, sx, nullptr, t, st_, rng_state_dev, as_stream(stream), &fuse);
}

extern "C" int clm4_rowdots(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                            float *d, void *stream)
{
    int rc = check_mvm_args("clm4_rowdots", A, sA, rows, cols, x, sx);
    if (rc) return rc;
    CLV_REQUIRE(d, "clm4_rowdots: null result pointer");
    if (!rows) return CLV_OK;
    return launch_mvm(A, sA, rows, cols, x, sx, d, nullptr, nullptr, nullptr, as_stream(stream));
}

extern "C" int clm4_quantize(const float *A, uint64_t rows, uint64_t cols, int8_t *q, float *s, uint64_t *rng_state_dev, void *stream)
{
    CLV_REQUIRE(A && q && s, "clm4_quantize: null pointer");
    CLV_REQUIRE(rows % 128 == 0 && cols % 128 == 0, "clm4_quantize: rows=%llu cols=%llu must be multiples of 128",
                (unsigned long long)rows, (unsigned long long)cols);
    if (!rows || !cols) return CLV_OK;
    const uint64_t tiles = (rows / 64) * (cols / 64);
    CLV_REQUIRE(tiles <= 0x7FFFFFFFull, "clm4_quantize: too many tiles");
    if (rng_state_dev) return clm4_quantize_stochastic(A, rows, cols, q, s, rng_state_dev, as_stream(stream));
    static const bool tile_kernel = getenv("CLV_M4Q_TILE") != nullptr;       // A/B switch: the older one-tile-per-workgroup kernel
    if (!tile_kernel) {
        const uint32_t strips_x = (uint32_t)((cols + 255) / 256);
        hipLaunchKernelGGL(k_m4_quantize_strip, dim3((unsigned)((rows / 64) * strips_x)), dim3(256), 0, as_stream(stream), A, cols,
                           (uint32_t *)q, s, strips_x, (uint32_t)(cols / 64));
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    hipLaunchKernelGGL(k_m4_quantize, dim3((unsigned)tiles), dim3(256), 0, as_stream(stream), A, cols, (uint32_t *)q, s,
                       (uint32_t)(cols / 64));
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clm4_gemm(const int8_t *A, const float *sA, uint64_t M, uint64_t K, const int8_t *B, const float *sB, uint64_t N,
                         float *C, void *stream)
{
    CLV_REQUIRE(A && sA && B && sB && C, "clm4_gemm: null pointer");
    CLV_REQUIRE(M % 128 == 0 && N % 128 == 0 && K % 128 == 0, "clm4_gemm: M=%llu N=%llu K=%llu must be multiples of 128",
                (unsigned long long)M, (unsigned long long)N, (unsigned long long)K);
    if (!M || !N) return CLV_OK;
    // CLV_GEMM_KERNEL (A/B runs): "i8" = the int8 MFMA kernel of gemm4.hip, "simple" = the scalar check kernel
    static const int which = [] { const char *e = getenv("CLV_GEMM_KERNEL"); return !e ? 0 : !strcmp(e, "simple") ? 2 : !strcmp(e, "i8") ? 1 : 0; }();
    if (which == 0 && K > 0) return clm4_gemm_fp6(A, sA, M, K, B, sB, N, C, as_stream(stream));
    if (which == 1 && K > 0) return clm4_gemm_mfma(A, sA, M, K, B, sB, N, C, as_stream(stream));
    hipLaunchKernelGGL(k_m4_gemm_simple, dim3((unsigned)(N / 16), (unsigned)(M / 16)), dim3(256), 0, as_stream(stream),
                       (const uint8_t *)A, sA, M, K, (const uint8_t *)B, sB, N, C);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
