// mixed8.hip -- mixed precision: 4-bit matrix x 8-bit vector ("mixed 4b x 8b is what the paper's
// best-accuracy runs use").  CloverVector8 (CloverVector8.h:35-140): int8 values in natural order, one fp32 scale per
// 64 elements, value = q * scale / 127.  Here: its quantize / restore and CloverMatrix4::mvm(CloverVector8, CloverVector8)
// (CloverMatrix4.h:1093-1441), bit-exact in the reference's SIMD order.
#include "rng_device.h"

#include <stdlib.h>

// =================================================================================================
// CloverVector8::quantize (CloverVector8.h:393-606), rounding disabled: lane = one float4 in, 4 bytes out; a
// 64-element block is one DPP row of 16 lanes.  Algorithmic bytes: 4 + 1 + 1/16 per element.
// =================================================================================================
#define V8_LOADS 8
__global__ __launch_bounds__(256) void k_v8_quantize(const f32x4 *__restrict__ x, uint32_t *__restrict__ q, float *__restrict__ s,
                                                     uint64_t nquads)
{
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const uint64_t f0 = wave * (64 * V8_LOADS);                 // first float4 of this wave's chunk
    f32x4 v[V8_LOADS];
#pragma unroll
    for (int j = 0; j < V8_LOADS; j++) {
        const uint64_t f = f0 + 64 * j + lane;
        v[j] = __builtin_nontemporal_load(&x[f < nquads ? f : 0]);       // nquads is a multiple of 32: whole 16-lane blocks are in or out
    }
#pragma unroll
    for (int j = 0; j < V8_LOADS; j++) {
        const uint64_t f = f0 + 64 * j + lane;
        float m = fmaxf(fmaxf(__builtin_fabsf(v[j].x), __builtin_fabsf(v[j].y)), fmaxf(__builtin_fabsf(v[j].z), __builtin_fabsf(v[j].w)));
        m = fix_zero_max(row16_max(m));
        const float k = 127.0f / m;                             // CloverVector8.h:455
        const uint32_t w = ((uint32_t)quant1_det(v[j].x, k) & 0xFFu) | (((uint32_t)quant1_det(v[j].y, k) & 0xFFu) << 8) |
                           (((uint32_t)quant1_det(v[j].z, k) & 0xFFu) << 16) | (((uint32_t)quant1_det(v[j].w, k) & 0xFFu) << 24);
        if (f < nquads) {
            __builtin_nontemporal_store(k < __builtin_inff() ? w : 0u, &q[f]);      // k == inf: see quant_pack8 (common.h)
            if ((lane & 15) == 0) s[f >> 4] = m;
        }
    }
}

// stochastic: the same segment walk as k_v4_quantize_st (rng4.hip): lane = 8 elements = 8 bytes out;
// noise group g = element/8 (draw g>>2, byte g&3), word W[element%8] (CloverVector8.h:472-520, 546-553)
template <int S>
__global__ __launch_bounds__(256) void k_v8_quantize_st(const f32x4 *__restrict__ x, u32x2 *__restrict__ q, float *__restrict__ s,
                                                        uint64_t nblocks, uint64_t *state, uint64_t seq, RngTables T)
{
    typedef StShape<S> Sh;
    __shared__ __attribute__((aligned(16))) uint64_t raw_all[4][Sh::NBR * 2 * 4];
    __shared__ uint64_t base[4];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    SegRows<Sh::NSEG> segs;
    segs.load(Sh::seg_table(T), wave * Sh::NSEG);
    rng_workgroup_begin(state, seq, T.pow_rows, blockIdx.x, Sh::SHIFT, 2 * nblocks, base);
    uint64_t *raw = raw_all[wave];
    const uint64_t blk0 = ((uint64_t)blockIdx.x * 4 + wave) * (Sh::SEGLEN * Sh::NSEG);
    const int seg = lane >> 2, k = lane & 3;
    uint64_t a = segs.starts(base);
    const int rho = lane & 7;

    if constexpr (Sh::NSEG == 16) {
        // Large vectors (r5): the mapping of k_v4_quantize_st's large-vector form (rng4.hip).  Lane = one float4; a round is 16 pieces of 4
        // consecutive blocks (one per segment), load j reads piece j = one contiguous KiB, all 16 loads are in flight while the generator lanes
        // step the stream; a block is a DPP row of 16 lanes (maximum by row rotations).  Element 4 c + t of a block (c = lane & 15) takes
        // draw c >> 3, word 4 (c & 1) + t, byte (c >> 1) & 3 (CloverVector8.h:462-520 -- the element <-> noise map of the 8-lane form below,
        // re-indexed), i.e. ONE 16-byte LDS read per lane, and the lane's four bytes are one output dword: a store instruction of the wave
        // writes the 256 contiguous bytes of its piece.  (The 8-elements-per-lane form ran 0.66 of the HBM peak at n = 2^30.)
        const int c = lane & 15, g = c >> 1;
        uint32_t *q32 = reinterpret_cast<uint32_t *>(q);
        for (int r = 0; r < Sh::ROUNDS; r++) {
            f32x4 v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t blk = blk0 + (uint64_t)j * Sh::SEGLEN + Sh::BPR * r + (lane >> 4);
                v[j] = __builtin_nontemporal_load(&x[blk < nblocks ? blk * 16 + c : 0]);
            }
            if (r) __syncthreads();                          // the previous round's noise has been consumed
            a = gen_blocks(a, Sh::BPR, raw + (size_t)(Sh::BPR * seg) * 8, k);
            __syncthreads();
            const u32x4 *noise = reinterpret_cast<const u32x4 *>(raw) + (size_t)((lane >> 4) * 2 + (g >> 2)) * 2 + (c & 1);   // + 16 per piece
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint64_t blk = blk0 + (uint64_t)j * Sh::SEGLEN + Sh::BPR * r + (lane >> 4);
                float m = fmaxf(fmaxf(__builtin_fabsf(v[j].x), __builtin_fabsf(v[j].y)), fmaxf(__builtin_fabsf(v[j].z), __builtin_fabsf(v[j].w)));
                m = fix_zero_max(row16_max(m));
                const float kq = 127.0f / m;
                const u32x4 W = noise[(size_t)Sh::BPR * 4 * j];
                uint32_t out = ((uint32_t)quant1_st(v[j].x, kq, noise_of(W.x, g & 3)) & 0xFFu) | (((uint32_t)quant1_st(v[j].y, kq, noise_of(W.y, g & 3)) & 0xFFu) << 8) |
                               (((uint32_t)quant1_st(v[j].z, kq, noise_of(W.z, g & 3)) & 0xFFu) << 16) | ((uint32_t)quant1_st(v[j].w, kq, noise_of(W.w, g & 3)) << 24);
                if (!(kq < __builtin_inff())) out = 0u;
                if (blk < nblocks) {
                    __builtin_nontemporal_store(out, &q32[blk * 16 + c]);
                    if (c == 0) s[blk] = m;
                }
            }
        }
        return;
    }
    for (int r = 0; r < Sh::ROUNDS; r++) {
        if (lane < 4 * Sh::NSEG) a = gen_blocks(a, Sh::BPR, raw + (size_t)(Sh::BPR * seg) * 8, k);
        __syncthreads();
        f32x4 lo[Sh::STEPS], hi[Sh::STEPS];
#pragma unroll
        for (int u = 0; u < Sh::STEPS; u++) {
            const uint64_t blk = Sh::block(blk0, r, 8 * u + (lane >> 3));
            const uint64_t i = blk < nblocks ? blk * 8 + rho : 0;
            lo[u] = __builtin_nontemporal_load(&x[2 * i]);
            hi[u] = __builtin_nontemporal_load(&x[2 * i + 1]);
        }
#pragma unroll
        for (int u = 0; u < Sh::STEPS; u++) {
            const int bl = 8 * u + (lane >> 3);
            const uint64_t blk = Sh::block(blk0, r, bl);
            const float v[8] = {lo[u].x, lo[u].y, lo[u].z, lo[u].w, hi[u].x, hi[u].y, hi[u].z, hi[u].w};
            float m = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[e]));
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fmaxf(m, __shfl_xor(m, 4));
            m = fix_zero_max(m);
            const float kq = 127.0f / m;
            const u32x4 *Wp = reinterpret_cast<const u32x4 *>(raw + (size_t)(bl * 2 + (rho >> 2)) * 4);
            const u32x4 W0 = Wp[0], W1 = Wp[1];
            const uint32_t W[8] = {W0.x, W0.y, W0.z, W0.w, W1.x, W1.y, W1.z, W1.w};
            uint32_t b[8];
#pragma unroll
            for (int e = 0; e < 8; e++) b[e] = (uint32_t)quant1_st(v[e], kq, noise_of(W[e], rho & 3)) & 0xFFu;
            u32x2 out;
            out.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            out.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            if (!(kq < __builtin_inff())) out.x = out.y = 0u;
            if (blk < nblocks) {
                q[blk * 8 + rho] = out;
                if (rho == 0) s[blk] = m;
            }
        }
        if (r + 1 < Sh::ROUNDS) __syncthreads();
    }
}

// CloverVector8::restore (CloverVector8.h:835-909): x = (float)q * (scale / 127.0f); lane = one dword in, one float4 out
template <bool NT>
__global__ __launch_bounds__(256) void k_v8_restore(const uint32_t *__restrict__ q, const float *__restrict__ s, f32x4 *__restrict__ x,
                                                    uint64_t nquads)
{
    const uint64_t f0 = ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 256;
    const int lane = threadIdx.x & 63;
    uint32_t w[4];
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t f = f0 + 64 * j + lane, fc = f < nquads ? f : 0;
        w[j] = q[fc];
        sc[j] = s[fc >> 4];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t f = f0 + 64 * j + lane;
        const float k = div127(sc[j]);
        f32x4 v;
        v.x = (float)((int)(w[j] << 24) >> 24) * k;
        v.y = (float)((int)(w[j] << 16) >> 24) * k;
        v.z = (float)((int)(w[j] << 8) >> 24) * k;
        v.w = (float)((int)w[j] >> 24) * k;
        if (f < nquads) {
            if (NT) __builtin_nontemporal_store(v, &x[f]);
            else x[f] = v;
        }
    }
}

// =================================================================================================
// CloverVector8::scaleAndAdd (CloverVector8.h:1063-1358): r = quantize8(u + a * v) per 64-block,
//   val = fma((float)qv, f32(f32(sv*a)/127), (float)qu * f32(su/127)).  lane = 16 elements (one dwordx4 of u and of v), a block =
//   4 lanes.  Algorithmic bytes: 3 * (1 + 1/16) per element.  r/sr may alias qu/su.
// =================================================================================================
__device__ __forceinline__ float byte_f(uint32_t w, int e) { return (float)((int)(w << (24 - 8 * e)) >> 24); }

// the 16 values of one lane; returns their maximum magnitude
__device__ __forceinline__ float saa8_values(const u32x4 wu, const u32x4 wv, float su_ps, float sv_ps, float v[16])
{
    const uint32_t U[4] = {wu.x, wu.y, wu.z, wu.w}, V[4] = {wv.x, wv.y, wv.z, wv.w};
    float m = 0.0f;
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float du = byte_f(U[d], e) * su_ps;
            v[4 * d + e] = __builtin_fmaf(byte_f(V[d], e), sv_ps, du);
            m = fmaxf(m, __builtin_fabsf(v[4 * d + e]));
        }
    return m;
}

// noise == nullptr <=> rounding disabled; noise[d] = the XORShift word whose byte e belongs to element 4d + e
__device__ __forceinline__ u32x4 saa8_pack(const float v[16], float k, const uint32_t *noise)
{
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int qv = noise ? quant1_st(v[4 * d + e], k, noise_of(noise[d], e)) : quant1_det(v[4 * d + e], k);
            w |= ((uint32_t)qv & 0xFFu) << (8 * e);
        }
        o[d] = k < __builtin_inff() ? w : 0u;
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}

template <bool NT>
__global__ __launch_bounds__(256) void k_v8_scale_and_add(const u32x4 *qu, const float *su, const u32x4 *qv,
                                                          const float *sv, float a, u32x4 *r, float *sr, uint64_t nq16)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq16; i += stride) {
        const uint64_t b = i >> 2;
        const u32x4 wu = NT ? __builtin_nontemporal_load(qu + i) : qu[i];
        const u32x4 wv = NT ? __builtin_nontemporal_load(qv + i) : qv[i];
        float v[16];
        float m = saa8_values(wu, wv, div127(su[b]), div127(sv[b] * a), v);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        m = fix_zero_max(m);
        const u32x4 o = saa8_pack(v, 127.0f / m, nullptr);
        if (NT) __builtin_nontemporal_store(o, r + i); else r[i] = o;
        if ((i & 3) == 0) sr[b] = m;
    }
}

// Large vectors (r5, as k_v4_scale_and_add_blk in scale_add4.hip): the plain kernel above pays three IEEE divisions (s / 127 twice, 127 / max)
// and the quad reductions in EVERY lane of a block -- ~1.6 of its ~9 VALU instructions per element, and it is bound by their count
// (0.63 of the HBM peak at n = 2^30).  Here a wave takes 64 blocks: four 16-byte loads per operand and lane (quarter blocks, one
// contiguous KiB per instruction), then the per-block arithmetic ONCE per block with lane = block (phases A, C) and ds_bpermute between
// that lane and the block's four quarter-block lanes.
template <bool NT>
__global__ __launch_bounds__(256) void k_v8_scale_and_add_blk(const u32x4 *qu, const float *su, const u32x4 *qv,
                                                              const float *sv, float a, u32x4 *r, float *sr, uint64_t nblocks)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave-uniform chunk index: scalar base addresses
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t c = (uint64_t)blockIdx.x * 4 + wave_in_wg; c * 64 < nblocks; c += nwaves) {
        const uint64_t b0 = c * 64;
        const uint32_t left = (uint32_t)(nblocks - b0 < 64 ? nblocks - b0 : 64);
        const bool full = left == 64;
        const u32x4 *pu = qu + 4 * b0, *pv = qv + 4 * b0;
        u32x4 *pr = r + 4 * b0;
        u32x4 wu[4], wv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t h = 64 * u + lane, hc = full || h < 4 * left ? h : 0;
            wu[u] = NT ? __builtin_nontemporal_load(pu + hc) : pu[hc];
            wv[u] = NT ? __builtin_nontemporal_load(pv + hc) : pv[hc];
        }
        const uint32_t blc = full || (uint32_t)lane < left ? lane : 0;
        const float fsu = su[b0 + blc], fsv = sv[b0 + blc];
        asm volatile("" ::: "memory");                                      // every load of the chunk precedes its stores (r may alias qu)
        // A: lane = block
        const float fa = div127(fsu), fb = div127(fsv * a);
        // B: lane = quarter block
        float v[4][16], mb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int src = 4 * (16 * u + (lane >> 2));
            const float cu = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(fa)));
            const float cv = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(fb)));
            float m = saa8_values(wu[u], wv[u], cu, cv, v[u]);
            m = fmaxf(m, __shfl_xor(m, 1));
            mb[u] = fmaxf(m, __shfl_xor(m, 2));
        }
        // C: lane = block L: its maximum sits in the quad at lanes 4 (L & 15) of step L >> 4
        const int from = 4 * (4 * (lane & 15));
        const float m0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[0])));
        const float m1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[1])));
        const float m2 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[2])));
        const float m3 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[3])));
        const float m = fix_zero_max(lane < 32 ? (lane < 16 ? m0 : m1) : (lane < 48 ? m2 : m3));
        const float k = 127.0f / m;                                         // IEEE-correct fp32 division (CloverVector8.h:1262)
        if (full || (uint32_t)lane < left) sr[b0 + lane] = m;
        // D: lane = quarter block
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const float kk = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (16 * u + (lane >> 2)), __float_as_int(k)));
            const u32x4 o = saa8_pack(v[u], kk, nullptr);
            const uint32_t h = 64 * u + lane;
            if (full || h < 4 * left) {
                if (NT) __builtin_nontemporal_store(o, pr + h); else pr[h] = o;
            }
        }
    }
}

// stochastic: the segment walk of the other vector kernels.  Element e of a block takes draw e>>5, byte e&3 of word (e&31)>>2
// (CloverVector8.h:1104-1126, 1193-1229): lane c (= 16 elements) reads the four words W[4 (c & 1) ..] of draw c >> 1.
template <int S>
__global__ __launch_bounds__(256) void k_v8_scale_and_add_st(const u32x4 *qu, const float *su, const u32x4 *qv,
                                                             const float *sv, float a, u32x4 *r, float *sr, uint64_t nblocks,
                                                             uint64_t *state, uint64_t seq, RngTables T)
{
    typedef StShape<S> Sh;
    __shared__ __attribute__((aligned(16))) uint64_t raw_all[4][Sh::NBR * 2 * 4];
    __shared__ uint64_t base[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    SegRows<Sh::NSEG> segs;
    segs.load(Sh::seg_table(T), wave * Sh::NSEG);
    rng_workgroup_begin(state, seq, T.pow_rows, blockIdx.x, Sh::SHIFT, 2 * nblocks, base);
    uint64_t *raw = raw_all[wave];
    const uint64_t blk0 = ((uint64_t)blockIdx.x * 4 + wave) * (Sh::SEGLEN * Sh::NSEG);
    const int seg = lane >> 2, k = lane & 3, c = lane & 3;
    uint64_t st = segs.starts(base);
    constexpr int STEPS = Sh::NBR / 16 > 0 ? Sh::NBR / 16 : 1;           // 16 blocks (64 lanes x 16 elements) per step
    for (int rr = 0; rr < Sh::ROUNDS; rr++) {
        if (lane < 4 * Sh::NSEG) st = gen_blocks(st, Sh::BPR, raw + (size_t)(Sh::BPR * seg) * 8, k);
        __syncthreads();
        u32x4 wu[STEPS], wv[STEPS];
        float fu[STEPS], fv[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; u++) {                                 // loads of the round first (r may alias qu)
            const int bl = 16 * u + (lane >> 2);
            const uint64_t blk = bl < Sh::NBR ? Sh::block(blk0, rr, bl) : nblocks;
            const uint64_t b = blk < nblocks ? blk : 0;
            wu[u] = qu[b * 4 + c];
            wv[u] = qv[b * 4 + c];
            fu[u] = su[b];
            fv[u] = sv[b];
        }
#pragma unroll
        for (int u = 0; u < STEPS; u++) {
            const int bl = 16 * u + (lane >> 2);
            const uint64_t blk = bl < Sh::NBR ? Sh::block(blk0, rr, bl) : nblocks;
            float v[16];
            float m = saa8_values(wu[u], wv[u], div127(fu[u]), div127(fv[u] * a), v);
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fix_zero_max(m);
            const u32x4 W = reinterpret_cast<const u32x4 *>(raw + (size_t)((bl < Sh::NBR ? bl : 0) * 2 + (c >> 1)) * 4)[c & 1];
            const uint32_t Wn[4] = {W.x, W.y, W.z, W.w};
            const u32x4 o = saa8_pack(v, 127.0f / m, Wn);
            if (blk < nblocks) {
                r[blk * 4 + c] = o;
                if (c == 0) sr[blk] = m;
            }
        }
        if (rr + 1 < Sh::ROUNDS) __syncthreads();
    }
}

// =================================================================================================
// CloverMatrix4::mvm(const CloverVector8 &, CloverVector8 &)  (CloverMatrix4.h:1093-1441)
//
// Arithmetic fixed by the reference: per row 8 sequential fp32 fma chains (the 8 AVX lanes of dot_product_acc); chain L
// takes, from every 64-column block b, the exact integer I = sum of q4*q8 over elements 4L..4L+3 and 32+4L..32+4L+3, and
// does acc = fma(c_b, (float)I, acc) with c_b = f32(f32(sA*1/7) * f32(sx*1/127)); then ((a0+a4)+(a2+a6)) + ((a1+a5)+(a3+a7)).
// Mapping: workgroup = 64 output rows (256 threads) with the 8-bit re-quantisation fused; lane = (row, m) owns chains
// 2m and 2m+1.  Per step the four lanes of a row load 64 contiguous bytes of it (two blocks, one dwordx4 each) and
// transpose the 4x4 dwords inside the quad (two DPP exchanges) so that lane m ends up with the dwords holding ITS
// chains' nibbles: word m and word 4+m of either block.  A nibble dword becomes two int8 dwords ((w & 0xF0F0F0F0) and
// ((w << 4) & 0xF0F0F0F0): 16*q, no sign extension), v_perm_b32 puts elements e..e+3 next to each other, v_dot4_i32_i8
// against the natural-order int8 of x from LDS gives 16*I.  x (int8) and c_b are staged in LDS per 32768-column chunk.
// Algorithmic bytes per call: rows*cols/2 + 4*(rows/64)*(cols/64) + 1.0625*(rows + cols).
// =================================================================================================
#define MVM8_CHUNK 32768u

// 8 nibbles of a dword (elements e0..e7) -> int8 dwords {e0..e3} and {e4..e7}, each value times 16
__device__ __forceinline__ void widen8(uint32_t w, uint32_t &d03, uint32_t &d47)
{
    const uint32_t lo = w & 0xF0F0F0F0u;                 // bytes [e0, e2, e4, e6] * 16
    const uint32_t hi = (w << 4) & 0xF0F0F0F0u;          // bytes [e1, e3, e5, e7] * 16
    d03 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);    // [lo.b0, hi.b0, lo.b1, hi.b1]
    d47 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);    // [lo.b2, hi.b2, lo.b3, hi.b3]
}

__device__ __forceinline__ int sdot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }

// one block (64 columns) for lane m: w_first = word m, w_second = word 4+m of the block, xb = the block's 64 int8 in LDS
__device__ __forceinline__ void mvm8_block(uint32_t w_first, uint32_t w_second, const u32x2 *xb, int m, float c, float &a_even, float &a_odd)
{
    uint32_t f03, f47, s03, s47;
    widen8(w_first, f03, f47);
    widen8(w_second, s03, s47);
    const u32x2 x0 = xb[m], x1 = xb[4 + m];          // elements 8m..8m+7 and 32+8m..32+8m+7
    const int ie = sdot4(s03, x1.x, sdot4(f03, x0.x, 0)) >> 4;      // chain 2m   (exact: the sum is a multiple of 16)
    const int io = sdot4(s47, x1.y, sdot4(f47, x0.y, 0)) >> 4;      // chain 2m+1
    a_even = __builtin_fmaf(c, (float)ie, a_even);
    a_odd = __builtin_fmaf(c, (float)io, a_odd);
}

// FUSE: the CloverVector8::scaleAndAdd that follows this mvm in the IHT / GD loops, done on the row group while it is still in
// the wave: r2 = quantize8(u + a * quantize8(A x)); its draws follow ALL the mvm draws in the stream, as in two separate calls.
struct Mvm8Fuse {
    const int8_t *qu;        // u, one 64-element block per row group
    const float *su;
    float a;
    int8_t *r2;              // may alias qu (the in-place overload)
    float *sr2;
};

// LDS tail shared by the two mvm kernels: 64 row dots, then the generator bases and raw draws (ST)
struct Mvm8Tail {
    float *dsh;
    uint64_t *rbase, *raw, *rbase2, *raw2;
    __device__ __forceinline__ explicit Mvm8Tail(float *p)
        : dsh(p), rbase(reinterpret_cast<uint64_t *>(p + 64)), raw(rbase + 4), rbase2(raw + 8), raw2(rbase2 + 4) {}
};
#define MVM8_TAIL_BYTES (64 * sizeof(float) + 256)

// generator bases for this row group (ST) and its block of u (FUSE), fetched before the streaming loop
template <bool ST, bool FUSE>
__device__ __forceinline__ void mvm8_prologue(const Mvm8Tail &t, uint64_t *rng_state, uint64_t seq, const uint64_t *__restrict__ pow_rows,
                                              const Mvm8Fuse &fuse, int &fuse_q, float &fuse_s)
{
    if (ST) {
        const uint64_t a0 = rng_workgroup_begin(rng_state, seq, pow_rows, blockIdx.x, 1, (FUSE ? 4ull : 2ull) * gridDim.x, t.rbase);
        if (FUSE) {
            const uint64_t b2 = wave_pow_apply(pow_rows, a0, (uint64_t)gridDim.x + blockIdx.x, 1);
            if ((threadIdx.x & 63) == 0) t.rbase2[threadIdx.x >> 6] = b2;
        }
    }
    fuse_q = 0;
    fuse_s = 0.0f;
    if (FUSE && threadIdx.x < 64) {
        fuse_q = fuse.qu[blockIdx.x * 64 + threadIdx.x];
        fuse_s = fuse.su[blockIdx.x];
    }
}

// t.dsh holds the 64 row dots of row group rb (written before the call, no barrier yet): re-quantise to 8 bits
// (CloverMatrix4.h:1246-1440; noise group g = l>>3 (draw g>>2, byte g&3), word W[l&7]) and, with FUSE, scaleAndAdd
template <bool ST, bool FUSE>
__device__ __forceinline__ void mvm8_epilogue(const Mvm8Tail &t, uint64_t rb, int8_t *r, float *sr, const Mvm8Fuse &fuse, int fuse_q, float fuse_s)
{
    const int tid = threadIdx.x;
    if (ST && tid < 4) {
        gen_blocks(t.rbase[tid], 1, t.raw, tid);
        if (FUSE) gen_blocks(t.rbase2[tid], 1, t.raw2, tid);
    }
    __syncthreads();
    if ((r || FUSE) && tid < 64) {
        const float d = t.dsh[tid];
        float noise = 0.0f;
        if (ST) {
            const int g = tid >> 3, j = tid & 7;
            noise = noise_of(reinterpret_cast<const uint32_t *>(t.raw + (size_t)(g >> 2) * 4)[j], g & 3);
        }
        float mx = wave_max(__builtin_fabsf(d));
        mx = fix_zero_max(mx);
        const int qv = quant1(d, 127.0f / mx, noise);
        if (r) {
            r[rb * 64 + tid] = (int8_t)qv;
            if (tid == 0) sr[rb] = mx;
        }
        if (FUSE) {
            // CloverVector8::scaleAndAdd on this block (CloverVector8.h:1089-1358); element l: draw l>>5, byte l&3, word (l&31)>>2
            const float val = __builtin_fmaf((float)qv, div127(mx * fuse.a), (float)fuse_q * div127(fuse_s));
            float noise2 = 0.0f;
            if (ST) noise2 = noise_of(reinterpret_cast<const uint32_t *>(t.raw2 + (size_t)(tid >> 5) * 4)[(tid & 31) >> 2], tid & 3);
            float m2 = wave_max(__builtin_fabsf(val));
            m2 = fix_zero_max(m2);
            fuse.r2[rb * 64 + tid] = (int8_t)quant1(val, 127.0f / m2, noise2);
            if (tid == 0) fuse.sr2[rb] = m2;
        }
    }
}

#ifndef MVM8_U
#define MVM8_U 8                  // 16-byte matrix loads in flight per lane
#endif
#ifndef MVM8_MIN_WAVES
#define MVM8_MIN_WAVES 4          // waves per SIMD the register allocation must leave room for
#endif
template <int U, bool NT, bool ST, bool FUSE>
__global__ __launch_bounds__(256, MVM8_MIN_WAVES) void k_m4_mvm8(const uint8_t *__restrict__ A, const float *__restrict__ sA, uint64_t cols,
                                                 const int8_t *__restrict__ x, const float *__restrict__ sx, float *__restrict__ d_out,
                                                 int8_t *r, float *sr, uint64_t *rng_state, uint64_t seq,
                                                 const uint64_t *__restrict__ pow_rows, Mvm8Fuse fuse)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *xs = reinterpret_cast<u32x4 *>(smem);                         // MVM8_CHUNK bytes of int8
    float *cs = reinterpret_cast<float *>(smem + MVM8_CHUNK);            // MVM8_CHUNK/64 floats
    const Mvm8Tail tail(cs + MVM8_CHUNK / 64);
    float *dsh = tail.dsh;
    int fuse_q;
    float fuse_s;
    mvm8_prologue<ST, FUSE>(tail, rng_state, seq, pow_rows, fuse, fuse_q, fuse_s);

    const uint64_t rb = blockIdx.x;
    const int tid = threadIdx.x;
    const int m = tid & 3;
    const int rho = tid >> 2;
    const uint64_t row = rb * 64 + rho;
    const u32x4 *Arow = reinterpret_cast<const u32x4 *>(A + row * (cols / 2));
    const float *sArow = sA + rb * (cols / 64);
    float a_even = 0.0f, a_odd = 0.0f;

    for (uint64_t c0 = 0; c0 < cols; c0 += MVM8_CHUNK) {
        const uint32_t cw = (uint32_t)((cols - c0) < MVM8_CHUNK ? (cols - c0) : MVM8_CHUNK);
        if (c0) __syncthreads();
        {
            constexpr int NX = MVM8_CHUNK / 16 / 256;         // 8 x 16 B per thread
            constexpr int NC = MVM8_CHUNK / 64 / 256;         // 2 factors per thread
            const u32x4 *xg = reinterpret_cast<const u32x4 *>(x + c0);
            u32x4 xr[NX];
            float sa[NC], sv[NC];
            const uint32_t nx = cw / 16, nc = cw / 64;
#pragma unroll
            for (int k = 0; k < NX; k++) { const uint32_t i = tid + 256 * k; xr[k] = xg[i < nx ? i : 0]; }
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const uint32_t i = tid + 256 * k, ii = i < nc ? i : 0;
                sa[k] = sArow[c0 / 64 + ii];
                sv[k] = sx[c0 / 64 + ii];
            }
#pragma unroll
            for (int k = 0; k < NX; k++) { const uint32_t i = tid + 256 * k; if (i < nx) xs[i] = xr[k]; }
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const uint32_t i = tid + 256 * k;
                if (i < nc) cs[i] = (sa[k] * (1.0f / 7.0f)) * (sv[k] * (1.0f / 127.0f));           // CloverMatrix4.h:1147-1149
            }
        }
        __syncthreads();

        const u32x4 *Ap = Arow + c0 / 32;
        const u32x2 *xq = reinterpret_cast<const u32x2 *>(xs);
        const uint32_t nsteps = cw / 128;                     // two blocks per step
        uint32_t t = 0;
        for (; t + U <= nsteps; t += U) {
            u32x4 a[U];
#pragma unroll
            for (int u = 0; u < U; u++) a[u] = NT ? __builtin_nontemporal_load(&Ap[4 * (t + u) + m]) : Ap[4 * (t + u) + m];
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint32_t w0 = a[u].x, w1 = a[u].y, w2 = a[u].z, w3 = a[u].w;
                quad_transpose4(w0, w1, w2, w3, m);           // now: word m, word 4+m of block 2(t+u); word m, word 4+m of the next
                const uint32_t b = 2 * (t + u);
                mvm8_block(w0, w1, xq + 8 * b, m, cs[b], a_even, a_odd);
                mvm8_block(w2, w3, xq + 8 * (b + 1), m, cs[b + 1], a_even, a_odd);
            }
        }
        for (; t < nsteps; t++) {
            const u32x4 av = NT ? __builtin_nontemporal_load(&Ap[4 * t + m]) : Ap[4 * t + m];
            uint32_t w0 = av.x, w1 = av.y, w2 = av.z, w3 = av.w;
            quad_transpose4(w0, w1, w2, w3, m);
            const uint32_t b = 2 * t;
            mvm8_block(w0, w1, xq + 8 * b, m, cs[b], a_even, a_odd);
            mvm8_block(w2, w3, xq + 8 * (b + 1), m, cs[b + 1], a_even, a_odd);
        }
    }

    // chain 2m / 2m+1 in lane m.  CloverMatrix4.h:1229-1234: h[L] = a[L+4] + a[L]; (h0 + h2) + (h1 + h3)
    const float he = a_even + __shfl_xor(a_even, 2), ho = a_odd + __shfl_xor(a_odd, 2);      // m = 0,2: h0, h1;  m = 1,3: h2, h3
    const float ge = he + __shfl_xor(he, 1), go = ho + __shfl_xor(ho, 1);                    // h0 + h2,  h1 + h3
    const float dot = ge + go;

    if (m == 0) {
        dsh[rho] = dot;
        if (d_out) d_out[row] = dot;
    }
    mvm8_epilogue<ST, FUSE>(tail, rb, r, sr, fuse, fuse_q, fuse_s);
}

// =================================================================================================
// CloverVector8::dot (CloverVector8.h:911-977; round 5 -- the one arithmetic method of the class that was still missing here).
//   Per 64-element block: I[w] = the 8 byte products that fall into 32-bit lane w of the block's two 32-byte halves (exact; |q| <= 127, so
//   the reference's maddubs never saturates), scale = f32(f32(su / 127*) f32(sv / 127*)) with 127* = the constant 1.0f / 127.0f, and ONE
//   accumulator register: acc[w] = fma(scale, (float)I[w], acc[w]) block after block -- 8 sequential chains of n / 64 steps -- then the
//   _mm256_haddf32_ps tree (CloverBase.h:149-157).
//   EXACT: that order, bit for bit, on the chain kernel of CloverVector4::dot (vector4.hip: one wave, 16 chain lanes, operands in a register
//          ring, generated asm): its lanes l = 8 a + w keep two accumulators a = 0, 1 per AVX lane w and its tree starts with
//          acc[0][w] + acc[1][w].  Here one fma step = ONE block, accumulator 0 carries the chain and accumulator 1 is fed zeros --
//          fma(0, 0, +0) = +0 and acc + 0 = acc exactly -- after which that kernel's tree IS _mm256_haddf32_ps over the 8 chains.
//          k_v8_dot_prep (all CUs) writes the operands in that kernel's layout.  Latency-bound by the definition: n / 64 dependent fmas
//          per chain, twice CloverVector4::dot's n / 128.
//   FAST:  exact block integers, per-block scale, fp32 partials by a fixed tree (dot_common.h): the memory-bound order, one launch --
//          what dot_parallel() means (the reference's own dot_parallel sums its threads' partials in unspecified order, :979-1060).
// =================================================================================================
#include "dot_common.h"

#define CLV_RCP127 (1.0f / 127.0f)      // clover_mm256_rcp_127_ps (CloverBase.h:87)
__device__ __forceinline__ int v8_word_isum(const uint32_t *__restrict__ qu, const uint32_t *__restrict__ qv, uint64_t blk, int w)
{
    // lane w of both halves: words w and 8 + w of the block's 16
    int I = __builtin_amdgcn_sdot4((int)qu[16 * blk + w], (int)qv[16 * blk + w], 0, false);
    return __builtin_amdgcn_sdot4((int)qu[16 * blk + 8 + w], (int)qv[16 * blk + 8 + w], I, false);
}

// The chain kernel's operand layout (k_v4_dot_prep2): thread = (block Q of 16 steps, chain lane l, piece 0..4); pieces 0..3 = f of steps
// 4 piece .. 4 piece + 3, piece 4 = c of steps 4 i + (l & 3), i = 0..3 (each lane of a quad keeps a quarter of the 16 factors).  Step p =
// CloverVector8 block p for the lanes of accumulator 0; accumulator 1 (l >= 8) and steps past the end hold zeros.
__global__ __launch_bounds__(256) void k_v8_dot_prep(const uint32_t *__restrict__ qu, const float *__restrict__ su, const uint32_t *__restrict__ qv,
                                                     const float *__restrict__ sv, uint64_t nblocks, uint64_t qblocks_padded, f32x4 *__restrict__ X)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < qblocks_padded * 80; t += stride) {
        const uint64_t Q = t / 80;
        const int r = (int)(t - Q * 80), l = r / 5, piece = r - 5 * l, w = l & 7;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t p = piece < 4 ? 16 * Q + 4 * piece + i : 16 * Q + 4 * i + (l & 3);
            v[i] = 0.0f;
            if (l < 8 && p < nblocks) v[i] = piece < 4 ? (float)v8_word_isum(qu, qv, p, w) : (su[p] * CLV_RCP127) * (sv[p] * CLV_RCP127);
        }
        X[t] = f32x4{v[0], v[1], v[2], v[3]};
    }
}

// FAST: lane = 16 bytes of each operand (a quarter block); the four lanes of a block add their integer sums, lane 0 of the quad folds
template <int U>
__global__ __launch_bounds__(DOT_FAST_THREADS) void k_v8_dot_fast1(const u32x4 *__restrict__ qu, const float *__restrict__ su, const u32x4 *__restrict__ qv,
                                                                   const float *__restrict__ sv, uint64_t nvec, unsigned long long *slots,
                                                                   float *__restrict__ out)
{
    __shared__ float sh[4];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += U * stride) {
        u32x4 a[U], b[U];
        float cu[U], cv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t j = i + u * stride, jc = j < nvec ? j : i;          // nvec, i - lane and stride are multiples of 4: quads stay whole
            a[u] = __builtin_nontemporal_load(&qu[jc]);
            b[u] = __builtin_nontemporal_load(&qv[jc]);
            cu[u] = su[jc >> 2];
            cv[u] = sv[jc >> 2];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int I = __builtin_amdgcn_sdot4((int)a[u].x, (int)b[u].x, 0, false);
            I = __builtin_amdgcn_sdot4((int)a[u].y, (int)b[u].y, I, false);
            I = __builtin_amdgcn_sdot4((int)a[u].z, (int)b[u].z, I, false);
            I = __builtin_amdgcn_sdot4((int)a[u].w, (int)b[u].w, I, false);
            I += __shfl_xor(I, 1);
            I += __shfl_xor(I, 2);
            const uint64_t j = i + u * stride;
            const float c = ((j & 3) == 0 && j < nvec) ? (cu[u] * CLV_RCP127) * (cv[u] * CLV_RCP127) : 0.0f;      // branch-free, see k_v4_dot_fast1
            acc = __builtin_fmaf(c, (float)I, acc);
        }
    }
    dot_hand_over_and_collect(block_sum_256(acc, sh), slots, out, sh);
}

extern "C" uint64_t clv8_dot_workspace_bytes(uint64_t n_pad) { return clv_internal_dot_chain_bytes(n_pad / 64); }

extern "C" int clv8_dot(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, uint64_t n_pad, int mode, float *out_dev,
                        void *workspace, void *stream)
{
    CLV_REQUIRE(qu && su && qv && sv && out_dev, "clv8_dot: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv8_dot: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    CLV_REQUIRE(mode == CLV_DOT_EXACT || mode == CLV_DOT_FAST, "clv8_dot: unknown mode %d", mode);
    hipStream_t st = as_stream(stream);
    if (!n_pad) { CLV_HIP(hipMemsetAsync(out_dev, 0, sizeof(float), st)); return CLV_OK; }
    if (mode == CLV_DOT_EXACT) {
        if (!workspace) {
            int rc = clv_internal_workspace(&workspace, clv8_dot_workspace_bytes(n_pad), st);
            if (rc) return rc;
        }
        const uint64_t qpad = clv_internal_dot_chain_blocks_padded(n_pad / 64);
        const uint64_t want = (qpad * 80 + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
        hipLaunchKernelGGL(k_v8_dot_prep, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, (const uint32_t *)qu, su, (const uint32_t *)qv, sv,
                           n_pad / 64, qpad, (f32x4 *)workspace);
        CLV_LAUNCH_CHECK();
        return clv_internal_dot_chain(workspace, qpad, out_dev, st);
    }
    const uint64_t nvec = n_pad / 16;
    // the collector reads at most DOT_FAST_THREADS * DOT_MAX_SLOTS_PER_THREAD slots (dot_common.h): the grid never exceeds that, whatever the CU count
    const uint64_t want = (nvec + DOT_FAST_THREADS - 1) / DOT_FAST_THREADS, cap_cu = (uint64_t)clv_cu_count() * 4,
                   cap = cap_cu < (uint64_t)DOT_FAST_THREADS * DOT_MAX_SLOTS_PER_THREAD ? cap_cu : (uint64_t)DOT_FAST_THREADS * DOT_MAX_SLOTS_PER_THREAD;
    const int grid = (int)(want < cap ? want : cap);
    void *slots = nullptr;
    int rc = clv_internal_sync_slots(&slots, (uint64_t)grid * 8, st);
    if (rc) return rc;
    const uint64_t per_thread = (nvec + (uint64_t)grid * DOT_FAST_THREADS - 1) / ((uint64_t)grid * DOT_FAST_THREADS);
    if (per_thread <= 2)
        hipLaunchKernelGGL(k_v8_dot_fast1<2>, dim3(grid), dim3(DOT_FAST_THREADS), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, nvec,
                           (unsigned long long *)slots, out_dev);
    else
        hipLaunchKernelGGL(k_v8_dot_fast1<1>, dim3(grid), dim3(DOT_FAST_THREADS), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, nvec,
                           (unsigned long long *)slots, out_dev);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" int clv8_quantize(const float *x, uint64_t n_pad, int8_t *q, float *s, uint64_t *rng_state_dev, void *stream)
{
    CLV_REQUIRE(x && q && s, "clv8_quantize: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv8_quantize: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    hipStream_t st = as_stream(stream);
    if (!rng_state_dev) {
        const uint64_t nquads = n_pad / 4, waves = (nquads + 64 * V8_LOADS - 1) / (64 * V8_LOADS);
        hipLaunchKernelGGL(k_v8_quantize, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const f32x4 *)x, (uint32_t *)q, s, nquads);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    RngTables T;
    int rc = clv_rng_tables(&T);
    if (rc) return rc;
    const uint64_t nb = n_pad / 64;
    const uint64_t seq = clv_rng_seq_for(rng_state_dev, st);
#define Q8_LAUNCH(S)                                                                                                           \
    hipLaunchKernelGGL(k_v8_quantize_st<S>, dim3((unsigned)((nb + 32 * S - 1) / (32 * S))), dim3(256), 0, st, (const f32x4 *)x, \
                       (u32x2 *)q, s, nb, rng_state_dev, seq, T)
    switch (clv_st_segments(nb, false)) {
    case 1: Q8_LAUNCH(1); break;
    case 4: Q8_LAUNCH(4); break;
    case 16: Q8_LAUNCH(16); break;
    default: Q8_LAUNCH(64); break;
    }
#undef Q8_LAUNCH
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv8_restore(const int8_t *q, const float *s, uint64_t n_pad, float *x, void *stream)
{
    CLV_REQUIRE(x && q && s, "clv8_restore: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv8_restore: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    const uint64_t nquads = n_pad / 4, waves = (nquads + 255) / 256;
    if (n_pad * sizeof(float) > (256ull << 20))
        hipLaunchKernelGGL(k_v8_restore<true>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), (const uint32_t *)q, s,
                           (f32x4 *)x, nquads);
    else
        hipLaunchKernelGGL(k_v8_restore<false>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, as_stream(stream), (const uint32_t *)q, s,
                           (f32x4 *)x, nquads);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

#ifndef SAA8_BLK_MIN_BLOCKS
#define SAA8_BLK_MIN_BLOCKS (1u << 21)      // n >= 2^27: where the operands leave the Infinity Cache; below it the plain kernel is as fast or faster (2^24: 9.6 against 10.7 us)
#endif
extern "C" int clv8_scale_and_add(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, float a, uint64_t n_pad, int8_t *r,
                                  float *sr, uint64_t *rng_state_dev, void *stream)
{
    CLV_REQUIRE(qu && su && qv && sv && r && sr, "clv8_scale_and_add: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv8_scale_and_add: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    hipStream_t st = as_stream(stream);
    const uint64_t nb = n_pad / 64;
    static const uint64_t blk_min = [] { const char *e = getenv("CLV_SAA8_BLK_MIN_BLOCKS"); return e ? strtoull(e, nullptr, 10) : (uint64_t)SAA8_BLK_MIN_BLOCKS; }();      // A/B runs
    if (!rng_state_dev && nb >= blk_min) {                                // once-per-block arithmetic (k_v8_scale_and_add_blk)
        const uint64_t want = (nb + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
        const dim3 grid((unsigned)(want < cap ? want : cap));
        if (3 * n_pad > (256ull << 20))
            hipLaunchKernelGGL(k_v8_scale_and_add_blk<true>, grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a, (u32x4 *)r, sr, nb);
        else
            hipLaunchKernelGGL(k_v8_scale_and_add_blk<false>, grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a, (u32x4 *)r, sr, nb);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (!rng_state_dev) {
        const uint64_t nq16 = n_pad / 16;
        const uint64_t want = (nq16 + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
        const dim3 grid((unsigned)(want < cap ? want : cap));
        if (3 * n_pad > (256ull << 20))
            hipLaunchKernelGGL(k_v8_scale_and_add<true>, grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a, (u32x4 *)r,
                               sr, nq16);
        else
            hipLaunchKernelGGL(k_v8_scale_and_add<false>, grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a, (u32x4 *)r,
                               sr, nq16);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    RngTables T;
    int rc = clv_rng_tables(&T);
    if (rc) return rc;
    const uint64_t seq = clv_rng_seq_for(rng_state_dev, st);
#define SAA8_LAUNCH(S)                                                                                                               \
    hipLaunchKernelGGL(k_v8_scale_and_add_st<S>, dim3((unsigned)((nb + 32 * S - 1) / (32 * S))), dim3(256), 0, st, (const u32x4 *)qu, su, \
                       (const u32x4 *)qv, sv, a, (u32x4 *)r, sr, nb, rng_state_dev, seq, T)
    switch (clv_st_segments(nb, true)) {
    case 1: SAA8_LAUNCH(1); break;
    case 4: SAA8_LAUNCH(4); break;
    case 16: SAA8_LAUNCH(16); break;
    default: SAA8_LAUNCH(64); break;
    }
#undef SAA8_LAUNCH
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

static int launch_mvm8(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, float *d,
                       int8_t *r, float *sr, uint64_t *rng, hipStream_t st, const Mvm8Fuse *fuse = nullptr)
{
    const size_t lds = MVM8_CHUNK + (MVM8_CHUNK / 64) * sizeof(float) + MVM8_TAIL_BYTES;
    const dim3 grid((unsigned)(rows / 64)), block(256);
    RngTables T = {nullptr, nullptr, nullptr};
    uint64_t seq = 0;
    if (rng) {
        int rc = clv_rng_tables(&T);
        if (rc) return rc;
        seq = clv_rng_seq_for(rng, st);
    }
    const bool streaming = rows * (cols / 2) > (256ull << 20);
    const Mvm8Fuse no_fuse = {nullptr, nullptr, 0.0f, nullptr, nullptr};
#define M8_LAUNCH_F(NT, ST, FUSE)                                                                                                      \
    hipLaunchKernelGGL((k_m4_mvm8<MVM8_U, NT, ST, FUSE>), grid, block, lds, st, (const uint8_t *)A, sA, cols, x, sx, d, r, sr, rng, seq,    \
                       T.pow_rows, FUSE ? *fuse : no_fuse)
#define M8_LAUNCH(NT, ST)                             \
    do {                                              \
        if (fuse) M8_LAUNCH_F(NT, ST, true);          \
        else M8_LAUNCH_F(NT, ST, false);              \
    } while (0)
    // (A matrix-core variant for matrices with few row groups -- v_mfma_i32_16x16x64_i8 with x masked per fma chain in the B
    //  columns, so that C[row][chain] is the chain's block integer -- was built and was bit-exact, but slower: 16.4 us against
    //  11.2 us at 4096 x 8192.  hipcc copies every MFMA result out of the accumulation registers right behind the MFMA, which
    //  serialises the blocks at MFMA latency; not worth hand-scheduling for this kernel.  Also tried: 8 lanes per row, one fma
    //  chain per lane (same loads, each lane widens half of every word): 13.2 us -- a SIMD issues one wave64 VALU instruction
    //  per 4 cycles however many waves it holds, so what counts is total VALU work per SIMD, and that variant has more.  r01.)
    if (streaming) {
        if (rng) M8_LAUNCH(true, true); else M8_LAUNCH(true, false);
    } else {
        if (rng) M8_LAUNCH(false, true); else M8_LAUNCH(false, false);
    }
#undef M8_LAUNCH
#undef M8_LAUNCH_F
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

static int check_mvm8_args(const char *fn, const void *A, const void *sA, uint64_t rows, uint64_t cols, const void *x, const void *sx)
{
    CLV_REQUIRE(A && sA && x && sx, "%s: null pointer", fn);
    // rows % 64: a row shard of a matrix (see check_mvm_args in matrix4.hip)
    CLV_REQUIRE(rows % 64 == 0 && cols % 128 == 0, "%s: rows=%llu must be a multiple of 64 and cols=%llu of 128", fn, (unsigned long long)rows,
                (unsigned long long)cols);
    CLV_REQUIRE(rows / 64 <= 0x7FFFFFFFull, "%s: too many rows", fn);
    return CLV_OK;
}

extern "C" int clm4_mvm_v8(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, int8_t *r,
                           float *sr, uint64_t *rng_state_dev, void *stream)
{
    int rc = check_mvm8_args("clm4_mvm_v8", A, sA, rows, cols, x, sx);
    if (rc) return rc;
    CLV_REQUIRE(r && sr, "clm4_mvm_v8: null result pointer");
    if (!rows) return CLV_OK;
    return launch_mvm8(A, sA, rows, cols, x, sx, nullptr, r, sr, rng_state_dev, as_stream(stream));
}

extern "C" int clm4_mvm_v8_scale_and_add(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx,
                                         const int8_t *qu, const float *su, float a, int8_t *t, float *st_, int8_t *r, float *sr,
                                         uint64_t *rng_state_dev, void *stream)
{
    int rc = check_mvm8_args("clm4_mvm_v8_scale_and_add", A, sA, rows, cols, x, sx);
    if (rc) return rc;
    CLV_REQUIRE(qu && su && r && sr, "clm4_mvm_v8_scale_and_add: null pointer");
    CLV_REQUIRE((t == nullptr) == (st_ == nullptr), "clm4_mvm_v8_scale_and_add: t and st must both be given or both be NULL");
    CLV_REQUIRE((const void *)r != (const void *)x && (const void *)sr != (const void *)sx,
                "clm4_mvm_v8_scale_and_add: the result must not alias the vector being multiplied");
    if (!rows) return CLV_OK;
    const Mvm8Fuse fuse = {qu, su, a, r, sr};
    return launch_mvm8(A, sA, rows, cols, x, sx, nullptr, t, st_, rng_state_dev, as_stream(stream), &fuse);
}

// Q_IHT / Q_GD (test/performance/01_measure.h:923-946, 999-1021) as the reference runs them for "4-bit": CloverMatrix4 with
// CloverVector8 vectors (02_bit04.cpp:140).  One call enqueues all iterations; 3 launches per iteration.
__global__ void k_v8_clear(uint32_t *q, float *s, uint64_t nwords, uint64_t nblocks)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) q[i] = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += stride) s[i] = 1.0f;
}

// iht_persist.hip: the whole loop as one persistent launch (N <= 8192, rounding disabled, threshold FAST or none); 1 = launched, 0 = no, < 0 = error
int clm4_iht_v8_persistent(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n, int8_t *x, float *sx,
                           uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2, float *st2, int8_t *t3, float *st3,
                           uint64_t iterations, uint64_t K, float mu, int threshold, uint64_t *rng, hipStream_t st);

extern "C" int clm4_iht_v8(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n, int8_t *x,
                           float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2, float *st2,
                           int8_t *t3, float *st3, uint64_t iterations, uint64_t K, float mu, int threshold, uint64_t *rng_state_dev,
                           void *stream)
{
    CLV_REQUIRE(Phi && sPhi && PhiT && sPhiT && x && sx && y && sy && t1 && st1 && t2 && st2 && t3 && st3, "clm4_iht_v8: null pointer");
    CLV_REQUIRE(m % 128 == 0 && n % 128 == 0 && x_len <= n, "clm4_iht_v8: m=%llu n=%llu x_len=%llu", (unsigned long long)m,
                (unsigned long long)n, (unsigned long long)x_len);
    hipStream_t st = as_stream(stream);
    {
        const int p = clm4_iht_v8_persistent(Phi, sPhi, PhiT, sPhiT, m, n, x, sx, x_len, y, sy, t1, st1, t2, st2, t3, st3, iterations, K, mu, threshold,
                                             rng_state_dev, st);
        if (p < 0) return CLV_ERR_HIP;
        if (p > 0) return CLV_OK;
    }
    hipLaunchKernelGGL(k_v8_clear, dim3(64), dim3(256), 0, st, (uint32_t *)x, sx, n / 4, n / 64);      // x.clear()
    CLV_LAUNCH_CHECK();
    for (uint64_t it = 0; it < iterations; it++) {
        int rc = clm4_mvm_v8_scale_and_add(Phi, sPhi, m, n, x, sx, y, sy, -1.0f, t1, st1, t2, st2, rng_state_dev, stream);      // t1 = Phi x; t2 = y - t1
        if (!rc) rc = clm4_mvm_v8_scale_and_add(PhiT, sPhiT, n, m, t2, st2, x, sx, mu, t3, st3, x, sx, rng_state_dev, stream);  // t3 = Phi' t2; x += mu t3
        if (!rc && threshold)                                                                    // keep the K largest (2: the reference's survivor order)
            rc = clv8_threshold_mode(x, sx, x_len, n, K, threshold == 2 ? CLV_THRESHOLD_REFERENCE : CLV_THRESHOLD_FAST, nullptr, stream);
        if (rc) return rc;
    }
    return CLV_OK;
}

extern "C" int clm4_rowdots_v8(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const int8_t *x, const float *sx, float *d,
                               void *stream)
{
    int rc = check_mvm8_args("clm4_rowdots_v8", A, sA, rows, cols, x, sx);
    if (rc) return rc;
    CLV_REQUIRE(d, "clm4_rowdots_v8: null result pointer");
    if (!rows) return CLV_OK;
    return launch_mvm8(A, sA, rows, cols, x, sx, d, nullptr, nullptr, nullptr, as_stream(stream));
}
