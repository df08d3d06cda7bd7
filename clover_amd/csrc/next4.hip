// next4.hip -- the callers either side of the hot path: scaleAndAdd, transpose, threshold (CloverVector4, and the
// large-vector / CloverVector8 threshold), the IHT / GD loop, and the mixed 4-bit x fp32 mvm.
// Together with mvm these are all five steps of the reference's quantized IHT / GD iterations
// (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "rng_device.h"

#include <type_traits>

// =================================================================================================
// f1  CloverVector4::scaleAndAdd (CloverVector4.h:1196-1478):  r = quantize(u + a * v), per 64-block
//     val = fma((float)qv, f32(f32(sv*a)/7), (float)qu * f32(su/7));  lane = 4 dwords (half a block) of u and of v.
//     algorithmic bytes: 3 * (1/2 + 1/16) = 1.6875 per element.
// =================================================================================================
// su7 = f32(su / 7), sv7 = f32(f32(sv * a) / 7).  Fast form: the nibbles come as q / 16 (unpack8_16th: one conversion per element, no
// masks) and the scales as 16 s -- bit-identical as long as 16 s does not overflow (times16_is_finite); blocks whose scale sits at the
// very top of the fp32 range take the plain form.
__device__ __forceinline__ void saa_values(uint32_t wu, uint32_t wv, float su7, float sv7, float v[8])
{
    const float su16 = su7 * 16.0f, sv16 = sv7 * 16.0f;
    float fu[8], fv[8];
    unpack8_16th(wu, fu);
    unpack8_16th(wv, fv);
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = __builtin_fmaf(fv[e], sv16, fu[e] * su16);
    // a WAVE-uniform branch around the plain form: hipcc would otherwise if-convert a per-lane one and run both forms everywhere
    if (!__all(times16_is_finite(su7) && times16_is_finite(sv7))) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float du = (float)unpack1(wu, e) * su7;
            v[e] = __builtin_fmaf((float)unpack1(wv, e), sv7, du);
        }
    }
}

// lane = 4 dwords (half a block) of u and of v: 16-byte loads/stores, one shuffle for the block maximum.  U grid-stride steps per
// iteration, all 2 U loads requested before the first value is computed.  Round 5 measured U = 1 / 2 / 4 on one box at n = 2^30
// (tools/build_variant.py, profiles/r05_weak_kernels_ab.txt): 0.3442 / 0.3430 / 0.3377 ms -- the loads in flight are not what bounds
// this 2-reads-1-write stream (5.3 TB/s, in family with the chip's copy-shaped kernels), so the plain U = 1 form stays.
#ifndef SAA_U
#define SAA_U 1
#endif
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_v4_scale_and_add(const u32x4 *qu, const float *su, const u32x4 *__restrict__ qv,
                                                          const float *__restrict__ sv, float a, u32x4 *r, float *sr,
                                                          uint64_t nquads)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < nquads; i0 += U * stride) {
        u32x4 wu[U], wv[U];
        float fu[U], fv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                          // r may alias qu: every load of the iteration precedes its stores, lane by lane
            const uint64_t i = i0 + u * stride, ic = i < nquads ? i : i0;      // nquads is even, i0 and stride have one parity: lane pairs stay whole
            wu[u] = NT ? __builtin_nontemporal_load(qu + ic) : qu[ic];
            wv[u] = NT ? __builtin_nontemporal_load(qv + ic) : qv[ic];
            fu[u] = su[ic >> 1];
            fv[u] = sv[ic >> 1];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = i0 + u * stride, b = i >> 1;
            const float su7 = div7(fu[u]);
            const float sv7 = div7(fv[u] * a);
            float v[4][8];
            saa_values(wu[u].x, wv[u].x, su7, sv7, v[0]);
            saa_values(wu[u].y, wv[u].y, su7, sv7, v[1]);
            saa_values(wu[u].z, wv[u].z, su7, sv7, v[2]);
            saa_values(wu[u].w, wv[u].w, su7, sv7, v[3]);
            float m = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[q][e]));
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fix_zero_max(m);
            const float k = 7.0f / m;
            u32x4 o;
            o.x = quant_pack8(v[0], k, nullptr);
            o.y = quant_pack8(v[1], k, nullptr);
            o.z = quant_pack8(v[2], k, nullptr);
            o.w = quant_pack8(v[3], k, nullptr);
            if (i < nquads) {
                if (NT) __builtin_nontemporal_store(o, r + i); else r[i] = o;
                if ((i & 1) == 0) sr[b] = m;
            }
        }
    }
}

// Large vectors (round 5): the same arithmetic with the PER-BLOCK work done once per block instead of once per lane.  Counters of the
// kernel above at n = 2^30 (profiles/r05_weak_kernels_pmc.txt): 8.4 VALU instructions per element = 87 % of the VALU issue slots at
// 1.84 GHz -- it is bound by its instruction count, not by HBM -- and ~1.2 of the 8.4 are block scalars (two div7, the 1/16 scaling, the
// exactness test, fix_zero, the division 7 / max, the overflow guard) that both lanes of a block compute for themselves.  Here a wave
// takes 64 blocks = two steps of 64 half-block lanes:
//   A  lane = BLOCK: the 64 scale pairs are one coalesced load each; su7, sv7 and their 16-fold (see unpack8_16th) once per block;
//      ds_bpermute hands lane (step, half-block) its block's two factors (the LDS crossbar, not the VALU);
//   B  lane = half a block, as above: nibbles as q / 16 (9 instead of 11 VALU per word), value = fma(qv/16, 16 sv7, qu/16 * 16 su7),
//      |max| of the 32 values, pair maximum by DPP;
//   C  lane = BLOCK again: maxima gathered by two ds_bpermute, 0 -> 1.0, k = 7 / max (k = 0 where the reference's cvttps overflow
//      makes every nibble 0, see quant_pack8), ONE coalesced store of the 64 scales, k back out by ds_bpermute;
//   D  quantise and store.
// All loads of a wave's chunk precede its stores lane by lane, so r may alias qu or qv as before.
__device__ __forceinline__ uint32_t quant_pack8_k(const float v[8], float k)      // quant_pack8 without its overflow guard (folded into k)
{
    uint32_t even = cvt_i32_byte0_first(v[0] * k), odd = cvt_i32_byte0_first(v[1] * k);
    cvt_i32_into_byte1(even, v[2] * k);
    cvt_i32_into_byte1(odd, v[3] * k);
    cvt_i32_into_byte2(even, v[4] * k);
    cvt_i32_into_byte2(odd, v[5] * k);
    cvt_i32_into_byte3(even, v[6] * k);
    cvt_i32_into_byte3(odd, v[7] * k);
    return nibbles_from_bytes(even, odd);
}

template <bool NT>
__global__ __launch_bounds__(256) void k_v4_scale_and_add_blk(const u32x4 *qu, const float *su, const u32x4 *__restrict__ qv,
                                                              const float *__restrict__ sv, float a, u32x4 *r, float *sr, uint64_t nblocks)
{
    const int lane = threadIdx.x & 63;
    // the chunk index is wave-uniform: told to the compiler (readfirstlane), so that the chunk's base addresses and the full / ragged
    // decision live in scalar registers and a lane adds only its own 32-bit offset (the 64-bit per-lane index arithmetic and the clamps
    // were 10 % of the kernel's instructions)
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t c = (uint64_t)blockIdx.x * 4 + wave_in_wg; c * 64 < nblocks; c += nwaves) {
        const uint64_t b0 = c * 64;
        const uint32_t left = (uint32_t)(nblocks - b0 < 64 ? nblocks - b0 : 64);      // blocks of this chunk: 64 except in the last one
        const u32x4 *pu = qu + 2 * b0, *pv = qv + 2 * b0;
        u32x4 *pr = r + 2 * b0;
        const float *psu = su + b0, *psv = sv + b0;
        float *psr = sr + b0;
        const bool full = left == 64;
        u32x4 wu[2], wv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t h = 64 * u + lane, hc = full || h < 2 * left ? h : 0;
            wu[u] = NT ? __builtin_nontemporal_load(pu + hc) : pu[hc];
            wv[u] = NT ? __builtin_nontemporal_load(pv + hc) : pv[hc];
        }
        const uint32_t blc = full || (uint32_t)lane < left ? lane : 0;
        const float fsu = psu[blc], fsv = psv[blc];
        asm volatile("" ::: "memory");
        // A: lane = block
        const float su7 = div7(fsu), sv7 = div7(fsv * a);
        const bool fast = __all(times16_is_finite(su7) && times16_is_finite(sv7));      // wave-uniform: a real branch below
        const float fa = fast ? su7 * 16.0f : su7, fb = fast ? sv7 * 16.0f : sv7;
        float cu[2], cv[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int src = 4 * (32 * u + (lane >> 1));
            cu[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(fa)));
            cv[u] = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(fb)));
        }
        // B: lane = half a block
        float v[2][4][8], mb[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t xu[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w}, xv[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
            if (fast) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float fu[8], fv[8];
                    unpack8_16th(xu[q], fu);
                    unpack8_16th(xv[q], fv);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[u][q][e] = __builtin_fmaf(fv[e], cv[u], fu[e] * cu[u]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int e = 0; e < 8; e++) v[u][q][e] = __builtin_fmaf((float)unpack1(xv[q], e), cv[u], (float)unpack1(xu[q], e) * cu[u]);
            }
            float m = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[u][q][e]));
            mb[u] = fmaxf(m, __shfl_xor(m, 1));
        }
        // C: lane = block (lanes 0..31: the blocks of step 0, lanes 32..63: step 1); the block's maximum sits in lanes 2 b', 2 b' + 1
        const int from = 4 * (2 * (lane & 31));
        const float m0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[0])));
        const float m1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[1])));
        const float m = fix_zero_max(lane < 32 ? m0 : m1);
        float k = 7.0f / m;                                   // IEEE-correct fp32 division (CloverVector4.h:1390)
        k = k < __builtin_inff() ? k : 0.0f;                  // 7 / max overflows: every nibble of the block is 0 (quant_pack8's guard)
        if (full || (uint32_t)lane < left) psr[lane] = m;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const float kk = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (32 * u + (lane >> 1)), __float_as_int(k)));
            // D
            u32x4 o;
            o.x = quant_pack8_k(v[u][0], kk);
            o.y = quant_pack8_k(v[u][1], kk);
            o.z = quant_pack8_k(v[u][2], kk);
            o.w = quant_pack8_k(v[u][3], kk);
            const uint32_t h = 64 * u + lane;
            if (full || h < 2 * left) {
                if (NT) __builtin_nontemporal_store(o, pr + h); else pr[h] = o;
            }
        }
    }
}

// stochastic variant: same segment walk as k_v4_quantize_st (rng4.hip); the nibbles are unpacked by bit
// position there, so noise group g of AVX lane j meets element 8j + (g ^ 1) (CloverVector4.h:1236-1243).
template <int S, bool NT = false>
__global__ __launch_bounds__(256) void k_v4_scale_and_add_st(const uint32_t *qu, const float *su, const uint32_t *__restrict__ qv,
                                                             const float *__restrict__ sv, float a, uint32_t *r, float *sr,
                                                             uint64_t nblocks, uint64_t *state, uint64_t seq, RngTables T)
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
    const int seg = lane >> 2, k = lane & 3, rho = lane & 7;
    uint64_t st = segs.starts(base);                          // workgroup base, then this segment's T^(16 e)
    for (int rr = 0; rr < Sh::ROUNDS; rr++) {
        if (lane < 4 * Sh::NSEG) st = gen_blocks(st, Sh::BPR, raw + (size_t)(Sh::BPR * seg) * 8, k);
        __syncthreads();
        if constexpr (S == 1) {
            // small vectors: lane = one dword, 8 blocks per wave
            uint32_t wu[Sh::STEPS], wv[Sh::STEPS];
            float fu[Sh::STEPS], fv[Sh::STEPS];
#pragma unroll
            for (int u = 0; u < Sh::STEPS; u++) {                 // all loads of the round first (r may alias qu: loads precede stores per block)
                const uint64_t blk = Sh::block(blk0, rr, 8 * u + (lane >> 3));
                const uint64_t b = blk < nblocks ? blk : 0;
                wu[u] = qu[b * 8 + rho];
                wv[u] = qv[b * 8 + rho];
                fu[u] = su[b];
                fv[u] = sv[b];
            }
#pragma unroll
            for (int u = 0; u < Sh::STEPS; u++) {
                const int bl = 8 * u + (lane >> 3);
                const uint64_t blk = Sh::block(blk0, rr, bl);
                float v[8];
                saa_values(wu[u], wv[u], div7(fu[u]), div7(fv[u] * a), v);
                float m = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[e]));
                m = fmaxf(m, __shfl_xor(m, 1));
                m = fmaxf(m, __shfl_xor(m, 2));
                m = fmaxf(m, __shfl_xor(m, 4));
                m = fix_zero_max(m);
                const float kq = 7.0f / m;
                const uint32_t *W32 = reinterpret_cast<const uint32_t *>(raw + (size_t)(bl * 2) * 4);
                const uint32_t Wd[2] = {W32[rho], W32[8 + rho]};          // W[j = rho] of draw 0 and draw 1
                float n0[4], n1[4], nz[8];
                noise4_of(Wd[0], n0);
                noise4_of(Wd[1], n1);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int g = e ^ 1;
                    nz[e] = (g >> 2) ? n1[g & 3] : n0[g & 3];
                }
                const uint32_t packed = quant_pack8(v, kq, nz);
                if (blk < nblocks) {
                    r[blk * 8 + rho] = packed;
                    if (rho == 0) sr[blk] = m;
                }
            }
        } else if constexpr (Sh::NBR == 64) {
            // 64 blocks per round = two steps of half-block lanes, the per-block work once per BLOCK (phases A..D of
            // k_v4_scale_and_add_blk; lane L of the block phases <-> local block L of the round)
            const int half = lane & 1;
            u32x4 wu[2], wv[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const uint64_t blk = Sh::block(blk0, rr, 32 * u + (lane >> 1));
                const uint64_t b = blk < nblocks ? blk : 0;
                const u32x4 *pu = reinterpret_cast<const u32x4 *>(qu) + (b * 2 + half), *pv = reinterpret_cast<const u32x4 *>(qv) + (b * 2 + half);
                wu[u] = NT ? __builtin_nontemporal_load(pu) : *pu;
                wv[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
            }
            const uint64_t blkL = Sh::block(blk0, rr, lane), bL = blkL < nblocks ? blkL : 0;
            const float fsu = su[bL], fsv = sv[bL];
            asm volatile("" ::: "memory");                        // every load of the round precedes its stores (r may alias qu)
            const float su7 = div7(fsu), sv7 = div7(fsv * a);
            const bool fast = __all(times16_is_finite(su7) && times16_is_finite(sv7));
            const float fa = fast ? su7 * 16.0f : su7, fb = fast ? sv7 * 16.0f : sv7;
            float v[2][4][8], mb[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int src = 4 * (32 * u + (lane >> 1));
                const float cu = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(fa)));
                const float cv = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(fb)));
                const uint32_t xu[4] = {wu[u].x, wu[u].y, wu[u].z, wu[u].w}, xv[4] = {wv[u].x, wv[u].y, wv[u].z, wv[u].w};
                if (fast) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) {
                        float gu[8], gv[8];
                        unpack8_16th(xu[q4], gu);
                        unpack8_16th(xv[q4], gv);
#pragma unroll
                        for (int e = 0; e < 8; e++) v[u][q4][e] = __builtin_fmaf(gv[e], cv, gu[e] * cu);
                    }
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                        for (int e = 0; e < 8; e++) v[u][q4][e] = __builtin_fmaf((float)unpack1(xv[q4], e), cv, (float)unpack1(xu[q4], e) * cu);
                }
                float m = 0.0f;
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                    for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[u][q4][e]));
                mb[u] = fmaxf(m, __shfl_xor(m, 1));
            }
            const int from = 4 * (2 * (lane & 31));
            const float m0 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[0])));
            const float m1 = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(mb[1])));
            const float mL = fix_zero_max(lane < 32 ? m0 : m1);
            float kL = 7.0f / mL;
            kL = kL < __builtin_inff() ? kL : 0.0f;               // 7 / max overflows: fma(v, 0, noise) truncates to 0 -- every nibble 0, as quant_pack8's guard
            if (blkL < nblocks) sr[blkL] = mL;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int bl = 32 * u + (lane >> 1);
                const uint64_t blk = Sh::block(blk0, rr, bl);
                const float kq = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * bl, __float_as_int(kL)));
                // W[j] of draw 0 and draw 1 for this lane's words j = 4 half .. 4 half + 3
                const u32x4 *W4 = reinterpret_cast<const u32x4 *>(raw + (size_t)(bl * 2) * 4);
                const u32x4 Wa = W4[half], Wb = W4[2 + half];
                const uint32_t W0[4] = {Wa.x, Wa.y, Wa.z, Wa.w}, W1[4] = {Wb.x, Wb.y, Wb.z, Wb.w};
                uint32_t o[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    float n0[4], n1[4], nz[8];
                    noise4_of(W0[q4], n0);
                    noise4_of(W1[q4], n1);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int g = e ^ 1;
                        nz[e] = (g >> 2) ? n1[g & 3] : n0[g & 3];
                    }
                    o[q4] = quant_pack8(v[u][q4], kq, nz);
                }
                if (blk < nblocks) {
                    u32x4 *pr = reinterpret_cast<u32x4 *>(r) + (blk * 2 + half);
                    if (NT) __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, pr); else *pr = u32x4{o[0], o[1], o[2], o[3]};
                }
            }
        } else {
            // lane = 4 dwords (half a block), 32 blocks per step: 16-byte loads and stores
            constexpr int STEPS4 = Sh::NBR / 32 > 0 ? Sh::NBR / 32 : 1;
            const int half = lane & 1;
            u32x4 wu[STEPS4], wv[STEPS4];
            float fu[STEPS4], fv[STEPS4];
#pragma unroll
            for (int u = 0; u < STEPS4; u++) {
                const uint64_t blk = Sh::block(blk0, rr, 32 * u + (lane >> 1));
                const uint64_t b = blk < nblocks ? blk : 0;
                const u32x4 *pu = reinterpret_cast<const u32x4 *>(qu) + (b * 2 + half), *pv = reinterpret_cast<const u32x4 *>(qv) + (b * 2 + half);
                wu[u] = NT ? __builtin_nontemporal_load(pu) : *pu;              // NT: operands + result beyond the Infinity Cache stream past it
                wv[u] = NT ? __builtin_nontemporal_load(pv) : *pv;
                fu[u] = su[b];
                fv[u] = sv[b];
            }
#pragma unroll
            for (int u = 0; u < STEPS4; u++) {
                const int bl = 32 * u + (lane >> 1);
                const uint64_t blk = Sh::block(blk0, rr, bl);
                const float su7 = div7(fu[u]), sv7 = div7(fv[u] * a);
                float v[4][8];
                saa_values(wu[u].x, wv[u].x, su7, sv7, v[0]);
                saa_values(wu[u].y, wv[u].y, su7, sv7, v[1]);
                saa_values(wu[u].z, wv[u].z, su7, sv7, v[2]);
                saa_values(wu[u].w, wv[u].w, su7, sv7, v[3]);
                float m = 0.0f;
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
                    for (int e = 0; e < 8; e++) m = fmaxf(m, __builtin_fabsf(v[q4][e]));
                m = fmaxf(m, __shfl_xor(m, 1));
                m = fix_zero_max(m);
                const float kq = 7.0f / m;
                // W[j] of draw 0 and draw 1 for this lane's words j = 4 half .. 4 half + 3
                const u32x4 *W4 = reinterpret_cast<const u32x4 *>(raw + (size_t)(bl * 2) * 4);
                const u32x4 Wa = W4[half], Wb = W4[2 + half];
                const uint32_t W0[4] = {Wa.x, Wa.y, Wa.z, Wa.w}, W1[4] = {Wb.x, Wb.y, Wb.z, Wb.w};
                uint32_t o[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    float n0[4], n1[4], nz[8];
                    noise4_of(W0[q4], n0);
                    noise4_of(W1[q4], n1);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int g = e ^ 1;
                        nz[e] = (g >> 2) ? n1[g & 3] : n0[g & 3];
                    }
                    o[q4] = quant_pack8(v[q4], kq, nz);
                }
                if (blk < nblocks) {
                    u32x4 *pr = reinterpret_cast<u32x4 *>(r) + (blk * 2 + half);
                    if (NT) __builtin_nontemporal_store(u32x4{o[0], o[1], o[2], o[3]}, pr); else *pr = u32x4{o[0], o[1], o[2], o[3]};
                    if (half == 0) sr[blk] = m;
                }
            }
        }
        if (rr + 1 < Sh::ROUNDS) __syncthreads();
    }
}

#ifndef SAA_BLK_MIN_BLOCKS
#define SAA_BLK_MIN_BLOCKS 4096u      // n >= 2^18: the block-scalar kernel; below that the launch-bound sizes of the IHT / GD loops keep the plain one
#endif

extern "C" int clv4_scale_and_add(const int8_t *qu, const float *su, const int8_t *qv, const float *sv, float a, uint64_t n_pad,
                                  int8_t *r, float *sr, uint64_t *rng_state_dev, void *stream)
{
    CLV_REQUIRE(qu && su && qv && sv && r && sr, "clv4_scale_and_add: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0, "clv4_scale_and_add: n_pad=%llu is not a multiple of 128", (unsigned long long)n_pad);
    if (!n_pad) return CLV_OK;
    hipStream_t st = as_stream(stream);
    const uint64_t nwords = n_pad / 8, nb = n_pad / 64;
    if (!rng_state_dev && nb >= SAA_BLK_MIN_BLOCKS) {
        // one wave per 64 blocks, at most 8 workgroups per CU (grid-stride over the chunks)
        const uint64_t want = (nb + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
        const dim3 grid((unsigned)(want < cap ? want : cap));
        if (3 * (n_pad / 2) > (256ull << 20))
            hipLaunchKernelGGL(k_v4_scale_and_add_blk<true>, grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a, (u32x4 *)r, sr, nb);
        else
            hipLaunchKernelGGL(k_v4_scale_and_add_blk<false>, grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a, (u32x4 *)r, sr, nb);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (!rng_state_dev) {
        const uint64_t nquads = nwords / 4;
        const uint64_t want = (nquads + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
        const dim3 grid((unsigned)(want < cap ? want : cap));
        if (3 * (n_pad / 2) > (256ull << 20))           // operands + result exceed the Infinity Cache: stream past it
            hipLaunchKernelGGL((k_v4_scale_and_add<true, SAA_U>), grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a,
                               (u32x4 *)r, sr, nquads);
        else
            hipLaunchKernelGGL((k_v4_scale_and_add<false, 1>), grid, dim3(256), 0, st, (const u32x4 *)qu, su, (const u32x4 *)qv, sv, a,
                               (u32x4 *)r, sr, nquads);
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    RngTables T;
    int rc = clv_rng_tables(&T);
    if (rc) return rc;
    const uint64_t seq = clv_rng_seq_for(rng_state_dev, st);
#define SAA_LAUNCH(S)                                                                                                                  \
    hipLaunchKernelGGL(k_v4_scale_and_add_st<S>, dim3((unsigned)((nb + 32 * S - 1) / (32 * S))), dim3(256), 0, st, (const uint32_t *)qu, \
                       su, (const uint32_t *)qv, sv, a, (uint32_t *)r, sr, nb, rng_state_dev, seq, T)
    switch (clv_st_segments(nb, true)) {
    case 1: SAA_LAUNCH(1); break;
    case 4: SAA_LAUNCH(4); break;
    case 16: SAA_LAUNCH(16); break;
    default:
        if (3 * (n_pad / 2) > (256ull << 20))
            hipLaunchKernelGGL((k_v4_scale_and_add_st<64, true>), dim3((unsigned)((nb + 32 * 64 - 1) / (32 * 64))), dim3(256), 0, st,
                               (const uint32_t *)qu, su, (const uint32_t *)qv, sv, a, (uint32_t *)r, sr, nb, rng_state_dev, seq, T);
        else
            SAA_LAUNCH(64);
        break;
    }
#undef SAA_LAUNCH
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

// =================================================================================================
// f2  CloverMatrix4::transpose (CloverMatrix4.h:1549-1663): out(j,i) = in(i,j) nibble-wise, tile scales
//     transposed (the reference calls IPP for those, :1657-1658).
//     workgroup = 256 x 256 elements staged through LDS so that BOTH the reads and the writes are 128-byte
//     runs (a row of the tile is 128 B on either side); a thread transposes 8x8 nibble blocks in registers.
//     LDS rows are padded to 33 words: the 8x8-block reads (lanes = 8 words x 4 row groups) are conflict-free,
//     the writes 2-way (free for ds_write_b32).  Algorithmic bytes: 2 * (1/2 + 4/4096) per element.
// =================================================================================================
#define TR_T 256                      // tile edge in elements
#define TR_W (TR_T / 8)               // 32 words per tile row
#define TR_S (TR_W + 1)               // padded LDS row stride in words
#ifndef TR_BH
#define TR_BH 8                      // tiles per XCD block: BH x BW (4x8: 0.214 ms, 8x8: 0.20-0.21, 16x8: 0.21, 8x16: 0.22, 2x16: 0.23 at 32768^2)
#define TR_BW 8
#endif

__global__ __launch_bounds__(256) void k_m4_transpose(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t rows,
                                                      uint64_t cols, uint32_t *__restrict__ qt, float *__restrict__ st,
                                                      uint32_t tiles_x)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tr_lds[];
    uint32_t *tin = tr_lds;                       // [256][33]: the input tile, then (in place) the output tile
    uint32_t *tout = tr_lds;
    // Tile order: workgroups are dealt to the 8 XCDs round-robin; the workgroups that run together on one XCD take 4 x 8 blocks of
    // tiles, so that what goes through that L2 at one time is 1 KiB of every input row and 512 B of every output row, not 128 B.
    uint32_t bj = blockIdx.x % tiles_x;
    uint64_t bi = blockIdx.x / tiles_x;
    {
        const uint32_t ntiles = gridDim.x, tiles_y = ntiles / tiles_x;
        constexpr uint32_t BH = TR_BH, BW = TR_BW;
        if (ntiles % 8 == 0 && tiles_x % BW == 0 && tiles_y % BH == 0) {
            const uint32_t t = (blockIdx.x & 7) * (ntiles / 8) + (blockIdx.x >> 3);
            const uint32_t blk = t / (BH * BW), in = t % (BH * BW), bx = tiles_x / BW;
            bi = (uint64_t)(blk / bx) * BH + in / BW;
            bj = (blk % bx) * BW + in % BW;
        }
    }
    const int tid = threadIdx.x;
    const uint64_t wcols = cols / 8, wrows = rows / 8;
    const uint64_t r0 = bi * TR_T, c0w = (uint64_t)bj * TR_W;       // tile origin: row, word column

    // 1. global -> LDS: 8 lanes x 16 B per tile row
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int idx = tid + 256 * k, r = idx >> 3, c = idx & 7;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(q + (r0 + r) * wcols + c0w + 4 * c);      // (nt loads / stores: no difference, r4)
        uint32_t *d = tin + r * TR_S + 4 * c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // 2. 8x8 nibble blocks: block (bg, w) = rows 8bg..8bg+7, word w.  lanes: w_lo = tid&7, bg_lo = (tid>>3)&3.  All four blocks of
    //    a thread are read into registers before anything is written back: ONE tile buffer (33 KiB, four workgroups per CU
    //    instead of two with separate in / out buffers)
    uint32_t wd[4][8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int w = (tid & 7) + 8 * ((tid >> 5) & 3);
        const int bg = ((tid >> 3) & 3) + 4 * ((tid >> 7) + 2 * k);
#pragma unroll
        for (int r = 0; r < 8; r++) wd[k][r] = tin[(8 * bg + r) * TR_S + w];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int w = (tid & 7) + 8 * ((tid >> 5) & 3);
        const int bg = ((tid >> 3) & 3) + 4 * ((tid >> 7) + 2 * k);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            uint32_t acc = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) acc |= ((wd[k][r] >> nib_shift(e)) & 0xFu) << nib_shift(r);
            tout[(8 * w + e) * TR_S + bg] = acc;
        }
    }
    __syncthreads();
    // 3. LDS -> global: output tile row j (a column of the input tile) = 32 words
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int idx = tid + 256 * k, r = idx >> 3, c = idx & 7;
        const uint32_t *d = tout + r * TR_S + 4 * c;
        const u32x4 v = {d[0], d[1], d[2], d[3]};
        *reinterpret_cast<u32x4 *>(qt + ((uint64_t)bj * TR_T + r) * wrows + bi * TR_W + 4 * c) = v;
    }
    // tile scales: this 256x256 tile covers a 4x4 patch of the 64x64 scale grid
    if (tid < 16) {
        const uint64_t ti = bi * 4 + (tid >> 2), tj = (uint64_t)bj * 4 + (tid & 3);
        st[tj * (rows / 64) + ti] = s[ti * (cols / 64) + tj];
    }
}

// rows or cols not divisible by 256 (they are multiples of 128): one 64x64 tile per 64-thread workgroup
__global__ __launch_bounds__(64) void k_m4_transpose_small(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t rows,
                                                           uint64_t cols, uint32_t *__restrict__ qt, float *__restrict__ st,
                                                           uint32_t tiles_x)
{
    const uint32_t bj = blockIdx.x % tiles_x;
    const uint64_t bi = blockIdx.x / tiles_x;
    const int cb = threadIdx.x & 7, rb = threadIdx.x >> 3;
    const uint64_t wcols = cols / 8, wrows = rows / 8;       // words per row of in / out
    uint32_t w[8];
#pragma unroll
    for (int r = 0; r < 8; r++) w[r] = q[(bi * 64 + rb * 8 + r) * wcols + bj * 8 + cb];
    uint32_t o[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        uint32_t acc = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) acc |= ((w[r] >> nib_shift(e)) & 0xFu) << nib_shift(r);
        o[e] = acc;
    }
#pragma unroll
    for (int e = 0; e < 8; e++) qt[((uint64_t)bj * 64 + cb * 8 + e) * wrows + bi * 8 + rb] = o[e];
    if (threadIdx.x == 0) st[(uint64_t)bj * (rows / 64) + bi] = s[bi * tiles_x + bj];
}

extern "C" int clm4_transpose(const int8_t *q, const float *s, uint64_t rows, uint64_t cols, int8_t *qt, float *st, void *stream)
{
    CLV_REQUIRE(q && s && qt && st, "clm4_transpose: null pointer");
    CLV_REQUIRE(rows % 128 == 0 && cols % 128 == 0, "clm4_transpose: rows=%llu cols=%llu must be multiples of 128",
                (unsigned long long)rows, (unsigned long long)cols);
    CLV_REQUIRE(q != qt, "clm4_transpose: in-place transposition is not supported");
    if (!rows || !cols) return CLV_OK;
    if (rows % TR_T == 0 && cols % TR_T == 0) {
        const uint64_t tiles = (rows / TR_T) * (cols / TR_T);
        CLV_REQUIRE(tiles <= 0x7FFFFFFFull, "clm4_transpose: too many tiles");
        const size_t lds = TR_T * TR_S * sizeof(uint32_t);                     // 33 KiB
        hipLaunchKernelGGL(k_m4_transpose, dim3((unsigned)tiles), dim3(256), lds, as_stream(stream), (const uint32_t *)q, s, rows, cols,
                           (uint32_t *)qt, st, (uint32_t)(cols / TR_T));
    } else {
        const uint64_t tiles = (rows / 64) * (cols / 64);
        CLV_REQUIRE(tiles <= 0x7FFFFFFFull, "clm4_transpose: too many tiles");
        hipLaunchKernelGGL(k_m4_transpose_small, dim3((unsigned)tiles), dim3(64), 0, as_stream(stream), (const uint32_t *)q, s, rows, cols,
                           (uint32_t *)qt, st, (uint32_t)(cols / 64));
    }
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

// =================================================================================================
// f3  CloverVector4::threshold(K) (CloverVector4.h:1913-1975): keep the K largest |value| among the first n
//     elements, zero the other nibbles, scales untouched.  The reference walks a K-entry min-heap
//     sequentially (O(n log K)); here: exact radix select on the fp32 bit pattern of
//     |value| = |f32(s/7) * q| (12 + 12 + 8 bits, three histogram passes over 0.56 B/element), then
//     one pass that keeps everything above the K-th value and the lowest-index ties.
//     The kept multiset of magnitudes is identical to the reference's; WHICH of several equal magnitudes
//     survive is heap-order dependent there and lowest-index-first here.
// =================================================================================================
// inclusive scan over the 64 lanes of a wave with DPP moves (no LDS crossbar): Hillis-Steele inside each row of 16,
// then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
#define DPP_ADD(ctrl, row_mask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xF, false)
    DPP_ADD(0x111, 0xF);      // row_shr:1
    DPP_ADD(0x112, 0xF);      // row_shr:2
    DPP_ADD(0x114, 0xF);      // row_shr:4
    DPP_ADD(0x118, 0xF);      // row_shr:8
    DPP_ADD(0x142, 0xA);      // row_bcast:15 -> rows 1, 3
    DPP_ADD(0x143, 0xC);      // row_bcast:31 -> rows 2, 3
#undef DPP_ADD
    return v;
}

struct ThreshState {
    uint32_t prefix;      // selected high bits so far
    uint32_t remaining;   // how many elements still to take inside the selected bin
    uint32_t tau;         // final: bit pattern of the K-th largest magnitude
    uint32_t ties_keep;   // final: how many elements == tau survive
};

__device__ __forceinline__ uint32_t mag_key(uint32_t w, int e, float s7)
{
    return __float_as_uint(__builtin_fabsf(s7 * (float)unpack1(w, e)));
}

// The large-n kernels below serve both element widths.  BITS = 4: CloverVector4, 8 elements per word, |value| = |f32(s/7) * q|
// (CloverVector4::get, :206-209).  BITS = 8: CloverVector8 (same algorithm, CloverVector8.h:1680-1740), 4 elements per word,
// |value| = |f32((float)q * s) / 127| (CloverVector8::get, :137-140).
template <int BITS>
struct ThreshElems {
    static constexpr int EPW = 32 / BITS;                                   // elements per 32-bit word
    static constexpr int WPB = 64 / EPW;                                    // words per 64-element block
    __device__ static __forceinline__ uint32_t mask(int e) { return BITS == 4 ? 0xFu << nib_shift(e) : 0xFFu << (8 * e); }
    __device__ static __forceinline__ uint32_t key(uint32_t w, int e, float sc)
    {
        if (BITS == 4) return mag_key(w, e, div7(sc));
        const float q = (float)((int)(w << (24 - 8 * e)) >> 24);
        return __float_as_uint(__builtin_fabsf(div127(q * sc)));
    }
};

// level 0: bins = key >> 20 (4096); level 1: key>>20 == prefix, bins = (key >> 8) & 0xFFF; level 2:
// key>>8 == prefix, bins = key & 0xFF
template <int LEVEL, int BITS>
__global__ __launch_bounds__(256) void k_thresh_hist(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                     const ThreshState *__restrict__ ts, uint32_t *__restrict__ hist)
{
    typedef ThreshElems<BITS> E;
    __shared__ uint32_t lh[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t prefix = LEVEL ? ts->prefix : 0;
    const uint64_t nwords = (n + E::EPW - 1) / E::EPW;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
#pragma unroll
        for (int e = 0; e < E::EPW; e++) {
            if (i * E::EPW + e >= n) break;
            const uint32_t key = E::key(w, e, sc);
            if (LEVEL == 0) atomicAdd(&lh[key >> 20], 1u);
            else if (LEVEL == 1) { if ((key >> 20) == prefix) atomicAdd(&lh[(key >> 8) & 0xFFF], 1u); }
            else { if ((key >> 8) == prefix) atomicAdd(&lh[key & 0xFF], 1u); }
        }
    }
    __syncthreads();
    const int nb = LEVEL == 2 ? 256 : 4096;
    for (int i = threadIdx.x; i < nb; i += 256) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// one WG: find, from the top, the bin in which the cumulative count reaches `remaining`; leaves the histogram zeroed
template <int LEVEL>
__global__ __launch_bounds__(256) void k_thresh_select(uint32_t *__restrict__ hist, ThreshState *__restrict__ ts, uint32_t k)
{
    __shared__ uint32_t wsum[4];
    constexpr int nb = LEVEL == 2 ? 256 : 4096;
    constexpr int per = nb / 256;
    const int t = threadIdx.x;
    const int top = (255 - t) * per + per - 1;                   // thread t owns the t-th run of `per` bins from the top
    uint32_t bins[per], sum = 0;
#pragma unroll
    for (int i = 0; i < per; i++) { bins[i] = hist[top - i]; sum += bins[i]; }
    uint32_t v = wave_scan_incl(sum);
    if ((t & 63) == 63) wsum[t >> 6] = v;
    __syncthreads();
    for (int w = 0; w < (t >> 6); w++) v += wsum[w];
    const uint32_t need = LEVEL == 0 ? k : ts->remaining;
    if (v >= need && v - sum < need) {
        uint32_t above = v - sum;
        int i = 0;
#pragma unroll
        for (int j = 0; j < per - 1; j++)
            if (i == j && above + bins[j] < need) { above += bins[j]; i = j + 1; }
        const uint32_t bin = (uint32_t)(top - i);
        const uint32_t prev = LEVEL == 0 ? 0 : ts->prefix;
        const uint32_t prefix = LEVEL == 0 ? bin : (LEVEL == 1 ? (prev << 12) | bin : (prev << 8) | bin);
        ts->prefix = prefix;
        ts->remaining = need - above;
        if (LEVEL == 2) { ts->tau = prefix; ts->ties_keep = need - above; }
    }
#pragma unroll
    for (int i = 0; i < per; i++) hist[top - i] = 0;             // ready for the next level
}

// chunked (not grid-stride) so that index order = (block, thread, element)
#define TH_WORDS_PER_BLOCK 2048      // 8 words per thread

template <int BITS>
__global__ __launch_bounds__(256) void k_thresh_count_ties(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                           const ThreshState *__restrict__ ts, uint32_t *__restrict__ block_ties)
{
    typedef ThreshElems<BITS> E;
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint32_t tau = ts->tau;
    const uint64_t w0 = (uint64_t)blockIdx.x * TH_WORDS_PER_BLOCK + threadIdx.x * 8;
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) {
        const uint64_t i = w0 + k;
        if (i * E::EPW >= n) break;
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
#pragma unroll
        for (int e = 0; e < E::EPW; e++) if (i * E::EPW + e < n && E::key(w, e, sc) == tau) c++;
    }
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) block_ties[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(256) void k_thresh_scan(uint32_t *__restrict__ block_ties, uint32_t nblocks)
{
    // exclusive scan by one WG (nblocks is n / 16384: small)
    __shared__ uint32_t wsum[4];
    const int t = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += 256) {
        const uint32_t i = base + t;
        const uint32_t v = i < nblocks ? block_ties[i] : 0;
        uint32_t incl = wave_scan_incl(v);
        __syncthreads();                                         // wsum of the previous round has been read
        if ((t & 63) == 63) wsum[t >> 6] = incl;
        __syncthreads();
        for (int w = 0; w < (t >> 6); w++) incl += wsum[w];
        if (i < nblocks) block_ties[i] = carry + incl - v;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void k_thresh_apply(uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n,
                                                      const ThreshState *__restrict__ ts, const uint32_t *__restrict__ block_ties)
{
    typedef ThreshElems<BITS> E;
    __shared__ uint32_t tcnt[256];
    const uint32_t tau = ts->tau, keep = ts->ties_keep;
    const uint64_t w0 = (uint64_t)blockIdx.x * TH_WORDS_PER_BLOCK + threadIdx.x * 8;
    uint32_t words[8];
    uint32_t tie_mask[8];
    uint32_t c = 0;
    for (int k = 0; k < 8; k++) {
        const uint64_t i = w0 + k;
        words[k] = 0;
        tie_mask[k] = 0;
        if (i * E::EPW >= n) continue;
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
        uint32_t outw = 0;
#pragma unroll
        for (int e = 0; e < E::EPW; e++) {
            if (i * E::EPW + e >= n) continue;
            const uint32_t key = E::key(w, e, sc);
            if (key > tau) outw |= w & E::mask(e);
            else if (key == tau) { tie_mask[k] |= 1u << e; c++; }
        }
        words[k] = outw;
    }
    tcnt[threadIdx.x] = c;
    __syncthreads();
    // exclusive prefix of tie counts inside the block (Hillis-Steele over 256 threads)
    uint32_t incl = c;
    for (int o = 1; o < 256; o <<= 1) {
        const uint32_t add = threadIdx.x >= (unsigned)o ? tcnt[threadIdx.x - o] : 0;
        __syncthreads();
        incl += add;
        tcnt[threadIdx.x] = incl;
        __syncthreads();
    }
    uint32_t rank = block_ties[blockIdx.x] + incl - c;
    for (int k = 0; k < 8; k++) {
        const uint64_t i = w0 + k;
        if (i * E::EPW >= n) break;
        uint32_t outw = words[k];
        const uint32_t w = q[i];
#pragma unroll
        for (int e = 0; e < E::EPW; e++)
            if (tie_mask[k] & (1u << e)) {
                if (rank < keep) outw |= w & E::mask(e);
                rank++;
            }
        // elements at or beyond n (the padding) are left as they are, like the reference's loop to `length`
        const uint64_t first = i * E::EPW;
        if (first + E::EPW > n) {
            for (int e = 0; e < E::EPW; e++) if (first + e >= n) outw |= w & E::mask(e);
        }
        q[i] = outw;
    }
}

// ---- single-workgroup path (n_pad <= 131072: every IHT size the reference benchmarks): one launch, the vector,
// the histograms and the scans all live in LDS.  One workgroup on one CU is latency-bound code, so this kernel is
// written to execute few instructions and few barriers:
//  * A block holds at most 9 distinct magnitudes (|nibble| in 0..8 times one scale), so the selection runs over
//    9 * n/64 weighted CANDIDATES (block, |nibble|), weight = how many elements of the block carry that magnitude,
//    instead of over the n elements.
//  * Radix select with four 8-bit levels: a 256-bin histogram is scanned by four waves in one step, where the
//    4096-bin levels used before cost 16 serial, bank-conflicting LDS reads per thread and level.
//  * Words are handled whole (SWAR over the 8 nibbles): magnitudes, per-block cut-offs and tie masks are bit
//    operations on the 32-bit word; there is no per-element float work after the selection.
#define TS_THREADS 1024
#define TS_MAXW 16

__device__ __forceinline__ uint32_t cand_key(float s7, int m) { return __float_as_uint(__builtin_fabsf(s7 * (float)m)); }

// element e of a word sits in nibble e after the two nibbles of every byte are swapped (even elements are stored high)
__device__ __forceinline__ uint32_t swap_nibbles(uint32_t w) { return ((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu); }
// |two's complement nibble| for all 8 nibbles: 0..8, no carries between nibbles
__device__ __forceinline__ uint32_t abs_nibbles(uint32_t w)
{
    const uint32_t sgn = (w >> 3) & 0x11111111u;
    return (w ^ (sgn * 0xFu)) + sgn;
}
// bit 3 of every nibble whose value (0..8) is >= t, t in 0..9
__device__ __forceinline__ uint32_t ge_nibbles(uint32_t ab, uint32_t t)
{
    if (t == 0) return 0x88888888u;
    if (t > 8) return 0u;
    return (ab + (8u - t) * 0x11111111u) & 0x88888888u;
}
// bit 3 of each of the first `count` nibbles (count 0..8)
__device__ __forceinline__ uint32_t first_nibbles(uint32_t count) { return count >= 8 ? 0x88888888u : 0x88888888u & ((1u << (4 * count)) - 1u); }

// Bit-sliced magnitude counts of a FULL block (64 nibbles in 8 words).  A 4 x 4 bit transpose inside every nibble column turns four words
// into the four bit planes of their 32 nibbles (plane j, bit 4e + k = bit j of nibble e of word k); |v| is then taken on the planes
// (two's complement: a1 = v1 ^ (sign & v0), a2 = v2 ^ (sign & (v1 | v0)), a3 = sign & ~(v2 | v1 | v0), i.e. only for -8) and every
// magnitude 1..7 is one three-input boolean + one popcount per half block: ~1.5 VALU per element against ~5 for the nibble-by-nibble
// walk (round 3: the count pass ran at 2.7 TB/s, VALU-bound).  Element order inside the block does not matter for counts.
__device__ __forceinline__ void th4_planes(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t P[4])
{
    uint32_t t;
    t = ((x0 >> 1) ^ x1) & 0x55555555u; x1 ^= t; x0 ^= t << 1;
    t = ((x2 >> 1) ^ x3) & 0x55555555u; x3 ^= t; x2 ^= t << 1;
    t = ((x0 >> 2) ^ x2) & 0x33333333u; x2 ^= t; x0 ^= t << 2;
    t = ((x1 >> 2) ^ x3) & 0x33333333u; x3 ^= t; x1 ^= t << 2;
    P[0] = x0; P[1] = x1; P[2] = x2; P[3] = x3;
}

__device__ __forceinline__ unsigned long long th4_count_full_block(const uint32_t w[8])
{
    uint32_t A[2][4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t v[4];
        th4_planes(w[4 * h], w[4 * h + 1], w[4 * h + 2], w[4 * h + 3], v);
        const uint32_t low = v[1] | v[0];
        A[h][0] = v[0];
        A[h][1] = v[1] ^ (v[3] & v[0]);
        A[h][2] = v[2] ^ (v[3] & low);
        A[h][3] = v[3] & ~(v[2] | low);
    }
    uint32_t c[9], sum = 0;
#pragma unroll
    for (int m = 1; m <= 7; m++) {
        uint32_t n = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t b2 = (m & 4) ? A[h][2] : ~A[h][2], b1 = (m & 2) ? A[h][1] : ~A[h][1], b0 = (m & 1) ? A[h][0] : ~A[h][0];
            n += __popc(b2 & b1 & b0);
        }
        c[m] = n;
        sum += n;
    }
    c[8] = __popc(A[0][3]) + __popc(A[1][3]);
    c[0] = 64u - sum - c[8];
    // fields of 7 bits at bit 7 m: m = 0..3 in the low word, m = 4 straddles bit 32
    const uint32_t lo = c[0] | (c[1] << 7) | (c[2] << 14) | (c[3] << 21) | (c[4] << 28);
    const uint32_t hi = (c[4] >> 4) | (c[5] << 3) | (c[6] << 10) | (c[7] << 17) | (c[8] << 24);
    return ((unsigned long long)hi << 32) | lo;
}

// one word's contribution to its block's table: field m += #(|nibble| == m) over the first `valid` elements of the word
__device__ __forceinline__ unsigned long long th4_count_word(uint32_t w, uint32_t valid)
{
    const uint32_t ab = abs_nibbles(swap_nibbles(w));
    unsigned long long acc = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++)
        if (e < valid) acc += 1ull << (7u * ((ab >> (4 * e)) & 0xFu));
    return acc;
}

// Round 4: the same selection with a third of the synchronisation (measured with cycle stamps at N = 8192, round-3 kernel: 16.5 k cycles
// = load 1.6 k, counts 1.8 k, four levels 7.2 k, cut-offs 0.8 k, ties + scan 4.0 k, apply 0.6 k; about 20 barriers):
//  * W = words per thread is a template parameter (1 at N = 8192): the thread's words live in registers, nothing is staged in LDS, no
//    16-slot loop with dead slots;
//  * a thread computes the (key, weight) of its <= 9 W / 8 + 1 candidates ONCE, in registers, before the levels;
//  * a level is: the candidates' LDS atomics, ONE barrier, and then EVERY wave scans the 256 bins itself (4 bins per lane, one DPP scan,
//    one ballot) -- no second and third barrier to publish the selected bin;
//  * the per-block cut-offs are computed by each thread for its own words' blocks (9 products): no table, no barrier;
//  * the tie ranks need one barrier (16 wave totals), not two.
// Same keys, same selection, same lowest-index tie rule: results are bit-identical (all threshold and IHT tests).
template <int W>
__global__ __launch_bounds__(TS_THREADS) void k_thresh_small(uint32_t *__restrict__ q, const float *__restrict__ s, uint32_t n, uint32_t k)
{
    constexpr int MAXB = TS_THREADS * TS_MAXW / 8;                       // 2048 blocks at most
    constexpr int NC = (9 * W + 7) / 8 + 1;                              // candidates per thread at most
    __shared__ __attribute__((aligned(16))) uint32_t hist[4 * 256];
    __shared__ unsigned long long cnt[MAXB];                             // per block: 9 fields of 7 bits, field m = #(|nibble| == m)
    __shared__ float s7[MAXB];
    __shared__ uint32_t wtot[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nwords = (n + 7) / 8, nblocks = (n + 63) / 64;
    const uint32_t w0 = tid * W;
    uint32_t w[W];
#pragma unroll
    for (int j = 0; j < W; j++) w[j] = q[w0 + j < nwords ? w0 + j : 0];
    for (uint32_t i = tid; i < nblocks; i += TS_THREADS) { s7[i] = div7(s[i]); if (W < 8) cnt[i] = 0ull; }
    hist[tid] = 0;
    __syncthreads();                                                     // tables zero before anybody adds to them; s7 visible (also for k = 0)

    uint32_t tau = 0x7F800000u, keep = 0;
    if (k != 0) {
        // magnitude tables.  W >= 8: a thread holds whole blocks (plain stores, bit-sliced counts for full blocks); else LDS atomics
        if constexpr (W >= 8) {
#pragma unroll
            for (int g = 0; g < W / 8; g++) {
                const uint32_t b = (w0 >> 3) + g;
                if (b < nblocks) {
                    unsigned long long acc = 0;
                    if (64u * b + 64u <= n) {
                        acc = th4_count_full_block(&w[8 * g]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const uint32_t first = 64u * b + 8u * j;
                            acc += th4_count_word(w[8 * g + j], first >= n ? 0u : (n - first < 8 ? n - first : 8u));
                        }
                    }
                    cnt[b] = acc;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < W; j++) {
                const uint32_t i = w0 + j;
                if (i < nwords) atomicAdd(&cnt[i >> 3], th4_count_word(w[j], n - 8 * i < 8 ? n - 8 * i : 8u));
            }
        }
        __syncthreads();
        // this thread's candidates (block, magnitude): key and weight, once
        uint32_t ckey[NC], cwgt[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const uint32_t ci = tid + TS_THREADS * c;
            const bool in = ci < 9 * nblocks;
            const uint32_t b = in ? ci / 9u : 0u, m = ci - 9u * b;
            cwgt[c] = in ? (uint32_t)(cnt[b] >> (7 * m)) & 0x7Fu : 0u;
            ckey[c] = cand_key(s7[b], (int)m);
        }
        uint32_t prefix = 0, need = k;
#pragma unroll
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
            uint32_t *h = hist + 256 * level;
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (cwgt[c] && (level == 0 || (ckey[c] >> (shift + 8)) == prefix)) atomicAdd(&h[(ckey[c] >> shift) & 0xFFu], cwgt[c]);
            __syncthreads();
            // every wave selects for itself: lane l owns bins 255 - 4 l ... 252 - 4 l (from the top)
            const u32x4 h4 = *reinterpret_cast<const u32x4 *>(h + 252 - 4 * lane);
            const uint32_t t0 = h4.w, t1 = h4.z, t2 = h4.y, t3 = h4.x;
            const uint32_t sum = t0 + t1 + t2 + t3;
            const uint32_t incl = wave_scan_incl(sum);
            const unsigned long long hit = __ballot(incl >= need && incl - sum < need);
            const int L = __builtin_ctzll(hit);                          // exactly one lane: the level's total weight is >= need
            uint32_t above = __shfl(incl - sum, L);
            const uint32_t T0 = __shfl(t0, L), T1 = __shfl(t1, L), T2 = __shfl(t2, L);
            uint32_t pick = 0;
            if (above + T0 < need) { above += T0; pick = 1;
                if (above + T1 < need) { above += T1; pick = 2;
                    if (above + T2 < need) { above += T2; pick = 3; } } }
            prefix = (prefix << 8) | (255u - 4u * (uint32_t)L - pick);
            need -= above;
        }
        tau = prefix;
        keep = need;
    }
    // survivors above tau, ties, the rank of this thread's first tie (ties in index order: the first `keep` survive)
    uint32_t keepbits[W], tiebits[W], c = 0;
    uint32_t lo_t = 0, hi_t = 0, cur_b = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t i = w0 + j;
        keepbits[j] = tiebits[j] = 0;
        if (i < nwords) {
            if ((i >> 3) != cur_b) {                                     // per block: magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it
                cur_b = i >> 3;
                const float sc = s7[cur_b];
                lo_t = hi_t = 0;
#pragma unroll
                for (int m = 0; m <= 8; m++) {
                    const uint32_t key = cand_key(sc, m);
                    lo_t += key < tau;
                    hi_t += key <= tau;
                }
            }
            const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
            const uint32_t valid = first_nibbles(n - 8 * i < 8 ? n - 8 * i : 8);
            const uint32_t above = ge_nibbles(ab, hi_t);
            keepbits[j] = above | (0x88888888u & ~valid);                 // padding is left alone
            tiebits[j] = ge_nibbles(ab, lo_t) & ~above & valid;
            c += __popc(tiebits[j]);
        }
    }
    const uint32_t v = wave_scan_incl(c);
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    const uint32_t tot = lane < 16 ? wtot[lane] : 0;
    const uint32_t inc = wave_scan_incl(tot);
    uint32_t rank = v - c + __shfl(inc - tot, wave);
#pragma unroll
    for (int j = 0; j < W; j++) {
        const uint32_t i = w0 + j;
        if (i < nwords) {
            uint32_t t = tiebits[j], kb = keepbits[j];
            const uint32_t nt = __popc(t), room = keep > rank ? keep - rank : 0;
            if (room >= nt) kb |= t;
            else for (uint32_t r = 0; r < room; r++) { kb |= t & (0u - t); t &= t - 1; }
            rank += nt;
            const uint32_t full = (kb >> 3) * 0xFu;                       // bit 3 -> whole nibble, then back to the stored nibble order
            q[i] = w[j] & swap_nibbles(full);
        }
    }
}

// ---- single-workgroup path for CloverVector8 (n_pad <= 32768): element keys live in registers (8 words = 32 elements per
// thread), radix select in four 8-bit levels straight over the elements (a block has up to 128 distinct magnitudes, so the
// candidate trick of the 4-bit kernel does not pay), same DPP scans, same lowest-index tie rule.
#define TS8_MAXW 8
template <int TS8_W>          // words per thread, compile-time so that the unrolled loops carry no dead slots
__global__ __launch_bounds__(TS_THREADS) void k_thresh8_small(uint32_t *__restrict__ q, const float *__restrict__ s, uint32_t n, uint32_t k)
{
    typedef ThreshElems<8> E;
    constexpr int COPIES = 8;                      // private histograms by lane & 7: the keys of a vector crowd into a few bins
    __shared__ __attribute__((aligned(16))) uint32_t hist[4 * COPIES * 256];
    __shared__ __attribute__((aligned(16))) uint32_t hsum[2 * 256];    // per-level sums over the copies (two buffers: a fast wave may already
    __shared__ uint32_t wsum[16];                                       // write the next level's while a slow one still reads this level's)
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t nwords = (n + 3) / 4;
    constexpr uint32_t W = TS8_W;                                     // contiguous words per thread: index order = thread order
    const uint32_t w0 = tid * W;
    uint32_t words[TS8_W], keys[TS8_W][4];
    uint32_t valid = 0;                                                // bit 4j+e: element e of word j exists (index < n)
#pragma unroll
    for (uint32_t j = 0; j < TS8_W; j++) {
        const uint32_t i = w0 + j;
        const bool in = j < W && i < nwords;
        words[j] = in ? q[i] : 0u;
        const float sc = in ? s[i / E::WPB] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            keys[j][e] = E::key(words[j], e, sc);
            if (in && i * 4 + e < n) valid |= 1u << (4 * j + e);
        }
    }
#pragma unroll
    for (int i = 0; i < 4 * COPIES * 256 / TS_THREADS; i++) hist[tid + TS_THREADS * i] = 0;
    __syncthreads();

    uint32_t tau = 0x7F800000u, keep = 0;
    if (k != 0) {
        uint32_t prefix = 0, need = k;
        for (int level = 0; level < 4; level++) {
            const int shift = 24 - 8 * level;
            uint32_t *h = hist + COPIES * 256 * level;
            uint32_t *hp = h + 256 * (tid & (COPIES - 1));
#pragma unroll
            for (uint32_t j = 0; j < TS8_W; j++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if ((valid >> (4 * j + e)) & 1u) {
                        const uint32_t key = keys[j][e];
                        if (level == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hp[(key >> shift) & 0xFFu], 1u);
                    }
            __syncthreads();
            // the private copies are added up once (256 threads, one bin each), then every wave selects for itself from the sums: two
            // barriers per level instead of three.  (Letting every wave add up the eight copies itself saves one more barrier but reads
            // 128 KiB of LDS per level: measured slower, 8.0 against 7.5 us at N = 8192.)
            if (tid < 256) {
                uint32_t mine = 0;
#pragma unroll
                for (int cpy = 0; cpy < COPIES; cpy++) mine += h[256 * cpy + tid];
                hsum[256 * (level & 1) + tid] = mine;
            }
            __syncthreads();
            const u32x4 h4 = *reinterpret_cast<const u32x4 *>(hsum + 256 * (level & 1) + 252 - 4 * lane);       // lane l: bins 255 - 4 l ... 252 - 4 l
            const uint32_t t0 = h4.w, t1 = h4.z, t2 = h4.y, t3 = h4.x;
            const uint32_t sum = t0 + t1 + t2 + t3;
            const uint32_t incl = wave_scan_incl(sum);
            const unsigned long long hit = __ballot(incl >= need && incl - sum < need);
            const int L = __builtin_ctzll(hit);                          // exactly one lane: the level's total is >= need
            uint32_t above = __shfl(incl - sum, L);
            const uint32_t T0 = __shfl(t0, L), T1 = __shfl(t1, L), T2 = __shfl(t2, L);
            uint32_t pick = 0;
            if (above + T0 < need) { above += T0; pick = 1;
                if (above + T1 < need) { above += T1; pick = 2;
                    if (above + T2 < need) { above += T2; pick = 3; } } }
            prefix = (prefix << 8) | (255u - 4u * (uint32_t)L - pick);
            need -= above;
        }
        tau = prefix;
        keep = need;
    }
    uint32_t c = 0;
#pragma unroll
    for (uint32_t j = 0; j < TS8_W; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) c += ((valid >> (4 * j + e)) & 1u) && keys[j][e] == tau;
    // ties in index order: the first `keep` of them survive (one barrier: the 16 wave totals)
    const uint32_t vinc = wave_scan_incl(c);
    if (lane == 63) wsum[tid >> 6] = vinc;
    __syncthreads();
    const uint32_t tot = lane < 16 ? wsum[lane] : 0;
    const uint32_t inc = wave_scan_incl(tot);
    uint32_t rank = vinc - c + __shfl(inc - tot, tid >> 6);
#pragma unroll
    for (uint32_t j = 0; j < TS8_W; j++) {
        const uint32_t i = w0 + j;
        if (j < W && i < nwords) {
            uint32_t outw = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t byte = words[j] & E::mask(e);
                if (!((valid >> (4 * j + e)) & 1u)) { outw |= byte; continue; }          // padding is left alone
                if (keys[j][e] > tau) outw |= byte;
                else if (keys[j][e] == tau) { if (rank < keep) outw |= byte; rank++; }
            }
            q[i] = outw;
        }
    }
}

// ---- large CloverVector4 vectors: the same weighted-candidate idea as k_thresh_small, across kernels ----------------------------
// One pass turns every 64-element block into its 9 magnitude counts (8 bytes); the three radix levels then run over these
// tables (12 bytes per block instead of 36 bytes of elements, 9 weighted histogram updates per block instead of 64), the tie
// counts come from the tables as well, and only the final pass touches the elements again: 2 element passes instead of 5.
// Radix levels: 12 + 12 + 8 bits (LEVEL 0 = most significant).  The fixed cost per workgroup of a level (16 KiB of LDS bins to
// clear and to flush with global atomics, which serialise per address at ~50 ns) is what such a kernel costs, not the 12 bytes per
// block it reads.  (A variant that let the last workgroup to finish do the selection was 6x slower: its one-ticket-per-workgroup
// atomic serialises the same way.)
// elements of block b equal to tau, from its table
__device__ __forceinline__ uint32_t th4_block_ties(unsigned long long c, float s7, uint32_t tau)
{
    uint32_t t = 0;
#pragma unroll
    for (int m = 0; m <= 8; m++) t += cand_key(s7, m) == tau ? (uint32_t)(c >> (7 * m)) & 0x7Fu : 0u;
    return t;
}

// ---- round 2: the same algorithm in 6 launches instead of 10 -------------------------------------------------------------------
// The three one-workgroup select kernels and the one-workgroup scan are gone: every workgroup of a kernel that needs the outcome
// of an earlier level recomputes it from that level's finished histogram (16 KiB out of L2, one block scan), and the apply kernel
// adds up the tie counts in front of its chunk itself (group totals + prefixes inside a group).  Each level has its own
// histogram, zeroed by the first kernel.  Same keys, same selection, same tie rule: results are unchanged bit for bit.
struct Th4Sel {
    uint32_t prefix, remaining;
};

// one workgroup of 256 or more threads (the first 256 do the work, everybody takes the barriers): from the top of `hist`, the bin in
// which the cumulative count reaches `need`
template <int LEVEL>
__device__ __forceinline__ Th4Sel th4_wg_select(const uint32_t *__restrict__ hist, uint32_t need, uint32_t prev, uint32_t *sel /* LDS[2] */,
                                                uint32_t *wsum /* LDS[4] */)
{
    constexpr int nb = LEVEL == 2 ? 256 : 4096;
    constexpr int per = nb / 256;
    const int t = threadIdx.x;
    const bool on = t < 256;
    const int top = (255 - (on ? t : 0)) * per + per - 1;        // thread t owns the t-th run of `per` bins from the top
    uint32_t bins[per], sum = 0;
#pragma unroll
    for (int i = 0; i < per; i++) bins[i] = hist[top - i];      // unconditional (the idle threads read thread 0's bins): the loads go out together
#pragma unroll
    for (int i = 0; i < per; i++) { if (!on) bins[i] = 0u; sum += bins[i]; }
    uint32_t v = wave_scan_incl(sum);
    __syncthreads();                                             // sel / wsum of an earlier call have been read
    if (on && (t & 63) == 63) wsum[t >> 6] = v;
    __syncthreads();
    if (on) {
        for (int w = 0; w < (t >> 6); w++) v += wsum[w];
        if (v >= need && v - sum < need) {
            uint32_t above = v - sum;
            int i = 0;
#pragma unroll
            for (int j = 0; j < per - 1; j++)
                if (i == j && above + bins[j] < need) { above += bins[j]; i = j + 1; }
            const uint32_t bin = (uint32_t)(top - i);
            sel[0] = LEVEL == 0 ? bin : (LEVEL == 1 ? (prev << 12) | bin : (prev << 8) | bin);
            sel[1] = need - above;
        }
    }
    __syncthreads();
    return Th4Sel{sel[0], sel[1]};
}

// the selections of levels 0 .. UPTO-1 (what level UPTO's histogram, or the tie count, needs)
template <int UPTO>
__device__ __forceinline__ Th4Sel th4_selected(const uint32_t *__restrict__ hists, uint32_t k, uint32_t *sel, uint32_t *wsum)
{
    Th4Sel r{0, k};
    if (UPTO >= 1) r = th4_wg_select<0>(hists, k, 0, sel, wsum);
    if (UPTO >= 2) r = th4_wg_select<1>(hists + 4096, r.remaining, r.prefix, sel, wsum);
    if (UPTO >= 3) r = th4_wg_select<2>(hists + 8192, r.remaining, r.prefix, sel, wsum);
    return r;
}

// every block's 9 magnitude counts, and a clean slate for all three histograms
__global__ __launch_bounds__(256) void k_th4_count6(const u32x4 *__restrict__ q, uint64_t n, unsigned long long *__restrict__ cnt, uint64_t nblocks,
                                                    uint32_t *hists)
{
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 3 * 4096; i += 256) hists[i] = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += stride) {
        const u32x4 lo = q[2 * b], hi = q[2 * b + 1];
        const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (b * 64 + 64 <= n) {                                  // every block but possibly the last
            cnt[b] = th4_count_full_block(w);
            continue;
        }
        unsigned long long acc = 0;                              // the block that n cuts: element by element, the first n - 64 b of them
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
            const uint64_t first = b * 64 + 8 * j;
            const uint32_t valid = first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u);
#pragma unroll
            for (uint32_t e = 0; e < 8; e++)
                if (e < valid) acc += 1ull << (7u * ((ab >> (4 * e)) & 0xFu));
        }
        cnt[b] = acc;
    }
}

#ifndef TH4_HU
#define TH4_HU 4             // blocks per thread and step of a histogram level (8 with half the workgroups: 30-37 us per level at n = 2^28, this: 20-25)
#endif
#ifndef TH4_HWG_PER_CU_X2
#define TH4_HWG_PER_CU_X2 4  // histogram workgroups (of 1024 threads) per TWO CUs
#endif
// radix level LEVEL over the candidate tables, with the earlier levels' selections recomputed per workgroup
template <int LEVEL>
__global__ __launch_bounds__(1024) void k_th4_hist6(const unsigned long long *__restrict__ cnt, const float *__restrict__ s, uint64_t nblocks,
                                                    uint32_t *__restrict__ hists, uint32_t k)
{
    __shared__ uint32_t lh[4096];
    __shared__ uint32_t sel[2], wsum[4];
    constexpr int NB = LEVEL == 2 ? 256 : 4096;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) lh[i] = 0;
    const uint32_t prefix = th4_selected<LEVEL>(hists, k, sel, wsum).prefix;       // ends with a barrier: lh is clear for everybody
    if (LEVEL == 0) __syncthreads();
    // The 64 lanes of a wave hold 64 neighbouring blocks, whose scales -- hence whose keys for one magnitude m -- mostly fall into the
    // same bin: with every lane on the same m, one LDS atomic instruction is up to 64 updates of ONE address, which the LDS serialises.
    // So the lanes walk the nine magnitudes in rotated order, lane l starting at m = l mod 9: one instruction then spreads over nine
    // bins.  Sums are order-free: same histogram.
    const uint32_t m0 = (threadIdx.x & 63) % 9u;
    auto add_block = [&](unsigned long long c, float sc) {
        const float s7 = div7(sc);
        uint32_t m = m0;
#pragma unroll
        for (int it = 0; it <= 8; it++) {
            const uint32_t wgt = (uint32_t)(c >> (7 * m)) & 0x7Fu;
            if (wgt) {
                const uint32_t key = __float_as_uint(__builtin_fabsf(s7 * (float)m));        // cand_key(s7, m)
                if (LEVEL == 0) atomicAdd(&lh[key >> 20], wgt);
                else if (LEVEL == 1) { if ((key >> 20) == prefix) atomicAdd(&lh[(key >> 8) & 0xFFF], wgt); }
                else { if ((key >> 8) == prefix) atomicAdd(&lh[key & 0xFF], wgt); }
            }
            m = m == 8 ? 0 : m + 1;
        }
    };
    // TH4_HU blocks per thread and step, all loads first: a thread walks only a handful of steps, so the level is as long as its
    // chain of load latencies (a block whose table is all zero adds nothing: the tail needs no branch)
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblocks; b += TH4_HU * stride) {
        unsigned long long c[TH4_HU];
        float sc[TH4_HU];
#pragma unroll
        for (int u = 0; u < TH4_HU; u++) {
            const uint64_t bu = b + u * stride;
            const bool in = bu < nblocks;
            const uint64_t bc = in ? bu : b;                                  // clamped address, unconditional load
            c[u] = cnt[bc];
            sc[u] = s[bc];
            if (!in) c[u] = 0ull;
        }
#pragma unroll
        for (int u = 0; u < TH4_HU; u++) add_block(c[u], sc[u]);
    }
    __syncthreads();
    uint32_t *hist = hists + 4096 * LEVEL;
    for (int i = threadIdx.x; i < NB; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

#define TH4_GROUPS 512u      // workgroups of the tie kernel (and entries the apply kernel adds up at most); every one re-reads the
                             // three histograms (33 KiB) to recompute tau: 2048 groups were slower (68 MB of L2 reads) than 512
#define TH4_MAX_CPG 512u     // chunks per group at most: n < 2^32 -> fewer than 2^18 chunks of 256 blocks, over 512 groups

// tie counts.  Workgroup g takes the chunks (of 256 blocks = one workgroup of the apply kernel) [g * cpg, (g + 1) * cpg): it
// leaves the EXCLUSIVE prefix of every chunk inside its group in chunk_ties and the group's total in group_ties; workgroup 0
// also publishes tau and the number of ties to keep.  k == 0: nothing survives (tau beyond any magnitude).
__global__ __launch_bounds__(256) void k_th4_ties6(const unsigned long long *__restrict__ cnt, const float *__restrict__ s, uint64_t nblocks,
                                                   const uint32_t *__restrict__ hists, uint32_t k, ThreshState *__restrict__ ts,
                                                   uint32_t *__restrict__ chunk_ties, uint32_t *__restrict__ group_ties, uint32_t nchunks, uint32_t cpg)
{
    __shared__ uint32_t sel[2], wsum[4];
    Th4Sel r{0x7F800000u, 0};
    if (k != 0) r = th4_selected<3>(hists, k, sel, wsum);
    const uint32_t tau = r.prefix, keep = r.remaining;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ts = ThreshState{tau, keep, tau, keep};
    // chunk totals first -- a wave per chunk, four lane-steps of 64 blocks and one wave reduction, no barrier inside the loop -- then ONE
    // block scan over the group's (at most 512) totals.  (Round 3 took a workgroup scan and two barriers per chunk: 24 us at n = 2^28.)
    __shared__ uint32_t tot[TH4_MAX_CPG];
    const uint32_t c0 = blockIdx.x * cpg, c1 = c0 + cpg < nchunks ? c0 + cpg : nchunks;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t c = c0 + wave; c < c1; c += 8) {                        // two chunks per step: all sixteen loads first
        unsigned long long cc[2][4];
        float sc[2][4];
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) {                              // (an all-zero table has no ties: the tails need no branch)
                const uint64_t b = (uint64_t)(c + 4 * u) * 256 + 64 * q4 + lane;
                const bool in = c + 4 * u < c1 && b < nblocks;
                const uint64_t bc = in ? b : 0;                               // clamped address, unconditional load: all sixteen go out
                cc[u][q4] = cnt[bc];                                          // back to back (a predicated load becomes a branch + wait)
                sc[u][q4] = s[bc];
                if (!in) cc[u][q4] = 0ull;
            }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            uint32_t t = 0;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) t += th4_block_ties(cc[u][q4], div7(sc[u][q4]), tau);
            t = wave_scan_incl(t);
            if (lane == 63 && c + 4 * u < c1) tot[c + 4 * u - c0] = t;
        }
    }
    __syncthreads();
    const uint32_t i0 = 2 * threadIdx.x, n_here = c1 > c0 ? c1 - c0 : 0;
    const uint32_t a = i0 < n_here ? tot[i0] : 0u, b2 = i0 + 1 < n_here ? tot[i0 + 1] : 0u;
    uint32_t incl = wave_scan_incl(a + b2);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = incl - (a + b2);
    for (uint32_t w = 0; w < wave; w++) before += wsum[w];
    if (i0 < n_here) chunk_ties[c0 + i0] = before;
    if (i0 + 1 < n_here) chunk_ties[c0 + i0 + 1] = before + a;
    if (threadIdx.x == 0) group_ties[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// thread = one 64-element block: survivors above tau and the first `keep` ties in index order, whole words at a time; the rank of
// the chunk's first tie is computed here: ties of all groups in front + the chunk's prefix inside its group
__global__ __launch_bounds__(256) void k_th4_apply6(u32x4 *__restrict__ q, const float *__restrict__ s, uint64_t n, uint64_t nblocks,
                                                    const unsigned long long *__restrict__ cnt, const ThreshState *__restrict__ ts,
                                                    const uint32_t *__restrict__ chunk_ties, const uint32_t *__restrict__ group_ties, uint32_t cpg)
{
    __shared__ uint32_t wsum[4], gsum[4];
    const uint32_t tau = ts->tau, keep = ts->ties_keep;
    const uint32_t group = blockIdx.x / cpg;
    uint32_t before = 0;
    for (uint32_t g = threadIdx.x; g < group; g += 256) before += group_ties[g];
    before = wave_scan_incl(before);
    if ((threadIdx.x & 63) == 63) gsum[threadIdx.x >> 6] = before;
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = b < nblocks;
    const float s7 = in ? div7(s[b]) : 1.0f;
    const uint32_t mine = in ? th4_block_ties(cnt[b], s7, tau) : 0u;
    const uint32_t incl = wave_scan_incl(mine);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t rank = gsum[0] + gsum[1] + gsum[2] + gsum[3] + chunk_ties[blockIdx.x] + incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) rank += wsum[w];
    if (!in) return;
    uint32_t lo_t = 0, hi_t = 0;                                   // magnitudes >= hi_t are above tau, [lo_t, hi_t) equal it
#pragma unroll
    for (int m = 0; m <= 8; m++) {
        const uint32_t key = cand_key(s7, m);
        lo_t += key < tau;
        hi_t += key <= tau;
    }
    const u32x4 lo = q[2 * b], hi = q[2 * b + 1];
    uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t ab = abs_nibbles(swap_nibbles(w[j]));
        const uint64_t first = b * 64 + 8 * j;
        const uint32_t valid = first_nibbles(first >= n ? 0u : (n - first < 8 ? (uint32_t)(n - first) : 8u));
        const uint32_t above = ge_nibbles(ab, hi_t);
        uint32_t kb = above | (0x88888888u & ~valid);              // padding is left alone
        uint32_t t = ge_nibbles(ab, lo_t) & ~above & valid;
        const uint32_t nt = __popc(t), room = keep > rank ? keep - rank : 0;
        if (room >= nt) kb |= t;
        else for (uint32_t r = 0; r < room; r++) { kb |= t & (0u - t); t &= t - 1; }
        rank += nt;
        w[j] &= swap_nibbles((kb >> 3) * 0xFu);
    }
    q[2 * b] = u32x4{w[0], w[1], w[2], w[3]};
    q[2 * b + 1] = u32x4{w[4], w[5], w[6], w[7]};
}

// (TH4_GROUPS / TH4_MAX_CPG are defined in front of k_th4_ties6)

// workspace layout: [3 histograms of 4096 u32][ThreshState, 256 B][group_ties: 512 u32][chunk_ties: nblocks/256 + 1 u32, padded to
// 256 B][cnt: nblocks u64]
static int threshold4_large(uint32_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, hipStream_t st)
{
    const uint64_t nblocks = (n + 63) / 64;
    const uint32_t nchunks = (uint32_t)((nblocks + 255) / 256);
    uint32_t *hists = (uint32_t *)workspace;
    ThreshState *ts = (ThreshState *)(hists + 3 * 4096);
    uint32_t *group_ties = (uint32_t *)((char *)ts + 256);
    uint32_t *chunk_ties = group_ties + TH4_GROUPS;
    unsigned long long *cnt = (unsigned long long *)((char *)chunk_ties + (((uint64_t)(n_pad / 64 / 256 + 1) * 4 + 255) & ~255ull));
    const uint64_t want = (nblocks + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
    hipLaunchKernelGGL(k_th4_count6, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, st, (const u32x4 *)q, n, cnt, nblocks, hists);
    if (k != 0) {
        // few, fat workgroups: every workgroup clears 16 KiB of bins and flushes its non-empty ones with global atomics, and those land on
        // the SAME few dozen addresses from every workgroup (the keys of one vector cluster), where they serialise at ~50 ns each -- the
        // flush, not the table read, was what a level cost at n = 2^28 (round 3: 1024 workgroups of 256 threads, 38 us per 12-bit level).
        // So: as many THREADS as before but in workgroups of 1024 -- a quarter of the flushes per address -- one per 4096 blocks, at
        // most two per CU.
        // Below 2^20 blocks (n < 2^26) a level is all fixed cost and the round-3 shape -- 256-thread workgroups, one per 1024 blocks --
        // is the faster one (n = 2^24: 6-9 us per level either way, 41 us per call against 49).
        const bool fat = nblocks >= (1ull << 20);
        const uint64_t hwant = fat ? (nblocks + 4095) / 4096 : (nblocks + 1023) / 1024;
        const uint64_t hcap = fat ? (uint64_t)clv_cu_count() * TH4_HWG_PER_CU_X2 / 2 : (uint64_t)clv_cu_count() * 4;
        const dim3 grid((unsigned)(hwant < hcap ? hwant : hcap));
        const dim3 hthreads(fat ? 1024 : 256);
        hipLaunchKernelGGL(k_th4_hist6<0>, grid, hthreads, 0, st, cnt, s, nblocks, hists, (uint32_t)k);
        hipLaunchKernelGGL(k_th4_hist6<1>, grid, hthreads, 0, st, cnt, s, nblocks, hists, (uint32_t)k);
        hipLaunchKernelGGL(k_th4_hist6<2>, grid, hthreads, 0, st, cnt, s, nblocks, hists, (uint32_t)k);
    }
    const uint32_t cpg = (nchunks + TH4_GROUPS - 1) / TH4_GROUPS;
    const uint32_t groups = (nchunks + cpg - 1) / cpg;
    hipLaunchKernelGGL(k_th4_ties6, dim3(groups), dim3(256), 0, st, cnt, s, nblocks, hists, (uint32_t)k, ts, chunk_ties, group_ties, nchunks, cpg);
    hipLaunchKernelGGL(k_th4_apply6, dim3(nchunks), dim3(256), 0, st, (u32x4 *)q, s, n, nblocks, cnt, ts, chunk_ties, group_ties, cpg);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

template <int BITS>
static int threshold_large(uint32_t *q, const float *s, uint64_t n, uint64_t k, void *workspace, hipStream_t st)
{
    uint32_t *hist = (uint32_t *)workspace;
    ThreshState *ts = (ThreshState *)(hist + 4096);
    uint32_t *block_ties = (uint32_t *)((char *)ts + 256);
    const uint64_t nwords = (n + ThreshElems<BITS>::EPW - 1) / ThreshElems<BITS>::EPW;
    const uint32_t nblocks = (uint32_t)((nwords + TH_WORDS_PER_BLOCK - 1) / TH_WORDS_PER_BLOCK);
    CLV_HIP(hipMemsetAsync(hist, 0, 4096 * sizeof(uint32_t) + 256, st));
    if (k == 0) {
        // keep nothing: tau = +inf pattern beyond any finite magnitude, no ties kept
        const ThreshState none = {0, 0, 0x7F800000u, 0};
        CLV_HIP(hipMemcpyAsync(ts, &none, sizeof none, hipMemcpyHostToDevice, st));
        CLV_HIP(hipStreamSynchronize(st));
    } else {
        const uint64_t want = (nwords + 255) / 256, cap = (uint64_t)clv_cu_count() * 4;
        const dim3 grid((unsigned)(want < cap ? want : cap));
        hipLaunchKernelGGL((k_thresh_hist<0, BITS>), grid, dim3(256), 0, st, (const uint32_t *)q, s, n, ts, hist);
        hipLaunchKernelGGL(k_thresh_select<0>, dim3(1), dim3(256), 0, st, hist, ts, (uint32_t)k);
        hipLaunchKernelGGL((k_thresh_hist<1, BITS>), grid, dim3(256), 0, st, (const uint32_t *)q, s, n, ts, hist);
        hipLaunchKernelGGL(k_thresh_select<1>, dim3(1), dim3(256), 0, st, hist, ts, (uint32_t)k);
        hipLaunchKernelGGL((k_thresh_hist<2, BITS>), grid, dim3(256), 0, st, (const uint32_t *)q, s, n, ts, hist);
        hipLaunchKernelGGL(k_thresh_select<2>, dim3(1), dim3(256), 0, st, hist, ts, (uint32_t)k);
        CLV_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_thresh_count_ties<BITS>, dim3(nblocks), dim3(256), 0, st, (const uint32_t *)q, s, n, ts, block_ties);
    hipLaunchKernelGGL(k_thresh_scan, dim3(1), dim3(256), 0, st, block_ties, nblocks);
    hipLaunchKernelGGL(k_thresh_apply<BITS>, dim3(nblocks), dim3(256), 0, st, q, s, n, ts, block_ties);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" uint64_t clv4_threshold_workspace_bytes(uint64_t n_pad)
{
    const uint64_t chunk_bytes = ((n_pad / 64 / 256 + 1) * 4 + 255) & ~255ull;
    return 3 * 4096 * sizeof(uint32_t) + 256 + TH4_GROUPS * sizeof(uint32_t) + chunk_bytes + (n_pad / 64) * sizeof(unsigned long long) + 256;
}

extern "C" uint64_t clv8_threshold_workspace_bytes(uint64_t n_pad)
{
    const uint64_t blocks = (n_pad / 4 + TH_WORDS_PER_BLOCK - 1) / TH_WORDS_PER_BLOCK;
    return 4096 * sizeof(uint32_t) + 256 + blocks * sizeof(uint32_t) + 256;
}

extern "C" int clv4_threshold(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, void *stream)
{
    CLV_REQUIRE(q && s, "clv4_threshold: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv4_threshold: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv4_threshold: vectors of 2^32 or more elements are not supported");
    hipStream_t st = as_stream(stream);
    if (k >= n || n == 0) return CLV_OK;                       // everything survives
    if (n_pad <= (uint64_t)TS_THREADS * TS_MAXW * 8) {
        const uint64_t wpt = ((n + 7) / 8 + TS_THREADS - 1) / TS_THREADS;         // words per thread
#define T4_LAUNCH(W) hipLaunchKernelGGL(k_thresh_small<W>, dim3(1), dim3(TS_THREADS), 0, st, (uint32_t *)q, s, (uint32_t)n, (uint32_t)k)
        // (fewer, fatter threads -- W = 8 so that every block is one thread's and the tables need no atomics -- were measured slower at
        //  N = 8192 and 32768: 7.7 / 12.4 us against 5.4 / 10.9)
        if (wpt <= 1) T4_LAUNCH(1);
        else if (wpt <= 2) T4_LAUNCH(2);
        else if (wpt <= 4) T4_LAUNCH(4);
        else if (wpt <= 8) T4_LAUNCH(8);
        else T4_LAUNCH(16);
#undef T4_LAUNCH
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (!workspace) {
        int rc = clv_internal_workspace(&workspace, clv4_threshold_workspace_bytes(n_pad), as_stream(stream));
        if (rc) return rc;
    }
    return threshold4_large((uint32_t *)q, s, n, n_pad, k, workspace, st);
}

// =================================================================================================
// f3'  threshold in the REFERENCE's survivor order (CLV_THRESHOLD_REFERENCE): the min-heap walk of CloverVector4.h:1927-1972 /
//      CloverVector8.h:1680-1740 with the heap helpers of CloverBase.h:208-249 (std::make_heap under gt_idx_t = libstdc++'s bottom-up
//      __adjust_heap; min_heapify with left-first ties) reproduced step by step.  Which of several EQUAL magnitudes survive is decided by
//      where they sit in the heap when a larger value arrives, i.e. by the whole history: the walk is sequential by definition, like the
//      16 fma chains of dot EXACT.  One wavefront runs it: the heap lives in LDS (k <= 20000 entries of {value, index}; beyond that its top
//      14 levels, the deeper entries in global memory); the stream of the n - k later elements is taken 64 at a time and a ballot against the current root skips every
//      chunk -- or chunk remainder -- that cannot enter the heap (the root only grows), so only the inserts cost a sift (five heap
//      levels per LDS round trip, see k_thr_ref_walk).  Around it: a parallel pass that writes |value| per element
//      (the same expression as CloverVector4::get / CloverVector8::get) and a parallel pass that clears every nibble / byte whose index
//      is not in the final heap.  Cost: ~0.4 us per insert: N = 8192, K = 1024 0.85 ms (round 4: 3.4).  The C ABI's default is the radix
//      select above; the C++ containers default to this mode (clover_device.h: the exactness switch).
// =================================================================================================
template <int BITS>
__global__ __launch_bounds__(256) void k_thr_ref_keys(const uint32_t *__restrict__ q, const float *__restrict__ s, uint64_t n, float *__restrict__ vals,
                                                      uint32_t *__restrict__ keep, uint64_t keep_words)
{
    typedef ThreshElems<BITS> E;
    const uint64_t nwords = (n + E::EPW - 1) / E::EPW, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint32_t w = q[i];
        const float sc = s[i / E::WPB];
#pragma unroll
        for (int e = 0; e < E::EPW; e++)
            if (i * E::EPW + e < n) vals[i * E::EPW + e] = __uint_as_float(E::key(w, e, sc));
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < keep_words; i += stride) keep[i] = 0;
}

#define THR_LDS_ENTRIES 16384u      // the LDS part of a heap that does not fit: the top 14 levels
#define THR_LDS_WHOLE_MAX 20000u    // largest k whose heap (+ sentinel) lives in LDS whole: 160 008 of the CU's 160 KiB
// the heap's LDS part: the whole heap + its sentinel when k <= THR_LDS_WHOLE_MAX (IN_LDS), else entries 0 .. THR_LDS_ENTRIES - 1 -- the top 14
// levels, where every sift spends two of its rounds -- with the deeper entries in global memory (r5: all of it was global before, 2.3 x slower)
extern __shared__ __attribute__((aligned(16))) uint2 thr_lheap[];

struct ThrHeap {            // heap storage: LDS (ds_read / ds_write), beyond its capacity global memory read past the vector L1
    // global-memory entries: a lane's store and ANOTHER lane's later load of the same entry are ordered by the memory model, not by the
    // in-order issue of one wavefront: release / acquire at wavefront scope (no instruction on gfx950 beyond a wait for the
    // store; the LDS part is ordered by the wave's own lgkmcnt waits)
    template <bool IN_LDS> __device__ static __forceinline__ void wave_fence()
    {
        if (!IN_LDS) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    template <bool IN_LDS> __device__ static __forceinline__ uint2 ld(const uint2 *g, unsigned long long i)
    {
        if (IN_LDS || i < THR_LDS_ENTRIES) return thr_lheap[i];
        const unsigned long long v = __hip_atomic_load((const unsigned long long *)(g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    }
    // this lane's own store, no fence (make_heap: a lane works inside its own subtree; wave_fence() separates the levels)
    template <bool IN_LDS> __device__ static __forceinline__ void st1(uint2 *g, unsigned long long i, uint2 v)
    {
        if (IN_LDS || i < THR_LDS_ENTRIES) thr_lheap[i] = v;
        else __hip_atomic_store((unsigned long long *)(g + i), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // every lane with `on` stores its own entry at its own index (the moves of one sift step); then every lane may read them
    template <bool IN_LDS> __device__ static __forceinline__ void st_lanes(uint2 *g, bool on, unsigned long long i, uint2 v)
    {
        if (on) st1<IN_LDS>(g, i, v);
        wave_fence<IN_LDS>();
    }
    // one wave-uniform entry, written by lane 0
    template <bool IN_LDS> __device__ static __forceinline__ void st(uint2 *g, unsigned long long i, uint2 v) { st_lanes<IN_LDS>(g, threadIdx.x == 0, i, v); }
};
#define THR_VAL(e) __uint_as_float((e).x)
// gt_idx_t (CloverBase.h:216-218): (a.value > b.value) || isnan(a.value).  The NaN clause decides only where a NaN magnitude (a block
// scale that is NaN, or infinite over a zero nibble) sits in the initial heap -- kept so that the walk is the reference's for every input
#define THR_GT(a, b) ((THR_VAL(a) > THR_VAL(b)) || THR_VAL(a) != THR_VAL(a))

// bit `lane` and the bits of its ancestors inside a 62-node subtree laid out as lane = 2^depth - 2 + offset (depth 1 .. 5): the lanes
// that must ALL have been chosen for `lane` to lie on the sift path
__device__ __forceinline__ unsigned long long thr_ancestors(uint32_t lane)
{
    if (lane >= 62) return ~0ull;                                       // never satisfied: only bits 0 .. 61 are ever set
    unsigned long long anc = 0;
    for (uint32_t l = lane;; l = (l - 2) >> 1) {
        anc |= 1ull << l;
        if (l < 2) break;
    }
    return anc;
}

template <bool IN_LDS>
__global__ __launch_bounds__(64) void k_thr_ref_walk(const float *__restrict__ vals, uint32_t n, uint32_t k, uint2 *__restrict__ gheap,
                                                     uint32_t *__restrict__ keep)
{
    // heap indices: 32 bits while the heap fits LDS (k <= 20000: (pos + 1) << 5 stays small), 64 when part of it is in global memory
    typedef typename std::conditional<IN_LDS, uint32_t, unsigned long long>::type hidx_t;
    uint2 *h = gheap;                                                   // the global part (unused when the heap fits LDS)
    const uint32_t lane = threadIdx.x;
    // "Copy the first K-elements" (CloverVector4.h:1933-1940); entry k is a sentinel (+inf: never smaller than anything) that every
    // fetch beyond the heap is clamped to
    for (uint32_t i = lane; i <= k; i += 64) {
        ThrHeap::st1<IN_LDS>(h, i, i < k ? make_uint2(__float_as_uint(vals[i]), i) : make_uint2(0x7F800000u, 0xFFFFFFFFu));
    }
    ThrHeap::wave_fence<IN_LDS>();
    __syncthreads();
    // std::make_heap(min_heap, min_heap + k, gt_idx_t) (:1944): libstdc++ __make_heap = __adjust_heap(first, parent, len, value) for
    // parent = (len - 2) / 2 ... 0, comp = gt_idx_t.  An adjust touches the subtree under its parent only, and the parents of ONE level have
    // disjoint subtrees: their adjusts commute, so a level is done by all lanes at once (lane = parent, the scalar algorithm per lane) and
    // the levels follow each other bottom-up -- the same heap as the sequential order, entry for entry (r5; one parent at a time before:
    // 0.4 ms at k = 1024).
    if (k >= 2) {
        const uint32_t last_parent = (k - 2) / 2;
        for (int level = 31 - __builtin_clz(last_parent + 1); level >= 0; level--) {
            const uint32_t lo = (1u << level) - 1u, hi = (2u << level) - 2u < last_parent ? (2u << level) - 2u : last_parent;
            for (uint32_t parent = lo + lane; parent <= hi; parent += 64) {
                const uint2 v = ThrHeap::ld<IN_LDS>(h, parent);
                const uint32_t top = parent;
                uint32_t hole = parent, child = parent;
                while (child < (k - 1) / 2) {
                    child = 2 * (child + 1);
                    const uint2 a = ThrHeap::ld<IN_LDS>(h, child);
                    const uint2 b = ThrHeap::ld<IN_LDS>(h, child - 1);
                    if (THR_GT(a, b)) {                                        // comp(first + child, first + (child - 1))
                        child--;
                        ThrHeap::st1<IN_LDS>(h, hole, b);
                    } else {
                        ThrHeap::st1<IN_LDS>(h, hole, a);
                    }
                    hole = child;
                }
                if ((k & 1) == 0 && child == (k - 2) / 2) {
                    child = 2 * (child + 1);
                    ThrHeap::st1<IN_LDS>(h, hole, ThrHeap::ld<IN_LDS>(h, child - 1));
                    hole = child - 1;
                }
                while (hole > top) {                                           // __push_heap
                    const uint32_t par = (hole - 1) / 2;
                    const uint2 pe = ThrHeap::ld<IN_LDS>(h, par);
                    if (!THR_GT(pe, v)) break;                                 // comp(first + parent, value)
                    ThrHeap::st1<IN_LDS>(h, hole, pe);
                    hole = par;
                }
                ThrHeap::st1<IN_LDS>(h, hole, v);
            }
            ThrHeap::wave_fence<IN_LDS>();                                     // the next level reads what other lanes wrote in this one
        }
    }
    // the walk over elements k ... n-1 (:1952-1962): strictly larger than the root -> replace the root, min_heapify(0)
    float root = THR_VAL(ThrHeap::ld<IN_LDS>(h, 0));
    float vnext = (k + lane < n) ? vals[k + lane] : -1.0f;
    // min_heapify(heap, 0, k) with heap[0] = m (CloverBase.h:226-249): the smaller child moves up while it is smaller than m, the LEFT
    // child on equal children.  FIVE levels per memory round trip (r5; one pair of dependent round trips PER level in round 4): the 62
    // descendants of `pos` down to depth 5 are fetched by 62 lanes at once -- lane = 2^depth - 2 + offset, so a left child sits on an even
    // lane with its right sibling beside it; indices beyond the heap are clamped to the +inf sentinel.  What a sibling pair decides depends
    // on the pair and on m only (a < m, b < a, b < m), never on the path: three compares in all lanes give every pair's verdict, a
    // handful of mask operations the set S of chosen children, and a lane lies on the sift path iff it and all its ancestors are in S
    // (one masked compare against a per-lane constant).  The entries on the path move up one level in ONE masked store.  A wave issues
    // one instruction per ~4 cycles whatever it is, so the instruction count of this loop IS its time: ~45 per round.
    const uint32_t ld2 = lane + 2, dep = lane < 62 ? 31u - (uint32_t)__builtin_clz(ld2) : 0u, off = lane < 62 ? ld2 - (1u << dep) : 0u;
    const unsigned long long anc = thr_ancestors(lane);
    const unsigned long long LEFT = 0x1555555555555555ull;                 // even lanes 0 .. 60: the left children
    for (uint32_t base = k; base < n; base += 64) {
        const float v = vnext;
        const uint32_t nb = base + 64 + lane;
        vnext = (base + 64 < n && nb < n) ? vals[nb] : -1.0f;             // |value| >= 0: -1 never enters
        unsigned long long mask = __ballot(v > root);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            const uint2 m = make_uint2((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j), base + (uint32_t)j);
            const float mv = THR_VAL(m);
            hidx_t pos = 0;
            float new_root = mv;
            for (;;) {
                hidx_t mine = ((pos + 1) << dep) - 1 + off;              // lanes 62, 63 fetch `pos` itself: harmless, never on a path
                mine = mine < (hidx_t)k ? mine : (hidx_t)k;
                const uint2 e = ThrHeap::ld<IN_LDS>(h, mine);
                const float a = THR_VAL(e);
                const float b = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e.x, 0xB1, 0xF, 0xF, false));     // lane ^ 1
                const unsigned long long alm = __ballot(a < mv), bla = __ballot(b < a), blm = __ballot(b < mv);
                const unsigned long long r = LEFT & ((alm & bla) | (~alm & blm));      // pairs whose RIGHT child moves up
                const unsigned long long S = (LEFT & alm & ~r) | (r << 1);             // the chosen child of every pair that has one
                const bool on_path = (S & anc) == anc;
                const unsigned long long path = __ballot(on_path);
                if (path == 0) break;
                const int last = 63 - __builtin_clzll(path);
                if (pos == 0) new_root = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)e.x, __builtin_ctzll(path)));
                ThrHeap::st_lanes<IN_LDS>(h, on_path, (mine - 1) >> 1, e);
                if (IN_LDS) pos = (hidx_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, last);
                else pos = (hidx_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)mine >> 32), last) << 32) |
                                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, last));
                if (last < 30) break;                                        // the path ended above depth 5: m stops at `pos`
            }
            ThrHeap::st<IN_LDS>(h, pos, m);
            root = new_root;
            mask = __ballot(v > root) & ~((2ull << j) - 1ull);             // later lanes of this chunk, against the new root
        }
    }
    __syncthreads();
    // "Only copy the max K elements" (:1966-1969): the indices left in the heap survive
    for (uint32_t i = lane; i < k; i += 64) {
        const uint32_t idx = ThrHeap::ld<IN_LDS>(h, i).y;
        atomicOr(&keep[idx >> 5], 1u << (idx & 31));
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void k_thr_ref_apply(uint32_t *__restrict__ q, uint64_t n, const uint32_t *__restrict__ keep)
{
    typedef ThreshElems<BITS> E;
    const uint64_t nwords = (n + E::EPW - 1) / E::EPW, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint64_t e0 = i * E::EPW;
        const uint32_t bits = keep[e0 >> 5] >> (e0 & 31);                 // EPW divides 32: the word's bits sit in one bitmap word
        uint32_t w = q[i];
#pragma unroll
        for (int e = 0; e < E::EPW; e++)
            if (e0 + e < n && !((bits >> e) & 1u)) w &= ~E::mask(e);       // setBits(i, 0) for i < length only (:1939, 1961)
        q[i] = w;
    }
}

extern "C" uint64_t clv_threshold_reference_workspace_bytes(uint64_t n_pad)
{
    // [|value| per element: 4 n][survivor bitmap: n / 8][heap entries beyond LDS when k > 20000: 8 n] + alignment slack
    return n_pad * 4 + ((n_pad / 8 + 255) & ~255ull) + n_pad * 8 + 512;
}

template <int BITS>
static int threshold_reference(uint32_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, hipStream_t st)
{
    // the walk steps a 32-bit element index 64 at a time (k_thr_ref_walk): base + 64 must not wrap
    CLV_REQUIRE(n <= 0xFFFFFFFFull - 64, "threshold (reference order): n=%llu, at most 2^32 - 65 elements", (unsigned long long)n);
    if (!workspace) {
        int rc = clv_internal_workspace(&workspace, clv_threshold_reference_workspace_bytes(n_pad), st);
        if (rc) return rc;
    }
    float *vals = (float *)workspace;
    uint32_t *keep = (uint32_t *)((char *)workspace + n_pad * 4);
    uint2 *gheap = (uint2 *)((char *)keep + ((n_pad / 8 + 255) & ~255ull));
    const uint64_t keep_words = (n + 31) / 32;
    const uint64_t nwords = (n + ThreshElems<BITS>::EPW - 1) / ThreshElems<BITS>::EPW;
    const uint64_t want = (nwords + 255) / 256, cap = (uint64_t)clv_cu_count() * 8;
    const dim3 grid((unsigned)(want < cap ? want : cap));
    hipLaunchKernelGGL(k_thr_ref_keys<BITS>, grid, dim3(256), 0, st, (const uint32_t *)q, s, n, vals, keep, keep_words);
    if (k == 0) {
        // nothing survives (the oracle's reading of k = 0): the bitmap the keys pass has just cleared goes to the apply pass as it is,
        // which clears the first n elements and leaves the padding alone
    } else if (k <= THR_LDS_WHOLE_MAX) {
        const size_t lds = ((size_t)k + 1) * sizeof(uint2);                                  // + the sentinel entry
        if (lds > 64 * 1024) CLV_HIP(hipFuncSetAttribute((const void *)k_thr_ref_walk<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_thr_ref_walk<true>, dim3(1), dim3(64), lds, st, vals, (uint32_t)n, (uint32_t)k, gheap, keep);
    } else {
        const size_t lds = (size_t)THR_LDS_ENTRIES * sizeof(uint2);                          // the top 14 levels; the rest in the workspace
        CLV_HIP(hipFuncSetAttribute((const void *)k_thr_ref_walk<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_thr_ref_walk<false>, dim3(1), dim3(64), lds, st, vals, (uint32_t)n, (uint32_t)k, gheap, keep);
    }
    hipLaunchKernelGGL(k_thr_ref_apply<BITS>, grid, dim3(256), 0, st, q, n, keep);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv4_threshold_mode(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, int mode, void *workspace, void *stream)
{
    if (mode == CLV_THRESHOLD_FAST) return clv4_threshold(q, s, n, n_pad, k, workspace, stream);
    CLV_REQUIRE(mode == CLV_THRESHOLD_REFERENCE, "clv4_threshold_mode: unknown mode %d", mode);
    CLV_REQUIRE(q && s, "clv4_threshold_mode: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv4_threshold_mode: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv4_threshold_mode: vectors of 2^32 or more elements are not supported");
    if (k >= n || n == 0) return CLV_OK;
    return threshold_reference<4>((uint32_t *)q, s, n, n_pad, k, workspace, as_stream(stream));
}

// CloverVector8::threshold(K) (CloverVector8.h:1680-1740): same algorithm and tie rule on |q * scale / 127|
extern "C" int clv8_threshold(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, void *workspace, void *stream)
{
    CLV_REQUIRE(q && s, "clv8_threshold: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv8_threshold: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv8_threshold: vectors of 2^32 or more elements are not supported");
    hipStream_t st = as_stream(stream);
    if (k >= n || n == 0) return CLV_OK;
    if (n_pad <= (uint64_t)TS_THREADS * TS8_MAXW * 4) {
        const uint64_t w = ((n + 3) / 4 + TS_THREADS - 1) / TS_THREADS;
#define T8_LAUNCH(W) hipLaunchKernelGGL(k_thresh8_small<W>, dim3(1), dim3(TS_THREADS), 0, st, (uint32_t *)q, s, (uint32_t)n, (uint32_t)k)
        if (w <= 1) T8_LAUNCH(1);
        else if (w <= 2) T8_LAUNCH(2);
        else if (w <= 4) T8_LAUNCH(4);
        else T8_LAUNCH(8);
#undef T8_LAUNCH
        CLV_LAUNCH_CHECK();
        return CLV_OK;
    }
    if (!workspace) {
        int rc = clv_internal_workspace(&workspace, clv8_threshold_workspace_bytes(n_pad), as_stream(stream));
        if (rc) return rc;
    }
    return threshold_large<8>((uint32_t *)q, s, n, k, workspace, st);
}

extern "C" int clv8_threshold_mode(int8_t *q, const float *s, uint64_t n, uint64_t n_pad, uint64_t k, int mode, void *workspace, void *stream)
{
    if (mode == CLV_THRESHOLD_FAST) return clv8_threshold(q, s, n, n_pad, k, workspace, stream);
    CLV_REQUIRE(mode == CLV_THRESHOLD_REFERENCE, "clv8_threshold_mode: unknown mode %d", mode);
    CLV_REQUIRE(q && s, "clv8_threshold_mode: null pointer");
    CLV_REQUIRE(n_pad % 128 == 0 && n <= n_pad, "clv8_threshold_mode: n=%llu n_pad=%llu", (unsigned long long)n, (unsigned long long)n_pad);
    CLV_REQUIRE(n < (1ull << 32), "clv8_threshold_mode: vectors of 2^32 or more elements are not supported");
    if (k >= n || n == 0) return CLV_OK;
    return threshold_reference<8>((uint32_t *)q, s, n, n_pad, k, workspace, as_stream(stream));
}

// =================================================================================================
// f4  The application loops that call the hot path: quantized Iterative Hard Thresholding / Gradient Descent
//     (test/performance/01_measure.h:923-946, 999-1021).  One call enqueues all iterations on the stream; nothing
//     returns to the host in between.
// =================================================================================================
__global__ void k_v4_clear(uint32_t *q, float *s, uint64_t nwords, uint64_t nblocks)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) q[i] = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nblocks; i += stride) s[i] = 1.0f;
}

static int iht_iteration(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n,
                         int8_t *x, float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2,
                         float *st2, int8_t *t3, float *st3, uint64_t K, float mu, int threshold, uint64_t *rng, void *stream)
{
    // each scaleAndAdd rides in the epilogue of the mvm before it (same bits, same XORShift positions): 3 launches, not 5
    int rc = clm4_mvm_scale_and_add(Phi, sPhi, m, n, x, sx, y, sy, -1.0f, t1, st1, t2, st2, rng, stream);     // t1 = Phi * x; t2 = y - t1
    if (!rc) rc = clm4_mvm_scale_and_add(PhiT, sPhiT, n, m, t2, st2, x, sx, mu, t3, st3, x, sx, rng, stream); // t3 = Phi' * t2; x += mu * t3
    if (!rc && threshold)                                                                    // keep the K largest (2: in the reference's survivor order)
        rc = clv4_threshold_mode(x, sx, x_len, n, K, threshold == 2 ? CLV_THRESHOLD_REFERENCE : CLV_THRESHOLD_FAST, nullptr, stream);
    return rc;
}

extern "C" int clm4_iht(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n,
                        int8_t *x, float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2,
                        float *st2, int8_t *t3, float *st3, uint64_t iterations, uint64_t K, float mu, int threshold,
                        uint64_t *rng_state_dev, void *stream)
{
    CLV_REQUIRE(Phi && sPhi && PhiT && sPhiT && x && sx && y && sy && t1 && st1 && t2 && st2 && t3 && st3, "clm4_iht: null pointer");
    CLV_REQUIRE(m % 128 == 0 && n % 128 == 0 && x_len <= n, "clm4_iht: m=%llu n=%llu x_len=%llu", (unsigned long long)m,
                (unsigned long long)n, (unsigned long long)x_len);
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(k_v4_clear, dim3(64), dim3(256), 0, st, (uint32_t *)x, sx, n / 8, n / 64);   // x.clear()
    CLV_LAUNCH_CHECK();
    if (!iterations) return CLV_OK;
    // plain launches: a captured-graph replay of the five kernels was measured SLOWER on MI355X (39 vs 33 us per
    // iteration at N = 8192: the per-replay cost exceeds the five launch gaps it removes), so none is used
    for (uint64_t it = 0; it < iterations; it++) {
        int rc = iht_iteration(Phi, sPhi, PhiT, sPhiT, m, n, x, sx, x_len, y, sy, t1, st1, t2, st2, t3, st3, K, mu, threshold,
                               rng_state_dev, stream);
        if (rc) return rc;
    }
    return CLV_OK;
}

// =================================================================================================
// f4'  mixed precision CloverMatrix4::mvm(const CloverVector32 &, CloverVector32 &) (CloverMatrix4.h:1451-1547)
//      fp32 vector in, fp32 row dots out.  The reference keeps 4 accumulators x 8 AVX lanes = 32 sequential
//      fma chains per row; chain (e mod 32) takes elements e, e+32, ...  Lane = (row, a = word index mod 4) owns the 8
//      chains of accumulator a.  The four lanes of a row load 64 contiguous bytes (one dwordx4 each) and transpose the
//      4x4 dwords inside the quad, which leaves lane a with words a, 4+a, 8+a, 12+a: its next four words.  x lives in
//      LDS as fp32 (16384-column chunks = 64 KiB) together with f32(s/7) per block.  Products are
//      f32((float)q * f32(s/7)) * x with one fma, as there.
// =================================================================================================
#ifndef MVF_CHUNK
#define MVF_CHUNK 16384u
#endif
#ifndef MVF_U
#define MVF_U 4                     // matrix loads (dwordx4 per lane) requested one step ahead
#endif
#ifndef MVF_FENCE
#define MVF_FENCE 0
#endif
#ifndef MVF_WAVES
#define MVF_WAVES 1                 // minimum waves per SIMD the register allocation must leave room for (A/B builds: 3, 4)
#endif

template <bool NT>
__global__ __launch_bounds__(256, MVF_WAVES) void k_m4_mvm_f32(const u32x4 *__restrict__ A, const float *__restrict__ sA, uint64_t cols,
                                                    const float *__restrict__ x, float *__restrict__ r)
{
    extern __shared__ __attribute__((aligned(16))) float mvf_x[];        // MVF_CHUNK floats of x, then MVF_CHUNK/64 block factors
    float *s7 = mvf_x + MVF_CHUNK;
    const int tid = threadIdx.x, a = tid & 3, rho = tid >> 2;
    const uint64_t row = (uint64_t)blockIdx.x * 64 + rho;
    const u32x4 *Arow = A + row * (cols / 32);
    const float *su = sA + (uint64_t)blockIdx.x * (cols / 64);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.0f;
    for (uint64_t c0 = 0; c0 < cols; c0 += MVF_CHUNK) {
        const uint32_t cw = (uint32_t)((cols - c0) < MVF_CHUNK ? (cols - c0) : MVF_CHUNK);
        if (c0) __syncthreads();
        int fast;
        {   // stage x and s/7: all loads first (one round trip), then the LDS writes
            constexpr int NX = MVF_CHUNK / 4 / 256;                       // 16 x 16 B per thread
            f32x4 xr[NX];
            const uint32_t nx = cw / 4, nb = cw / 64;
#pragma unroll
            for (int k = 0; k < NX; k++) { const uint32_t i = tid + 256 * k; xr[k] = reinterpret_cast<const f32x4 *>(x + c0)[i < nx ? i : 0]; }
            const float sv = su[c0 / 64 + ((uint32_t)tid < nb ? tid : 0)];
#pragma unroll
            for (int k = 0; k < NX; k++) { const uint32_t i = tid + 256 * k; if (i < nx) reinterpret_cast<f32x4 *>(mvf_x)[i] = xr[k]; }
            if ((uint32_t)tid < nb) s7[tid] = div7(sv);
            fast = (uint32_t)tid >= nb || times16_is_finite(div7(sv));
        }
        // the barrier the staging needs anyway also tells whether every block factor c of the chunk survives a multiplication by 16
        // (no overflow): then the nibbles are taken as q / 16 (one v_cvt_off_f32_i4 each, common.h) and (q / 16) * (16 c) rounds like q * c
        fast = __builtin_amdgcn_readfirstlane(__syncthreads_and(fast));      // scalar: a real branch, not two predicated bodies
        const u32x4 *Ap = Arow + c0 / 32;
        const uint32_t ngroups = cw / 128;                                // 16 words = 128 columns per quad and step
        constexpr int U = MVF_U;
        // FAST: every block factor c of the chunk survives 16 c (see the barrier above)
        auto chunk = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
            auto group = [&](const u32x4 av, uint32_t g) {
                uint32_t w[4] = {av.x, av.y, av.z, av.w};
                quad_transpose4(w[0], w[1], w[2], w[3], a);                   // words a, 4+a, 8+a, 12+a of group g
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t wi = 16 * g + 4 * i + a;                   // word index inside the chunk
                    const float sc = s7[wi >> 3];
                    const f32x4 xl = reinterpret_cast<const f32x4 *>(mvf_x)[2 * wi];
                    const f32x4 xh = reinterpret_cast<const f32x4 *>(mvf_x)[2 * wi + 1];
                    const float xv[8] = {xl.x, xl.y, xl.z, xl.w, xh.x, xh.y, xh.z, xh.w};
                    if constexpr (FAST) {
                        const float sc16 = sc * 16.0f;
                        float f16[8];
                        unpack8_16th(w[i], f16);                                  // q / 16 (r5: 9 VALU per word instead of 11)
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[j] = __builtin_fmaf(xv[j], f16[j] * sc16, acc[j]);     // rounded product first
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[j] = __builtin_fmaf(xv[j], (float)unpack1(w[i], j) * sc, acc[j]);
                    }
                }
            };
            const uint32_t last = ngroups - 1;
            auto ld = [&](uint32_t gg) {                                   // clamped: a step past the end re-reads the last group
                const uint32_t gc = gg < last ? gg : last;
                return NT ? __builtin_nontemporal_load(&Ap[4 * gc + a]) : Ap[4 * gc + a];
            };
            uint32_t g = 0;
            u32x4 cur[U];
#pragma unroll
            for (int u = 0; u < U; u++) cur[u] = ld(u);
            for (; g + U <= ngroups; g += U) {
                u32x4 nxt[U];
#pragma unroll
                for (int u = 0; u < U; u++) nxt[u] = ld(g + U + u);
                asm volatile("" ::: "memory");                              // the next step's loads are issued HERE, before this step's arithmetic
#pragma unroll
                for (int u = 0; u < U; u++) {
                    group(cur[u], g + u);
#if MVF_FENCE
                    asm volatile("" ::: "memory");                          // one group's x reads at a time: keeps the live registers of a step down
#endif
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < U; u++) cur[u] = nxt[u];
            }
#pragma unroll
            for (int u = 0; u < U - 1; u++)
                if (g + u < ngroups) group(cur[u], g + u);
        };
        if (fast) chunk(std::true_type{});
        else chunk(std::false_type{});
    }
    // (acc1 + acc2) + (acc3 + acc4) per AVX lane j, then the CloverBase.h:149-157 tree over the 8 lanes
    float s3[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float s12 = acc[j] + __shfl_xor(acc[j], 1);        // lanes a=0,1 -> acc1+acc2 ; a=2,3 -> acc3+acc4
        s3[j] = s12 + __shfl_xor(s12, 2);
    }
    const float t0 = s3[4] + s3[0], t1 = s3[5] + s3[1], t2 = s3[6] + s3[2], t3 = s3[7] + s3[3];
    const float dot = (t0 + t2) + (t1 + t3);
    if (a == 0) r[row] = dot;
}

extern "C" int clm4_mvm_f32(const int8_t *A, const float *sA, uint64_t rows, uint64_t cols, const float *x, float *r, void *stream)
{
    CLV_REQUIRE(A && sA && x && r, "clm4_mvm_f32: null pointer");
    CLV_REQUIRE(rows % 64 == 0 && cols % 128 == 0 && rows / 64 <= 0x7FFFFFFFull, "clm4_mvm_f32: rows=%llu must be a multiple of 64 (a row shard) and cols=%llu of 128",
                (unsigned long long)rows, (unsigned long long)cols);
    if (!rows) return CLV_OK;
    const size_t lds = (MVF_CHUNK + MVF_CHUNK / 64) * sizeof(float);                     // 65 KiB
    const dim3 grid((unsigned)(rows / 64));
    if (rows * (cols / 2) > (256ull << 20)) {
        CLV_HIP(hipFuncSetAttribute((const void *)k_m4_mvm_f32<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_m4_mvm_f32<true>, grid, dim3(256), lds, as_stream(stream), (const u32x4 *)A, sA, cols, x, r);
    } else {
        CLV_HIP(hipFuncSetAttribute((const void *)k_m4_mvm_f32<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_m4_mvm_f32<false>, grid, dim3(256), lds, as_stream(stream), (const u32x4 *)A, sA, cols, x, r);
    }
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
