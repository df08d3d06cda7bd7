// scale_add4.hip -- CloverVector4::scaleAndAdd, deterministic and stochastic.
// One of the callers either side of the hot path.  With mvm these are the five steps of the reference's quantized IHT / GD
// iterations (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "rng_device.h"

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
__global__ __launch_bounds__(256) void k_v4_scale_and_add(const u32x4 *qu, const float *su, const u32x4 *qv,
                                                          const float *sv, float a, u32x4 *r, float *sr,
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
__global__ __launch_bounds__(256) void k_v4_scale_and_add_blk(const u32x4 *qu, const float *su, const u32x4 *qv,
                                                              const float *sv, float a, u32x4 *r, float *sr, uint64_t nblocks)
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
__global__ __launch_bounds__(256) void k_v4_scale_and_add_st(const uint32_t *qu, const float *su, const uint32_t *qv,
                                                             const float *sv, float a, uint32_t *r, float *sr,
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
