// mvm_f32.hip -- the mixed-precision CloverMatrix4::mvm(const CloverVector32 &, CloverVector32 &).
// One of the callers either side of the hot path.  With mvm these are the five steps of the reference's quantized IHT / GD
// iterations (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "common.h"

#include <type_traits>

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
