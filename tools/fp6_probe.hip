// fp6_probe.hip -- what v_mfma_scale_f32_16x16x128_f8f6f4 does with FP6 (E2M3) operands on gfx950 (experiment, not product).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/fp6_probe tools/fp6_probe.hip
// Part 1 (semantics): A[16][128], B[16][128] of random integers in [-7,7] encoded as sign | 00 | magnitude (subnormal and
// first-binade E2M3 codes: value = magnitude / 8), packed 32 x 6 bits per lane (lane = row + 16 g holds k = 32 g .. 32 g + 31,
// element p in bits [6p, 6p+6)); D compared with the integer sums / 64.  Also the zero-half trick (two K-blocks per instruction).
// Part 2 (rate): MFMA + fold loops, cycles per 16x16 tile and K-block, against the int8 path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ONE_SCALE 0x7F7F7F7F

__global__ void k_one(const uint32_t *A, const uint32_t *B, float *D, int fmt)
{
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (int)A[lane * 8 + i]; b[i] = (int)B[lane * 8 + i]; }
    f32x4 c = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 d;
    if (fmt == 2) d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, ONE_SCALE, 0, ONE_SCALE);
    else d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, ONE_SCALE, 0, ONE_SCALE);
    for (int t = 0; t < 4; t++) D[lane * 4 + t] = d[t];
}

// rate loops: NT tiles per iteration, FOLD = fold instructions per element (0, 1 = fma, 2 = sub + fma)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters, float c)
{
    const int lane = threadIdx.x & 63;
    i32x8 a[4], b[2];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) a[i][j] = (j < 6) ? (lane * 7 + i + j) & 0x07070707 : 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) b[i][j] = (j < 6) ? (lane * 5 + i + j) & 0x07070707 : 0;
    float acc[4][2][4] = {};
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 1) {          // fp6 MX, fold = 1 fma per element (MODE 1) or none (MODE 0)
            f32x4 S[4][2];
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 2; y++)
                    S[x][y] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[x], b[y], z, 2, 2, 0, ONE_SCALE, 0, ONE_SCALE);
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 2; y++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (MODE == 1) acc[x][y][t] = __builtin_fmaf(c, S[x][y][t], acc[x][y][t]);
                        else acc[x][y][t] += S[x][y][t] * 0.0f + (it == -1 ? 1.0f : 0.0f);
                    }
        } else {                               // int8 16x16x64, fold = sub + fma (MODE 3) or none (MODE 2)
            i32x4 S[4][2];
            const i32x4 bias = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 2; y++) {
                    const i32x4 fa = {a[x][0], a[x][1], a[x][2], a[x][3]}, fb = {b[y][0], b[y][1], b[y][2], b[y][3]};
                    S[x][y] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, fb, bias, 0, 0, 0);
                }
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = 0; y < 2; y++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (MODE == 3) acc[x][y][t] = __builtin_fmaf(c, __int_as_float(S[x][y][t]) - 12582912.0f, acc[x][y][t]);
                        else acc[x][y][t] += (float)(S[x][y][t] & (it == -1 ? 1 : 0));
                    }
        }
        a[0][0] ^= it;       // keep the loop body from being hoisted
    }
    float s = 0.0f;
    for (int x = 0; x < 4; x++) for (int y = 0; y < 2; y++) for (int t = 0; t < 4; t++) s += acc[x][y][t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static uint32_t code_of(int q) { return q < 0 ? (0x20u | (uint32_t)(-q)) : (uint32_t)q; }

static void pack_fp6(const int *M, uint32_t *regs)      // M[16][128] -> regs[64][8]
{
    for (int lane = 0; lane < 64; lane++) {
        const int row = lane & 15, g = lane >> 4;
        uint64_t w[4] = {0, 0, 0, 0};
        for (int p = 0; p < 32; p++) {
            const uint64_t c = code_of(M[row * 128 + 32 * g + p]);
            const int bit = 6 * p;
            w[bit / 64] |= c << (bit % 64);
            if (bit % 64 > 58) w[bit / 64 + 1] |= c >> (64 - bit % 64);
        }
        for (int i = 0; i < 8; i++) regs[lane * 8 + i] = (uint32_t)(w[i / 2] >> (32 * (i & 1)));
    }
}

template <int MODE>
static void rate(const char *name, float *dout, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, dout, 100, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks*4 waves over 1024 SIMDs
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    const double tiles = (double)iters * 8 * waves_per_simd;
    printf("%-28s blocks=%d  %.3f ms  %.1f ns per tile-Kstep per SIMD (x2.4 = cycles: %.1f)\n", name, blocks, ms, ms * 1e6 / tiles, ms * 1e6 / tiles * 2.4);
}


typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_rate32(float *out, int iters, float c)
{
    const int lane = threadIdx.x & 63;
    i32x8 a[2], b[2];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 8; j++) { a[i][j] = (j < 6) ? (lane * 7 + i + j) & 0x07070707 : 0; b[i][j] = (j < 6) ? (lane * 5 + i + j) & 0x07070707 : 0; }
    f32x16 acc[2];
    for (int x = 0; x < 2; x++) for (int t = 0; t < 16; t++) acc[x][t] = 0.0f;
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 sp = z;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int x = 0; x < 2; x++) {
                if (MODE == 4) acc[x] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[x], b[j], acc[x], 2, 2, 0, 0x82828282, 0, 0x82828282);
                else if (MODE == 6) {
                    // software-pipelined: MFMA of this tile first, then the fold of the previous tile's result beside it
                    const f32x16 s = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[x], b[j], z, 2, 2, 0, 0x82828282, 0, 0x82828282);
                    const int px = x ^ 1;
#pragma unroll
                    for (int t = 0; t < 16; t++) acc[px][t] = __builtin_fmaf(c, sp[t], acc[px][t]);
                    sp = s;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                } else {
                    const f32x16 s = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[x], b[j], z, 2, 2, 0, 0x82828282, 0, 0x82828282);
#pragma unroll
                    for (int t = 0; t < 16; t++) acc[x][t] = __builtin_fmaf(c, s[t], acc[x][t]);
                }
            }
        a[0][0] ^= it;
    }
    float s = 0.0f;
    for (int x = 0; x < 2; x++) for (int t = 0; t < 16; t++) s += acc[x][t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void rate32(const char *name, float *dout, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL(k_rate32<MODE>, dim3(blocks), dim3(512), 0, 0, dout, 100, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate32<MODE>, dim3(blocks), dim3(512), 0, 0, dout, iters, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks * 8.0 / 1024.0;
    const double mfmas = (double)iters * 4 * waves_per_simd;
    printf("%-28s blocks=%d  %.3f ms  %.1f ns per 32x32x64 MFMA per SIMD (x2.4 = cycles: %.1f); 8192^3 at this rate: %.3f ms\n", name, blocks, ms,
           ms * 1e6 / mfmas, ms * 1e6 / mfmas * 2.4, ms * 1e6 / mfmas * 8192.0 * 1e-6);
}

int main()
{
    std::vector<int> A(16 * 128), B(16 * 128);
    srand(7);
    for (auto &v : A) v = rand() % 15 - 7;
    for (auto &v : B) v = rand() % 15 - 7;
    std::vector<uint32_t> ra(64 * 8), rb(64 * 8);
    pack_fp6(A.data(), ra.data());
    pack_fp6(B.data(), rb.data());
    uint32_t *dA, *dB;
    float *dD;
    hipMalloc(&dA, ra.size() * 4); hipMalloc(&dB, rb.size() * 4); hipMalloc(&dD, 64 * 4 * 4);
    hipMemcpy(dA, ra.data(), ra.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, rb.data(), rb.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dD, 2);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int t = 0; t < 4; t++) {
            const int col = lane & 15, row = 4 * (lane >> 4) + t;
            long s = 0;
            for (int k = 0; k < 128; k++) s += (long)A[row * 128 + k] * B[col * 128 + k];
            const float want = (float)s / 64.0f;
            if (D[lane * 4 + t] != want) { if (bad < 8) printf("mismatch row %d col %d: got %g want %g\n", row, col, D[lane * 4 + t], want); bad++; }
        }
    printf("fp6 full K=128: %s (%d mismatches)\n", bad ? "FAIL" : "exact", bad);

    // zero-half trick: B keeps only lane groups 0,1 -> sum over k < 64
    std::vector<uint32_t> rb0 = rb;
    for (int lane = 32; lane < 64; lane++) for (int i = 0; i < 8; i++) rb0[lane * 8 + i] = 0;
    hipMemcpy(dB, rb0.data(), rb0.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_one, dim3(1), dim3(64), 0, 0, dA, dB, dD, 2);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int t = 0; t < 4; t++) {
            const int col = lane & 15, row = 4 * (lane >> 4) + t;
            long s = 0;
            for (int k = 0; k < 64; k++) s += (long)A[row * 128 + k] * B[col * 128 + k];
            if (D[lane * 4 + t] != (float)s / 64.0f) bad++;
        }
    printf("fp6 first half only: %s (%d mismatches)\n", bad ? "FAIL" : "exact", bad);

    float *dout;
    hipMalloc(&dout, 4096 * 256 * 4);
    for (int blocks : {256, 512, 1024}) {
        rate<0>("fp6 MX, no fold", dout, blocks);
        rate<1>("fp6 MX, fma fold", dout, blocks);
        rate<2>("int8, no fold", dout, blocks);
        rate<3>("int8, sub+fma fold", dout, blocks);
    }
    for (int blocks : {256, 512}) {
        rate32<4>("fp6 32x32x64, no fold", dout, blocks);
        rate32<5>("fp6 32x32x64, 16 fma fold", dout, blocks);
        rate32<6>("fp6 32x32x64, pipelined fold", dout, blocks);
    }
    return 0;
}
