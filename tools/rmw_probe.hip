// rmw_probe.hip -- the ceiling of an in-place pass on MI355X: read 128 MiB, change it, write it back (the traffic of the threshold's apply
// pass, k_th4_apply3, without its arithmetic), for the lane mappings an apply kernel could take.
//   hipcc --offload-arch=gfx950 -O3 -o tools/rmw_probe tools/rmw_probe.hip && tools/rmw_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: lane = 32 contiguous bytes (two 16-byte accesses at stride 32 over the wave), one per thread
// MODE 1: lane = 16 bytes, twice, each access contiguous over the wave (1 KiB apart)
// MODE 2: as 0, two such pairs per thread (a second workgroup-width away)
// MODE 3: as 1, four accesses per thread
template <int MODE, int NT>
__global__ __launch_bounds__(256) void k_rmw(u32x4 *__restrict__ q, const float *__restrict__ s)
{
    constexpr int NV = MODE >= 2 ? 4 : 2;
    const uint64_t base = (uint64_t)blockIdx.x * 256 * NV;
    uint64_t idx[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
        if (MODE == 0) idx[i] = base + 2 * threadIdx.x + i;
        else if (MODE == 1 || MODE == 3) idx[i] = base + (threadIdx.x >> 6) * 64 * NV + 64 * i + (threadIdx.x & 63);
        else idx[i] = base + 512 * (i >> 1) + 2 * threadIdx.x + (i & 1);
    }
    u32x4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = (NT & 1) ? __builtin_nontemporal_load(&q[idx[i]]) : q[idx[i]];
    const uint32_t m = __float_as_uint(s[(base >> 1) + threadIdx.x]) | 0x0F0F0F0Fu;     // one scale per 32 bytes, as the apply pass reads
#pragma unroll
    for (int i = 0; i < NV; i++) {
        v[i] &= m;
        if (NT & 2) __builtin_nontemporal_store(v[i], &q[idx[i]]); else q[idx[i]] = v[i];
    }
}

template <int MODE, int NT>
static void run(u32x4 *q, float *s, uint64_t bytes)
{
    constexpr int NV = MODE >= 2 ? 4 : 2;
    const unsigned grid = (unsigned)(bytes / 16 / 256 / NV);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((k_rmw<MODE, NT>), dim3(grid), dim3(256), 0, 0, q, s);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    printf("mode %d nt %d: %7.2f us  %.2f TB/s (read + write + scales)\n", MODE, NT, best * 1e3, (2.0 * bytes + bytes / 8) / (best * 1e-3) / 1e12);
}

int main()
{
    const uint64_t bytes = 128ull << 20;
    u32x4 *q;
    float *s;
    if (hipMalloc(&q, bytes) != hipSuccess || hipMalloc(&s, bytes / 8) != hipSuccess) return 1;
    (void)hipMemset(q, 0x5A, bytes);
    (void)hipMemset(s, 0x3F, bytes / 8);
    run<0, 0>(q, s, bytes); run<0, 3>(q, s, bytes);
    run<1, 0>(q, s, bytes); run<1, 3>(q, s, bytes);
    run<2, 0>(q, s, bytes); run<2, 3>(q, s, bytes);
    run<3, 0>(q, s, bytes); run<3, 3>(q, s, bytes);
    return 0;
}
