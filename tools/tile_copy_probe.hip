// tile_copy_probe.hip -- what a tile-shaped access pattern costs on MI355X, without any transposition: every workgroup reads a tile of
// ROWS rows x RUN bytes out of a row-major 32768^2-nibble matrix (row pitch 16 KiB) and writes a tile of the same shape at the mirrored
// tile position of a second matrix -- the traffic of CloverMatrix4::transpose with RUN = 128 (k_m4_transpose: 256 x 256 elements).
//   hipcc --offload-arch=gfx950 -O3 -o tools/tile_copy_probe tools/tile_copy_probe.hip && tools/tile_copy_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int RUN, int ROWS, int NT>
__global__ __launch_bounds__(256) void k_tile_copy(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, uint64_t pitch, uint32_t tiles_x, int xcd_blocks)
{
    constexpr int LPR = RUN / 16;                 // lanes per row
    constexpr int RPI = 256 / LPR;                // rows per load instruction of the workgroup
    constexpr int NI = ROWS / RPI;                // 16-byte loads per lane
    uint32_t bj = blockIdx.x % tiles_x, bi = blockIdx.x / tiles_x;
    const uint32_t ntiles = gridDim.x, tiles_y = ntiles / tiles_x;
    if (xcd_blocks && ntiles % 8 == 0 && tiles_x % 8 == 0 && tiles_y % 8 == 0) {
        const uint32_t t = (blockIdx.x & 7) * (ntiles / 8) + (blockIdx.x >> 3);
        const uint32_t blk = t / 64, inb = t % 64, bx = tiles_x / 8;
        bi = (blk / bx) * 8 + inb / 8;
        bj = (blk % bx) * 8 + inb % 8;
    }
    const int c = threadIdx.x % LPR, r = threadIdx.x / LPR;
    u32x4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(in + ((uint64_t)bi * ROWS + r + RPI * i) * pitch + (uint64_t)bj * RUN + 16 * c);
        v[i] = (NT & 1) ? __builtin_nontemporal_load(p) : *p;
    }
    // the mirrored tile: tile (bj, bi) of a matrix whose tiles are ROWS x RUN as well (tiles_y tiles per row there)
    const uint64_t opitch = (uint64_t)tiles_y * RUN;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        u32x4 *p = reinterpret_cast<u32x4 *>(out + ((uint64_t)bj * ROWS + r + RPI * i) * opitch + (uint64_t)bi * RUN + 16 * c);
        if (NT & 2) __builtin_nontemporal_store(v[i], p); else *p = v[i];
    }
}

template <int RUN, int ROWS, int NT>
static void run(const uint8_t *in, uint8_t *out, uint64_t n, int xcd)
{
    const uint64_t pitch = n / 2;
    const uint32_t tiles_x = (uint32_t)(pitch / RUN), tiles_y = (uint32_t)(n / ROWS);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_tile_copy<RUN, ROWS, NT>), dim3(tiles_x * tiles_y), dim3(256), 0, 0, in, out, pitch, tiles_x, xcd);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_tile_copy<RUN, ROWS, NT>), dim3(tiles_x * tiles_y), dim3(256), 0, 0, in, out, pitch, tiles_x, xcd);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("nt %d run %4d B x %4d rows, xcd blocks %d: %.4f ms  %.0f GB/s  %.3f of 8 TB/s\n", NT, RUN, ROWS, xcd, ms, 2.0 * n * n / 2 / ms / 1e6, 2.0 * n * n / 2 / ms / 1e6 / 8000);
}

int main()
{
    const uint64_t n = 32768;
    uint8_t *in, *out;
    if (hipMalloc(&in, n * n / 2) != hipSuccess || hipMalloc(&out, n * n / 2) != hipSuccess) return 1;
    (void)hipMemset(in, 1, n * n / 2);
    (void)hipMemset(out, 2, n * n / 2);
    run<128, 256, 0>(in, out, n, 1);
    run<128, 256, 1>(in, out, n, 1);
    run<128, 256, 2>(in, out, n, 1);
    run<128, 256, 3>(in, out, n, 1);
    run<512, 64, 0>(in, out, n, 1);
    run<512, 64, 3>(in, out, n, 1);
    run<4096, 16, 0>(in, out, n, 0);
    run<4096, 16, 1>(in, out, n, 0);
    run<4096, 16, 2>(in, out, n, 0);
    run<4096, 16, 3>(in, out, n, 0);
    return 0;
}
