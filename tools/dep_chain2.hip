// dep_chain2.hip -- what costs time in the exact-order dot chain: fma latency, LDS reads, permlane hops?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int batches)
{
    __shared__ f32x4 lds[2304];
    for (int i = threadIdx.x; i < 2304; i += 64) lds[i] = f32x4{1e-9f * i, 1.0f, 0.5f, 0.25f};
    __syncthreads();
    float acc = threadIdx.x;
    const int j = threadIdx.x & 15, row = threadIdx.x >> 4;
    f32x4 f = lds[j], c = lds[2048 + (j >> 3)];
    for (int b = 0; b < batches; b++) {
        f32x4 fn = f, cn = c;
        if (MODE >= 1) { const int g = (b & 127); fn = lds[(4 * (g & 31) + row) * 16 + j]; cn = lds[2048 + (4 * (g & 31) + row) * 2 + (j >> 3)]; }
#pragma unroll
        for (int h = 0; h < 4; h++) {
            acc = __builtin_fmaf(c.x, f.x, acc);
            acc = __builtin_fmaf(c.y, f.y, acc);
            acc = __builtin_fmaf(c.z, f.z, acc);
            acc = __builtin_fmaf(c.w, f.w, acc);
            if (MODE == 2) {
                auto r = (h & 1) ? __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false)
                                 : __builtin_amdgcn_permlane16_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
                acc = __uint_as_float(h < 2 ? r[0] : r[1]);
            }
        }
        f = fn; c = cn;
    }
    if (acc == 123.f) out[0] = acc;
}
template <int MODE> void run(const char *name)
{
    float *out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int batches = 8192;
    k<MODE><<<1, 64>>>(out, 16);
    hipEventRecord(a); k<MODE><<<1, 64>>>(out, batches); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s %.3f ms -> %.2f ns per chain step\n", name, ms, ms * 1e6 / (batches * 16.0));
}
int main() { run<0>("16 dependent fma per batch, registers"); run<1>("+ 2 ds_read_b128 per batch (prefetched)"); run<2>("+ 4 permlane hops per batch"); return 0; }
