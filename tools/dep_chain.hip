// dep_chain.hip -- latency of a dependent v_fmac_f32 chain (what bounds the exact-order dot), with 1 wave and with a busy chip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void k_chain(float *out, int iters, int lanes)
{
    float a = threadIdx.x * 1e-9f, b = 1.0000001f, c = 1e-9f;
    if ((int)threadIdx.x < lanes) {
        for (int it = 0; it < iters; it++) {
            REP16(asm volatile("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %0, %1, %2\n" : "+v"(a) : "v"(b), "v"(c));)
        }
    }
    if (a == 123.0f) out[0] = a;
}
int main()
{
    float *out; hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4096;
    struct { int blocks, threads, lanes; const char *name; } cfg[] = {
        {1, 64, 64, "1 wave, 64 lanes"}, {1, 64, 16, "1 wave, 16 lanes"}, {1, 256, 256, "1 WG of 4 waves"},
        {1024, 256, 256, "1024 WGs (busy chip)"}, {256, 64, 16, "256 WGs x 1 wave, 16 lanes"}};
    for (auto &c : cfg) {
        k_chain<<<c.blocks, c.threads>>>(out, 16, c.lanes);
        hipEventRecord(a);
        k_chain<<<c.blocks, c.threads>>>(out, iters, c.lanes);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %.3f ms -> %.2f ns per dependent fmac\n", c.name, ms, ms * 1e6 / (iters * 64.0));
    }
    return 0;
}
