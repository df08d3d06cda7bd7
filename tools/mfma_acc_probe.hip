// mfma_acc_probe.hip -- does v_mfma_scale_f32_32x32x64_f8f6f4 (FP6 operands) issue slower when it accumulates in place (srcC = vDst,
// 16 more registers read per instruction) than with srcC = 0?  Which register file for the accumulators?  (experiment, not product)
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_acc_probe tools/mfma_acc_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define M_(d, a, b, c) "v_mfma_scale_f32_32x32x64_f8f6f4 " d ", " a ", " b ", " c ", v124, v124 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"
#define FA0 "v[100:105]"
#define FA1 "v[106:111]"
#define FB0 "v[112:117]"
#define FB1 "v[118:123]"

#define LOOP(BODY)                                                                                                                   \
    "s_mov_b32 s42, %[iters]\n v_mov_b32 v124, 0x82828282\n"                                                                         \
    "v_mov_b32 v100, %[f]\n v_mul_lo_u32 v101, v100, %[g]\n v_mul_lo_u32 v102, v101, %[g]\n v_mul_lo_u32 v103, v102, %[g]\n v_mul_lo_u32 v104, v103, %[g]\n v_mul_lo_u32 v105, v104, %[g]\n" \
    "v_mul_lo_u32 v106, v105, %[g]\n v_mul_lo_u32 v107, v106, %[g]\n v_mul_lo_u32 v108, v107, %[g]\n v_mul_lo_u32 v109, v108, %[g]\n v_mul_lo_u32 v110, v109, %[g]\n v_mul_lo_u32 v111, v110, %[g]\n" \
    "v_mul_lo_u32 v112, v111, %[g]\n v_mul_lo_u32 v113, v112, %[g]\n v_mul_lo_u32 v114, v113, %[g]\n v_mul_lo_u32 v115, v114, %[g]\n v_mul_lo_u32 v116, v115, %[g]\n v_mul_lo_u32 v117, v116, %[g]\n" \
    "v_mul_lo_u32 v118, v117, %[g]\n v_mul_lo_u32 v119, v118, %[g]\n v_mul_lo_u32 v120, v119, %[g]\n v_mul_lo_u32 v121, v120, %[g]\n v_mul_lo_u32 v122, v121, %[g]\n v_mul_lo_u32 v123, v122, %[g]\n" \
    "s_nop 4\n"                                                                                                                      \
    "1:\n" BODY "s_sub_u32 s42, s42, 1\n s_cmp_lg_u32 s42, 0\n s_cbranch_scc1 1b\n s_nop 15\n v_mov_b32 %[o], v0\n"

#define RUNASM(BODY) asm volatile(LOOP(BODY) : [o] "=v"(o) : [iters] "s"(iters), [f] "v"(f), [g] "v"(g) : "memory", "scc", "s42", CL)

#define CL                                                                                                                            \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21",   \
        "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41",    \
        "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61",    \
        "v62", "v63", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",  \
        "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10",    \
        "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30",    \
        "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50",    \
        "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"

template <int V>
__global__ __launch_bounds__(256) void k_probe(float *out, int iters, int f, int g, int rnd)
{
    float o;
    if (rnd) {      // different bits in every lane (and, through the v_mov chain below, the same per register): toggling operands
        uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        f = (int)(h & 0x3F3F3F3Fu ^ (rnd == 2 ? 0u : 0u));
        g = (int)((h * 3266489917u) >> 1);
    }
    if (V == 0)            // srcC = 0, four destinations
        RUNASM(M_("v[0:15]", FA0, FB0, "0") M_("v[16:31]", FA0, FB1, "0") M_("v[32:47]", FA1, FB1, "0") M_("v[48:63]", FA1, FB0, "0")
               M_("v[0:15]", FA0, FB0, "0") M_("v[16:31]", FA0, FB1, "0") M_("v[32:47]", FA1, FB1, "0") M_("v[48:63]", FA1, FB0, "0"));
    else if (V == 1)       // accumulate in place, VGPRs
        RUNASM(M_("v[0:15]", FA0, FB0, "v[0:15]") M_("v[16:31]", FA0, FB1, "v[16:31]") M_("v[32:47]", FA1, FB1, "v[32:47]") M_("v[48:63]", FA1, FB0, "v[48:63]")
               M_("v[0:15]", FA0, FB0, "v[0:15]") M_("v[16:31]", FA0, FB1, "v[16:31]") M_("v[32:47]", FA1, FB1, "v[32:47]") M_("v[48:63]", FA1, FB0, "v[48:63]"));
    else if (V == 2)       // accumulate in place, AGPRs
        RUNASM(M_("a[0:15]", FA0, FB0, "a[0:15]") M_("a[16:31]", FA0, FB1, "a[16:31]") M_("a[32:47]", FA1, FB1, "a[32:47]") M_("a[48:63]", FA1, FB0, "a[48:63]")
               M_("a[0:15]", FA0, FB0, "a[0:15]") M_("a[16:31]", FA0, FB1, "a[16:31]") M_("a[32:47]", FA1, FB1, "a[32:47]") M_("a[48:63]", FA1, FB0, "a[48:63]"));
    else if (V == 3)       // srcC = 0, AGPR destinations
        RUNASM(M_("a[0:15]", FA0, FB0, "0") M_("a[16:31]", FA0, FB1, "0") M_("a[32:47]", FA1, FB1, "0") M_("a[48:63]", FA1, FB0, "0")
               M_("a[0:15]", FA0, FB0, "0") M_("a[16:31]", FA0, FB1, "0") M_("a[32:47]", FA1, FB1, "0") M_("a[48:63]", FA1, FB0, "0"));
    else if (V == 4)       // srcC = another register set (read 16, write a different 16)
        RUNASM(M_("v[0:15]", FA0, FB0, "v[32:47]") M_("v[16:31]", FA0, FB1, "v[48:63]") M_("v[32:47]", FA1, FB1, "v[0:15]") M_("v[48:63]", FA1, FB0, "v[16:31]")
               M_("v[0:15]", FA0, FB0, "v[32:47]") M_("v[16:31]", FA0, FB1, "v[48:63]") M_("v[32:47]", FA1, FB1, "v[0:15]") M_("v[48:63]", FA1, FB0, "v[16:31]"));
    else if (V == 5)       // accumulate in place, operands from AGPRs? no: same fragments, one accumulator only (dependent chain)
        RUNASM(M_("v[0:15]", FA0, FB0, "v[0:15]") M_("v[0:15]", FA0, FB1, "v[0:15]") M_("v[0:15]", FA1, FB1, "v[0:15]") M_("v[0:15]", FA1, FB0, "v[0:15]")
               M_("v[0:15]", FA0, FB0, "v[0:15]") M_("v[0:15]", FA0, FB1, "v[0:15]") M_("v[0:15]", FA1, FB1, "v[0:15]") M_("v[0:15]", FA1, FB0, "v[0:15]"));
    out[blockIdx.x * blockDim.x + threadIdx.x] = o;
}

template <int V>
static void run(const char *name, float *dout, int wg_per_cu, int rnd)
{
    const int iters = 2000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<V>, dim3(blocks), dim3(256), 0, 0, dout, 50, 0x05030107, 0x02060401, rnd);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<V>, dim3(blocks), dim3(256), 0, 0, dout, iters, 0x05030107, 0x02060401, rnd);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (iters * 8.0 * wg_per_cu);
    printf("%-44s rnd=%d waves/SIMD=%d  %.3f ms  %.2f ns per MFMA per SIMD  -> 8192^3 MFMAs: %.3f ms\n", name, rnd, wg_per_cu, ms, ns, ns * 8192.0 * 1e-6);
}

int main()
{
    float *dout;
    hipMalloc(&dout, 256 * 8 * 256 * 4);
    for (int rep = 0; rep < 2; rep++)
        for (int rnd = 0; rnd <= 1; rnd++)
            for (int w = 1; w <= 4; w += 3) {
                run<0>("srcC = 0, VGPR destinations", dout, w, rnd);
                run<1>("accumulate in place, VGPR", dout, w, rnd);
                run<2>("accumulate in place, AGPR", dout, w, rnd);
                printf("\n");
            }
    return 0;
}
