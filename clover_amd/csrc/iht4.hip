// iht4.hip -- the quantized IHT / GD loop on the device: the five steps of an iteration, nothing copied back between them.
// One of the callers either side of the hot path.  With mvm these are the five steps of the reference's quantized IHT / GD
// iterations (test/performance/01_measure.h:923-946, 999-1021), so x, t1..t3 can stay in HBM across iterations.
#include "common.h"

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

// iht_persist.hip: the whole loop as one persistent launch with Phi and PhiT in LDS (N <= 8192, rounding disabled, threshold FAST or none);
// 1 = launched, 0 = does not qualify, < 0 = error
int clm4_iht_persistent(const int8_t *Phi, const float *sPhi, const int8_t *PhiT, const float *sPhiT, uint64_t m, uint64_t n, int8_t *x,
                        float *sx, uint64_t x_len, const int8_t *y, const float *sy, int8_t *t1, float *st1, int8_t *t2, float *st2, int8_t *t3,
                        float *st3, uint64_t iterations, uint64_t K, float mu, int threshold, uint64_t *rng, hipStream_t st);

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
    {
        const int p = clm4_iht_persistent(Phi, sPhi, PhiT, sPhiT, m, n, x, sx, x_len, y, sy, t1, st1, t2, st2, t3, st3, iterations, K, mu, threshold,
                                          rng_state_dev, st);
        if (p < 0) return CLV_ERR_HIP;
        if (p > 0) return CLV_OK;
    }
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
