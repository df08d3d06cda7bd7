// iht_dropin.cpp -- what one quantized IHT iteration costs THROUGH THE DROP-IN SURFACE: Q_IHT of include/CloverIHT.h (the reference's caller,
// test/performance/01_measure.h:923-946) on CloverMatrix4 / CloverVector4 (and CloverVector8) objects, N = 8192 (Phi 4096 x 8192, K = 1024) --
// every step a method call on containers whose host blocks are mirrored in HBM (clover_device.h: a lock + state check per object per call, in
// the page-tracked build an mprotect when an object changes sides).  bench.py's `extras` quotes the C ABI's own loop (clm4_iht) beside it.
// Both settings of the headers' exactness switch are timed in one process (clover_hip::set_exactness); the container build is chosen at
// compile time:
//   g++ -std=c++11 -O2 -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1 [-DCLOVER_HIP_EXPLICIT_SYNC] -Iinclude tools/iht_dropin.cpp -o /tmp/iht_dropin
//       -Lclover_amd/lib -lclover_hip -Wl,-rpath,$PWD/clover_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib && /tmp/iht_dropin
// Prints one JSON object.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "CloverIHT.h"
#include "CloverMatrix32.h"
#include "CloverVector32.h"
#include "CloverVector8.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// microseconds per iteration: (time of `hi` iterations - time of `lo` iterations) / (hi - lo), median of `reps`; the read of one
// element at the end waits for the device and brings x back, as a caller looking at the result would
template <class V>
static double per_iteration_us(CloverMatrix4 &Phi, CloverMatrix4 &PhiT, V &x, V &y, V &t1, V &t2, V &t3, uint64_t K, int lo, int hi, int reps)
{
    std::vector<double> us;
    for (int r = 0; r < reps; r++) {
        double t[2];
        const int it[2] = {lo, hi};
        for (int w = 0; w < 2; w++) {
            const double t0 = now();
            Q_IHT(Phi, PhiT, x, y, t1, t2, t3, (uint64_t)it[w], K, 1e-3f);
            (void)x.getBits(0);
            t[w] = now() - t0;
        }
        us.push_back((t[1] - t[0]) / (hi - lo) * 1e6);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

// the reference's threshold (CloverVector4.h:1929-1970) on ONE host core: std::make_heap over the first k, every later element against the
// root, min_heapify (CloverBase.h:226-249) -- magnitudes decoded per element as getAbs does.  What the device's REFERENCE mode is up against.
struct HostHeapItem { float value; uint64_t idx; };
static bool host_gt(const HostHeapItem &a, const HostHeapItem &b) { return (a.value > b.value) || a.value != a.value; }
static double host_heap_walk_us(const CloverVector4 &v, uint64_t n, uint64_t k, int reps)
{
    std::vector<double> us;
    std::vector<HostHeapItem> h(k);
    uint64_t sink = 0;
    for (int r = 0; r < reps; r++) {
        const double t0 = now();
        for (uint64_t i = 0; i < k; i++) { h[i].value = v.getAbs(i); h[i].idx = i; }
        std::make_heap(h.begin(), h.end(), host_gt);
        for (uint64_t i = k; i < n; i++) {
            const float value = v.getAbs(i);
            if (value > h[0].value) {
                h[0].value = value;
                h[0].idx = i;
                uint32_t pos = 0, smallest = 0;
                for (;;) {
                    const uint32_t l = pos * 2 + 1, rr = pos * 2 + 2;
                    if (l < k && h[l].value < h[smallest].value) smallest = l;
                    if (rr < k && h[rr].value < h[smallest].value) smallest = rr;
                    if (smallest == pos) break;
                    std::swap(h[pos], h[smallest]);
                    pos = smallest;
                }
            }
        }
        for (uint64_t i = 0; i < k; i++) sink += h[i].idx;
        us.push_back((now() - t0) * 1e6);
    }
    if (sink == 1) std::printf(" ");
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

// one x.threshold(K) through the headers on a fresh copy of `src`, microseconds (median), incl. waiting for the device
static double device_threshold_us(const CloverVector4 &src, uint64_t k, int reps)
{
    std::vector<double> us;
    for (int r = 0; r < reps; r++) {
        CloverVector4 w(src);
        w.toDevice();
        const double t0 = now();
        w.threshold(k);
        clv_device_sync();
        us.push_back((now() - t0) * 1e6);
    }
    std::sort(us.begin(), us.end());
    return us[us.size() / 2];
}

int main()
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("{\"no_device\": true}\n"); return 0; }
    const uint64_t M = 4096, N = 8192, K = M / 4;
    CloverMatrix4 Phi(M, N), PhiT(N, M);
    {
        CloverMatrix32 Phi32(M, N);
        Phi32.setRandomInteger(10, 7);
        Phi.quantize(Phi32);
        Phi.transpose(PhiT);
    }
    CloverVector32 y32(M);
    y32.setRandomInteger(10, 9);
    CloverVector4 x(N), y(M), t1(M), t2(M), t3(N);
    y.quantize(y32);
    CloverVector8 x8(N), y8(y32), u1(M), u2(M), u3(N);
    Q_IHT(Phi, PhiT, x, y, t1, t2, t3, 20, K, 1e-3f);          // warm-up: mirrors created, operands uploaded, clocks up
    Q_IHT(Phi, PhiT, x8, y8, u1, u2, u3, 20, K, 1e-3f);
    (void)x.getBits(0);
    (void)x8.getBits(0);

    clover_hip::set_exactness(clover_hip::FAST);
    const double fast4 = per_iteration_us(Phi, PhiT, x, y, t1, t2, t3, K, 100, 300, 5);
    const double fast8 = per_iteration_us(Phi, PhiT, x8, y8, u1, u2, u3, K, 100, 300, 5);
    clover_hip::set_exactness(clover_hip::REFERENCE_BITS);
    const double ref4 = per_iteration_us(Phi, PhiT, x, y, t1, t2, t3, K, 4, 12, 3);
    const double ref8 = per_iteration_us(Phi, PhiT, x8, y8, u1, u2, u3, K, 4, 12, 3);
    // the generic template (five method calls per iteration, no mvm + scaleAndAdd pairing): what unchanged reference-style code costs
    clover_hip::set_exactness(clover_hip::FAST);
    std::vector<double> gen;
    for (int r = 0; r < 5; r++) {
        double t[2];
        const int it[2] = {100, 300};
        for (int w = 0; w < 2; w++) {
            const double t0 = now();
            Q_IHT<CloverMatrix4, CloverVector4>(Phi, PhiT, x, y, t1, t2, t3, (uint64_t)it[w], K, 1e-3f);
            (void)x.getBits(0);
            t[w] = now() - t0;
        }
        gen.push_back((t[1] - t[0]) / 200 * 1e6);
    }
    std::sort(gen.begin(), gen.end());
    // the same at the reference's own K = 25 % of N (00_test.cpp:702, 750; performance.txt:566)
    const uint64_t K25 = N / 4;
    const double fast4_k25 = per_iteration_us(Phi, PhiT, x, y, t1, t2, t3, K25, 100, 300, 5);
    const double fast8_k25 = per_iteration_us(Phi, PhiT, x8, y8, u1, u2, u3, K25, 100, 300, 5);
    clover_hip::set_exactness(clover_hip::REFERENCE_BITS);
    const double ref4_k25 = per_iteration_us(Phi, PhiT, x, y, t1, t2, t3, K25, 4, 12, 3);
    // threshold alone, REFERENCE order: the device's single-wavefront walk against the same walk on one host core
    CloverVector32 r32(N);
    r32.setRandomInteger(10, 11);
    CloverVector4 xr(N);
    xr.quantize(r32);
    (void)xr.getBits(0);
    const double dev_thr[2] = {device_threshold_us(xr, K, 9), device_threshold_us(xr, K25, 9)};
    const double host_thr[2] = {host_heap_walk_us(xr, N, K, 21), host_heap_walk_us(xr, N, K25, 21)};
    clover_hip::set_exactness(clover_hip::FAST);
    const double dev_thr_fast[2] = {device_threshold_us(xr, K, 9), device_threshold_us(xr, K25, 9)};
#ifdef CLOVER_HIP_EXPLICIT_SYNC
    const char *build = "explicit residency (-DCLOVER_HIP_EXPLICIT_SYNC)";
#else
    const char *build = "page-tracked mirrors (default)";
#endif
    std::printf("{\"build\": \"%s\", \"N\": %llu, \"M\": %llu, \"K\": %llu, "
                "\"us_per_iteration\": {\"fast\": %.2f, \"fast_v8\": %.2f, \"fast_generic_five_calls\": %.2f, \"reference_bits\": %.1f, \"reference_bits_v8\": %.1f}, "
                "\"us_per_iteration_K2048_reference_ratio\": {\"fast\": %.2f, \"fast_v8\": %.2f, \"reference_bits\": %.1f}, "
                "\"threshold_alone_us\": {\"K1024\": {\"device_reference_walk\": %.1f, \"one_host_core_same_walk\": %.1f, \"device_fast\": %.1f}, "
                "\"K2048\": {\"device_reference_walk\": %.1f, \"one_host_core_same_walk\": %.1f, \"device_fast\": %.1f}, "
                "\"note\": \"x.threshold(K) through the headers incl. the wait for the device, against the reference's heap walk (std::make_heap + "
                "min_heapify over getAbs) on one host core; the walk is sequential by definition: REFERENCE mode buys the reference's survivor set, not speed\"}}\n",
                build, (unsigned long long)N, (unsigned long long)M, (unsigned long long)K, fast4, fast8, gen[2], ref4, ref8, fast4_k25, fast8_k25, ref4_k25,
                dev_thr[0], host_thr[0], dev_thr_fast[0], dev_thr[1], host_thr[1], dev_thr_fast[1]);
    return 0;
}
