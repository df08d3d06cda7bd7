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
#ifdef CLOVER_HIP_EXPLICIT_SYNC
    const char *build = "explicit residency (-DCLOVER_HIP_EXPLICIT_SYNC)";
#else
    const char *build = "page-tracked mirrors (default)";
#endif
    std::printf("{\"build\": \"%s\", \"N\": %llu, \"M\": %llu, \"K\": %llu, "
                "\"us_per_iteration\": {\"fast\": %.2f, \"fast_v8\": %.2f, \"fast_generic_five_calls\": %.2f, \"reference_bits\": %.1f, \"reference_bits_v8\": %.1f}}\n",
                build, (unsigned long long)N, (unsigned long long)M, (unsigned long long)K, fast4, fast8, gen[2], ref4, ref8);
    return 0;
}
