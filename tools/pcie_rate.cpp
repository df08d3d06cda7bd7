// pcie_rate.cpp -- what a caller of the C++ containers sees when the data starts and ends in HOST memory (the PCIe-inclusive
// rate, never bench.py's `value`).  BASELINE configs[1]: n = 2^24, quantize + dot.  Build and run on the GPU box:
//   g++ -std=c++11 -O2 -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1 -Iinclude tools/pcie_rate.cpp -o /tmp/pcie_rate -Lclover_amd/lib -lclover_hip \
//       -Wl,-rpath,$PWD/clover_amd/lib -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib && /tmp/pcie_rate
#include <chrono>
#include <cstdio>

#include "CloverVector32.h"
#include "CloverVector4.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }
    const uint64_t n = 1ull << 24;
    CloverVector32 x(n), y(n);
    float *px = x.getData(), *py = y.getData();
    for (uint64_t i = 0; i < n; i++) { px[i] = (float)((int)(i * 2654435761u >> 8) % 21 - 10); py[i] = (float)((int)(i * 40503u >> 4) % 21 - 10); }
    CloverVector4 qx(n), qy(n);
    qy.quantize(y);
    for (int rep = 0; rep < 3; rep++) {
        ((volatile float *)px)[0] = px[0];                      // host write: the device copy of x is stale, the next quantize uploads 64 MiB
        const double t0 = now();
        qx.quantize(x);                                         // upload + kernel
        const double t1 = now();
        const volatile int8_t first = qx.getData()[0];          // pull the 9 MiB result back (fault -> device -> host copy)
        (void)first;
        const double t2 = now();
        const float d = qx.dot_parallel(qy);                    // operands resident: launch pair + 4-byte read-back
        const double t3 = now();
        std::printf("rep %d: quantize from host memory %.3f ms (%.1f GB/s of fp32 source over PCIe + kernel), result back in host memory +%.3f ms, "
                    "dot_parallel with resident operands %.3f ms (dot = %g)\n", rep, (t1 - t0) * 1e3, 4.0 * n / (t1 - t0) / 1e9, (t2 - t1) * 1e3,
                    (t3 - t2) * 1e3, (double)d);
    }
    return 0;
}
