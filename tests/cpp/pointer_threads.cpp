// pointer_threads.cpp -- the containers' raw-pointer contract under host threads, on the GPU (the state machine itself
// is stress-tested without a GPU in mirror_threads.cpp).  In the reference any number of threads may read through getData() /
// getScales() pointers (CloverVector4.h:229-237) while others compute with the same read-only operands.  Here that means concurrent
// first uploads of shared operands, concurrent faults on one device-written block, concurrent writes into one uploaded block.
// Prints "pointer threads ok" or the first failures.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "CloverMatrix4.h"
#include "CloverVector32.h"
#include "CloverVector4.h"

static std::atomic<int> failures(0);
#define EXPECT(cond) do { if (!(cond)) { if (failures.fetch_add(1) < 10) std::printf("FAILED line %d: %s\n", __LINE__, #cond); } } while (0)

struct Barrier {
    std::atomic<int> count, generation;
    const int n;
    explicit Barrier(int n_) : count(0), generation(0), n(n_) {}
    void wait()
    {
        const int g = generation.load();
        if (count.fetch_add(1) + 1 == n) { count.store(0); generation.fetch_add(1); }
        else while (generation.load() == g) std::this_thread::yield();
    }
};

static const int T = 4;

int main()
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }
    const uint64_t M = 1024, N = 2048;
    CloverMatrix32 A32(M, N);
    CloverVector32 x32(N);
    {
        float *a = A32.getData(), *x = x32.getData();
        unsigned s = 12345;
        for (uint64_t i = 0; i < M * N; i++) { s = s * 1664525u + 1013904223u; a[i] = (float)((int)(s >> 20) % 21 - 10); }
        for (uint64_t i = 0; i < N; i++) { s = s * 1664525u + 1013904223u; x[i] = (float)((int)(s >> 20) % 21 - 10); }
    }
    CloverMatrix4 A4(M, N);
    CloverVector4 x4(N), ref(M);
    A4.quantize(A32);
    x4.quantize(x32);
    A4.mvm(x4, ref);
    std::vector<int8_t> want(ref.getData(), ref.getData() + M / 2);
    std::vector<float> want_s(ref.getScales(), ref.getScales() + M / 64);

    // 1. four threads multiply with the SAME matrix and vector, each into its own result, and read the result through a pointer taken
    //    before the call; two more threads keep reading the shared operands' host blocks.  The operands' host blocks are made dirty
    //    before every round (a write through the kept pointer), so the four mvm calls race for the upload.
    {
        int8_t *qa = A4.getData();
        int8_t *qx = x4.getData();
        const int rounds = 40;
        Barrier bar(T + 1);
        std::atomic<bool> stop(false);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&] {
                CloverVector4 r(M);
                const int8_t *qr = r.getData();
                const float *sr = r.getScales();
                for (int k = 0; k < rounds; k++) {
                    bar.wait();
                    A4.mvm(x4, r);
                    EXPECT(std::memcmp(qr, want.data(), M / 2) == 0 && std::memcmp(sr, want_s.data(), M / 16) == 0);
                    bar.wait();
                }
            });
        std::vector<std::thread> readers;
        for (int t = 0; t < 2; t++)
            readers.emplace_back([&] {
                unsigned long sum = 0;
                while (!stop.load()) {
                    for (uint64_t i = 0; i < M * N / 2; i += 4096) sum += (uint8_t)qa[i];
                    for (uint64_t i = 0; i < N / 2; i += 64) sum += (uint8_t)qx[i];
                }
                if (sum == 1) std::printf(" ");
            });
        for (int k = 0; k < rounds; k++) {
            qa[0] = qa[0];                       // host write: the matrix's device copy is stale again (same bytes)
            qx[0] = qx[0];
            bar.wait();
            bar.wait();
        }
        stop.store(true);
        for (auto &x : th) x.join();
        for (auto &x : readers) x.join();
    }
    // 2. one device-written vector, four threads reading it through ONE kept pointer at the same moment: all of them see the result
    {
        const uint64_t n = 1 << 20;
        CloverVector32 src(n), back(n);
        float *ps = src.getData();
        const float *pb = back.getData();
        CloverVector4 q(n);
        Barrier bar(T + 1);
        const int rounds = 30;
        std::atomic<int> version(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int k = 0; k < rounds; k++) {
                    bar.wait();
                    const float wantv = (float)(version.load() % 7 + 1);
                    unsigned long bad = 0;
                    for (uint64_t i = (uint64_t)t * 17; i < n; i += 257) bad += pb[i] != wantv;
                    EXPECT(bad == 0);
                    bar.wait();
                }
            });
        for (int k = 0; k < rounds; k++) {
            version.store(k);
            const float v = (float)(k % 7 + 1);
            for (uint64_t i = 0; i < n; i++) ps[i] = v;          // host writes through the kept pointer (fault on the first one)
            q.quantize(src);                                     // device
            q.restore(back);                                     // device writes `back`: its host block is stale now
            bar.wait();
            bar.wait();
        }
        for (auto &x : th) x.join();
    }
    // 3. four threads write their quarter of an uploaded (read-only) fp32 block at the same moment; the next quantize sees all of it
    {
        const uint64_t n = 1 << 16;
        CloverVector32 src(n);
        CloverVector4 q(n);
        float *ps = src.getData();
        for (uint64_t i = 0; i < n; i++) ps[i] = 1.0f;
        Barrier bar(T + 1);
        const int rounds = 50;
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int k = 0; k < rounds; k++) {
                    bar.wait();
                    for (uint64_t i = (uint64_t)t * (n / T); i < (uint64_t)(t + 1) * (n / T); i += 64) ps[i] = (float)(2 + t + k % 3);
                    bar.wait();
                }
            });
        for (int k = 0; k < rounds; k++) {
            q.quantize(src);                                     // uploads: src is SHARED (read-only) now
            bar.wait();
            bar.wait();
            q.quantize(src);
            const float *sc = q.getScales();
            unsigned long bad = 0;
            for (int t = 0; t < T; t++)
                for (uint64_t b = (uint64_t)t * (n / T) / 64; b < (uint64_t)(t + 1) * (n / T) / 64; b++) bad += sc[b] != (float)(2 + t + k % 3);
            EXPECT(bad == 0);
        }
        for (auto &x : th) x.join();
    }
    std::printf(failures.load() ? "pointer threads FAILED\n" : "pointer threads ok\n");
    return failures.load() ? 1 : 0;
}
