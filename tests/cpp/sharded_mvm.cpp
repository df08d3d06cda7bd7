// sharded_mvm.cpp -- clm4_sharded_mvm / clm4_sharded_gemm against the unsharded clm4_mvm / clm4_gemm through the C ABI.
//
// Runs with however many GPUs are visible (1 on the test box, 8 on a full node: then the exchange is RCCL's all-gather),
// and, on device 0 alone, with the partitions an 8-GPU / 3-GPU / 5-GPU node would get (devices listed repeatedly: same
// shard arithmetic, plain copies instead of RCCL) -- including ragged partitions whose shards are ODD multiples of 64 rows.
// The MI355X counterpart of the reference's "mvm vs mvm_parallel" check (test/validate/03_matrix.cpp:495-573).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "clover_hip.h"

#define CHECK(call)                                                                               \
    do {                                                                                          \
        int rc__ = (call);                                                                        \
        if (rc__ != CLV_OK) {                                                                     \
            std::printf("FAILED %s -> %d: %s\n", #call, rc__, clv_last_error());                  \
            std::exit(2);                                                                         \
        }                                                                                         \
    } while (0)

struct Whole {
    uint64_t rows, cols;
    int8_t *A, *x, *r;
    float *sA, *sx, *sr;
    std::vector<int8_t> r_host, x_host;
    std::vector<float> sr_host, sx_host;
};

static Whole make_whole(uint64_t rows, uint64_t cols, uint64_t seed)
{
    Whole w{rows, cols, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, {}, {}, {}, {}};
    CHECK(clv_malloc((void **)&w.A, rows * cols / 2));
    CHECK(clv_malloc((void **)&w.sA, (rows / 64) * (cols / 64) * 4));
    CHECK(clv_malloc((void **)&w.x, cols / 2));
    CHECK(clv_malloc((void **)&w.sx, cols / 16));
    CHECK(clv_malloc((void **)&w.r, rows / 2));
    CHECK(clv_malloc((void **)&w.sr, rows / 16));
    CHECK(clv_fill_random_nibbles(w.A, rows * cols / 2, seed, 0, nullptr));
    CHECK(clv_fill_random_scales(w.sA, (rows / 64) * (cols / 64), seed + 1, 0, nullptr));
    CHECK(clv_fill_random_nibbles(w.x, cols / 2, seed + 2, 0, nullptr));
    CHECK(clv_fill_random_scales(w.sx, cols / 64, seed + 3, 0, nullptr));
    CHECK(clm4_mvm(w.A, w.sA, rows, cols, w.x, w.sx, w.r, w.sr, nullptr, nullptr));
    w.r_host.resize(rows / 2); w.sr_host.resize(rows / 64); w.x_host.resize(cols / 2); w.sx_host.resize(cols / 64);
    CHECK(clv_memcpy_d2h(w.r_host.data(), w.r, rows / 2, nullptr));
    CHECK(clv_memcpy_d2h(w.sr_host.data(), w.sr, rows / 16, nullptr));
    CHECK(clv_memcpy_d2h(w.x_host.data(), w.x, cols / 2, nullptr));
    CHECK(clv_memcpy_d2h(w.sx_host.data(), w.sx, cols / 16, nullptr));
    CHECK(clv_device_sync());
    return w;
}

static void free_whole(Whole &w)
{
    void *p[] = {w.A, w.sA, w.x, w.sx, w.r, w.sr};
    for (void *q : p) CHECK(clv_free(q));
}

// one layout: nparts shards on `devices` (NULL = devices 0..nparts-1); every shard's gathered copy must equal the whole result
static int run_layout(const char *name, Whole &w, int nparts, const int *devices, uint64_t seed, bool gemm)
{
    clm4_shard_ctx *ctx = nullptr;
    CHECK(clm4_sharded_create(&ctx, nparts, devices, w.rows, w.cols));
    CHECK(clm4_sharded_fill_random(ctx, seed));
    std::vector<int8_t> r(w.rows / 2);
    std::vector<float> sr(w.rows / 64);
    int before = -1, after = -1;
    CHECK(clv_get_device(&before));
    for (int rep = 0; rep < 2; rep++) {                     // the second call re-uses every buffer and event
        std::memset(r.data(), 0x5a, r.size());
        CHECK(clm4_sharded_mvm(ctx, rep ? w.x : w.x_host.data(), rep ? w.sx : w.sx_host.data(), rep ? 0 : 1, r.data(), sr.data()));
        if (std::memcmp(r.data(), w.r_host.data(), r.size()) || std::memcmp(sr.data(), w.sr_host.data(), sr.size() * 4)) {
            std::printf("%s: gathered result differs from clm4_mvm (rep %d)\n", name, rep);
            return 1;
        }
    }
    CHECK(clv_get_device(&after));
    if (before != after) { std::printf("%s: current device changed %d -> %d\n", name, before, after); return 1; }
    int odd = 0, ranks = -1, equal = -1;
    float kmax = 0.0f, gmax = 0.0f;
    for (int p = 0; p < nparts; p++) {
        const int8_t *rd; const float *srd;
        uint64_t b, c; int dev;
        CHECK(clm4_sharded_result(ctx, p, &rd, &srd));
        CHECK(clm4_sharded_info(ctx, p, &dev, &b, &c, nullptr, nullptr));
        odd += (c / 64) & 1;
        CHECK(clv_set_device(dev));
        std::vector<int8_t> rp(w.rows / 2);
        std::vector<float> srp(w.rows / 64);
        CHECK(clv_memcpy_d2h(rp.data(), rd, rp.size(), nullptr));
        CHECK(clv_memcpy_d2h(srp.data(), srd, srp.size() * 4, nullptr));
        CHECK(clv_device_sync());
        CHECK(clv_set_device(before));
        if (std::memcmp(rp.data(), w.r_host.data(), rp.size()) || std::memcmp(srp.data(), w.sr_host.data(), srp.size() * 4)) {
            std::printf("%s: shard %d's copy of the gathered result differs\n", name, p);
            return 1;
        }
        float k, g;
        CHECK(clm4_sharded_timing(ctx, p, &k, &g));
        kmax = k > kmax ? k : kmax; gmax = g > gmax ? g : gmax;
    }
    CHECK(clm4_sharded_comm_info(ctx, &ranks, &equal));
    int gemm_ok = -1;
    if (gemm) {
        // C = A * B^T, B = the first 128 rows of an independent random matrix; shards must be multiples of 128 rows
        const uint64_t N = 128, K = w.cols;
        int8_t *B; float *sB, *C;
        CHECK(clv_malloc((void **)&B, N * K / 2));
        CHECK(clv_malloc((void **)&sB, (N / 64) * (K / 64) * 4));
        CHECK(clv_malloc((void **)&C, w.rows * N * 4));
        CHECK(clv_fill_random_nibbles(B, N * K / 2, seed + 9, 0, nullptr));
        CHECK(clv_fill_random_scales(sB, (N / 64) * (K / 64), seed + 10, 0, nullptr));
        CHECK(clm4_gemm(w.A, w.sA, w.rows, K, B, sB, N, C, nullptr));
        std::vector<float> C1(w.rows * N), C2(w.rows * N);
        CHECK(clv_memcpy_d2h(C1.data(), C, C1.size() * 4, nullptr));
        CHECK(clv_device_sync());
        CHECK(clm4_sharded_gemm(ctx, B, sB, N, 0, C2.data()));
        gemm_ok = std::memcmp(C1.data(), C2.data(), C1.size() * 4) == 0;
        // the loop form: three steps, the C row panels all-gathered (RCCL where the layout has a communicator, copies in the same-device
        // layout): EVERY shard's copy of the whole C, in the buffer the last step wrote, equals the unsharded product
        if (gemm_ok) {
            CHECK(clm4_sharded_gemm_begin(ctx, B, sB, N, 0, 3));
            for (int step = 0; step < 3; step++) CHECK(clm4_sharded_gemm_enqueue(ctx, step, 1));
            CHECK(clm4_sharded_sync(ctx));
            for (int p = 0; p < nparts && gemm_ok; p++) {
                const float *cf; int dev;
                CHECK(clm4_sharded_gemm_full(ctx, p, 2 & 1, &cf));
                CHECK(clm4_sharded_info(ctx, p, &dev, nullptr, nullptr, nullptr, nullptr));
                CHECK(clv_set_device(dev));
                CHECK(clv_memcpy_d2h(C2.data(), cf, C2.size() * 4, nullptr));
                CHECK(clv_device_sync());
                CHECK(clv_set_device(before));
                gemm_ok = std::memcmp(C1.data(), C2.data(), C1.size() * 4) == 0;
                float k, g;
                CHECK(clm4_sharded_step_timing(ctx, p, 2, &k, &g));
                if (!(k > 0.0f) || !(g >= 0.0f)) gemm_ok = 0;
            }
            if (!gemm_ok) { std::printf("%s: a shard's all-gathered C differs from clm4_gemm\n", name); return 1; }
            // the other two exchange modes (round 6): panels to shard 0's device only -- its buffer is the whole C, every other device holds
            // (at least) its own panel -- and no exchange at all: every device's buffer holds its own panel.  The buffers are poisoned first.
            for (int mode = CLM4_GEMM_GATHER_ROOT; mode <= CLM4_GEMM_SHARDED && gemm_ok; mode++) {
                CHECK(clm4_sharded_gemm_begin_mode(ctx, B, sB, N, 0, 3, mode));
                for (int p = 0; p < nparts; p++)
                    for (int buf = 0; buf < 2; buf++) {
                        const float *cf; int dev;
                        CHECK(clm4_sharded_gemm_full(ctx, p, buf, &cf));
                        CHECK(clm4_sharded_info(ctx, p, &dev, nullptr, nullptr, nullptr, nullptr));
                        CHECK(clv_set_device(dev));
                        CHECK(clv_memset((void *)cf, 0xFF, w.rows * N * 4, nullptr));
                        CHECK(clv_device_sync());
                        CHECK(clv_set_device(before));
                    }
                for (int step = 0; step < 3; step++) CHECK(clm4_sharded_gemm_enqueue(ctx, step, 1));
                CHECK(clm4_sharded_sync(ctx));
                for (int p = 0; p < nparts && gemm_ok; p++) {
                    const float *cf; int dev; uint64_t rb, rc;
                    CHECK(clm4_sharded_gemm_full(ctx, p, 2 & 1, &cf));
                    CHECK(clm4_sharded_info(ctx, p, &dev, &rb, &rc, nullptr, nullptr));
                    CHECK(clv_set_device(dev));
                    CHECK(clv_memcpy_d2h(C2.data(), cf, C2.size() * 4, nullptr));
                    CHECK(clv_device_sync());
                    CHECK(clv_set_device(before));
                    if (mode == CLM4_GEMM_GATHER_ROOT && p == 0) gemm_ok = std::memcmp(C1.data(), C2.data(), C1.size() * 4) == 0;      // the whole C
                    else gemm_ok = std::memcmp(C1.data() + rb * N, C2.data() + rb * N, rc * N * 4) == 0;                               // its own panel
                    float k, g;
                    CHECK(clm4_sharded_step_timing(ctx, p, 2, &k, &g));
                    if (!(k > 0.0f) || !(g >= 0.0f)) gemm_ok = 0;
                }
                if (!gemm_ok) { std::printf("%s: sharded GEMM mode %d differs from clm4_gemm\n", name, mode); return 1; }
            }
        }
        CHECK(clv_free(B)); CHECK(clv_free(sB)); CHECK(clv_free(C));
        if (!gemm_ok) { std::printf("%s: sharded GEMM differs from clm4_gemm\n", name); return 1; }
    }
    std::printf("%s parts=%d odd_shards=%d rccl_ranks=%d equal=%d kernel_ms=%.4f gather_ms=%.4f gemm=%d ok\n", name, nparts, odd, ranks, equal, kmax, gmax, gemm_ok);
    CHECK(clm4_sharded_destroy(ctx));
    return 0;
}

int main()
{
    int ndev = 0;
    CHECK(clv_device_count(&ndev));
    if (ndev == 0) { std::printf("no_device\n"); return 0; }
    CHECK(clv_set_device(0));
    int bad = 0;
    // 1. the node as it is: one shard per visible GPU (RCCL all-gather when ndev > 1).  8192 rows: multiples of 128 for 1,2,4,8
    Whole w = make_whole(8192, 4096, 101);
    bad += run_layout("node", w, ndev, nullptr, 101, true);
    // 2. on device 0 alone: the 8-way equal partition, and ragged ones with odd 64-row shards
    const int zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bad += run_layout("loop8", w, 8, zeros, 101, true);
    free_whole(w);
    Whole v = make_whole(1664, 2048, 202);                // 26 blocks of 64 rows: 3 -> 9,9,8   5 -> 6,5,5,5,5   7 -> 4,4,4,4,4,3,3
    bad += run_layout("loop3", v, 3, zeros, 202, false);
    bad += run_layout("loop5", v, 5, zeros, 202, false);
    bad += run_layout("loop7", v, 7, zeros, 202, false);
    free_whole(v);
    std::printf(bad ? "sharded FAILED\n" : "sharded all ok ndev=%d\n", ndev);
    return bad ? 1 : 0;
}
