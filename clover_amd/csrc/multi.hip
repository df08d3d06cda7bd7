// multi.hip -- row-sharded CloverMatrix4::mvm across the GPUs of one node (BASELINE config 5).
//
// The reference's only parallelism is mvm_parallel's contiguous split of 64-row blocks over OpenMP threads
// (CloverMatrix4.h:1700-1705).  The MI355X equivalent: contiguous row shards (multiples of 64 rows) per GPU,
// x replicated (36 KiB for 65536 columns), the packed result all-gathered over xGMI with RCCL.  No partial
// sum ever crosses a device, so the result is bit-identical to the single-GPU one -- never an all-reduce.
// The gather moves rows/2 + rows/16 bytes in total (C5: 576 KiB): latency-bound, so it is issued as one
// grouped set of broadcasts (works for unequal shards too) on each device's stream right behind its kernel.
//
// One process drives all devices (one stream + one RCCL communicator per device).  RCCL is dlopen'ed on
// first use so that libclover_hip.so itself does not depend on it.  bench.py uses the other idiom (one
// process per GPU under torch.distributed); both shard identically.
#include "common.h"

#include <dlfcn.h>

#include <vector>

// ---- the few RCCL entry points used, resolved at run time ---------------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
enum { ncclInt8 = 0, ncclFloat32 = 7 };
struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char *(*GetErrorString)(ncclResult_t);
    bool ok;
};

static RcclApi *rccl()
{
    static RcclApi api = [] {
        RcclApi a{};
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
        a.Broadcast = (decltype(a.Broadcast))dlsym(h, "ncclBroadcast");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.CommInitAll && a.CommDestroy && a.Broadcast && a.GroupStart && a.GroupEnd && a.GetErrorString;
        return a;
    }();
    return &api;
}

#define CLV_NCCL(call)                                                                                  \
    do {                                                                                                \
        ncclResult_t r__ = (call);                                                                      \
        if (r__ != 0) {                                                                                 \
            clv_set_error("%s failed: %s (%s:%d)", #call, rccl()->GetErrorString(r__), __FILE__, __LINE__); \
            return CLV_ERR_HIP;                                                                         \
        }                                                                                               \
    } while (0)

struct clm4_shard_ctx {
    int ndev = 0;
    uint64_t rows = 0, cols = 0;
    std::vector<int> dev;
    std::vector<uint64_t> row_begin, row_count;
    std::vector<int8_t *> A, x, r;          // per device: shard nibbles, x nibbles, FULL result nibbles
    std::vector<float *> sA, sx, sr;        // per device: shard tile scales, x scales, FULL result scales
    std::vector<hipStream_t> st;
    std::vector<ncclComm_t> comm;
    // GEMM (clm4_sharded_gemm): B replicated, one row shard of C per device
    uint64_t gemm_n = 0;
    std::vector<int8_t *> B;
    std::vector<float *> sB, C;
};

// contiguous shards in units of 64 rows, remainder spread over the first ranks (as a static OpenMP split)
extern "C" int clm4_shard_partition(uint64_t rows, int nparts, int part, uint64_t *row_begin, uint64_t *row_count)
{
    CLV_REQUIRE(nparts > 0 && part >= 0 && part < nparts && rows % 64 == 0 && row_begin && row_count, "clm4_shard_partition: bad argument");
    const uint64_t blocks = rows / 64, base = blocks / nparts, extra = blocks % nparts;
    const uint64_t b0 = (uint64_t)part * base + ((uint64_t)part < extra ? (uint64_t)part : extra);
    *row_begin = b0 * 64;
    *row_count = (base + ((uint64_t)part < extra ? 1 : 0)) * 64;
    return CLV_OK;
}

extern "C" int clm4_sharded_destroy(clm4_shard_ctx *c)
{
    if (!c) return CLV_OK;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < (int)c->dev.size(); d++) {
        (void)hipSetDevice(c->dev[d]);
        if (d < (int)c->comm.size() && c->comm[d]) rccl()->CommDestroy(c->comm[d]);
        if (d < (int)c->st.size() && c->st[d]) (void)hipStreamDestroy(c->st[d]);
        void *ptrs[] = {d < (int)c->A.size() ? c->A[d] : nullptr, d < (int)c->x.size() ? c->x[d] : nullptr, d < (int)c->r.size() ? c->r[d] : nullptr,
                        d < (int)c->sA.size() ? c->sA[d] : nullptr, d < (int)c->sx.size() ? c->sx[d] : nullptr, d < (int)c->sr.size() ? c->sr[d] : nullptr};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        void *gptrs[] = {d < (int)c->B.size() ? (void *)c->B[d] : nullptr, d < (int)c->sB.size() ? (void *)c->sB[d] : nullptr,
                         d < (int)c->C.size() ? (void *)c->C[d] : nullptr};
        for (void *p : gptrs) if (p) (void)hipFree(p);
    }
    (void)hipSetDevice(cur);
    delete c;
    return CLV_OK;
}

extern "C" int clm4_sharded_create(clm4_shard_ctx **out, int ndev, const int *devices, uint64_t rows, uint64_t cols)
{
    CLV_REQUIRE(out && ndev > 0, "clm4_sharded_create: bad argument");
    CLV_REQUIRE(rows % 128 == 0 && cols % 128 == 0 && rows / 64 >= (uint64_t)ndev, "clm4_sharded_create: rows=%llu cols=%llu ndev=%d",
                (unsigned long long)rows, (unsigned long long)cols, ndev);
    int avail = 0;
    CLV_HIP(hipGetDeviceCount(&avail));
    CLV_REQUIRE(ndev <= avail, "clm4_sharded_create: %d devices requested, %d visible", ndev, avail);
    int cur = 0;
    CLV_HIP(hipGetDevice(&cur));
    clm4_shard_ctx *c = new clm4_shard_ctx;
    c->ndev = ndev;
    c->rows = rows;
    c->cols = cols;
    c->dev.resize(ndev);
    c->row_begin.resize(ndev);
    c->row_count.resize(ndev);
    c->A.assign(ndev, nullptr); c->x.assign(ndev, nullptr); c->r.assign(ndev, nullptr);
    c->sA.assign(ndev, nullptr); c->sx.assign(ndev, nullptr); c->sr.assign(ndev, nullptr);
    c->st.assign(ndev, nullptr);
    c->comm.assign(ndev, nullptr);
    int rc = CLV_OK;
    for (int d = 0; d < ndev && rc == CLV_OK; d++) {
        c->dev[d] = devices ? devices[d] : d;
        clm4_shard_partition(rows, ndev, d, &c->row_begin[d], &c->row_count[d]);
        auto alloc = [&](void **p, uint64_t bytes) { return hipMalloc(p, bytes ? bytes : 1) == hipSuccess; };
        if (hipSetDevice(c->dev[d]) != hipSuccess || hipStreamCreateWithFlags(&c->st[d], hipStreamNonBlocking) != hipSuccess ||
            !alloc((void **)&c->A[d], c->row_count[d] * cols / 2) || !alloc((void **)&c->sA[d], (c->row_count[d] / 64) * (cols / 64) * 4) ||
            !alloc((void **)&c->x[d], cols / 2) || !alloc((void **)&c->sx[d], cols / 16) || !alloc((void **)&c->r[d], rows / 2) ||
            !alloc((void **)&c->sr[d], rows / 16)) {
            clv_set_error("clm4_sharded_create: device %d setup failed: %s", c->dev[d], hipGetErrorString(hipGetLastError()));
            rc = CLV_ERR_HIP;
        }
    }
    if (rc == CLV_OK && ndev > 1) {
        if (!rccl()->ok) {
            clv_set_error("clm4_sharded_create: librccl.so could not be loaded");
            rc = CLV_ERR_UNSUPPORTED;
        } else if (ncclResult_t r = rccl()->CommInitAll(c->comm.data(), ndev, c->dev.data())) {
            clv_set_error("ncclCommInitAll failed: %s", rccl()->GetErrorString(r));
            rc = CLV_ERR_HIP;
        }
    }
    (void)hipSetDevice(cur);
    if (rc != CLV_OK) { clm4_sharded_destroy(c); return rc; }
    *out = c;
    return CLV_OK;
}

extern "C" int clm4_sharded_info(const clm4_shard_ctx *c, int part, int *device, uint64_t *row_begin, uint64_t *row_count,
                                 int8_t **A_dev, float **sA_dev)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev, "clm4_sharded_info: bad argument");
    if (device) *device = c->dev[part];
    if (row_begin) *row_begin = c->row_begin[part];
    if (row_count) *row_count = c->row_count[part];
    if (A_dev) *A_dev = c->A[part];
    if (sA_dev) *sA_dev = c->sA[part];
    return CLV_OK;
}

// scatter a whole matrix (reference layout, host memory) over the shards
extern "C" int clm4_sharded_upload(clm4_shard_ctx *c, const int8_t *A_host, const float *sA_host)
{
    CLV_REQUIRE(c && A_host && sA_host, "clm4_sharded_upload: null argument");
    int cur = 0;
    CLV_HIP(hipGetDevice(&cur));
    const uint64_t hb = c->cols / 64;
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipMemcpyAsync(c->A[d], A_host + c->row_begin[d] * c->cols / 2, c->row_count[d] * c->cols / 2, hipMemcpyHostToDevice, c->st[d]));
        CLV_HIP(hipMemcpyAsync(c->sA[d], sA_host + (c->row_begin[d] / 64) * hb, (c->row_count[d] / 64) * hb * 4, hipMemcpyHostToDevice, c->st[d]));
    }
    for (int d = 0; d < c->ndev; d++) { CLV_HIP(hipSetDevice(c->dev[d])); CLV_HIP(hipStreamSynchronize(c->st[d])); }
    CLV_HIP(hipSetDevice(cur));
    return CLV_OK;
}

// synthetic matrix: the same bytes the unsharded clv_fill_random_* calls would produce for (seed, seed+1)
extern "C" int clm4_sharded_fill_random(clm4_shard_ctx *c, uint64_t seed)
{
    CLV_REQUIRE(c, "clm4_sharded_fill_random: null argument");
    int cur = 0;
    CLV_HIP(hipGetDevice(&cur));
    const uint64_t hb = c->cols / 64;
    int rc = CLV_OK;
    for (int d = 0; d < c->ndev && rc == CLV_OK; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        rc = clv_fill_random_nibbles(c->A[d], c->row_count[d] * c->cols / 2, seed, c->row_begin[d] * c->cols / 2, c->st[d]);
        if (rc == CLV_OK) rc = clv_fill_random_scales(c->sA[d], (c->row_count[d] / 64) * hb, seed + 1, (c->row_begin[d] / 64) * hb, c->st[d]);
    }
    for (int d = 0; d < c->ndev; d++) { CLV_HIP(hipSetDevice(c->dev[d])); CLV_HIP(hipStreamSynchronize(c->st[d])); }
    CLV_HIP(hipSetDevice(cur));
    return rc;
}

// r = A * x on all shards, result all-gathered so that EVERY device ends up with the full packed result.
// x / sx: cols/2 bytes + cols/64 floats in host memory (x_on_host != 0) or on device part 0.
// r_host / sr_host (optional): receive rows/2 bytes + rows/64 floats.
extern "C" int clm4_sharded_mvm(clm4_shard_ctx *c, const int8_t *x, const float *sx, int x_on_host, int8_t *r_host, float *sr_host)
{
    CLV_REQUIRE(c && x && sx, "clm4_sharded_mvm: null argument");
    int cur = 0;
    CLV_HIP(hipGetDevice(&cur));
    const uint64_t cols = c->cols;
    // 1. x to device 0, then to everyone
    CLV_HIP(hipSetDevice(c->dev[0]));
    const hipMemcpyKind kind = x_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    CLV_HIP(hipMemcpyAsync(c->x[0], x, cols / 2, kind, c->st[0]));
    CLV_HIP(hipMemcpyAsync(c->sx[0], sx, cols / 16, kind, c->st[0]));
    if (c->ndev > 1) {
        CLV_NCCL(rccl()->GroupStart());
        for (int d = 0; d < c->ndev; d++) {
            CLV_NCCL(rccl()->Broadcast(c->x[d], c->x[d], cols / 2, ncclInt8, 0, c->comm[d], c->st[d]));
            CLV_NCCL(rccl()->Broadcast(c->sx[d], c->sx[d], cols / 64, ncclFloat32, 0, c->comm[d], c->st[d]));
        }
        CLV_NCCL(rccl()->GroupEnd());
    }
    // 2. every shard multiplies, writing its slice of its own copy of the full result
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        int rc = clm4_mvm(c->A[d], c->sA[d], c->row_count[d], cols, c->x[d], c->sx[d], c->r[d] + c->row_begin[d] / 2,
                          c->sr[d] + c->row_begin[d] / 64, nullptr, c->st[d]);
        if (rc != CLV_OK) { (void)hipSetDevice(cur); return rc; }
    }
    // 3. all-gather of the packed slices = one broadcast per owner, grouped
    if (c->ndev > 1) {
        CLV_NCCL(rccl()->GroupStart());
        for (int root = 0; root < c->ndev; root++)
            for (int d = 0; d < c->ndev; d++) {
                int8_t *pr = c->r[d] + c->row_begin[root] / 2;
                float *ps = c->sr[d] + c->row_begin[root] / 64;
                CLV_NCCL(rccl()->Broadcast(pr, pr, c->row_count[root] / 2, ncclInt8, root, c->comm[d], c->st[d]));
                CLV_NCCL(rccl()->Broadcast(ps, ps, c->row_count[root] / 64, ncclFloat32, root, c->comm[d], c->st[d]));
            }
        CLV_NCCL(rccl()->GroupEnd());
    }
    for (int d = 0; d < c->ndev; d++) { CLV_HIP(hipSetDevice(c->dev[d])); CLV_HIP(hipStreamSynchronize(c->st[d])); }
    if (r_host) { CLV_HIP(hipSetDevice(c->dev[0])); CLV_HIP(hipMemcpy(r_host, c->r[0], c->rows / 2, hipMemcpyDeviceToHost)); }
    if (sr_host) { CLV_HIP(hipSetDevice(c->dev[0])); CLV_HIP(hipMemcpy(sr_host, c->sr[0], c->rows / 16, hipMemcpyDeviceToHost)); }
    CLV_HIP(hipSetDevice(cur));
    return CLV_OK;
}

// device pointers of the full result held by shard `part` (valid after clm4_sharded_mvm)
extern "C" int clm4_sharded_result(const clm4_shard_ctx *c, int part, const int8_t **r_dev, const float **sr_dev)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev, "clm4_sharded_result: bad argument");
    if (r_dev) *r_dev = c->r[part];
    if (sr_dev) *sr_dev = c->sr[part];
    return CLV_OK;
}


// C = A * B^T with A the sharded matrix (rows x cols) and B an N x cols CloverMatrix4 replicated on every device: device d ends
// with rows [row_begin_d, +row_count_d) of C (fp32, row-major, N columns).  Every element of C is its own fma chain over the
// K-blocks, so the shards equal the unsharded clm4_gemm bit for bit and nothing needs to be exchanged; C_host
// (optional) receives the whole C.  clm4_gemm works in units of 128 rows: every shard must be a multiple of 128.
extern "C" int clm4_sharded_gemm(clm4_shard_ctx *c, const int8_t *B, const float *sB, uint64_t N, int b_on_host, float *C_host)
{
    CLV_REQUIRE(c && B && sB && N && N % 128 == 0, "clm4_sharded_gemm: bad argument");
    for (int d = 0; d < c->ndev; d++)
        CLV_REQUIRE(c->row_count[d] % 128 == 0, "clm4_sharded_gemm: shard %d has %llu rows, not a multiple of 128", d, (unsigned long long)c->row_count[d]);
    int cur = 0;
    CLV_HIP(hipGetDevice(&cur));
    const uint64_t K = c->cols, b_bytes = N * K / 2, sb_count = (N / 64) * (K / 64);
    if (c->gemm_n != N) {                                      // (re)allocate B and the C shards for this N
        c->B.resize(c->ndev, nullptr); c->sB.resize(c->ndev, nullptr); c->C.resize(c->ndev, nullptr);
        for (int d = 0; d < c->ndev; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            CLV_HIP(hipStreamSynchronize(c->st[d]));
            if (c->B[d]) CLV_HIP(hipFree(c->B[d]));
            if (c->sB[d]) CLV_HIP(hipFree(c->sB[d]));
            if (c->C[d]) CLV_HIP(hipFree(c->C[d]));
            c->B[d] = nullptr; c->sB[d] = nullptr; c->C[d] = nullptr;
            CLV_HIP(hipMalloc((void **)&c->B[d], b_bytes));
            CLV_HIP(hipMalloc((void **)&c->sB[d], sb_count * sizeof(float)));
            CLV_HIP(hipMalloc((void **)&c->C[d], c->row_count[d] * N * sizeof(float)));
        }
        c->gemm_n = N;
    }
    // 1. B to device 0, then to everyone
    CLV_HIP(hipSetDevice(c->dev[0]));
    const hipMemcpyKind kind = b_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    CLV_HIP(hipMemcpyAsync(c->B[0], B, b_bytes, kind, c->st[0]));
    CLV_HIP(hipMemcpyAsync(c->sB[0], sB, sb_count * sizeof(float), kind, c->st[0]));
    if (c->ndev > 1) {
        CLV_NCCL(rccl()->GroupStart());
        for (int d = 0; d < c->ndev; d++) {
            CLV_NCCL(rccl()->Broadcast(c->B[d], c->B[d], b_bytes, ncclInt8, 0, c->comm[d], c->st[d]));
            CLV_NCCL(rccl()->Broadcast(c->sB[d], c->sB[d], sb_count, ncclFloat32, 0, c->comm[d], c->st[d]));
        }
        CLV_NCCL(rccl()->GroupEnd());
    }
    // 2. every device multiplies its row shard
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        int rc = clm4_gemm(c->A[d], c->sA[d], c->row_count[d], K, c->B[d], c->sB[d], N, c->C[d], c->st[d]);
        if (rc != CLV_OK) { (void)hipSetDevice(cur); return rc; }
    }
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipStreamSynchronize(c->st[d]));
        if (C_host) CLV_HIP(hipMemcpy(C_host + c->row_begin[d] * N, c->C[d], c->row_count[d] * N * sizeof(float), hipMemcpyDeviceToHost));
    }
    CLV_HIP(hipSetDevice(cur));
    return CLV_OK;
}

// device pointer of shard `part`'s rows of C (valid after clm4_sharded_gemm)
extern "C" int clm4_sharded_gemm_result(const clm4_shard_ctx *c, int part, const float **C_dev)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev && C_dev && c->gemm_n, "clm4_sharded_gemm_result: bad argument");
    *C_dev = c->C[part];
    return CLV_OK;
}
