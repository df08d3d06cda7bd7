// multi.hip -- row-sharded CloverMatrix4::mvm across the GPUs of one node (BASELINE config 5).
//
// The reference's only parallelism is mvm_parallel's contiguous split of 64-row blocks over OpenMP threads
// (CloverMatrix4.h:1700-1705).  The MI355X equivalent: contiguous row shards (multiples of 64 rows) per GPU,
// x replicated (36 KiB for 65536 columns), the packed result all-gathered over xGMI with RCCL.  No partial
// sum ever crosses a device, so the result is bit-identical to the single-GPU one -- never an all-reduce.
// The gather moves rows/2 + rows/16 bytes in total (C5: 576 KiB): latency-bound.  Equal shards (every BASELINE
// configuration): ONE grouped pair of in-place ncclAllGather (nibbles, scales) per device, right behind its kernel on
// the device's stream.  Ragged shards (rows/64 not divisible by ndev): one grouped set of broadcasts per owner.
// Shards that repeat a device (a test layout: all the shard arithmetic on one GPU) exchange with plain copies.
//
// One process drives all devices (one stream + one RCCL communicator per device).  RCCL is dlopen'ed on
// first use so that libclover_hip.so itself does not depend on it.  bench.py uses the other idiom (one
// process per GPU under torch.distributed); both shard identically.
#include "common.h"

#include <dlfcn.h>
#include <string.h>

#include <vector>

// ---- the few RCCL entry points used, resolved at run time ---------------------------------------------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
enum { ncclInt8 = 0, ncclFloat32 = 7 };
struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t);      // optional (gather-to-root of the GEMM panels)
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int *);      // optional
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
        a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
        a.Send = (decltype(a.Send))dlsym(h, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(h, "ncclRecv");
        a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
        a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
        a.ok = a.CommInitAll && a.CommDestroy && a.Broadcast && a.AllGather && a.GroupStart && a.GroupEnd && a.GetErrorString;
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

// restores the caller's current device on every return path
struct DeviceGuard {
    int cur = -1;
    DeviceGuard() { if (hipGetDevice(&cur) != hipSuccess) cur = -1; }
    ~DeviceGuard() { if (cur >= 0) (void)hipSetDevice(cur); }
};

// an RCCL group that is closed on every return path (an error between GroupStart and GroupEnd must not leave it open)
struct RcclGroup {
    bool open = false;
    ncclResult_t start() { ncclResult_t r = rccl()->GroupStart(); open = (r == 0); return r; }
    ncclResult_t end() { open = false; return rccl()->GroupEnd(); }
    ~RcclGroup() { if (open) (void)rccl()->GroupEnd(); }
};

struct clm4_shard_ctx {
    int ndev = 0;
    bool loopback = false;                  // some device listed twice: exchanges are plain copies, no RCCL (test layout)
    bool equal = false;                     // all shards have the same number of rows: the gather is one ncclAllGather pair
    bool use_rccl = false;                  // exchanges go through RCCL: more than one device -- or ONE device under
                                            // CLV_SHARDED_RCCL_SELFTEST=1 (a communicator of one rank: the same calls, groups and in-place
                                            // buffers as on a node, so that the RCCL path has run at least once where only one GPU exists);
                                            // CLV_SHARDED_RCCL_SELFTEST=ragged takes the per-owner broadcast form instead of the all-gather
    std::vector<hipEvent_t> ev;             // 3 per shard: before the kernel, after it, after the gather
    int rccl_ranks = 0;                     // communicator size RCCL reports having built (0: no communicator)
    uint64_t rows = 0, cols = 0;
    std::vector<int> dev;
    std::vector<uint64_t> row_begin, row_count;
    std::vector<int8_t *> A, x, r;          // per device: shard nibbles, x nibbles, FULL result nibbles
    std::vector<float *> sA, sx, sr;        // per device: shard tile scales, x scales, FULL result scales
    std::vector<hipStream_t> st;
    std::vector<ncclComm_t> comm;
    // timed-loop form (clm4_sharded_loop_begin / _mvm_enqueue): a second result buffer and an exchange stream per device, so the
    // gather of step i runs beside the kernel of step i+1; nothing in it synchronises with the host
    bool loop_ready = false;                // clm4_sharded_loop_begin built cs / r2 / sr2 / kdone / gdone completely
    int slots = 0;                          // steps whose events are kept (3 events per step and device)
    std::vector<hipStream_t> cs;            // per device: the stream the exchanges of the loop run on
    std::vector<int8_t *> r2;               // per device: FULL result nibbles, buffer 1 (buffer 0 is r)
    std::vector<float *> sr2;
    std::vector<hipEvent_t> kdone, gdone;   // [2*d + buf]: kernel of the step that last wrote buf finished / its gather finished
    std::vector<char> gpending;             // [2*d + buf]: gdone was recorded
    std::vector<hipEvent_t> slot_ev;        // [(3*step + e) * ndev + d]
    // GEMM (clm4_sharded_gemm): B replicated, one row shard of C per device
    uint64_t gemm_n = 0, gemm_b_n = 0;      // N the C shards / the replicated B are allocated for
    std::vector<int8_t *> B;
    std::vector<float *> sB, C;
    // GEMM loop form (clm4_sharded_gemm_begin / _enqueue): two FULL C buffers per device, [2*d + buf]; device d computes its row panel
    // in place and the panels are all-gathered on the exchange stream ("shard rows of A, replicate B, all-gather C row panels")
    uint64_t gemm_loop_n = 0;
    int gemm_mode = 0;          // CLM4_GEMM_ALL_GATHER / _GATHER_ROOT / _SHARDED: what clm4_sharded_gemm_enqueue exchanges
    std::vector<float *> Cf;
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
    DeviceGuard guard;
    for (int d = 0; d < (int)c->dev.size(); d++) {
        (void)hipSetDevice(c->dev[d]);
        if (d < (int)c->comm.size() && c->comm[d]) rccl()->CommDestroy(c->comm[d]);
        if (d < (int)c->st.size() && c->st[d]) {
            clv_internal_workspace_forget(c->st[d]);         // the FP6 images clm4_gemm keeps per (device, stream)
            (void)hipStreamDestroy(c->st[d]);
        }
        if (d < (int)c->cs.size() && c->cs[d]) (void)hipStreamDestroy(c->cs[d]);
        for (int e = 0; e < 2; e++) {
            if (2 * d + e < (int)c->kdone.size() && c->kdone[2 * d + e]) (void)hipEventDestroy(c->kdone[2 * d + e]);
            if (2 * d + e < (int)c->gdone.size() && c->gdone[2 * d + e]) (void)hipEventDestroy(c->gdone[2 * d + e]);
        }
        for (size_t k = d; k < c->slot_ev.size(); k += c->dev.size())
            if (c->slot_ev[k]) (void)hipEventDestroy(c->slot_ev[k]);
        if (d < (int)c->r2.size() && c->r2[d]) (void)hipFree(c->r2[d]);
        if (d < (int)c->sr2.size() && c->sr2[d]) (void)hipFree(c->sr2[d]);
        for (int e = 0; e < 3; e++)
            if (3 * d + e < (int)c->ev.size() && c->ev[3 * d + e]) (void)hipEventDestroy(c->ev[3 * d + e]);
        void *ptrs[] = {d < (int)c->A.size() ? c->A[d] : nullptr, d < (int)c->x.size() ? c->x[d] : nullptr, d < (int)c->r.size() ? c->r[d] : nullptr,
                        d < (int)c->sA.size() ? c->sA[d] : nullptr, d < (int)c->sx.size() ? c->sx[d] : nullptr, d < (int)c->sr.size() ? c->sr[d] : nullptr};
        for (void *p : ptrs) if (p) (void)hipFree(p);
        void *gptrs[] = {d < (int)c->B.size() ? (void *)c->B[d] : nullptr, d < (int)c->sB.size() ? (void *)c->sB[d] : nullptr,
                         d < (int)c->C.size() ? (void *)c->C[d] : nullptr};
        for (void *p : gptrs) if (p) (void)hipFree(p);
        for (int e = 0; e < 2; e++)
            if (2 * d + e < (int)c->Cf.size() && c->Cf[2 * d + e]) (void)hipFree(c->Cf[2 * d + e]);
    }
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
    DeviceGuard guard;
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
    c->ev.assign(3 * (size_t)ndev, nullptr);
    int rc = CLV_OK;
    for (int d = 0; d < ndev && rc == CLV_OK; d++) {
        c->dev[d] = devices ? devices[d] : d;
        if (c->dev[d] < 0 || c->dev[d] >= avail) {
            clv_set_error("clm4_sharded_create: device %d requested, %d visible", c->dev[d], avail);
            rc = CLV_ERR_INVALID;
            break;
        }
        for (int e = 0; e < d; e++) c->loopback |= (c->dev[e] == c->dev[d]);
        clm4_shard_partition(rows, ndev, d, &c->row_begin[d], &c->row_count[d]);
        auto alloc = [&](void **p, uint64_t bytes) { return hipMalloc(p, bytes ? bytes : 1) == hipSuccess; };
        bool ok = hipSetDevice(c->dev[d]) == hipSuccess && hipStreamCreateWithFlags(&c->st[d], hipStreamNonBlocking) == hipSuccess;
        for (int e = 0; e < 3 && ok; e++) ok = hipEventCreate(&c->ev[3 * d + e]) == hipSuccess;
        if (!ok || !alloc((void **)&c->A[d], c->row_count[d] * cols / 2) || !alloc((void **)&c->sA[d], (c->row_count[d] / 64) * (cols / 64) * 4) ||
            !alloc((void **)&c->x[d], cols / 2) || !alloc((void **)&c->sx[d], cols / 16) || !alloc((void **)&c->r[d], rows / 2) ||
            !alloc((void **)&c->sr[d], rows / 16)) {
            clv_set_error("clm4_sharded_create: device %d setup failed: %s", c->dev[d], hipGetErrorString(hipGetLastError()));
            rc = CLV_ERR_HIP;
        }
    }
    c->equal = (rows / 64) % (uint64_t)ndev == 0;
    const char *selftest = getenv("CLV_SHARDED_RCCL_SELFTEST");
    c->use_rccl = !c->loopback && (ndev > 1 || (selftest && selftest[0] && selftest[0] != '0'));
    if (c->use_rccl && ndev == 1 && !strcmp(selftest, "ragged")) c->equal = false;
    if (rc == CLV_OK && c->use_rccl) {
        if (!rccl()->ok) {
            clv_set_error("clm4_sharded_create: librccl.so could not be loaded");
            rc = CLV_ERR_UNSUPPORTED;
        } else if (ncclResult_t r = rccl()->CommInitAll(c->comm.data(), ndev, c->dev.data())) {
            clv_set_error("ncclCommInitAll failed: %s", rccl()->GetErrorString(r));
            rc = CLV_ERR_HIP;
        } else {
            c->rccl_ranks = ndev;
            if (rccl()->CommCount) (void)rccl()->CommCount(c->comm[0], &c->rccl_ranks);
        }
    }
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
    DeviceGuard guard;
    const uint64_t hb = c->cols / 64;
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipMemcpyAsync(c->A[d], A_host + c->row_begin[d] * c->cols / 2, c->row_count[d] * c->cols / 2, hipMemcpyHostToDevice, c->st[d]));
        CLV_HIP(hipMemcpyAsync(c->sA[d], sA_host + (c->row_begin[d] / 64) * hb, (c->row_count[d] / 64) * hb * 4, hipMemcpyHostToDevice, c->st[d]));
    }
    for (int d = 0; d < c->ndev; d++) { CLV_HIP(hipSetDevice(c->dev[d])); CLV_HIP(hipStreamSynchronize(c->st[d])); }
    return CLV_OK;
}

// synthetic matrix: the same bytes the unsharded clv_fill_random_* calls would produce for (seed, seed+1)
extern "C" int clm4_sharded_fill_random(clm4_shard_ctx *c, uint64_t seed)
{
    CLV_REQUIRE(c, "clm4_sharded_fill_random: null argument");
    DeviceGuard guard;
    const uint64_t hb = c->cols / 64;
    int rc = CLV_OK;
    for (int d = 0; d < c->ndev && rc == CLV_OK; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        rc = clv_fill_random_nibbles(c->A[d], c->row_count[d] * c->cols / 2, seed, c->row_begin[d] * c->cols / 2, c->st[d]);
        if (rc == CLV_OK) rc = clv_fill_random_scales(c->sA[d], (c->row_count[d] / 64) * hb, seed + 1, (c->row_begin[d] / 64) * hb, c->st[d]);
    }
    for (int d = 0; d < c->ndev; d++) { CLV_HIP(hipSetDevice(c->dev[d])); CLV_HIP(hipStreamSynchronize(c->st[d])); }
    return rc;
}

// replicate `bytes` at p[0] (already enqueued on st[0]) to p[d] for every shard
static int replicate_from_0(clm4_shard_ctx *c, void *const *p, uint64_t bytes)
{
    if (c->ndev == 1 && !c->use_rccl) return CLV_OK;
    if (c->loopback) {
        // same-process test layout: order the copies behind shard 0's stream with an event
        CLV_HIP(hipSetDevice(c->dev[0]));
        CLV_HIP(hipEventRecord(c->ev[2], c->st[0]));
        for (int d = 1; d < c->ndev; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            CLV_HIP(hipStreamWaitEvent(c->st[d], c->ev[2], 0));
            CLV_HIP(hipMemcpyAsync(p[d], p[0], bytes, hipMemcpyDeviceToDevice, c->st[d]));
        }
        return CLV_OK;
    }
    RcclGroup g;
    CLV_NCCL(g.start());
    for (int d = 0; d < c->ndev; d++) CLV_NCCL(rccl()->Broadcast(p[d], p[d], bytes, ncclInt8, 0, c->comm[d], c->st[d]));
    CLV_NCCL(g.end());
    return CLV_OK;
}

// r = A * x on all shards, result all-gathered so that EVERY device ends up with the full packed result.
// x / sx: cols/2 bytes + cols/64 floats in host memory (x_on_host != 0) or on device part 0.
// r_host / sr_host (optional): receive rows/2 bytes + rows/64 floats.
extern "C" int clm4_sharded_mvm(clm4_shard_ctx *c, const int8_t *x, const float *sx, int x_on_host, int8_t *r_host, float *sr_host)
{
    CLV_REQUIRE(c && x && sx, "clm4_sharded_mvm: null argument");
    DeviceGuard guard;
    const uint64_t cols = c->cols;
    const int n = c->ndev;
    // 1. x to shard 0, then to everyone
    CLV_HIP(hipSetDevice(c->dev[0]));
    const hipMemcpyKind kind = x_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    CLV_HIP(hipMemcpyAsync(c->x[0], x, cols / 2, kind, c->st[0]));
    CLV_HIP(hipMemcpyAsync(c->sx[0], sx, cols / 16, kind, c->st[0]));
    int rc = replicate_from_0(c, (void *const *)c->x.data(), cols / 2);
    if (rc == CLV_OK) rc = replicate_from_0(c, (void *const *)c->sx.data(), cols / 16);
    if (rc != CLV_OK) return rc;
    // 2. every shard multiplies, writing its slice of its own copy of the full result
    for (int d = 0; d < n; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipEventRecord(c->ev[3 * d + 0], c->st[d]));
        rc = clm4_mvm(c->A[d], c->sA[d], c->row_count[d], cols, c->x[d], c->sx[d], c->r[d] + c->row_begin[d] / 2,
                      c->sr[d] + c->row_begin[d] / 64, nullptr, c->st[d]);
        if (rc != CLV_OK) return rc;
        CLV_HIP(hipEventRecord(c->ev[3 * d + 1], c->st[d]));
    }
    // 3. all-gather of the packed slices
    if (n > 1 && c->loopback) {
        // every shard's slice is complete at its ev[1]; the consumers wait for it and copy
        for (int d = 0; d < n; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            for (int o = 0; o < n; o++) {
                if (o == d) continue;
                CLV_HIP(hipStreamWaitEvent(c->st[d], c->ev[3 * o + 1], 0));
                CLV_HIP(hipMemcpyAsync(c->r[d] + c->row_begin[o] / 2, c->r[o] + c->row_begin[o] / 2, c->row_count[o] / 2, hipMemcpyDeviceToDevice, c->st[d]));
                CLV_HIP(hipMemcpyAsync(c->sr[d] + c->row_begin[o] / 64, c->sr[o] + c->row_begin[o] / 64, c->row_count[o] / 16, hipMemcpyDeviceToDevice, c->st[d]));
            }
        }
    } else if (c->use_rccl && c->equal) {
        // in place: rank d's contribution already sits at offset d * count of its receive buffer
        const uint64_t rc_rows = c->row_count[0];
        RcclGroup g;
        CLV_NCCL(g.start());
        for (int d = 0; d < n; d++) {
            CLV_NCCL(rccl()->AllGather(c->r[d] + c->row_begin[d] / 2, c->r[d], rc_rows / 2, ncclInt8, c->comm[d], c->st[d]));
            CLV_NCCL(rccl()->AllGather(c->sr[d] + c->row_begin[d] / 64, c->sr[d], rc_rows / 64, ncclFloat32, c->comm[d], c->st[d]));
        }
        CLV_NCCL(g.end());
    } else if (c->use_rccl) {
        RcclGroup g;
        CLV_NCCL(g.start());
        for (int root = 0; root < n; root++)
            for (int d = 0; d < n; d++) {
                int8_t *pr = c->r[d] + c->row_begin[root] / 2;
                float *ps = c->sr[d] + c->row_begin[root] / 64;
                CLV_NCCL(rccl()->Broadcast(pr, pr, c->row_count[root] / 2, ncclInt8, root, c->comm[d], c->st[d]));
                CLV_NCCL(rccl()->Broadcast(ps, ps, c->row_count[root] / 64, ncclFloat32, root, c->comm[d], c->st[d]));
            }
        CLV_NCCL(g.end());
    }
    for (int d = 0; d < n; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipEventRecord(c->ev[3 * d + 2], c->st[d]));
    }
    for (int d = 0; d < n; d++) { CLV_HIP(hipSetDevice(c->dev[d])); CLV_HIP(hipStreamSynchronize(c->st[d])); }
    if (r_host) { CLV_HIP(hipSetDevice(c->dev[0])); CLV_HIP(hipMemcpy(r_host, c->r[0], c->rows / 2, hipMemcpyDeviceToHost)); }
    if (sr_host) { CLV_HIP(hipSetDevice(c->dev[0])); CLV_HIP(hipMemcpy(sr_host, c->sr[0], c->rows / 16, hipMemcpyDeviceToHost)); }
    return CLV_OK;
}

// what the last clm4_sharded_mvm spent on shard `part`: its kernel and the exchange behind it (HIP events on the shard's stream)
extern "C" int clm4_sharded_timing(const clm4_shard_ctx *c, int part, float *mvm_ms, float *gather_ms)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev, "clm4_sharded_timing: bad argument");
    DeviceGuard guard;
    CLV_HIP(hipSetDevice(c->dev[part]));
    if (mvm_ms) CLV_HIP(hipEventElapsedTime(mvm_ms, c->ev[3 * part + 0], c->ev[3 * part + 1]));
    if (gather_ms) CLV_HIP(hipEventElapsedTime(gather_ms, c->ev[3 * part + 1], c->ev[3 * part + 2]));
    return CLV_OK;
}

// how the shards exchange: ranks in the RCCL communicator (0: none -- one shard, or the same-device test layout),
// and whether the gather is the single all-gather (equal shards) or the per-owner broadcasts (ragged)
extern "C" int clm4_sharded_comm_info(const clm4_shard_ctx *c, int *rccl_ranks, int *equal_shards)
{
    CLV_REQUIRE(c, "clm4_sharded_comm_info: null argument");
    if (rccl_ranks) *rccl_ranks = c->rccl_ranks;
    if (equal_shards) *equal_shards = c->equal ? 1 : 0;
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


// ---- timed-loop form: the same sharded mvm without any host synchronisation -----------------------------------------------------
// clm4_sharded_mvm is the blocking convenience call (it ends with a stream synchronise per device).  A loop that multiplies many
// times -- bench.py --mode one-process, an IHT iteration over a sharded matrix -- uses these instead:
//     clm4_sharded_set_x        x to every device (stream-ordered);
//     clm4_sharded_loop_begin   second result buffer, exchange stream and `slots` event triples per device;
//     clm4_sharded_mvm_enqueue  step i: kernel into result buffer i & 1 on the device's compute stream, the all-gather of that buffer
//                               on its exchange stream behind it -- so the gather of step i overlaps the kernel of step i + 1, and
//                               the kernel of step i + 2 waits (on the device) until the gather of step i has left its buffer;
//     clm4_sharded_sync         the one host wait, when the caller wants the results;
//     clm4_sharded_step_timing  kernel and exchange time of a timed step, from events on the device's own streams.
extern "C" int clm4_sharded_set_x(clm4_shard_ctx *c, const int8_t *x, const float *sx, int x_on_host)
{
    CLV_REQUIRE(c && x && sx, "clm4_sharded_set_x: null argument");
    DeviceGuard guard;
    CLV_HIP(hipSetDevice(c->dev[0]));
    const hipMemcpyKind kind = x_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    CLV_HIP(hipMemcpyAsync(c->x[0], x, c->cols / 2, kind, c->st[0]));
    CLV_HIP(hipMemcpyAsync(c->sx[0], sx, c->cols / 16, kind, c->st[0]));
    int rc = replicate_from_0(c, (void *const *)c->x.data(), c->cols / 2);
    if (rc == CLV_OK) rc = replicate_from_0(c, (void *const *)c->sx.data(), c->cols / 16);
    return rc;
}

extern "C" int clm4_sharded_loop_begin(clm4_shard_ctx *c, int slots)
{
    CLV_REQUIRE(c && slots >= 0 && slots <= (1 << 20), "clm4_sharded_loop_begin: bad argument");
    DeviceGuard guard;
    const int n = c->ndev;
    if (!c->loop_ready) {
        // built into locals and committed only when every device succeeded: a failure half way leaves the context as it was (no
        // half-initialised streams / buffers for a later enqueue to trip over), and what was created is released here
        std::vector<hipStream_t> cs(n, nullptr);
        std::vector<int8_t *> r2(n, nullptr);
        std::vector<float *> sr2(n, nullptr);
        std::vector<hipEvent_t> kdone(2 * (size_t)n, nullptr), gdone(2 * (size_t)n, nullptr);
        hipError_t err = hipSuccess;
        for (int d = 0; d < n && err == hipSuccess; d++) {
            if ((err = hipSetDevice(c->dev[d])) != hipSuccess) break;
            if ((err = hipStreamCreateWithFlags(&cs[d], hipStreamNonBlocking)) != hipSuccess) break;
            if ((err = hipMalloc((void **)&r2[d], c->rows / 2)) != hipSuccess) break;
            if ((err = hipMalloc((void **)&sr2[d], c->rows / 16)) != hipSuccess) break;
            for (int e = 0; e < 2 && err == hipSuccess; e++) {
                err = hipEventCreateWithFlags(&kdone[2 * d + e], hipEventDisableTiming);
                if (err == hipSuccess) err = hipEventCreateWithFlags(&gdone[2 * d + e], hipEventDisableTiming);
            }
        }
        if (err != hipSuccess) {
            for (int d = 0; d < n; d++) {
                (void)hipSetDevice(c->dev[d]);
                if (cs[d]) (void)hipStreamDestroy(cs[d]);
                if (r2[d]) (void)hipFree(r2[d]);
                if (sr2[d]) (void)hipFree(sr2[d]);
                for (int e = 0; e < 2; e++) {
                    if (kdone[2 * d + e]) (void)hipEventDestroy(kdone[2 * d + e]);
                    if (gdone[2 * d + e]) (void)hipEventDestroy(gdone[2 * d + e]);
                }
            }
            clv_set_error("clm4_sharded_loop_begin: %s", hipGetErrorString(err));
            return CLV_ERR_HIP;
        }
        c->cs.swap(cs); c->r2.swap(r2); c->sr2.swap(sr2); c->kdone.swap(kdone); c->gdone.swap(gdone);
        c->gpending.assign(2 * (size_t)n, 0);
        c->loop_ready = true;
    }
    if (slots > c->slots) {
        // events are appended one at a time and c->slots only moves over complete steps: a failure keeps what exists (destroyed with the
        // context) and a retry continues behind it instead of re-creating over live events
        const size_t want = 3 * (size_t)slots * n;
        c->slot_ev.reserve(want);
        while (c->slot_ev.size() < want) {
            hipEvent_t e = nullptr;
            CLV_HIP(hipSetDevice(c->dev[c->slot_ev.size() % n]));
            CLV_HIP(hipEventCreate(&e));
            c->slot_ev.push_back(e);
            c->slots = (int)(c->slot_ev.size() / (3 * (size_t)n));
        }
    }
    return CLV_OK;
}

extern "C" int clm4_sharded_mvm_enqueue(clm4_shard_ctx *c, int step, int timed)
{
    CLV_REQUIRE(c && step >= 0, "clm4_sharded_mvm_enqueue: bad argument");
    CLV_REQUIRE(c->loop_ready, "clm4_sharded_mvm_enqueue: call clm4_sharded_loop_begin first");
    CLV_REQUIRE(!timed || step < c->slots, "clm4_sharded_mvm_enqueue: step %d has no event slot (%d reserved)", step, c->slots);
    DeviceGuard guard;
    const int n = c->ndev, b = step & 1;
    const bool exchange = n > 1 || c->use_rccl;
    auto slot = [&](int e, int d) { return c->slot_ev[(3 * (size_t)step + e) * n + d]; };
    auto rb = [&](int d) { return b ? c->r2[d] : c->r[d]; };
    auto sb = [&](int d) { return b ? c->sr2[d] : c->sr[d]; };
    for (int d = 0; d < n; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        // the gather of step - 2 has left this buffer.  RCCL: the collective on cs[d] is the only reader AND writer of device d's buffer,
        // so its own gdone is enough.  Same-device test layout: the exchange is pull-based -- every OTHER shard's exchange stream copies
        // this shard's slice out of THIS buffer -- so the kernel waits for every shard's gather of step - 2, not only its own
        if (c->loopback && n > 1) {
            for (int o = 0; o < n; o++)
                if (c->gpending[2 * o + b]) CLV_HIP(hipStreamWaitEvent(c->st[d], c->gdone[2 * o + b], 0));
        } else if (c->gpending[2 * d + b]) {
            CLV_HIP(hipStreamWaitEvent(c->st[d], c->gdone[2 * d + b], 0));
        }
        if (timed) CLV_HIP(hipEventRecord(slot(0, d), c->st[d]));
        int rc = clm4_mvm(c->A[d], c->sA[d], c->row_count[d], c->cols, c->x[d], c->sx[d], rb(d) + c->row_begin[d] / 2,
                          sb(d) + c->row_begin[d] / 64, nullptr, c->st[d]);
        if (rc != CLV_OK) return rc;
        if (timed) CLV_HIP(hipEventRecord(slot(1, d), c->st[d]));
        if (exchange) CLV_HIP(hipEventRecord(c->kdone[2 * d + b], c->st[d]));
    }
    if (!exchange) {
        // one shard and no communicator: nothing leaves the device, the exchange stream stays out of it (every event record and
        // cross-stream wait is a barrier packet in the queue: ~5 us each between back-to-back launches)
        if (timed) CLV_HIP(hipEventRecord(slot(2, 0), c->st[0]));
        return CLV_OK;
    }
    if (n > 1 && c->loopback) {
        for (int d = 0; d < n; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            for (int o = 0; o < n; o++) {
                CLV_HIP(hipStreamWaitEvent(c->cs[d], c->kdone[2 * o + b], 0));
                if (o == d) continue;
                CLV_HIP(hipMemcpyAsync(rb(d) + c->row_begin[o] / 2, rb(o) + c->row_begin[o] / 2, c->row_count[o] / 2, hipMemcpyDeviceToDevice, c->cs[d]));
                CLV_HIP(hipMemcpyAsync(sb(d) + c->row_begin[o] / 64, sb(o) + c->row_begin[o] / 64, c->row_count[o] / 16, hipMemcpyDeviceToDevice, c->cs[d]));
            }
        }
    } else if (c->use_rccl) {
        for (int d = 0; d < n; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            CLV_HIP(hipStreamWaitEvent(c->cs[d], c->kdone[2 * d + b], 0));
        }
        RcclGroup g;
        CLV_NCCL(g.start());
        if (c->equal) {
            const uint64_t rc_rows = c->row_count[0];
            for (int d = 0; d < n; d++) {
                CLV_NCCL(rccl()->AllGather(rb(d) + c->row_begin[d] / 2, rb(d), rc_rows / 2, ncclInt8, c->comm[d], c->cs[d]));
                CLV_NCCL(rccl()->AllGather(sb(d) + c->row_begin[d] / 64, sb(d), rc_rows / 64, ncclFloat32, c->comm[d], c->cs[d]));
            }
        } else {
            for (int root = 0; root < n; root++)
                for (int d = 0; d < n; d++) {
                    int8_t *pr = rb(d) + c->row_begin[root] / 2;
                    float *ps = sb(d) + c->row_begin[root] / 64;
                    CLV_NCCL(rccl()->Broadcast(pr, pr, c->row_count[root] / 2, ncclInt8, root, c->comm[d], c->cs[d]));
                    CLV_NCCL(rccl()->Broadcast(ps, ps, c->row_count[root] / 64, ncclFloat32, root, c->comm[d], c->cs[d]));
                }
        }
        CLV_NCCL(g.end());
    }
    for (int d = 0; d < n; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        if (timed) CLV_HIP(hipEventRecord(slot(2, d), c->cs[d]));
        CLV_HIP(hipEventRecord(c->gdone[2 * d + b], c->cs[d]));
        c->gpending[2 * d + b] = 1;
    }
    return CLV_OK;
}

extern "C" int clm4_sharded_sync(clm4_shard_ctx *c)
{
    CLV_REQUIRE(c, "clm4_sharded_sync: null argument");
    DeviceGuard guard;
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipStreamSynchronize(c->st[d]));
        if (d < (int)c->cs.size() && c->cs[d]) CLV_HIP(hipStreamSynchronize(c->cs[d]));
    }
    return CLV_OK;
}

extern "C" int clm4_sharded_step_timing(const clm4_shard_ctx *c, int part, int step, float *mvm_ms, float *gather_ms)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev && step >= 0 && step < c->slots, "clm4_sharded_step_timing: bad argument");
    DeviceGuard guard;
    CLV_HIP(hipSetDevice(c->dev[part]));
    const size_t n = c->ndev;
    if (mvm_ms) CLV_HIP(hipEventElapsedTime(mvm_ms, c->slot_ev[(3 * (size_t)step + 0) * n + part], c->slot_ev[(3 * (size_t)step + 1) * n + part]));
    if (gather_ms) CLV_HIP(hipEventElapsedTime(gather_ms, c->slot_ev[(3 * (size_t)step + 1) * n + part], c->slot_ev[(3 * (size_t)step + 2) * n + part]));
    return CLV_OK;
}

// device pointers of the full result in buffer `buf` (= step & 1 of the enqueue that wrote it) held by shard `part`
extern "C" int clm4_sharded_result_buf(const clm4_shard_ctx *c, int part, int buf, const int8_t **r_dev, const float **sr_dev)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev && (buf == 0 || (buf == 1 && !c->r2.empty())), "clm4_sharded_result_buf: bad argument");
    if (r_dev) *r_dev = buf ? c->r2[part] : c->r[part];
    if (sr_dev) *sr_dev = buf ? c->sr2[part] : c->sr[part];
    return CLV_OK;
}

// B (N x cols nibbles + tile scales) on every device, (re)allocated when N changes; shared by the blocking and the loop form
static int gemm_ensure_B(clm4_shard_ctx *c, uint64_t N)
{
    if (c->gemm_b_n == N) return CLV_OK;
    const uint64_t b_bytes = N * c->cols / 2, sb_count = (N / 64) * (c->cols / 64);
    c->B.resize(c->ndev, nullptr); c->sB.resize(c->ndev, nullptr);
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipStreamSynchronize(c->st[d]));
        if (c->B[d]) CLV_HIP(hipFree(c->B[d]));
        if (c->sB[d]) CLV_HIP(hipFree(c->sB[d]));
        c->B[d] = nullptr; c->sB[d] = nullptr;
        CLV_HIP(hipMalloc((void **)&c->B[d], b_bytes));
        CLV_HIP(hipMalloc((void **)&c->sB[d], sb_count * sizeof(float)));
    }
    c->gemm_b_n = N;
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
    DeviceGuard guard;
    const uint64_t K = c->cols, b_bytes = N * K / 2, sb_count = (N / 64) * (K / 64);
    {
        int brc = gemm_ensure_B(c, N);
        if (brc != CLV_OK) return brc;
    }
    if (c->gemm_n != N) {                                      // (re)allocate the C shards for this N
        c->C.resize(c->ndev, nullptr);
        for (int d = 0; d < c->ndev; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            CLV_HIP(hipStreamSynchronize(c->st[d]));
            if (c->C[d]) CLV_HIP(hipFree(c->C[d]));
            c->C[d] = nullptr;
            CLV_HIP(hipMalloc((void **)&c->C[d], c->row_count[d] * N * sizeof(float)));
        }
        c->gemm_n = N;
    }
    // 1. B to device 0, then to everyone
    CLV_HIP(hipSetDevice(c->dev[0]));
    const hipMemcpyKind kind = b_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    CLV_HIP(hipMemcpyAsync(c->B[0], B, b_bytes, kind, c->st[0]));
    CLV_HIP(hipMemcpyAsync(c->sB[0], sB, sb_count * sizeof(float), kind, c->st[0]));
    int rrc = replicate_from_0(c, (void *const *)c->B.data(), b_bytes);
    if (rrc == CLV_OK) rrc = replicate_from_0(c, (void *const *)c->sB.data(), sb_count * sizeof(float));
    if (rrc != CLV_OK) return rrc;
    // 2. every device multiplies its row shard
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        int rc = clm4_gemm(c->A[d], c->sA[d], c->row_count[d], K, c->B[d], c->sB[d], N, c->C[d], c->st[d]);
        if (rc != CLV_OK) return rc;
    }
    for (int d = 0; d < c->ndev; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        CLV_HIP(hipStreamSynchronize(c->st[d]));
        if (C_host) CLV_HIP(hipMemcpy(C_host + c->row_begin[d] * N, c->C[d], c->row_count[d] * N * sizeof(float), hipMemcpyDeviceToHost));
    }
    return CLV_OK;
}

// device pointer of shard `part`'s rows of C (valid after clm4_sharded_gemm)
extern "C" int clm4_sharded_gemm_result(const clm4_shard_ctx *c, int part, const float **C_dev)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev && C_dev && c->gemm_n, "clm4_sharded_gemm_result: bad argument");
    *C_dev = c->C[part];
    return CLV_OK;
}

// ---- GEMM, loop form with the C row panels all-gathered (round 5) -----------------------------------------------------------------
// "GEMM: shard rows of A, replicate B, all-gather C row panels" -- the row split of mvm_parallel (CloverMatrix4.h:1700-1705)
// applied to the M rows of C = A * B^T.  Every element of C is its own fma chain over the K-blocks, so a panel computed by
// device d equals those rows of the unsharded clm4_gemm bit for bit; the exchange moves fp32 panels (rows_d x N x 4 bytes: 32 MiB per
// device for configs[3] split 8 ways), ONE in-place ncclAllGather per device and step on the exchange stream.  Two full C buffers per
// device: the gather of step i runs beside the kernel of step i + 1, and the kernel of step i + 2 waits on the device until the
// gather of step i has left its buffer.  Ragged shards: one broadcast per owner; shards that repeat a device (test layout): copies.
//     clm4_sharded_gemm_begin    B replicated (stream-ordered), both C buffers, exchange streams and `slots` event triples;
//     clm4_sharded_gemm_enqueue  step i: clm4_gemm of the shard into its panel of C buffer i & 1, then the all-gather of that buffer;
//     clm4_sharded_sync          the one host wait;
//     clm4_sharded_step_timing   kernel (re-code + MFMA kernel) and exchange time of a timed step;
//     clm4_sharded_gemm_full     device pointer of the whole C (rows x N fp32) in buffer `buf` on shard `part`.
// What happens to the C row panels after a step's kernels (clm4_sharded_gemm_begin_mode):
//   CLM4_GEMM_ALL_GATHER   every device ends with the whole C (SURVEY 8(e) as written).  (N - 1) / N of C enters EVERY device: at configs[3]
//                          split 8 ways 224 MiB per device and step against 0.06 ms of kernel -- exchange-bound by construction;
//   CLM4_GEMM_GATHER_ROOT  the panels go to shard 0's device only (grouped ncclSend / ncclRecv): the consumer of C sits on one device; 7 of 8
//                          devices send 32 MiB and receive nothing;
//   CLM4_GEMM_SHARDED      nothing is exchanged: C stays row-sharded like A (the natural form when the consumer is sharded the same way, e.g.
//                          the next product's left operand); the step is the kernel.
// The panel of shard d is at rows row_begin[d] of its device's C buffer in every mode; bits never depend on the mode.
extern "C" int clm4_sharded_gemm_begin_mode(clm4_shard_ctx *c, const int8_t *B, const float *sB, uint64_t N, int b_on_host, int slots, int mode)
{
    CLV_REQUIRE(c, "clm4_sharded_gemm_begin_mode: bad argument");
    CLV_REQUIRE(mode == CLM4_GEMM_ALL_GATHER || mode == CLM4_GEMM_GATHER_ROOT || mode == CLM4_GEMM_SHARDED, "clm4_sharded_gemm_begin_mode: unknown mode %d", mode);
    CLV_REQUIRE(mode != CLM4_GEMM_GATHER_ROOT || !c->use_rccl || (rccl()->Send && rccl()->Recv),
                "clm4_sharded_gemm_begin_mode: this librccl has no ncclSend / ncclRecv");
    int rc = clm4_sharded_gemm_begin(c, B, sB, N, b_on_host, slots);
    if (rc == CLV_OK) c->gemm_mode = mode;
    return rc;
}

extern "C" int clm4_sharded_gemm_begin(clm4_shard_ctx *c, const int8_t *B, const float *sB, uint64_t N, int b_on_host, int slots)
{
    CLV_REQUIRE(c && B && sB && N && N % 128 == 0, "clm4_sharded_gemm_begin: bad argument");
    c->gemm_mode = CLM4_GEMM_ALL_GATHER;
    for (int d = 0; d < c->ndev; d++)
        CLV_REQUIRE(c->row_count[d] % 128 == 0, "clm4_sharded_gemm_begin: shard %d has %llu rows, not a multiple of 128", d, (unsigned long long)c->row_count[d]);
    int rc = clm4_sharded_loop_begin(c, slots);
    if (rc != CLV_OK) return rc;
    DeviceGuard guard;
    const int n = c->ndev;
    const uint64_t K = c->cols, b_bytes = N * K / 2, sb_count = (N / 64) * (K / 64);
    rc = gemm_ensure_B(c, N);
    if (rc != CLV_OK) return rc;
    if (c->gemm_loop_n != N) {
        c->Cf.resize(2 * (size_t)n, nullptr);
        for (int d = 0; d < n; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            CLV_HIP(hipStreamSynchronize(c->st[d]));
            CLV_HIP(hipStreamSynchronize(c->cs[d]));
            for (int e = 0; e < 2; e++) {
                if (c->Cf[2 * d + e]) CLV_HIP(hipFree(c->Cf[2 * d + e]));
                c->Cf[2 * d + e] = nullptr;
                CLV_HIP(hipMalloc((void **)&c->Cf[2 * d + e], c->rows * N * sizeof(float)));
            }
        }
        c->gemm_loop_n = N;
    }
    CLV_HIP(hipSetDevice(c->dev[0]));
    const hipMemcpyKind kind = b_on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    CLV_HIP(hipMemcpyAsync(c->B[0], B, b_bytes, kind, c->st[0]));
    CLV_HIP(hipMemcpyAsync(c->sB[0], sB, sb_count * sizeof(float), kind, c->st[0]));
    rc = replicate_from_0(c, (void *const *)c->B.data(), b_bytes);
    if (rc == CLV_OK) rc = replicate_from_0(c, (void *const *)c->sB.data(), sb_count * sizeof(float));
    return rc;
}

extern "C" int clm4_sharded_gemm_enqueue(clm4_shard_ctx *c, int step, int timed)
{
    CLV_REQUIRE(c && step >= 0, "clm4_sharded_gemm_enqueue: bad argument");
    CLV_REQUIRE(c->loop_ready && c->gemm_loop_n, "clm4_sharded_gemm_enqueue: call clm4_sharded_gemm_begin first");
    CLV_REQUIRE(!timed || step < c->slots, "clm4_sharded_gemm_enqueue: step %d has no event slot (%d reserved)", step, c->slots);
    DeviceGuard guard;
    const int n = c->ndev, b = step & 1, mode = c->gemm_mode;
    const bool exchange = (n > 1 || c->use_rccl) && mode != CLM4_GEMM_SHARDED;
    const bool to_root = mode == CLM4_GEMM_GATHER_ROOT;
    const uint64_t N = c->gemm_loop_n, K = c->cols;
    auto slot = [&](int e, int d) { return c->slot_ev[(3 * (size_t)step + e) * n + d]; };
    auto Cb = [&](int d) { return c->Cf[2 * d + b]; };
    for (int d = 0; d < n; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        // the exchange of step - 2 has left this buffer (see clm4_sharded_mvm_enqueue: the same-device test layout is pull-based)
        if (c->loopback && n > 1) {
            for (int o = 0; o < n; o++)
                if (c->gpending[2 * o + b]) CLV_HIP(hipStreamWaitEvent(c->st[d], c->gdone[2 * o + b], 0));
        } else if (c->gpending[2 * d + b]) {
            CLV_HIP(hipStreamWaitEvent(c->st[d], c->gdone[2 * d + b], 0));
        }
        if (timed) CLV_HIP(hipEventRecord(slot(0, d), c->st[d]));
        int rc = clm4_gemm(c->A[d], c->sA[d], c->row_count[d], K, c->B[d], c->sB[d], N, Cb(d) + c->row_begin[d] * N, c->st[d]);
        if (rc != CLV_OK) return rc;
        if (timed) CLV_HIP(hipEventRecord(slot(1, d), c->st[d]));
        if (exchange) CLV_HIP(hipEventRecord(c->kdone[2 * d + b], c->st[d]));
    }
    if (!exchange) {                                            // one shard and no communicator (see clm4_sharded_mvm_enqueue), or C stays sharded
        if (timed)
            for (int d = 0; d < (mode == CLM4_GEMM_SHARDED ? n : 1); d++) {
                CLV_HIP(hipSetDevice(c->dev[d]));
                CLV_HIP(hipEventRecord(slot(2, d), c->st[d]));
            }
        return CLV_OK;
    }
    if (n > 1 && c->loopback) {
        for (int d = 0; d < (to_root ? 1 : n); d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            for (int o = 0; o < n; o++) {
                CLV_HIP(hipStreamWaitEvent(c->cs[d], c->kdone[2 * o + b], 0));
                if (o == d) continue;
                CLV_HIP(hipMemcpyAsync(Cb(d) + c->row_begin[o] * N, Cb(o) + c->row_begin[o] * N, c->row_count[o] * N * sizeof(float),
                                       hipMemcpyDeviceToDevice, c->cs[d]));
            }
        }
    } else {
        for (int d = 0; d < n; d++) {
            CLV_HIP(hipSetDevice(c->dev[d]));
            CLV_HIP(hipStreamWaitEvent(c->cs[d], c->kdone[2 * d + b], 0));
        }
        if (c->use_rccl && to_root) {
            RcclGroup g;
            CLV_NCCL(g.start());
            for (int d = 1; d < n; d++) {                         // shard d's panel: device d -> device 0, same offset in both C buffers
                CLV_NCCL(rccl()->Send(Cb(d) + c->row_begin[d] * N, c->row_count[d] * N, ncclFloat32, 0, c->comm[d], c->cs[d]));
                CLV_NCCL(rccl()->Recv(Cb(0) + c->row_begin[d] * N, c->row_count[d] * N, ncclFloat32, d, c->comm[0], c->cs[0]));
            }
            CLV_NCCL(g.end());
        } else if (c->use_rccl) {
            RcclGroup g;
            CLV_NCCL(g.start());
            if (c->equal) {
                // in place: rank d's panel already sits at offset d * count of its receive buffer
                for (int d = 0; d < n; d++)
                    CLV_NCCL(rccl()->AllGather(Cb(d) + c->row_begin[d] * N, Cb(d), c->row_count[0] * N, ncclFloat32, c->comm[d], c->cs[d]));
            } else {
                for (int root = 0; root < n; root++)
                    for (int d = 0; d < n; d++) {
                        float *p = Cb(d) + c->row_begin[root] * N;
                        CLV_NCCL(rccl()->Broadcast(p, p, c->row_count[root] * N, ncclFloat32, root, c->comm[d], c->cs[d]));
                    }
            }
            CLV_NCCL(g.end());
        }
    }
    for (int d = 0; d < n; d++) {
        CLV_HIP(hipSetDevice(c->dev[d]));
        if (timed) CLV_HIP(hipEventRecord(slot(2, d), c->cs[d]));
        CLV_HIP(hipEventRecord(c->gdone[2 * d + b], c->cs[d]));
        c->gpending[2 * d + b] = 1;
    }
    return CLV_OK;
}

// device pointer of the whole C (rows x N fp32, row-major) in buffer `buf` (= step & 1 of the enqueue that wrote it) on shard `part`
extern "C" int clm4_sharded_gemm_full(const clm4_shard_ctx *c, int part, int buf, const float **C_dev)
{
    CLV_REQUIRE(c && part >= 0 && part < c->ndev && (buf == 0 || buf == 1) && C_dev && c->gemm_loop_n, "clm4_sharded_gemm_full: bad argument");
    *C_dev = c->Cf[2 * part + buf];
    return CLV_OK;
}
