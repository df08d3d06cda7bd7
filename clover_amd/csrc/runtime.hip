// runtime.hip -- device/stream/memory plumbing, XORShift key handling and synthetic-data fills.
#include "common.h"
#include "rng_device.h"

#include <mutex>
#include <unordered_set>

#include <atomic>

#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

// ---- errors -----------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void clv_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char *clv_last_error(void) { return g_err; }
// the bench-only probe build (clover_amd/build.py build_probe_library) defines this symbol; the product library does not, the weak
// reference is then null.  A probe build announces itself, and clover_amd.lib_binding.load_library refuses it unless asked for.
extern "C" __attribute__((weak)) const char *clvx_probe_tag(void);
extern "C" const char *clv_version(void) { return clvx_probe_tag ? clvx_probe_tag() : "clover_hip 0.1 (gfx950)"; }

// ---- devices ----------------------------------------------------------------------------------
extern "C" int clv_device_count(int *count)
{
    CLV_REQUIRE(count, "clv_device_count: null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { n = 0; (void)hipGetLastError(); }
    *count = n;
    return CLV_OK;
}

extern "C" int clv_set_device(int device) { CLV_HIP(hipSetDevice(device)); return CLV_OK; }
extern "C" int clv_get_device(int *device)
{
    CLV_REQUIRE(device, "clv_get_device: null argument");
    CLV_HIP(hipGetDevice(device));
    return CLV_OK;
}

extern "C" int clv_device_info(char *name, int name_len, int *compute_units, uint64_t *hbm_bytes)
{
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    CLV_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return CLV_OK;
}

#define CLV_MAX_DEVICES 64
static int g_cu[CLV_MAX_DEVICES];

int clv_cu_count()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CLV_MAX_DEVICES) return 256;
    if (g_cu[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cu[dev] = n;
    }
    return g_cu[dev];
}

// Grow-only scratch, one buffer per (device, stream): used when a caller passes workspace == NULL (clv4_dot, threshold) and by
// clm4_gemm for the FP6 operand images.  Calls on DIFFERENT streams never share a buffer, so they may overlap; calls on one
// stream are ordered by the stream.  Growing a buffer waits for that stream only (its old buffer may still be in use there).
struct WsEntry {
    int dev;
    hipStream_t stream;
    void *ptr;
    uint64_t bytes;
};
static std::mutex g_ws_mutex;
static std::vector<WsEntry> g_ws;

int clv_internal_workspace(void **ptr, uint64_t bytes, hipStream_t stream)
{
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    WsEntry *e = nullptr;
    for (auto &w : g_ws)
        if (w.dev == dev && w.stream == stream) { e = &w; break; }
    if (!e) {
        g_ws.push_back(WsEntry{dev, stream, nullptr, 0});
        e = &g_ws.back();
    }
    if (e->bytes < bytes) {
        if (e->ptr) {
            CLV_HIP(hipStreamSynchronize(stream));
            CLV_HIP(hipFree(e->ptr));
            e->ptr = nullptr;
            e->bytes = 0;
        }
        uint64_t want = bytes < (1ull << 20) ? (1ull << 20) : bytes;
        CLV_HIP(hipMalloc(&e->ptr, want));
        e->bytes = want;
    }
    *ptr = e->ptr;
    return CLV_OK;
}

// Hand-over slots, one small buffer per (device, stream), ZERO when handed out for the first time and zero again after every kernel
// that used them (the consumer clears what it read): single-launch reductions (clv4_dot FAST) pass their workgroup partials through
// them without a second kernel.  Separate from the scratch above, which other calls on the stream overwrite.
#define CLV_SYNC_SLOT_BYTES CLV_SYNC_SLOT_BYTES_TOTAL
static std::vector<WsEntry> g_slots;

int clv_internal_sync_slots(void **ptr, uint64_t bytes, hipStream_t stream)
{
    CLV_REQUIRE(bytes <= CLV_SYNC_SLOT_BYTES, "clv_internal_sync_slots: %llu bytes wanted, %u kept per stream", (unsigned long long)bytes, CLV_SYNC_SLOT_BYTES);
    int dev = 0;
    CLV_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (auto &w : g_slots)
        if (w.dev == dev && w.stream == stream) { *ptr = w.ptr; return CLV_OK; }
    // first use on this stream: an allocation, which a stream capture must not contain (hipMalloc is illegal under a global-mode capture and
    // the memset would become a graph node that re-zeroes the slots under a running collector on replay)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusNone) {
        clv_set_error("this stream is being captured and has not run a single-launch reduction yet: make one ordinary clv4_dot / clv8_dot (FAST) call "
                      "on it before hipStreamBeginCapture");
        return CLV_ERR_INVALID;
    }
    void *p = nullptr;
    CLV_HIP(hipMalloc(&p, CLV_SYNC_SLOT_BYTES));
    if (hipMemsetAsync(p, 0, CLV_SYNC_SLOT_BYTES, stream) != hipSuccess) {       // stream-ordered in front of the first kernel that uses them
        (void)hipFree(p);
        clv_set_error("clv_internal_sync_slots: hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
        return CLV_ERR_HIP;
    }
    g_slots.push_back(WsEntry{dev, stream, p, CLV_SYNC_SLOT_BYTES});
    *ptr = p;
    return CLV_OK;
}

// a destroyed stream's handle may be re-used by the runtime for a new stream: drop its scratch with it
void clv_internal_workspace_forget(hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (size_t k = 0; k < g_slots.size();) {
        if (g_slots[k].stream == stream && stream != nullptr) {
            if (g_slots[k].ptr) (void)hipFree(g_slots[k].ptr);
            g_slots.erase(g_slots.begin() + (long)k);
        } else {
            k++;
        }
    }
    for (size_t k = 0; k < g_ws.size();) {
        if (g_ws[k].stream == stream && stream != nullptr) {
            if (g_ws[k].ptr) (void)hipFree(g_ws[k].ptr);
            g_ws.erase(g_ws.begin() + (long)k);
        } else {
            k++;
        }
    }
}

// ---- memory / streams / events ------------------------------------------------------------------
extern "C" int clv_malloc(void **ptr, uint64_t bytes)
{
    CLV_REQUIRE(ptr, "clv_malloc: null argument");
    CLV_HIP(hipMalloc(ptr, bytes ? bytes : 1));
    return CLV_OK;
}
static void rng_graph_forget(const void *ptr);      // below: a freed buffer must not leave its address in graph mode
extern "C" int clv_free(void *ptr) { if (ptr) { rng_graph_forget(ptr); CLV_HIP(hipFree(ptr)); } return CLV_OK; }
extern "C" int clv_memset(void *ptr, int value, uint64_t bytes, void *stream)
{
    CLV_HIP(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return CLV_OK;
}
extern "C" int clv_memcpy_h2d(void *dst, const void *src, uint64_t bytes, void *stream)
{
    CLV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return CLV_OK;
}
extern "C" int clv_memcpy_d2h(void *dst, const void *src, uint64_t bytes, void *stream)
{
    CLV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    CLV_HIP(hipStreamSynchronize(as_stream(stream)));
    return CLV_OK;
}
extern "C" int clv_memcpy_d2d(void *dst, const void *src, uint64_t bytes, void *stream)
{
    CLV_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return CLV_OK;
}
extern "C" int clv_host_alloc(void **ptr, uint64_t bytes)
{
    CLV_REQUIRE(ptr, "clv_host_alloc: null argument");
    CLV_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return CLV_OK;
}
extern "C" int clv_host_free(void *ptr) { if (ptr) CLV_HIP(hipHostFree(ptr)); return CLV_OK; }

extern "C" int clv_stream_create(void **stream)
{
    CLV_REQUIRE(stream, "clv_stream_create: null argument");
    hipStream_t s;
    CLV_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return CLV_OK;
}
extern "C" int clv_stream_destroy(void *stream)
{
    clv_internal_workspace_forget(as_stream(stream));
    clv_internal_persist_forget(as_stream(stream));
    CLV_HIP(hipStreamDestroy(as_stream(stream)));
    return CLV_OK;
}
extern "C" int clv_stream_sync(void *stream) { CLV_HIP(hipStreamSynchronize(as_stream(stream))); return CLV_OK; }
extern "C" int clv_device_sync(void) { CLV_HIP(hipDeviceSynchronize()); return CLV_OK; }

extern "C" int clv_event_create(void **event)
{
    CLV_REQUIRE(event, "clv_event_create: null argument");
    hipEvent_t e;
    CLV_HIP(hipEventCreate(&e));
    *event = e;
    return CLV_OK;
}
extern "C" int clv_event_destroy(void *event) { CLV_HIP(hipEventDestroy((hipEvent_t)event)); return CLV_OK; }
extern "C" int clv_event_record(void *event, void *stream)
{
    CLV_HIP(hipEventRecord((hipEvent_t)event, as_stream(stream)));
    return CLV_OK;
}
extern "C" int clv_event_sync(void *event) { CLV_HIP(hipEventSynchronize((hipEvent_t)event)); return CLV_OK; }
extern "C" int clv_event_elapsed_ms(void *start, void *stop, float *ms)
{
    CLV_REQUIRE(ms, "clv_event_elapsed_ms: null argument");
    CLV_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return CLV_OK;
}

// ---- XORShift keys ------------------------------------------------------------------------------
// Lane 0 is the seed pair; lanes 1..3 are successive 2^64-step jumps of the canonical xorshift128+
// (reference include/simdxorshift128plus.h:38-92).
static void canon_step(uint64_t &a, uint64_t &b)
{
    uint64_t s1 = a;
    const uint64_t s0 = b;
    a = s0;
    s1 ^= s1 << 23;
    b = s1 ^ s0 ^ (s1 >> 18) ^ (s0 >> 5);
}

static void canon_jump(uint64_t in0, uint64_t in1, uint64_t &out0, uint64_t &out1)
{
    static const uint64_t poly[2] = {0x8a5cd789635d2dffull, 0x121fd2155c472f96ull};
    uint64_t a = 0, b = 0;
    for (int i = 0; i < 2; i++)
        for (int bit = 0; bit < 64; bit++) {
            if (poly[i] & (1ull << bit)) { a ^= in0; b ^= in1; }
            canon_step(in0, in1);
        }
    out0 = a;
    out1 = b;
}

static std::atomic<uint64_t> g_rng_seq{1};
uint64_t clv_rng_next_seq() { return g_rng_seq.fetch_add(1, std::memory_order_relaxed); }
// host numbers must stay above every stamp a state carries: a state that leaves graph mode brings device-made stamps along
static void rng_seq_raise_above(uint64_t stamp)
{
    uint64_t cur = g_rng_seq.load(std::memory_order_relaxed);
    while (cur <= stamp && !g_rng_seq.compare_exchange_weak(cur, stamp + 1, std::memory_order_relaxed)) {}
}

// ---- graph mode: the launch sequence of a state kept on the device -----------------------------------------------------------------
// A stochastic kernel tells the state slot its predecessor wrote from the one it writes itself by a per-launch number.  Ordinarily the
// host hands that number over as a kernel argument -- which a captured hipGraph would replay.  For a state in graph mode the number
// lives in the state buffer (RNG_TICK_WORD): every stochastic call first enqueues k_rng_tick (one thread: counter += 1) and passes 0,
// the kernels read the counter (rng_effective_seq).  Price: one extra ~2 us launch per stochastic call -- hence opt-in.
static std::mutex g_graph_mutex;
static std::unordered_set<const uint64_t *> g_graph_states;
static std::atomic<int> g_graph_count{0};       // size of the set, readable without the mutex: the ordinary (non-graph) launch path takes no lock

// hipMalloc hands addresses out again: clv_free drops the address, so a later, unrelated state does not inherit graph mode
static void rng_graph_forget(const void *ptr)
{
    if (g_graph_count.load(std::memory_order_acquire) == 0) return;
    std::lock_guard<std::mutex> g(g_graph_mutex);
    if (g_graph_states.erase((const uint64_t *)ptr)) g_graph_count.fetch_sub(1, std::memory_order_release);
}

__global__ void k_rng_tick(uint64_t *state) { state[RNG_TICK_WORD] += 1; }
// the counter starts behind every stamp present, so the first tick is newer than both slots
__global__ void k_rng_tick_init(uint64_t *state)
{
    const uint64_t s0 = state[RNG_STAMP_WORD], s1 = state[RNG_SLOT_WORDS + RNG_STAMP_WORD];
    state[RNG_TICK_WORD] = s0 > s1 ? s0 : s1;
}

uint64_t clv_rng_seq_for(uint64_t *state, hipStream_t stream)
{
    if (g_graph_count.load(std::memory_order_acquire) == 0) return clv_rng_next_seq();
    bool graph;
    {
        std::lock_guard<std::mutex> g(g_graph_mutex);
        graph = g_graph_states.count(state) != 0;
    }
    if (!graph) return clv_rng_next_seq();
    hipLaunchKernelGGL(k_rng_tick, dim3(1), dim3(1), 0, stream, state);
    // a failed tick launch is left in hipGetLastError for the CLV_LAUNCH_CHECK that follows the stochastic kernel of this very call
    // (every caller launches and checks right behind this): the call then fails instead of drawing from a stale sequence number
    return 0;
}

extern "C" int clv_rng_graph_mode(uint64_t *state_dev, int on, void *stream)
{
    CLV_REQUIRE(state_dev, "clv_rng_graph_mode: null argument");
    std::lock_guard<std::mutex> g(g_graph_mutex);
    if (on) {
        hipLaunchKernelGGL(k_rng_tick_init, dim3(1), dim3(1), 0, as_stream(stream), state_dev);
        CLV_LAUNCH_CHECK();
        if (g_graph_states.insert(state_dev).second) g_graph_count.fetch_add(1, std::memory_order_release);
    } else if (g_graph_states.erase(state_dev)) {
        g_graph_count.fetch_sub(1, std::memory_order_release);
        uint64_t tick = 0;                           // the stamps this state carries were made on the device: host numbers continue above them
        CLV_HIP(hipMemcpyAsync(&tick, state_dev + RNG_TICK_WORD, sizeof tick, hipMemcpyDeviceToHost, as_stream(stream)));
        CLV_HIP(hipStreamSynchronize(as_stream(stream)));
        rng_seq_raise_above(tick);
    }
    return CLV_OK;
}

extern "C" int clv_rng_set(uint64_t *state_dev, const uint64_t key1[4], const uint64_t key2[4], void *stream)
{
    CLV_REQUIRE(state_dev && key1 && key2, "clv_rng_set: null argument");
    uint64_t st[CLV_RNG_STATE_BYTES / 8] = {0};      // slot 0 = keys + stamp, slot 1 empty with stamp 0 (rng_device.h)
    memcpy(st, key1, 32);
    memcpy(st + 4, key2, 32);
    st[8] = clv_rng_next_seq();
    st[RNG_TICK_WORD] = st[8];                       // graph mode: the device counter continues from this stamp
    CLV_HIP(hipMemcpyAsync(state_dev, st, sizeof st, hipMemcpyHostToDevice, as_stream(stream)));
    CLV_HIP(hipStreamSynchronize(as_stream(stream)));   // st is a stack buffer
    return CLV_OK;
}

extern "C" int clv_rng_seed(uint64_t *state_dev, uint64_t key1, uint64_t key2, void *stream)
{
    uint64_t s0[4], s1[4];
    s0[0] = key1;
    s1[0] = key2;
    for (int l = 1; l < 4; l++) canon_jump(s0[l - 1], s1[l - 1], s0[l], s1[l]);
    return clv_rng_set(state_dev, s0, s1, stream);
}

extern "C" int clv_rng_get(const uint64_t *state_dev, uint64_t key1[4], uint64_t key2[4], void *stream)
{
    CLV_REQUIRE(state_dev && key1 && key2, "clv_rng_get: null argument");
    uint64_t st[CLV_RNG_STATE_BYTES / 8];
    CLV_HIP(hipMemcpyAsync(st, state_dev, sizeof st, hipMemcpyDeviceToHost, as_stream(stream)));
    CLV_HIP(hipStreamSynchronize(as_stream(stream)));
    const uint64_t *cur = st[16 + 8] > st[8] ? st + 16 : st;      // the slot stamped last
    memcpy(key1, cur, 32);
    memcpy(key2, cur + 4, 32);
    return CLV_OK;
}

// ---- synthetic data -----------------------------------------------------------------------------
// one dword (8 nibbles) per splitmix64 draw; nibble = (byte * 15 >> 8) - 7, uniform on [-7,7]
__global__ void k_fill_nibbles(uint32_t *q, uint64_t nwords, uint64_t seed, uint64_t word_offset)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) {
        const uint64_t r = splitmix64(seed ^ ((i + word_offset) * 0xD6E8FEB86659FD93ull));
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int v = (int)((((r >> (8 * e)) & 0xFF) * 15) >> 8) - 7;
            w |= ((uint32_t)v & 0xFu) << (4 * e);
        }
        q[i] = w;
    }
}

__global__ void k_fill_scales(float *s, uint64_t n, uint64_t seed, uint64_t offset)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix64(seed ^ ((i + offset) * 0xA24BAED4963EE407ull));
        // [0.5, 2): 0.5 + 1.5 * u, u a 24-bit fraction
        s[i] = 0.5f + 1.5f * ((float)(r >> 40) * (1.0f / 16777216.0f));
    }
}

__global__ void k_fill_ints_f32(float *x, uint64_t n, int range, uint64_t seed, uint64_t offset)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t span = 2ull * (uint64_t)range + 1ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix64(seed ^ ((i + offset) * 0x9FB21C651E98DF25ull));
        x[i] = (float)((int)(((r >> 32) * span) >> 32) - range);
    }
}

static inline int fill_grid(uint64_t n)
{
    const uint64_t want = (n + 255) / 256;
    const uint64_t cap = (uint64_t)clv_cu_count() * 16;
    return (int)(want < cap ? (want ? want : 1) : cap);
}

extern "C" int clv_fill_random_nibbles(int8_t *q, uint64_t bytes, uint64_t seed, uint64_t byte_offset, void *stream)
{
    CLV_REQUIRE(q && (bytes % 4 == 0) && (byte_offset % 4 == 0), "clv_fill_random_nibbles: size/offset must be multiples of 4");
    if (!bytes) return CLV_OK;
    hipLaunchKernelGGL(k_fill_nibbles, dim3(fill_grid(bytes / 4)), dim3(256), 0, as_stream(stream),
                       (uint32_t *)q, bytes / 4, seed, byte_offset / 4);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv_fill_random_scales(float *s, uint64_t count, uint64_t seed, uint64_t index_offset, void *stream)
{
    CLV_REQUIRE(s, "clv_fill_random_scales: null argument");
    if (!count) return CLV_OK;
    hipLaunchKernelGGL(k_fill_scales, dim3(fill_grid(count)), dim3(256), 0, as_stream(stream), s, count, seed, index_offset);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}

extern "C" int clv_fill_random_ints_f32(float *x, uint64_t count, int range, uint64_t seed, uint64_t index_offset, void *stream)
{
    CLV_REQUIRE(x && range >= 0, "clv_fill_random_ints_f32: bad argument");
    if (!count) return CLV_OK;
    hipLaunchKernelGGL(k_fill_ints_f32, dim3(fill_grid(count)), dim3(256), 0, as_stream(stream), x, count, range, seed, index_offset);
    CLV_LAUNCH_CHECK();
    return CLV_OK;
}
