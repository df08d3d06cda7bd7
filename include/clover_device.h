/*
 * clover_device.h -- host/device mirror plumbing shared by the drop-in containers in this directory
 * (CloverVector32/4/8, CloverMatrix32/4).  Own code; the reference has no counterpart because it is CPU-only.
 *
 * Model: every container owns ONE page-aligned host block laid out exactly like the reference's
 * (CloverVector4.h:68-103: values immediately followed by scales, so user-visible pointer arithmetic on
 * getData()/getScales() keeps working) and a lazily created HBM mirror of the same bytes.
 *
 * The reference hands out raw pointers (getData()/getScales()) that stay valid for the life of the object and always
 * show the current contents, because there is only one copy.  To keep that contract with two copies, an owned host
 * block is tracked with page protection -- three states, switched with mprotect() on the whole block:
 *
 *     HOST_DIRTY    host current, writable         device stale      (PROT_READ|PROT_WRITE)
 *     SHARED        host and device both current                     (PROT_READ: the first host WRITE faults -> HOST_DIRTY)
 *     DEVICE_DIRTY  device current, host stale                       (PROT_NONE: the first host ACCESS faults -> the block is
 *                                                                     copied back -> SHARED)
 *
 * A SIGSEGV handler (installed once, chained in front of whatever was there) resolves faults that hit a tracked block, so
 * a pointer kept across a device operation reads the kernel's results and a write through it reaches the next kernel --
 * at the price of one fault (a few microseconds) per state change, none while an object stays on one side.  Accessor
 * methods (get(), set(), getBits() ...) switch state directly and never fault.  Limits, all loud rather than silent: a
 * system call given a pointer into a DEVICE_DIRTY block returns EFAULT instead of faulting (use the accessors or touch the
 * memory first); debuggers stop on the resolved SIGSEGVs (gdb: `handle SIGSEGV nostop noprint`).  Building with
 * -DCLOVER_HIP_NO_PAGE_TRACKING disables all of it: getData() then conservatively invalidates the device copy, and a pointer
 * kept across a device operation is stale (the round-1 behaviour).
 *
 * Non-owning views (CloverVector4(n, values, scales), CloverVector4.h:114-119) alias caller memory that cannot be
 * protected (not page-granular): they are written through -- every device result is copied back into the caller's memory
 * before the method returns -- and re-uploaded before every device read, so both directions are visible immediately as in
 * the reference.
 *
 * Errors follow the reference's convention (CloverVector4.h:100-101, CloverMatrix4.h:779-782): a message
 * on stdout and exit(1) -- the C ABI underneath returns status codes instead.
 */
#ifndef CLOVER_DEVICE_H
#define CLOVER_DEVICE_H

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>

#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include "clover_hip.h"

namespace clover_hip {

inline void check(int rc, const char *what)
{
    if (rc != CLV_OK) {
        std::cout << what << " failed: " << clv_last_error() << ". Exiting ..." << std::endl;
        exit(1);
    }
}

inline uint64_t round_up(uint64_t v, uint64_t m) { return v % m ? v + m - (v % m) : v; }

class Mirror;

namespace detail {

/* process-wide list of tracked blocks + the chained SIGSEGV handler.  Lives in an inline function so that every
 * translation unit including this header shares one instance (C++11). */
struct Tracker {
    std::atomic_flag lock;
    Mirror *head;
    struct sigaction previous;
    bool installed;
    Tracker() : head(nullptr), installed(false) { lock.clear(); }
    void acquire() { while (lock.test_and_set(std::memory_order_acquire)) {} }
    void release() { lock.clear(std::memory_order_release); }
};
inline Tracker &tracker()
{
    static Tracker t;
    return t;
}
inline void fault_handler(int sig, siginfo_t *info, void *uctx);
inline bool install_handler_once()
{
    Tracker &t = tracker();
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = fault_handler;
    sa.sa_flags = SA_SIGINFO | SA_NODEFER;      /* NODEFER: a genuine crash inside the handler is not masked into a hang */
    sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &t.previous);
    t.installed = true;
    return true;
}
inline void install_handler()
{
    static const bool once = install_handler_once();      /* C++11: initialised exactly once, also under concurrent first use */
    (void)once;
}

/* Read (and for writes: rewrite) one byte at both ends of a host range on THIS thread before handing the pointer to the
 * HIP runtime: if the range lies inside some container's protected block, the fault is resolved here and not inside a
 * runtime copy. */
inline void touch_for_read(const void *p, uint64_t bytes)
{
    if (!p || !bytes) return;
    const volatile uint8_t *b = static_cast<const volatile uint8_t *>(p);
    (void)b[0];
    (void)b[bytes - 1];
}
inline void touch_for_write(void *p, uint64_t bytes)
{
    if (!p || !bytes) return;
    volatile uint8_t *b = static_cast<volatile uint8_t *>(p);
    b[0] = b[0];
    b[bytes - 1] = b[bytes - 1];
}

}  // namespace detail

/* One host block + one device block of `bytes` bytes, kept coherent as described at the top of this file. */
class Mirror {
public:
    enum State { HOST_DIRTY, SHARED, DEVICE_DIRTY };

    Mirror() : host_(nullptr), dev_(nullptr), bytes_(0), span_(0), state_(HOST_DIRTY), owns_host_(true), pending_(false), version_(0), next_(nullptr), prev_(nullptr) {}
    ~Mirror() { release(); }
    Mirror(const Mirror &) = delete;
    Mirror &operator=(const Mirror &) = delete;

    void allocate(uint64_t bytes)
    {
        release();
        bytes_ = bytes;
        const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE);
        span_ = round_up(bytes ? bytes : 1, page);                 /* whole pages: protection must not touch a neighbour */
        void *p = nullptr;
        if (posix_memalign(&p, (size_t)page, span_) != 0) {
            std::cout << "Could not allocate host memory. Exiting ..." << std::endl;
            exit(1);
        }
        host_ = static_cast<uint8_t *>(p);
        owns_host_ = true;
        state_ = HOST_DIRTY;
#ifndef CLOVER_HIP_NO_PAGE_TRACKING
        detail::install_handler();
        link();
#endif
    }

    /* non-owning view over user memory (CloverVector4(n, values, scales), CloverVector4.h:114-119) */
    void adopt(void *host, uint64_t bytes)
    {
        release();
        host_ = static_cast<uint8_t *>(host);
        bytes_ = bytes;
        span_ = 0;
        owns_host_ = false;
        state_ = HOST_DIRTY;
    }

    uint64_t bytes() const { return bytes_; }
    State state() const { return state_; }
    /* counts every event after which the DEVICE copy may hold other bytes than before (an upload, a kernel that was handed a
     * writable pointer): what is derived from the device copy -- CloverMatrix4's cached GEMM operand image -- is valid for one
     * value of it.  Read it AFTER dev_ro(): a pending upload is counted there. */
    uint64_t device_version() const { return version_; }

    /* the pointer getData() hands out: valid for the life of the object.  Tracked blocks: made readable now, later reads
     * and writes through it are caught by the page protection.  Untracked build: invalidates the device copy (the caller
     * may write). */
    uint8_t *host_ptr()
    {
#ifdef CLOVER_HIP_NO_PAGE_TRACKING
        return host_rw();
#else
        host_ro();                                 /* (a view aliases caller memory, which is always current: write-through) */
        return host_;
#endif
    }
    /* host pointer that the CALLER IS ABOUT TO WRITE THROUGH (accessor methods: set(), clear(), ...) */
    uint8_t *host_rw()
    {
        if (!owns_host_) { commit(); return host_; }
        if (state_ == DEVICE_DIRTY) pull(HOST_DIRTY);
        else if (state_ == SHARED) set_state(HOST_DIRTY);
        return host_;
    }
    /* host pointer for reading only */
    const uint8_t *host_ro()
    {
        if (!owns_host_) { commit(); return host_; }
        if (state_ == DEVICE_DIRTY) pull(SHARED);
        return host_;
    }
    /* device pointer for reading: uploads if the host side is newer */
    const uint8_t *dev_ro()
    {
        ensure_dev();
        if (!owns_host_) {                         /* view: the caller may have written its memory at any time */
            detail::touch_for_read(host_, bytes_);
            check(clv_memcpy_h2d(dev_, host_, bytes_, nullptr), "host->device copy");
            check(clv_stream_sync(nullptr), "stream sync");
            ++version_;
        } else if (state_ == HOST_DIRTY) {
            check(clv_memcpy_h2d(dev_, host_, bytes_, nullptr), "host->device copy");
            check(clv_stream_sync(nullptr), "stream sync");
            set_state(SHARED);
            ++version_;
        }
        return dev_;
    }
    /* device pointer that a kernel is about to overwrite completely; follow the launch with commit() */
    uint8_t *dev_wo()
    {
        ensure_dev();
        ++version_;
        if (owns_host_) set_state(DEVICE_DIRTY);
        else pending_ = true;
        return dev_;
    }
    /* device pointer that a kernel updates in place; follow the launch with commit() */
    uint8_t *dev_rw()
    {
        dev_ro();
        ++version_;
        if (owns_host_) set_state(DEVICE_DIRTY);
        else pending_ = true;
        return dev_;
    }
    /* after the launches that wrote through dev_wo()/dev_rw(): views copy the result back into the caller's memory now */
    void commit()
    {
        if (owns_host_ || !pending_) return;
        pending_ = false;
        detail::touch_for_write(host_, bytes_);
        check(clv_memcpy_d2h(host_, dev_, bytes_, nullptr), "device->host copy");
    }
    bool device_is_current() const { return state_ != HOST_DIRTY; }

    /* called by the SIGSEGV handler: true if `addr` lies in this block and the fault was resolved */
    bool contains(const void *addr) const
    {
        const uint8_t *a = static_cast<const uint8_t *>(addr);
        return owns_host_ && host_ && a >= host_ && a < host_ + span_;
    }
    bool resolve_fault(const void *addr)
    {
        if (!contains(addr)) return false;
        if (state_ == DEVICE_DIRTY) { pull(SHARED); return true; }       /* a write faults once more and lands below */
        if (state_ == SHARED) { set_state(HOST_DIRTY); return true; }
        return false;                                                   /* HOST_DIRTY is unprotected: not ours */
    }
    Mirror *next_tracked() const { return next_; }

private:
    void ensure_dev()
    {
        if (!dev_) {
            void *p = nullptr;
            check(clv_malloc(&p, bytes_ ? bytes_ : 1), "device allocation");
            dev_ = static_cast<uint8_t *>(p);
        }
    }
    void protect(int prot)
    {
#ifndef CLOVER_HIP_NO_PAGE_TRACKING
        if (owns_host_ && host_ && mprotect(host_, span_, prot) != 0) {
            std::cout << "mprotect failed. Exiting ..." << std::endl;
            exit(1);
        }
#else
        (void)prot;
#endif
    }
    void set_state(State s)
    {
        if (s == state_) return;
        state_ = s;
        protect(s == HOST_DIRTY ? (PROT_READ | PROT_WRITE) : s == SHARED ? PROT_READ : PROT_NONE);
    }
    /* device -> host, ending in `after` (SHARED or HOST_DIRTY) */
    void pull(State after)
    {
        protect(PROT_READ | PROT_WRITE);
        check(clv_memcpy_d2h(host_, dev_, bytes_, nullptr), "device->host copy");
        state_ = HOST_DIRTY;
        set_state(after);
    }
    void link()
    {
        detail::Tracker &t = detail::tracker();
        t.acquire();
        next_ = t.head;
        prev_ = nullptr;
        if (t.head) t.head->prev_ = this;
        t.head = this;
        t.release();
    }
    void unlink()
    {
        detail::Tracker &t = detail::tracker();
        t.acquire();
        if (prev_) prev_->next_ = next_;
        else if (t.head == this) t.head = next_;
        if (next_) next_->prev_ = prev_;
        next_ = prev_ = nullptr;
        t.release();
    }
    void release()
    {
        if (dev_) clv_free(dev_);
        if (host_ && owns_host_) {
#ifndef CLOVER_HIP_NO_PAGE_TRACKING
            unlink();
            mprotect(host_, span_, PROT_READ | PROT_WRITE);            /* hand the pages back to the allocator usable */
#endif
            free(host_);
        }
        host_ = nullptr;
        dev_ = nullptr;
        bytes_ = 0;
        span_ = 0;
        state_ = HOST_DIRTY;
        pending_ = false;
    }

    uint8_t *host_;
    uint8_t *dev_;
    uint64_t bytes_, span_;
    volatile State state_;             /* also written by the fault handler: always re-read */
    bool owns_host_;
    volatile bool pending_;
    uint64_t version_;                 /* see device_version() */
    Mirror *next_, *prev_;             /* intrusive list of tracked blocks (detail::Tracker) */
};

namespace detail {
inline void fault_handler(int sig, siginfo_t *info, void *uctx)
{
    Tracker &t = tracker();
    const void *addr = info ? info->si_addr : nullptr;
    Mirror *hit = nullptr;
    if (addr) {
        t.acquire();
        for (Mirror *m = t.head; m && !hit; m = m->next_tracked())
            if (m->contains(addr)) hit = m;
        t.release();
        if (hit && hit->resolve_fault(addr)) return;       /* resolved: the faulting instruction is restarted */
    }
    /* not ours: behave as if this handler had never been installed */
    if (t.previous.sa_flags & SA_SIGINFO) {
        if (t.previous.sa_sigaction) { t.previous.sa_sigaction(sig, info, uctx); return; }
    } else if (t.previous.sa_handler == SIG_IGN) {
        return;
    } else if (t.previous.sa_handler != SIG_DFL && t.previous.sa_handler) {
        t.previous.sa_handler(sig);
        return;
    }
    signal(SIGSEGV, SIG_DFL);                          /* default action: the re-executed access terminates the process */
}
}  // namespace detail

/* One persistent 4-byte device slot + one pinned host word per thread: where dot() receives its result.  (A per-call
 * hipMalloc/hipFree cost more than the C1 kernel itself.) */
class ResultSlot {
public:
    ResultSlot() : dev_(nullptr), host_(nullptr) {}
    ~ResultSlot()
    {
        if (dev_) clv_free(dev_);
        if (host_) clv_host_free(host_);
    }
    float *device()
    {
        if (!dev_) {
            void *p = nullptr, *h = nullptr;
            check(clv_malloc(&p, 64), "device allocation");
            check(clv_host_alloc(&h, 64), "pinned host allocation");
            dev_ = static_cast<float *>(p);
            host_ = static_cast<float *>(h);
        }
        return dev_;
    }
    float fetch()
    {
        check(clv_memcpy_d2h(host_, dev_, sizeof(float), nullptr), "device->host copy");
        return *host_;
    }

private:
    float *dev_, *host_;
};
inline ResultSlot &result_slot()
{
    static thread_local ResultSlot slot;
    return slot;
}

/*
 * Per-object XORShift state (CloverRandom.h:45-114).  The reference seeds from RDRAND at construction;
 * here std::random_device supplies the two seeds and the state lives in device memory, where the
 * stochastic kernels advance it.  Created lazily: objects that never quantize stochastically cost nothing.
 */
class RandomState {
public:
    RandomState() : dev_(nullptr) {}
    ~RandomState() { if (dev_) clv_free(dev_); }
    RandomState(const RandomState &) = delete;
    RandomState &operator=(const RandomState &) = delete;

    uint64_t *device()
    {
        if (!dev_) {
            std::random_device rd;
            const uint64_t k1 = ((uint64_t)rd() << 32) | rd(), k2 = ((uint64_t)rd() << 32) | rd();
            alloc();
            check(clv_rng_seed(dev_, k1 | 1, k2 | 1, nullptr), "rng seed");
        }
        return dev_;
    }
    /* avx_xorshift128plus_init(key1, key2) */
    void seed(uint64_t key1, uint64_t key2)
    {
        alloc();
        check(clv_rng_seed(dev_, key1, key2, nullptr), "rng seed");
    }
    /* CloverRandom::setRandomKeys(random_key1, random_key2): the 4 x 64-bit lanes of each __m256i */
    void set(const uint64_t key1[4], const uint64_t key2[4])
    {
        alloc();
        check(clv_rng_set(dev_, key1, key2, nullptr), "rng set");
    }
    void get(uint64_t key1[4], uint64_t key2[4])
    {
        check(clv_rng_get(device(), key1, key2, nullptr), "rng get");
    }

private:
    void alloc()
    {
        if (!dev_) {
            void *p = nullptr;
            check(clv_malloc(&p, CLV_RNG_STATE_BYTES), "device allocation");
            dev_ = static_cast<uint64_t *>(p);
        }
    }
    uint64_t *dev_;
};

/* rounding mode switch, same macro as the reference build (CMakeLists.txt:78-80) */
inline uint64_t *rng_or_null(RandomState &r)
{
#ifdef CLOVER_STOCHASTIC_ROUNDING_DISABLED
    (void)r;
    return nullptr;
#else
    return r.device();
#endif
}

}  // namespace clover_hip

/* CloverRandom::setRandomKeys(__m256i key1, __m256i key2) (CloverRandom.h:90-94), for callers compiled with AVX: the
 * containers get an overload that takes the two key registers as the reference does. */
#if defined(__AVX__)
#include <immintrin.h>
namespace clover_hip {
inline void set_keys_m256(RandomState &r, __m256i key1, __m256i key2)
{
    uint64_t k1[4], k2[4];
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(k1), key1);
    _mm256_storeu_si256(reinterpret_cast<__m256i *>(k2), key2);
    r.set(k1, k2);
}
}  // namespace clover_hip
#define CLOVER_HIP_M256_KEYS 1
#endif

#endif
