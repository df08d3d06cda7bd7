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
 * methods (get(), set(), getBits() ...) switch state directly and never fault.
 *
 * Threads (round 3).  The reference's raw pointers may be read by any number of host threads at once; so may these:
 *   - every state change of a block -- from a method or from a fault -- runs under that block's own lock and re-checks the state
 *     after taking it, so two threads faulting on one block resolve it once (the second finds it done and just retries its access),
 *     and a fault racing an upload / a kernel hand-over on another thread is ordered with it;
 *   - the handler finds the block by binary search in an address-sorted table under the tracker lock and PINS it before dropping that
 *     lock; a destructor unlinks the block first and then waits for the pins to drain, so the handler never touches a dead object;
 *   - the handler itself only does atomics, mprotect and futex calls.  The device -> host copy a DEVICE_DIRTY fault needs is done by a
 *     helper thread (started with the first tracked block) in ordinary context -- no HIP runtime call, allocation or stdio happens in
 *     signal context.  (-DCLOVER_HIP_FAULT_INLINE, or a process that forked, copies from the handler as rounds 1-2 did.)
 *   What stays the caller's job, as in the reference: two threads CALLING METHODS that write one object need their own ordering
 *   (objects are not thread-safe, CloverVector4.h:59-66: mutable RNG state, shared scratch).
 * Limits, all loud rather than silent: a system call given a pointer into a DEVICE_DIRTY block returns EFAULT instead of faulting, and
 * a HIP runtime call given such a pointer faults inside the runtime, where the helper's copy may need a lock the faulting thread
 * holds (use the accessors or touch the memory first -- the containers do that for their own copies, detail::touch_for_*);
 * debuggers stop on the resolved SIGSEGVs (gdb: `handle SIGSEGV nostop noprint`); fork(): tracked blocks are MAP_SHARED memory files, so a
 * child shares them with its parent instead of getting copy-on-write pages, and it has no helper thread -- a child must not touch the
 * parent's containers (exec or _exit).
 *
 * EXPLICIT RESIDENCY (-DCLOVER_HIP_EXPLICIT_SYNC; the older spelling -DCLOVER_HIP_NO_PAGE_TRACKING is the same switch).  For hosts that
 * own SIGSEGV themselves (a JVM, sanitizers, a debugger session) or must fork freely: NO signal handler is installed, NO page is ever
 * protected, no helper thread, no memory files -- the host block is one plain posix_memalign allocation as in the reference.  Coherence is
 * then by call instead of by fault:
 *     getData() / getScales()   bring the device's results to the host NOW (if the device copy is newer) and mark the host copy as the
 *                               one that counts: whatever the caller writes through the pointer reaches the next device operation;
 *     accessors (get, set, getBits, setBits, clear, toString ...) synchronise by themselves, as in the default build;
 *     toDevice() / toHost()     optional: move the bytes at a moment of the caller's choosing (e.g. outside a timed region).
 * The one rule: RE-TAKE the pointer after a device operation (quantize, mvm, scaleAndAdd, threshold, transpose, restore ...) on the
 * object -- a pointer kept across one still points at valid memory, but shows the bytes from before it, and writes through it are not
 * seen.  (The reference's contract, CloverVector4.h:229-237, is "valid for the life of the object"; that half still holds.)
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

#include <errno.h>
#include <pthread.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#if defined(__linux__)
#include <linux/futex.h>
#endif
#if defined(__x86_64__)
#include <ucontext.h>
#endif

#include "clover_hip.h"

#if defined(CLOVER_HIP_EXPLICIT_SYNC) && !defined(CLOVER_HIP_NO_PAGE_TRACKING)
#define CLOVER_HIP_NO_PAGE_TRACKING 1
#endif

#ifdef CLOVER_HIP_TEST_MPROTECT_ENOMEM
static int clover_hip_test_mprotect_enomem = 0;      /* tests set it to 1: the next restricting mprotect calls "fail" with ENOMEM */
#endif

namespace clover_hip {

inline void check(int rc, const char *what)
{
    if (rc != CLV_OK) {
        std::cout << what << " failed: " << clv_last_error() << ". Exiting ..." << std::endl;
        exit(1);
    }
}

inline uint64_t round_up(uint64_t v, uint64_t m) { return v % m ? v + m - (v % m) : v; }

/* ONE exactness switch for the containers (round 5).  Two methods of this path have a definition-bound slow form and a fast form:
 *
 *   method                     reference's bits                                            fast
 *   dot()                      CLV_DOT_EXACT: the reference's 16 sequential fma chains     CLV_DOT_FAST: exact block integers, fp32 tree order,
 *                              (n/128 dependent fmas: 0.38 ms at n = 2^24, what one host    within 2e-6 * sum|terms| (the reference's own
 *                              core takes for the same order)                               dot_parallel is "any order" too); memory-bound
 *   threshold()                CLV_THRESHOLD_REFERENCE: the reference's K-entry min-heap   CLV_THRESHOLD_FAST: radix select; the same MULTISET of
 *                              walk, survivors index for index (one wavefront, about 0.4 us  magnitudes survives, ties at the K-th value go to
 *                              per heap insert: 0.85 ms at N = 8192, K = 1024)              the lowest indices (5 us at N = 8192)
 *
 *   -DCLOVER_REFERENCE_BITS   both methods return the reference's bits.  THE DEFAULT when neither macro is given: a drop-in first of all
 *                             reproduces what it replaces (a Q_IHT through CloverIHT.h then follows the reference's trajectory tie for tie);
 *   -DCLOVER_FAST             both take the fast form (a quantized IHT iteration at N = 8192: 20 us instead of 0.83 ms).
 * At run time: clover_hip::set_exactness(clover_hip::REFERENCE_BITS | clover_hip::FAST) switches both; CLV_EXACTNESS=fast|reference in the
 * environment picks the start value of a build without either macro.  Finer: set_dot_mode() / set_threshold_mode(), and the older
 * single-method macros -DCLOVER_DOT_FAST, -DCLOVER_THRESHOLD_REFERENCE / -DCLOVER_THRESHOLD_FAST and CLV_THRESHOLD_REFERENCE=0|1 still
 * override their one method.  dot_parallel() / dot_fast() are always the fast order, dot_exact() always the reference's. */
#if defined(CLOVER_REFERENCE_BITS) && defined(CLOVER_FAST)
#error "-DCLOVER_REFERENCE_BITS and -DCLOVER_FAST exclude each other"
#endif
enum Exactness { REFERENCE_BITS = 0, FAST = 1 };
inline int start_exactness()
{
#if defined(CLOVER_FAST)
    return FAST;
#elif defined(CLOVER_REFERENCE_BITS)
    return REFERENCE_BITS;
#else
    const char *e = getenv("CLV_EXACTNESS");
    return (e && (e[0] == 'f' || e[0] == 'F')) ? FAST : REFERENCE_BITS;
#endif
}
inline int &threshold_mode_slot()
{
#if defined(CLOVER_THRESHOLD_REFERENCE)
    static int mode = CLV_THRESHOLD_REFERENCE;
#elif defined(CLOVER_THRESHOLD_FAST)
    static int mode = CLV_THRESHOLD_FAST;
#else
    static int mode = [] {
        const char *e = getenv("CLV_THRESHOLD_REFERENCE");
        if (e && e[0]) return e[0] != '0' ? CLV_THRESHOLD_REFERENCE : CLV_THRESHOLD_FAST;
        return start_exactness() == FAST ? CLV_THRESHOLD_FAST : CLV_THRESHOLD_REFERENCE;
    }();
#endif
    return mode;
}
inline int &dot_mode_slot()
{
#if defined(CLOVER_DOT_FAST)
    static int mode = CLV_DOT_FAST;
#else
    static int mode = start_exactness() == FAST ? CLV_DOT_FAST : CLV_DOT_EXACT;
#endif
    return mode;
}
inline int threshold_mode() { return threshold_mode_slot(); }
inline void set_threshold_mode(int mode) { threshold_mode_slot() = mode; }
inline int dot_mode() { return dot_mode_slot(); }
inline void set_dot_mode(int mode) { dot_mode_slot() = mode; }
inline void set_exactness(int e)
{
    set_dot_mode(e == FAST ? CLV_DOT_FAST : CLV_DOT_EXACT);
    set_threshold_mode(e == FAST ? CLV_THRESHOLD_FAST : CLV_THRESHOLD_REFERENCE);
}

class Mirror;

/* blocks of this process that fell back from page tracking to explicit residency because mprotect ran out of map entries (see
 * Mirror::protect): 0 in a healthy process.  A host that keeps pointers across device operations can assert on it. */
inline std::atomic<uint64_t> &untracked_count()
{
    static std::atomic<uint64_t> n(0);
    return n;
}
inline uint64_t untracked_blocks() { return untracked_count().load(std::memory_order_relaxed); }

namespace detail {

inline long futex(std::atomic<int> *word, int op, int val)
{
#if defined(__linux__)
    return syscall(SYS_futex, reinterpret_cast<int *>(word), op, val, nullptr, nullptr, 0);      /* a plain system call: async-signal-safe */
#else
    (void)word; (void)op; (void)val;
    return 0;
#endif
}

/* Lock on a lock-free atomic, usable from the signal handler: 0 free, 1 held, 2 held with sleepers.  Spins briefly (the common hold
 * time is a state flip and one mprotect), then sleeps on the word with futex -- a block's lock is also held across a blocking copy of
 * the whole block (milliseconds for a large one), and waiters must not burn cores that a cgroup-throttled host needs for the copy. */
struct SpinLock {
    std::atomic<int> v;
    SpinLock() : v(0) {}
    void lock()
    {
        int c = 0;
        if (v.compare_exchange_strong(c, 1, std::memory_order_acquire)) return;
        for (int spin = 0; spin < 400; spin++) {
            c = 0;
            if (v.load(std::memory_order_relaxed) == 0 && v.compare_exchange_strong(c, 1, std::memory_order_acquire)) return;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
#if defined(__linux__)
        c = v.exchange(2, std::memory_order_acquire);
        while (c != 0) {
            futex(&v, FUTEX_WAIT, 2);
            c = v.exchange(2, std::memory_order_acquire);
        }
#else
        while (v.exchange(1, std::memory_order_acquire)) {}
#endif
    }
    void unlock()
    {
#if defined(__linux__)
        if (v.exchange(0, std::memory_order_release) == 2) futex(&v, FUTEX_WAKE, 1);
#else
        v.store(0, std::memory_order_release);
#endif
    }
};
struct Guard {
    SpinLock &l;
    explicit Guard(SpinLock &lk) : l(lk) { l.lock(); }
    ~Guard() { l.unlock(); }
    Guard(const Guard &) = delete;
    Guard &operator=(const Guard &) = delete;
};

/* process-wide table of tracked blocks (sorted by host address), the chained SIGSEGV handler and the helper thread that performs
 * device -> host copies on behalf of faulting threads.  Lives in an inline function so that every translation unit including
 * this header shares one instance (C++11). */
struct Tracker {
    SpinLock lock;                     /* guards table / count / cap */
    Mirror **table;
    size_t count, cap;
    struct sigaction previous;
    bool installed;
    /* helper thread mailbox: one request at a time (req_lock), word: 0 idle, 1 posted, 2 done */
    SpinLock req_lock;
    std::atomic<int> req_word;
    Mirror *req_mirror;
    int req_after;
    std::atomic<int> helper_pid;       /* pid of the process the helper thread lives in (0: none): a forked child has no helper */
    pthread_t helper;
    Tracker() : table(nullptr), count(0), cap(0), installed(false), req_word(0), req_mirror(nullptr), req_after(0), helper_pid(0), helper() {}
};
inline Tracker &tracker()
{
    static Tracker t;
    return t;
}
inline void fault_handler(int sig, siginfo_t *info, void *uctx);
inline void *helper_main(void *);
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
#if !defined(CLOVER_HIP_FAULT_INLINE) && defined(__linux__)
    sigset_t all, old;                          /* the helper never handles signals itself */
    sigfillset(&all);
    pthread_sigmask(SIG_SETMASK, &all, &old);
    if (pthread_create(&t.helper, nullptr, helper_main, nullptr) == 0) {
        pthread_detach(t.helper);
        t.helper_pid.store((int)getpid(), std::memory_order_release);
    }
    pthread_sigmask(SIG_SETMASK, &old, nullptr);
#endif
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

    Mirror() : host_(nullptr), alias_(nullptr), dev_(nullptr), bytes_(0), span_(0), state_(HOST_DIRTY), owns_host_(true), mapped_(false), pending_(false), untracked_(false), restricted_(false), version_(0), pins_(0), spurious_(0) {}
    /* false once the kernel refused to protect this block's pages (ENOMEM: vm.max_map_count) while they were fully open: the block then
     * follows the EXPLICIT-RESIDENCY rules -- re-take getData() after a device operation -- for the rest of its life.  Process-wide:
     * clover_hip::untracked_blocks(). */
    bool tracked() const { return !untracked_.load(std::memory_order_acquire); }
    ~Mirror() { release(); }
    Mirror(const Mirror &) = delete;
    Mirror &operator=(const Mirror &) = delete;

    void allocate(uint64_t bytes)
    {
        release();
        bytes_ = bytes;
        const uint64_t page = (uint64_t)sysconf(_SC_PAGESIZE);
        span_ = round_up(bytes ? bytes : 1, page);                 /* whole pages: protection must not touch a neighbour */
        map_block((size_t)page);
        owns_host_ = true;
        state_.store(HOST_DIRTY, std::memory_order_release);
#ifndef CLOVER_HIP_NO_PAGE_TRACKING
        detail::install_handler();
        link();
#endif
    }

    /* non-owning view over user memory (CloverVector4(n, values, scales), CloverVector4.h:114-119) */
    void adopt(void *host, uint64_t bytes)
    {
        release();
        host_ = alias_ = static_cast<uint8_t *>(host);
        bytes_ = bytes;
        span_ = 0;
        owns_host_ = false;
        state_.store(HOST_DIRTY, std::memory_order_release);
    }

    uint64_t bytes() const { return bytes_; }
    State state() const { return (State)state_.load(std::memory_order_acquire); }
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
        if (!tracked()) return host_rw();          /* a block whose protection the kernel refused (ENOMEM): explicit-residency rules */
        host_ro();                                 /* (a view aliases caller memory, which is always current: write-through) */
        return host_;
#endif
    }
    /* host pointer that the CALLER IS ABOUT TO WRITE THROUGH (accessor methods: set(), clear(), ...) */
    uint8_t *host_rw()
    {
        detail::Guard g(lock_);
        if (!owns_host_) { commit_locked(); return host_; }
        const State s = state();
        if (s == DEVICE_DIRTY) pull(HOST_DIRTY);
        else if (s == SHARED) set_state(HOST_DIRTY);
        return host_;
    }
    /* host pointer for reading only */
    const uint8_t *host_ro()
    {
        detail::Guard g(lock_);
        if (!owns_host_) { commit_locked(); return host_; }
        if (state() == DEVICE_DIRTY) pull(SHARED);
        return host_;
    }
    /* device pointer for reading: uploads if the host side is newer */
    const uint8_t *dev_ro()
    {
        detail::Guard g(lock_);
        return dev_ro_locked();
    }
    /* device pointer that a kernel is about to overwrite completely; follow the launch with commit() */
    uint8_t *dev_wo()
    {
        detail::Guard g(lock_);
        ensure_dev();
        ++version_;
        if (owns_host_) set_state(DEVICE_DIRTY);
        else pending_ = true;
        return dev_;
    }
    /* device pointer that a kernel updates in place; follow the launch with commit() */
    uint8_t *dev_rw()
    {
        detail::Guard g(lock_);
        dev_ro_locked();
        ++version_;
        if (owns_host_) set_state(DEVICE_DIRTY);
        else pending_ = true;
        return dev_;
    }
    /* after the launches that wrote through dev_wo()/dev_rw(): views copy the result back into the caller's memory now */
    void commit()
    {
        detail::Guard g(lock_);
        commit_locked();
    }
    bool device_is_current() const { return state() != HOST_DIRTY; }
    /* true when the host block is two mappings of one memory file (copies land behind the protection); false: plain allocation */
    bool double_mapped() const { return mapped_; }

    /* ---- used by the tracker / the SIGSEGV handler ---- */
    const uint8_t *block_begin() const { return host_; }
    bool contains(const void *addr) const
    {
        const uint8_t *a = static_cast<const uint8_t *>(addr);
        return owns_host_ && host_ && a >= host_ && a < host_ + span_;
    }
    void pin() { pins_.fetch_add(1, std::memory_order_acq_rel); }
    void unpin() { pins_.fetch_sub(1, std::memory_order_acq_rel); }
    /* Signal context, block pinned by the caller.  access: 0 read, 1 write, -1 unknown (no error code on this platform: a fault in a
     * SHARED block is then taken for a write, as in rounds 1-2).  True = resolved (or already resolved by another thread): the
     * faulting instruction is simply restarted. */
    bool resolve_fault(const void *addr, int access)
    {
        detail::Guard g(lock_);
        if (!contains(addr)) return false;                               /* released and re-used between lookup and lock */
        const State s = state();
        if (s == DEVICE_DIRTY) {
            pull_from_handler(access == 1 ? HOST_DIRTY : SHARED);         /* an access of unknown kind that was a write faults once more */
            spurious_ = 0;
            return true;
        }
        if (s == SHARED) {
            if (access != 0) { set_state(HOST_DIRTY); spurious_ = 0; return true; }
            return ++spurious_ < 1000;                                   /* a read of a readable block: another thread pulled it while we waited */
        }
        /* HOST_DIRTY pages are unprotected: either another thread resolved this very fault while we waited for the lock (retry), or
         * the fault is not a protection fault of ours (the bound turns a loop into the crash it should be) */
        return ++spurious_ < 1000;
    }
    /* helper thread: the copy a faulting thread asked for (that thread holds lock_ and waits) */
    void pull_for_request(int after) { pull((State)after); }

private:
    const uint8_t *dev_ro_locked()
    {
        ensure_dev();
        if (!owns_host_) {                         /* view: the caller may have written its memory at any time */
            detail::touch_for_read(host_, bytes_);
            check(clv_memcpy_h2d(dev_, host_, bytes_, nullptr), "host->device copy");
            check(clv_stream_sync(nullptr), "stream sync");
            ++version_;
        } else if (state() == HOST_DIRTY) {
            /* read-only FIRST, then the copy (from the alias), then SHARED: a write through a kept pointer on another thread either
             * lands before the protection (and is uploaded) or faults, waits for lock_ and finds SHARED -> HOST_DIRTY: never lost */
            protect(PROT_READ);
            check(clv_memcpy_h2d(dev_, alias_, bytes_, nullptr), "host->device copy");
            check(clv_stream_sync(nullptr), "stream sync");
            state_.store(SHARED, std::memory_order_release);
            spurious_ = 0;
            ++version_;
        }
        return dev_;
    }
    void commit_locked()
    {
        if (owns_host_ || !pending_) return;
        pending_ = false;
        detail::touch_for_write(host_, bytes_);
        check(clv_memcpy_d2h(host_, dev_, bytes_, nullptr), "device->host copy");
    }
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
        if (!tracked() || !owns_host_ || !host_) return;
        int rc;
#ifdef CLOVER_HIP_TEST_MPROTECT_ENOMEM                                    /* tests: pretend the kernel ran out of map entries */
        if (prot != (PROT_READ | PROT_WRITE) && clover_hip_test_mprotect_enomem) { rc = -1; errno = ENOMEM; }
        else
#endif
        rc = mprotect(host_, span_, prot);
        /* restricted_ = the pages are NOT fully open right now (it follows the last successful call, r5: a long-lived block that is back
         * to read/write when the kernel runs out of map entries can fall back like a fresh one) */
        if (rc == 0) { restricted_ = prot != (PROT_READ | PROT_WRITE); return; }
        if (errno == ENOMEM && prot != (PROT_READ | PROT_WRITE) && !restricted_) {
            /* vm.max_map_count reached (a plain-allocation block: protecting a part of the heap splits its mapping).  The block is fully
             * open at this moment, so it can simply stay open: from here on it follows the EXPLICIT-RESIDENCY rules (getData() pulls and
             * marks the host copy; a pointer kept across a device operation shows the bytes from before it) instead of ending the
             * process.  Observable: Mirror::tracked(), clover_hip::untracked_blocks(), and one line on stderr per block. */
            untracked_.store(true, std::memory_order_release);
            untracked_count().fetch_add(1, std::memory_order_relaxed);
            static const char msg[] = "clover_hip: mprotect: out of map entries, a block falls back to explicit residency\n";
            ssize_t w = write(2, msg, sizeof(msg) - 1);
            (void)w;
            return;
        }
        static const char msg[] = "mprotect failed. Exiting ...\n";          /* may run in signal context: write(2), _exit(2) */
        ssize_t w = write(1, msg, sizeof(msg) - 1);
        (void)w;
        _exit(1);
#else
        (void)prot;
#endif
    }
    void set_state(State s)                                  /* lock_ held */
    {
        if (s == state()) return;
        state_.store(s, std::memory_order_release);
        protect(s == HOST_DIRTY ? (PROT_READ | PROT_WRITE) : s == SHARED ? PROT_READ : PROT_NONE);
        spurious_ = 0;
    }
    /* device -> host, ending in `after` (SHARED or HOST_DIRTY); lock_ held; ordinary context */
    void pull(State after)
    {
        /* the copy lands through the alias mapping while the user-visible one is still PROT_NONE: a second thread reading through a
         * kept pointer faults and waits for lock_ instead of seeing half a copy.  (Without the alias -- memfd_create unavailable --
         * the pages have to be opened first, and such a reader may see bytes of either version until the copy ends.) */
        if (alias_ == host_) protect(PROT_READ | PROT_WRITE);
        check(clv_memcpy_d2h(alias_, dev_, bytes_, nullptr), "device->host copy");
        if (alias_ == host_) state_.store(HOST_DIRTY, std::memory_order_release);
        set_state(after);
    }
    /* the same from the SIGSEGV handler: hand the copy to the helper thread and wait for it on a futex */
    void pull_from_handler(State after)
    {
        detail::Tracker &t = detail::tracker();
#if !defined(CLOVER_HIP_FAULT_INLINE) && defined(__linux__)
        const int hp = t.helper_pid.load(std::memory_order_acquire);
        if (hp != 0 && hp == (int)getpid() && !pthread_equal(pthread_self(), t.helper)) {
            t.req_lock.lock();
            t.req_mirror = this;
            t.req_after = (int)after;
            t.req_word.store(1, std::memory_order_release);
            detail::futex(&t.req_word, FUTEX_WAKE, 1);
            while (t.req_word.load(std::memory_order_acquire) != 2) detail::futex(&t.req_word, FUTEX_WAIT, 1);
            t.req_word.store(0, std::memory_order_release);
            t.req_lock.unlock();
            return;
        }
#endif
        (void)t;
        pull(after);                                /* no helper (inline build, forked child): as rounds 1-2 */
    }
    void link();
    void unlink();
    void release()
    {
        if (host_ && owns_host_) {
#ifndef CLOVER_HIP_NO_PAGE_TRACKING
            unlink();                                                   /* no new fault can find this block ... */
            while (pins_.load(std::memory_order_acquire) != 0) {}       /* ... and the ones that did have finished with it */
            lock_.lock();
            if (!mapped_) mprotect(host_, span_, PROT_READ | PROT_WRITE);   /* hand the pages back to the allocator usable */
            lock_.unlock();
#endif
            if (mapped_) {
                munmap(host_, span_);
                munmap(alias_, span_);
            } else {
                free(host_);
            }
        }
        if (dev_) clv_free(dev_);
        host_ = nullptr;
        alias_ = nullptr;
        mapped_ = false;
        dev_ = nullptr;
        bytes_ = 0;
        span_ = 0;
        state_.store(HOST_DIRTY, std::memory_order_release);
        pending_ = false;
        untracked_.store(false, std::memory_order_release);
        restricted_ = false;
    }
    /* the host block: two mappings of one anonymous memory file where the platform has memfd_create (tracked builds), else one
     * page-aligned allocation as in the reference (CloverVector4.h:70-79) */
    void map_block(size_t page)
    {
#if !defined(CLOVER_HIP_NO_PAGE_TRACKING) && defined(__linux__) && defined(SYS_memfd_create)
        const int fd = (int)syscall(SYS_memfd_create, "clover_block", 1u /* MFD_CLOEXEC */);
        if (fd >= 0) {
            void *a = MAP_FAILED, *b = MAP_FAILED;
            if (ftruncate(fd, (off_t)span_) == 0) {
                a = mmap(nullptr, span_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (a != MAP_FAILED) b = mmap(nullptr, span_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            }
            close(fd);
            if (a != MAP_FAILED && b != MAP_FAILED) {
                host_ = static_cast<uint8_t *>(a);
                alias_ = static_cast<uint8_t *>(b);
                mapped_ = true;
                return;
            }
            if (a != MAP_FAILED) munmap(a, span_);
        }
#endif
        void *p = nullptr;
        if (posix_memalign(&p, page, span_) != 0) {
            std::cout << "Could not allocate host memory. Exiting ..." << std::endl;
            exit(1);
        }
        host_ = alias_ = static_cast<uint8_t *>(p);
        mapped_ = false;
    }

    uint8_t *host_;                    /* the address the user sees: this mapping carries the protection */
    uint8_t *alias_;                   /* the same pages, always read/write, never handed out: where device -> host copies land */
    uint8_t *dev_;
    uint64_t bytes_, span_;
    std::atomic<int> state_;           /* a State; written under lock_, read anywhere */
    bool owns_host_;
    bool mapped_;                      /* host_/alias_ are two mmaps of one memfd (else: posix_memalign, alias_ == host_) */
    bool pending_;
    std::atomic<bool> untracked_;      /* mprotect refused with ENOMEM while the block was fully open: explicit-residency rules from then on
                                        * (read without lock_ by host_ptr() / tracked(): atomic) */
    bool restricted_;                  /* the pages are not fully open at the moment (under lock_) */
    uint64_t version_;                 /* see device_version() */
    detail::SpinLock lock_;            /* every state change of this block, from methods and from the fault handler */
    std::atomic<int> pins_;            /* fault handlers between table lookup and resolution: the destructor waits for 0 */
    int spurious_;                     /* faults that found nothing to do since the last state change (under lock_) */
};

inline void Mirror::link()
{
    detail::Tracker &t = detail::tracker();
    detail::Guard g(t.lock);
    if (t.count == t.cap) {
        const size_t ncap = t.cap ? 2 * t.cap : 64;
        Mirror **nt = static_cast<Mirror **>(realloc(t.table, ncap * sizeof(Mirror *)));
        if (!nt) {
            std::cout << "Could not allocate host memory. Exiting ..." << std::endl;
            exit(1);
        }
        t.table = nt;
        t.cap = ncap;
    }
    size_t lo = 0, hi = t.count;                       /* first entry whose block begins above ours */
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (t.table[mid]->block_begin() < host_) lo = mid + 1;
        else hi = mid;
    }
    memmove(t.table + lo + 1, t.table + lo, (t.count - lo) * sizeof(Mirror *));
    t.table[lo] = this;
    t.count++;
}
inline void Mirror::unlink()
{
    detail::Tracker &t = detail::tracker();
    detail::Guard g(t.lock);
    for (size_t i = 0; i < t.count; i++)
        if (t.table[i] == this) {
            memmove(t.table + i, t.table + i + 1, (t.count - i - 1) * sizeof(Mirror *));
            t.count--;
            break;
        }
}

namespace detail {
/* the tracked block that contains addr, pinned; or NULL.  Signal context: a spin lock and a binary search. */
inline Mirror *find_and_pin(const void *addr)
{
    Tracker &t = tracker();
    Guard g(t.lock);
    size_t lo = 0, hi = t.count;                       /* last entry whose block begins at or below addr */
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (static_cast<const void *>(t.table[mid]->block_begin()) <= addr) lo = mid + 1;
        else hi = mid;
    }
    if (lo == 0) return nullptr;
    Mirror *m = t.table[lo - 1];
    if (!m->contains(addr)) return nullptr;
    m->pin();
    return m;
}
inline void *helper_main(void *)
{
    Tracker &t = tracker();
    for (;;) {
        int w;
        while ((w = t.req_word.load(std::memory_order_acquire)) != 1) futex(&t.req_word, FUTEX_WAIT, w);
        t.req_mirror->pull_for_request(t.req_after);
        t.req_word.store(2, std::memory_order_release);
        futex(&t.req_word, FUTEX_WAKE, 64);
    }
    return nullptr;
}
inline void fault_handler(int sig, siginfo_t *info, void *uctx)
{
    Tracker &t = tracker();
    const int saved_errno = errno;
    const void *addr = info ? info->si_addr : nullptr;
    int access = -1;
#if defined(__x86_64__) && defined(REG_ERR)
    if (uctx) {
        const unsigned long err = (unsigned long)static_cast<ucontext_t *>(uctx)->uc_mcontext.gregs[REG_ERR];
        access = (err & 16) ? 2 : (err & 2) ? 1 : 0;                /* page-fault error code: bit 1 write, bit 4 instruction fetch */
    }
#endif
    if (addr && access != 2) {
        if (Mirror *hit = find_and_pin(addr)) {
            const bool ok = hit->resolve_fault(addr, access);
            hit->unpin();
            if (ok) { errno = saved_errno; return; }                 /* resolved: the faulting instruction is restarted */
        }
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
    /* stochastic methods of the owning object become capturable into a hipGraph (clv_rng_graph_mode, clover_hip.h) */
    void graph_mode(bool on) { check(clv_rng_graph_mode(device(), on ? 1 : 0, nullptr), "rng graph mode"); }

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

/* the reference's heap entry (CloverBase.h:202-211) for threshold_min_heap(idx_t *, k): callers that bring their own heap memory */
union Restorator {
    int32_t i;
    float f;
};
typedef struct {
    float value;
    Restorator bits;
    uint64_t idx;
} idx_t;

/* threshold_min_heap of the 4- and 8-bit vectors: the reference's walk on the device (clvN_threshold_heap) and its heap -- {|value|, index}
 * per entry, in the reference's array order -- copied into the caller's idx_t array; the caller fills in `bits` from its own values */
typedef int (*threshold_heap_fn)(int8_t *, const float *, uint64_t, uint64_t, uint64_t, void *, void *, void *);
inline void threshold_heap_to_host(threshold_heap_fn fn, int8_t *dev_values, const float *dev_scales, uint64_t length, uint64_t length_pad,
                                   idx_t *min_heap, uint64_t k, const char *what)
{
    void *heap_dev = nullptr;
    check(clv_malloc(&heap_dev, k * 8), what);
    const int rc = fn(dev_values, dev_scales, length, length_pad, k, heap_dev, nullptr, nullptr);
    if (rc) { clv_free(heap_dev); check(rc, what); }
    uint32_t *pairs = static_cast<uint32_t *>(malloc(k * 8));
    if (!pairs) { std::cout << "We ran out of memory, while allocating thresholding memory. Exiting ..." << std::endl; exit(1); }
    check(clv_memcpy_d2h(pairs, heap_dev, k * 8, nullptr), what);
    clv_free(heap_dev);
    for (uint64_t i = 0; i < k; i++) {
        memcpy(&min_heap[i].value, &pairs[2 * i], 4);
        min_heap[i].idx = pairs[2 * i + 1];
        min_heap[i].bits.i = 0;
    }
    free(pairs);
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
