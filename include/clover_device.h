/*
 * clover_device.h -- host/device mirror plumbing shared by the drop-in containers in this directory
 * (CloverVector32/4, CloverMatrix32/4).  Own code; the reference has no counterpart because it is CPU-only.
 *
 * Model: every container owns ONE page-aligned host block laid out exactly like the reference's
 * (CloverVector4.h:68-103: values immediately followed by scales, so user-visible pointer arithmetic on
 * getData()/getScales() keeps working) and a lazily created HBM mirror of the same bytes.  Two validity
 * flags track which side is current; kernels run on the device mirror, host accessors pull it back.
 * Handing out a mutable host pointer (getData()) conservatively invalidates the device side.
 *
 * Errors follow the reference's convention (CloverVector4.h:100-101, CloverMatrix4.h:779-782): a message
 * on stdout and exit(1) -- the C ABI underneath returns status codes instead.
 */
#ifndef CLOVER_DEVICE_H
#define CLOVER_DEVICE_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>

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

/* One host block + one device block of `bytes` bytes with validity tracking. */
class Mirror {
public:
    Mirror() : host_(nullptr), dev_(nullptr), bytes_(0), host_valid_(true), dev_valid_(false), owns_host_(true) {}
    ~Mirror() { release(); }
    Mirror(const Mirror &) = delete;
    Mirror &operator=(const Mirror &) = delete;

    void allocate(uint64_t bytes)
    {
        release();
        bytes_ = bytes;
        void *p = nullptr;
        if (posix_memalign(&p, (size_t)sysconf(_SC_PAGESIZE), bytes ? bytes : 1) != 0) {
            std::cout << "Could not allocate host memory. Exiting ..." << std::endl;
            exit(1);
        }
        host_ = static_cast<uint8_t *>(p);
        owns_host_ = true;
        host_valid_ = true;
        dev_valid_ = false;
    }

    /* non-owning view over user memory (CloverVector4(n, values, scales), CloverVector4.h:114-119) */
    void adopt(void *host, uint64_t bytes)
    {
        release();
        host_ = static_cast<uint8_t *>(host);
        bytes_ = bytes;
        owns_host_ = false;
        host_valid_ = true;
        dev_valid_ = false;
    }

    uint64_t bytes() const { return bytes_; }

    /* host pointer for reading AND writing: pulls the device copy back and invalidates it */
    uint8_t *host_rw()
    {
        pull();
        dev_valid_ = false;
        return host_;
    }
    /* host pointer for reading only */
    const uint8_t *host_ro()
    {
        pull();
        return host_;
    }
    /* device pointer for reading: uploads if the host side is newer */
    const uint8_t *dev_ro()
    {
        ensure_dev();
        if (!dev_valid_) {
            check(clv_memcpy_h2d(dev_, host_, bytes_, nullptr), "host->device copy");
            check(clv_stream_sync(nullptr), "stream sync");
            dev_valid_ = true;
        }
        return dev_;
    }
    /* device pointer that a kernel is about to overwrite completely */
    uint8_t *dev_wo()
    {
        ensure_dev();
        dev_valid_ = true;
        host_valid_ = false;
        return dev_;
    }
    /* device pointer that a kernel updates in place */
    uint8_t *dev_rw()
    {
        dev_ro();
        host_valid_ = false;
        return dev_;
    }
    bool device_is_current() const { return dev_valid_; }

private:
    void ensure_dev()
    {
        if (!dev_) {
            void *p = nullptr;
            check(clv_malloc(&p, bytes_ ? bytes_ : 1), "device allocation");
            dev_ = static_cast<uint8_t *>(p);
        }
    }
    void pull()
    {
        if (!host_valid_) {
            check(clv_memcpy_d2h(host_, dev_, bytes_, nullptr), "device->host copy");
            host_valid_ = true;
        }
    }
    void release()
    {
        if (dev_) clv_free(dev_);
        if (host_ && owns_host_) free(host_);
        host_ = nullptr;
        dev_ = nullptr;
        bytes_ = 0;
    }

    uint8_t *host_;
    uint8_t *dev_;
    uint64_t bytes_;
    bool host_valid_, dev_valid_, owns_host_;
};

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

#endif
