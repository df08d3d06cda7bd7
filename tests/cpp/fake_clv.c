/* fake_clv.c -- a host-memory stand-in for the handful of C-ABI calls clover_hip::Mirror makes, so that the page-tracking
 * state machine of include/clover_device.h can be unit-tested on a machine without a GPU (tests/cpp/mirror_states.cpp).
 * TEST DOUBLE ONLY: "device memory" is malloc'ed host memory; nothing here is shipped or linked into the product. */
#define _POSIX_C_SOURCE 199309L
#include <stdlib.h>
#include <string.h>

#include "clover_hip.h"

#include <time.h>

int fake_copies_d2h = 0, fake_copies_h2d = 0;       /* updated atomically: tests/cpp/mirror_threads.cpp copies from several threads */
int fake_d2h_delay_us = 0;                          /* stretches a device -> host copy (a 1 GiB block takes tens of ms on PCIe) */

const char *clv_last_error(void) { return "fake"; }
int clv_malloc(void **ptr, uint64_t bytes) { *ptr = malloc(bytes ? bytes : 1); return *ptr ? CLV_OK : CLV_ERR_HIP; }
int clv_free(void *ptr) { free(ptr); return CLV_OK; }
int clv_memcpy_h2d(void *dst, const void *src, uint64_t bytes, void *stream)
{
    (void)stream;
    memcpy(dst, src, bytes);
    __atomic_fetch_add(&fake_copies_h2d, 1, __ATOMIC_SEQ_CST);
    return CLV_OK;
}
int clv_memcpy_d2h(void *dst, const void *src, uint64_t bytes, void *stream)
{
    (void)stream;
    /* first half, pause, second half: a reader that got in during the copy would see two versions */
    memcpy(dst, src, bytes / 2);
    if (fake_d2h_delay_us) {
        struct timespec ts;
        ts.tv_sec = 0;
        ts.tv_nsec = 1000L * fake_d2h_delay_us;
        nanosleep(&ts, NULL);
    }
    memcpy((char *)dst + bytes / 2, (const char *)src + bytes / 2, bytes - bytes / 2);
    __atomic_fetch_add(&fake_copies_d2h, 1, __ATOMIC_SEQ_CST);
    return CLV_OK;
}
int clv_stream_sync(void *stream) { (void)stream; return CLV_OK; }
int clv_host_alloc(void **ptr, uint64_t bytes) { *ptr = malloc(bytes ? bytes : 1); return CLV_OK; }
int clv_host_free(void *ptr) { free(ptr); return CLV_OK; }
int clv_rng_seed(uint64_t *s, uint64_t a, uint64_t b, void *st) { (void)s; (void)a; (void)b; (void)st; return CLV_OK; }
int clv_rng_set(uint64_t *s, const uint64_t a[4], const uint64_t b[4], void *st) { (void)s; (void)a; (void)b; (void)st; return CLV_OK; }
int clv_rng_get(const uint64_t *s, uint64_t a[4], uint64_t b[4], void *st) { (void)s; (void)a; (void)b; (void)st; return CLV_OK; }
int clv_rng_graph_mode(uint64_t *s, int on, void *st) { (void)s; (void)on; (void)st; return CLV_OK; }
