/* fake_clv.c -- a host-memory stand-in for the handful of C-ABI calls clover_hip::Mirror makes, so that the page-tracking
 * state machine of include/clover_device.h can be unit-tested on a machine without a GPU (tests/cpp/mirror_states.cpp).
 * TEST DOUBLE ONLY: "device memory" is malloc'ed host memory; nothing here is shipped or linked into the product. */
#include <stdlib.h>
#include <string.h>

#include "clover_hip.h"

int fake_copies_d2h = 0, fake_copies_h2d = 0;

const char *clv_last_error(void) { return "fake"; }
int clv_malloc(void **ptr, uint64_t bytes) { *ptr = malloc(bytes ? bytes : 1); return *ptr ? CLV_OK : CLV_ERR_HIP; }
int clv_free(void *ptr) { free(ptr); return CLV_OK; }
int clv_memcpy_h2d(void *dst, const void *src, uint64_t bytes, void *stream) { (void)stream; memcpy(dst, src, bytes); fake_copies_h2d++; return CLV_OK; }
int clv_memcpy_d2h(void *dst, const void *src, uint64_t bytes, void *stream) { (void)stream; memcpy(dst, src, bytes); fake_copies_d2h++; return CLV_OK; }
int clv_stream_sync(void *stream) { (void)stream; return CLV_OK; }
int clv_host_alloc(void **ptr, uint64_t bytes) { *ptr = malloc(bytes ? bytes : 1); return CLV_OK; }
int clv_host_free(void *ptr) { free(ptr); return CLV_OK; }
int clv_rng_seed(uint64_t *s, uint64_t a, uint64_t b, void *st) { (void)s; (void)a; (void)b; (void)st; return CLV_OK; }
int clv_rng_set(uint64_t *s, const uint64_t a[4], const uint64_t b[4], void *st) { (void)s; (void)a; (void)b; (void)st; return CLV_OK; }
int clv_rng_get(const uint64_t *s, uint64_t a[4], uint64_t b[4], void *st) { (void)s; (void)a; (void)b; (void)st; return CLV_OK; }
