/*
 * ref_xorshift.cpp -- harness around the REFERENCE's own generator (TEST INFRASTRUCTURE ONLY).
 *
 * The only part of the reference that builds in this image without stand-ins is
 * /root/reference/include/simdxorshift128plus.h: it includes nothing but <stdint.h> and <x86intrin.h>.
 * (CloverVector4.h / CloverMatrix4.h pull in CloverBase.h, whose lines 35-36 include Intel ipp.h / mkl.h.)
 * This file contains no reference code: it #includes that header from where it lies (the Makefile passes
 * -I$(REFERENCE)/include) and wraps its three entry points
 *     avx_xorshift128plus_init   simdxorshift128plus.h:81-92
 *     avx_xorshift128plus        simdxorshift128plus.h:97-109
 *     avx_xorshift128plus_jump   simdxorshift128plus.h:115-127
 * behind a C ABI (oracle/_ref/libxorshift_ref.so, which travels to the GPU box like any built .so), and -- built
 * with -DREF_XS_MAIN -- prints the golden fixture tests/golden/xorshift_ref.json.
 *
 * State convention everywhere in this repository: s0[4] = random_key1 (part1), s1[4] = random_key2 (part2),
 * one uint64 per AVX lane (CloverRandom.h:90-94).
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "simdxorshift128plus.h"

static inline void load_state(const uint64_t s0[4], const uint64_t s1[4], __m256i &p1, __m256i &p2) {
    p1 = _mm256_loadu_si256((const __m256i *)s0);
    p2 = _mm256_loadu_si256((const __m256i *)s1);
}

static inline void store_state(uint64_t s0[4], uint64_t s1[4], __m256i p1, __m256i p2) {
    _mm256_storeu_si256((__m256i *)s0, p1);
    _mm256_storeu_si256((__m256i *)s1, p2);
}

extern "C" {

void ref_xs_init(uint64_t key1, uint64_t key2, uint64_t s0[4], uint64_t s1[4]) {
    __m256i p1, p2;
    avx_xorshift128plus_init(key1, key2, p1, p2);
    store_state(s0, s1, p1, p2);
}

/* `count` consecutive draws; W receives 8 uint32 per draw in memory order of the returned __m256i */
void ref_xs_draw(uint64_t s0[4], uint64_t s1[4], uint64_t count, uint32_t *W) {
    __m256i p1, p2;
    load_state(s0, s1, p1, p2);
    for (uint64_t i = 0; i < count; ++i) {
        __m256i r = avx_xorshift128plus(p1, p2);
        _mm256_storeu_si256((__m256i *)(W + 8 * i), r);
    }
    store_state(s0, s1, p1, p2);
}

/* `count` draws, nothing stored: xor-fold and wrapping sum of all 64-bit lane outputs (long-stream checks) */
void ref_xs_digest(uint64_t s0[4], uint64_t s1[4], uint64_t count, uint64_t *xor_fold, uint64_t *sum) {
    __m256i p1, p2, x = _mm256_setzero_si256(), a = _mm256_setzero_si256();
    load_state(s0, s1, p1, p2);
    for (uint64_t i = 0; i < count; ++i) {
        __m256i r = avx_xorshift128plus(p1, p2);
        x = _mm256_xor_si256(x, r);
        a = _mm256_add_epi64(a, r);
    }
    store_state(s0, s1, p1, p2);
    uint64_t xv[4], av[4];
    _mm256_storeu_si256((__m256i *)xv, x);
    _mm256_storeu_si256((__m256i *)av, a);
    *xor_fold = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
    *sum = av[0] + av[1] + av[2] + av[3];
}

void ref_xs_jump(uint64_t s0[4], uint64_t s1[4]) {
    __m256i p1, p2;
    load_state(s0, s1, p1, p2);
    avx_xorshift128plus_jump(p1, p2);
    store_state(s0, s1, p1, p2);
}

}  /* extern "C" */

#ifdef REF_XS_MAIN
static void put_u64s(const char *name, const uint64_t v[4], const char *tail) {
    printf("    \"%s\": [\"0x%016llx\", \"0x%016llx\", \"0x%016llx\", \"0x%016llx\"]%s\n", name,
           (unsigned long long)v[0], (unsigned long long)v[1], (unsigned long long)v[2], (unsigned long long)v[3], tail);
}

int main() {
    /* (12345, 67890): the survey's probe keys; the second pair is the reference's own deterministic seed
       (test/random/00_random.cpp:42) */
    static const uint64_t seeds[2][2] = {{12345ULL, 67890ULL}, {445560390295639063ULL, 2935984234003016713ULL}};
    enum { DRAWS = 256, LONG = 1 << 20 };
    static uint32_t W[8 * DRAWS];
    printf("{\n  \"generator\": \"oracle/ref_xorshift.cpp over /root/reference/include/simdxorshift128plus.h (make -C oracle ref-fixtures)\",\n");
    printf("  \"draws\": %d,\n  \"long_draws\": %d,\n  \"streams\": [\n", (int)DRAWS, (int)LONG);
    for (int k = 0; k < 2; ++k) {
        uint64_t s0[4], s1[4], xf, sm;
        ref_xs_init(seeds[k][0], seeds[k][1], s0, s1);
        printf("   {\n    \"key1\": \"%llu\", \"key2\": \"%llu\",\n", (unsigned long long)seeds[k][0], (unsigned long long)seeds[k][1]);
        put_u64s("init_s0", s0, ",");
        put_u64s("init_s1", s1, ",");
        ref_xs_draw(s0, s1, DRAWS, W);
        printf("    \"draw_words_hex\": \"");
        for (int i = 0; i < 8 * DRAWS; ++i) printf("%08x", W[i]);
        printf("\",\n");
        put_u64s("after_draws_s0", s0, ",");
        put_u64s("after_draws_s1", s1, ",");
        ref_xs_digest(s0, s1, LONG, &xf, &sm);
        printf("    \"long_xor_fold\": \"0x%016llx\", \"long_sum\": \"0x%016llx\",\n", (unsigned long long)xf, (unsigned long long)sm);
        put_u64s("after_long_s0", s0, ",");
        put_u64s("after_long_s1", s1, ",");
        ref_xs_jump(s0, s1);
        put_u64s("after_jump_s0", s0, ",");
        put_u64s("after_jump_s1", s1, "");
        printf("   }%s\n", k == 0 ? "," : "");
    }
    printf("  ]\n}\n");
    return 0;
}
#endif
