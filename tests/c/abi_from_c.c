/* abi_from_c.c -- the boundary is a C ABI: this file is compiled as C99 (gcc, not g++) against include/clover_hip.h alone.
 * With a GPU it quantizes the README vectors (a = 1, b = 2, n = 128), takes their dot in the reference's order and prints it;
 * without one it reports the status code and exits 0 (the test then only checks that it compiled, linked and ran). */
#include <stdio.h>
#include <stdlib.h>

#include "clover_hip.h"

int main(void)
{
    int count = 0;
    if (clv_device_count(&count) != CLV_OK || count < 1) {
        printf("no_device status_text=%s\n", clv_last_error());
        return 0;
    }
    enum { N = 128 };
    float ha[N], hb[N], result = 0.0f;
    void *a = NULL, *b = NULL, *qa = NULL, *qb = NULL, *sa = NULL, *sb = NULL, *out = NULL;
    for (int i = 0; i < N; i++) { ha[i] = 1.0f; hb[i] = 2.0f; }
    if (clv_malloc(&a, sizeof ha) || clv_malloc(&b, sizeof hb) || clv_malloc(&qa, N / 2) || clv_malloc(&qb, N / 2) ||
        clv_malloc(&sa, N / 64 * sizeof(float)) || clv_malloc(&sb, N / 64 * sizeof(float)) || clv_malloc(&out, sizeof(float))) {
        printf("alloc failed: %s\n", clv_last_error());
        return 1;
    }
    int rc = clv_memcpy_h2d(a, ha, sizeof ha, NULL);
    if (!rc) rc = clv_memcpy_h2d(b, hb, sizeof hb, NULL);
    if (!rc) rc = clv4_quantize((const float *)a, N, (int8_t *)qa, (float *)sa, NULL, NULL);
    if (!rc) rc = clv4_quantize((const float *)b, N, (int8_t *)qb, (float *)sb, NULL, NULL);
    if (!rc) rc = clv4_dot((const int8_t *)qa, (const float *)sa, (const int8_t *)qb, (const float *)sb, N, CLV_DOT_EXACT, (float *)out, NULL, NULL);
    if (!rc) rc = clv_memcpy_d2h(&result, out, sizeof result, NULL);
    if (rc) {
        printf("failed: %s\n", clv_last_error());
        return 1;
    }
    printf("dot=%.1f\n", result);      /* README.md: 256 */
    clv_free(a); clv_free(b); clv_free(qa); clv_free(qb); clv_free(sa); clv_free(sb); clv_free(out);
    return result == 256.0f ? 0 : 2;
}
