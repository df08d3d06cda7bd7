// threshold_min_heap.cpp -- CloverVector4 / CloverVector8 ::threshold_min_heap(idx_t *, k) (CloverVector4.h:1929-1970, CloverVector8.h:1696-1737)
// through the headers: the heap the caller gets back must be the reference's, entry for entry.  The reference's method is restated here on
// the host with the real std::make_heap (the walk of CloverVector4.h:1933-1969 over getAbs / getBits) and compared with the device's.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "CloverVector32.h"
#include "CloverVector4.h"
#include "CloverVector8.h"

typedef clover_hip::idx_t idx_t;
static bool gt_idx_t(const idx_t &a, const idx_t &b) { return (a.value > b.value) || std::isnan(a.value); }
static void min_heapify(idx_t *heap, uint32_t pos, const uint32_t k)
{
    uint32_t smallest = pos;
    while (true) {
        const uint32_t l = pos * 2 + 1, r = pos * 2 + 2;
        if (l < k && heap[l].value < heap[smallest].value) smallest = l;
        if (r < k && heap[r].value < heap[smallest].value) smallest = r;
        if (smallest == pos) break;
        std::swap(heap[pos], heap[smallest]);
        pos = smallest;
    }
}

template <class QVector>
static int run(const char *name, uint64_t n, uint64_t k, unsigned seed)
{
    CloverVector32 x(n);
    srand(seed);
    for (uint64_t i = 0; i < n; i++) x.set(i, (float)((rand() % 2001) - 1000) * 0.01f * (1.0f + (float)((i / 64) % 5)));
    QVector q(n), ref(n);
    q.quantize(x);
    ref.quantize(x);
    // the reference's walk on the host copy of `ref`
    std::vector<idx_t> want(k);
    std::vector<int8_t> kept_bits(n, 0);
    for (uint64_t i = 0; i < k; i++) { want[i].value = ref.getAbs(i); want[i].bits.i = ref.getBits(i); want[i].idx = i; }
    std::make_heap(want.begin(), want.end(), gt_idx_t);
    for (uint64_t i = k; i < n; i++) {
        const float v = ref.getAbs(i);
        if (v > want[0].value) { want[0].value = v; want[0].idx = i; want[0].bits.i = ref.getBits(i); min_heapify(want.data(), 0, (uint32_t)k); }
    }
    for (uint64_t i = 0; i < k; i++) kept_bits[want[i].idx] = (int8_t)want[i].bits.i;
    std::vector<idx_t> got(k);
    q.threshold_min_heap(got.data(), k);
    int bad = 0;
    for (uint64_t i = 0; i < k; i++)
        if (got[i].value != want[i].value || got[i].idx != want[i].idx || got[i].bits.i != want[i].bits.i) bad++;
    for (uint64_t i = 0; i < n; i++)
        if (q.getBits(i) != kept_bits[i]) bad++;
    printf("%s n=%llu k=%llu mismatches=%d\n", name, (unsigned long long)n, (unsigned long long)k, bad);
    return bad;
}

int main()
{
    int bad = 0;
    bad += run<CloverVector4>("v4", 1024, 100, 1);
    bad += run<CloverVector4>("v4", 8192, 2048, 2);
    bad += run<CloverVector4>("v4", 1000, 1, 3);
    bad += run<CloverVector4>("v4", 640, 640, 4);
    bad += run<CloverVector8>("v8", 1024, 100, 5);
    bad += run<CloverVector8>("v8", 4096, 1024, 6);
    CloverVector4 p(2048);
    std::vector<idx_t> heap(64);
    CloverVector32 z(2048);
    for (uint64_t i = 0; i < 2048; i++) z.set(i, (float)(i % 17) - 8.0f);
    p.quantize(z);
    p.threshold_min_heap_parallel(heap.data(), 64);
    int nz = 0;
    for (uint64_t i = 0; i < 2048; i++) nz += p.getBits(i) != 0;
    printf("parallel nonzero=%d\n", nz);
    bad += nz > 64;
    printf("threshold_min_heap %s\n", bad ? "FAILED" : "OK");
    return bad ? 1 : 0;
}
