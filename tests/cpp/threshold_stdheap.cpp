// threshold_stdheap.cpp -- independent check of the oracle's hand-written std::make_heap emulation:
// the same K-entry min-heap walk as CloverVector4::threshold (CloverVector4.h:1913-1975), but using the
// real libstdc++ std::make_heap.  Reads "n k" then n magnitudes (float) from stdin, prints the kept indices.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

struct item { float value; uint64_t idx; };
static bool gt(const item &a, const item &b) { return (a.value > b.value) || std::isnan(a.value); }

static void min_heapify(std::vector<item> &h, uint32_t pos, uint32_t k)
{
    uint32_t smallest = pos;
    while (true) {
        const uint32_t l = pos * 2 + 1, r = pos * 2 + 2;
        if (l < k && h[l].value < h[smallest].value) smallest = l;
        if (r < k && h[r].value < h[smallest].value) smallest = r;
        if (smallest == pos) break;
        std::swap(h[pos], h[smallest]);
        pos = smallest;
    }
}

int main()
{
    unsigned long long n, k;
    if (scanf("%llu %llu", &n, &k) != 2) return 1;
    std::vector<float> v(n);
    for (auto &x : v) if (scanf("%f", &x) != 1) return 1;
    std::vector<item> h(k);
    for (uint64_t i = 0; i < k; i++) h[i] = {v[i], i};
    std::make_heap(h.begin(), h.end(), gt);
    for (uint64_t i = k; i < n; i++)
        if (v[i] > h[0].value) { h[0] = {v[i], i}; min_heapify(h, 0, (uint32_t)k); }
    std::vector<uint64_t> kept;
    for (auto &e : h) kept.push_back(e.idx);
    std::sort(kept.begin(), kept.end());
    for (auto i : kept) printf("%llu\n", (unsigned long long)i);
    return 0;
}
