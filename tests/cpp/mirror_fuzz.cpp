// mirror_fuzz.cpp -- random sequences of container operations against a host-only model: device kernels (quantize, scaleAndAdd, restore,
// dot), host accessors (get / set / getBits / setBits), raw writes and reads through getData() / getScales() pointers KEPT from the start
// (the reference's contract: one copy, pointers stay good -- CloverVector4.h:229-237), copies, and views over caller memory, in any order.
// The model is plain arrays driven by the scalar twins of clover_scalar.h; after every step one random object is compared with it byte for
// byte THROUGH THE KEPT POINTERS, so a missed upload, a missed download or a stale mapping shows at the step that caused it.
// With -DCLOVER_HIP_EXPLICIT_SYNC the pointers are re-taken before each raw access instead (that build's rule).
//   g++ -std=c++11 -O2 -DCLOVER_STOCHASTIC_ROUNDING_DISABLED=1 [-DCLOVER_HIP_EXPLICIT_SYNC] -Iinclude tests/cpp/mirror_fuzz.cpp -lclover_hip ...
//   ./mirror_fuzz <seed> <steps>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "CloverVector32.h"
#include "CloverVector4.h"
#include "clover_scalar.h"

namespace sc = clover_hip::scalar;

static uint64_t rng_state;
static uint64_t rnd()
{
    rng_state += 0x9E3779B97F4A7C15ull;
    uint64_t r = rng_state;
    r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
    r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
    return r ^ (r >> 31);
}
static uint64_t below(uint64_t n) { return rnd() % n; }
static int8_t random_byte() { return (int8_t)((((int)below(15) - 7) << 4) | (((int)below(15) - 7) & 0xF)); }      // two nibbles in -7 .. 7

struct Obj {
    CloverVector4 *v;
    int8_t *q;             // kept pointers (tracked build)
    float *s;
    std::vector<int8_t> mq;      // the model
    std::vector<float> ms;
    std::vector<int8_t> own_q;   // caller memory when the object is a view
    std::vector<float> own_s;
};

int main(int argc, char **argv)
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }
    rng_state = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 300;
    const uint64_t n = 128 * (1 + below(24)), nb = n / 64;
    const int NOBJ = 4;
    std::vector<Obj> objs(NOBJ);
    for (int i = 0; i < NOBJ; i++) {
        Obj &o = objs[i];
        if (i == NOBJ - 1) {                                    // the last one is a view over caller memory (CloverVector4.h:114-119)
            o.own_q.assign(n / 2, 0);
            o.own_s.assign(nb, 1.0f);
            o.v = new CloverVector4(n, o.own_q.data(), o.own_s.data());
        } else {
            o.v = new CloverVector4(n);
            o.v->clear();
        }
        o.q = o.v->getData();
        o.s = o.v->getScales();
        o.mq.assign(n / 2, 0);
        o.ms.assign(nb, 1.0f);
        for (uint64_t b = 0; b < nb; b++) { o.s[b] = 1.0f; }
        for (uint64_t j = 0; j < n / 2; j++) o.q[j] = 0;
    }
    CloverVector32 x32(n);
    std::vector<float> mx(n);
#ifdef CLOVER_HIP_EXPLICIT_SYNC
#define PTRS(o) do { (o).q = (o).v->getData(); (o).s = (o).v->getScales(); } while (0)
#else
#define PTRS(o) do { } while (0)
#endif
    int failures = 0;
    for (int step = 0; step < steps && !failures; step++) {
        Obj &a = objs[below(NOBJ)], &b = objs[below(NOBJ)];
        const int op = (int)below(11);
        switch (op) {
        case 0: {                                              // quantize from a fresh fp32 vector (host-written, then uploaded)
            for (uint64_t i = 0; i < n; i++) { const float f = (float)((int)below(41) - 20) * 0.25f; x32.set(i, f); mx[i] = f; }
            a.v->quantize(x32);
            sc::quantize4(mx.data(), n, a.mq.data(), a.ms.data());
            break;
        }
        case 1: {                                              // device scaleAndAdd in place (a == b allowed)
            const float alpha = (float)((int)below(9) - 4) * 0.5f;
            a.v->scaleAndAdd(*b.v, alpha);
            sc::scale_and_add4(a.mq.data(), a.ms.data(), b.mq.data(), b.ms.data(), alpha, n, a.mq.data(), a.ms.data());
            break;
        }
        case 2: {                                              // raw writes through the kept value pointer
            PTRS(a);
            for (int r = 0; r < 5; r++) { const uint64_t j = below(n / 2); const int8_t v = random_byte(); a.q[j] = v; a.mq[j] = v; }
            break;
        }
        case 3: {                                              // raw write through the kept scale pointer
            PTRS(a);
            const uint64_t j = below(nb);
            const float v = 0.25f * (float)(1 + below(16));
            a.s[j] = v;
            a.ms[j] = v;
            break;
        }
        case 4: {                                              // setBits / getBits
            const uint64_t pos = below(n);
            const int8_t bits = (int8_t)((int)below(15) - 7);
            a.v->setBits(pos, bits);
            int8_t &byte = a.mq[pos >> 1];
            byte = (pos & 1) ? (int8_t)((byte & 0xF0) | (bits & 0xF)) : (int8_t)((bits << 4) | (byte & 0xF));
            if (a.v->getBits(pos) != bits) { std::printf("step %d: getBits after setBits\n", step); failures++; }
            break;
        }
        case 5: {                                              // get(i) == scale / 7 * nibble
            const uint64_t pos = below(n);
            const int8_t byte = a.mq[pos >> 1];
            const float want = (a.ms[pos >> 6] / 7.0f) * (float)((pos & 1) ? sc::nibble_lo(byte) : sc::nibble_hi(byte));
            const float got = a.v->get(pos);
            if (std::memcmp(&want, &got, 4)) { std::printf("step %d: get(%llu) = %g, model %g\n", step, (unsigned long long)pos, got, want); failures++; }
            break;
        }
        case 6: {                                              // restore on the device, read back through the fp32 accessor
            a.v->restore(x32);
            sc::restore4(a.mq.data(), a.ms.data(), n, mx.data());
            for (uint64_t i = 0; i < n; i += 1 + below(7)) {
                const float got = x32.get(i);
                if (std::memcmp(&got, &mx[i], 4)) { std::printf("step %d: restore element %llu\n", step, (unsigned long long)i); failures++; break; }
            }
            break;
        }
        case 7: {                                              // dot on the device vs the host twin of the same object
            const float d = a.v->dot(*b.v), h = a.v->dot_scalar(*b.v);
            double mag = 0;
            for (uint64_t bk = 0; bk < nb; bk++) mag += 64.0 * std::fabs((double)a.ms[bk] * b.ms[bk]);
            if (!(std::fabs((double)d - h) <= 1e-5 * mag + 1e-6)) { std::printf("step %d: dot %g vs host %g\n", step, d, h); failures++; }
            break;
        }
        case 8: {                                              // copy construction: from whichever side is current
            CloverVector4 c(*a.v);
            if (std::memcmp(c.getData(), a.mq.data(), n / 2) || std::memcmp(c.getScales(), a.ms.data(), nb * 4)) { std::printf("step %d: copy\n", step); failures++; }
            break;
        }
        case 9: {                                              // clear
            a.v->clear();
            std::fill(a.mq.begin(), a.mq.end(), 0);
            std::fill(a.ms.begin(), a.ms.end(), 1.0f);
            break;
        }
        default: {                                             // set(pos, value): the reference's per-element re-quantise against the stored scale
            const uint64_t pos = below(n);
            const float val = a.ms[pos >> 6] * (float)((int)below(15) - 7) / 7.0f;
            a.v->set(pos, val);
            PTRS(a);
            a.mq[pos >> 1] = a.q[pos >> 1];                    // the accessor is host code; only its coherence is under test here
            break;
        }
        }
        // after every step: one object against the model, through the kept pointers
        Obj &c = objs[below(NOBJ)];
        PTRS(c);
        if (std::memcmp(c.q, c.mq.data(), n / 2) || std::memcmp(c.s, c.ms.data(), nb * 4)) {
            std::printf("step %d (op %d): object differs from the model\n", step, op);
            failures++;
        }
        if (!c.own_q.empty() && (c.q != c.own_q.data() || c.s != c.own_s.data())) { std::printf("step %d: a view must alias the caller's memory\n", step); failures++; }
    }
    for (auto &o : objs) delete o.v;
    std::printf(failures ? "FAILED\n" : "ok n=%llu steps=%d\n", (unsigned long long)n, steps);
    return failures ? 1 : 0;
}
