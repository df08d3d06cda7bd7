// pointer_coherence.cpp -- the reference's raw-pointer contract on top of two copies (host block + HBM mirror).
//
// In the reference getData()/getScales() alias the one and only copy (CloverVector4.h:229-237), a view constructed from two
// pointers aliases the caller's memory (:114-119), and users keep such pointers for as long as they like.  Here the cases
// that go wrong with a naive host/device mirror: a pointer kept ACROSS a device operation must read the operation's
// result, a write through it must reach the next device operation, and views must write through in both directions.
// Also: the per-call cost of dot() through the headers at the README size (BASELINE config 1, n = 128).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "CloverMatrix4.h"
#include "CloverVector32.h"
#include "CloverVector4.h"
#include "CloverVector8.h"

static int failures = 0;
#define EXPECT(cond)                                                      \
    do {                                                                  \
        if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); failures++; } \
    } while (0)

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv)
{
    if (argc > 1 && !std::strcmp(argv[1], "crash")) {
        // a genuine wild access must still kill the process with SIGSEGV (the tracking handler chains to the default action)
        CloverVector32 keep(128);
        volatile int *bad = reinterpret_cast<volatile int *>(16);
        std::printf("about to crash\n");
        std::fflush(stdout);
        *bad = 1;
        std::printf("not reached\n");
        return 0;
    }
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }

    const uint64_t n = 128;
    CloverVector32 a32(n), b32(n);
    float *pa = a32.getData();                       // kept for the whole test
    for (uint64_t i = 0; i < n; i++) { pa[i] = 1.0f; b32.set(i, 2.0f); }
    CloverVector4 a4(n), b4(n);
    int8_t *qa = a4.getData();                       // taken BEFORE the device writes the vector
    float *sa = a4.getScales();
    a4.quantize(a32);
    b4.quantize(b32);
    // 1. read through pointers kept across a device operation: README answer (bytes 0x77, scale 1.0)
    EXPECT((uint8_t)qa[0] == 0x77 && (uint8_t)qa[63] == 0x77);
    EXPECT(sa[0] == 1.0f && sa[1] == 1.0f);
    EXPECT(bits(a4.dot(b4)) == 0x43800000u);         // 256.0
    // 2. write through the kept pointer AFTER device operations: the next device operation must see it
    qa[0] = 0x00;                                    // elements 0 and 1 become 0: the dot loses 2 * (7*7/49 * 1 * 2) = 4
    EXPECT(bits(a4.dot(b4)) == 0x437C0000u);         // 252.0
    sa[1] = 3.0f;                                    // second block's scale 1 -> 3: its 128 become 384
    {
        CloverVector4 fresh(a4);                     // built from the host bytes: what the device must have seen
        const float d = a4.dot(b4);
        EXPECT(bits(d) == bits(fresh.dot(b4)) && std::fabs(d - 508.0f) < 1e-3f && std::fabs(a4.dot_scalar(b4) - 508.0f) < 1e-3f);
    }
    // ... and the other way round again: device result after host writes through the same pointers
    a4.quantize(a32);
    EXPECT((uint8_t)qa[0] == 0x77 && sa[1] == 1.0f);
    // 3. the fp32 side: a kept float* sees restore()'s result and feeds the next quantize
    CloverVector32 r32(n);
    float *pr = r32.getData();
    b4.restore(r32);
    EXPECT(pr[0] == 2.0f && pr[127] == 2.0f);
    pa[5] = -4.0f;                                   // a32 was uploaded for quantize: this write must invalidate that copy
    a4.quantize(a32);
    EXPECT(sa[0] == 4.0f && std::fabs(a4.get(5) + 4.0f) < 1e-6f && (uint8_t)qa[2] == 0x19);      // elements 4,5 -> trunc(1*7/4)=1, -7
    // 4. views over plain caller memory write through: quantize into the view, read the caller's arrays directly
    std::vector<int8_t> user_values(n / 2, 0x55);
    std::vector<float> user_scales(n / 64, -1.0f);
    {
        CloverVector4 view(n, user_values.data(), user_scales.data());
        view.quantize(b32);
        EXPECT((uint8_t)user_values[0] == 0x77 && (uint8_t)user_values[63] == 0x77 && user_scales[0] == 2.0f && user_scales[1] == 2.0f);
        user_scales[0] = 4.0f;                       // the caller changes its memory: the view's next device read must see it
        EXPECT(std::fabs(view.dot(b4) - 768.0f) < 1e-3f);    // block 0: 64 * 49 * (4*2/49) = 512, block 1: 256
    }
    // 5. a view over another container's pointers aliases that container
    {
        CloverVector4 alias(n, a4.getData(), a4.getScales());
        alias.quantize(b32);                         // writes "through" a4's storage
        EXPECT(sa[0] == 2.0f && std::fabs(a4.get(0) - 2.0f) < 1e-6f);
        EXPECT(bits(a4.dot(b4)) == bits(b4.dot(b4)));
    }
    // 6. 8-bit container, same contract
    CloverVector8 a8(n);
    int8_t *q8 = a8.getData();
    a8.quantize(b32);
    EXPECT(q8[0] == 127 && a8.getScales()[0] == 2.0f);
    // 7. mvm result into a vector whose pointer was taken before
    CloverMatrix32 A32(128, 128);
    for (uint64_t i = 0; i < 128; i++) for (uint64_t j = 0; j < 128; j++) A32.set(i, j, i == j ? 1.0f : 0.0f);
    CloverMatrix4 A4(128, 128);
    A4.quantize(A32);
    CloverVector4 y4(128);
    int8_t *qy = y4.getData();
    A4.mvm(b4, y4);                                  // identity * b = b (2.0 everywhere: nibble 7, scale 2)
    EXPECT((uint8_t)qy[0] == 0x77 && y4.getScales()[0] == 2.0f);
    // 8. C1 through the headers: per-call cost of dot() at n = 128 (no allocation per call any more)
    const int reps = 2000;
    float acc = 0;
    for (int i = 0; i < 50; i++) acc += b4.dot(b4);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) acc += b4.dot(b4);
    const auto t1 = std::chrono::steady_clock::now();
    const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
    std::printf("c1_header_dot_us=%.2f acc=%g\n", us, (double)acc);
    std::printf(failures ? "coherence FAILED\n" : "coherence ok\n");
    return failures ? 1 : 0;
}
