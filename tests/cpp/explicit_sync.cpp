// explicit_sync.cpp -- the containers built with -DCLOVER_HIP_EXPLICIT_SYNC (clover_device.h): no SIGSEGV handler, no mprotect, no helper
// thread -- usable inside a host that owns SIGSEGV itself.  Checks:
//   1. the host's own SIGSEGV disposition is untouched after the containers have been used (and its handler is never entered);
//   2. the reference's README example and a quantize / mvm / scaleAndAdd / threshold sequence give the answers of the default build
//      (the default build's answers are pinned elsewhere; here: device results == scalar host twins, bit for bit);
//   3. the documented rule: a pointer RE-TAKEN after a device operation shows its result; a write through a freshly taken pointer
//      reaches the next device operation; toDevice() / toHost() move the bytes at the caller's moment;
//   4. the host block is a plain allocation (no memory file): Mirror::double_mapped() is false.
// The same source compiles in the default (page-tracked) build: there the kept-pointer checks of step 3 hold without re-taking.
#include <signal.h>

#include <cmath>
#include <cstdio>
#include <cstring>

#include "CloverMatrix4.h"
#include "CloverVector4.h"
#include "CloverVector8.h"

static int failures = 0;
static volatile sig_atomic_t host_handler_entered = 0;
static void host_segv(int) { host_handler_entered = 1; }

static void expect(bool ok, const char *what)
{
    if (!ok) { std::printf("FAILED %s\n", what); failures++; }
}

int main()
{
    int ndev = 0;
    if (clv_device_count(&ndev) != CLV_OK || ndev == 0) { std::printf("no_device\n"); return 0; }
    // a host with its own SIGSEGV logic
    struct sigaction mine, seen;
    memset(&mine, 0, sizeof mine);
    mine.sa_handler = host_segv;
    sigaction(SIGSEGV, &mine, nullptr);

    const uint64_t n = 1024, M = 256;
    CloverVector32 a(n), b(n), back(n);
    for (uint64_t i = 0; i < n; i++) { a.set(i, 1.0f); b.set(i, 2.0f); }
    CloverVector4 qa(n), qb(n);
    qa.quantize(a);
    qb.quantize(b);
    expect(qa.dot(qb) == 2.0f * n, "README example: dot of ones and twos");
    expect(std::fabs(qa.dot_scalar(qb) - 2.0f * n) <= 0.02f, "README example: dot_scalar");

    // rule, read side: the pointer taken AFTER the device operation shows its result
    const int8_t *d = qa.getData();
    expect((uint8_t)d[0] == 0x77 && (uint8_t)d[n / 2 - 1] == 0x77 && qa.getScales()[0] == 1.0f && qb.getScales()[3] == 2.0f, "getData after quantize");
    // rule, write side: a write through a freshly taken pointer reaches the next device operation
    qa.getData()[0] = (int8_t)0x17;                     // element 0: 7 -> 1, i.e. 1.0 -> 1/7: the dot loses 2 - 2/7
    const float want1 = 2.0f * n - 2.0f + 2.0f / 7.0f;
    expect(std::fabs(qa.dot(qb) - want1) <= 1e-3f && std::fabs(qa.dot_scalar(qb) - want1) <= 0.02f, "write through getData reaches the next kernel");
    qa.getScales()[0] = 2.0f;                           // block 0 doubled: its 63 ones and the 1/7 count twice
    const float want2 = want1 + 2.0f * 63.0f + 2.0f / 7.0f;
    expect(std::fabs(qa.dot(qb) - want2) <= 1e-3f && std::fabs(qa.dot_scalar(qb) - want2) <= 0.02f, "write through getScales reaches the next kernel");
    // accessors synchronise by themselves
    qa.setBits(0, 7);
    qa.getScales()[0] = 1.0f;
    expect(qa.dot(qb) == 2.0f * n && qa.getBits(0) == 7 && qa.get(5) == 1.0f, "accessors");

    // a device -> device chain with explicit moves around it: quantize, mvm, scaleAndAdd, threshold; results == scalar host twins
    CloverMatrix32 A(M, n);
    A.setRandomInteger(10, 5);
    CloverVector32 x(n);
    x.setRandomInteger(10, 6);
    CloverMatrix4 qA(M, n), qS(M, n);
    qA.quantize(A);
    qS.quantize_scalar(A);
    expect(memcmp(qA.getData(), qS.getData(), qA.getBytes()) == 0, "matrix quantize == quantize_scalar");
    CloverVector4 qx(n), r(M), rs(M), acc(M), accs(M);
    qx.quantize(x);
    qA.toDevice();
    qx.toDevice();
    qA.mvm(qx, r);
    qA.mvm_scalar(qx, rs);
    r.toHost();
    expect(memcmp(r.getData(), rs.getData(), r.getBytes()) == 0, "mvm == mvm_scalar");
    r.scaleAndAdd(rs, 0.5f, acc);
    r.scaleAndAdd_scalar(rs, 0.5f, accs);
    expect(memcmp(acc.getData(), accs.getData(), acc.getBytes()) == 0, "scaleAndAdd == scaleAndAdd_scalar");
    acc.threshold(32);
    uint64_t nz = 0;
    for (uint64_t i = 0; i < M; i++) nz += acc.getBits(i) != 0;
    expect(nz > 0 && nz <= 32, "threshold keeps at most k");
    qa.restore(back);
    expect(back.getData()[7] == 1.0f && back.get(n - 1) == 1.0f, "restore, pointer taken after it");
    CloverVector8 e(n), es(n);
    e.quantize(x);
    es.quantize_scalar(x);
    expect(memcmp(e.getData(), es.getData(), e.getBytes()) == 0, "8-bit quantize == quantize_scalar");

    // the host's SIGSEGV disposition was never replaced and its handler never ran
    sigaction(SIGSEGV, nullptr, &seen);
#if defined(CLOVER_HIP_NO_PAGE_TRACKING)
    expect(!(seen.sa_flags & SA_SIGINFO) && seen.sa_handler == host_segv, "SIGSEGV disposition untouched");
    expect(host_handler_entered == 0, "host handler never entered");
    std::printf("mode=explicit\n");
#else
    expect((seen.sa_flags & SA_SIGINFO) != 0, "default build: the tracking handler is installed (chained in front of the host's)");
    std::printf("mode=tracked\n");
#endif
    std::printf(failures ? "explicit_sync FAILED (%d)\n" : "explicit_sync ok\n", failures);
    return failures ? 1 : 0;
}
