// mirror_states.cpp -- the host/device coherence state machine of include/clover_device.h, exercised without a GPU
// (linked against tests/cpp/fake_clv.c, where "device memory" is plain host memory and a "kernel" is a memset/memcpy).
#include <cstdio>
#include <cstring>

#include "clover_device.h"

extern "C" int fake_copies_d2h, fake_copies_h2d;
using clover_hip::Mirror;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAILED line %d: %s\n", __LINE__, #cond); failures++; } } while (0)

int main()
{
    Mirror m;
    m.allocate(10000);                                  // 3 pages
    uint8_t *p = m.host_ptr();                          // "getData()", kept for the whole test
    EXPECT(m.state() == Mirror::HOST_DIRTY);
    memset(p, 1, 10000);                                // plain host writes: no fault, no copy
    EXPECT(fake_copies_d2h == 0 && fake_copies_h2d == 0);

    const uint8_t *d = m.dev_ro();                      // device read: upload, both sides current
    EXPECT(m.state() == Mirror::SHARED && fake_copies_h2d == 1 && d[9999] == 1);
    EXPECT(p[5] == 1 && m.state() == Mirror::SHARED);   // host reads do not disturb SHARED
    m.dev_ro();
    EXPECT(fake_copies_h2d == 1);                       // no second upload

    p[5] = 7;                                           // host WRITE through the kept pointer: fault -> HOST_DIRTY
    EXPECT(m.state() == Mirror::HOST_DIRTY && p[5] == 7);
    d = m.dev_ro();
    EXPECT(fake_copies_h2d == 2 && d[5] == 7);          // the next device read sees it

    uint8_t *w = m.dev_wo();                            // a "kernel" overwrites the device copy
    memset(w, 9, 10000);
    m.commit();
    EXPECT(m.state() == Mirror::DEVICE_DIRTY && fake_copies_d2h == 0);
    EXPECT(p[9000] == 9);                               // host READ through the kept pointer: fault -> copy back -> SHARED
    EXPECT(m.state() == Mirror::SHARED && fake_copies_d2h == 1);
    EXPECT(p[0] == 9 && fake_copies_d2h == 1);          // only one copy back

    w = m.dev_rw();                                     // in-place device update
    w[1] = 3;
    EXPECT(m.state() == Mirror::DEVICE_DIRTY);
    p[2] = 4;                                           // host WRITE into a DEVICE_DIRTY block: pull, then upgrade
    EXPECT(m.state() == Mirror::HOST_DIRTY && p[1] == 3 && p[2] == 4 && fake_copies_d2h == 2);
    d = m.dev_ro();
    EXPECT(d[1] == 3 && d[2] == 4);

    // accessor paths switch state without faulting
    m.dev_wo()[0] = 42;
    EXPECT(m.host_ro()[0] == 42 && m.state() == Mirror::SHARED);
    m.host_rw()[0] = 43;
    EXPECT(m.state() == Mirror::HOST_DIRTY && m.dev_ro()[0] == 43);

    // a view over caller memory is written through (commit) and re-uploaded on every device read
    uint8_t user[256];
    memset(user, 5, sizeof(user));
    Mirror v;
    v.adopt(user, sizeof(user));
    const int h0 = fake_copies_h2d;
    EXPECT(v.dev_ro()[10] == 5);
    user[10] = 6;
    EXPECT(v.dev_ro()[10] == 6 && fake_copies_h2d == h0 + 2);
    memset(v.dev_wo(), 8, sizeof(user));
    EXPECT(user[0] == 5);                               // not yet: the launch is "in flight"
    v.commit();
    EXPECT(user[0] == 8 && user[255] == 8);
    // ... also when the caller memory is another mirror's protected block
    Mirror owner;
    owner.allocate(4096);
    uint8_t *op = owner.host_ptr();
    memset(op, 1, 4096);
    owner.dev_ro();                                     // SHARED: block is read-only now
    Mirror alias;
    alias.adopt(op, 4096);
    memset(alias.dev_wo(), 2, 4096);
    alias.commit();                                     // writes into owner's block: must un-protect it first
    EXPECT(op[100] == 2 && owner.state() == Mirror::HOST_DIRTY && owner.dev_ro()[100] == 2);
    memset(owner.dev_wo(), 3, 4096);                    // owner DEVICE_DIRTY (PROT_NONE)
    EXPECT(alias.dev_ro()[7] == 3);                     // the alias reads through the protection: owner pulled first

    // many objects: the handler finds the right one; destruction un-protects before free
    {
        Mirror many[32];
        uint8_t *ptrs[32];
        for (int i = 0; i < 32; i++) { many[i].allocate(5000); ptrs[i] = many[i].host_ptr(); memset(many[i].dev_wo(), i, 5000); }
        for (int i = 31; i >= 0; i--) EXPECT(ptrs[i][4999] == i);
    }
    clover_hip::ResultSlot &slot = clover_hip::result_slot();
    *slot.device() = 2.5f;
    EXPECT(slot.fetch() == 2.5f);
    std::printf(failures ? "mirror FAILED\n" : "mirror ok\n");
    return failures ? 1 : 0;
}
