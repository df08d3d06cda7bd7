// mirror_threads.cpp -- include/clover_device.h under threads, against the host-memory fake of the ABI (tests/cpp/fake_clv.c;
// the same scenarios run against the real library in tests/cpp/pointer_threads.cpp on the GPU box).
//
// The reference hands out raw pointers that any number of host threads may read at once (CloverVector4.h:229-237).  With a host block
// and an HBM mirror that needs: one resolution per state change however many threads fault on the block, no window in which a reader
// sees half of a device -> host copy, no host write lost to a racing upload, a block table that can change while faults are being
// looked up, and a destructor that waits for a fault in flight.  Prints "mirror threads ok" or the first failures.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "clover_device.h"

extern "C" int fake_copies_d2h, fake_copies_h2d, fake_d2h_delay_us;
using clover_hip::Mirror;

static std::atomic<int> failures(0);
#define EXPECT(cond) do { if (!(cond)) { if (failures.fetch_add(1) < 10) std::printf("FAILED line %d: %s\n", __LINE__, #cond); } } while (0)

// a reusable barrier (C++11 has none)
struct Barrier {
    std::atomic<int> count, generation;
    const int n;
    explicit Barrier(int n_) : count(0), generation(0), n(n_) {}
    void wait()
    {
        const int g = generation.load();
        if (count.fetch_add(1) + 1 == n) { count.store(0); generation.fetch_add(1); }
        else while (generation.load() == g) std::this_thread::yield();
    }
};

static const int T = 4;

int main()
{
    bool strict = true;         // "no reader ever sees half a copy" needs the double mapping (memfd_create); without it the pages open before the copy
    {
        Mirror probe;
        probe.allocate(4096);
        strict = probe.double_mapped();
        if (!strict) std::printf("note: memfd_create unavailable -- single mapping, torn-read checks relaxed\n");
    }
    // 1. four threads read one DEVICE_DIRTY block through a kept pointer at the same moment: ONE copy back per round, and nobody
    //    sees a byte of the previous round (the copy is stretched: first half, pause, second half)
    {
        const uint64_t bytes = 1 << 20;
        Mirror m;
        m.allocate(bytes);
        const uint8_t *p = m.host_ptr();
        fake_d2h_delay_us = 300;
        const int rounds = 60;
        Barrier bar(T + 1);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int r = 0; r < rounds; r++) {
                    bar.wait();                                        // the "kernel" of this round has written the device copy
                    const uint8_t want = (uint8_t)(r + 1);
                    uint64_t bad = 0;
                    // every thread starts somewhere else and walks the whole block
                    for (uint64_t i = 0; i < bytes; i += 64) bad += p[(i + (uint64_t)t * (bytes / T)) % bytes] != want;
                    EXPECT(bad == 0 || !strict);
                    bar.wait();
                }
            });
        for (int r = 0; r < rounds; r++) {
            const int before = fake_copies_d2h;
            memset(m.dev_wo(), r + 1, bytes);
            m.commit();
            bar.wait();
            bar.wait();
            EXPECT(fake_copies_d2h == before + 1);                     // resolved once, not once per thread
            EXPECT(m.state() == Mirror::SHARED);
        }
        for (auto &x : th) x.join();
        fake_d2h_delay_us = 0;
    }
    // 2. four threads WRITE into one SHARED block at the same moment (each its own quarter): every write is in the next upload
    {
        const uint64_t bytes = 1 << 16;
        Mirror m;
        m.allocate(bytes);
        uint8_t *p = m.host_ptr();
        memset(p, 0, bytes);
        const int rounds = 200;
        Barrier bar(T + 1);
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (int r = 0; r < rounds; r++) {
                    bar.wait();
                    for (uint64_t i = (uint64_t)t * (bytes / T); i < (uint64_t)(t + 1) * (bytes / T); i += 512) p[i] = (uint8_t)(r + t + 1);
                    bar.wait();
                }
            });
        for (int r = 0; r < rounds; r++) {
            m.dev_ro();                                                // SHARED: the block is read-only now
            EXPECT(m.state() == Mirror::SHARED);
            bar.wait();
            bar.wait();
            EXPECT(m.state() == Mirror::HOST_DIRTY);
            const uint8_t *d = m.dev_ro();
            uint64_t bad = 0;
            for (int t = 0; t < T; t++)
                for (uint64_t i = (uint64_t)t * (bytes / T); i < (uint64_t)(t + 1) * (bytes / T); i += 512) bad += d[i] != (uint8_t)(r + t + 1);
            EXPECT(bad == 0);
        }
        for (auto &x : th) x.join();
    }
    // 3. a writer racing uploads: thread W keeps counting in place through a kept pointer while the main thread uploads over and over;
    //    whatever W wrote last must be on the device after the final upload (a write is either inside an upload or faults and
    //    marks the block dirty again -- never dropped)
    {
        Mirror m;
        m.allocate(8192);
        volatile uint32_t *p = reinterpret_cast<volatile uint32_t *>(m.host_ptr());
        p[0] = 0; p[1024] = 0;
        std::atomic<bool> stop(false);
        std::thread w([&] { while (!stop.load()) { p[0] = p[0] + 1; p[1024] = p[1024] + 1; } });
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(300);
        int uploads = 0;
        while (std::chrono::steady_clock::now() < t_end) { m.dev_ro(); uploads++; }
        stop.store(true);
        w.join();
        const uint32_t *d = reinterpret_cast<const uint32_t *>(m.dev_ro());
        EXPECT(d[0] == p[0] && d[1024] == p[1024] && d[0] == d[1024] && d[0] > 100 && uploads > 100);
    }
    // 4. the table of tracked blocks changes (one thread creates and destroys blocks of assorted sizes) while three threads keep
    //    faulting on blocks of their own
    {
        std::atomic<bool> stop(false);
        std::thread churn([&] {
            uint64_t k = 0;
            while (!stop.load()) {
                Mirror a[7];
                for (int i = 0; i < 7; i++) { a[i].allocate(4096 * (1 + (k + i) % 5)); a[i].host_ptr()[0] = (uint8_t)i; a[i].dev_ro(); }
                for (int i = 0; i < 7; i++) EXPECT(a[i].host_ptr()[0] == (uint8_t)i);
                k++;
            }
        });
        std::vector<std::thread> th;
        for (int t = 0; t < 3; t++)
            th.emplace_back([&, t] {
                Mirror m;
                m.allocate(20000);
                uint8_t *p = m.host_ptr();
                for (int r = 0; r < 3000; r++) {
                    memset(m.dev_wo(), (r + t) & 0xFF, 20000);         // DEVICE_DIRTY
                    EXPECT(p[19999] == (uint8_t)((r + t) & 0xFF));     // read fault -> SHARED
                    p[5] = 1;                                          // write fault -> HOST_DIRTY
                    EXPECT(m.dev_ro()[5] == 1);
                }
            });
        for (auto &x : th) x.join();
        stop.store(true);
        churn.join();
    }
    // 5. a destructor waits for a fault in flight: the handler pins the block between lookup and resolution (simulated here by
    //    holding a pin), and ~Mirror must not free the block before the pin is dropped
    {
        Mirror *m = new Mirror;
        m->allocate(4096);
        m->pin();
        std::atomic<bool> destroyed(false);
        std::thread d([&] { delete m; destroyed.store(true); });
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        EXPECT(!destroyed.load());
        m->unpin();
        d.join();
        EXPECT(destroyed.load());
    }
    // 6. a read fault and a method racing on one block: thread R reads through the kept pointer while the main thread keeps handing the
    //    block to a "kernel" that writes an increasing version number into every word and then pulls it back (or R's fault does: both go
    //    through the lock).  A word near the end is read AFTER a word in the first half: it can never hold an older version -- which is
    //    exactly what a reader that got in during the (stretched) copy would see
    {
        const uint64_t words = 1 << 14;
        Mirror m;
        m.allocate(words * 4);
        const volatile uint32_t *p = reinterpret_cast<const volatile uint32_t *>(m.host_ptr());
        uint32_t *dev = reinterpret_cast<uint32_t *>(m.dev_wo());         // the device block (its address does not change)
        // the "kernel" has finished before the block is handed over: a host thread reading an object WHILE a kernel writes it is a data
        // race in the caller's program (as it would be in the reference), not something the mirror can order
        auto fill = [&](uint32_t v) { for (uint64_t i = 0; i < words; i++) dev[i] = v; m.dev_wo(); };
        fill(1);
        fake_d2h_delay_us = 50;
        std::atomic<bool> stop(false);
        std::thread r([&] {
            uint32_t last = 0;
            while (!stop.load()) {
                const uint32_t b = p[words / 2 - 1];
                const uint32_t c = p[words - 1];
                EXPECT((c >= b && b >= last) || !strict);
                last = b;
            }
        });
        for (uint32_t v = 2; v < 600; v++) {
            if (v & 1) m.host_ro();                                                   // the method pulls (racing the reader's fault) ...
            else while (m.state() == Mirror::DEVICE_DIRTY) std::this_thread::yield();  // ... or the reader's fault does
            fill(v);
        }
        stop.store(true);
        r.join();
        fake_d2h_delay_us = 0;
    }
    std::printf(failures.load() ? "mirror threads FAILED\n" : "mirror threads ok\n");
    return failures.load() ? 1 : 0;
}
