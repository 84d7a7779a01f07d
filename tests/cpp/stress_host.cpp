// tests/cpp/stress_host.cpp - a native, many-threaded client of the public C ABI (include/hs_ed25519.h only), built to run
// under ThreadSanitizer and AddressSanitizer + UBSan against sanitizer builds of the host side of libhs_ed25519.so
// ("FFI must be thread-safe"; VERDICT r3 missing #4: 1,300 lines of pools, tickets, per-thread groups and rings
// had never been under a race detector).  Python cannot host those runtimes usefully, so the scenarios of
// tests/test_gpu_tickets.py, test_gpu_fuzz_apis.py, test_concurrent_host_calls_are_thread_safe, test_mixed_concurrent_callers
// and the ring tests are restated here in C++ and run AT ONCE for a few seconds:
//   sync      threads calling hs_ed25519_verify_multi / _shared on slices of every size class (1 ... 6,667 signatures)
//   tickets   threads keeping windows of tickets out in bursts (HS_ED25519_SUBMIT_MORE), collecting in another order
//   handoff   tickets submitted by one thread and waited on by another
//   ring      a four-slot round ring driven by one thread (zero-copy and graph forms)
//   keyset    key sets created, used by two threads at once (cached calls, cached tickets) and destroyed, repeatedly
//   sharded   hs_ed25519_verify_multi_sharded[_cached] over two virtual device slots, both key placements
//   device    device-pointer contexts created / used / destroyed beside everything else
//   escape    a thread holding every wide context as tickets makes a synchronous call (the pool's over-cap escape)
// Inputs are signed by the library itself (hs_ed25519_sign_batch), every 37th signature is then broken (one bit of s), and
// every verdict bitmap is compared with that known mask - a run is green only if no verdict was wrong and the sanitizer said
// nothing.   usage: stress_host [seconds = 4] [threads scale = 1]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/hs_ed25519.h"

static std::atomic<long> g_calls{0}, g_wrong{0}, g_errors{0};
static std::atomic<bool> g_stop{false};

#define REQUIRE(cond, what) do { if (!(cond)) { fprintf(stderr, "stress_host: %s failed (%s) at line %d\n", what, hs_ed25519_last_error(), __LINE__); g_errors++; } } while (0)

struct dataset {
    size_t n = 0;
    std::vector<uint8_t> dg, pk, sg, valid;      // per-tuple digests (32 n), keys, signatures; valid[i]
    uint8_t shared_dg[32];
    std::vector<uint8_t> sg_shared;               // signatures over shared_dg
};
static uint64_t splitmix(uint64_t &s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

static bool make_data(dataset &d, size_t n) {
    d.n = n; d.dg.resize(32 * n); d.pk.resize(32 * n); d.sg.resize(64 * n); d.sg_shared.resize(64 * n); d.valid.assign(n, 1);
    std::vector<uint8_t> seeds(32 * n), secrets(64 * n);
    uint64_t s = 20260922;
    for (size_t i = 0; i < 32 * n; i += 8) { uint64_t v = splitmix(s); memcpy(&seeds[i], &v, 8); v = splitmix(s); memcpy(&d.dg[i], &v, 8); }
    for (int i = 0; i < 32; i += 8) { uint64_t v = splitmix(s); memcpy(&d.shared_dg[i], &v, 8); }
    if (hs_ed25519_public_keys(seeds.data(), n, d.pk.data()) != 0) return false;
    for (size_t i = 0; i < n; i++) { memcpy(&secrets[64 * i], &seeds[32 * i], 32); memcpy(&secrets[64 * i + 32], &d.pk[32 * i], 32); }
    if (hs_ed25519_sign_batch(secrets.data(), d.dg.data(), 32, n, d.sg.data()) != 0) return false;
    if (hs_ed25519_sign_batch(secrets.data(), d.shared_dg, 0, n, d.sg_shared.data()) != 0) return false;
    for (size_t i = 5; i < n; i += 37) { d.sg[64 * i + 32] ^= 8; d.sg_shared[64 * i + 32] ^= 8; d.valid[i] = 0; }
    return true;
}
// bitmap of [lo, lo + n) against the known mask; verdict rc must agree too
static void check(const dataset &d, size_t lo, size_t n, int rc, const uint8_t *bm, const char *who) {
    g_calls++;
    if (rc < 0) { fprintf(stderr, "stress_host: %s: engine error %d (%s)\n", who, rc, hs_ed25519_last_error()); g_errors++; return; }
    bool all = true;
    for (size_t i = 0; i < n; i++) {
        const bool got = (bm[i >> 3] >> (i & 7)) & 1, want = d.valid[lo + i] != 0;
        all = all && want;
        if (got != want) { g_wrong++; fprintf(stderr, "stress_host: %s: wrong verdict at %zu (+%zu of %zu)\n", who, lo + i, i, n); return; }
    }
    if ((rc == 1) != all) { g_wrong++; fprintf(stderr, "stress_host: %s: call verdict %d but mask all-valid = %d\n", who, rc, (int)all); }
}
static const size_t kSizes[] = {1, 2, 63, 65, 700, 1024, 1025, 2000, 6667};

static void sync_worker(const dataset &d, unsigned id) {
    uint64_t s = 1000 + id;
    std::vector<uint8_t> bm(1024);
    while (!g_stop.load()) {
        const size_t n = kSizes[splitmix(s) % (sizeof kSizes / sizeof kSizes[0])], lo = splitmix(s) % (d.n - n);
        const bool shared = splitmix(s) & 1;
        const uint32_t mode = splitmix(s) % 3;        // BATCH / STRICT / BATCH_RLC: the same per-signature law on these inputs
        memset(bm.data(), 0, bm.size());
        const int rc = shared ? hs_ed25519_verify_shared(d.shared_dg, &d.pk[32 * lo], &d.sg_shared[64 * lo], n, mode, bm.data())
                              : hs_ed25519_verify_multi(&d.dg[32 * lo], &d.pk[32 * lo], &d.sg[64 * lo], n, mode, bm.data());
        check(d, lo, n, rc, bm.data(), "sync");
    }
}
struct out_ticket { hs_ed25519_ticket *t; size_t lo, n; };
static void ticket_worker(const dataset &d, unsigned id) {
    uint64_t s = 2000 + id;
    std::deque<out_ticket> win;
    std::vector<uint8_t> bm(1024);
    auto collect = [&](bool newest) {
        out_ticket o = newest ? win.back() : win.front();
        if (newest) win.pop_back(); else win.pop_front();
        memset(bm.data(), 0, bm.size());
        const int rc = hs_ed25519_ticket_wait(o.t, bm.data());
        check(d, o.lo, o.n, rc, bm.data(), "ticket");
    };
    while (!g_stop.load()) {
        const unsigned burst = 1 + splitmix(s) % 5;
        for (unsigned k = 0; k < burst; k++) {
            const size_t n = (splitmix(s) % 8) ? 1 + splitmix(s) % 1024 : 2048, lo = splitmix(s) % (d.n - n);
            hs_ed25519_ticket *t = nullptr;
            const int rc = hs_ed25519_submit(&t, nullptr, &d.dg[32 * lo], 32, &d.pk[32 * lo], &d.sg[64 * lo], n, splitmix(s) % 2, k + 1 < burst ? HS_ED25519_SUBMIT_MORE : 0);
            if (rc == 0) win.push_back({t, lo, n});
            else if (rc == HS_ED25519_E_BUSY) { if (!win.empty()) collect(false); else std::this_thread::yield(); }
            else { REQUIRE(false, "hs_ed25519_submit"); }
        }
        if (!win.empty() && splitmix(s) % 3 == 0) (void)hs_ed25519_ticket_ready(win.back().t);      // polling launches an open burst
        while (win.size() > 6) collect(splitmix(s) & 1);
    }
    while (!win.empty()) collect(false);
}
// tickets submitted here, waited on THERE
struct handoff { std::mutex mu; std::condition_variable cv; std::deque<out_ticket> q; bool done = false; };
static void handoff_producer(const dataset &d, handoff &h) {
    uint64_t s = 3000;
    while (!g_stop.load()) {
        const size_t n = 1 + splitmix(s) % 1024, lo = splitmix(s) % (d.n - n);
        hs_ed25519_ticket *t = nullptr;
        const int rc = hs_ed25519_submit(&t, nullptr, d.shared_dg, 0, &d.pk[32 * lo], &d.sg_shared[64 * lo], n, HS_ED25519_BATCH, (splitmix(s) & 1) ? HS_ED25519_SUBMIT_MORE : 0);
        if (rc == HS_ED25519_E_BUSY) { std::this_thread::yield(); continue; }
        if (rc != 0) { REQUIRE(false, "handoff submit"); continue; }
        std::unique_lock<std::mutex> lk(h.mu);
        h.q.push_back({t, lo, n});
        h.cv.notify_one();
        h.cv.wait(lk, [&] { return h.q.size() < 8 || g_stop.load(); });
    }
    std::lock_guard<std::mutex> lk(h.mu);
    h.done = true;
    h.cv.notify_all();
}
static void handoff_consumer(const dataset &d, handoff &h) {
    std::vector<uint8_t> bm(256);
    for (;;) {
        out_ticket o;
        {
            std::unique_lock<std::mutex> lk(h.mu);
            h.cv.wait(lk, [&] { return !h.q.empty() || h.done; });
            if (h.q.empty()) return;
            o = h.q.front(); h.q.pop_front();
            h.cv.notify_all();
        }
        memset(bm.data(), 0, bm.size());
        const int rc = hs_ed25519_ticket_wait(o.t, bm.data());
        check(d, o.lo, o.n, rc, bm.data(), "handoff");
    }
}
static void ring_worker(const dataset &d, uint32_t flags) {
    hs_ed25519_ring *ring = nullptr;
    const size_t cap = 3000;
    if (hs_ed25519_ring_create(&ring, cap, 4, nullptr, flags) != 0) { REQUIRE(false, "ring_create"); return; }
    uint64_t s = 4000 + flags;
    size_t lo[4] = {0, 0, 0, 0}, nn[4] = {0, 0, 0, 0};
    bool inflight[4] = {false, false, false, false};
    std::vector<uint8_t> bm(1024);
    unsigned slot = 0;
    while (!g_stop.load()) {
        if (inflight[slot]) {
            memset(bm.data(), 0, bm.size());
            const int rc = hs_ed25519_ring_wait(ring, slot, bm.data());
            check(d, lo[slot], nn[slot], rc, bm.data(), "ring");
            inflight[slot] = false;
        }
        const size_t n = 1 + splitmix(s) % cap, l = splitmix(s) % (d.n - n);
        uint8_t *pd, *pk, *ps;
        if (hs_ed25519_ring_begin(ring, slot, n, 0, &pd, &pk, &ps) != 0) { REQUIRE(false, "ring_begin"); break; }
        memcpy(pd, d.shared_dg, 32); memcpy(pk, &d.pk[32 * l], 32 * n); memcpy(ps, &d.sg_shared[64 * l], 64 * n);
        if (hs_ed25519_ring_submit(ring, slot, HS_ED25519_BATCH) != 0) { REQUIRE(false, "ring_submit"); break; }
        lo[slot] = l; nn[slot] = n; inflight[slot] = true;
        slot = (slot + 1) % 4;
    }
    for (unsigned k = 0; k < 4; k++) if (inflight[k]) { memset(bm.data(), 0, bm.size()); const int rc = hs_ed25519_ring_wait(ring, k, bm.data()); check(d, lo[k], nn[k], rc, bm.data(), "ring"); }
    hs_ed25519_ring_destroy(ring);
}
static void keyset_worker(const dataset &d) {
    uint64_t s = 5000;
    const size_t nk = 1536;
    while (!g_stop.load()) {
        const size_t base = splitmix(s) % (d.n - nk);
        hs_ed25519_keyset *ks = nullptr;
        if (hs_ed25519_keyset_create(&ks, &d.pk[32 * base], nk) != 0) { REQUIRE(false, "keyset_create"); return; }
        auto user = [&](unsigned id) {
            uint64_t r = 5100 + id;
            std::vector<uint8_t> bm(256);
            std::vector<uint32_t> idx(nk);
            for (int it = 0; it < 6 && !g_stop.load(); it++) {
                const size_t n = 1 + splitmix(r) % 1200, l = splitmix(r) % (nk - n);
                for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)(l + i);
                memset(bm.data(), 0, bm.size());
                int rc;
                if (it & 1) {
                    hs_ed25519_ticket *t = nullptr;
                    rc = hs_ed25519_submit(&t, ks, &d.dg[32 * (base + l)], 32, idx.data(), &d.sg[64 * (base + l)], n, HS_ED25519_STRICT, HS_ED25519_SUBMIT_MORE);
                    if (rc == HS_ED25519_E_BUSY) continue;
                    if (rc != 0) { REQUIRE(false, "cached submit"); continue; }
                    rc = hs_ed25519_ticket_wait(t, bm.data());
                } else rc = hs_ed25519_verify_shared_cached(ks, d.shared_dg, idx.data(), &d.sg_shared[64 * (base + l)], n, HS_ED25519_BATCH, bm.data());
                check(d, base + l, n, rc, bm.data(), "keyset");
            }
        };
        std::thread a(user, 0u), b(user, 1u);
        a.join(); b.join();
        // epoch change: the set is destroyed while a ticket on it is still out; another thread collects it a little later.
        // Destroy waits for the host API's uncollected work on the set (no device-wide synchronisation - that would break
        // the ring worker's graph captures), so the late verdict is still the right one.
        const size_t n = 1 + splitmix(s) % 1200, l = splitmix(s) % (nk - n);
        std::vector<uint32_t> idx(n);
        for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)(l + i);
        hs_ed25519_ticket *t = nullptr;
        const int rc = hs_ed25519_submit(&t, ks, d.shared_dg, 0, idx.data(), &d.sg_shared[64 * (base + l)], n, HS_ED25519_BATCH, 0);
        if (rc != 0 && rc != HS_ED25519_E_BUSY) REQUIRE(false, "cached submit before destroy");
        std::thread late([&, t] {
            if (!t) return;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            std::vector<uint8_t> bm(256, 0);
            const int v = hs_ed25519_ticket_wait(t, bm.data());
            check(d, base + l, n, v, bm.data(), "ticket collected during keyset_destroy");
        });
        hs_ed25519_keyset_destroy(ks);
        late.join();
    }
}
static void sharded_worker(const dataset &d) {
    uint64_t s = 6000;
    std::vector<uint8_t> bm(2048);
    const size_t nk = 2048;
    int round = 0;
    while (!g_stop.load()) {
        const size_t n = 500 + splitmix(s) % 6000, lo = splitmix(s) % (d.n - n);
        memset(bm.data(), 0, bm.size());
        int rc = hs_ed25519_verify_multi_sharded(&d.dg[32 * lo], &d.pk[32 * lo], &d.sg[64 * lo], n, HS_ED25519_STRICT, bm.data());
        check(d, lo, n, rc, bm.data(), "sharded");
        hs_ed25519_keyset_sharded *kss = nullptr;
        const size_t base = splitmix(s) % (d.n - nk);
        if (hs_ed25519_keyset_sharded_create(&kss, &d.pk[32 * base], nk, (round++ & 1) ? HS_ED25519_KEYS_BY_OWNER : HS_ED25519_KEYS_REPLICATE) != 0) { REQUIRE(false, "keyset_sharded_create"); return; }
        std::vector<uint32_t> idx(nk);
        // a QC is an arbitrary subset: a strided walk through the set
        const size_t m = 1 + splitmix(s) % (nk / 3);
        for (size_t i = 0; i < m; i++) idx[i] = (uint32_t)((i * 3 + 1) % nk);
        std::vector<uint8_t> sg(64 * m), vmask(m);
        for (size_t i = 0; i < m; i++) memcpy(&sg[64 * i], &d.sg_shared[64 * (base + idx[i])], 64);
        memset(bm.data(), 0, bm.size());
        rc = hs_ed25519_verify_shared_sharded_cached(kss, d.shared_dg, idx.data(), sg.data(), m, HS_ED25519_BATCH, bm.data());
        g_calls++;
        if (rc < 0) { g_errors++; fprintf(stderr, "stress_host: sharded cached: %d %s\n", rc, hs_ed25519_last_error()); }
        else for (size_t i = 0; i < m; i++) if ((bool)((bm[i >> 3] >> (i & 7)) & 1) != (d.valid[base + idx[i]] != 0)) { g_wrong++; break; }
        hs_ed25519_keyset_sharded_destroy(kss);
    }
}
static void escape_worker(const dataset &d) {
    // every wide context out as this thread's tickets, then a synchronous large call: must come back (the pool lends one over its cap)
    std::vector<hs_ed25519_ticket *> held;
    const size_t n = 2000;
    for (int i = 0; i < 8; i++) {
        hs_ed25519_ticket *t = nullptr;
        const int rc = hs_ed25519_submit(&t, nullptr, d.dg.data(), 32, d.pk.data(), d.sg.data(), n, HS_ED25519_STRICT, 0);
        if (rc == HS_ED25519_E_BUSY) break;
        if (rc != 0) { REQUIRE(false, "escape submit"); break; }
        held.push_back(t);
    }
    std::vector<uint8_t> bm(512, 0);
    const int rc = hs_ed25519_verify_multi(&d.dg[32 * 100], &d.pk[32 * 100], &d.sg[64 * 100], n, HS_ED25519_STRICT, bm.data());
    check(d, 100, n, rc, bm.data(), "escape");
    for (hs_ed25519_ticket *t : held) { memset(bm.data(), 0, bm.size()); const int r2 = hs_ed25519_ticket_wait(t, bm.data()); check(d, 0, n, r2, bm.data(), "escape-held"); }
}
static void ctx_churn_worker() {      // contexts created and destroyed beside everything else (no device buffers of the caller's: the ABI surface only)
    while (!g_stop.load()) {
        hs_ed25519_ctx *c = nullptr;
        if (hs_ed25519_ctx_create_on(&c, 2048, 1) != 0) { REQUIRE(false, "ctx_create_on"); return; }
        (void)hs_ed25519_ctx_set_concurrent(c, 1);
        (void)hs_ed25519_ctx_set_rlc_min(c, 64);
        size_t calls = 0;
        REQUIRE(hs_ed25519_ctx_rlc_last(c, &calls) == HS_ED25519_RLC_NOT_RUN, "ctx_rlc_last on a fresh context");
        hs_ed25519_ctx_destroy(c);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}

// `stress_host canary`: a deliberate data race and a deliberate heap overflow of one byte - run once per sanitizer build to show
// that the detector is alive in this process on this box (exit code 66 / 67 from TSAN_OPTIONS / ASAN_OPTIONS exitcode=...)
static int g_canary;
static int canary() {
    std::thread a([] { for (int i = 0; i < 100000; i++) g_canary++; }), b([] { for (int i = 0; i < 100000; i++) g_canary++; });
    a.join(); b.join();
    volatile char *p = (volatile char *)malloc(16);
    p[16] = 1;                                          // one past the end
    const int v = p[16];
    free((void *)p);
    printf("stress_host: canary ran (%d, %d)\n", g_canary, v);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "canary")) return canary();
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    const unsigned scale = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    const int ordinals[2] = {0, 0};                     // two virtual device slots on one GPU: the sharded paths and the round-robin of cold calls
    if (hs_ed25519_init_devices(ordinals, 2, 8192) != 0) { fprintf(stderr, "stress_host: init failed: %s\n", hs_ed25519_last_error()); return 2; }
    dataset d;
    if (!make_data(d, 12288)) { fprintf(stderr, "stress_host: cannot sign the data set: %s\n", hs_ed25519_last_error()); return 2; }
    {   // the data set itself: one plain call, whole
        std::vector<uint8_t> bm((d.n + 7) / 8 + 8, 0);
        const int rc = hs_ed25519_verify_multi(d.dg.data(), d.pk.data(), d.sg.data(), d.n, HS_ED25519_STRICT, bm.data());
        check(d, 0, d.n, rc, bm.data(), "whole set");
        if (g_wrong.load() || g_errors.load()) { fprintf(stderr, "stress_host: the data set does not verify as built\n"); return 2; }
    }
    handoff h;
    std::vector<std::thread> th;
    for (unsigned i = 0; i < 4 * scale; i++) th.emplace_back(sync_worker, std::cref(d), i);
    for (unsigned i = 0; i < 3 * scale; i++) th.emplace_back(ticket_worker, std::cref(d), i);
    th.emplace_back(handoff_producer, std::cref(d), std::ref(h));
    th.emplace_back(handoff_consumer, std::cref(d), std::ref(h));
    th.emplace_back(ring_worker, std::cref(d), HS_ED25519_RING_ZERO_COPY);
    th.emplace_back(ring_worker, std::cref(d), HS_ED25519_RING_GRAPH);
    th.emplace_back(keyset_worker, std::cref(d));
    th.emplace_back(sharded_worker, std::cref(d));
    th.emplace_back(ctx_churn_worker);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds * 0.5));
    std::thread esc(escape_worker, std::cref(d));
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds * 0.5));
    g_stop.store(true);
    { std::lock_guard<std::mutex> lk(h.mu); h.cv.notify_all(); }
    esc.join();
    for (std::thread &t : th) t.join();
    hs_ed25519_shard_reset();
    hs_ed25519_shutdown();
    printf("stress_host: %ld calls checked, %ld wrong verdicts, %ld engine errors in %.1f s (%zu threads)\n", g_calls.load(), g_wrong.load(), g_errors.load(),
           seconds, th.size() + 1);
    const int code = (g_wrong.load() || g_errors.load()) ? 1 : 0;
    // HS_TEST_QUICK_EXIT=1 (the ASan leg of tools/gpu_visit.sh): leave without running the HIP runtime's exit-time finalizers.  ROCm's
    // ASan keeps freed pinned-memory chunks in a quarantine and recycles them from INSIDE that finalizer when the quarantine overflows
    // there - after it has marked the device runtime unloaded - and then fails its own CHECK (sanitizer_allocator_device.h:125),
    // taking this program's buffered output and exit code with it (profiles/r6_t_sanitizer_runs.txt).  Everything this program checks
    // has been checked by now, hs_ed25519_shutdown() included.
    if (const char *q = getenv("HS_TEST_QUICK_EXIT")) if (q[0] == '1') { fflush(stdout); fflush(stderr); _exit(code); }
    return code;
}
