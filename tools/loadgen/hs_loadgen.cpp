// hs_loadgen.cpp - load generator over the PUBLIC C ABI of libhs_ed25519.so (include/hs_ed25519.h).  A measurement
// aid, deliberately NOT part of the shipped library (round 2 exported it from there): tools/libhs_loadgen.so, built
// by __graft_entry__.build_tools(), links against the product like any other client.
//
// SURVEY.md 8d defines the headline metric "end-to-end from host buffers to host verdict".  Driving the
// host-pointer API from interpreter threads measures the interpreter (a thread returning from a call waits
// for the interpreter lock); this drives it from `threads` native threads instead: thread t loops over batch
// t mod nbatches of the caller's packed host buffers (nbatches x n tuples, all valid; with a key set `ks` the
// *_cached forms, `pks` then holding nbatches x n uint32 key indices) for `seconds`.
//   inflight <= 1   the synchronous calls (hs_ed25519_verify_multi / _shared[_cached])
//   inflight = k    every thread keeps k tickets outstanding through hs_ed25519_submit / hs_ed25519_ticket_wait
//                   (latency = submit to verdict), submitted in bursts of batches_per_launch
//                   (HS_ED25519_SUBMIT_MORE on all but the last of a burst)
// Whether a call is staged or read in place is decided by the engine from where the buffers live
// (hs_ed25519_host_alloc memory is read in place): the caller chooses by how it allocates them.
// Reports completed calls, the median call latency and the elapsed time.  Returns 0, or a negative HS_ED25519_E_* /
// -100 (bad arguments) / -101 (a call rejected its valid batch).
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <thread>
#include <vector>
#include "../../include/hs_ed25519.h"

extern "C" int hs_loadgen(const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride, const uint8_t *pks, const uint8_t *sigs,
                          size_t n, size_t nbatches, uint32_t mode, unsigned threads, unsigned inflight, unsigned batches_per_launch,
                          double seconds, uint64_t *calls_out, double *p50_ms_out, double *elapsed_s_out) {
    if (!digests || !pks || !sigs || !n || !nbatches || !threads || threads > 256 || seconds <= 0 || !calls_out) return -100;
    if (hs_ed25519_device_count() <= 0) return HS_ED25519_E_NOT_INIT;
    std::atomic<bool> go{false}, stop{false};
    std::atomic<int> bad{0};
    std::vector<uint64_t> calls(threads, 0);
    std::vector<std::vector<float>> lat(threads);
    std::vector<std::thread> th;
    const size_t dg_bytes = digest_stride ? 32 * n : 32;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            const size_t b = t % nbatches;
            const uint8_t *d = digests + b * dg_bytes, *p = pks + b * (ks ? 4 : 32) * n, *s = sigs + b * 64 * n;
            lat[t].reserve(4096);
            while (!go.load()) std::this_thread::yield();
            if (inflight > 1) {          // a window of tickets: submit until full (or the pool is busy), then collect the oldest
                const unsigned burst = batches_per_launch < 1 ? 1 : (batches_per_launch > inflight ? inflight : batches_per_launch);
                const bool any_order = getenv("HS_LOADGEN_ANY") != nullptr;
                std::deque<std::pair<hs_ed25519_ticket *, std::chrono::steady_clock::time_point>> win;
                auto collect = [&]() -> bool {
                    const int rc = hs_ed25519_ticket_wait(win.front().first, nullptr);
                    lat[t].push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - win.front().second).count());
                    win.pop_front();
                    if (rc != 1) { bad.store(rc == 0 ? 1 : rc); return false; }
                    calls[t]++;
                    return true;
                };
                bool ok = true;
                while (ok && !stop.load()) {
                    // refill in bursts of `burst` (one launch each): MORE on all but the last of a burst
                    while (ok && win.size() + burst <= inflight) {
                        unsigned sent = 0;
                        for (unsigned k = 0; k < burst; k++) {
                            hs_ed25519_ticket *tk = nullptr;
                            const auto t0 = std::chrono::steady_clock::now();
                            const int rc = hs_ed25519_submit(&tk, ks, d, digest_stride, p, s, n, mode, k + 1 < burst ? HS_ED25519_SUBMIT_MORE : 0);
                            if (rc == 0) { win.emplace_back(tk, t0); sent++; continue; }
                            if (rc != HS_ED25519_E_BUSY) { bad.store(rc); ok = false; }
                            break;
                        }
                        if (sent < burst) break;       // pool busy (or failed): collect first
                    }
                    if (!ok) break;
                    if (win.empty()) { std::this_thread::yield(); continue; }
                    if (any_order && win.size() > burst) {
                        // an async host awaits its tickets concurrently: take whichever burst has completed (hs_ed25519_ticket_ready on
                        // its first ticket) before blocking on the oldest
                        for (size_t i = burst; i + burst <= win.size(); i += burst)
                            if (hs_ed25519_ticket_ready(win[i].first) == 1) { std::rotate(win.begin(), win.begin() + (long)i, win.begin() + (long)(i + burst)); break; }
                    }
                    ok = collect();
                }
                while (!win.empty()) if (!collect()) { while (!win.empty()) { (void)hs_ed25519_ticket_wait(win.front().first, nullptr); win.pop_front(); } }
                return;
            }
            while (!stop.load()) {
                const auto t0 = std::chrono::steady_clock::now();
                int rc;
                if (ks) rc = digest_stride ? hs_ed25519_verify_multi_cached(ks, d, (const uint32_t *)p, s, n, mode, nullptr)
                                           : hs_ed25519_verify_shared_cached(ks, d, (const uint32_t *)p, s, n, mode, nullptr);
                else rc = digest_stride ? hs_ed25519_verify_multi(d, p, s, n, mode, nullptr)
                                        : hs_ed25519_verify_shared(d, p, s, n, mode, nullptr);
                lat[t].push_back(std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count());
                if (rc != 1) { bad.store(rc == 0 ? 1 : rc); return; }
                calls[t]++;
            }
        });
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    stop.store(true);
    for (auto &x : th) x.join();
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (bad.load()) return bad.load() < 0 ? bad.load() : -101;
    uint64_t tot = 0;
    std::vector<float> all;
    for (unsigned t = 0; t < threads; t++) { tot += calls[t]; all.insert(all.end(), lat[t].begin(), lat[t].end()); }
    std::sort(all.begin(), all.end());
    *calls_out = tot;
    if (p50_ms_out) *p50_ms_out = all.empty() ? 0.0 : all[all.size() / 2];
    if (elapsed_s_out) *elapsed_s_out = el;
    return 0;
}
