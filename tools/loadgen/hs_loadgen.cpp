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

// ---- the callers either side of the path, timed through the C++ twin of the reference's interface -----------
// hs_callers_bench: a committee of `validators` (keys and signatures made on the GPU through the C ABI), its keys registered;
//   (1) vote ingest (consensus/src/core.rs:203-225 + aggregator.rs:74-94, batched: include/hotstuff_crypto.hpp VoteIngest): the
//       q = floor(2N/3) + 1 votes of one round arrive one by one; wall time from the first vote to the QC, and the engine calls it took;
//   (2) timeout storm (messages.rs:250-265, core.rs:227-238): `timeouts` timeouts of one view change, each embedding the SAME high_qc,
//       through VerifiedQCCache: the embedded certificate is verified once, the rest are fingerprint hits.  (The timeouts' own
//       signatures - one strict single each, which the shim keeps on dalek - are not part of this figure.)
// out[0] = ingest ms, out[1] = ingest engine calls, out[2] = storm ms, out[3] = cache misses, out[4] = cache hits, out[5] = one uncached
// QC::verify of the same certificate in ms (what the reference pays PER TIMEOUT, on the GPU instead of the CPU).  Returns 0 or < 0.
#include "../../include/hotstuff_crypto.hpp"
extern "C" int hs_callers_bench(unsigned validators, unsigned timeouts, double *out) {
    using namespace hotstuff;
    if (!out || validators < 4 || hs_ed25519_device_count() <= 0) return -100;
    const size_t N = validators, q = 2 * N / 3 + 1;
    std::vector<uint8_t> seeds(32 * N), pks(32 * N), secrets(64 * N), sigs(64 * q);
    for (size_t i = 0; i < N; i++) for (int j = 0; j < 32; j++) seeds[32 * i + j] = (uint8_t)((i * 2654435761u + 97u * (unsigned)j + 13u) >> (8 * (j & 3)));
    if (hs_ed25519_public_keys(seeds.data(), N, pks.data()) != 0) return -1;
    for (size_t i = 0; i < N; i++) { memcpy(&secrets[64 * i], &seeds[32 * i], 32); memcpy(&secrets[64 * i + 32], &pks[32 * i], 32); }
    Committee committee;
    std::vector<PublicKey> names(N);
    for (size_t i = 0; i < N; i++) { memcpy(names[i].b.data(), &pks[32 * i], 32); committee.authorities[names[i]] = 1; }
    if (!committee.cache_keys()) return -2;
    Vote proto; proto.round = 7;
    for (int j = 0; j < 32; j++) proto.hash.b[j] = (uint8_t)(3 * j + 1);
    const Digest vd = proto.digest();
    if (hs_ed25519_sign_batch(secrets.data(), vd.b.data(), 0, q, sigs.data()) != 0) return -3;
    std::vector<Vote> votes(q, proto);
    for (size_t i = 0; i < q; i++) { votes[i].author = names[i]; votes[i].signature = Signature::from_bytes(&sigs[64 * i]); }
    std::unique_ptr<QC> qc;
    size_t calls = 0;
    for (int rep = 0; rep < 3; rep++) {          // the last repetition is the one reported (pools, workspaces, tables warm)
        VoteIngest ing(committee);
        qc.reset();
        const auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < q && !qc; i++) {
            VoteIngest::Outcome o = ing.add_vote(votes[i]);
            if (!o.status.is_ok()) return -4;
            if (o.qc) qc = std::move(o.qc);
        }
        out[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        calls = ing.verify_calls;
        if (!qc || qc->votes.size() != q) return -5;
    }
    out[1] = (double)calls;
    {
        const auto t0 = std::chrono::steady_clock::now();
        const int reps = 20;
        for (int i = 0; i < reps; i++) if (qc->verify(committee).is_err()) return -6;
        out[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    VerifiedQCCache cache;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned t = 0; t < timeouts; t++) if (cache.verify(*qc, committee).is_err()) return -7;
    out[2] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    out[3] = (double)cache.misses; out[4] = (double)cache.hits;
    return 0;
}
