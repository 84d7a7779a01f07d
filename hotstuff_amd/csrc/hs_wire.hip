// hs_wire.hip - what sits either side of the verification path: the native load generator
// used to measure the host-pointer API end to end, [wire decode and batch signing follow below].
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include "engine.h"

using namespace hs;
#define fail hs::hs_fail

// ---- load generator ---------------------------------------------------------------------------------
// SURVEY.md 8d defines the headline metric "end-to-end from host buffers to host verdict".  Driving the
// host-pointer API from interpreter threads measures the interpreter (a thread returning from a call waits
// for the interpreter lock); this drives it from `threads` native threads instead: each loops over
// hs_ed25519_verify_multi / _shared on its own batch (batch t of `nbatches`, round-robin) for `seconds`.
extern "C" int hs_ed25519_loadgen(const uint8_t *digests, size_t digest_stride, const uint8_t *pks, const uint8_t *sigs,
                                  size_t n, size_t nbatches, uint32_t mode, unsigned threads, double seconds,
                                  uint64_t *calls_out, double *p50_ms_out, double *elapsed_s_out) {
    if (!digests || !pks || !sigs || !n || !nbatches || !threads || threads > 256 || seconds <= 0 || !calls_out)
        return fail(HS_ED25519_E_ARG, "loadgen: bad arguments");
    if (!engine_ready()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    std::atomic<bool> go{false}, stop{false};
    std::atomic<int> bad{0};
    std::vector<uint64_t> calls(threads, 0);
    std::vector<std::vector<float>> lat(threads);
    std::vector<std::thread> th;
    const size_t dg_bytes = digest_stride ? 32 * n : 32;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            const size_t b = t % nbatches;
            const uint8_t *d = digests + b * dg_bytes, *p = pks + b * 32 * n, *s = sigs + b * 64 * n;
            lat[t].reserve(4096);
            while (!go.load()) std::this_thread::yield();
            while (!stop.load()) {
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = digest_stride ? hs_ed25519_verify_multi(d, p, s, n, mode, nullptr)
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
    if (bad.load()) return fail(HS_ED25519_E_DEVICE, "loadgen: a call rejected its (valid) batch or failed");
    uint64_t tot = 0;
    std::vector<float> all;
    for (unsigned t = 0; t < threads; t++) { tot += calls[t]; all.insert(all.end(), lat[t].begin(), lat[t].end()); }
    std::sort(all.begin(), all.end());
    *calls_out = tot;
    if (p50_ms_out) *p50_ms_out = all.empty() ? 0.0 : all[all.size() / 2];
    if (elapsed_s_out) *elapsed_s_out = el;
    return 0;
}
