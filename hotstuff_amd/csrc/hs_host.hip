// hs_host.hip - the host-pointer side of libhs_ed25519.so (include/hs_ed25519.h): what the reference's shim binds.
//   pools of stream contexts per device slot (wide / narrow = confined to a slice of the CU mask: the same share of every XCD, profiles/r6_u_cu_mask_probe.md)
//   hs_ed25519_verify_{shared,multi}[_cached], hs_ed25519_verify_tc[_cached]      synchronous, host buffers in, verdict out
//   hs_ed25519_submit / ticket_ready / ticket_wait                                asynchronous; the small calls of a burst share a launch
//   hs_ed25519_ring_*                                                             streamed rounds through pinned slots (BASELINE config 5)
// Replaces the call sites of ed25519-dalek at /root/reference/crypto/src/lib.rs:200-219 as seen from the host;
// the kernels and the device-pointer entry points it enqueues live in hs_ed25519.hip / hs_rlc.hip.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <new>
#include <mutex>
#include <shared_mutex>
#include <algorithm>
#include <condition_variable>
#include <vector>
#include <chrono>
#include "engine.h"

using namespace hs;
#define fail hs::hs_fail

// ---- host-pointer API ----------------------------------------------------------------------
// the call's verdict from the accept bitmap (the host forms pass no reject counter to the kernels);
// the bitmap words of [0, n) are zeroed before every submit, so a launch that did not write them
// can never be read as "all accepted"
bool hs::bitmap_all_ones(const uint8_t *bm, size_t n) {
    const size_t words = n / 64;
    for (size_t i = 0; i < words; i++) { uint64_t w; memcpy(&w, bm + 8 * i, 8); if (w != ~(uint64_t)0) return false; }
    for (size_t i = 64 * words; i < n; i++) if (!((bm[i >> 3] >> (i & 7)) & 1)) return false;
    return true;
}
// How long a synchronous call waits for a context of its pool before it asks whether waiting can end at all: three orders
// of magnitude above a call's duration, so ordinary contention (more synchronous callers than contexts) never gets near it
// and the pools stay at their caps - which is what keeps the streams within the 16 hardware queues.
static const std::chrono::milliseconds kPoolPatience(1000);
// Contexts this thread has out as TICKETS (hs_ed25519_submit), per (device slot, pool).  Counted on the context, not on the
// ticket: a ticket may be collected by another thread, and the count must follow the context back to the pool wherever that
// happens - hence shared counters that outlive the thread.  Per slot AND pool: a thread with tickets out on slot A
// must not be lent an over-cap context on slot B, whose contexts other threads hold and will return.
// Hand-off caveat (include/hs_ed25519.h, tickets): the escape below looks at what the CALLING thread submitted; a thread that
// collects tickets submitted by another has a count of zero and waits like any other caller.
namespace {
struct held_counts {
    std::vector<std::shared_ptr<std::atomic<int>>> n;      // [2 * slot + pool]
    std::shared_ptr<std::atomic<int>> &at(int slot, int k) {
        const size_t i = 2 * (size_t)slot + (size_t)k;
        if (n.size() <= i) n.resize(i + 1);
        if (!n[i]) n[i] = std::make_shared<std::atomic<int>>(0);
        return n[i];
    }
};
thread_local held_counts t_held;
}
int hs::pool_get(hs_ed25519_ctx **out, int slot, size_t n, bool nowait) {
    const int k = n <= kPartitionMaxBatch ? 1 : 0;
    std::unique_lock<std::mutex> lk(g_mu);
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (slot < 0 || slot >= (int)g_devs.size()) return fail(HS_ED25519_E_ARG, "no such device slot");
    bool waiting = false, overflow = false;
    std::chrono::steady_clock::time_point deadline;
    for (;;) {
        if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_shutdown was called");
        dev_state &d = g_devs[slot];
        const bool have_free = !d.free_ctx[k].empty();
        // a new context when none is free - and, in the narrow pool, until every CU slice has one
        // (narrow pool: kNarrowDepth contexts queue on each of its kMaxPool[1] streams)
        if (d.live[k] < kMaxPool[k] * (k == 1 ? kNarrowDepth : 1) && (!have_free || (k == 1 && d.live[k] < (int)kPoolPartitions))) { d.live[k]++; break; }
        if (have_free) {
            // narrow pool: two streams share each CU slice and kNarrowDepth contexts each stream - take a free context whose
            // stream is idle if there is one, and among those the slice with the fewest launches in flight (with eight of
            // sixteen contexts busy, a last-in-first-out pick left half the chip idle half of the time: 42 against 74 M
            // verifies/s for four threads keeping eight sixteen-batch launches in flight)
            size_t pick = d.free_ctx[k].size() - 1;
            if (k == 1) {
                unsigned best = ~0u;
                for (size_t i = d.free_ctx[k].size(); i-- > 0;) {
                    const hs_ed25519_ctx *f = d.free_ctx[k][i];
                    const unsigned busy = d.lane_busy[f->lane % 64] * 256u + d.part_busy[f->part % 64];
                    if (busy < best) { best = busy; pick = i; }
                }
            }
            hs_ed25519_ctx *c = d.free_ctx[k][pick];
            d.free_ctx[k].erase(d.free_ctx[k].begin() + (long)pick);
            if (k == 1) { d.part_busy[c->part % 64]++; d.lane_busy[c->lane % 64]++; }
            // a whole-chip call that finds another one in flight on its device runs the throughput forms (enqueue_cold)
            else c->several_in_flight = d.live[0] - (int)d.free_ctx[0].size() > 1;
            if (nowait) { c->held = t_held.at(slot, k); c->held->fetch_add(1, std::memory_order_relaxed); }
            *out = c;
            return 0;
        }
        if (nowait) return HS_ED25519_E_BUSY;       // not an error worth a message: the caller retries after a wait
        // A synchronous call waits for a context to come back.  Whoever holds them will return them - unless that is the
        // CALLING thread itself: contexts it has out as tickets it has not collected yet (four large ones are enough) come
        // back only after this call returns.  So: after kPoolPatience without a context, a caller that HOLDS contexts of this
        // pool is lent one over the cap (marked `overflow`: pool_put destroys exactly that one, so the pool and its CU slices
        // stay as they were); a caller that holds none keeps waiting, as it always did - under genuine saturation by other
        // threads' long calls the pools must not grow past the hardware-queue budget (the
        // escape used to fire for ANY caller after one second, without bound).  At most one overflow context per calling
        // thread exists at a time: the call that got it returns it before the thread can make another.
        // tests/test_gpu_tickets.py: a thread holding every wide context, then verifying; other threads waiting meanwhile.
        if (!waiting) { waiting = true; deadline = std::chrono::steady_clock::now() + kPoolPatience; }
        if (g_cv.wait_until(lk, deadline) == std::cv_status::timeout) {          // (other pools' returns wake it too: one deadline per wait)
            if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_shutdown was called");
            if (g_devs[slot].free_ctx[k].empty()) {
                if (t_held.at(slot, k)->load(std::memory_order_relaxed) > 0) { g_devs[slot].live[k]++; overflow = true; break; }
                waiting = false;                     // others hold them: wait on
            }
        }
    }
    const unsigned lane = (unsigned)(g_devs[slot].live[k] - 1) % (unsigned)(kMaxPool[k] > 0 ? kMaxPool[k] : 1);
    // Lane l serves slice l mod (slices): CONSECUTIVELY created streams go to DIFFERENT slices.  Not a detail (round 6, profiles/
    // r6_x_lane_map.md): the streams get their hardware queues in creation order and consecutive queues sit on different pipes of the
    // command processor, so this way the streams of ONE slice share a pipe (lanes s, s + 4, s + 8, s + 12).  The other way round -
    // lanes 4 s ... 4 s + 3 on slice s, every pipe serving all four slices - the same sixteen streams run 85 - 86 M verifies/s instead
    // of 103 - 104 M: a dispatch that cannot place its wavefronts yet (its slice's SIMDs are full) holds its pipe, and the queues of
    // the OTHER slices behind it on that pipe wait although their SIMDs are free.  Which is what a pool created by sixteen threads
    // at once got wrong at random (86 - 88 M, r6_j): the ORDER of creation, not its concurrency - hence creation in turn, below.
    const unsigned part = lane % kPoolPartitions, cus = g_devs[slot].cus;
    const bool others_busy = k == 0 && g_devs[slot].live[0] - (int)g_devs[slot].free_ctx[0].size() > 1;
    const int ordinal = g_devs[slot].ordinal;
    lk.unlock();
    hs_ed25519_ctx *c = nullptr;
    int rc = -1;
    // this tenant's share of the chip (the whole of it by default)
    const unsigned t_lo = cus * kTenantIndex / kTenants, t_cus = cus * (kTenantIndex + 1) / kTenants - t_lo;
    auto slice_mask = [&](unsigned p) {          // slice p of the tenant's CUs
        std::vector<uint32_t> mask((cus + 31) / 32, 0u);
        const unsigned lo = t_lo + t_cus * p / kPoolPartitions, hi = t_lo + t_cus * (p + 1) / kPoolPartitions;
        for (unsigned u = lo; u < hi; u++) mask[u / 32] |= 1u << (u % 32);
        return mask;
    };
    // Pool contexts are created one at a time, and the narrow pool's CU-masked streams all at once, in lane order, by whichever
    // thread needs the first of them (round 6).  Sixteen threads whose first calls created the pool CONCURRENTLY left the
    // process at 86 - 88 M verifies/s for good where a pool created by one thread ran 102 - 105 M on the same GPU, same load
    // (profiles/r6_j_pool_creation_order.md): created in parallel the streams get their hardware queues - and with them their
    // pipes - in a random order (the masks themselves are applied either way); see `part` above for why the order matters.
    static std::mutex s_create_mu;
    std::lock_guard<std::mutex> creating(s_create_mu);
    if (k == 1 && cus >= kPoolPartitions) {      // a stream confined to slice `part` of the tenant's CUs
        lk.lock();
        bool need = g_init.load() && slot < (int)g_devs.size() && !g_devs[slot].lanes;
        lk.unlock();
        if (need && hipSetDevice(ordinal) == hipSuccess) {
            auto *made = new (std::nothrow) std::vector<std::shared_ptr<stream_box>>((size_t)(kMaxPool[1] > 0 ? kMaxPool[1] : 1));
            std::vector<std::shared_ptr<stream_box>> none, &boxes = made ? *made : none;
            for (size_t l = 0; l < boxes.size(); l++) {
                const std::vector<uint32_t> mask = slice_mask((unsigned)l % kPoolPartitions);
                hipStream_t st = nullptr;
                if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { (void)hipGetLastError(); break; }      // CU masking unavailable: plain streams, one per context
                boxes[l] = std::shared_ptr<stream_box>(new (std::nothrow) stream_box());
                if (!boxes[l]) { (void)hipStreamDestroy(st); break; }
                boxes[l]->s = st; boxes[l]->device = ordinal;
            }
            lk.lock();
            if (made && g_init.load() && slot < (int)g_devs.size() && !g_devs[slot].lanes) { g_devs[slot].lanes = made; made = nullptr; }
            lk.unlock();
            delete made;       // (the engine was shut down meanwhile: the streams go with it)
        }
        std::shared_ptr<stream_box> share;
        lk.lock();
        if (g_init.load() && slot < (int)g_devs.size() && g_devs[slot].lanes && lane < g_devs[slot].lanes->size()) share = (*g_devs[slot].lanes)[lane];
        lk.unlock();
        const std::vector<uint32_t> mask = slice_mask(part);
        if (share) rc = ctx_new(&c, 64, slot, mask.data(), (uint32_t)mask.size(), share);
        if (rc) (void)hipGetLastError();          // CU masking unavailable: a plain stream serves as well
    } else if (k == 0 && kTenants > 1 && kTenantWide && t_cus > 0 && t_cus < cus) {
        // co-tenants: the whole-chip calls of this process stay on its own share too, or four tenants' QC-sized calls
        // stack their wavefronts on the same SIMDs while each tenant's small calls queue behind them
        std::vector<uint32_t> mask((cus + 31) / 32, 0u);
        for (unsigned u = t_lo; u < t_lo + t_cus; u++) mask[u / 32] |= 1u << (u % 32);
        rc = ctx_new(&c, 64, slot, mask.data(), (uint32_t)mask.size());
        if (rc) (void)hipGetLastError();
    }
    if (rc) rc = hs_ed25519_ctx_create_on(&c, 64, slot);
    if (rc) { lk.lock(); if (g_init.load() && slot < (int)g_devs.size()) g_devs[slot].live[k]--; g_cv.notify_all(); return rc; }
    c->narrow = k == 1;
    c->part = part;
    c->lane = lane;
    c->overflow = overflow;
    c->several_in_flight = others_busy;
    if (nowait) { c->held = t_held.at(slot, k); c->held->fetch_add(1, std::memory_order_relaxed); }
    if (k == 1) {
        lk.lock();
        if (g_init.load() && slot < (int)g_devs.size()) { g_devs[slot].part_busy[part % 64]++; g_devs[slot].lane_busy[lane % 64]++; }
        lk.unlock();
    }
    *out = c;
    return 0;
}
void hs::pool_put(hs_ed25519_ctx *c) {
    if (c->held) { c->held->fetch_sub(1, std::memory_order_relaxed); c->held.reset(); }
    std::unique_lock<std::mutex> lk(g_mu);
    if (!g_init.load() || c->slot >= (int)g_devs.size()) { lk.unlock(); hs_ed25519_ctx_destroy(c); return; }
    const int k = c->narrow ? 1 : 0;
    if (c->narrow && g_devs[c->slot].part_busy[c->part % 64]) g_devs[c->slot].part_busy[c->part % 64]--;
    if (c->narrow && g_devs[c->slot].lane_busy[c->lane % 64]) g_devs[c->slot].lane_busy[c->lane % 64]--;
    if (c->overflow) {                                  // lent over the cap: exactly this one goes, the pool keeps its own
        g_devs[c->slot].live[k]--;
        lk.unlock();
        hs_ed25519_ctx_destroy(c);
        return;
    }
    g_devs[c->slot].free_ctx[k].push_back(c);
    g_cv.notify_all();
}

// ---- caller-owned pinned buffers ---------------------------------------------------------------
// hs_ed25519_host_alloc hands out pinned, device-mapped host memory and remembers the range.  A host-pointer call whose
// digests, keys and signatures ALL lie inside such ranges (and are aligned as the device entry points ask) is not staged:
// the kernels read the caller's memory over PCIe directly, as they read the engine's own pinned staging - the 128 bytes per
// signature are then written once (by whoever packs them: the shim copying out of Vec<(PublicKey, Signature)>) instead of
// twice.  One thread submitting bursts of sixteen 1,024-signature batches spends its time in that second copy (2 MB per
// launch): bench.py `end_to_end`, rows with "caller_pinned".  Such buffers belong to the call until its verdict has been
// collected (ticket_wait / the synchronous return); ordinary buffers may still be reused as soon as submit returns.
namespace {
std::shared_mutex r_mu;
std::vector<std::pair<uintptr_t, size_t>> r_ranges;      // sorted by start
bool pinned_range(const void *p, size_t len) {
    const uintptr_t a = (uintptr_t)p;
    std::shared_lock<std::shared_mutex> lk(r_mu);
    auto it = std::upper_bound(r_ranges.begin(), r_ranges.end(), std::make_pair(a, (size_t)-1));
    if (it == r_ranges.begin()) return false;
    --it;
    return a >= it->first && a + len <= it->first + it->second;
}
// all three inputs of a call in caller-owned pinned memory, aligned for the kernels
bool caller_pinned(const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride, const void *keys, const uint8_t *sigs, size_t n) {
    if (misaligned(digests, 16) || misaligned(keys, ks ? 4 : 16) || misaligned(sigs, 16)) return false;
    { std::shared_lock<std::shared_mutex> lk(r_mu); if (r_ranges.empty()) return false; }
    return pinned_range(digests, digest_stride ? 32 * n : 32) && pinned_range(keys, (ks ? 4 : 32) * n) && pinned_range(sigs, 64 * n);
}
}  // namespace
extern "C" void *hs_ed25519_host_alloc(size_t bytes) {
    if (!bytes || !g_init.load()) { (void)fail(HS_ED25519_E_ARG, "host_alloc: zero bytes, or hs_ed25519_init has not succeeded"); return nullptr; }
    void *p = nullptr;
    // Portable: visible to every device slot of the process (a sharded call reads it from several GPUs)
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) { (void)fail(HS_ED25519_E_NOMEM, "host_alloc: hipHostMalloc", e); return nullptr; }
    std::unique_lock<std::shared_mutex> lk(r_mu);
    const auto v = std::make_pair((uintptr_t)p, bytes);
    r_ranges.insert(std::upper_bound(r_ranges.begin(), r_ranges.end(), v), v);
    return p;
}
extern "C" void hs_ed25519_host_free(void *p) {
    if (!p) return;
    {
        std::unique_lock<std::shared_mutex> lk(r_mu);
        for (size_t i = 0; i < r_ranges.size(); i++) if (r_ranges[i].first == (uintptr_t)p) { r_ranges.erase(r_ranges.begin() + (long)i); break; }
    }
    (void)hipHostFree(p);
}

// One host-pointer verification on one device slot, split into submit (stage + enqueue, asynchronous)
// and finish (wait + verdict) so that hs_shard.hip can have several devices in flight from one thread.
// Up to kZeroCopyMax signatures the kernels read the pinned staging and write the pinned bitmap
// directly over PCIe (hipHostMalloc memory is mapped on the device; every input byte is read
// exactly once): no copy-engine round trips, -10..20 us per call.  Above, bulk H2D / D2H copies.
//   keys: n*32 key bytes (ks == NULL) or n uint32 indices into `ks`
int hs::host_submit(host_job &job, int slot, const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride,
                    const void *keys, const uint8_t *sigs, size_t n, uint32_t mode, bool nowait, const uint32_t *gather) {
    job.c = nullptr; job.n = n;
    int rc = pool_get(&job.c, slot, n, nowait);
    if (rc) return rc;
    hs_ed25519_ctx *c = job.c;
    job.ks = ks; ks_hold(ks);                      // released with the context: host_finish / host_abandon
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    const size_t hint = c->narrow ? kPartitionMaxBatch : g_max_batch.load();
    rc = ctx_reserve(c, n > hint ? n : hint, true);
    if (rc) return rc;
    // staging layout: [digests][keys: 32n bytes or 4n index bytes][sigs: 64n], each section 16-byte aligned
    const size_t dg_bytes = digest_stride ? 32 * n : 32, key_bytes = ks ? 4 * n : 32 * n;
    const size_t off_k = round_up(dg_bytes, 16), off_sig = round_up(off_k + key_bytes, 16), in_bytes = off_sig + 64 * n;
    const bool own = !gather && caller_pinned(ks, digests, digest_stride, keys, sigs, n);     // caller-owned pinned buffers: no staging copy
    if (gather) {                  // a routed shard (hs_shard.hip, keys by owner): packed here, once
        if (digest_stride) for (size_t p = 0; p < n; p++) memcpy(c->h_in + 32 * p, digests + (size_t)32 * gather[p], 32);
        else memcpy(c->h_in, digests, 32);
        memcpy(c->h_in + off_k, keys, key_bytes);
        uint8_t *dst = c->h_in + off_sig;
        for (size_t p = 0; p < n; p++) memcpy(dst + 64 * p, sigs + (size_t)64 * gather[p], 64);
    } else if (!own) {
        memcpy(c->h_in, digests, dg_bytes);
        memcpy(c->h_in + off_k, keys, key_bytes);
        memcpy(c->h_in + off_sig, sigs, 64 * n);
    }
    job.zc = n <= kZeroCopyMax;
    uint8_t *in = job.zc ? c->h_in : c->d_in, *outp = job.zc ? c->h_out : c->d_out;
    const uint8_t *p_dg = in, *p_k = in + off_k, *p_sg = in + off_sig;
    if (job.zc) {
        memset(c->h_out, 0, bitmap_bytes(n));
        if (own) { p_dg = digests; p_k = (const uint8_t *)keys; p_sg = sigs; }
    } else {
        HS_HIP(hipMemsetAsync(c->d_out, 0, bitmap_bytes(n), c->stream), "clear bitmap");
        if (own) {
            HS_HIP(hipMemcpyAsync(c->d_in, digests, dg_bytes, hipMemcpyHostToDevice, c->stream), "H2D digests");
            HS_HIP(hipMemcpyAsync(c->d_in + off_k, keys, key_bytes, hipMemcpyHostToDevice, c->stream), "H2D keys");
            HS_HIP(hipMemcpyAsync(c->d_in + off_sig, sigs, 64 * n, hipMemcpyHostToDevice, c->stream), "H2D signatures");
        } else HS_HIP(hipMemcpyAsync(c->d_in, c->h_in, in_bytes, hipMemcpyHostToDevice, c->stream), "H2D");
    }
    if (ks) rc = hs_ed25519_verify_device_cached(c, ks, p_dg, digest_stride, (const uint32_t *)p_k, p_sg, n, mode, outp, nullptr, c->stream);
    else rc = hs_ed25519_verify_device(c, p_dg, digest_stride, p_k, p_sg, n, mode, outp, nullptr, c->stream);
    if (rc) return rc;
    if (!job.zc) HS_HIP(hipMemcpyAsync(c->h_out, c->d_out, bitmap_bytes(n), hipMemcpyDeviceToHost, c->stream), "D2H");
    return 0;
}
// returns 1 / 0 / <0; always gives the context back.  bitmap (optional): ceil(n/8) bytes.
int hs::host_finish(host_job &job, uint8_t *bitmap) {
    hs_ed25519_ctx *c = job.c;
    if (!c) return fail(HS_ED25519_E_ARG, "host_finish: nothing submitted");
    job.c = nullptr;
    struct putter { hs_ed25519_ctx *c; const hs_ed25519_keyset *ks; ~putter() { hs::pool_put(c); hs::ks_release(ks); } } guard{c, job.ks};
    job.ks = nullptr;
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    HS_HIP(hipStreamSynchronize(c->stream), "stream sync");
    if (bitmap) memcpy(bitmap, c->h_out, (job.n + 7) / 8);
    return bitmap_all_ones(c->h_out, job.n) ? 1 : 0;
}
void hs::host_abandon(host_job &job) {
    if (!job.c) return;
    (void)hipSetDevice(job.c->device);
    (void)hipStreamSynchronize(job.c->stream);
    pool_put(job.c);
    ks_release(job.ks);
    job.c = nullptr; job.ks = nullptr;
}
const uint8_t *hs::host_job_bitmap(const host_job &job) { return job.c ? job.c->h_out : nullptr; }

// ---- small host calls: several per launch ------------------------------------------------------
// One 1,024-signature batch is one wavefront per SIMD of its context's CU slice, and a lone
// wavefront issues the 64-bit multiply-adds at half the rate two co-resident ones reach; the streams
// cannot be multiplied instead (16 hardware queues).  So the small host calls (n <= kPartitionMaxBatch)
// of one BURST - consecutive submits of one thread, all but the last flagged HS_ED25519_SUBMIT_MORE -
// form a GROUP: one pool context, one launch (hs_ed25519_verify_device_batches[_cached], which picks its kernel by the
// group's size: one signature per quad, the pairs form, or the one-lane pair from eight members),
// each member with its own slice of the pinned staging and its own bitmap.  A group launches when a
// member arrives without the flag, when it is full, when a call of another kind follows, or when any
// of its tickets is waited on or polled - never by a timer, and never later than that: a lone
// synchronous call launches at once, exactly as before.  Groups are per thread (a burst is not
// diluted by other threads' calls); tickets may be waited on from any thread.
static const unsigned kGroupMax = 16;       // = HS_MAX_BATCHES_PER_LAUNCH: one launch of hs_ed25519_verify_device_batches[_cached]
struct host_group {
    std::mutex mu;
    std::condition_variable cv;
    hs_ed25519_ctx *c = nullptr;
    const hs_ed25519_keyset *ks = nullptr;
    int slot = 0;
    // all under mu
    unsigned members = 0, staged = 0;
    unsigned refs = 0;            // members not yet collected + the submitting thread's reference to its open group: the struct lives while > 0
    unsigned member_refs = 0;     // members not yet collected: the context and the key set go back when the last one is (the
                                  // thread's reference used to pin both until that thread's next submit - a key set could not be destroyed)
    bool closed = false, launched = false;
    int launch_rc = 0;
    hs_ed25519_batch desc[kGroupMax];
};
static const size_t kGroupInStride = 128 * kPartitionMaxBatch;       // staging bytes per member
static size_t group_out_stride() { return bitmap_bytes(kPartitionMaxBatch); }

// no more members; then on its stream (idempotent; waits until every member's bytes are staged)
static int group_close_and_launch(host_group *g) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->closed = true;
    if (g->launched || !g->c) return g->launch_rc;       // (no context: every member was collected - the thread's stale reference)
    g->cv.wait(lk, [g] { return g->staged == g->members || g->launched; });
    if (g->launched) return g->launch_rc;
    int rc = 0;
    if (hipSetDevice(g->c->device) != hipSuccess) rc = fail(HS_ED25519_E_DEVICE, "hipSetDevice");
    else if (g->ks) rc = hs_ed25519_verify_device_batches_cached(g->c, g->ks, g->desc, g->members, g->c->stream);
    else rc = hs_ed25519_verify_device_batches(g->c, g->desc, g->members, g->c->stream);
    if (!rc && hipEventRecord(g->c->done, g->c->stream) != hipSuccess) rc = fail(HS_ED25519_E_DEVICE, "hipEventRecord");
    g->launch_rc = rc;
    g->launched = true;
    g->cv.notify_all();
    return rc;
}
// member: a member's reference (its verdict was collected, or its submit failed); else the submitting thread's.  A group whose
// last member has been collected is closed (group_finish closes before it reads): nothing can join it, so what it holds - the pool
// context, the hold on the key set - goes back at once, whatever thread-local pointer still names the struct.
static void group_unref(host_group *g, bool member) {
    bool last;
    hs_ed25519_ctx *c = nullptr;
    const hs_ed25519_keyset *ks = nullptr;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        if (member && --g->member_refs == 0) g->closed = true;
        last = --g->refs == 0;
        if ((last || (g->member_refs == 0 && g->closed)) && g->c) { c = g->c; ks = g->ks; g->c = nullptr; g->ks = nullptr; }
    }
    if (c) { pool_put(c); ks_release(ks); }
    if (last) delete g;
}
// this thread's open group (it holds a reference); a thread that exits launches what it left open
struct open_group_slot {
    host_group *g = nullptr;
    void drop(bool launch) { if (g) { if (launch) (void)group_close_and_launch(g); group_unref(g, false); g = nullptr; } }
    ~open_group_slot() { drop(true); }
};
static thread_local open_group_slot t_open;

// Host calls by key bytes can run on any device slot: with several GPUs open they go round the slots, one launch
// (group or big call) at a time - the single node process of the reference then uses every GPU without sharding its
// small certificates.  Calls through a key set run where the key set lives.
static std::atomic<unsigned> g_cold_rr{0};
static int next_cold_slot() {
    const int n = hs::dev_slots();
    return n > 1 ? (int)(g_cold_rr.fetch_add(1, std::memory_order_relaxed) % (unsigned)n) : 0;
}
// join this thread's open group or start one; stage; launch unless `more`.  *idx_out = the member's position.
static int group_submit(host_group **g_out, unsigned *idx_out, const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride,
                        const void *keys, const uint8_t *sigs, size_t n, uint32_t mode, bool more, bool nowait) {
    host_group *g = t_open.g;
    unsigned idx = 0;
    bool closer = false;
    if (g) {
        bool joined = false;
        {
            std::lock_guard<std::mutex> lk(g->mu);
            if (!g->closed && g->ks == ks && g->members < kGroupMax) {      // same kind (a key set pins its device slot, cold calls take any)
                idx = g->members++; g->refs++; g->member_refs++; joined = true;
                if (!more || g->members == kGroupMax) { g->closed = true; closer = true; }
            }
        }
        if (!joined) { t_open.drop(true); g = nullptr; }       // waited on meanwhile, or another kind of call: that group goes now
    }
    if (!g) {
        const int slot = ks ? ks->slot : next_cold_slot();
        hs_ed25519_ctx *c = nullptr;
        int rc = pool_get(&c, slot, n, nowait);
        if (rc) return rc;
        rc = hipSetDevice(c->device) == hipSuccess ? 0 : fail(HS_ED25519_E_DEVICE, "hipSetDevice");
        if (!rc) rc = ctx_reserve(c, kGroupMax * kPartitionMaxBatch, true);
        if (!rc && !c->done && hipEventCreateWithFlags(&c->done, hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess)
            rc = fail(HS_ED25519_E_DEVICE, "hipEventCreate");
        if (!rc && !(g = new (std::nothrow) host_group())) rc = fail(HS_ED25519_E_NOMEM, "out of memory");
        if (rc) { pool_put(c); return rc; }
        g->c = c; g->ks = ks; g->slot = slot; ks_hold(ks);      // released when the last member's reference goes (group_unref)
        g->members = 1; g->refs = 1; g->member_refs = 1; idx = 0;
        if (more) { g->refs++; t_open.g = g; }                  // the thread's reference to its open group
        else { g->closed = true; closer = true; }
    }
    // this member's slice of the pinned staging: [digests][keys: 32n bytes or 4n index bytes][sigs: 64n]
    hs_ed25519_ctx *c = g->c;
    uint8_t *in = c->h_in + idx * kGroupInStride, *outp = c->h_out + idx * group_out_stride();
    const size_t dg_bytes = digest_stride ? 32 * n : 32, key_bytes = ks ? 4 * n : 32 * n;
    const size_t off_k = round_up(dg_bytes, 16), off_sig = round_up(off_k + key_bytes, 16);
    const bool own = caller_pinned(ks, digests, digest_stride, keys, sigs, n);     // caller-owned pinned buffers: no staging copy
    if (!own) {
        memcpy(in, digests, dg_bytes);
        memcpy(in + off_k, keys, key_bytes);
        memcpy(in + off_sig, sigs, 64 * n);
    }
    memset(outp, 0, bitmap_bytes(n));
    {
        std::lock_guard<std::mutex> lk(g->mu);
        hs_ed25519_batch &d = g->desc[idx];
        d.digest_stride = digest_stride; d.n = n;
        if (own) { d.d_digests = digests; d.d_pks = (const uint8_t *)keys; d.d_sigs = sigs; }
        else { d.d_digests = in; d.d_pks = in + off_k; d.d_sigs = in + off_sig; }
        d.mode = mode == HS_ED25519_STRICT ? HS_ED25519_STRICT : HS_ED25519_BATCH;   // BATCH_RLC this small is the per-signature BATCH law
        d.d_bitmap = outp; d.d_nreject = nullptr;
        g->staged++;
        g->cv.notify_all();
    }
    *g_out = g; *idx_out = idx;
    if (closer) {
        if (t_open.g == g) t_open.drop(false);
        const int rc = group_close_and_launch(g);
        if (rc) { group_unref(g, true); *g_out = nullptr; return rc; }
    }
    return 0;
}
// verdict of member idx; always drops the member's reference
static int group_finish(host_group *g, unsigned idx, size_t n, uint8_t *bitmap) {
    int rc = group_close_and_launch(g);
    if (!rc && hipSetDevice(g->c->device) != hipSuccess) rc = fail(HS_ED25519_E_DEVICE, "hipSetDevice");
    if (!rc) { const hipError_t e = hipEventSynchronize(g->c->done); if (e != hipSuccess) rc = fail(HS_ED25519_E_DEVICE, hipGetErrorString(e)); }
    int verdict = rc;
    if (!rc) {
        const uint8_t *outp = g->c->h_out + idx * group_out_stride();
        if (bitmap) memcpy(bitmap, outp, (n + 7) / 8);
        verdict = bitmap_all_ones(outp, n) ? 1 : 0;
    } else if (g->launched) (void)hipStreamSynchronize(g->c->stream);      // nothing of this group may still run when the context goes back
    group_unref(g, true);
    return verdict;
}
static bool group_sized(size_t n, uint32_t mode) { return n && n <= kPartitionMaxBatch && mode <= HS_ED25519_BATCH_RLC; }

static int verify_host(const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride, const void *keys,
                       const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *bitmap) {
    if (mode > HS_ED25519_BATCH_RLC) return fail(HS_ED25519_E_ARG, "verify: bad mode");
    if (n == 0) return g_init.load() ? 1 : fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!digests || !keys || !sigs) return fail(HS_ED25519_E_ARG, "verify: null pointer");
    if (group_sized(n, mode)) {
        host_group *g = nullptr; unsigned idx = 0;
        const int rc = group_submit(&g, &idx, ks, digests, digest_stride, keys, sigs, n, mode, false, false);
        if (rc) return rc;
        return group_finish(g, idx, n, bitmap);
    }
    t_open.drop(true);          // a burst left open launches before a call of another kind (the documented rule)
    host_job job;
    int rc = host_submit(job, ks ? ks->slot : next_cold_slot(), ks, digests, digest_stride, keys, sigs, n, mode);
    if (rc) { host_abandon(job); return rc; }
    return host_finish(job, bitmap);
}

// ---- the same, asynchronous: submit now, collect the verdict later -------------------------------
// A ticket owns a pool context (or a share of a group's) from submit to wait.  The reference is a
// tokio program: its verify_batch call sites sit on async tasks (consensus/src/core.rs, mempool),
// so the binding can submit from the task and await the ticket on a blocking-pool thread - or keep
// several certificates in flight from one thread, which is what fills the GPU.
struct hs_ed25519_ticket { host_job job; host_group *g = nullptr; unsigned idx = 0; size_t n = 0; };
extern "C" int hs_ed25519_submit(hs_ed25519_ticket **out, const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride,
                                 const void *pks_or_idx, const uint8_t *sigs, size_t n, uint32_t mode, uint32_t flags) {
    if (!out) return fail(HS_ED25519_E_ARG, "submit: null ticket slot");
    *out = nullptr;
    if (mode > HS_ED25519_BATCH_RLC || (flags & ~(uint32_t)HS_ED25519_SUBMIT_MORE)) return fail(HS_ED25519_E_ARG, "submit: bad mode / flags");
    if (digest_stride != 0 && digest_stride != 32) return fail(HS_ED25519_E_ARG, "submit: digest_stride must be 0 or 32");
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (n && (!digests || !pks_or_idx || !sigs)) return fail(HS_ED25519_E_ARG, "submit: null pointer");
    hs_ed25519_ticket *t = new (std::nothrow) hs_ed25519_ticket();
    if (!t) return fail(HS_ED25519_E_NOMEM, "submit: out of memory");
    t->n = n;
    int rc = 0;
    if (group_sized(n, mode)) rc = group_submit(&t->g, &t->idx, ks, digests, digest_stride, pks_or_idx, sigs, n, mode, flags & HS_ED25519_SUBMIT_MORE, true);
    else if (n) {
        t_open.drop(true);      // a burst left open launches before a call of another kind (the documented rule)
        rc = host_submit(t->job, ks ? ks->slot : next_cold_slot(), ks, digests, digest_stride, pks_or_idx, sigs, n, mode, true);
        if (rc) host_abandon(t->job);
    }
    if (rc) { delete t; return rc; }
    *out = t;
    return 0;
}
extern "C" int hs_ed25519_ticket_ready(hs_ed25519_ticket *t) {
    if (!t) return fail(HS_ED25519_E_ARG, "ticket_ready: null ticket");
    hs_ed25519_ctx *c = t->g ? t->g->c : t->job.c;
    if (!c) return 1;
    if (t->g) { const int rc = group_close_and_launch(t->g); if (rc) return 1; }      // the error is reported by wait
    if (hipSetDevice(c->device) != hipSuccess) return fail(HS_ED25519_E_DEVICE, "hipSetDevice");
    const hipError_t e = t->g ? hipEventQuery(c->done) : hipStreamQuery(c->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(HS_ED25519_E_DEVICE, hipGetErrorString(e));
}
extern "C" int hs_ed25519_ticket_device_slot(const hs_ed25519_ticket *t) {
    if (!t) return fail(HS_ED25519_E_ARG, "ticket_device_slot: null ticket");
    const hs_ed25519_ctx *c = t->g ? t->g->c : t->job.c;
    return c ? c->slot : 0;
}
extern "C" int hs_ed25519_ticket_wait(hs_ed25519_ticket *t, uint8_t *bitmap) {
    if (!t) return fail(HS_ED25519_E_ARG, "ticket_wait: null ticket");
    int rc = 1;                                                        // an empty batch is Ok (crypto/src/lib.rs:212-219 on no votes)
    if (t->g) rc = group_finish(t->g, t->idx, t->n, bitmap);
    else if (t->job.c) rc = host_finish(t->job, bitmap);
    delete t;
    return rc;
}

extern "C" int hs_ed25519_verify_shared(const uint8_t digest[32], const uint8_t *pks, const uint8_t *sigs,
                                        size_t n, uint32_t mode, uint8_t *bitmap) {
    return verify_host(nullptr, digest, 0, pks, sigs, n, mode, bitmap);
}
extern "C" int hs_ed25519_verify_multi(const uint8_t *digests, const uint8_t *pks, const uint8_t *sigs,
                                       size_t n, uint32_t mode, uint8_t *bitmap) {
    return verify_host(nullptr, digests, 32, pks, sigs, n, mode, bitmap);
}
// host-pointer forms of the cached path: key_idx[i] indexes the key set
extern "C" int hs_ed25519_verify_shared_cached(const hs_ed25519_keyset *ks, const uint8_t digest[32], const uint32_t *key_idx,
                                               const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *bitmap) {
    if (!ks) return fail(HS_ED25519_E_ARG, "verify: null key set");
    return verify_host(ks, digest, 0, key_idx, sigs, n, mode, bitmap);
}
extern "C" int hs_ed25519_verify_multi_cached(const hs_ed25519_keyset *ks, const uint8_t *digests, const uint32_t *key_idx,
                                              const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *bitmap) {
    if (!ks) return fail(HS_ED25519_E_ARG, "verify: null key set");
    return verify_host(ks, digests, 32, key_idx, sigs, n, mode, bitmap);
}

// TC::verify from host buffers: the per-vote digests SHA-512(round || high_qc_round)[..32]
// (consensus/src/messages.rs:307-311) are built on the device; strict semantics.
// ks == NULL: keys by bytes (pks_or_idx = n*32 key bytes); else pks_or_idx = n uint32 indices.
static int verify_host_tc(const hs_ed25519_keyset *ks, uint64_t round, const uint64_t *hq_rounds, const void *pks_or_idx,
                          const uint8_t *sigs, size_t n, uint8_t *bitmap) {
    if (n == 0) return g_init.load() ? 1 : fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!hq_rounds || !pks_or_idx || !sigs) return fail(HS_ED25519_E_ARG, "verify_tc: null pointer");
    hs_ed25519_ctx *c = nullptr;
    int rc = pool_get(&c, ks ? ks->slot : 0, n);
    if (rc) return rc;
    // (an early return below may leave work on the stream: the guard drains it before the context and the key set are let go)
    struct putter { hs_ed25519_ctx *c; const hs_ed25519_keyset *ks; bool done;
                    ~putter() { if (!done) (void)hipStreamSynchronize(c->stream); hs::pool_put(c); hs::ks_release(ks); } } guard{c, ks, false};
    ks_hold(ks);
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    const size_t hint = c->narrow ? kPartitionMaxBatch : g_max_batch.load();
    rc = ctx_reserve(c, n > hint ? n : hint, true);
    if (rc) return rc;
    // staging (128 B per signature available): [hq: 8n][keys: 32n or 4n][sigs: 64n]
    const size_t key_bytes = ks ? 4 * n : 32 * n;
    const size_t off_k = round_up(8 * n, 16), off_sig = round_up(off_k + key_bytes, 16), in_bytes = off_sig + 64 * n;
    memcpy(c->h_in, hq_rounds, 8 * n);
    memcpy(c->h_in + off_k, pks_or_idx, key_bytes);
    memcpy(c->h_in + off_sig, sigs, 64 * n);
    const bool zc = n <= kZeroCopyMax;
    uint8_t *in = zc ? c->h_in : c->d_in, *outp = zc ? c->h_out : c->d_out;
    if (zc) memset(c->h_out, 0, bitmap_bytes(n));
    else {
        HS_HIP(hipMemsetAsync(c->d_out, 0, bitmap_bytes(n), c->stream), "clear bitmap");
        HS_HIP(hipMemcpyAsync(c->d_in, c->h_in, in_bytes, hipMemcpyHostToDevice, c->stream), "H2D");
    }
    if (ks)
        rc = hs_ed25519_verify_device_tc_cached(c, ks, round, (const uint64_t *)in, (const uint32_t *)(in + off_k),
                                                in + off_sig, n, outp, nullptr, c->stream);
    else
        rc = hs_ed25519_verify_device_tc(c, round, (const uint64_t *)in, in + off_k, in + off_sig, n, outp, nullptr, c->stream);
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    if (!zc) HS_HIP(hipMemcpyAsync(c->h_out, c->d_out, bitmap_bytes(n), hipMemcpyDeviceToHost, c->stream), "D2H");
    HS_HIP(hipStreamSynchronize(c->stream), "stream sync");
    guard.done = true;
    if (bitmap) memcpy(bitmap, c->h_out, (n + 7) / 8);
    return bitmap_all_ones(c->h_out, n) ? 1 : 0;
}
extern "C" int hs_ed25519_verify_tc(uint64_t round, const uint64_t *high_qc_rounds, const uint8_t *pks, const uint8_t *sigs,
                                    size_t n, uint8_t *bitmap) {
    return verify_host_tc(nullptr, round, high_qc_rounds, pks, sigs, n, bitmap);
}
extern "C" int hs_ed25519_verify_tc_cached(const hs_ed25519_keyset *ks, uint64_t round, const uint64_t *high_qc_rounds,
                                           const uint32_t *key_idx, const uint8_t *sigs, size_t n, uint8_t *bitmap) {
    if (!ks) return fail(HS_ED25519_E_ARG, "verify_tc_cached: null key set");
    return verify_host_tc(ks, round, high_qc_rounds, key_idx, sigs, n, bitmap);
}

// ---- round ring: streamed rounds through pinned slots, one hipGraph launch per round -----------
// BASELINE config 5.  A slot owns a stream, pinned host staging and device buffers.  The caller
// writes a round's packed tuples straight into the slot's pinned memory (begin), submits it
// (one hipGraphLaunch: H2D copy -> verification kernel(s) -> D2H copy of the bitmap) and collects
// the verdict later (wait / poll); slots in flight overlap their copies and kernels on separate
// streams.  The graph of a slot is captured once and re-launched while (n, digest form, mode) stay
// the same - the steady state of a fixed committee, whose QCs carry exactly 2f+1 votes - and is
// updated in place (hipGraphExecUpdate) when they change.
struct ring_slot {
    hs_ed25519_ctx *ctx = nullptr;
    hipGraphExec_t exec = nullptr;
    size_t g_n = 0, g_stride = 0; uint32_t g_mode = 0;                      // what `exec` was captured for
    size_t n = 0, stride = 0, off_key = 0, off_sig = 0, in_bytes = 0;       // the round begun in this slot
    bool begun = false, inflight = false;
};
struct hs_ed25519_ring {
    std::vector<ring_slot> slots;
    size_t cap = 0;
    const hs_ed25519_keyset *ks = nullptr;
    bool use_graph = false;
    bool zero_copy = false;      // kernels read the pinned slot / write the pinned bitmap directly over PCIe
    size_t captures = 0, launches = 0;
};

extern "C" void hs_ed25519_ring_destroy(hs_ed25519_ring *r) {
    if (!r) return;
    for (ring_slot &s : r->slots) {
        if (s.ctx) { (void)hipSetDevice(s.ctx->device); (void)hipStreamSynchronize(s.ctx->stream); }
        if (s.exec) (void)hipGraphExecDestroy(s.exec);
        if (s.ctx) hs_ed25519_ctx_destroy(s.ctx);
    }
    ks_release(r->ks);
    delete r;
}

extern "C" int hs_ed25519_ring_create(hs_ed25519_ring **out, size_t slot_batch, unsigned slots,
                                      const hs_ed25519_keyset *ks, uint32_t flags) {
    if (!out || slot_batch == 0 || slots == 0 || slots > 64 || slot_batch > 0x3fffffffu || flags > 3u)
        return fail(HS_ED25519_E_ARG, "ring_create: bad arguments");
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    hs_ed25519_ring *r = new (std::nothrow) hs_ed25519_ring();
    if (!r) return fail(HS_ED25519_E_NOMEM, "ring_create: out of memory");
    r->cap = round_up(slot_batch, 64); r->ks = ks; ks_hold(ks);      // for the ring's life: its rounds read the set's tables
    // AUTO follows the measurements (profiles/r1_ring_sweep.md): zero-copy wins or ties up to QC-sized rounds of a
    // 10,000-validator committee; above that the copy engines' bulk transfers are as good and plain stream
    // operations overlap best.  The graph form never won on gfx950 / ROCm 7 and is kept selectable.
    if (flags == HS_ED25519_RING_AUTO) flags = r->cap <= kZeroCopyMax ? HS_ED25519_RING_ZERO_COPY : HS_ED25519_RING_STREAM_OPS;
    r->zero_copy = flags == HS_ED25519_RING_ZERO_COPY;
    r->use_graph = flags == HS_ED25519_RING_GRAPH;
    r->slots.resize(slots);
    for (ring_slot &s : r->slots) {
        int rc = hs_ed25519_ctx_create_on(&s.ctx, 64, ks ? ks->slot : 0);
        if (!rc) rc = ctx_reserve(s.ctx, r->cap, true);      // all allocation happens here, none under capture
        if (!rc) s.ctx->several_in_flight = slots > 1;
        if (rc) { hs_ed25519_ring_destroy(r); return rc; }
    }
    *out = r;
    return 0;
}

extern "C" int hs_ed25519_ring_begin(hs_ed25519_ring *r, unsigned slot, size_t n, size_t digest_stride,
                                     uint8_t **digests, uint8_t **keys, uint8_t **sigs) {
    if (!r || slot >= r->slots.size() || !digests || !keys || !sigs) return fail(HS_ED25519_E_ARG, "ring_begin: bad arguments");
    if (n == 0 || n > r->cap) return fail(HS_ED25519_E_ARG, "ring_begin: n must be in [1, slot_batch]");
    if (digest_stride != 0 && digest_stride != 32) return fail(HS_ED25519_E_ARG, "ring_begin: bad digest_stride");
    ring_slot &s = r->slots[slot];
    if (s.inflight) return fail(HS_ED25519_E_ARG, "ring_begin: slot still in flight (wait for it first)");
    // one contiguous transfer: [digests][keys: 32n bytes, or 4n index bytes with a key set][sigs: 64n]
    const size_t dg_bytes = digest_stride ? 32 * n : 32;
    s.n = n; s.stride = digest_stride;
    s.off_key = round_up(dg_bytes, 16);
    s.off_sig = round_up(s.off_key + (r->ks ? 4 : 32) * n, 16);
    s.in_bytes = s.off_sig + 64 * n;
    s.begun = true;
    *digests = s.ctx->h_in; *keys = s.ctx->h_in + s.off_key; *sigs = s.ctx->h_in + s.off_sig;
    return 0;
}

static int ring_enqueue(hs_ed25519_ring *r, ring_slot &s, uint32_t mode) {
    hs_ed25519_ctx *c = s.ctx;
    // zero-copy: hipHostMalloc memory is mapped into the device's address space, so the kernels can
    // take the slot's pinned buffers as they are - every input byte is read exactly once anyway
    uint8_t *in = r->zero_copy ? c->h_in : c->d_in, *outp = r->zero_copy ? c->h_out : c->d_out;
    if (!r->zero_copy) {
        HS_HIP(hipMemsetAsync(c->d_out, 0, bitmap_bytes(s.n), c->stream), "ring clear bitmap");
        HS_HIP(hipMemcpyAsync(c->d_in, c->h_in, s.in_bytes, hipMemcpyHostToDevice, c->stream), "ring H2D");
    }
    int rc;
    if (r->ks)
        rc = hs_ed25519_verify_device_cached(c, r->ks, in, s.stride, (const uint32_t *)(in + s.off_key),
                                             in + s.off_sig, s.n, mode, outp, nullptr, c->stream);
    else
        rc = hs_ed25519_verify_device(c, in, s.stride, in + s.off_key, in + s.off_sig, s.n, mode,
                                      outp, nullptr, c->stream);
    if (rc) return rc;
    if (!r->zero_copy)
        HS_HIP(hipMemcpyAsync(c->h_out, c->d_out, bitmap_bytes(s.n), hipMemcpyDeviceToHost, c->stream), "ring D2H");
    return 0;
}

extern "C" int hs_ed25519_ring_submit(hs_ed25519_ring *r, unsigned slot, uint32_t mode) {
    if (!r || slot >= r->slots.size() || mode > HS_ED25519_BATCH_RLC) return fail(HS_ED25519_E_ARG, "ring_submit: bad arguments");
    ring_slot &s = r->slots[slot];
    if (!s.begun || s.inflight) return fail(HS_ED25519_E_ARG, "ring_submit: slot not begun / already in flight");
    HS_HIP(hipSetDevice(s.ctx->device), "hipSetDevice");
    if (mode == HS_ED25519_BATCH_RLC) {      // the MSM workspace is allocated here, never under graph capture
        int rc = rlc_reserve(s.ctx, s.n);
        if (rc) return rc;
        if (!dev_base_table(s.ctx->slot)) return HS_ED25519_E_DEVICE;
    }
    hipStream_t st = s.ctx->stream;
    // the verdict is read from the pinned bitmap: clear it so that a round whose kernels did not
    // write it can never be read as accepted (the zero-copy kernels write this memory directly)
    memset(s.ctx->h_out, 0, bitmap_bytes(s.n));
    if (!r->use_graph) {
        int rc = ring_enqueue(r, s, mode);
        if (rc) return rc;
    } else {
        if (!s.exec || s.g_n != s.n || s.g_stride != s.stride || s.g_mode != mode) {
            // Thread-local capture: calls of other threads are not recorded and the "unsafe" ones (allocation, stream and event
            // creation, work on their own streams) leave it alone.  NOT every call does, on ROCm 7.0: a hipDeviceSynchronize or a
            // legacy-stream hipMemcpy / hipMemset issued by any thread while this capture is open invalidates it ("operation
            // failed due to a previous error during capture") or crashes inside the runtime - tools/capture_probe.hip,
            // profiles/r5_h_capture_probe.json.  The engine issues neither after hs_ed25519_init (round 5: key-set registration
            // left the legacy stream, hs_ed25519_keyset_destroy lost its device-wide synchronisation); a host that does must not
            // select the graph form (include/hs_ed25519.h; AUTO never does).
            hipGraph_t g = nullptr;
            HS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
            int rc = ring_enqueue(r, s, mode);
            hipError_t e = hipStreamEndCapture(st, &g);
            if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
            if (e != hipSuccess) return fail(HS_ED25519_E_DEVICE, "hipStreamEndCapture", e);
            if (s.exec) {                       // same topology unless the kernel choice changed: try the in-place update
                hipGraphNode_t bad = nullptr;
                hipGraphExecUpdateResult res;
                if (hipGraphExecUpdate(s.exec, g, &bad, &res) != hipSuccess) {
                    (void)hipGetLastError();      // a refused update is handled here: do not leave it as the thread's sticky error
                    (void)hipGraphExecDestroy(s.exec); s.exec = nullptr;
                }
            }
            if (!s.exec) {
                e = hipGraphInstantiate(&s.exec, g, nullptr, nullptr, 0);
                if (e != hipSuccess) { (void)hipGraphDestroy(g); s.exec = nullptr; return fail(HS_ED25519_E_DEVICE, "hipGraphInstantiate", e); }
            }
            (void)hipGraphDestroy(g);
            s.g_n = s.n; s.g_stride = s.stride; s.g_mode = mode;
            r->captures++;
        }
        HS_HIP(hipGraphLaunch(s.exec, st), "hipGraphLaunch");
    }
    r->launches++;
    s.inflight = true; s.begun = false;
    return 0;
}

// 1 = the slot's round has completed (or nothing is in flight), 0 = still running
extern "C" int hs_ed25519_ring_poll(hs_ed25519_ring *r, unsigned slot) {
    if (!r || slot >= r->slots.size()) return fail(HS_ED25519_E_ARG, "ring_poll: bad arguments");
    ring_slot &s = r->slots[slot];
    if (!s.inflight) return 1;
    hipError_t e = hipStreamQuery(s.ctx->stream);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) return 0;
    return fail(HS_ED25519_E_DEVICE, "hipStreamQuery", e);
}

extern "C" int hs_ed25519_ring_wait(hs_ed25519_ring *r, unsigned slot, uint8_t *bitmap) {
    if (!r || slot >= r->slots.size()) return fail(HS_ED25519_E_ARG, "ring_wait: bad arguments");
    ring_slot &s = r->slots[slot];
    if (!s.inflight) return fail(HS_ED25519_E_ARG, "ring_wait: nothing submitted in this slot");
    hipError_t e = hipStreamSynchronize(s.ctx->stream);
    s.inflight = false;
    if (e != hipSuccess) return fail(HS_ED25519_E_DEVICE, "ring stream sync", e);
    if (bitmap) memcpy(bitmap, s.ctx->h_out, (s.n + 7) / 8);
    return bitmap_all_ones(s.ctx->h_out, s.n) ? 1 : 0;
}

extern "C" hs_ed25519_ctx *hs_ed25519_ring_slot_ctx(hs_ed25519_ring *r, unsigned slot) {
    if (!r || slot >= r->slots.size()) { (void)fail(HS_ED25519_E_ARG, "ring_slot_ctx: bad arguments"); return nullptr; }
    return r->slots[slot].ctx;
}

extern "C" int hs_ed25519_ring_stats(const hs_ed25519_ring *r, size_t *graph_captures, size_t *launches) {
    if (!r || !graph_captures || !launches) return fail(HS_ED25519_E_ARG, "ring_stats: null");
    *graph_captures = r->captures; *launches = r->launches;
    return 0;
}
