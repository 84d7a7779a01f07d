// engine.h - internal host-side declarations shared by the translation units of libhs_ed25519.so
// (hs_ed25519.hip: kernels of the per-signature paths + init, contexts, key sets, device-pointer API;
//  hs_host.hip: context pools, host-pointer API (synchronous, tickets), round ring;
//  hs_rlc.hip: the batch random-linear-combination MSM path; hs_shard.hip: multi-GPU fan-out;
//  hs_wire.hip: wire decode + batch signing).  Not installed; the public surface is include/hs_ed25519.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <vector>
#include "verify.h"
#include "../../include/hs_ed25519.h"

namespace hs {

// ---- errors -----------------------------------------------------------------------------------
int hs_fail(int code, const char *what, hipError_t e = hipSuccess);
#define HS_HIP(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hs::hs_fail(HS_ED25519_E_DEVICE, what, e_); } while (0)

inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
inline size_t bitmap_bytes(size_t n) { return 8 * ((n + 63) / 64); }
inline bool misaligned(const void *p, size_t a) { return ((uintptr_t)p & (a - 1)) != 0; }
bool bitmap_all_ones(const uint8_t *bm, size_t n);

// ---- devices ----------------------------------------------------------------------------------
// A "slot" is one entry of the device list given to hs_ed25519_init_devices; slot 0 is the primary
// device (what hs_ed25519_init(device) selects).  The same HIP ordinal may appear in several slots
// (tests shard one GPU into several virtual devices).
int dev_slots();                    // 0 before init
int dev_ordinal(int slot);          // HIP ordinal of a slot
const u32 *dev_base_table(int slot);// the base point's window table on that device (built on first use), or nullptr on error
size_t max_batch_hint();
bool engine_ready();

}  // namespace hs

// kernel choices of a context (hs_ed25519_ctx_set_kernel)
enum { HS_K_AUTO = 0, HS_K_LANE = 1, HS_K_FUSED = 3, HS_K_FUSED_PAIRS = 4,
       HS_K_CACHED_C = 5,         // key cache: chains + batched finish by compression at any size (cold calls: as HS_K_AUTO)
       HS_K_OCT = 6, HS_K_OCT_SPLIT = 7 };   // cold single calls of <= 8 signatures per SIMD: k_verify_fused8 / k_verify_fused8h whatever the autotune chose

// selectors that only concern cold single calls of up to 8 signatures per SIMD behave as HS_K_AUTO everywhere else
inline bool hs_k_auto_like(int k) { return k == HS_K_AUTO || k == HS_K_OCT || k == HS_K_OCT_SPLIT; }

namespace hs {
// a stream that several pool contexts launch on (the narrow pool: kNarrowDepth contexts - each with its own staging, workspace
// and completion event - queue on one CU-masked stream, so that a slice's next launch is already queued when one ends)
struct stream_box { hipStream_t s = nullptr; int device = 0; ~stream_box(); };
}
struct hs_ed25519_ctx {
    int slot = 0, device = 0;
    hipStream_t stream = nullptr;
    std::shared_ptr<hs::stream_box> stream_share;   // set: `stream` belongs to this box (destroyed with its last context), not to the context
    unsigned lane = 0;            // narrow pool: which of the pool's streams
    hipEvent_t done = nullptr;    // host-call groups: recorded behind the launch
    bool narrow = false;          // a pool context whose stream is confined to one slice of the CUs (small host-API calls)
    unsigned part = 0;            // which slice (narrow pool contexts)
    unsigned cus = 0;             // compute units the context's own stream may use (the whole device unless the stream is CU-masked):
                                  // what the kernel choice sizes "one wavefront per SIMD" from (a tenant's share, not the chip)
    bool overflow = false;        // lent over the pool's cap (pool_get's escape from a self-deadlock): destroyed on return, never pooled
    std::shared_ptr<std::atomic<int>> held;   // taken for a TICKET: the submitting thread's count of contexts it has out (pool_put drops it)
    size_t cap = 0;               // signatures
    hs::u32 *d_ws = nullptr;      // workspace of the two-kernel cold path
    uint8_t *d_in = nullptr;      // digests | keys | sigs   (host API only)
    uint8_t *d_out = nullptr;     // bitmap (8*ceil(cap/64)) | 16 spare bytes
    uint8_t *h_in = nullptr;      // pinned mirror of d_in
    uint8_t *h_out = nullptr;     // pinned mirror of d_out
    hs::u32 *d_tc = nullptr;      // timeout-certificate digests built by k_tc_digests (tc_cap signatures)
    size_t tc_cap = 0;
    uint8_t *d_wire = nullptr;    // hs_wire.hip: decoded (pk | sig | round) of wire frames, wire_cap signatures
    size_t wire_cap = 0;
    bool staging = false;
    int kernel = HS_K_AUTO;       // per-context kernel choice (tests, A/B); HS_K_AUTO = by batch size
    bool several_in_flight = false;  // other whole-chip calls run beside this context's (ring slot, busy wide pool, hs_ed25519_ctx_set_concurrent): throughput forms
    void *rlc = nullptr;          // hs_rlc.hip: this context's MSM workspace (rlc_ws *), freed by rlc_release
    // optional profiling: a ring of (start, after-prepare, after-verify) HIP events attached to the
    // kernel dispatches, read back by hs_ed25519_ctx_kernel_times after a synchronize
    std::vector<hipEvent_t> ev;
    std::vector<char> ev_mid;     // slot recorded the middle event (a two-kernel call)
    size_t ev_slots = 0, ev_calls = 0;
    size_t ev_every = 1, ev_seq = 0;   // record events on every ev_every-th call only
};

struct hs_ed25519_keyset {
    int slot = 0;
    size_t nkeys = 0;
    hs::u32 *d_tab = nullptr, *d_pk = nullptr, *d_flag = nullptr;
    hs::u32 *d_niels = nullptr;   // [key][32 words]: affine Niels form of +A (y+x, y-x, 2dxy, pad) for the RLC path
    // host-API work that names this set and has not been collected yet: jobs and groups (synchronous calls, tickets), round
    // rings (for their whole life).  hs_ed25519_keyset_destroy waits for it to drain instead of synchronising the device -
    // a device-wide synchronisation from one thread breaks a hipGraph capture in progress on another (round 5).
    mutable std::atomic<int> users{0};
};

namespace hs {
// ---- process state (defined in hs_ed25519.hip; hs_host.hip serves the host-pointer API from it) -----------------
struct dev_state {
    int ordinal = -1;
    u32 *base_tab = nullptr;                 // key-cache path: the base point's own window table (8-bit windows; also the RLC path's [b]B)
    u32 *base_tab16 = nullptr;               // the base point's 16-bit window table (verify_cached.h: 16 x 32,769 x 128 B), built from base_tab
    // the host API's pools of stream contexts on this device: [0] wide (whole chip), [1] narrow (one slice of the CU mask)
    std::vector<hs_ed25519_ctx *> free_ctx[2];
    int live[2] = {0, 0};
    unsigned part_busy[64] = {0};            // narrow contexts in flight per CU slice (pool_get balances over the slices)
    unsigned lane_busy[64] = {0};            // narrow contexts in flight per stream of the pool (an idle stream before a queued one)
    // the narrow pool's streams, by lane.  On the heap, and freed by hs_ed25519_shutdown only: a process that exits without
    // shutting the engine down leaks them - like the pooled contexts - instead of calling into the HIP runtime from a static destructor
    std::vector<std::shared_ptr<stream_box>> *lanes = nullptr;
    unsigned cus = 0;                        // compute units (for the narrow pool's CU masks)
    // hs_ed25519_autotune: kernel time of the two eight-lane cold forms on a canned 6,667-signature batch on THIS device
    // ([0] k_verify_fused8, [1] k_verify_fused8h; 0 = not measured) and the faster one (-1 = not measured: the split form)
    float oct_ms[2] = {0.f, 0.f};            // (written and read under g_mu: hs_ed25519_autotune, hs_ed25519_config_dump)
};
extern std::mutex g_mu;
extern std::condition_variable g_cv;
extern std::atomic<bool> g_init;
extern std::vector<dev_state> g_devs;
extern std::atomic<size_t> g_max_batch;
constexpr int kMaxSlots = 64;                // hs_ed25519_init_devices takes at most this many device slots
extern std::atomic<int> g_oct_pick[kMaxSlots];   // per slot: the faster eight-lane cold form as measured (0 one wavefront, 1 split, -1 not measured); read by
                                             // every cold call without a lock, written by hs_ed25519_autotune - hence atomics beside g_devs, not inside it
extern int kMaxPool[2];                      // STREAMS per pool: [0] wide, [1] narrow
extern int kNarrowDepth;                     // narrow pool: contexts per stream (launches queued per stream, each with its own staging)
extern unsigned kPoolPartitions;             // CU slices the narrow pool's streams are spread over
extern unsigned kTenants, kTenantIndex;      // this process is tenant kTenantIndex of kTenants on its GPU (hs_ed25519_set_tenancy)
extern bool kTenantWide;                     // co-tenants: confine the wide pool's streams to the tenant's CU share as well
constexpr size_t kPartitionMaxBatch = 1024;  // host calls up to this many signatures are served by the narrow pool
constexpr size_t kZeroCopyMax = 16384;       // host-pointer calls up to this size let the kernels read pinned memory directly
int ctx_new(hs_ed25519_ctx **out, size_t max_batch, int slot, const uint32_t *cu_mask, uint32_t mask_words,
            std::shared_ptr<stream_box> share = nullptr);
int ctx_reserve(hs_ed25519_ctx *c, size_t n, bool staging);
// a narrow (CU-partitioned) context for small n, a wide one above; nowait: HS_ED25519_E_BUSY instead of waiting for a free one
int pool_get(hs_ed25519_ctx **out, int slot, size_t n, bool nowait = false);
void pool_put(hs_ed25519_ctx *c);
int make_digest_src(msg_src &m, const uint8_t *d_digests, size_t digest_stride);
// per-signature cold path over n signatures, optionally skipped when *d_skip != 0 (then the bitmap is
// written all-ones: the RLC fast-accept already vouched for every signature)
int enqueue_cold(hs_ed25519_ctx *ctx, msg_src msgs, const uint8_t *d_pks, const uint8_t *d_sigs, size_t n,
                 uint32_t mode, uint8_t *d_bitmap, uint32_t *d_nreject, void *stream, const u32 *d_skip = nullptr);
int enqueue_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint32_t *d_key_idx,
                   const uint8_t *d_sigs, size_t n, uint32_t mode, uint8_t *d_bitmap, uint32_t *d_nreject, void *stream,
                   const u32 *d_skip = nullptr);
// one host-pointer verification on one device slot, split so that several devices can be in flight from one thread
struct host_job { hs_ed25519_ctx *c = nullptr; size_t n = 0; bool zc = false; const hs_ed25519_keyset *ks = nullptr; /* held from submit to finish / abandon */ };
inline void ks_hold(const hs_ed25519_keyset *ks) { if (ks) ks->users.fetch_add(1, std::memory_order_acq_rel); }
inline void ks_release(const hs_ed25519_keyset *ks) { if (ks) ks->users.fetch_sub(1, std::memory_order_acq_rel); }
// gather (optional): element p of the call is signature gather[p] of `sigs` (and digest gather[p] of `digests` when they are
// per tuple) - a routed shard is packed straight into the pinned staging, one copy instead of two; `keys` stay contiguous
int host_submit(host_job &job, int slot, const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride,
                const void *keys, const uint8_t *sigs, size_t n, uint32_t mode, bool nowait = false, const uint32_t *gather = nullptr);
int host_finish(host_job &job, uint8_t *bitmap);          // 1 / 0 / <0; returns the context to its pool
void host_abandon(host_job &job);                          // after a failed submit
const uint8_t *host_job_bitmap(const host_job &job);       // the pinned bitmap (valid until finish)
// hs_wire.hip
void wire_release(hs_ed25519_ctx *c);
int public_keys_device(hs_ed25519_ctx *ctx, const uint8_t *d_seeds, size_t n, uint8_t *d_pks, void *stream);   // n x 32-byte seeds -> n x 32-byte keys
// hs_tune.hip
int autotune_slot(int slot, double ms[2]);
// hs_rlc.hip
void rlc_release(hs_ed25519_ctx *c);
int rlc_reserve(hs_ed25519_ctx *ctx, size_t n, void *stream = nullptr);     // all allocation of an n-signature RLC call, ahead of time
int enqueue_rlc(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint8_t *d_pks,
                const uint32_t *d_key_idx, const uint8_t *d_sigs, size_t n, u32 **d_ok_flag, void *stream);
// a combination cut over several devices (hs_shard.hip): one key per call, one partial record per shard, one join
struct rlc_key_words { u32 w[8]; };
int rlc_call_key(rlc_key_words &out, bool &pinned);
int enqueue_rlc_partial(hs_ed25519_ctx *ctx, msg_src msgs, const uint8_t *d_pks, const uint8_t *d_sigs, size_t n, u32 index_base,
                        const rlc_key_words &key, u32 **d_partial, void *stream);
int enqueue_rlc_join(hs_ed25519_ctx *ctx, const u32 *d_parts, u32 count, u32 **d_flag, void *stream);
constexpr size_t kRlcPartBytes = 384;        // HS_RLC_PART_WORDS x 4 (rlc.h)
}  // namespace hs
