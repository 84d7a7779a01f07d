/* hs_ed25519.h - the EXTENDED C ABI of the MI355X batch Ed25519 verification engine.
 *
 * include/hs_ed25519_core.h (included below) holds what the reference's shim binds - the twelve functions that replace
 * crypto/src/lib.rs:200-219 - together with the data formats, verdict semantics, return values and thread-safety
 * rules, which apply to every function here as well.  This header adds everything else: several GPUs, tenancy,
 * caller-owned pinned buffers, device-pointer calls, batch lists, the round ring, tickets, sharded calls, wire
 * decode, batch signing, and the probes / read-outs the bench and the tests use.  A maintainer reviewing the
 * drop-in reads the core header; nothing in it depends on this one.
 */
#ifndef HS_ED25519_H
#define HS_ED25519_H

#include "hs_ed25519_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Several GPUs behind one process (the reference node is one process: consensus/src/messages.rs:197
 * is called from the single Core task).  `ordinals[0..count)` are HIP device ordinals; entry s becomes
 * device slot s, slot 0 being the primary device every non-sharded call uses.  The same ordinal may
 * appear more than once (virtual devices: tests shard one GPU).  Idempotent for the same list; a
 * different list after a successful init is HS_ED25519_E_ARG (shutdown first). */
int hs_ed25519_init_devices(const int *ordinals, size_t count, size_t max_batch);
int hs_ed25519_device_count(void);        /* device slots, 0 before init */
/* Several node processes on ONE GPU (the reference's `fab local` runs four node processes on one host,
 * benchmark/benchmark/local.py:94-104).  Tell the engine, before hs_ed25519_init, that this
 * process is tenant `index` of `tenants` (1..8) on its GPU: it then takes 1/tenants of the stream budget (the
 * device multiplexes every process's streams; past ~16 in flight throughput collapses) and confines its small
 * calls to its own 1/tenants of the compute units, so co-tenants do not stack their batches on the same CUs.
 * Environment HS_ED25519_TENANTS / HS_ED25519_TENANT_INDEX do the same for an unmodified host.  Verdicts are
 * identical with or without it.  The pools then want hs_ed25519_hw_queue_budget(tenants) hardware queues: the HOST
 * exports that as GPU_MAX_HW_QUEUES before the HIP runtime starts (hs_ed25519_core.h, hs_ed25519_init). */
int hs_ed25519_set_tenancy(unsigned tenants, unsigned index);
/* The GPU_MAX_HW_QUEUES value the host should export (in main(), before any thread starts and before the HIP runtime
 * initialises; hs_ed25519_init itself only sets it under HS_ED25519_SET_ENV=1) for a process that is one of `tenants` (1..8) on its GPU: 16 for a
 * process alone; for co-tenants one hardware queue per pool context + 1 (11 / 6 / 4 for 2 / 4 / 8 tenants).  The GPU's
 * hardware scheduler runs ~28 user queues at once across ALL processes; past that every process is time-sliced
 * (profiles/r4_c_queue_probe_pools_of_7.md), so co-tenants keep well below it.  Pure function, no device needed - a host whose runtime
 * is initialised before this library (PyTorch, hotstuff_amd/_lib.py) sets the variable itself from it. */
unsigned hs_ed25519_hw_queue_budget(unsigned tenants);
/* Caller-owned pinned buffers.  Memory from hs_ed25519_host_alloc is page-locked and mapped on every device
 * slot; a host-pointer call (synchronous, submit, sharded) whose digests, keys and signatures all lie in such
 * memory (16-byte aligned; key indices 4-byte) is NOT staged - the kernels read it in place over PCIe - so the
 * shim's own packing of Vec<(PublicKey, Signature)> is the only host copy.  Such buffers
 * belong to the call until its verdict has been collected; any other buffer may be reused as soon as the call
 * or hs_ed25519_submit returns.  NULL on failure (hs_ed25519_last_error). */
void *hs_ed25519_host_alloc(size_t bytes);
void hs_ed25519_host_free(void *p);
/* HS_ED25519_BATCH_RLC draws its 127-bit coefficients from SHA-512 keyed with a 32-byte secret taken
 * from the OS at first use, mixed with a per-call sequence number (host side, at enqueue) and a per-execution
 * counter kept in the context's workspace on the device (so a round replayed from a captured hipGraph draws fresh
 * coefficients as well); this pins the key and switches both counters off (tests, reproducible runs) for every call
 * enqueued - or graph captured - afterwards.  key == NULL: back to the default - a fresh secret from the OS, counters on. */
int hs_ed25519_set_rlc_key(const uint8_t key[32]);
/* HS_ED25519_BATCH_RLC calls below this size run as plain HS_ED25519_BATCH (the five-stage MSM pipeline
 * only pays from a few thousand signatures up; profiles/r2_rlc_ab.md).  A context can override the
 * threshold (tests run the pipeline from n = 1); 0 restores the default. */
size_t hs_ed25519_rlc_min_batch(void);

/* ---- device-resident variant (streamed rounds, benchmarks, multi-GPU shards) ----
 * All data pointers are DEVICE pointers on the initialised GPU, 16-byte aligned.
 *   d_digests     n*32 bytes, or 32 bytes when digest_stride == 0 (shared digest)
 *   digest_stride 0 or 32
 *   d_bitmap      8*ceil(n/64) bytes, written in full (bits >= n are 0)
 *   d_nreject     one uint32, incremented by the number of rejected signatures
 *                 (zero it before the call); may be NULL
 *   stream        hipStream_t to enqueue on (NULL = the context's own stream)
 * Enqueues and returns immediately: 0 on success or HS_ED25519_E_*.  The scratch
 * workspace belongs to `ctx`; a context serves one call at a time per stream order.
 * NOTE: stream = NULL selects the context's own NON-BLOCKING stream, not HIP's null stream;
 * work on other streams (including the null stream) that produces the inputs or consumes the
 * bitmap must be ordered against it explicitly (event / stream wait / synchronize). */
typedef struct hs_ed25519_ctx hs_ed25519_ctx;
int hs_ed25519_ctx_create(hs_ed25519_ctx **out, size_t max_batch);                 /* on device slot 0 */
int hs_ed25519_ctx_create_on(hs_ed25519_ctx **out, size_t max_batch, int slot);
int hs_ed25519_ctx_device(const hs_ed25519_ctx *ctx);                                /* HIP ordinal */
int hs_ed25519_ctx_set_rlc_min(hs_ed25519_ctx *ctx, size_t min_batch);
/* After the context's stream has drained: 1 = the last HS_ED25519_BATCH_RLC call on it accepted the
 * batch by the combination alone, 0 = it fell through to the per-signature kernels,
 * HS_ED25519_RLC_NOT_RUN = the RLC stage has not run (a state, not an error: errors are < 0);
 * *calls (optional) = how many calls ran it. */
#define HS_ED25519_RLC_NOT_RUN 2
int hs_ed25519_ctx_rlc_last(hs_ed25519_ctx *ctx, size_t *calls);
void hs_ed25519_ctx_destroy(hs_ed25519_ctx *ctx);
/* Like hs_ed25519_ctx_create, but the context's own stream is restricted to partition `part`
 * of `nparts` equal contiguous slices of the GPU's compute units (CU-masked stream), so that
 * small batches in flight on different contexts occupy disjoint CUs.  Pass stream = NULL to
 * hs_ed25519_verify_device to use it; hs_ed25519_ctx_stream returns the hipStream_t.
 * (A slice is the same share of EVERY XCD: the mask's bits are dealt round robin over the XCDs; `nparts` should divide
 * CUs / 8.)  The ORDER of creation matters on MI355X: create the contexts one after the other, context j on partition
 * j mod nparts - consecutively created streams sit on different pipes of the command processor, so the streams of one
 * slice then share a pipe; with consecutive contexts on the SAME partition, or contexts created by several threads at once,
 * a launch that cannot place its wavefronts yet stalls the queues of other slices (sixteen streams on four slices:
 * 85 - 88 M verifies/s instead of 103 - 104 M, profiles/r6_x_lane_map.md).  The host API's pools do this themselves. */
int hs_ed25519_ctx_create_partition(hs_ed25519_ctx **out, size_t max_batch, unsigned part, unsigned nparts);
void *hs_ed25519_ctx_stream(hs_ed25519_ctx *ctx);
int hs_ed25519_verify_device(hs_ed25519_ctx *ctx, const uint8_t *d_digests, size_t digest_stride,
                             const uint8_t *d_pks, const uint8_t *d_sigs, size_t n, uint32_t mode,
                             uint8_t *d_bitmap, uint32_t *d_nreject, void *stream);

/* Several INDEPENDENT small batches in one launch (up to 16 per grid; longer lists are cut into several
 * launches): a node under load has many certificates pending, and a 1,024-signature batch alone is only one
 * wavefront per SIMD of the slice of the chip its stream runs on.  Every batch keeps its own digests, bitmap and reject counter;
 * n <= 32,768 and mode BATCH or STRICT for each.  Pointer rules as for hs_ed25519_verify_device.
 * The library picks the kernel by what one call hands over (hs_ed25519_batches_kernel_for): up to three batches the
 * one-signature-per-quad form (latency: 0.56 ms per two 1,024-signature batches); from four the PAIRS form
 * (k_verify_fused_pairs_batches: every quad of lanes verifies two signatures one after the other, so the four lanes
 * decompress A, R of both at once instead of two of them idling - 11 % less arithmetic, 1 ms per four batches); from
 * eight batches totalling 8,192 signatures the ONE-LANE pair over the batch list (k_prepare_batches +
 * k_verify_batches: no lane algebra, the least work per verification - 72 M verifies/s at eight, 86-88 M at sixteen
 * batches per launch against 64 M and 57 M for the forms above - at 1.7 / 3 ms per launch pair; it uses the context's
 * workspace, which grows on first use).  Same accept set in every form. */
typedef struct hs_ed25519_batch {
    const uint8_t *d_digests; size_t digest_stride;
    const uint8_t *d_pks, *d_sigs;
    size_t n; uint32_t mode;
    uint8_t *d_bitmap; uint32_t *d_nreject;
} hs_ed25519_batch;
int hs_ed25519_verify_device_batches(hs_ed25519_ctx *ctx, const hs_ed25519_batch *batches, size_t count, void *stream);

/* ---- committee key cache: the extended forms (hs_ed25519_core.h declares keyset_create / _destroy and
 * verify_shared_cached / verify_tc_cached and describes the cache) ---------------------------------------------
 * A verification through a key set is 32 additions from the key's table + 16 from the base point's own
 * 16-bit-window table (67 MB per GPU, built inside hs_ed25519_init) + 1 square root or a share of an inversion for R. */
int hs_ed25519_keyset_create_on(hs_ed25519_keyset **out, const uint8_t *pks, size_t n_keys, int slot);
size_t hs_ed25519_keyset_size(const hs_ed25519_keyset *ks);
size_t hs_ed25519_keyset_bytes(size_t n_keys);   /* HBM footprint of a key set of that size */
int hs_ed25519_verify_multi_cached(const hs_ed25519_keyset *ks, const uint8_t *digests, const uint32_t *key_idx,
                                   const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *bitmap);
int hs_ed25519_verify_device_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, const uint8_t *d_digests,
                                    size_t digest_stride, const uint32_t *d_key_idx, const uint8_t *d_sigs, size_t n,
                                    uint32_t mode, uint8_t *d_bitmap, uint32_t *d_nreject, void *stream);
/* Several small key-cache batches in one launch (hs_ed25519_verify_device_batches with key indices): batches[i].d_pks
 * holds batch i's n uint32 key indices (4-byte aligned) instead of key bytes; mode BATCH or STRICT, n <= 32,768 each. */
int hs_ed25519_verify_device_batches_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, const hs_ed25519_batch *batches,
                                            size_t count, void *stream);

/* ---- TC::verify with the per-vote digests built on the device: the device-pointer forms (host forms: core header) ----
 * high_qc_rounds: 8-byte aligned. */
int hs_ed25519_verify_device_tc(hs_ed25519_ctx *ctx, uint64_t round, const uint64_t *d_high_qc_rounds,
                                const uint8_t *d_pks, const uint8_t *d_sigs, size_t n, uint8_t *d_bitmap,
                                uint32_t *d_nreject, void *stream);
int hs_ed25519_verify_device_tc_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, uint64_t round,
                                       const uint64_t *d_high_qc_rounds, const uint32_t *d_key_idx,
                                       const uint8_t *d_sigs, size_t n, uint8_t *d_bitmap, uint32_t *d_nreject,
                                       void *stream);

/* ---- round ring: streamed rounds (BASELINE config 5) -----------------------------------------
 * The reference's Core task verifies one certificate per message as rounds stream by
 * (consensus/src/core.rs:379, 401 -> QC::verify messages.rs:197).  A ring keeps `slots` rounds in
 * flight: each slot owns a HIP stream, PINNED host staging and device buffers; copies and kernels
 * of different slots overlap.  Protocol per slot:  begin (get the pinned pointers, write the
 * packed tuples in place - no second host copy) -> submit (asynchronous) -> poll / wait
 * (verdict + bitmap).
 *   ks != NULL: the ring verifies through that key set; `keys` then holds n uint32 indices,
 *               otherwise n*32 key bytes.
 *   digest_stride 0: `digests` holds one shared 32-byte digest; 32: one per signature.
 *   flags: how a round travels (all four have the same results):
 *     HS_ED25519_RING_GRAPH       H2D copy -> kernel(s) -> D2H copy as ONE hipGraph launch, captured
 *                                 once per slot and re-launched / updated in place.  HOST RULE for this form
 *                                 only (ROCm 7.0, tools/capture_probe.hip): while a round is being captured
 *                                 no thread of the process may call hipDeviceSynchronize or a legacy-stream
 *                                 hipMemcpy / hipMemset - the runtime invalidates the capture (submit then
 *                                 returns HS_ED25519_E_DEVICE) or crashes.  The engine calls neither after
 *                                 hs_ed25519_init; allocation, stream / event creation and work on other
 *                                 streams are harmless.  It never won on gfx950 (profiles/r1_ring_sweep.md).
 *     HS_ED25519_RING_STREAM_OPS  the same three operations issued to the slot's stream one by one
 *     HS_ED25519_RING_ZERO_COPY   no copies: the kernels read the pinned slot and write the pinned
 *                                 bitmap directly over PCIe (every input byte is read exactly once)
 *     HS_ED25519_RING_AUTO        the engine's choice by measurement (profiles/r1_ring_sweep.md):
 *                                 zero-copy for slot_batch <= 16,384, stream operations above
 * A ring is driven by one thread at a time (the reference's single Core task); different rings
 * are independent.  wait returns 1 / 0 / <0 like every verify call; poll returns 1 when the
 * slot's round has completed, 0 while it runs.  The slot's pinned buffers belong to the round
 * from begin until wait returns: do not write them while it is in flight.  A key set given to
 * hs_ed25519_ring_create must outlive the ring; destroy rings, contexts and key sets before
 * hs_ed25519_shutdown. */
#define HS_ED25519_RING_AUTO 0u
#define HS_ED25519_RING_GRAPH 1u
#define HS_ED25519_RING_STREAM_OPS 2u
#define HS_ED25519_RING_ZERO_COPY 3u
typedef struct hs_ed25519_ring hs_ed25519_ring;
int hs_ed25519_ring_create(hs_ed25519_ring **out, size_t slot_batch, unsigned slots,
                           const hs_ed25519_keyset *ks, uint32_t flags);
void hs_ed25519_ring_destroy(hs_ed25519_ring *ring);
int hs_ed25519_ring_begin(hs_ed25519_ring *ring, unsigned slot, size_t n, size_t digest_stride,
                          uint8_t **digests, uint8_t **keys, uint8_t **sigs);
int hs_ed25519_ring_submit(hs_ed25519_ring *ring, unsigned slot, uint32_t mode);
int hs_ed25519_ring_poll(hs_ed25519_ring *ring, unsigned slot);
int hs_ed25519_ring_wait(hs_ed25519_ring *ring, unsigned slot, uint8_t *bitmap);
/* Number of graph captures / updates and of rounds submitted so far. */
int hs_ed25519_ring_stats(const hs_ed25519_ring *ring, size_t *graph_captures, size_t *launches);
/* The context a slot runs on - BORROWED: it dies with the ring, never destroy it, and use it only while the slot
 * has nothing in flight.  For the per-context knobs and read-outs (hs_ed25519_ctx_set_rlc_min,
 * hs_ed25519_ctx_rlc_last, hs_ed25519_ctx_set_concurrent, hs_ed25519_ctx_set_kernel).  NULL on bad arguments. */
hs_ed25519_ctx *hs_ed25519_ring_slot_ctx(hs_ed25519_ring *ring, unsigned slot);

/* Profiling: keep a ring of `slots` HIP-event triples attached to the kernel dispatches of each
 * hs_ed25519_verify_device* call on `ctx` (hipExtLaunchKernel start/stop events, on the launch
 * stream; 0 = off).  After the
 * stream has been synchronised, hs_ed25519_ctx_kernel_times returns the summed durations
 * (ms) of the prepare and verify kernels over the recorded calls and their count. */
int hs_ed25519_ctx_set_profiling(hs_ed25519_ctx *ctx, size_t slots);
/* Record the events on every `every`-th call only (default 1).  Event markers are queue
 * packets; with many small batches in flight sampling keeps the measurement from perturbing
 * what it measures. */
int hs_ed25519_ctx_set_profiling_stride(hs_ed25519_ctx *ctx, size_t every);
int hs_ed25519_ctx_kernel_times(hs_ed25519_ctx *ctx, double *prepare_ms_sum, double *verify_ms_sum,
                                size_t *launches);

/* Per-context kernel choice, for tests and A/B measurements (the host-pointer calls always use
 * the default): 0 = by batch size (cold: eight lanes per signature up to 8,192 signatures, the fused
 * four-lanes-per-signature kernels up to 32,768, k_prepare + k_verify above; key cache: k_verify_cached_lane2r - two lanes per
 * signature and a third wavefront decompressing the R's beside them - up to 98,304 signatures and for batch lists; above
 * (from 49,152 on a context marked concurrent) the two-launch form: two-lane chains, then four or eight signatures per finish
 * lane compressed with ONE inversion), 1 = cold calls: the one-lane-per-signature pair (key-cache calls: as 3 - their one-lane
 * kernel lost its size range and left the library in round 5), 3 = the fused /
 * two-lane + R-wavefront kernels at any size, 4 = as 3, and hs_ed25519_verify_device_batches uses the pairs form whatever
 * the batch count (3 pins it to the one-signature-per-quad form, 1 to the one-lane pair over the
 * batch list), 5 = key cache: the two-launch form at any size, batch lists included (cold calls as 0, except that a cold batch
 * list stays on the quad kernels), 6 / 7 = cold single calls of up to 8 signatures per SIMD: the one-wavefront eight-lane form
 * (k_verify_fused8) / the split form with a helper wavefront (k_verify_fused8h) whatever hs_ed25519_autotune chose (larger calls
 * and batch lists as 0) - what bench.py uses to time BOTH forms on the box it runs on.  All have the same accept set.  Returns
 * the previous choice or HS_ED25519_E_ARG. */
#define HS_ED25519_K_AUTO 0
#define HS_ED25519_K_LANE 1
#define HS_ED25519_K_FUSED 3
#define HS_ED25519_K_FUSED_PAIRS 4
#define HS_ED25519_K_CACHED_C 5
#define HS_ED25519_K_OCT 6
#define HS_ED25519_K_OCT_SPLIT 7
int hs_ed25519_ctx_set_kernel(hs_ed25519_ctx *ctx, int which);

/* Latency form or throughput form.  By default a context assumes its call has the GPU to itself (the reference's
 * node verifies one certificate at a time: consensus/src/core.rs:379 -> messages.rs:197) and picks the forms with the
 * shortest time for ONE call: eight lanes per signature up to 8,192 signatures, and - above 65,536 - the sliver past
 * whole rounds of wavefronts as four-lane workgroups inside the second launch.  Both cost throughput when several
 * whole-chip calls run at once (profiles/r3_oct_ab.txt, r3_concurrent_ab.txt).  A caller that keeps several contexts
 * busy says so here (concurrent != 0) and gets four lanes per signature and no sliver; the verdicts are the same.  The library
 * sets it by itself where it knows: the slots of a round ring, and host-pointer calls that find another large call
 * of the same device in flight.  Returns the previous setting (0 / 1) or HS_ED25519_E_ARG. */
int hs_ed25519_ctx_set_concurrent(hs_ed25519_ctx *ctx, int concurrent);

/* The dominant kernel's symbol name (what to look for in a rocprofv3 kernel trace). */
const char *hs_ed25519_dominant_kernel(void);              /* for a 1,024-signature batch */
const char *hs_ed25519_dominant_kernel_for(size_t n);      /* for an n-signature cold-path call */
const char *hs_ed25519_cached_kernel_for(size_t n);        /* for a LONE n-signature key-cache call (k_verify_cached_lane2r up to 98,304
                                                            * signatures; above, k_cached_chain2 followed by k_cached_finish; on a
                                                            * context marked concurrent the two-launch form from 49,152) */
const char *hs_ed25519_batches_kernel_for(const hs_ed25519_ctx *ctx, size_t count);   /* hs_ed25519_verify_device_batches of `count` batches */

/* ---- measured kernel choice, and what the library resolved ---------------------------------------------------
 * Between 4,097 and 8,192 signatures (more than half a wavefront slot per SIMD) the faster of the two eight-lane cold forms
 * depends on the GPU in a way no static property predicts (round 4: boxes of one SKU, VBIOS, firmware and clocks differ by
 * 3 - 10 % in either direction; the driver's box contradicted the instruction-feed proxy that round used).  So the library
 * measures the thing itself: hs_ed25519_init - and this call, on demand (after a power-cap or clock change, say) - times
 * k_verify_fused8 and k_verify_fused8h on a canned, valid 6,667-signature batch (the QC of N = 10,000; signed on the
 * device; three rounds of a block of five runs per form, the first of a block untimed, minima of the dispatch timestamps;
 * ~10 ms in all, on a wide pool context: no stream of its own) on device slot `slot` and keeps the
 * faster for that slot.  A process that is one of several tenants of its GPU does not tune (HS_ED25519_E_ARG here): its
 * calls of that size run on its share of the CUs, where the eight-lane forms do not apply.  ms[0] = one-wavefront form, ms[1] = split form (ms may be NULL).  Returns the pick (0 / 1) or
 * HS_ED25519_E_*; a form whose run did not accept the whole canned batch is an engine error, never a pick.
 * HS_ED25519_OCT_SPLIT=0 / 1 in the environment pins the form instead; HS_ED25519_AUTOTUNE=0 skips the measurement at
 * init (the split form is then used throughout - the choice of two boxes in three). */
int hs_ed25519_autotune(int slot, double ms[2]);
/* Every value the library resolved at init - pools, tenancy, thresholds, environment overrides seen, the hardware-queue
 * budget against the GPU_MAX_HW_QUEUES it found, per device slot the autotune times and pick - as one JSON object.
 * Writes at most cap bytes (NUL-terminated) and returns the length the full text needs (snprintf convention). */
size_t hs_ed25519_config_dump(char *buf, size_t cap);

/* ---- one call over several GPUs (BASELINE configs[3]: a 100,000-validator QC on 8 x MI355X) ----
 * After hs_ed25519_init_devices with k slots these cut the batch into k contiguous shards of
 * ceil(n/k) signatures rounded up to a multiple of 64 (hs_ed25519_shard_plan), stage and launch all
 * shards from the calling thread - the devices run concurrently - and bring the per-shard accept
 * bitmaps together.  Same arguments, verdicts and bitmap as hs_ed25519_verify_shared / _multi; with
 * one slot they are equivalent to them.  The exchange of the bitmap slices is selectable:
 *   HS_ED25519_EXCHANGE_HOST  each device's slice (<= ~1 KB) lands in pinned host memory, the host
 *                             concatenates (default)
 *   HS_ED25519_EXCHANGE_RCCL  in-process RCCL all-gather over xGMI (ncclCommInitAll; librccl is
 *                             dlopen'ed on first use), then one device-to-host copy
 * hs_ed25519_set_shard_exchange returns the previous setting. */
#define HS_ED25519_EXCHANGE_HOST 0
#define HS_ED25519_EXCHANGE_RCCL 1
int hs_ed25519_shard_plan(size_t n, unsigned nshards, unsigned shard, size_t *lo, size_t *hi);
int hs_ed25519_set_shard_exchange(int how);
void hs_ed25519_shard_reset(void);        /* drop RCCL communicators / gather buffers (before shutdown) */
int hs_ed25519_shard_rccl_ranks(void);    /* ranks of the in-process RCCL exchange (0 until it first ran) */
/* mode HS_ED25519_BATCH_RLC through the two calls below is ONE random linear combination over the whole batch, cut over the
 * devices (SURVEY.md 8e "batch-MSM mode"; dalek's verify_batch is one combination, crypto/src/lib.rs:218): every device
 * reduces its shard to one partial point under the call's coefficient key, the 384-byte records are brought together (pinned
 * host memory, or the RCCL all-gather with HS_ED25519_EXCHANGE_RCCL), one device adds them and tests the identity once.
 * Accept -> 1 and an all-ones bitmap; otherwise the per-signature kernels localise, per shard, and the result is
 * HS_ED25519_BATCH's.  Runs from hs_ed25519_shard_msm_min() signatures up (default hs_ed25519_rlc_min_batch() x device slots:
 * below ~40,000 signatures per device the per-signature kernels answer sooner; env HS_ED25519_SHARD_MSM_MIN or
 * hs_ed25519_set_shard_msm_min override), below that as HS_ED25519_BATCH.  hs_ed25519_shard_msm_stats: calls that ran the
 * combination and how many of them it accepted outright (returns 0).  hs_ed25519_shard_rccl_fallbacks: the exchanges (bitmap
 * slices or partial points) that were asked to travel by RCCL but went through pinned host memory because the collective
 * could not be SET UP (no librccl, a device list ncclCommInitAll refuses, no memory for the gather buffers) - every device
 * still holds its part then, so the call verifies instead of failing, hs_ed25519_last_error says why, and the refusal is
 * remembered until hs_ed25519_shard_reset (no new attempt per call); a device error while staging, or after the grouped call
 * has started, is returned as an error. */
size_t hs_ed25519_shard_msm_min(void);
int hs_ed25519_set_shard_msm_min(size_t n);
int hs_ed25519_shard_msm_stats(unsigned long long *calls, unsigned long long *accepts);
unsigned long long hs_ed25519_shard_rccl_fallbacks(void);
int hs_ed25519_verify_shared_sharded(const uint8_t digest[32], const uint8_t *pks, const uint8_t *sigs, size_t n,
                                     uint32_t mode, uint8_t *bitmap);
int hs_ed25519_verify_multi_sharded(const uint8_t *digests, const uint8_t *pks, const uint8_t *sigs, size_t n,
                                    uint32_t mode, uint8_t *bitmap);
/* Committee key cache over several GPUs.  A QC is an arbitrary 2f+1 subset of the committee, so the
 * votes of a call do not line up with any slicing of the key table:
 *   HS_ED25519_KEYS_REPLICATE  every slot holds the whole table (100,000 validators: 53 GB of 288 GB
 *                              per GPU); shards stay contiguous
 *   HS_ED25519_KEYS_BY_OWNER   slot s holds the keys of hs_ed25519_shard_plan(n_keys, k, s); the
 *                              call routes each vote to its key's owner and scatters the verdict bits
 *                              back (1/k of the memory; shard sizes follow the votes)
 * key_idx[i] indexes the WHOLE committee in both placements. */
#define HS_ED25519_KEYS_REPLICATE 0
#define HS_ED25519_KEYS_BY_OWNER 1
typedef struct hs_ed25519_keyset_sharded hs_ed25519_keyset_sharded;
int hs_ed25519_keyset_sharded_create(hs_ed25519_keyset_sharded **out, const uint8_t *pks, size_t n_keys, int placement);
void hs_ed25519_keyset_sharded_destroy(hs_ed25519_keyset_sharded *kss);
size_t hs_ed25519_keyset_sharded_bytes_per_device(const hs_ed25519_keyset_sharded *kss, int slot);
int hs_ed25519_verify_shared_sharded_cached(const hs_ed25519_keyset_sharded *kss, const uint8_t digest[32],
                                            const uint32_t *key_idx, const uint8_t *sigs, size_t n, uint32_t mode,
                                            uint8_t *bitmap);
int hs_ed25519_verify_multi_sharded_cached(const hs_ed25519_keyset_sharded *kss, const uint8_t *digests,
                                           const uint32_t *key_idx, const uint8_t *sigs, size_t n, uint32_t mode,
                                           uint8_t *bitmap);

/* ---- wire decode --------------------------------------------------------------
 * votes: Vec<(PublicKey, Signature)> of a QC travels as n bincode elements of 116 bytes each -
 * u64 LE 44 | 44 base64 characters (crypto/src/lib.rs:94-101 serialises a key as a string) | 64 signature
 * bytes - and a TC's (PublicKey, Signature, Round) elements carry a u64 LE round more (124 bytes).  These
 * take the elements as received: the base64 decode and the transposition into the packed arrays run on the
 * device ahead of the verification kernels, on the same stream.  Contract: the canonical encoding the
 * reference's serialiser always emits (length 44, standard alphabet, one '=', zero trailing bits); any other
 * element is rejected whatever its signature says (bit i = 0) - the shim re-parses such a frame with the
 * reference's own deserialiser.  Verdicts, bitmap and modes as hs_ed25519_verify_shared / _tc. */
int hs_ed25519_verify_wire_shared(const uint8_t digest[32], const uint8_t *frames, size_t n, uint32_t mode, uint8_t *bitmap);
int hs_ed25519_verify_wire_tc(uint64_t round, const uint8_t *frames, size_t n, uint8_t *bitmap);
/* Decode only, device buffers: frame_bytes 116 (d_rounds may be NULL) or 124; d_pks n*32, d_sigs n*64 (16-byte
 * aligned), d_rounds n uint64, *d_nbad incremented per non-canonical element (may be NULL). */
int hs_ed25519_wire_decode_device(hs_ed25519_ctx *ctx, const uint8_t *d_frames, size_t frame_bytes, size_t n,
                                  uint8_t *d_pks, uint8_t *d_sigs, uint64_t *d_rounds, uint32_t *d_nbad, void *stream);

/* ---- batch signing for load generation -----------------------------------------
 * Signature::new (crypto/src/lib.rs:185-191) = RFC 8032 5.1.6 with the key pair's stored public key, for n
 * (secret, digest) pairs at once: secrets n x 64 bytes (seed || public key, the layout of crypto::SecretKey),
 * digests n x 32 (digest_stride 32) or one shared (0), signatures n x 64 out - bit-identical to dalek / OpenSSL
 * (signing is deterministic).  Not on a node's hot path (one signature per block / vote / timeout): it makes
 * 10^5 - 10^6 signature synthetic sets take milliseconds.  hs_ed25519_public_keys derives n public keys from
 * n 32-byte seeds.  Return 0 or HS_ED25519_E_*. */
int hs_ed25519_sign_batch(const uint8_t *secrets, const uint8_t *digests, size_t digest_stride, size_t n, uint8_t *sigs_out);
int hs_ed25519_sign_device(hs_ed25519_ctx *ctx, const uint8_t *d_secrets, const uint8_t *d_digests, size_t digest_stride,
                           size_t n, uint8_t *d_sigs, void *stream);
int hs_ed25519_public_keys(const uint8_t *seeds, size_t n, uint8_t *pks_out);

/* ---- asynchronous host-pointer verification -------------------------------------------------
 * hs_ed25519_submit stages the caller's buffers (they may be reused as soon as it returns) and
 * enqueues the verification; hs_ed25519_ticket_wait blocks until it is done, writes the optional
 * bitmap and returns the call's verdict exactly like the synchronous forms (1 / 0 / <0), consuming
 * the ticket.  hs_ed25519_ticket_ready polls (1 = wait will not block, 0 = not yet).
 *   ks == NULL: pks_or_idx = n x 32 key bytes;  else n uint32 indices into the key set
 *   digest_stride: 0 = one shared digest (QC shape), 32 = one per signature
 * A ticket holds one of the pool's contexts (16 for n <= 1,024 - shared by the tickets of one burst - and 4 above,
 * per device slot): when none
 * is free, submit returns HS_ED25519_E_BUSY WITHOUT blocking (a thread that holds every ticket itself
 * would otherwise deadlock) - wait on an outstanding ticket and retry.  Every ticket must be waited on.
 * The SYNCHRONOUS calls (verify_shared / _multi / _tc, sign_batch, public_keys ...) wait for a context instead, however
 * long the holders take - unless the calling thread itself has contexts out as uncollected tickets (then nobody else may
 * ever return one): after one second such a caller runs on one context lent over the pool's cap, destroyed afterwards.
 * A caller that holds none never grows the pools (they are sized to the GPU's hardware queues).
 * flags: HS_ED25519_SUBMIT_MORE on every batch but the last of a burst lets calls of n <= 1,024 signatures share
 * one launch (the TCP_CORK idea): a 1,024-signature batch alone is one wavefront per SIMD of its CU slice, two per
 * launch reach the chip's full multiply-add issue rate.  Without the flag the call launches at once.  A batch
 * submitted with the flag launches when its successor arrives, when a call of another kind (another key set, a
 * larger batch) follows, or when any ticket of its launch is waited on or polled - never later.  Bursts are per
 * thread (another thread's calls never join or delay them); tickets may be waited on from any thread.
 * With several device slots open (hs_ed25519_init_devices) the host calls by key bytes - these and the synchronous
 * ones - go round the slots, one launch at a time; calls through a key set run on the key set's device.
 * Replaces nothing in the reference (crypto/src/lib.rs:212-219 is synchronous); it is what lets one
 * tokio task keep several certificates in flight. */
typedef struct hs_ed25519_ticket hs_ed25519_ticket;
#define HS_ED25519_SUBMIT_MORE 1u  /* another submit follows at once: this launch may wait for it (up to 4 batches per launch) */
int hs_ed25519_submit(hs_ed25519_ticket **out, const hs_ed25519_keyset *ks, const uint8_t *digests, size_t digest_stride,
                      const void *pks_or_idx, const uint8_t *sigs, size_t n, uint32_t mode, uint32_t flags);
int hs_ed25519_ticket_ready(hs_ed25519_ticket *t);
int hs_ed25519_ticket_device_slot(const hs_ed25519_ticket *t);  /* which device slot verifies it (diagnostics) */
int hs_ed25519_ticket_wait(hs_ed25519_ticket *t, uint8_t *bitmap);

#ifdef __cplusplus
}
#endif
#endif
