/* hs_ed25519_core.h - the CORE C ABI of the MI355X batch Ed25519 verification engine: the twelve functions the
 * Rust shim of INTEGRATION.md sections 2-6 binds, and nothing else.  include/hs_ed25519.h (which includes this file)
 * adds the extended surface: device-pointer calls, tickets, the round ring, sharding, wire decode, signing, probes.
 *
 * The reference (asonnino/hotstuff) has no FFI today; its boundary for this
 * path is two Rust associated functions in crypto/src/lib.rs:
 *
 *   Signature::verify(&self, &Digest, &PublicKey)            lib.rs:200-204
 *   Signature::verify_batch(&Digest, I: IntoIterator<...>)   lib.rs:206-219
 *
 * reached from consensus/src/messages.rs:64,144,258 (single), :197 (QC::verify,
 * one shared digest) and :307-313 (TC::verify, one digest per vote).  These
 * entry points are what a Rust `extern "C"` block in crypto/src/lib.rs binds
 * (INTEGRATION.md shows the shim).  Plain pointers and sizes only.
 *
 * Data formats (reference types, crypto/src/lib.rs:21-22, 65-66, 178-182):
 *   digest  32 bytes                         (Digest)
 *   pk      32 bytes, compressed Edwards y   (PublicKey)
 *   sig     64 bytes, R (part1) || s (part2) (Signature::flatten, lib.rs:193-198)
 * Arrays are packed: element i of `pks` is at pks + 32*i, of `sigs` at
 * sigs + 64*i, of `digests` at digests + 32*i.
 *
 * Verdict semantics (mode):
 *   HS_ED25519_BATCH  (0)  the equation ed25519-dalek 1.0.1 `verify_batch`
 *                          checks, decided per signature: s < L, A and R on the
 *                          curve, [k](-A) + [s]B == R as points, cofactorless,
 *                          no small-order rejection.  The call accepts iff every
 *                          signature does.  (dalek's randomised sum gives the same
 *                          answer on every input where dalek is deterministic; on
 *                          crafted inputs with a torsion component in A or R its
 *                          answer depends on its random z_i - there the engine
 *                          returns this per-signature verdict.  DESIGN.md section 3.)
 *   HS_ED25519_STRICT (1)  `PublicKey::verify_strict`: BATCH plus reject when
 *                          [8]R == O or [8]A == O.
 *   HS_ED25519_BATCH_RLC (2)  ed25519-dalek 1.0.1 `verify_batch` as it computes it: ONE
 *                          random linear combination  -[sum z_i s_i]B + sum [z_i]R_i +
 *                          sum [z_i h_i mod L]A_i == O  over the whole call (127-bit z_i - dalek
 *                          draws 128 bits; one bit less keeps the signed radix-256 recoding inside
 *                          16 windows - Pippenger multi-scalar multiplication on the GPU).  If it holds the
 *                          call accepts (bitmap all ones); if it does not, the per-signature
 *                          BATCH kernels run in the same call and localise the rejects, so the
 *                          verdict and bitmap are then exactly BATCH's.  Identical to BATCH on
 *                          every input where dalek is deterministic (SURVEY.md 8a classes i / ii);
 *                          on crafted torsion inputs (class iii) it can accept where BATCH
 *                          rejects, with dalek's own probability.  z_i = low 127 bits of
 *                          SHA-512((process secret ^ call number) || h_i || s_i || i):
 *                          unpredictable to whoever built the batch and independent from call to
 *                          call, as dalek's thread_rng draws are (hs_ed25519_set_rlc_key pins the
 *                          key AND drops the call number: reproducible coefficients for tests).
 *
 * Return value of every verify call:
 *    1  every signature accepted (n == 0 -> 1, as lib.rs:206-219 returns Ok)
 *    0  at least one signature rejected
 *   <0  engine error (HS_ED25519_E_*); the caller must treat it as Err - the
 *       engine never reports success on failure and has no CPU fallback inside.
 * bitmap (optional, may be NULL): bit (i & 7) of byte (i >> 3) is 1 iff
 * signature i was accepted; ceil(n/8) bytes are written.
 *
 * Thread safety: every function may be called concurrently from any thread
 * (the reference's tests call verify from several threads of one process,
 * SURVEY.md 8b); calls are served from an internal pool of stream contexts.
 *
 * HIP state a call touches: it makes the engine's GPU the calling thread's CURRENT HIP device (hipSetDevice) and leaves
 * it so - a host that drives other GPUs through HIP itself calls hipSetDevice again afterwards; the reference does not
 * use HIP at all.  Nothing else of the thread's or the process's HIP state is changed: no device-wide synchronisation
 * and no legacy-stream operation after hs_ed25519_init (both break stream captures of other threads on ROCm 7.0,
 * include/hs_ed25519.h, rings), no environment variable written, no device flags set.
 */
#ifndef HS_ED25519_CORE_H
#define HS_ED25519_CORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: (major << 16) | minor.  A change of argument lists, struct layouts, return conventions or
 * verdict semantics bumps the major; additions bump the minor.  The shim compares hs_ed25519_abi_version() - what the
 * loaded libhs_ed25519.so was BUILT from - with this constant at load time and refuses a library whose major differs
 * (INTEGRATION.md section 2); tests/test_abi_and_host_logic.py holds the two equal. */
#define HS_ED25519_ABI_VERSION 0x00050000u
uint32_t hs_ed25519_abi_version(void);

#define HS_ED25519_BATCH 0u
#define HS_ED25519_STRICT 1u
#define HS_ED25519_BATCH_RLC 2u

#define HS_ED25519_E_NOT_INIT (-1) /* hs_ed25519_init not called or failed        */
#define HS_ED25519_E_DEVICE (-2)   /* HIP runtime error (see hs_ed25519_last_error) */
#define HS_ED25519_E_ARG (-3)      /* null / misaligned pointer, bad mode           */
#define HS_ED25519_E_NOMEM (-4)    /* allocation failure                            */
#define HS_ED25519_E_BUSY (-5)     /* hs_ed25519_submit only: every pool context is in flight - wait on a ticket, retry */

/* Select the GPU (HIP ordinal), build the base-point tables on it (67 MB, a few ms), time the two forms of the
 * eight-lane cold kernel on a canned 6,667-signature batch and keep the faster for this GPU (~10 ms;
 * hs_ed25519_autotune in hs_ed25519.h), and pre-size the context pool for batches of `max_batch` signatures
 * (contexts grow on demand).  Idempotent and thread-safe.  Returns 0 or HS_ED25519_E_*.
 * The environment is NOT modified (round 5; a library must not call setenv in a threaded host): the pools want
 * hs_ed25519_hw_queue_budget(tenants) hardware queues, which the HIP runtime reads from GPU_MAX_HW_QUEUES when IT
 * initialises - the host's main() exports it before any thread starts.  When the
 * variable is missing or smaller than the budget, init still succeeds and hs_ed25519_last_error() holds a warning
 * text ("hs_ed25519_init: warning: ...").  HS_ED25519_SET_ENV=1 restores the old behaviour (setenv without
 * overwrite) for single-threaded hosts that want it.
 * The timing of the two kernel forms runs AFTER the engine has been marked initialised (it uses ordinary calls on a pool
 * context): a second thread whose calls arrive while another thread is still inside its first hs_ed25519_init is served, and
 * the pick is then measured under that load - call init once, before the callers start. */
int hs_ed25519_init(int device, size_t max_batch);
void hs_ed25519_shutdown(void);
/* Human-readable text of the last engine error (or init warning) seen by the calling thread. */
const char *hs_ed25519_last_error(void);
/* The batch size below which a call should stay on the CPU path of the shim (dalek): the GPU round
 * trip costs one wavefront's serial chain (~0.3 ms cold, ~0.1 ms with registered keys) whatever n
 * is.  keys_registered != 0: the caller would use a *_cached entry point.  bench.py measures the
 * crossover (`small_batch_crossover`); env HS_ED25519_MIN_GPU_BATCH overrides. */
size_t hs_ed25519_min_gpu_batch(int keys_registered);

/* Signature::verify_batch / QC::verify shape: one digest shared by n pairs.
 * Replaces crypto/src/lib.rs:206-219 (called at consensus/src/messages.rs:197). */
int hs_ed25519_verify_shared(const uint8_t digest[32], const uint8_t *pks, const uint8_t *sigs,
                             size_t n, uint32_t mode, uint8_t *bitmap);
/* TC::verify shape: one digest per pair.  Replaces the loop of
 * Signature::verify calls at consensus/src/messages.rs:307-313 (mode STRICT),
 * and with n = 1 a single crypto/src/lib.rs:200-204 call. */
int hs_ed25519_verify_multi(const uint8_t *digests, const uint8_t *pks, const uint8_t *sigs,
                            size_t n, uint32_t mode, uint8_t *bitmap);
/* TC::verify with the per-vote digests built on the device: consensus/src/messages.rs:307-313 hashes
 * SHA-512(round_le || high_qc_round_le)[..32] on the host for every vote before each Signature::verify; this takes
 * the certificate's round and the n high_qc_round values (uint64, host order) and hashes on the GPU.  Strict. */
int hs_ed25519_verify_tc(uint64_t round, const uint64_t *high_qc_rounds, const uint8_t *pks, const uint8_t *sigs,
                         size_t n, uint8_t *bitmap);

/* Committee key cache.  Every key the reference verifies belongs to the committee, which is static per epoch
 * (consensus/src/config.rs:39-43).  hs_ed25519_keyset_create decompresses the n_keys public keys once and builds
 * per-key window tables in HBM (528,420 bytes per key) so that a verification needs no doublings and no
 * decompression of A.  The *_cached calls take key_idx[i] = position of signature i's key in the set instead of the
 * key bytes (the shim resolves it with the same map QC::verify / TC::verify already use for the stake lookup,
 * consensus/src/messages.rs:182-194); an index >= n_keys rejects that signature.  Accept set, return values, bitmap
 * format and modes are identical to the calls above.  A key set is immutable and may be used concurrently from any
 * number of threads.  hs_ed25519_keyset_destroy follows free()'s rule - no call may be using the set - with one courtesy:
 * host-API work still in flight on it (tickets not collected, synchronous calls of other threads that have not returned,
 * round rings not destroyed) is counted, and destroy waits up to ten seconds for the count to drain before it frees the
 * tables; if it does not drain, the tables are LEAKED (never freed under a running kernel) and hs_ed25519_last_error says
 * so.  It does not synchronise the device (rounds 1-4 did: a device-wide synchronisation breaks a hipGraph capture
 * in progress on any other thread).  Calls through the device-pointer entry points on the caller's own streams
 * (hs_ed25519.h) are not counted: drain those streams first. */
typedef struct hs_ed25519_keyset hs_ed25519_keyset;
int hs_ed25519_keyset_create(hs_ed25519_keyset **out, const uint8_t *pks, size_t n_keys);   /* device slot 0 */
void hs_ed25519_keyset_destroy(hs_ed25519_keyset *ks);
int hs_ed25519_verify_shared_cached(const hs_ed25519_keyset *ks, const uint8_t digest[32], const uint32_t *key_idx,
                                    const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *bitmap);
int hs_ed25519_verify_tc_cached(const hs_ed25519_keyset *ks, uint64_t round, const uint64_t *high_qc_rounds,
                                const uint32_t *key_idx, const uint8_t *sigs, size_t n, uint8_t *bitmap);

#ifdef __cplusplus
}
#endif
#endif
