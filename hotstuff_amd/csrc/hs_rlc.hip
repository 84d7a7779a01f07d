// hs_rlc.hip - gfx950 kernels and host side of HS_ED25519_BATCH_RLC: ed25519-dalek 1.0.1
// `verify_batch` as it is computed (one random linear combination over the whole call, evaluated as
// a Pippenger multi-scalar multiplication), reached in the reference from
// /root/reference/crypto/src/lib.rs:218 <- consensus/src/messages.rs:197 (QC::verify).
// The arithmetic of every stage is in rlc.h (host+device, validated on the CPU by tests/emu);
// this file maps it onto the chip:
//
//   k_rlc_prepare   one lane per point (2n lanes; n when A comes from a key set's Niels records):
//                   decompress -> 128-byte affine Niels record (one cache line, dwordx4 stores);
//                   the R lane hashes, draws z_i, z_i h_i and z_i s_i mod L, writes 2 x 32 digit bytes
//                   [window][point] and adds z_i s_i into eight 64-bit limb sums (atomics, carry-free).
//   k_rlc_sort      one 1,024-lane workgroup per window: counting sort of the window's points by
//                   |digit| in LDS (histogram -> offsets -> scatter of (record id | sign)).
//   k_rlc_accum     one lane per (window, bucket, segment): a wavefront covers 64 consecutive buckets of
//                   one window and one segment, so its lanes walk runs of similar length; each step gathers
//                   one whole 128-byte record and does one mixed addition.  Partials go out
//                   structure-of-arrays so that k_rlc_reduce reads them coalesced.
//   k_rlc_reduce    two wavefronts per window, one lane per bucket: fold the segments, then
//                   sum_j j * bucket_j WITHOUT the 256-step running sum: a suffix scan across the
//                   wavefront (6 shuffle + add steps), a butterfly reduction of the suffix sums (6 more)
//                   and 64 x the upper wavefront's total (6 doublings) - 12 point additions deep.
//   k_rlc_final     one quad: Horner over the 32 window sums (248 doublings, one coordinate per lane,
//                   DPP glue of verify_quad.h), + [-(sum z_i s_i)]B from the base point's window table,
//                   identity test -> one flag word.
// The per-signature kernels of hs_ed25519.hip then run with that flag as their `skip` word: they leave
// at once with an all-ones bitmap when the combination held and localise the rejects when it did not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include "rlc.h"
#include "engine.h"

using namespace hs;
#define fail hs::hs_fail

// ------------------------------------------------------------------------------------------ device
struct rlc_keys_src {            // where signature i's key bytes / curve flag come from
    const u32 *pks;              // cold: n x 8 words (and the A points are decompressed here)
    const u32 *key_idx;          // key set: n indices ...
    const u32 *ks_pk, *ks_flag;  // ... into the set's key bytes and flags
    u32 nkeys;
};

__global__ void __launch_bounds__(64, 3)
k_rlc_prepare(rlc_view v, rlc_key key, msg_src msgs, rlc_keys_src keys, const u32 *__restrict__ sigs, u32 n) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    if (t == 0) rlc_store_identity_record(v.rec + (size_t)HS_RLC_REC_WORDS * 2 * v.np);
    const u32 tasks = keys.pks ? 2 * n : n;
    if (t >= tasks) return;
    const bool is_r = t < n;
    const u32 i = is_r ? t : t - n;
    const u32 *sig = sigs + (size_t)16 * i;
    const u32 *pk;
    bool ok = true;
    if (keys.pks) pk = keys.pks + (size_t)8 * i;
    else {
        const u32 ki = keys.key_idx[i];
        const bool known = ki < keys.nkeys;
        pk = keys.ks_pk + (size_t)8 * (known ? ki : 0);
        ok = known && (keys.ks_flag[known ? ki : 0] & HS_F_ONCURVE);
    }
    u32 enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) enc[j] = is_r ? sig[j] : pk[j];
    ok = rlc_point(v, is_r ? i : v.np + i, enc) && ok;
    if (is_r) {
        u32 msg[8];
        load_msg(msg, msgs, i);
        ok = rlc_scalars(v, i, key, pk, sig, msg) && ok;
    }
    if (!ok) atomicOr(v.failed, 1u);
}

// counting sort of one window's points by |digit|
__global__ void __launch_bounds__(1024)
k_rlc_sort(const uint8_t *__restrict__ dig, u32 NP, u32 np, const u32 *__restrict__ key_idx, u32 *__restrict__ sorted,
           u32 *__restrict__ boff) {
    __shared__ u32 hist[HS_RLC_BUCKETS + 1], cur[HS_RLC_BUCKETS + 1];
    const u32 w = blockIdx.x;
    const u32 *d4 = reinterpret_cast<const u32 *>(dig + (size_t)w * NP);
    for (u32 j = threadIdx.x; j <= HS_RLC_BUCKETS; j += 1024) hist[j] = 0;
    __syncthreads();
    for (u32 q = threadIdx.x; q < NP / 4; q += 1024) {
        const u32 four = d4[q];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int dd = (int)((four >> (8 * b)) & 0xff) - 128;
            const u32 m = (u32)(dd < 0 ? -dd : dd);
            if (m) atomicAdd(&hist[m], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 acc = 0;
        for (u32 j = 1; j <= HS_RLC_BUCKETS; j++) { cur[j] = acc; boff[w * (HS_RLC_BUCKETS + 1) + j - 1] = acc; acc += hist[j]; }
        boff[w * (HS_RLC_BUCKETS + 1) + HS_RLC_BUCKETS] = acc;
    }
    __syncthreads();
    u32 *out = sorted + (size_t)w * NP;
    for (u32 q = threadIdx.x; q < NP / 4; q += 1024) {
        const u32 four = d4[q];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int dd = (int)((four >> (8 * b)) & 0xff) - 128;
            const u32 m = (u32)(dd < 0 ? -dd : dd);
            if (m) {
                const u32 p = 4 * q + b;
                const u32 id = (key_idx && p >= np) ? (HS_RLC_ID_CACHED | key_idx[p - np]) : p;
                out[atomicAdd(&cur[m], 1u)] = id | (dd < 0 ? HS_RLC_ID_NEG : 0u);
            }
        }
    }
}

// one lane per (window, bucket, segment); block b -> window b / (2 S), segment, bucket half
__global__ void __launch_bounds__(64)
k_rlc_accum(const u32 *__restrict__ sorted, const u32 *__restrict__ boff, const u32 *__restrict__ rec,
            const u32 *__restrict__ ksn, u32 NP, u32 S, u32 *__restrict__ part) {
    const u32 w = blockIdx.x / (2 * S), rem = blockIdx.x % (2 * S), s = rem >> 1, half = rem & 1;
    const u32 j = half * 64 + threadIdx.x;                       // bucket j holds |digit| = j + 1
    const u32 lo0 = boff[w * (HS_RLC_BUCKETS + 1) + j], hi0 = boff[w * (HS_RLC_BUCKETS + 1) + j + 1];
    const u32 len = hi0 - lo0;
    const u32 a = lo0 + (u32)((u64)len * s / S), b = lo0 + (u32)((u64)len * (s + 1) / S);
    u32 trip = b - a;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) trip = max(trip, (u32)__shfl_xor((int)trip, off, 64));
    ge_p3 acc;
    rlc_accum_lane(acc, sorted + (size_t)w * NP, a, b, trip, rec, ksn, NP);
    u32 *o = part + ((size_t)(w * S + s) * 40) * HS_RLC_BUCKETS + j;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        o[(size_t)i * HS_RLC_BUCKETS] = acc.X.v[i]; o[(size_t)(10 + i) * HS_RLC_BUCKETS] = acc.Y.v[i];
        o[(size_t)(20 + i) * HS_RLC_BUCKETS] = acc.Z.v[i]; o[(size_t)(30 + i) * HS_RLC_BUCKETS] = acc.T.v[i];
    }
}

__device__ __forceinline__ void pt_shfl_down(ge_p3 &r, const ge_p3 &p, int delta) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        r.X.v[i] = (u32)__shfl_down((int)p.X.v[i], delta, 64); r.Y.v[i] = (u32)__shfl_down((int)p.Y.v[i], delta, 64);
        r.Z.v[i] = (u32)__shfl_down((int)p.Z.v[i], delta, 64); r.T.v[i] = (u32)__shfl_down((int)p.T.v[i], delta, 64);
    }
}
__device__ __forceinline__ void pt_shfl_xor(ge_p3 &r, const ge_p3 &p, int mask) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        r.X.v[i] = (u32)__shfl_xor((int)p.X.v[i], mask, 64); r.Y.v[i] = (u32)__shfl_xor((int)p.Y.v[i], mask, 64);
        r.Z.v[i] = (u32)__shfl_xor((int)p.Z.v[i], mask, 64); r.T.v[i] = (u32)__shfl_xor((int)p.T.v[i], mask, 64);
    }
}
__device__ __forceinline__ void pt_shfl_lane0(ge_p3 &r, const ge_p3 &p) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        r.X.v[i] = (u32)__shfl((int)p.X.v[i], 0, 64); r.Y.v[i] = (u32)__shfl((int)p.Y.v[i], 0, 64);
        r.Z.v[i] = (u32)__shfl((int)p.Z.v[i], 0, 64); r.T.v[i] = (u32)__shfl((int)p.T.v[i], 0, 64);
    }
}

// segment partials -> bucket sums -> window sum, by wavefront shuffles (rlc_bucket_fold is the
// serial definition): with V_l = sum_{m >= l} B_m (suffix scan), sum_l (l + 1) B_l = sum_l V_l.
__global__ void __launch_bounds__(128)
k_rlc_reduce(const u32 *__restrict__ part, u32 S, u32 *__restrict__ wsum) {
    __shared__ u32 s_up[40];
    const u32 w = blockIdx.x, j = threadIdx.x, wave = j >> 6, lane = j & 63u;
    ge_p3 V;
    {
        const u32 *p = part + ((size_t)(w * S) * 40) * HS_RLC_BUCKETS + j;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            V.X.v[i] = p[(size_t)i * HS_RLC_BUCKETS]; V.Y.v[i] = p[(size_t)(10 + i) * HS_RLC_BUCKETS];
            V.Z.v[i] = p[(size_t)(20 + i) * HS_RLC_BUCKETS]; V.T.v[i] = p[(size_t)(30 + i) * HS_RLC_BUCKETS];
        }
    }
#pragma nounroll
    for (u32 s = 1; s < S; s++) {
        const u32 *p = part + ((size_t)(w * S + s) * 40) * HS_RLC_BUCKETS + j;
        ge_p3 q;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            q.X.v[i] = p[(size_t)i * HS_RLC_BUCKETS]; q.Y.v[i] = p[(size_t)(10 + i) * HS_RLC_BUCKETS];
            q.Z.v[i] = p[(size_t)(20 + i) * HS_RLC_BUCKETS]; q.T.v[i] = p[(size_t)(30 + i) * HS_RLC_BUCKETS];
        }
        ge_add_p3(V, V, q);
    }
    ge_p3 ident;
    ge_identity(ident);
    // suffix scan over the wavefront's 64 buckets
#pragma nounroll
    for (int off = 1; off < 64; off <<= 1) {
        ge_p3 q;
        pt_shfl_down(q, V, off);
        const bool in = lane + (u32)off < 64;
        fe_select(q.X, ident.X, q.X, in); fe_select(q.Y, ident.Y, q.Y, in);
        fe_select(q.Z, ident.Z, q.Z, in); fe_select(q.T, ident.T, q.T, in);
        ge_add_p3(V, V, q);
    }
    ge_p3 T;                           // this wavefront's plain total = V of lane 0
    pt_shfl_lane0(T, V);
    // R = sum over lanes of the suffix sums
#pragma nounroll
    for (int off = 32; off >= 1; off >>= 1) {
        ge_p3 q;
        pt_shfl_xor(q, V, off);
        ge_add_p3(V, V, q);
    }
    if (wave == 1) {                   // buckets 64..127 weigh 64 more each: + 64 T
#pragma nounroll
        for (int d = 0; d < 6; d++) { ge_p1p1 t; ge_dbl(t, T); ge_p1p1_to_p3(T, t, true); }
        ge_add_p3(V, V, T);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 10; i++) { s_up[i] = V.X.v[i]; s_up[10 + i] = V.Y.v[i]; s_up[20 + i] = V.Z.v[i]; s_up[30 + i] = V.T.v[i]; }
        }
    }
    __syncthreads();
    if (wave == 0) {
        ge_p3 q;
#pragma unroll
        for (int i = 0; i < 10; i++) { q.X.v[i] = s_up[i]; q.Y.v[i] = s_up[10 + i]; q.Z.v[i] = s_up[20 + i]; q.T.v[i] = s_up[30 + i]; }
        ge_add_p3(V, V, q);
        if (lane == 0) {
            u32 *o = wsum + (size_t)40 * w;
#pragma unroll
            for (int i = 0; i < 10; i++) { o[i] = V.X.v[i]; o[10 + i] = V.Y.v[i]; o[20 + i] = V.Z.v[i]; o[30 + i] = V.T.v[i]; }
        }
    }
}

__global__ void __launch_bounds__(64)
k_rlc_final(const u32 *__restrict__ wsum, const u32 *__restrict__ btab, const unsigned long long *__restrict__ csum,
            const u32 *__restrict__ failed, u32 *__restrict__ ok_flag) {
    if (threadIdx.x >= 4) return;      // one quad
    const qflags f = rlc_final_quad(wsum, btab, csum);
    const bool all = q_quad_all(f);
    if (threadIdx.x == 0) *ok_flag = (all && *failed == 0) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------ host
namespace {
struct rlc_ws {
    size_t np = 0; u32 S = 0;
    u32 *rec = nullptr; uint8_t *dig = nullptr; u32 *sorted = nullptr, *boff = nullptr, *part = nullptr, *wsum = nullptr;
    uint8_t *small = nullptr;      // csum[8] (64 B) | fail (4) | ok (4) | pad
    size_t min_batch = 0;          // 0 = the library default
    size_t calls = 0;
};
std::mutex k_mu;
rlc_key g_key;
bool g_key_set = false;

// Below this many signatures the five-stage pipeline costs more than the per-signature kernels
// (one wavefront's serial chain either way; profiles/r2_rlc_ab.md): the call then runs plain BATCH.
const size_t kRlcMinBatch = 2048;

void ws_free(rlc_ws *w) {
    if (w->rec) (void)hipFree(w->rec);
    if (w->dig) (void)hipFree(w->dig);
    if (w->sorted) (void)hipFree(w->sorted);
    if (w->boff) (void)hipFree(w->boff);
    if (w->part) (void)hipFree(w->part);
    if (w->wsum) (void)hipFree(w->wsum);
    if (w->small) (void)hipFree(w->small);
    w->rec = w->sorted = w->boff = w->part = w->wsum = nullptr; w->dig = w->small = nullptr; w->np = 0; w->S = 0;
}
u32 segments_for(size_t n) {       // lanes per bucket: runs of ~16 points per lane, at most 64 segments
    const size_t per_bucket = 2 * n / HS_RLC_BUCKETS;
    u32 s = 1;
    while (s < 64 && (size_t)s * 16 < per_bucket) s <<= 1;
    return s;
}
rlc_ws *ws_of(hs_ed25519_ctx *ctx) {
    if (!ctx->rlc) ctx->rlc = new (std::nothrow) rlc_ws();
    return (rlc_ws *)ctx->rlc;
}
}  // namespace

void hs::rlc_release(hs_ed25519_ctx *c) {
    if (!c->rlc) return;
    rlc_ws *w = (rlc_ws *)c->rlc;
    ws_free(w);
    delete w;
    c->rlc = nullptr;
}

// all allocation of an n-signature RLC call (none happens later, e.g. under graph capture)
int hs::rlc_reserve(hs_ed25519_ctx *ctx, size_t n) {
    rlc_ws *w = ws_of(ctx);
    if (!w) return fail(HS_ED25519_E_NOMEM, "rlc: out of memory");
    const size_t np = round_up(n, 64);
    const u32 S = segments_for(n);
    if (np <= w->np && S <= w->S) return 0;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    HS_HIP(hipStreamSynchronize(ctx->stream), "stream sync");       // the old buffers may be in use
    const size_t cap = np > w->np ? np : w->np;
    const u32 Scap = S > w->S ? S : w->S;
    ws_free(w);
    const size_t NP = 2 * cap;
    HS_HIP(hipMalloc((void **)&w->rec, (NP + 1) * 128), "hipMalloc rlc records");
    HS_HIP(hipMalloc((void **)&w->dig, (size_t)HS_RLC_WINDOWS * NP), "hipMalloc rlc digits");
    HS_HIP(hipMalloc((void **)&w->sorted, (size_t)HS_RLC_WINDOWS * NP * 4), "hipMalloc rlc sorted");
    HS_HIP(hipMalloc((void **)&w->boff, (size_t)HS_RLC_WINDOWS * (HS_RLC_BUCKETS + 1) * 4), "hipMalloc rlc offsets");
    HS_HIP(hipMalloc((void **)&w->part, (size_t)HS_RLC_WINDOWS * Scap * 40 * HS_RLC_BUCKETS * 4), "hipMalloc rlc partials");
    HS_HIP(hipMalloc((void **)&w->wsum, (size_t)HS_RLC_WINDOWS * 40 * 4), "hipMalloc rlc window sums");
    HS_HIP(hipMalloc((void **)&w->small, 128), "hipMalloc rlc flags");
    w->np = cap; w->S = Scap;
    return 0;
}

extern "C" int hs_ed25519_set_rlc_key(const uint8_t key[32]) {
    if (!key) return fail(HS_ED25519_E_ARG, "set_rlc_key: null");
    std::lock_guard<std::mutex> lk(k_mu);
    memcpy(g_key.w, key, 32);
    g_key_set = true;
    return 0;
}
static int current_key(rlc_key &out) {
    std::lock_guard<std::mutex> lk(k_mu);
    if (!g_key_set) {              // dalek: thread_rng(); here 32 bytes from the OS, once per process
        FILE *f = fopen("/dev/urandom", "rb");
        const bool ok = f && fread(g_key.w, 1, 32, f) == 32;
        if (f) fclose(f);
        if (!ok) return fail(HS_ED25519_E_DEVICE, "rlc: cannot read /dev/urandom for the coefficient key");
        g_key_set = true;
    }
    out = g_key;
    return 0;
}
// tests / tuning: run the RLC stage from `min_batch` signatures up on this context (0 = library default)
extern "C" int hs_ed25519_ctx_set_rlc_min(hs_ed25519_ctx *ctx, size_t min_batch) {
    if (!ctx) return fail(HS_ED25519_E_ARG, "ctx_set_rlc_min: null context");
    rlc_ws *w = ws_of(ctx);
    if (!w) return fail(HS_ED25519_E_NOMEM, "rlc: out of memory");
    w->min_batch = min_batch;
    return 0;
}
extern "C" size_t hs_ed25519_rlc_min_batch(void) { return kRlcMinBatch; }
// after the context's stream has drained: how many calls ran the RLC stage, and whether the last one
// accepted the batch outright (1), fell through to the per-signature kernels (0), or did not run (-1)
extern "C" int hs_ed25519_ctx_rlc_last(hs_ed25519_ctx *ctx, size_t *calls) {
    if (!ctx) return fail(HS_ED25519_E_ARG, "ctx_rlc_last: null context");
    rlc_ws *w = (rlc_ws *)ctx->rlc;
    if (calls) *calls = w ? w->calls : 0;
    if (!w || !w->small || !w->calls) return -1;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    HS_HIP(hipStreamSynchronize(ctx->stream), "stream sync");
    u32 flags[2] = {0, 0};
    HS_HIP(hipMemcpy(flags, w->small + 64, 8, hipMemcpyDeviceToHost), "read rlc flags");
    return flags[1] ? 1 : 0;
}

int hs::enqueue_rlc(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint8_t *d_pks,
                    const uint32_t *d_key_idx, const uint8_t *d_sigs, size_t n, u32 **d_ok_flag, void *stream) {
    *d_ok_flag = nullptr;
    if (!d_sigs || (!ks && !d_pks) || (ks && !d_key_idx)) return fail(HS_ED25519_E_ARG, "verify_device: null pointer");
    if (misaligned(d_sigs, 16) || misaligned(d_pks, 16) || misaligned(d_key_idx, 4)) return fail(HS_ED25519_E_ARG, "verify_device: misaligned pointer");
    if (n > 0x1fffffffu) return fail(HS_ED25519_E_ARG, "verify_device: bad n");
    if (ks && ks->slot != ctx->slot) return fail(HS_ED25519_E_ARG, "verify_device_cached: the key set lives on another device slot than the context");
    rlc_ws *w = ws_of(ctx);
    if (!w) return fail(HS_ED25519_E_NOMEM, "rlc: out of memory");
    const size_t min_batch = w->min_batch ? w->min_batch : kRlcMinBatch;
    if (n < min_batch) return 0;           // too small to pay for the pipeline: plain per-signature BATCH
    const u32 *btab = dev_base_table(ctx->slot);
    if (!btab) return HS_ED25519_E_DEVICE;
    int rc = rlc_reserve(ctx, n);
    if (rc) return rc;
    rlc_key key;
    rc = current_key(key);
    if (rc) return rc;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const u32 np = (u32)round_up(n, 64), NP = 2 * np, S = segments_for(n), nn = (u32)n;
    rlc_view v;
    v.rec = w->rec; v.dig = w->dig; v.csum = (unsigned long long *)w->small; v.failed = (u32 *)(w->small + 64); v.np = np;
    u32 *ok_flag = (u32 *)(w->small + 68);
    HS_HIP(hipMemsetAsync(w->small, 0, 128, st), "rlc clear flags");
    HS_HIP(hipMemsetAsync(w->dig, 0x80, (size_t)HS_RLC_WINDOWS * NP, st), "rlc clear digits");
    rlc_keys_src keys;
    keys.pks = ks ? nullptr : (const u32 *)d_pks; keys.key_idx = ks ? d_key_idx : nullptr;
    keys.ks_pk = ks ? ks->d_pk : nullptr; keys.ks_flag = ks ? ks->d_flag : nullptr; keys.nkeys = ks ? (u32)ks->nkeys : 0;
    const u32 *p_sg = (const u32 *)d_sigs;
    const u32 tasks = ks ? nn : 2 * nn;
    {
        void *a[] = { &v, &key, &msgs, &keys, &p_sg, (void *)&nn };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_prepare, dim3((tasks + 63) / 64), dim3(64), a, 0, st), "launch k_rlc_prepare");
    }
    {
        const uint8_t *p_dig = w->dig; const u32 *p_ki = keys.key_idx; u32 *p_sorted = w->sorted, *p_boff = w->boff;
        void *a[] = { &p_dig, (void *)&NP, (void *)&np, &p_ki, &p_sorted, &p_boff };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_sort, dim3(HS_RLC_WINDOWS), dim3(1024), a, 0, st), "launch k_rlc_sort");
    }
    {
        const u32 *p_sorted = w->sorted, *p_boff = w->boff, *p_rec = w->rec, *p_ksn = ks ? ks->d_niels : nullptr;
        u32 *p_part = w->part;
        void *a[] = { &p_sorted, &p_boff, &p_rec, &p_ksn, (void *)&NP, (void *)&S, &p_part };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_accum, dim3(HS_RLC_WINDOWS * 2 * S), dim3(64), a, 0, st), "launch k_rlc_accum");
    }
    {
        const u32 *p_part = w->part; u32 *p_wsum = w->wsum;
        void *a[] = { &p_part, (void *)&S, &p_wsum };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_reduce, dim3(HS_RLC_WINDOWS), dim3(128), a, 0, st), "launch k_rlc_reduce");
    }
    {
        const u32 *p_wsum = w->wsum; const unsigned long long *p_csum = v.csum; const u32 *p_fail = v.failed;
        void *a[] = { &p_wsum, &btab, &p_csum, &p_fail, &ok_flag };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_final, dim3(1), dim3(64), a, 0, st), "launch k_rlc_final");
    }
    w->calls++;
    *d_ok_flag = ok_flag;
    return 0;
}
