// hs_rlc.hip - gfx950 kernels and host side of HS_ED25519_BATCH_RLC: ed25519-dalek 1.0.1
// `verify_batch` as it is computed (one random linear combination over the whole call, evaluated as
// a Pippenger multi-scalar multiplication), reached in the reference from
// /root/reference/crypto/src/lib.rs:218 <- consensus/src/messages.rs:197 (QC::verify).
// The arithmetic of every stage is in rlc.h (host+device, validated on the CPU by tests/emu);
// this file maps it onto the chip:
//
//   k_rlc_prepare   one lane per task, three kinds in separate wavefronts: decompress R_i / decompress A_i
//                   (absent when A comes from a key set's Niels records) -> 128-byte affine Niels record (one
//                   cache line, dwordx4 stores); hash, draw z_i, z_i h_i and z_i s_i mod L, write 2 x 32 digit
//                   bytes [window][point] and add z_i s_i into eight 64-bit limb sums (atomics, carry-free).
//   k_rlc_hist / k_rlc_offsets / k_rlc_scatter   counting sort of every window's points by |digit|:
//                   LDS histograms per (slice, window) workgroup, one lane per window turns them into
//                   bucket offsets and chunk offsets, then each workgroup reserves its share of every
//                   bucket's run with one global atomic and scatters (record id | sign) through LDS cursors.
//   k_rlc_accum     one lane per chunk of <= C consecutive points of ONE bucket (a bucket of length len
//                   gets ceil(len / C) lanes: the 5-bit top window of the 253-bit scalars puts n / 16
//                   points into each of 16 buckets - that costs lanes, not time).  Each step gathers one
//                   whole 128-byte record and does one mixed addition; the wavefront then folds the
//                   chunks of equal buckets with a segmented suffix scan (shuffles) and stores one
//                   partial per bucket stretch.
//   k_rlc_reduce    two wavefronts per window, one lane per bucket: fold the stretch partials, then
//                   sum_j j * bucket_j WITHOUT the 256-step running sum: a suffix scan across the
//                   wavefront (6 shuffle + add steps), a butterfly reduction of the suffix sums (6 more)
//                   and 64 x the upper wavefront's total (6 doublings) - 12 point additions deep.
//                   A 33rd workgroup computes [-(sum z_i s_i)]B from the base point's window table.
//   k_rlc_final     one quad: Horner over the 32 window sums (248 doublings, one coordinate per lane,
//                   DPP glue of verify_quad.h), + the [b]B point, identity test -> one flag word.
// The per-signature kernels of hs_ed25519.hip then run with that flag as their `skip` word: they leave
// at once with an all-ones bitmap when the combination held and localise the rejects when it did not.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include "rlc.h"
#include "engine.h"

using namespace hs;
#define fail hs::hs_fail

// ------------------------------------------------------------------------------------------ device
#define HS_RLC_CSUM_SLOTS 64
struct rlc_keys_src {            // where signature i's key bytes / curve flag come from
    const u32 *pks;              // cold: n x 8 words (and the A points are decompressed here)
    const u32 *key_idx;          // key set: n indices ...
    const u32 *ks_pk, *ks_flag;  // ... into the set's key bytes and flags
    u32 nkeys;
};

// Three kinds of task, one lane each, in separate wavefronts: [0, n) decompress R_i; [n, 2n) hash, draw z_i, form the
// scalars and digits of signature i; [2n, 3n) decompress A_i (absent when A comes from a key set).  The scalar work
// (two SHA-512 blocks, two 256 x 256-bit products mod L, 64 digit bytes) is as long a chain as a decompression: on
// the lane that also decompressed R it made k_rlc_prepare 0.29 ms at 66,667 signatures, of which only 0.04 ms went
// away when the A decompressions did.
// (64, 2), not (64, 3): asked for three wavefronts per SIMD the compiler pins 168 VGPRs and spills 252 B per lane; asked
// for two it needs 159 and spills nothing - three still fit a SIMD.  Same speed (A/B: -DHS_RLC_PREPARE_WAVES=3), no scratch.
#ifndef HS_RLC_PREPARE_WAVES
#define HS_RLC_PREPARE_WAVES 2
#endif
__global__ void __launch_bounds__(64, HS_RLC_PREPARE_WAVES)
k_rlc_prepare(rlc_view v, rlc_key key, const u32 *__restrict__ epoch, msg_src msgs, rlc_keys_src keys,
              const u32 *__restrict__ sigs, u32 n, u32 index_base) {
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    // The coefficient key of THIS EXECUTION: the host mixed its call number into `key` at enqueue time, but a launch
    // replayed from a captured graph (the round ring's GRAPH form) carries that by-value argument unchanged for ever.
    // `epoch` is a 64-bit counter in the workspace that k_rlc_final bumps at the end of every execution, so a replay
    // draws fresh coefficients too.  Null when the key is pinned (hs_ed25519_set_rlc_key: reproducible runs).
    if (epoch) { key.w[4] ^= epoch[0]; key.w[5] ^= epoch[1]; }
    if (t == 0) rlc_store_identity_record(v.rec + (size_t)HS_RLC_REC_WORDS * 2 * v.np);
    u32 kind = t / v.np;                                     // np lanes per kind: whole wavefronts are of one kind
    const u32 i0 = t - kind * v.np;
    if (kind >= (keys.pks ? 3u : 2u)) return;
    // launch order R, A, scalars: the scalar tasks are ~0.4 of a decompression long, so the wavefronts that do not fit the
    // chip's first round (3,126 against 3,072 slots at 66,667 signatures) are short ones: k_rlc_prepare 270 -> 234 us there
    if (keys.pks) kind = kind == 1 ? 2u : (kind == 2 ? 1u : 0u);
    const bool active = i0 < n;
    if (kind != 1 && !active) return;
    const u32 i = active ? i0 : n - 1;                       // idle lanes of a scalar wavefront shadow the last signature (shuffles below)
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
    if (kind == 1) {
        u32 msg[8], c[8];
        load_msg(msg, msgs, i);
        ok = (rlc_scalars(v, i, key, pk, sig, msg, c, index_base + i) && ok) || !active;
        // sum z_i s_i over the wavefront (limb-wise, carry-free in 64 bits), then ONE atomic per limb
        unsigned long long c64[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned long long x = active ? c[j] : 0ull;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
            c64[j] = x;
        }
        // one of HS_RLC_CSUM_SLOTS copies of the eight limb sums, by wavefront: 1,042 wavefronts adding into the SAME eight
        // addresses cost 70 of k_rlc_prepare's 270 us at 66,667 signatures (same-address atomics serialise in the L2)
        if (threadIdx.x == 0) rlc_add_csum(v.csum + 8 * (blockIdx.x % HS_RLC_CSUM_SLOTS), c64);
    } else {
        u32 enc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) enc[j] = kind == 0 ? sig[j] : pk[j];
        ok = rlc_point(v, kind == 0 ? i : v.np + i, enc) && (kind == 0 || ok);
    }
    if (!ok) atomicOr(v.failed, 1u);
}

// ---- counting sort of every window's points by |digit|, in three small launches -------------------
// grid (G, 32): block (g, w) handles slice g of window w's NP digit bytes
__device__ __forceinline__ void rlc_slice(u32 &q0, u32 &q1, u32 NP, u32 G) {
    const u32 words = NP / 4, per = (words + G - 1) / G;
    q0 = blockIdx.x * per; q1 = min(words, q0 + per);
}
__global__ void __launch_bounds__(1024)
k_rlc_hist(const uint8_t *__restrict__ dig, u32 NP, u32 *__restrict__ ghist) {
    __shared__ u32 hist[HS_RLC_BUCKETS + 1];
    const u32 w = blockIdx.y;
    const u32 *d4 = reinterpret_cast<const u32 *>(dig + (size_t)w * NP);
    for (u32 j = threadIdx.x; j <= HS_RLC_BUCKETS; j += 1024) hist[j] = 0;
    __syncthreads();
    u32 q0, q1;
    rlc_slice(q0, q1, NP, gridDim.x);
    for (u32 q = q0 + threadIdx.x; q < q1; q += 1024) {
        const u32 four = d4[q];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int dd = (int)((four >> (8 * b)) & 0xff) - 128;
            const u32 m = (u32)(dd < 0 ? -dd : dd);
            if (m) atomicAdd(&hist[m], 1u);
        }
    }
    __syncthreads();
    for (u32 j = 1 + threadIdx.x; j <= HS_RLC_BUCKETS; j += 1024)
        if (hist[j]) atomicAdd(&ghist[w * (HS_RLC_BUCKETS + 1) + j], hist[j]);
}
// one lane per window: bucket offsets, the scatter cursors, and the chunk (lane) offsets of stage 3:
// bucket j gets ceil(len_j / C) lanes
__global__ void __launch_bounds__(64)
k_rlc_offsets(const u32 *__restrict__ ghist, u32 C, u32 *__restrict__ boff, u32 *__restrict__ segoff, u32 *__restrict__ gcur) {
    const u32 w = threadIdx.x;
    if (w >= HS_RLC_WINDOWS) return;
    const u32 base = w * (HS_RLC_BUCKETS + 1);
    u32 acc = 0, lanes = 0;
    for (u32 j = 0; j < HS_RLC_BUCKETS; j++) {
        const u32 len = ghist[base + j + 1];
        boff[base + j] = acc; gcur[base + j] = acc; segoff[base + j] = lanes;
        acc += len; lanes += (len + C - 1) / C;
    }
    boff[base + HS_RLC_BUCKETS] = acc; segoff[base + HS_RLC_BUCKETS] = lanes;
}
__global__ void __launch_bounds__(1024)
k_rlc_scatter(const uint8_t *__restrict__ dig, u32 NP, u32 np, const u32 *__restrict__ key_idx, u32 nkeys,
              u32 *__restrict__ gcur, u32 *__restrict__ sorted) {
    __shared__ u32 hist[HS_RLC_BUCKETS + 1], cur[HS_RLC_BUCKETS + 1];
    const u32 w = blockIdx.y;
    const u32 *d4 = reinterpret_cast<const u32 *>(dig + (size_t)w * NP);
    for (u32 j = threadIdx.x; j <= HS_RLC_BUCKETS; j += 1024) hist[j] = 0;
    __syncthreads();
    u32 q0, q1;
    rlc_slice(q0, q1, NP, gridDim.x);
    for (u32 q = q0 + threadIdx.x; q < q1; q += 1024) {
        const u32 four = d4[q];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int dd = (int)((four >> (8 * b)) & 0xff) - 128;
            const u32 m = (u32)(dd < 0 ? -dd : dd);
            if (m) atomicAdd(&hist[m], 1u);
        }
    }
    __syncthreads();
    // reserve this block's share of every bucket's run with one global atomic per bucket
    for (u32 j = 1 + threadIdx.x; j <= HS_RLC_BUCKETS; j += 1024)
        cur[j] = hist[j] ? atomicAdd(&gcur[w * (HS_RLC_BUCKETS + 1) + j - 1], hist[j]) : 0u;
    __syncthreads();
    u32 *out = sorted + (size_t)w * NP;
    for (u32 q = q0 + threadIdx.x; q < q1; q += 1024) {
        const u32 four = d4[q];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int dd = (int)((four >> (8 * b)) & 0xff) - 128;
            const u32 m = (u32)(dd < 0 ? -dd : dd);
            if (m) {
                const u32 p = 4 * q + b;
                u32 id = p;
                if (key_idx && p >= np) {       // A served from the key set; an unknown key already failed the batch
                    const u32 ki = key_idx[p - np];
                    id = ki < nkeys ? (HS_RLC_ID_CACHED | ki) : NP;
                }
                out[atomicAdd(&cur[m], 1u)] = id | (dd < 0 ? HS_RLC_ID_NEG : 0u);
            }
        }
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
__device__ __forceinline__ void pt_add_if(ge_p3 &v, const ge_p3 &q, bool take) {     // v += take ? q : O, lane-uniform control flow
    ge_p3 ident, a;
    ge_identity(ident);
    fe_select(a.X, ident.X, q.X, take); fe_select(a.Y, ident.Y, q.Y, take);
    fe_select(a.Z, ident.Z, q.Z, take); fe_select(a.T, ident.T, q.T, take);
    ge_add_p3(v, v, a);
}
__device__ __forceinline__ void pt_load40(ge_p3 &p, const u32 *src) {
    const uint4 *q = reinterpret_cast<const uint4 *>(src);
    u32 w[40];
#pragma unroll
    for (int i = 0; i < 10; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#pragma unroll
    for (int i = 0; i < 10; i++) { p.X.v[i] = w[i]; p.Y.v[i] = w[10 + i]; p.Z.v[i] = w[20 + i]; p.T.v[i] = w[30 + i]; }
}
__device__ __forceinline__ void pt_store40(u32 *dst, const ge_p3 &p) {
    uint4 *q = reinterpret_cast<uint4 *>(dst);
    u32 w[40];
#pragma unroll
    for (int i = 0; i < 10; i++) { w[i] = p.X.v[i]; w[10 + i] = p.Y.v[i]; w[20 + i] = p.Z.v[i]; w[30 + i] = p.T.v[i]; }
#pragma unroll
    for (int i = 0; i < 10; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// Stage 3.  grid (ceil(Lcap / 64), 32): lane l of window w owns chunk l of the window's chunk list (bucket j
// owns chunks segoff[j] .. segoff[j + 1]); chunks of one bucket sit in consecutive lanes, so the wavefront then
// folds them with a SEGMENTED suffix scan (6 shuffle + add steps keyed by the bucket) and only the first lane of
// each bucket's stretch writes a partial: a bucket ends up with one partial per wavefront it spans.
#ifndef HS_RLC_ACCUM_WAVES
#define HS_RLC_ACCUM_WAVES 2       // wavefronts per SIMD the register allocator leaves room for (A/B: -DHS_RLC_ACCUM_WAVES=3)
#endif
__global__ void __launch_bounds__(64, HS_RLC_ACCUM_WAVES)
k_rlc_accum(const u32 *__restrict__ sorted, const u32 *__restrict__ boff, const u32 *__restrict__ segoff,
            const u32 *__restrict__ rec, const u32 *__restrict__ ksn, u32 NP, u32 C, u32 Lcap, u32 *__restrict__ part) {
    __shared__ u32 s_seg[HS_RLC_BUCKETS + 1], s_off[HS_RLC_BUCKETS + 1];
    const u32 w = blockIdx.y, lane = threadIdx.x, l = blockIdx.x * 64 + lane;
    const u32 base = w * (HS_RLC_BUCKETS + 1);
    for (u32 j = lane; j <= HS_RLC_BUCKETS; j += 64) { s_seg[j] = segoff[base + j]; s_off[j] = boff[base + j]; }
    __syncthreads();
    const u32 total = s_seg[HS_RLC_BUCKETS];
    if (blockIdx.x * 64 >= total) return;                   // the whole wavefront is past the window's last chunk
    const bool live = l < total;
    u32 j = 0;                                               // largest j with s_seg[j] <= l  (binary search, 7 steps)
#pragma unroll
    for (u32 step = 64; step >= 1; step >>= 1) { const u32 t = j + step; if (t < HS_RLC_BUCKETS && s_seg[t] <= l) j = t; }
    u32 a = 0, b = 0;
    if (live) {
        a = s_off[j] + (l - s_seg[j]) * C;
        b = min(a + C, s_off[j + 1]);
    }
    ge_p3 V;
    rlc_accum_lane(V, sorted + (size_t)w * NP, a, b, C, rec, ksn, NP);
    // segmented suffix scan; idle lanes carry a key of their own
    const int key = live ? (int)j : -1 - (int)lane;
#pragma nounroll
    for (int off = 1; off < 64; off <<= 1) {
        ge_p3 q;
        pt_shfl_down(q, V, off);
        const int kq = __shfl_down(key, off, 64);
        pt_add_if(V, q, lane + (u32)off < 64 && kq == key);
    }
    const int kprev = __shfl_up(key, 1, 64);
    if (live && (lane == 0 || kprev != key)) pt_store40(part + ((size_t)w * Lcap + l) * 40, V);
}

// Stage 4.  blocks 0..31: two wavefronts per window, one lane per bucket: fold the bucket's partials (one per
// wavefront of stage 3 that it spans), then sum_j j * bucket_j WITHOUT the 256-step running sum: with
// V_l = sum_{m >= l} B_m (suffix scan), sum_l (l + 1) B_l = sum_l V_l (rlc_bucket_fold is the serial definition).
// block 32: [b]B beside it.
__global__ void __launch_bounds__(128)
k_rlc_reduce(const u32 *__restrict__ part, const u32 *__restrict__ segoff, u32 Lcap, const u32 *__restrict__ btab,
             const unsigned long long *__restrict__ csum, u32 *__restrict__ wsum) {
    __shared__ u32 s_up[40];
    const u32 w = blockIdx.x, j = threadIdx.x, wave = j >> 6, lane = j & 63u;
    if (w == HS_RLC_WINDOWS) {         // [b]B on one quad: 32 table additions of two vector multiplications each
        if (j >= 4) return;
        unsigned long long tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // fold the slot copies (64-bit lazy sums: 2^20 signatures of 2^32 fit)
#pragma nounroll
        for (u32 sl = 0; sl < HS_RLC_CSUM_SLOTS; sl++) {
#pragma unroll
            for (int t = 0; t < 8; t++) tot[t] += csum[8 * sl + t];
        }
        const qv bb = rlc_b_point_quad(btab, tot);
        rlc_store_wsum_quad(wsum + (size_t)HS_RLC_WSUM_WORDS * HS_RLC_WINDOWS, bb);
        return;
    }
    ge_p3 V;
    ge_identity(V);
    {
        const u32 p0 = segoff[w * (HS_RLC_BUCKETS + 1) + j], end = segoff[w * (HS_RLC_BUCKETS + 1) + j + 1];
        const u32 *pw = part + (size_t)w * Lcap * 40;
        if (p0 < end) {
            pt_load40(V, pw + (size_t)p0 * 40);
            for (u32 p = (p0 & ~63u) + 64; p < end; p += 64) { ge_p3 q; pt_load40(q, pw + (size_t)p * 40); ge_add_p3(V, V, q); }
        }
    }
    // suffix scan over the wavefront's 64 buckets
#pragma nounroll
    for (int off = 1; off < 64; off <<= 1) {
        ge_p3 q;
        pt_shfl_down(q, V, off);
        pt_add_if(V, q, lane + (u32)off < 64);
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
        if (lane == 0) rlc_store_wsum(wsum + (size_t)HS_RLC_WSUM_WORDS * w, V);
    }
}

__global__ void __launch_bounds__(64)
k_rlc_final(const u32 *__restrict__ wsum, const u32 *__restrict__ failed, u32 *__restrict__ ok_flag, u32 *__restrict__ epoch,
            u32 *__restrict__ partial) {
    if (threadIdx.x >= 4) return;      // one quad
    const qv acc = rlc_final_point_quad(wsum);
    if (partial) {                     // a shard of a combination cut over several devices: hand the point over, test nothing here
        rlc_store_wsum_quad(partial, acc);
        if (threadIdx.x == 0) { partial[80] = *failed; *ok_flag = 0u; }
    } else {
        const qflags f = rlc_identity_quad(acc);
        const bool all = q_quad_all(f);
        if (threadIdx.x == 0) *ok_flag = (all && *failed == 0) ? 1u : 0u;
    }
    if (threadIdx.x == 0 && epoch) {   // the next execution on this workspace (a graph replay included) gets other coefficients
        const u32 lo = epoch[0] + 1;
        epoch[0] = lo;
        if (lo == 0) epoch[1] += 1;
    }
}
// the partial points of `count` shards (HS_RLC_PART_WORDS each) added up, the identity tested once: 1 = the whole batch accepted
__global__ void __launch_bounds__(64)
k_rlc_join(const u32 *__restrict__ parts, u32 count, u32 *__restrict__ ok_flag) {
    if (threadIdx.x >= 4) return;
    const qflags f = rlc_join_quad(parts, count);
    const bool all = q_quad_all(f);
    u32 failed = 0;
    for (u32 s = 0; s < count; s++) failed |= parts[(size_t)HS_RLC_PART_WORDS * s + 80];
    if (threadIdx.x == 0) *ok_flag = (all && failed == 0) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------ host
namespace {
struct rlc_ws {
    size_t np = 0;                 // capacity in (padded) signatures
    u32 *rec = nullptr; uint8_t *dig = nullptr; u32 *sorted = nullptr, *part = nullptr, *wsum = nullptr;
    u32 *tabs = nullptr;           // ghist | boff | segoff | gcur, each [32][129]
    uint8_t *small = nullptr;      // failed (4) | ok (4) | pad to 64 | csum[HS_RLC_CSUM_SLOTS][8] (64 B each) | epoch (8, never cleared per call)
    size_t min_batch = 0;          // 0 = the library default
    size_t calls = 0;
};
std::mutex k_mu;
rlc_key g_key;
bool g_key_set = false, g_key_pinned = false;
std::atomic<unsigned long long> g_call_seq{1};

// Below this many signatures the pipeline (five stages, the last a 248-doubling chain on one quad) costs
// more than the per-signature kernels: the call then runs plain BATCH.  Round 2 switched at 20,480; the per-signature
// kernels' half-size form (halfscalar.h) moved the latency crossover to ~45,000 signatures (profiles/r3_sweeps.md:
// 0.65 vs 0.76 ms at 32,768, 0.85 vs 0.81 ms at 49,152), while the RLC path keeps winning THROUGHPUT from ~8,000.
// A QC is one call on the node's critical path: the switch follows latency.
const size_t kRlcMinBatch = 40960;
const u32 kTabWords = HS_RLC_WINDOWS * (HS_RLC_BUCKETS + 1);
const size_t kSmallBytes = 64 + (size_t)HS_RLC_CSUM_SLOTS * 64;
const size_t kPartBytes = (size_t)HS_RLC_PART_WORDS * 4;
static_assert(HS_RLC_PART_WORDS * 4 == hs::kRlcPartBytes, "engine.h and rlc.h disagree on the partial record");
const u32 kMinChunk = 8;

void ws_free(rlc_ws *w) {
    if (w->rec) (void)hipFree(w->rec);
    if (w->dig) (void)hipFree(w->dig);
    if (w->sorted) (void)hipFree(w->sorted);
    if (w->part) (void)hipFree(w->part);
    if (w->wsum) (void)hipFree(w->wsum);
    if (w->tabs) (void)hipFree(w->tabs);
    if (w->small) (void)hipFree(w->small);
    w->rec = w->sorted = w->part = w->wsum = w->tabs = nullptr; w->dig = w->small = nullptr; w->np = 0;
}
u32 chunk_for(size_t n) {          // points per lane of stage 3: enough lanes to fill the chip, runs long enough to amortise the scan
    return n <= 16384 ? 8u : (n <= 131072 ? 16u : 32u);      // profiles/r2_rlc_tuning_ab.txt: 32 from 32,768 up was measured and lost
}
u32 lane_cap(size_t np, u32 C) { return (u32)round_up(2 * np / C + HS_RLC_BUCKETS, 64); }   // chunks per window, upper bound
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
int hs::rlc_reserve(hs_ed25519_ctx *ctx, size_t n, void *stream) {
    rlc_ws *w = ws_of(ctx);
    if (!w) return fail(HS_ED25519_E_NOMEM, "rlc: out of memory");
    const size_t np = round_up(n, 64);
    if (np <= w->np) return 0;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    HS_HIP(hipStreamSynchronize(ctx->stream), "stream sync");       // the old buffers may be in use
    if (stream && (hipStream_t)stream != ctx->stream) HS_HIP(hipStreamSynchronize((hipStream_t)stream), "stream sync");
    ws_free(w);
    const size_t NP = 2 * np;
    HS_HIP(hipMalloc((void **)&w->rec, (NP + 1) * 128), "hipMalloc rlc records");
    HS_HIP(hipMalloc((void **)&w->dig, (size_t)HS_RLC_WINDOWS * NP), "hipMalloc rlc digits");
    HS_HIP(hipMalloc((void **)&w->sorted, (size_t)HS_RLC_WINDOWS * NP * 4), "hipMalloc rlc sorted");
    HS_HIP(hipMalloc((void **)&w->part, (size_t)HS_RLC_WINDOWS * lane_cap(np, kMinChunk) * 160), "hipMalloc rlc partials");
    HS_HIP(hipMalloc((void **)&w->wsum, (size_t)(HS_RLC_WINDOWS + 1) * HS_RLC_WSUM_WORDS * 4), "hipMalloc rlc window sums");
    HS_HIP(hipMalloc((void **)&w->tabs, (size_t)4 * kTabWords * 4), "hipMalloc rlc offsets");
    HS_HIP(hipMalloc((void **)&w->small, kSmallBytes + 64 + kPartBytes + 64), "hipMalloc rlc flags");      // ... | epoch | partial record | join flag
    // the execution counter starts at 0 and only k_rlc_final writes it.  On the context's own stream, not the legacy stream: a legacy-
    // stream operation fails (hipErrorStreamCaptureImplicit) while another thread captures a round ring's graph (round 5, TSan run)
    HS_HIP(hipMemsetAsync(w->small + kSmallBytes, 0, 64, ctx->stream), "rlc clear epoch");
    HS_HIP(hipStreamSynchronize(ctx->stream), "rlc clear epoch");
    w->np = np;
    return 0;
}

extern "C" int hs_ed25519_set_rlc_key(const uint8_t key[32]) {
    std::lock_guard<std::mutex> lk(k_mu);
    if (!key) { g_key_set = false; g_key_pinned = false; return 0; }      // unpin: a fresh secret at the next call
    memcpy(g_key.w, key, 32);
    g_key_set = true;
    g_key_pinned = true;           // pinned: the same coefficients on every call
    return 0;
}
// The key of ONE call.  dalek draws fresh thread_rng() coefficients for every verify_batch call; a process-wide key
// alone would give the same signature at the same index the same z_i on every call, which on crafted torsion inputs
// (SURVEY.md 8a class iii) turns dalek's independent 1/8 chance per verification into a permanent, learnable verdict
// (ADVICE r2).  So unless the key was pinned with hs_ed25519_set_rlc_key (tests, reproducible runs), every call mixes its
// own 64-bit sequence number into the process secret: z_i = SHA-512(secret ^ call || h_i || s_i || i) - a keyed PRF on a
// never-repeating input, i.e. coefficients independent from call to call.
static int current_key(rlc_key &out, bool &pinned) {
    std::lock_guard<std::mutex> lk(k_mu);
    pinned = g_key_pinned;
    if (!g_key_set) {              // dalek: thread_rng(); here 32 bytes from the OS, once per process
        FILE *f = fopen("/dev/urandom", "rb");
        const bool ok = f && fread(g_key.w, 1, 32, f) == 32;
        if (f) fclose(f);
        if (!ok) return fail(HS_ED25519_E_DEVICE, "rlc: cannot read /dev/urandom for the coefficient key");
        g_key_set = true;
    }
    out = g_key;
    if (!g_key_pinned) {
        const unsigned long long seq = g_call_seq.fetch_add(1, std::memory_order_relaxed);
        out.w[6] ^= (u32)seq; out.w[7] ^= (u32)(seq >> 32);
    }
    return 0;
}
int hs::rlc_call_key(rlc_key_words &out, bool &pinned) {
    rlc_key k;
    const int rc = current_key(k, pinned);
    if (!rc) memcpy(out.w, k.w, 32);
    return rc;
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
// accepted the batch outright (1), fell through to the per-signature kernels (0), or did not run
// (HS_ED25519_RLC_NOT_RUN = 2: a state, deliberately outside the negative HS_ED25519_E_* range)
extern "C" int hs_ed25519_ctx_rlc_last(hs_ed25519_ctx *ctx, size_t *calls) {
    if (!ctx) return fail(HS_ED25519_E_ARG, "ctx_rlc_last: null context");
    rlc_ws *w = (rlc_ws *)ctx->rlc;
    if (calls) *calls = w ? w->calls : 0;
    if (!w || !w->small || !w->calls) return HS_ED25519_RLC_NOT_RUN;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    HS_HIP(hipStreamSynchronize(ctx->stream), "stream sync");
    u32 flags[2] = {0, 0};
    HS_HIP(hipMemcpyAsync(flags, w->small, 8, hipMemcpyDeviceToHost, ctx->stream), "read rlc flags");        // failed, ok (never the legacy stream)
    HS_HIP(hipStreamSynchronize(ctx->stream), "read rlc flags");
    return flags[1] ? 1 : 0;
}

namespace {
// what distinguishes a shard of a combination cut over several devices from a whole call
struct rlc_shard_opts { const rlc_key *key = nullptr; u32 index_base = 0; bool partial = false; };
int enqueue_rlc_body(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint8_t *d_pks,
                     const uint32_t *d_key_idx, const uint8_t *d_sigs, size_t n, u32 **d_ok_flag, void *stream, const rlc_shard_opts &opt);
}
int hs::enqueue_rlc(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint8_t *d_pks,
                    const uint32_t *d_key_idx, const uint8_t *d_sigs, size_t n, u32 **d_ok_flag, void *stream) {
    return enqueue_rlc_body(ctx, ks, msgs, d_pks, d_key_idx, d_sigs, n, d_ok_flag, stream, rlc_shard_opts());
}
// One shard's part of a combination: the pipeline over this shard's n signatures under the CALL's key, coefficients drawn for
// positions index_base ... index_base + n - 1, ending in the partial record (*d_partial: HS_RLC_PART_WORDS words in the context's
// workspace) instead of an identity test.  Any n >= 1 (no minimum: the caller chose this path).
int hs::enqueue_rlc_partial(hs_ed25519_ctx *ctx, msg_src msgs, const uint8_t *d_pks, const uint8_t *d_sigs, size_t n, u32 index_base,
                            const rlc_key_words &key, u32 **d_partial, void *stream) {
    rlc_key k;
    memcpy(k.w, key.w, 32);
    rlc_shard_opts opt;
    opt.key = &k; opt.index_base = index_base; opt.partial = true;
    u32 *flag = nullptr;
    const int rc = enqueue_rlc_body(ctx, nullptr, msgs, d_pks, nullptr, d_sigs, n, &flag, stream, opt);
    if (rc) return rc;
    rlc_ws *w = (rlc_ws *)ctx->rlc;
    *d_partial = (u32 *)(w->small + kSmallBytes + 64);
    return 0;
}
// `count` partial records at d_parts (this device's memory, or pinned host memory mapped on it) -> one verdict word in the
// context's workspace (*d_flag); the context must have an RLC workspace (rlc_reserve)
int hs::enqueue_rlc_join(hs_ed25519_ctx *ctx, const u32 *d_parts, u32 count, u32 **d_flag, void *stream) {
    rlc_ws *w = (rlc_ws *)ctx->rlc;
    if (!w || !w->small || !count) return fail(HS_ED25519_E_ARG, "rlc join: no workspace");
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    u32 *flag = (u32 *)(w->small + kSmallBytes + 64 + kPartBytes);
    void *a[] = { &d_parts, &count, &flag };
    HS_HIP(hipLaunchKernel((const void *)k_rlc_join, dim3(1), dim3(64), a, 0, st), "launch k_rlc_join");
    *d_flag = flag;
    return 0;
}

namespace {
int enqueue_rlc_body(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint8_t *d_pks,
                     const uint32_t *d_key_idx, const uint8_t *d_sigs, size_t n, u32 **d_ok_flag, void *stream, const rlc_shard_opts &opt) {
    *d_ok_flag = nullptr;
    if (!d_sigs || (!ks && !d_pks) || (ks && !d_key_idx)) return fail(HS_ED25519_E_ARG, "verify_device: null pointer");
    if (misaligned(d_sigs, 16) || misaligned(d_pks, 16) || misaligned(d_key_idx, 4)) return fail(HS_ED25519_E_ARG, "verify_device: misaligned pointer");
    if (ks && ks->slot != ctx->slot) return fail(HS_ED25519_E_ARG, "verify_device_cached: the key set lives on another device slot than the context");
    rlc_ws *w = ws_of(ctx);
    if (!w) return fail(HS_ED25519_E_NOMEM, "rlc: out of memory");
    const size_t min_batch = opt.partial ? 1 : (w->min_batch ? w->min_batch : kRlcMinBatch);
    if (n < min_batch || n > 0x0fffffffu) return opt.partial ? fail(HS_ED25519_E_ARG, "rlc partial: bad n") : 0;      // too small to pay for the pipeline (or absurdly large): plain per-signature BATCH
    const u32 *btab = dev_base_table(ctx->slot);
    if (!btab) return HS_ED25519_E_DEVICE;
    int rc = rlc_reserve(ctx, n, stream);
    if (rc) return rc;
    rlc_key key;
    bool pinned = false;
    if (opt.key) { key = *opt.key; pinned = true; }      // a shard: the call's key as drawn by the caller, no per-execution counter
    else {
        rc = current_key(key, pinned);
        if (rc) return rc;
    }
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const u32 np = (u32)round_up(n, 64), NP = 2 * np, nn = (u32)n, C = chunk_for(n), Lcap = lane_cap(np, C);
    u32 G = NP / 16384; G = G < 1 ? 1 : (G > 16 ? 16 : G);
    rlc_view v;
    v.rec = w->rec; v.dig = w->dig; v.csum = (unsigned long long *)(w->small + 64); v.failed = (u32 *)w->small; v.np = np;
    u32 *ok_flag = (u32 *)(w->small + 4);
    u32 *epoch = pinned ? nullptr : (u32 *)(w->small + kSmallBytes);
    u32 *ghist = w->tabs, *boff = w->tabs + kTabWords, *segoff = w->tabs + 2 * kTabWords, *gcur = w->tabs + 3 * kTabWords;
    HS_HIP(hipMemsetAsync(w->small, 0, kSmallBytes, st), "rlc clear flags");
    HS_HIP(hipMemsetAsync(ghist, 0, (size_t)kTabWords * 4, st), "rlc clear histograms");
    HS_HIP(hipMemsetAsync(w->dig, 0x80, (size_t)HS_RLC_WINDOWS * NP, st), "rlc clear digits");
    rlc_keys_src keys;
    keys.pks = ks ? nullptr : (const u32 *)d_pks; keys.key_idx = ks ? d_key_idx : nullptr;
    keys.ks_pk = ks ? ks->d_pk : nullptr; keys.ks_flag = ks ? ks->d_flag : nullptr; keys.nkeys = ks ? (u32)ks->nkeys : 0;
    const u32 *p_sg = (const u32 *)d_sigs;
    const u32 tasks = (ks ? 2u : 3u) * np;                  // np lanes per task kind (k_rlc_prepare)
    const uint8_t *p_dig = w->dig;
    {
        const u32 *p_epoch = epoch;
        u32 index_base = opt.index_base;
        void *a[] = { &v, &key, &p_epoch, &msgs, &keys, &p_sg, (void *)&nn, &index_base };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_prepare, dim3((tasks + 63) / 64), dim3(64), a, 0, st), "launch k_rlc_prepare");
    }
    {
        void *a[] = { &p_dig, (void *)&NP, &ghist };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_hist, dim3(G, HS_RLC_WINDOWS), dim3(1024), a, 0, st), "launch k_rlc_hist");
    }
    {
        const u32 *p_hist = ghist;
        void *a[] = { &p_hist, (void *)&C, &boff, &segoff, &gcur };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_offsets, dim3(1), dim3(64), a, 0, st), "launch k_rlc_offsets");
    }
    {
        const u32 *p_ki = keys.key_idx; u32 *p_sorted = w->sorted;
        void *a[] = { &p_dig, (void *)&NP, (void *)&np, &p_ki, &keys.nkeys, &gcur, &p_sorted };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_scatter, dim3(G, HS_RLC_WINDOWS), dim3(1024), a, 0, st), "launch k_rlc_scatter");
    }
    {
        const u32 *p_sorted = w->sorted, *p_boff = boff, *p_seg = segoff, *p_rec = w->rec, *p_ksn = ks ? ks->d_niels : nullptr;
        u32 *p_part = w->part;
        void *a[] = { &p_sorted, &p_boff, &p_seg, &p_rec, &p_ksn, (void *)&NP, (void *)&C, (void *)&Lcap, &p_part };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_accum, dim3(Lcap / 64, HS_RLC_WINDOWS), dim3(64), a, 0, st), "launch k_rlc_accum");
    }
    {
        const u32 *p_part = w->part, *p_seg = segoff; const unsigned long long *p_csum = v.csum; u32 *p_wsum = w->wsum;
        void *a[] = { &p_part, &p_seg, (void *)&Lcap, &btab, &p_csum, &p_wsum };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_reduce, dim3(HS_RLC_WINDOWS + 1), dim3(128), a, 0, st), "launch k_rlc_reduce");
    }
    {
        const u32 *p_wsum = w->wsum; const u32 *p_fail = v.failed;
        u32 *p_partial = opt.partial ? (u32 *)(w->small + kSmallBytes + 64) : nullptr;
        void *a[] = { &p_wsum, &p_fail, &ok_flag, &epoch, &p_partial };
        HS_HIP(hipLaunchKernel((const void *)k_rlc_final, dim3(1), dim3(64), a, 0, st), "launch k_rlc_final");
    }
    w->calls++;
    *d_ok_flag = ok_flag;
    return 0;
}
}  // namespace
