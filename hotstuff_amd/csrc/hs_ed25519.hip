// hs_ed25519.hip - gfx950 kernels of the per-signature paths and the C-ABI host side of the batch
// Ed25519 verification engine (include/hs_ed25519.h).  Replaces the ed25519-dalek calls at
// /root/reference/crypto/src/lib.rs:203 (verify_strict) and :218 (verify_batch).
//
// Device side (one wavefront = 64 lanes per role; DESIGN.md section 5 has the table):
//   cold path   k_verify_fused            n <= 32,768: four lanes per signature, one launch
//               k_verify_fused_batches    the same for up to 16 independent small batches per grid
//               k_prepare + k_verify      above: one lane per point / signature, SoA workspace
//   key cache   k_keyset_decompress, k_keyset_build   registration (tables in HBM)
//               k_verify_cached_lane2r    up to 98,304 signatures and for batch lists: two lanes per signature + a
//                                         wavefront decompressing the R's beside the chains
//               k_cached_chain2 + k_cached_finish   above: two-lane chains, then a batched finish by compression
//   k_tc_digests                          timeout-certificate digests ahead of any of the above
//   (the beaten forms k_verify_cached, _5w, _lane2 of rounds 1-3 and the one-lane k_verify_cached_lane[_batches] - by selector
//    only since it lost its size range - left the tree in round 5: `git show cd3b106:hotstuff_amd/csrc/hs_ed25519.hip`,
//    numbers in profiles/HISTORY.md section 9)
// Every verify kernel takes an optional `skip` word: when the batch random-linear-combination
// check (hs_rlc.hip) has already accepted the whole batch it writes an all-ones bitmap and leaves.
// Host side here: init / device slots (one or more GPUs), contexts (stream + device buffers + optional
// pinned staging), key sets, the device-pointer entry points.  The host-pointer API (context pools,
// synchronous calls, tickets, the round ring) is hs_host.hip; hs_shard.hip fans one call out over
// the slots.  No CPU fallback anywhere.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <new>
#include <mutex>
#include <condition_variable>
#include <vector>
#include <string>
#include <utility>
#include "verify.h"
#include "verify_quad.h"
#include "verify_cached.h"
#include "engine.h"

using namespace hs;

// ------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------
__device__ const u32 *g_b16;                      // window 0 of the base point's 16-bit table: j*B, j <= 32,768, 128-byte entries
                                                  // (verify_cached.h; built and set by hs_ed25519_init_devices)

#ifndef HS_LANE_WAVES
#define HS_LANE_WAVES 2                 // wavefronts per SIMD the one-lane stage 2 is compiled for (3 = 168 VGPRs with 36 B of spills: measured equal)
#endif
// ... and THREE for its batch-list form (round 6).  Device-resident, deep queues: equal (104.6 - 105.1 against 103.9 - 104.6 M).  From
// host buffers, where every stream holds one or two launches: a third sixteen-batch launch now FITS its 64-CU slice instead of
// holding the slice's pipe until one of two ends (hs_host.hip pool_get, profiles/r6_x_lane_map.md) - bursts of eight 85 - 86 ->
// 88 - 90 M verifies/s, one thread with tickets 100 - 102 -> 104 M, and four tenants at bursts of sixteen 83.6 -> 92 M, whose
// blocked pipes are shared with the OTHER tenants' queues (profiles/r6_af_lane_waves_ab.md).
#ifndef HS_LANE_LIST_WAVES
#define HS_LANE_LIST_WAVES 3
#endif
// (64, 3): at most 168 VGPRs so that three wavefronts fit a SIMD next to the quad kernels
__global__ void __launch_bounds__(64, 3)
k_prepare(workspace ws, msg_src msgs, const u32 *__restrict__ pks, const u32 *__restrict__ sigs, u32 n, u32 mode,
          const u32 *__restrict__ skip) {
    if (skip && *skip) return;          // the batch fast-accept already vouched for every signature (hs_rlc.hip)
    const u32 t = blockIdx.x * 64 + threadIdx.x;
    if (t >= 2 * n) return;
    const bool is_r = t >= n;
    const u32 i = is_r ? t - n : t;
    u32 msg[8];
    if (is_r) load_msg(msg, msgs, i);
    prepare_point(ws, i, is_r, pks + (size_t)8 * i, sigs + (size_t)16 * i, msg, mode);
}

__global__ void __launch_bounds__(64, HS_LANE_WAVES)
k_verify(workspace ws, u32 n, u32 mode, unsigned long long *__restrict__ bitmap, u32 *__restrict__ nreject,
         const u32 *__restrict__ skip) {
    if (skip && *skip) {
        const u32 lanes = min(64u, n - blockIdx.x * 64);
        if (threadIdx.x == 0) bitmap[blockIdx.x] = lanes == 64 ? ~0ull : ((1ull << lanes) - 1);
        return;
    }
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    const bool active = i < n;
    // inactive tail lanes run lane min(i, n-1)'s data so the wavefront stays convergent
    const bool ok = verify_lane(ws, active ? i : n - 1, mode, g_b16) && active;
    const unsigned long long mask = __ballot(ok);
    if (threadIdx.x == 0) {
        bitmap[blockIdx.x] = mask;
        if (nreject) {
            const u32 lanes = min(64u, n - blockIdx.x * 64);
            const u32 bad = lanes - (u32)__popcll(mask);
            if (bad) atomicAdd(nreject, bad);
        }
    }
}

// The same pair over a list of independent batches (hs_ed25519_verify_device_batches from 16 batches per call): with
// enough signatures per launch to fill the chip the one-lane kernels do the least work per verification
// (profiles/r2_sweeps.md section 2: 74 M/s against the pairs form's 65), at several times the latency of a launch.
// Every batch has its own workspace slice, digests, bitmap and reject counter.
#define HS_MAX_BATCHES_PER_LAUNCH 16
#ifndef HS_MAX_LANE_BATCHES
#define HS_MAX_LANE_BATCHES 16          // measured with 32 (3 KB of kernel arguments): 87.5 M verifies/s against 87.2 at 16 - saturated
#endif
struct lane_batch { workspace ws; msg_src msgs; const u32 *pks, *sigs; unsigned long long *bitmap; u32 *nreject; u32 n, mode, first_p, first_v; };
struct lane_batch_list { lane_batch b[HS_MAX_LANE_BATCHES]; u32 count; };
// Stage 1 over the list, with a wavefront's inputs brought in COALESCED and exactly once: a wavefront is 64 A tasks or 64 R tasks
// of one batch (blocks [0, ceil(n/64)) of a batch are its A blocks, the next ceil(n/64) its R blocks), reads its 64 keys
// (2 KB) - and, an R wavefront, its 64 signatures (4 KB) and 64 digests (2 KB) - as consecutive 16-byte pieces, lane l
// piece l, l + 64, ... (every load instruction covers one contiguous kilobyte), parks them in LDS and picks its own record
// from there.  Per-lane record loads (k_prepare's form) are dwordx4 loads at a lane stride of 32 / 64 bytes: every instruction
// touches 64 different lines and uses 16 bytes of each.  Same box, alternating (profiles/r4_f_staged_ab.md): HBM-resident
// rows unchanged (the L2 forgives the strided form), host calls - whose inputs are pinned memory read over PCIe - +1 - 3.5 %,
// and the kernel lost its 20 bytes of scratch.
template <int WORDS>      // words per record: 8 (keys, digests) or 16 (signatures)
__device__ __forceinline__ void stage_records(u32 *lds, const u32 *__restrict__ src, u32 first, u32 count) {
    const uint4 *g = reinterpret_cast<const uint4 *>(src + (size_t)WORDS * first);
    uint4 *s = reinterpret_cast<uint4 *>(lds);
    const u32 pieces = count * (WORDS / 4);
#pragma unroll
    for (int j = 0; j < WORDS / 4; j++) {
        const u32 o = j * 64 + threadIdx.x;
        if (o < pieces) s[o] = g[o];
    }
}
__global__ void __launch_bounds__(64, 3)
k_prepare_batches(lane_batch_list L) {
    __shared__ __attribute__((aligned(16))) u32 s_in[64 * 32];      // [keys 64 x 8][signatures 64 x 16][digests 64 x 8]
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_LANE_BATCHES; t++) if (t < L.count && blockIdx.x >= L.b[t].first_p) k = t;
    const lane_batch &B = L.b[k];
    const u32 nb = (B.n + 63) / 64, blk = blockIdx.x - B.first_p;
    const bool is_r = blk >= nb;
    const u32 i0 = (is_r ? blk - nb : blk) * 64;
    const u32 cnt = min(64u, B.n - i0);
    u32 *s_pk = s_in, *s_sig = s_in + 64 * 8, *s_msg = s_in + 64 * 24;
    stage_records<8>(s_pk, B.pks, i0, cnt);
    const bool per_tuple = B.msgs.stride_words != 0;
    if (is_r) {
        stage_records<16>(s_sig, B.sigs, i0, cnt);
        if (per_tuple) stage_records<8>(s_msg, B.msgs.digests, i0, cnt);
    }
    __syncthreads();
    if (threadIdx.x >= cnt) return;
    u32 msg[8];
    if (is_r) {
        if (per_tuple) {
#pragma unroll
            for (int t = 0; t < 8; t++) msg[t] = s_msg[8 * threadIdx.x + t];
        } else load_msg(msg, B.msgs, 0);
    }
    prepare_point(B.ws, i0 + threadIdx.x, is_r, s_pk + 8 * threadIdx.x, s_sig + 16 * threadIdx.x, msg, B.mode);
}
__global__ void __launch_bounds__(64, HS_LANE_LIST_WAVES)
k_verify_batches(lane_batch_list L) {
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_LANE_BATCHES; t++) if (t < L.count && blockIdx.x >= L.b[t].first_v) k = t;
    const lane_batch &B = L.b[k];
    const u32 block = blockIdx.x - B.first_v;
    const u32 i = block * 64 + threadIdx.x;
    const bool active = i < B.n;
    const bool ok = verify_lane(B.ws, active ? i : B.n - 1, B.mode, g_b16) && active;
    const unsigned long long mask = __ballot(ok);
    if (threadIdx.x == 0) {
        B.bitmap[block] = mask;
        if (B.nreject) {
            const u32 lanes = min(64u, B.n - block * 64);
            const u32 bad = lanes - (u32)__popcll(mask);
            if (bad) atomicAdd(B.nreject, bad);
        }
    }
}

// TC::verify's per-vote digests SHA-512(round_le || high_qc_round_le)[..32]
// (consensus/src/messages.rs:307-311), one lane per vote, ahead of the verify kernels.
__global__ void __launch_bounds__(64)
k_tc_digests(unsigned long long round, const unsigned long long *__restrict__ hq_rounds, u32 n, u32 *__restrict__ out) {
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 d[8];
    tc_vote_digest(d, round, hq_rounds[i]);
    uint4 *o = reinterpret_cast<uint4 *>(out + (size_t)8 * i);
    o[0] = make_uint4(d[0], d[1], d[2], d[3]);
    o[1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// Four lanes per signature (verify_quad.h): a wavefront covers 16 signatures, so a
// 1,024-signature batch is 64 wavefronts on 64 CUs instead of 16.

// Whole verification in one launch (verify_quad_fused): reads the packed inputs, writes the bitmap.
// `block` = this wavefront's index within its batch (16 signatures per wavefront).
template <int W>
__device__ __forceinline__ void fused_block(u32 *s_atab, u32 block, msg_src msgs, const u32 *__restrict__ pks,
                                            const u32 *__restrict__ sigs, u32 n, u32 mode, unsigned short *__restrict__ bitmap16,
                                            u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    if (skip && *skip) {
        const u32 first = block * 16;
        const u32 here = first < n ? min(16u, n - first) : 0u;
        if (threadIdx.x == 0) bitmap16[block] = (unsigned short)((1u << here) - 1);
        return;
    }
    qtab atab, rtab; atab.lds = s_atab; rtab.lds = s_atab + HS_QTAB_WORDS_OF(W);
    const u32 i = block * 16 + (threadIdx.x >> 2);
    const bool active = i < n;
    const u32 j = active ? i : n - 1;
    u32 msg[8];
    load_msg(msg, msgs, j);
    const bool ok = verify_quad_fused<W>(pks + (size_t)8 * j, sigs + (size_t)16 * j, msg, mode, g_b16, atab, rtab);
    unsigned long long m = __ballot(ok && active);
    m &= m >> 1; m &= m >> 2; m &= 0x1111111111111111ull;
    m = (m | (m >> 3)) & 0x0303030303030303ull;
    m = (m | (m >> 6)) & 0x000f000f000f000full;
    m = (m | (m >> 12)) & 0x000000ff000000ffull;
    m = (m | (m >> 24)) & 0xffffull;
    if (threadIdx.x == 0) {
        bitmap16[block] = (unsigned short)m;
        if (nreject) {
            const u32 first = block * 16;
            const u32 sigs_here = first < n ? min(16u, n - first) : 0u;
            const u32 bad = sigs_here - (u32)__popcll(m);
            if (bad) atomicAdd(nreject, bad);
        }
    }
}
// 16-bit ballot of a wavefront of quads: signature q accepted iff all four lanes of quad q agree
__device__ __forceinline__ u32 quad_ballot16(bool ok) {
    unsigned long long m = __ballot(ok);
    m &= m >> 1; m &= m >> 2; m &= 0x1111111111111111ull;
    m = (m | (m >> 3)) & 0x0303030303030303ull;
    m = (m | (m >> 6)) & 0x000f000f000f000full;
    m = (m | (m >> 12)) & 0x000000ff000000ffull;
    m = (m | (m >> 24)) & 0xffffull;
    return (u32)m;
}
// The pairs form (verify_quad.h, quad_pair_*): `block` = this wavefront's index within its batch, 32 signatures per
// wavefront - quad q verifies signature 32 block + q, then 32 block + 16 + q, so each pass fills one 16-bit bitmap word.
__device__ __forceinline__ void fused_pairs_block(u32 *s_atab, u32 block, msg_src msgs, const u32 *__restrict__ pks,
                                                  const u32 *__restrict__ sigs, u32 n, u32 mode, unsigned short *__restrict__ bitmap16,
                                                  u32 *__restrict__ nreject) {
    qtab atab, rtab; atab.lds = s_atab; rtab.lds = s_atab + HS_QTAB_WORDS_OF(3);
    const u32 q = threadIdx.x >> 2;
    const u32 i0 = block * 32 + q, i1 = i0 + 16;
    const u32 j0 = i0 < n ? i0 : n - 1, j1 = i1 < n ? i1 : n - 1;
    const quad_pair_points pp = quad_pair_decompress<3>(pks, sigs, msgs, j0, j1, mode);
    u32 bad = 0;
#pragma nounroll
    for (u32 h = 0; h < 2; h++) {
        const u32 i = h ? i1 : i0;
        const bool ok = quad_pair_pass<3>(pp, h, mode, g_b16, atab, rtab);
        const u32 m = quad_ballot16(ok && i < n);
        if (threadIdx.x == 0) bitmap16[2 * block + h] = (unsigned short)m;
        const u32 first = block * 32 + 16 * h;
        bad += (first < n ? min(16u, n - first) : 0u) - (u32)__popc(m);
    }
    if (threadIdx.x == 0 && nreject && bad) atomicAdd(nreject, bad);
}
// The latency form: 4-bit windows, both tables of eight entries in LDS (40,960 B: one wavefront per SIMD - a single batch
// of up to 16,384 signatures is at most that anyway)
__global__ void __launch_bounds__(64, 1)
k_verify_fused(msg_src msgs, const u32 *__restrict__ pks, const u32 *__restrict__ sigs, u32 n, u32 mode,
               unsigned short *__restrict__ bitmap16, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    __shared__ u32 s_atab[2 * HS_QTAB_WORDS_OF(4)];    // this wavefront's tables j*(+-A) and j*(-R), j = 1..8 (verify_quad.h)
    fused_block<4>(s_atab, blockIdx.x, msgs, pks, sigs, n, mode, bitmap16, nreject, skip);
}
// EIGHT lanes per signature (verify_oct_fused): one quad per per-signature table, 8 signatures per wavefront - single batches
// of up to 8,192 signatures, which are at most one wavefront per SIMD even so.  15 % off the cold latency floor.
__global__ void __launch_bounds__(64, 2)
k_verify_fused8(msg_src msgs, const u32 *__restrict__ pks, const u32 *__restrict__ sigs, u32 n, u32 mode,
                unsigned char *__restrict__ bitmap8, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    __shared__ u32 s_tab[HS_QTAB_WORDS_OF(4)];      // per quad: the eight entries of ITS table
    const u32 block = blockIdx.x;
    if (skip && *skip) {
        const u32 first = block * 8;
        const u32 here = first < n ? min(8u, n - first) : 0u;
        if (threadIdx.x == 0) bitmap8[block] = (unsigned char)((1u << here) - 1);
        return;
    }
    qtab tab; tab.lds = s_tab;
    const u32 i = block * 8 + (threadIdx.x >> 3);
    const bool active = i < n;
    const u32 j = active ? i : n - 1;
    u32 msg[8];
    load_msg(msg, msgs, j);
    const bool ok = verify_oct_fused(pks + (size_t)8 * j, sigs + (size_t)16 * j, msg, mode, g_b16, tab);
    unsigned long long m = __ballot(ok && active);
    // signature o accepted iff all eight lanes of its octet agree; gather bit 8 o -> bit o
    m &= m >> 1; m &= m >> 2; m &= m >> 4; m &= 0x0101010101010101ull;
    m = (m | (m >> 7)) & 0x0003000300030003ull;
    m = (m | (m >> 14)) & 0x0000000f0000000full;
    m = (m | (m >> 28)) & 0xffull;
    if (threadIdx.x == 0) {
        bitmap8[block] = (unsigned char)m;
        if (nreject) {
            const u32 first = block * 8;
            const u32 here = first < n ? min(8u, n - first) : 0u;
            const u32 bad = here - (u32)__popcll(m);
            if (bad) atomicAdd(nreject, bad);
        }
    }
}
// The eight-lane form split over TWO wavefronts per eight signatures (verify_quad.h, oct_split_*): the helper wavefront hashes
// and reduces while the main one takes the square roots and builds its tables - 51 us of scalar work off a 273 us chain.  The
// lone call's default (hs_ed25519_ctx_set_concurrent and ring slots keep four lanes per signature, as for k_verify_fused8).
__global__ void __launch_bounds__(128, 2)
k_verify_fused8h(msg_src msgs, const u32 *__restrict__ pks, const u32 *__restrict__ sigs, u32 n, u32 mode,
                 unsigned char *__restrict__ bitmap8, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    __shared__ u32 s_tab[HS_QTAB_WORDS_OF(4)];      // per quad of the main wavefront: the eight entries of ITS table
    __shared__ u32 s_sc[HS_OCT_SC_WORDS * 64];      // the scalars, helper -> main, [word][lane]
    const u32 block = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (skip && *skip) {
        const u32 first = block * 8;
        const u32 here = first < n ? min(8u, n - first) : 0u;
        if (threadIdx.x == 0) bitmap8[block] = (unsigned char)((1u << here) - 1);
        return;
    }
    qtab tab; tab.lds = s_tab;
    const u32 i = block * 8 + (lane >> 3);
    const bool active = i < n;
    const u32 j = active ? i : n - 1;
    const u32 *pk = pks + (size_t)8 * j, *sig = sigs + (size_t)16 * j;
    u32 fl = 0;
    if (wave == 1) {
        u32 msg[8];
        load_msg(msg, msgs, j);
        oct_split_scalars(s_sc, lane, pk, sig, msg);
    } else {
        fl = oct_split_point(pk, sig, mode, tab);
    }
    __syncthreads();
    if (wave == 1) return;
    const bool ok = oct_split_walk(s_sc, fl, mode, g_b16, tab);
    unsigned long long m = __ballot(ok && active);
    // signature o accepted iff all eight lanes of its octet agree; gather bit 8 o -> bit o
    m &= m >> 1; m &= m >> 2; m &= m >> 4; m &= 0x0101010101010101ull;
    m = (m | (m >> 7)) & 0x0003000300030003ull;
    m = (m | (m >> 14)) & 0x0000000f0000000full;
    m = (m | (m >> 28)) & 0xffull;
    if (lane == 0) {
        bitmap8[block] = (unsigned char)m;
        if (nreject) {
            const u32 first = block * 8;
            const u32 here = first < n ? min(8u, n - first) : 0u;
            const u32 bad = here - (u32)__popcll(m);
            if (bad) atomicAdd(nreject, bad);
        }
    }
}
// The same with 3-bit windows: four entries per table, 20,480 B of LDS, two wavefronts per SIMD - single batches between
// 16,384 and 32,768 signatures
__global__ void __launch_bounds__(64, 2)
k_verify_fused3(msg_src msgs, const u32 *__restrict__ pks, const u32 *__restrict__ sigs, u32 n, u32 mode,
                unsigned short *__restrict__ bitmap16, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    __shared__ u32 s_atab[2 * HS_QTAB_WORDS_OF(3)];
    fused_block<3>(s_atab, blockIdx.x, msgs, pks, sigs, n, mode, bitmap16, nreject, skip);
}
// The one-lane stage 2 over whole rounds of wavefronts AND, in the same grid, the sliver past them through the four-lane
// fused form with 3-bit windows (20,480 B of LDS per workgroup: two wavefronts per SIMD still fit, so the one-lane
// wavefronts lose nothing).  The first `tail_blocks` workgroups are the sliver's (16 signatures each, dispatched first); the
// rest are k_verify's.  One launch on one stream: no second stream, no events, the same under graph capture (enqueue_cold).
__global__ void __launch_bounds__(64, HS_LANE_WAVES)
k_verify_tail(workspace ws, u32 n, u32 mode, unsigned long long *__restrict__ bitmap, u32 *__restrict__ nreject,
              msg_src tmsgs, const u32 *__restrict__ t_pks, const u32 *__restrict__ t_sigs, u32 tn,
              unsigned short *__restrict__ t_bitmap16, u32 tail_blocks) {
    __shared__ u32 s_atab[2 * HS_QTAB_WORDS_OF(3)];
    if (blockIdx.x < tail_blocks) {
        fused_block<3>(s_atab, blockIdx.x, tmsgs, t_pks, t_sigs, tn, mode, t_bitmap16, nreject, nullptr);
        return;
    }
    const u32 block = blockIdx.x - tail_blocks;
    const u32 i = block * 64 + threadIdx.x;
    const bool active = i < n;
    const bool ok = verify_lane(ws, active ? i : n - 1, mode, g_b16) && active;
    const unsigned long long mask = __ballot(ok);
    if (threadIdx.x == 0) {
        bitmap[block] = mask;
        if (nreject) {
            const u32 lanes = min(64u, n - block * 64);
            const u32 bad = lanes - (u32)__popcll(mask);
            if (bad) atomicAdd(nreject, bad);
        }
    }
}
// Several independent small batches in ONE grid (hs_ed25519_verify_device_batches): a 1,024-signature
// batch is only 64 wavefronts - at most one per SIMD of the CU slice its stream is confined to - and a lone
// wavefront issues at about half the rate two co-resident ones reach (profiles/r1_ubench_valu.txt).
// Back-to-back launches of one stream do not overlap and more than 16 streams collapse
// (profiles/r1_stream_sweep.txt), so the way to more wavefronts per SIMD is more batches per launch.
struct fused_batch { msg_src msgs; const u32 *pks, *sigs; unsigned short *bitmap16; u32 *nreject; u32 n, mode, first_block, pad; };
struct fused_batch_list { fused_batch b[HS_MAX_BATCHES_PER_LAUNCH]; u32 count; };
__global__ void __launch_bounds__(64, 2)
k_verify_fused_batches(fused_batch_list L) {
    __shared__ u32 s_atab[2 * HS_QTAB_WORDS_OF(3)];    // this wavefront's tables j*(+-A) and j*(-R), j = 1..4: two wavefronts per SIMD
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_BATCHES_PER_LAUNCH; t++) if (t < L.count && blockIdx.x >= L.b[t].first_block) k = t;
    const fused_batch &B = L.b[k];
    fused_block<3>(s_atab, blockIdx.x - B.first_block, B.msgs, B.pks, B.sigs, B.n, B.mode, B.bitmap16, B.nreject, nullptr);
}
// The same list served by the pairs form: first_block counts 32-signature wavefronts.
__global__ void __launch_bounds__(64, 2)
k_verify_fused_pairs_batches(fused_batch_list L) {
    __shared__ u32 s_atab[2 * HS_QTAB_WORDS_OF(3)];
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_BATCHES_PER_LAUNCH; t++) if (t < L.count && blockIdx.x >= L.b[t].first_block) k = t;
    const fused_batch &B = L.b[k];
    fused_pairs_block(s_atab, blockIdx.x - B.first_block, B.msgs, B.pks, B.sigs, B.n, B.mode, B.bitmap16, B.nreject);
}

// ---- committee key cache (verify_cached.h) -------------------------------------------------
// registration pass 1: one lane per key - decompress, flags, keep the key bytes
__global__ void __launch_bounds__(64, 3)
k_keyset_decompress(const u32 *__restrict__ pks, u32 nkeys, u32 *__restrict__ axy, u32 *__restrict__ pk_out,
                    u32 *__restrict__ flag, u32 *__restrict__ niels) {
    const u32 k = blockIdx.x * 64 + threadIdx.x;
    if (k >= nkeys) return;
    u32 enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { enc[j] = pks[(size_t)8 * k + j]; pk_out[(size_t)8 * k + j] = enc[j]; }
    ge_p3 p;
    u32 fl = ge_decompress(p, enc) ? HS_F_ONCURVE : 0;
    fl |= ge_is_small_order_affine(p.Y) ? HS_F_SMALL : 0;
    flag[k] = fl;
#pragma unroll
    for (int j = 0; j < 10; j++) { axy[(size_t)20 * k + j] = p.X.v[j]; axy[(size_t)20 * k + 10 + j] = p.Y.v[j]; }
    // affine Niels record of +A (y+x, y-x, 2dxy; 128 bytes) for the batch MSM path (rlc.h)
    fe ypx, ymx, xy2d;
    fe_add(ypx, p.Y, p.X); fe_carry(ypx);
    fe_sub(ymx, p.Y, p.X); fe_carry(ymx);
    fe_mul(xy2d, p.T, fe_d2());
    u32 *rec = niels + (size_t)32 * k;
#pragma unroll
    for (int j = 0; j < 10; j++) { rec[j] = ypx.v[j]; rec[10 + j] = ymx.v[j]; rec[20 + j] = xy2d.v[j]; }
    rec[30] = fl; rec[31] = 0;
}
// registration pass 2: one quad per (window, key); the 16 quads of a wavefront share a window
// so that their 8w doublings stay convergent
__global__ void __launch_bounds__(64)
k_keyset_build(u32 *__restrict__ tab, const u32 *__restrict__ axy, u32 nkeys, u32 nkeys_pad, u32 negate) {
    const u32 g = blockIdx.x * 16 + (threadIdx.x >> 2);
    const u32 w = g / nkeys_pad, k = g % nkeys_pad;
    if (k >= nkeys) return;     // whole quads leave together; DPP quad_perm never crosses a quad
    fe ax, ay;
#pragma unroll
    for (int j = 0; j < 10; j++) { ax.v[j] = axy[(size_t)20 * k + j]; ay.v[j] = axy[(size_t)20 * k + 10 + j]; }
    keyset_build_window(tab + (size_t)k * HS_KS_KEY_PROJ_WORDS + (size_t)w * HS_KS_ENTRIES * HS_KS_PROJ_WORDS,
                        ax, ay, negate != 0, (int)w);
}
// registration pass 3: one lane per group of eight entries - projective scratch -> the affine 128-byte entries the
// verification kernels read, one inversion per group (keyset_affine_group)
__global__ void __launch_bounds__(64, 2)
k_keyset_affine(const u32 *__restrict__ proj, u32 *__restrict__ tab, unsigned long long groups) {
    const unsigned long long g = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (g >= groups) return;
    keyset_affine_group(tab + g * (HS_KS_AFFINE_GROUP * HS_KS_ENTRY_WORDS), proj + g * (HS_KS_AFFINE_GROUP * HS_KS_PROJ_WORDS));
}
// the base point's 16-bit table (verify_cached.h): one lane per entry, projective into the scratch; k_keyset_affine
// then normalises it in groups of eight
__global__ void __launch_bounds__(64)
k_base16_build(const u32 *__restrict__ t8, u32 *__restrict__ proj) {
    const u32 e = blockIdx.x * 64 + threadIdx.x;
    if (e >= HS_B16_WINDOWS * HS_B16_ENTRIES) return;
    base16_entry_projective(proj + (size_t)e * HS_KS_PROJ_WORDS, t8, e / HS_B16_ENTRIES, e % HS_B16_ENTRIES);
}
// Two lanes per signature with the square root off the chains: a workgroup of THREE wavefronts
// covers 64 signatures.  Wavefronts 0 and 1 run the paired addition chains (32 signatures each);
// wavefront 2 meanwhile decompresses the 64 R's one lane each and hands (x_R, y_R, flags) over
// through LDS; the chains then finish with a projective comparison instead of an inversion that both
// lanes of a pair would execute.  Per 64 signatures: 2 x 52 K + 36 K instructions instead of 2 x 82 K.
// `block` = this workgroup's index within its batch (64 signatures per workgroup)
// HS_CACHED_STAGED (the batch-list kernel): the workgroup's 64 signatures (4 KB) and per-tuple digests (2 KB) come in as
// consecutive 16-byte pieces over all 192 lanes - every load instruction a contiguous 3 KB, every input byte fetched once -
// and are picked out of LDS; the per-lane record loads (false; the lone-call kernel) fetch the signature twice (chain pair, R
// wavefront) as 16-byte pieces at a 64-byte stride (profiles/r4_f_staged_ab.md).
#define HS_CACHED_STAGE_WORDS (64 * 24)     // [signatures 64 x 16][digests 64 x 8]
template <bool HS_CACHED_STAGED>
__device__ __forceinline__ void cached_lane2r_block(u32 (*s_r)[64], u32 *s_in, u32 block, const keyset_view &ks, u32 nkeys, const u32 *__restrict__ b16,
                                                    msg_src msgs, const u32 *__restrict__ key_idx, const u32 *__restrict__ sigs, u32 n, u32 mode,
                                                    u32 *__restrict__ bitmap32, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (skip && *skip) {
        const u32 first = block * 64 + wave * 32;
        const u32 here = first < n ? min(32u, n - first) : 0u;
        if (wave < 2 && lane == 0) bitmap32[block * 2 + wave] = here == 32 ? 0xffffffffu : ((1u << here) - 1);
        return;
    }
    const bool strict = (mode & HS_MODE_STRICT) != 0;
    ge_p3 acc;
    bool ok = false, active = false;
    const u32 i0 = block * 64;
    const u32 *s_sig = s_in, *s_msg = s_in + 64 * 16;
    if (HS_CACHED_STAGED) {
        const u32 cnt = min(64u, n - i0);
        const uint4 *gs = reinterpret_cast<const uint4 *>(sigs + (size_t)16 * i0);
        uint4 *ss = reinterpret_cast<uint4 *>(s_in);
        if (threadIdx.x < cnt * 4) ss[threadIdx.x] = gs[threadIdx.x];
        if (threadIdx.x + 192 < cnt * 4) ss[threadIdx.x + 192] = gs[threadIdx.x + 192];
        if (msgs.stride_words) {
            const uint4 *gm = reinterpret_cast<const uint4 *>(msgs.digests + (size_t)8 * i0);
            uint4 *sm = reinterpret_cast<uint4 *>(s_in + 64 * 16);
            if (threadIdx.x < cnt * 2) sm[threadIdx.x] = gm[threadIdx.x];
        }
        __syncthreads();
    }
    if (wave == 2) {
        const u32 i = block * 64 + lane;
        const u32 j = i < n ? i : n - 1;
        u32 enc[8];
#pragma unroll
        for (int t = 0; t < 8; t++) enc[t] = HS_CACHED_STAGED ? s_sig[16 * (j - i0) + t] : sigs[(size_t)16 * j + t];
        ge_p3 R;
        u32 fl = ge_decompress(R, enc) ? HS_F_ONCURVE : 0;
        if (strict) fl |= ge_is_small_order_affine(R.Y) ? HS_F_SMALL : 0;
#pragma unroll
        for (int t = 0; t < 10; t++) { s_r[t][lane] = R.X.v[t]; s_r[10 + t][lane] = R.Y.v[t]; }
        s_r[20][lane] = fl;
    } else {
        const u32 i = block * 64 + wave * 32 + (lane >> 1);
        active = i < n;
        const u32 j = active ? i : n - 1;
        const u32 ki_raw = key_idx[j];
        const bool known = ki_raw < nkeys;
        const u32 ki = known ? ki_raw : 0;
        u32 kw[8], sw[8], msg[8];
        if (HS_CACHED_STAGED) {
            if (msgs.stride_words) {
#pragma unroll
                for (int t = 0; t < 8; t++) msg[t] = s_msg[8 * (j - i0) + t];
            } else load_msg(msg, msgs, 0);
            ok = prepare_scalars(kw, sw, ks.pk + (size_t)8 * ki, s_sig + 16 * (j - i0), msg, true, true);
        } else {
            load_msg(msg, msgs, j);
            ok = prepare_scalars(kw, sw, ks.pk + (size_t)8 * ki, sigs + (size_t)16 * j, msg, true, true);
        }
        const u32 fl = ks.flag[ki];
        ok = ok && known && (fl & HS_F_ONCURVE);
        if (strict) ok = ok && !(fl & HS_F_SMALL);
        cached_lane2_chain(acc, ks.tab + (size_t)ki * HS_KS_KEY_WORDS, b16, kw, sw);
    }
    __syncthreads();
    if (wave == 2) return;
    const u32 slot = wave * 32 + (lane >> 1);
    fe rx, ry;
#pragma unroll
    for (int t = 0; t < 10; t++) { rx.v[t] = s_r[t][slot]; ry.v[t] = s_r[10 + t][slot]; }
    const bool curve = cached_lane_match_point(acc, rx, ry, s_r[20][slot], strict);
    unsigned long long m = __ballot(ok && curve && active);
    m &= m >> 1;                         // both lanes of the pair
    m &= 0x5555555555555555ull;          // gather bit 2q -> bit q
    m = (m | (m >> 1)) & 0x3333333333333333ull;
    m = (m | (m >> 2)) & 0x0f0f0f0f0f0f0f0full;
    m = (m | (m >> 4)) & 0x00ff00ff00ff00ffull;
    m = (m | (m >> 8)) & 0x0000ffff0000ffffull;
    m = (m | (m >> 16)) & 0xffffffffull;
    if (lane == 0) {
        bitmap32[block * 2 + wave] = (u32)m;
        if (nreject) {
            const u32 first = block * 64 + wave * 32;
            const u32 sigs_here = first < n ? min(32u, n - first) : 0u;
            const u32 bad = sigs_here - (u32)__popcll(m);
            if (bad) atomicAdd(nreject, bad);
        }
    }
}
#ifndef HS_CACHED2R_WAVES
#define HS_CACHED2R_WAVES 3             // wavefronts per SIMD the kernel is compiled for: 3 = 165 VGPRs, no scratch (4 = at most 128: A/B below)
#endif
// a lone call (the QC): per-lane record loads - the staged form's extra barrier costs the latency path 1 % (0.1232 against
// 0.1220 ms at 6,667 signatures, host API 0.233 against 0.229 ms; profiles/r4_f_staged_ab.md)
__global__ void __launch_bounds__(192, HS_CACHED2R_WAVES)
k_verify_cached_lane2r(keyset_view ks, u32 nkeys, const u32 *__restrict__ b16, msg_src msgs,
                       const u32 *__restrict__ key_idx, const u32 *__restrict__ sigs, u32 n, u32 mode,
                       u32 *__restrict__ bitmap32, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    __shared__ u32 s_r[21][64];          // [limb of x (0-9), limb of y (10-19), flags (20)][signature in block]
    cached_lane2r_block<false>(s_r, nullptr, blockIdx.x, ks, nkeys, b16, msgs, key_idx, sigs, n, mode, bitmap32, nreject, skip);
}
// several independent small key-cache batches in one grid (hs_ed25519_verify_device_batches_cached): a 1,024-signature
// batch is only 16 workgroups of three wavefronts here.  Inputs staged: host groups of four / eight +4 - 4.5 %.
struct cached_batch { msg_src msgs; const u32 *key_idx, *sigs; u32 *bitmap32; u32 *nreject; u32 n, mode, first_block, pad; };
struct cached_batch_list { cached_batch b[HS_MAX_BATCHES_PER_LAUNCH]; u32 count; };
__global__ void __launch_bounds__(192)
k_verify_cached_lane2r_batches(keyset_view ks, u32 nkeys, const u32 *__restrict__ b16, cached_batch_list L) {
    __shared__ u32 s_r[21][64];
    __shared__ __attribute__((aligned(16))) u32 s_in[HS_CACHED_STAGE_WORDS];
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_BATCHES_PER_LAUNCH; t++) if (t < L.count && blockIdx.x >= L.b[t].first_block) k = t;
    const cached_batch &B = L.b[k];
    cached_lane2r_block<true>(s_r, s_in, blockIdx.x - B.first_block, ks, nkeys, b16, B.msgs, B.key_idx, B.sigs, B.n, B.mode, B.bitmap32, B.nreject, nullptr);
}

// ---- the "c" forms: chains first, then a batched finish by compression (verify_cached.h, cfin_*) -------------------------
// The latency kernel above spends a third of its instructions on the R wavefronts' square roots (3.4 M of 10.8 M VALU
// instructions at 6,667 signatures, profiles/r4_boxes/box_*_pmc_sq.json).  Where a call is not alone on the chip - batch
// lists, contexts marked concurrent, ring slots - or is large, latency is not what counts and that third can go: a
// workgroup is TWO chain wavefronts (64 signatures, two lanes each, inputs staged through LDS) that leave R' = (X : Y : Z)
// and the flags in the context's workspace, and a second, tiny launch finishes HS_CFIN_M signatures per lane with one
// inversion (finish lane L of a batch takes signatures L, L + nl, L + 2 nl, ...: consecutive lanes read consecutive
// signatures of every workspace row, and step j of a wavefront is one 64-bit word of the accept bitmap).
template <bool STAGED>
__device__ __forceinline__ void cached_chain2_block(u32 *s_in, u32 block, const keyset_view &ks, u32 nkeys, const u32 *__restrict__ b16,
                                                    msg_src msgs, const u32 *__restrict__ key_idx, const u32 *__restrict__ sigs, u32 n, u32 mode,
                                                    u32 *__restrict__ fin, u32 np, const u32 *__restrict__ skip) {
    if (skip && *skip) return;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool strict = (mode & HS_MODE_STRICT) != 0;
    const u32 i0 = block * 64;
    const u32 *s_sig = s_in, *s_msg = s_in + 64 * 16;
    if (STAGED) {
        const u32 cnt = min(64u, n - i0);
        const uint4 *gs = reinterpret_cast<const uint4 *>(sigs + (size_t)16 * i0);
        uint4 *ss = reinterpret_cast<uint4 *>(s_in);
        if (threadIdx.x < cnt * 4) ss[threadIdx.x] = gs[threadIdx.x];
        if (threadIdx.x + 128 < cnt * 4) ss[threadIdx.x + 128] = gs[threadIdx.x + 128];
        if (msgs.stride_words) {
            const uint4 *gm = reinterpret_cast<const uint4 *>(msgs.digests + (size_t)8 * i0);
            uint4 *sm = reinterpret_cast<uint4 *>(s_in + 64 * 16);
            if (threadIdx.x < cnt * 2) sm[threadIdx.x] = gm[threadIdx.x];
        }
        __syncthreads();
    }
    const u32 i = i0 + wave * 32 + (lane >> 1);
    const bool active = i < n;
    const u32 j = active ? i : n - 1;
    const u32 ki_raw = key_idx[j];
    const bool known = ki_raw < nkeys;
    const u32 ki = known ? ki_raw : 0;
    u32 kw[8], sw[8], msg[8];
    bool ok;
    if (STAGED) {
        if (msgs.stride_words) {
#pragma unroll
            for (int t = 0; t < 8; t++) msg[t] = s_msg[8 * (j - i0) + t];
        } else load_msg(msg, msgs, 0);
        ok = prepare_scalars(kw, sw, ks.pk + (size_t)8 * ki, s_sig + 16 * (j - i0), msg, true, true);
    } else {
        load_msg(msg, msgs, j);
        ok = prepare_scalars(kw, sw, ks.pk + (size_t)8 * ki, sigs + (size_t)16 * j, msg, true, true);
    }
    const u32 fl = ks.flag[ki];
    ok = ok && known && (fl & HS_F_ONCURVE);
    if (strict) ok = ok && !(fl & HS_F_SMALL);
    ge_p3 acc;
    cached_lane2_chain(acc, ks.tab + (size_t)ki * HS_KS_KEY_WORDS, b16, kw, sw);
    if (active && !(lane & 1u)) cfin_store(fin, np, i, acc, ok);
}
// one wavefront of finish lanes: lanes fblock * 64 ... of a batch whose finish grid is nl lanes wide
__device__ __forceinline__ void cached_finish_block(u32 fblock, u32 *__restrict__ fin, u32 np, u32 nl, u32 m, u32 n, const u32 *__restrict__ sigs, u32 mode,
                                                    unsigned long long *__restrict__ bitmap, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    const u32 words_per_step = nl / 64;
    if (skip && *skip) {                 // the combination already vouched for every signature (hs_rlc.hip)
        if (threadIdx.x == 0)
            for (u32 j = 0; j < m; j++) {
                const u32 word = j * words_per_step + fblock, first = word * 64;
                if (first < n) { const u32 lanes = min(64u, n - first); bitmap[word] = lanes == 64 ? ~0ull : ((1ull << lanes) - 1); }
            }
        return;
    }
    const u32 bits = cfin_lane(fin, np, fblock * 64 + threadIdx.x, nl, n, sigs, (mode & HS_MODE_STRICT) != 0, (int)m);
    u32 bad = 0;
#pragma nounroll
    for (u32 j = 0; j < m; j++) {
        const u32 word = j * words_per_step + fblock, first = word * 64;
        if (first < n) {                 // uniform over the wavefront
            const unsigned long long mask = __ballot((bits >> j) & 1u);
            if (threadIdx.x == 0) { bitmap[word] = mask; bad += min(64u, n - first) - (u32)__popcll(mask); }
        }
    }
    if (threadIdx.x == 0 && nreject && bad) atomicAdd(nreject, bad);
}
__global__ void __launch_bounds__(128, HS_CACHED2R_WAVES)
k_cached_chain2(keyset_view ks, u32 nkeys, const u32 *__restrict__ b16, msg_src msgs, const u32 *__restrict__ key_idx,
                const u32 *__restrict__ sigs, u32 n, u32 mode, u32 *__restrict__ fin, u32 np, const u32 *__restrict__ skip) {
    __shared__ __attribute__((aligned(16))) u32 s_in[HS_CACHED_STAGE_WORDS];
    cached_chain2_block<true>(s_in, blockIdx.x, ks, nkeys, b16, msgs, key_idx, sigs, n, mode, fin, np, skip);
}
__global__ void __launch_bounds__(64)
k_cached_finish(u32 *__restrict__ fin, u32 np, u32 nl, u32 m, u32 n, const u32 *__restrict__ sigs, u32 mode,
                unsigned long long *__restrict__ bitmap, u32 *__restrict__ nreject, const u32 *__restrict__ skip) {
    cached_finish_block(blockIdx.x, fin, np, nl, m, n, sigs, mode, bitmap, nreject, skip);
}
// the same pair over a list of batches
struct cachedc_batch { msg_src msgs; const u32 *key_idx, *sigs; unsigned long long *bitmap; u32 *nreject; u32 *fin; u32 n, mode, np, nl, first_block, first_f; };
struct cachedc_batch_list { cachedc_batch b[HS_MAX_BATCHES_PER_LAUNCH]; u32 count, m; };
__global__ void __launch_bounds__(128, HS_CACHED2R_WAVES)
k_cached_chain2_batches(keyset_view ks, u32 nkeys, const u32 *__restrict__ b16, cachedc_batch_list L) {
    __shared__ __attribute__((aligned(16))) u32 s_in[HS_CACHED_STAGE_WORDS];
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_BATCHES_PER_LAUNCH; t++) if (t < L.count && blockIdx.x >= L.b[t].first_block) k = t;
    const cachedc_batch &B = L.b[k];
    cached_chain2_block<true>(s_in, blockIdx.x - B.first_block, ks, nkeys, b16, B.msgs, B.key_idx, B.sigs, B.n, B.mode, B.fin, B.np, nullptr);
}
__global__ void __launch_bounds__(64)
k_cached_finish_batches(cachedc_batch_list L) {
    u32 k = 0;
#pragma unroll
    for (u32 t = 1; t < HS_MAX_BATCHES_PER_LAUNCH; t++) if (t < L.count && blockIdx.x >= L.b[t].first_f) k = t;
    const cachedc_batch &B = L.b[k];
    cached_finish_block(blockIdx.x - B.first_f, B.fin, B.np, B.nl, L.m, B.n, B.sigs, B.mode, B.bitmap, B.nreject, nullptr);
}

// ------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------
static thread_local char t_err[256] = "";
int hs::hs_fail(int code, const char *what, hipError_t e) {
    if (e != hipSuccess) snprintf(t_err, sizeof t_err, "%s: %s", what, hipGetErrorString(e));
    else snprintf(t_err, sizeof t_err, "%s", what);
    return code;
}
#define fail hs::hs_fail

std::mutex hs::g_mu;
std::condition_variable hs::g_cv;
std::atomic<bool> hs::g_init{false};
std::vector<dev_state> hs::g_devs;           // written under g_mu by init / shutdown only
std::atomic<size_t> hs::g_max_batch{0};
std::atomic<int> hs::g_oct_pick[hs::kMaxSlots];      // set to -1 by init
// Pool sizes per device slot.  Every context owns a stream and HIP multiplexes streams onto 16 hardware queues
// (more streams in flight than queues collapse, profiles/r1_stream_sweep.txt), so the pools stay small: sixteen
// narrow contexts - two per CU partition - for the small calls that come in bulk, four wide ones for the big calls.
int hs::kMaxPool[2] = {4, 16};                 // HS_ED25519_POOL_NARROW overrides the second (read at init; measurement aid)
// Round 6: two contexts per narrow stream.  A group's launch (2 - 3 ms at sixteen batches) used to be the only work on its stream:
// from its end until the host had woken, collected sixteen verdicts, staged the next 2 MB and launched, the stream's share of its
// CU slice stood idle.  With a second context (own pinned staging, workspace and event) queued on the same stream the next
// launch is already there.  Streams - hardware queues - stay at sixteen.  HS_ED25519_POOL_DEPTH overrides (measurement aid).
int hs::kNarrowDepth = 2;
hs::stream_box::~stream_box() { if (s) { (void)hipSetDevice(device); (void)hipStreamDestroy(s); } }
// key cache, single calls: where the two-launch form - chains, then the batched finish by compression - takes over.  Its
// second launch is a tail of 0.07 - 0.11 ms (four to sixteen signatures per lane and one 254-squaring inversion on a handful of
// lone wavefronts), so it pays only where the chain phase is much longer than that (profiles/r4_j_cached_c_ab.md, same box):
// four calls in flight 341 -> 414 M signatures/s at 49,152 - 66,667, 353 / 382 -> 436 - 441 at 131,072 (343 against 346 - 353
// at 32,768: not yet); a LONE call with four signatures per lane 0.369 ms at 131,072 against the R-wavefront kernel's 0.44
// (0.290 against 0.262 at 66,667: not yet) - above 98,304 (a one-lane kernel, k_verify_cached_lane, used to take over there: 0.40 ms at
// 131,072, ahead by 7 % on some GPUs and behind by 8 % on others, profiles/r4_t_cached_c_slow_box.md; retired in round 5).  Batch lists of small batches lose outright (185 against 269 M verifies/s at four 1,024-signature batches per launch:
// the tail is as long as the chains) and keep the R-wavefront kernel.  HS_ED25519_CACHED_C_MIN / _CACHED_C_MIN_LONE override.
// A lone call takes the two-launch form above 98,304 signatures on every GPU (round 5: round 4 switched there by an init-time proxy
// for "the box's class", worth +8 % / -7 % in a range past every BASELINE config; a size switch that needs a per-box oracle and
// buys < 10 % is not worth its branch - VERDICT r4 item 4).
static std::atomic<size_t> g_cached_c_min{49152}, g_cached_c_min_lone{98305};
static size_t cached_c_min_lone() { return g_cached_c_min_lone.load(); }
// hs_ed25519_verify_device_batches picks its kernel by what the caller hands over at once (measured, DESIGN.md
// section 5): up to three batches the one-signature-per-quad form (latency), from four the pairs form, from eight
// batches that total 8,192 signatures the one-lane pair over the batch list (k_prepare_batches + k_verify_batches: the
// least work per verification once the chip is full, at 2-3x the latency of a launch).
// HS_ED25519_PAIRS_MIN_BATCHES / HS_ED25519_LANE_MIN_BATCHES override the counts for A/B runs (0 = never).
static std::atomic<size_t> g_pairs_min_batches{4}, g_lane_min_batches{8};
static const size_t kLaneMinTotal = 8192;
// fused quad kernel up to here (latency), lane kernels above (throughput).  32,768 is also the API's bound on a list entry.
// With the half-size form the quad kernels hold TWO per-signature tables in LDS (40,960 B per wavefront: one wavefront per
// SIMD), so a single batch fills the chip's 1,024 SIMDs at 16,384 signatures; HS_ED25519_QUAD_MAX_BATCH overrides (A/B runs).
static std::atomic<size_t> g_quad_max_batch{32768};
static std::atomic<size_t> g_oct_max_batch{8192};      // HS_ED25519_OCT_MAX_BATCH: 0 switches the eight-lane latency kernel off (A/B runs)
static std::atomic<int> g_oct_split{-1};               // HS_ED25519_OCT_SPLIT=0 / 1: the one-wavefront eight-lane form (k_verify_fused8) / the split one at every size; -1 = by size and measurement
static std::atomic<int> g_autotune{1};                 // HS_ED25519_AUTOTUNE=0: no measurement at init (the split form throughout)
// The split form (main + helper wavefront) up to half a wavefront slot per SIMD - every main and helper alone on a SIMD: it wins
// on every box measured (0.276 -> 0.238 ms at 64 signatures, 0.283 -> 0.268 at 4,096; 0.302 -> 0.259, 0.305 -> 0.278 on others).
// Above, mains share SIMDs with other workgroups' helpers for their first ~50 us, and which form wins depends on the GPU - by
// 3 - 10 % in either direction on boxes of one SKU, firmware and clock (profiles/r4_c_oct_split_ab.md, r4_m_oct_split_ab_fast_box.md).
// Round 4 inferred it from an instruction-feed probe; the driver's box then ran the inferred form 14 % slower than the probe's
// class promised.  Round 5 measures the thing itself: hs_ed25519_autotune times both forms on a canned
// 6,667-signature batch at init and `pick` is the faster one on THIS device (-1 = not measured: the split form).
// `whole_chip`: the context's stream may use every CU.  On a stream confined to a slice of the chip (a narrow pool context's
// 32 CUs, a tenant's share) the measurement does not transfer: above half a slot per SIMD the split form there takes 0.365 - 0.370 ms
// against the one-wavefront form's 0.255 - 0.259 (1,024 and 667 signatures on one 32-CU slice, a GPU whose whole-chip pick is a 4 %
// matter; below half a slot the split form wins on a slice as on the chip, 0.215 against 0.255: profiles/r5_y_narrow_oct_ab.md) - which
// is what a lone synchronous 1,024-signature host call paid on every GPU whose pick was the split form (0.385 against 0.271 ms,
// `end_to_end_call_p50_ms_one_sync_caller_1024`).  A confined context takes the split form up to half a slot and never above.
static bool oct_split_for(size_t n, size_t simds, int pick, bool whole_chip = true) {
    const int forced = g_oct_split.load();
    if (forced >= 0) return forced != 0;
    if (n * 2 <= simds * 8) return true;
    return whole_chip && pick != 0;
}
// the largest sliver past whole rounds of one-lane wavefronts that goes through the four-lane form inside the stage-2 launch
// (enqueue_cold): -19 % on the call's p50 up to 16,384 signatures past the round (1.03 against 1.27 ms at 66,667 ... 73,728,
// 1.07 against 1.31 at 81,920), nothing at 32,768 (profiles/r3_concurrent_ab.txt)
static const size_t kTailMax = 16384;
static std::atomic<size_t> g_tail_max{kTailMax};      // HS_ED25519_TAIL_MAX: sweeps
// Host-pointer calls of up to this many signatures (64 wavefronts of the fused kernel) are served by the NARROW
// pool, whose streams are confined to one slice of the CUs each: concurrent callers then occupy their own slices
// of the chip instead of being stacked onto the same CUs by the dispatcher (bench.py `end_to_end`); one caller
// alone loses nothing, its batch is at most one wavefront per SIMD either way.
// The narrow pool's streams are confined to slices of the CU mask (until round 5: 8 slices of 32 CUs, two contexts on each; a slice is
// the same share of EVERY XCD - the mask's bits are dealt round robin over the XCDs, profiles/r6_u_cu_mask_probe.md - not one XCD).
// Two launches of two 1,024-signature batches then share a slice at two wavefronts per SIMD: 49-50 M verifies/s from
// host buffers through hs_ed25519_submit bursts (16 slices of 16 CUs with one context each: 41.7 M - the two
// wavefronts of a SIMD then come from ONE launch and run 18 % longer; profiles/r2_e2e_async.txt).  The synchronous
// calls pay for it at 16 concurrent callers (21.5 against 26.8 M verifies/s; equal up to 8 callers).
// Round 6: FOUR slices of 64 CUs, four streams each - the shape of the device-resident headline.  With the pool created in turn
// (pool_get) the same GPU, same run (profiles/r6_k_e2e_pool_shape.md): bursts of sixteen 99 - 103 M verifies/s from host buffers
// on four slices against 74 - 83 M on eight; bursts of one to four equal (36 / 60 / 76 M); a lone 1,024-signature call 0.257 ms
// against 0.270 (half a wavefront slot per SIMD of its slice: the split eight-lane form); only sixteen threads with two bursts of
// EIGHT each in flight prefer eight slices (92 - 103 against 86 - 90 M: a launch of one wavefront per SIMD of its slice is the shape
// that fills it).
unsigned hs::kPoolPartitions = 4;               // HS_ED25519_POOL_PARTITIONS overrides (read at init; measurement aid)
// Several node processes on one GPU (the reference's `fab local` runs four node processes on one host,
// benchmark/benchmark/local.py:94-104).  Every process brings its own pools - 16 narrow + 4 wide stream contexts - and
// the device multiplexes all of them; past ~16 streams in flight throughput collapses (profiles/r1_stream_sweep.txt).
// A process that knows it is tenant `index` of `tenants` on its GPU takes 1/tenants of the stream budget and confines its
// narrow pool to its own 1/tenants of the CUs (of every XCD: see above), so co-tenants neither share hardware
// queues nor stack their small batches on the same CUs.  hs_ed25519_set_tenancy before init, or the environment
// (HS_ED25519_TENANTS / HS_ED25519_TENANT_INDEX); default 1 / 0 = the whole chip.
unsigned hs::kTenants = 1, hs::kTenantIndex = 0;
bool hs::kTenantWide = true;
extern "C" int hs_ed25519_set_tenancy(unsigned tenants, unsigned index) {
    if (tenants == 0 || tenants > 8 || index >= tenants) return hs::hs_fail(HS_ED25519_E_ARG, "set_tenancy: tenants in [1, 8], index < tenants");
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_init.load()) return hs::hs_fail(HS_ED25519_E_ARG, "set_tenancy: call it before hs_ed25519_init (or after hs_ed25519_shutdown)");
    kTenants = tenants; kTenantIndex = index;
    return 0;
}

int hs::dev_slots() { return g_init.load() ? (int)g_devs.size() : 0; }
int hs::dev_ordinal(int slot) { return (slot >= 0 && slot < hs::dev_slots()) ? g_devs[slot].ordinal : -1; }
size_t hs::max_batch_hint() { return g_max_batch.load(); }
bool hs::engine_ready() { return g_init.load(); }

static void ctx_release(hs_ed25519_ctx *c) {
    if (c->d_ws) (void)hipFree(c->d_ws);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    c->d_ws = nullptr; c->d_in = c->d_out = c->h_in = c->h_out = nullptr; c->cap = 0;
}
int hs::ctx_reserve(hs_ed25519_ctx *c, size_t n, bool staging) {
    if (n <= c->cap && (!staging || c->staging)) return 0;
    size_t cap = round_up(n < 64 ? 64 : n, 64);
    if (cap < c->cap) cap = c->cap;
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    if (c->stream && c->cap) HS_HIP(hipStreamSynchronize(c->stream), "stream sync");   // the old buffers may be in use
    ctx_release(c);
    HS_HIP(hipMalloc((void **)&c->d_ws, workspace_words(cap) * 4), "hipMalloc workspace");
    if (staging || c->staging) {
        const size_t in_bytes = cap * 128 + 64, out_bytes = bitmap_bytes(cap) + 16;
        HS_HIP(hipMalloc((void **)&c->d_in, in_bytes), "hipMalloc inputs");
        HS_HIP(hipMalloc((void **)&c->d_out, out_bytes), "hipMalloc outputs");
        // Portable: the pinned staging is mapped on every device of the process (a sharded call reads it from several)
        HS_HIP(hipHostMalloc((void **)&c->h_in, in_bytes, hipHostMallocPortable | hipHostMallocMapped), "hipHostMalloc inputs");
        HS_HIP(hipHostMalloc((void **)&c->h_out, out_bytes, hipHostMallocPortable | hipHostMallocMapped), "hipHostMalloc outputs");
        c->staging = true;
    }
    c->cap = cap;
    return 0;
}

int hs::ctx_new(hs_ed25519_ctx **out, size_t max_batch, int slot, const uint32_t *cu_mask, uint32_t mask_words,
                std::shared_ptr<stream_box> share) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (slot < 0 || slot >= (int)g_devs.size()) return fail(HS_ED25519_E_ARG, "ctx_create: no such device slot");
    const int device = g_devs[slot].ordinal;
    HS_HIP(hipSetDevice(device), "hipSetDevice");
    hs_ed25519_ctx *c = new (std::nothrow) hs_ed25519_ctx();
    if (!c) return fail(HS_ED25519_E_NOMEM, "ctx_create: out of memory");
    c->slot = slot; c->device = device;
    c->cus = g_devs[slot].cus;
    if (cu_mask) { unsigned pc = 0; for (uint32_t w = 0; w < mask_words; w++) pc += (unsigned)__builtin_popcount(cu_mask[w]); if (pc) c->cus = pc; }
    if (share) { c->stream = share->s; c->stream_share = share; }      // a context of the narrow pool: it queues on one of the pool's streams
    else {
        hipError_t e = cu_mask ? hipExtStreamCreateWithCUMask(&c->stream, mask_words, cu_mask)
                               : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(HS_ED25519_E_DEVICE, cu_mask ? "hipExtStreamCreateWithCUMask" : "hipStreamCreate", e); }
    }
    int rc = ctx_reserve(c, max_batch ? max_batch : 64, false);
    if (rc) { if (!c->stream_share) (void)hipStreamDestroy(c->stream); ctx_release(c); delete c; return rc; }
    *out = c;
    return 0;
}
extern "C" int hs_ed25519_ctx_create_on(hs_ed25519_ctx **out, size_t max_batch, int slot) {
    if (!out) return fail(HS_ED25519_E_ARG, "ctx_create: null out");
    return ctx_new(out, max_batch, slot, nullptr, 0);
}
extern "C" int hs_ed25519_ctx_create(hs_ed25519_ctx **out, size_t max_batch) { return hs_ed25519_ctx_create_on(out, max_batch, 0); }
// A context whose own stream may only run on a contiguous slice of the chip's CUs
// (hipExtStreamCreateWithCUMask).  Concurrent small batches then occupy disjoint CUs instead
// of being stacked on the same ones by the workgroup dispatcher.
extern "C" int hs_ed25519_ctx_create_partition(hs_ed25519_ctx **out, size_t max_batch, unsigned part, unsigned nparts) {
    if (!out || nparts == 0 || part >= nparts) return fail(HS_ED25519_E_ARG, "ctx_create_partition: bad arguments");
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    hipDeviceProp_t prop;
    HS_HIP(hipGetDeviceProperties(&prop, g_devs[0].ordinal), "hipGetDeviceProperties");
    const unsigned cus = (unsigned)prop.multiProcessorCount;
    if (nparts > cus) return fail(HS_ED25519_E_ARG, "ctx_create_partition: more partitions than CUs");
    std::vector<uint32_t> mask((cus + 31) / 32, 0u);
    // contiguous ranges of mask bits: measured best (16 x 16 CUs: 29.3 M verifies/s; CU c -> partition c mod 16: 19.4 M;
    // 20 / 24 / 28 slices: 8-12 M) - profiles/r1_stream_sweep.txt.  Round 6 found why (profiles/r6_u_cu_mask_probe.md): the
    // mask's bits are dealt round robin over the eight XCDs, so a contiguous range is the same share of every XCD, a range
    // whose length is no multiple of eight is lopsided, and a strided mask that leaves an XCD empty is not applied at all.
    const unsigned lo = (unsigned)((uint64_t)cus * part / nparts), hi = (unsigned)((uint64_t)cus * (part + 1) / nparts);
    for (unsigned c = lo; c < hi; c++) mask[c / 32] |= 1u << (c % 32);
    return ctx_new(out, max_batch, 0, mask.data(), (uint32_t)mask.size());
}
extern "C" void *hs_ed25519_ctx_stream(hs_ed25519_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int hs_ed25519_ctx_device(const hs_ed25519_ctx *c) { return c ? c->device : -1; }

extern "C" void hs_ed25519_ctx_destroy(hs_ed25519_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); if (!c->stream_share) (void)hipStreamDestroy(c->stream); }
    c->stream_share.reset();          // a shared stream goes with the last context (or pool entry) that names it
    if (c->done) (void)hipEventDestroy(c->done);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->d_tc) (void)hipFree(c->d_tc);
    hs::rlc_release(c);
    hs::wire_release(c);
    ctx_release(c);
    delete c;
}
// Per-context kernel choice (tests and A/B measurements; the default is by batch size):
// HS_K_AUTO 0, HS_K_LANE 1 (cold: k_prepare + k_verify; key-cache calls: as HS_K_FUSED),
// HS_K_FUSED 3 (cold: k_verify_fused; cached: k_verify_cached_lane2r at any size).  Returns the previous choice.
extern "C" int hs_ed25519_ctx_set_concurrent(hs_ed25519_ctx *c, int concurrent) {
    if (!c) return fail(HS_ED25519_E_ARG, "ctx_set_concurrent: null context");
    const int was = c->several_in_flight ? 1 : 0;
    c->several_in_flight = concurrent != 0;
    return was;
}
extern "C" int hs_ed25519_ctx_set_kernel(hs_ed25519_ctx *c, int which) {
    if (!c) return fail(HS_ED25519_E_ARG, "ctx_set_kernel: null context");
    const int old = c->kernel;
    if (which == HS_K_AUTO || which == HS_K_LANE || which == HS_K_FUSED || which == HS_K_FUSED_PAIRS || which == HS_K_CACHED_C ||
        which == HS_K_OCT || which == HS_K_OCT_SPLIT) c->kernel = which;
    else return fail(HS_ED25519_E_ARG, "ctx_set_kernel: unknown kernel");
    return old;
}

// The base point's tables on one device (called by hs_ed25519_init_devices with that device current): the 8-bit
// key-cache table through the key-set registration kernels, then the 16-bit table from it - one addition per entry into
// a projective scratch (84 MB, freed at once) and the shared-inversion normalisation the key tables get.
static int build_tables(u32 *d_tab, const u32 *d_axy, size_t nkeys, bool negate, hipStream_t st);
static void free_base_tables(dev_state &d) {
    if (d.base_tab || d.base_tab16) (void)hipSetDevice(d.ordinal);
    if (d.base_tab) { (void)hipFree(d.base_tab); d.base_tab = nullptr; }
    if (d.base_tab16) { (void)hipFree(d.base_tab16); d.base_tab16 = nullptr; }
}
static int build_base_tables(dev_state &d) {
    ge_p3 b; ge_basepoint(b);
    u32 h_axy[20];
    for (int j = 0; j < 10; j++) { h_axy[j] = b.X.v[j]; h_axy[10 + j] = b.Y.v[j]; }
    u32 *d_axy = nullptr, *d_t16 = nullptr, *d_proj = nullptr;
    const size_t entries = (size_t)HS_B16_WINDOWS * HS_B16_ENTRIES;
    hipError_t e = hipMalloc((void **)&d_axy, sizeof h_axy);
    if (e == hipSuccess) e = hipMalloc((void **)&d.base_tab, (size_t)HS_KS_KEY_WORDS * 4);
    if (e == hipSuccess) e = hipMemcpy(d_axy, h_axy, sizeof h_axy, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? build_tables(d.base_tab, d_axy, 1, false, nullptr) : 0;
    if (e == hipSuccess && !rc) e = hipMalloc((void **)&d_t16, HS_B16_WORDS * 4);
    if (e == hipSuccess && !rc) e = hipMalloc((void **)&d_proj, entries * HS_KS_PROJ_WORDS * 4);
    if (e == hipSuccess && !rc) {
        const u32 *p_t8 = d.base_tab;
        void *a1[] = { &p_t8, &d_proj };
        e = hipLaunchKernel((const void *)k_base16_build, dim3((unsigned)((entries + 63) / 64)), dim3(64), a1, 0, nullptr);
    }
    if (e == hipSuccess && !rc) {
        const u32 *p_proj = d_proj;
        unsigned long long groups = entries / HS_KS_AFFINE_GROUP;
        void *a2[] = { &p_proj, &d_t16, &groups };
        e = hipLaunchKernel((const void *)k_keyset_affine, dim3((unsigned)((groups + 63) / 64)), dim3(64), a2, 0, nullptr);
    }
    if (e == hipSuccess && !rc) e = hipDeviceSynchronize();
    if (d_proj) (void)hipFree(d_proj);
    if (d_axy) (void)hipFree(d_axy);
    d.base_tab16 = d_t16;
    if (e != hipSuccess && !rc) rc = fail(HS_ED25519_E_DEVICE, "base-point table build", e);
    if (rc) free_base_tables(d);
    return rc;
}
// Pools of a process that is one of `tenants` on its GPU, and the hardware queues they ask for.  The budget that matters is
// the GPU's, not the process's: the driver's hardware scheduler runs ~28 user queues at once ACROSS ALL PROCESSES, and one
// queue more oversubscribes its runlist - every process is then time-sliced (four tenants with seven queues each: 75 M
// verifies/s alone, 66 M and a 1.6x spread with ONE more queue held by an idle fifth process, 0.1 - 33 M and up to 160x with
// 8 - 32 more: tools/queue_probe.py, profiles/r4_c_queue_probe_pools_of_7.md).  So co-tenants take few queues and leave
// headroom for whatever else opens the GPU (a monitoring agent, a bench harness).  A tenant's pools are the single process's
// (16 narrow streams, 4 wide) divided by the tenants: for four, 4 narrow streams on two 32-CU slices of the tenant's 64 CUs
// - two launches per slice is what keeps two wavefronts on every SIMD for bursts of four AND of sixteen (round 6, four streams
// on ONE 64-CU slice, the single process's shape: 73.2 M at bursts of four against 77.5, 85.5 against 84.0 M at bursts of
// sixteen, profiles/r6_l_mp_sweep.txt; round 4, three contexts on one slice: 54.6 M at bursts of four) - one wide context, one
// queue for the process's own default stream: 6 queues per tenant, 24 for four, 22 for two.
static unsigned tenant_partitions(unsigned tenants) { return tenants <= 1 ? 4u : (tenants <= 4 ? 2u : 1u); }      // 4 x 64 CUs; two tenants 2 x 64 each; four 2 x 32 each; eight 1 x 32
static int pool_narrow_for(unsigned tenants) { return tenants <= 1 ? 16 : 2 * (int)(8 / tenants < 1 ? 1u : 8 / tenants); }      // streams; x kNarrowDepth contexts
static int pool_wide_for(unsigned tenants) { return tenants <= 1 ? 4 : (4 / tenants < 1 ? 1 : 4 / (int)tenants); }
extern "C" unsigned hs_ed25519_hw_queue_budget(unsigned tenants) {
    if (tenants < 1) tenants = 1;
    if (tenants > 8) tenants = 8;
    if (tenants == 1) return 16;
    return (unsigned)(pool_narrow_for(tenants) + pool_wide_for(tenants) + 1);
}
// the environment knobs init found set (name, value), for hs_ed25519_config_dump; written under g_mu by init only
static std::vector<std::pair<std::string, std::string>> g_env_seen;
static int g_hw_queues_env = 0;             // GPU_MAX_HW_QUEUES as init found it (0 = unset: the HIP runtime's default of 4)
static const char *env_knob(const char *name) {
    const char *e = getenv(name);
    if (e) g_env_seen.emplace_back(name, e);
    return e;
}
extern "C" uint32_t hs_ed25519_abi_version(void) { return HS_ED25519_ABI_VERSION; }
extern "C" int hs_ed25519_init_devices(const int *ordinals, size_t count, size_t max_batch) {
    if (!ordinals || count == 0 || count > (size_t)kMaxSlots) return fail(HS_ED25519_E_ARG, "hs_ed25519_init_devices: bad device list");
    std::unique_lock<std::mutex> lk(g_mu);
    if (g_init.load()) {
        // idempotent for the same device list; a different one would silently run on the wrong GPUs
        bool same = g_devs.size() == count;
        for (size_t i = 0; same && i < count; i++) same = g_devs[i].ordinal == ordinals[i];
        if (!same) return fail(HS_ED25519_E_ARG, "hs_ed25519_init: already initialised with a different device list "
                                                   "(hs_ed25519_shutdown first)");
        if (max_batch > g_max_batch.load()) g_max_batch.store(max_batch);
        return 0;
    }
    // Every knob is read here, once (hs_ed25519_config_dump reports what was resolved; INTEGRATION.md has the table).
    g_env_seen.clear();
    if (const char *e = env_knob("HS_ED25519_TENANTS")) { const int v = atoi(e); if (v >= 1 && v <= 8) { kTenants = (unsigned)v; if (kTenantIndex >= kTenants) kTenantIndex = 0; } }
    if (const char *e = env_knob("HS_ED25519_TENANT_INDEX")) { const int v = atoi(e); if (v >= 0 && (unsigned)v < kTenants) kTenantIndex = (unsigned)v; }
    kMaxPool[0] = 4; kMaxPool[1] = 16; kPoolPartitions = 4; kNarrowDepth = 2;
    if (kTenants > 1) {
        kMaxPool[1] = pool_narrow_for(kTenants);
        kMaxPool[0] = pool_wide_for(kTenants);
        kPoolPartitions = tenant_partitions(kTenants);
    }
    if (const char *e = env_knob("HS_ED25519_TENANT_WIDE")) kTenantWide = e[0] != '0';
    // The pools, the ring slots and the CU-partitioned contexts each want a hardware queue of their own; HIP's default of 4
    // multiplexes them (streamed rounds: 22 vs 34 M signatures/s).  The runtime reads GPU_MAX_HW_QUEUES when IT initialises,
    // so the variable belongs to the host's main(), before any thread exists: glibc's setenv is not
    // safe against a concurrent getenv, and the reference's node is a multi-thread tokio runtime (node/src/main.rs:57) - a
    // library must not touch the environment from inside it.  Only HS_ED25519_SET_ENV=1 - a host that
    // says it is single-threaded here and has not started HIP yet - still gets the old setenv (without overwrite).
    // Otherwise: compare, and leave a warning for hs_ed25519_last_error when the budget is not there.
    const unsigned budget = hs_ed25519_hw_queue_budget(kTenants);
    {
        const char *set_env = env_knob("HS_ED25519_SET_ENV");
        if (set_env && set_env[0] == '1') {
            char q[8];
            snprintf(q, sizeof q, "%u", budget);
            setenv("GPU_MAX_HW_QUEUES", q, 0);
        }
        const char *have_q = getenv("GPU_MAX_HW_QUEUES");
        g_hw_queues_env = have_q ? atoi(have_q) : 0;
    }
    g_pairs_min_batches.store(4); g_lane_min_batches.store(8); g_oct_split.store(-1); g_autotune.store(1);
    g_cached_c_min.store(49152); g_cached_c_min_lone.store(98305); g_oct_max_batch.store(8192); g_tail_max.store(kTailMax); g_quad_max_batch.store(32768);
    if (const char *e = env_knob("HS_ED25519_PAIRS_MIN_BATCHES")) { const long v = atol(e); g_pairs_min_batches.store(v > 0 ? (size_t)v : (size_t)-1); }
    if (const char *e = env_knob("HS_ED25519_LANE_MIN_BATCHES")) { const long v = atol(e); g_lane_min_batches.store(v > 0 ? (size_t)v : (size_t)-1); }
    if (const char *e = env_knob("HS_ED25519_OCT_SPLIT")) g_oct_split.store(e[0] != '0' ? 1 : 0);
    if (const char *e = env_knob("HS_ED25519_AUTOTUNE")) g_autotune.store(e[0] != '0' ? 1 : 0);
    if (const char *e = env_knob("HS_ED25519_CACHED_C_MIN")) { const long long v = atoll(e); if (v >= 0) g_cached_c_min.store((size_t)v); }
    if (const char *e = env_knob("HS_ED25519_CACHED_C_MIN_LONE")) { const long long v = atoll(e); if (v > 0) g_cached_c_min_lone.store((size_t)v); }
    if (const char *e = env_knob("HS_ED25519_OCT_MAX_BATCH")) { const long v = atol(e); if (v >= 0 && v <= 32768) g_oct_max_batch.store((size_t)v); }
    if (const char *e = env_knob("HS_ED25519_TAIL_MAX")) { const long v = atol(e); if (v >= 0 && v <= 65536) g_tail_max.store((size_t)v); }
    if (const char *e = env_knob("HS_ED25519_QUAD_MAX_BATCH")) { const long v = atol(e); if (v >= 64 && v <= 32768) g_quad_max_batch.store((size_t)v); }
    if (const char *e = env_knob("HS_ED25519_POOL_NARROW")) { const int v = atoi(e); if (v >= 1 && v <= 64) kMaxPool[1] = v; }
    if (const char *e = env_knob("HS_ED25519_POOL_PARTITIONS")) { const int v = atoi(e); if (v >= 1 && v <= 64) kPoolPartitions = (unsigned)v; }
    if (const char *e = env_knob("HS_ED25519_POOL_DEPTH")) { const int v = atoi(e); if (v >= 1 && v <= 4) kNarrowDepth = v; }
    (void)env_knob("HS_ED25519_MIN_GPU_BATCH"); (void)env_knob("HS_ED25519_SHARD_MSM_MIN");      // read where they are used; listed in the dump
    int have = 0;
    HS_HIP(hipGetDeviceCount(&have), "hipGetDeviceCount");
    for (size_t i = 0; i < count; i++)
        if (ordinals[i] < 0 || ordinals[i] >= have) return fail(HS_ED25519_E_ARG, "hs_ed25519_init: no such device");
    std::vector<dev_state> devs(count);
    for (size_t i = 0; i < count; i++) {
        devs[i].ordinal = ordinals[i];
        // a failure on any device gives back what the earlier ones had built (67 MB each): init either succeeds as a whole or leaves nothing
        hipError_t e = hipSetDevice(ordinals[i]);
        const char *what = "hipSetDevice";
        int rc = 0;
        // the base point's window tables (8-bit: key-cache layout, also the RLC path's; 16-bit: every verify kernel's [s]B)
        if (e == hipSuccess) rc = build_base_tables(devs[i]);
        if (e == hipSuccess && !rc) {
            const u32 *b16 = devs[i].base_tab16;
            what = "set the base-point table pointer";
            e = hipMemcpyToSymbol(HIP_SYMBOL(g_b16), &b16, sizeof b16);
            if (e == hipSuccess) { what = "hipDeviceSynchronize"; e = hipDeviceSynchronize(); }
            hipDeviceProp_t prop;
            if (e == hipSuccess) { what = "hipGetDeviceProperties"; e = hipGetDeviceProperties(&prop, ordinals[i]); }
            if (e == hipSuccess) devs[i].cus = (unsigned)prop.multiProcessorCount;
        }
        if (e != hipSuccess && !rc) rc = fail(HS_ED25519_E_DEVICE, what, e);
        if (rc) { for (size_t k = 0; k <= i; k++) free_base_tables(devs[k]); return rc; }
    }
    if (const hipError_t e = hipSetDevice(ordinals[0]); e != hipSuccess) {
        for (dev_state &d : devs) free_base_tables(d);
        return fail(HS_ED25519_E_DEVICE, "hipSetDevice", e);
    }
    for (int i = 0; i < kMaxSlots; i++) g_oct_pick[i].store(-1, std::memory_order_relaxed);
    g_devs.swap(devs);
    g_max_batch.store(max_batch ? max_batch : 1024);
    g_init.store(true);
    const bool tune = g_autotune.load() && g_oct_split.load() < 0 && kTenants == 1;      // a co-tenant does not tune (hs_tune.hip)
    const int queues = g_hw_queues_env;
    lk.unlock();
    // measured kernel choice (hs_tune.hip), outside the lock: it runs ordinary calls on a context of its own.  A failure
    // costs the measurement, not the engine: the slot keeps the default form and the text says why.
    char warn[256] = "";
    if (tune)
        for (size_t i = 0; i < count; i++)
            if (hs::autotune_slot((int)i, nullptr) < 0)
                snprintf(warn, sizeof warn, "hs_ed25519_init: warning: autotune on device slot %zu failed (%.120s); the split eight-lane form stays the default", i, t_err);
    if (!warn[0] && (unsigned)queues < budget)
        snprintf(warn, sizeof warn, "hs_ed25519_init: warning: GPU_MAX_HW_QUEUES is %s%.0d but the pools want %u hardware queues "
                 "(hs_ed25519_hw_queue_budget): export it in main() before the HIP runtime starts, or streams will share queues",
                 queues ? "" : "unset (HIP's default: 4)", queues, budget);
    snprintf(t_err, sizeof t_err, "%s", warn);
    (void)hipSetDevice(ordinals[0]);
    return 0;
}
extern "C" int hs_ed25519_init(int device, size_t max_batch) { return hs_ed25519_init_devices(&device, 1, max_batch); }
extern "C" int hs_ed25519_device_count(void) { return hs::dev_slots(); }
extern "C" void hs_ed25519_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_init.load()) return;
    g_init.store(false);
    for (dev_state &d : g_devs) {
        for (int k = 0; k < 2; k++) {
            for (hs_ed25519_ctx *c : d.free_ctx[k]) hs_ed25519_ctx_destroy(c);
            d.free_ctx[k].clear(); d.live[k] = 0;
        }
        delete d.lanes; d.lanes = nullptr;
        free_base_tables(d);
        // tickets still outstanding keep their group alive; its context is destroyed when they are waited on
    }
    g_devs.clear();
}
extern "C" const char *hs_ed25519_last_error(void) { return t_err; }

// what init resolved, as JSON (include/hs_ed25519.h); no device call
extern "C" size_t hs_ed25519_config_dump(char *buf, size_t cap) {
    std::string o;
    char t[2048];      // the longest chunk below is ~620 characters with every value at its 20-digit widest (512 could truncate into invalid JSON)
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t pm = g_pairs_min_batches.load(), lm = g_lane_min_batches.load();
    snprintf(t, sizeof t, "{\"abi_version\": %u, \"initialised\": %s, \"tenants\": %u, \"tenant_index\": %u, \"tenant_wide\": %s, "
             "\"pool_wide\": %d, \"pool_narrow\": %d, \"pool_narrow_depth\": %d, \"pool_partitions\": %u, \"hw_queue_budget\": %u, \"gpu_max_hw_queues_env\": %d, ",
             (unsigned)HS_ED25519_ABI_VERSION, g_init.load() ? "true" : "false", kTenants, kTenantIndex, kTenantWide ? "true" : "false",
             kMaxPool[0], kMaxPool[1], kNarrowDepth, kPoolPartitions, hs_ed25519_hw_queue_budget(kTenants), g_hw_queues_env);
    o += t;
    snprintf(t, sizeof t, "\"max_batch_hint\": %zu, \"narrow_pool_max_batch\": %zu, \"zero_copy_max\": %zu, \"oct_max_batch\": %zu, \"oct_split_forced\": %d, "
             "\"autotune\": %d, \"quad_max_batch\": %zu, \"tail_max\": %zu, \"pairs_min_batches\": %lld, \"lane_min_batches\": %lld, \"lane_min_total\": %zu, "
             "\"cached_c_min_concurrent\": %zu, \"cached_c_min_lone\": %zu, \"rlc_min_batch\": %zu, "
             "\"min_gpu_batch\": {\"cold\": %zu, \"key_cache\": %zu}, ",
             g_max_batch.load(), (size_t)kPartitionMaxBatch, (size_t)kZeroCopyMax, g_oct_max_batch.load(), g_oct_split.load(), g_autotune.load(),
             g_quad_max_batch.load(), g_tail_max.load(), pm == (size_t)-1 ? -1ll : (long long)pm, lm == (size_t)-1 ? -1ll : (long long)lm, kLaneMinTotal,
             g_cached_c_min.load(), g_cached_c_min_lone.load(), hs_ed25519_rlc_min_batch(),
             hs_ed25519_min_gpu_batch(0), hs_ed25519_min_gpu_batch(1));
    o += t;
    o += "\"devices\": [";
    for (size_t i = 0; i < g_devs.size(); i++) {
        const dev_state &d = g_devs[i];
        const int pick = g_oct_pick[i].load(std::memory_order_relaxed);
        snprintf(t, sizeof t, "%s{\"slot\": %zu, \"ordinal\": %d, \"cus\": %u, \"oct_ms\": {\"k_verify_fused8\": %.4f, \"k_verify_fused8h\": %.4f}, "
                 "\"oct_pick\": \"%s\", \"oct_form_4097_8192\": \"%s\"}", i ? ", " : "", i, d.ordinal, d.cus, d.oct_ms[0], d.oct_ms[1],
                 pick < 0 ? "not measured" : (pick ? "k_verify_fused8h" : "k_verify_fused8"),
                 oct_split_for(6667, (size_t)d.cus * 4, pick) ? "k_verify_fused8h" : "k_verify_fused8");
        o += t;
    }
    o += "], \"env_seen\": {";
    for (size_t i = 0; i < g_env_seen.size(); i++) {
        std::string v;
        for (char ch : g_env_seen[i].second) if (ch != '"' && ch != '\\' && (unsigned char)ch >= 0x20) v += ch;
        o += (i ? ", \"" : "\"") + g_env_seen[i].first + "\": \"" + v + "\"";
    }
    o += "}}";
    if (buf && cap) { const size_t k = o.size() < cap - 1 ? o.size() : cap - 1; memcpy(buf, o.data(), k); buf[k] = 0; }
    return o.size();
}

// Below this many signatures per call the GPU round trip (launch + one wavefront's serial chain,
// ~0.40 ms cold / ~0.13 ms with the keys registered) costs more than verifying on one CPU core
// (~40-50 us per ed25519-dalek strict verification): the Rust shim keeps such calls - the
// reference's Block / Vote / Timeout::verify singles (consensus/src/messages.rs:64,144,258) and
// 4-node committees - on dalek.  Crossover measured by bench.py `small_batch_crossover`;
// env HS_ED25519_MIN_GPU_BATCH overrides.
extern "C" size_t hs_ed25519_min_gpu_batch(int keys_registered) {
    if (const char *e = getenv("HS_ED25519_MIN_GPU_BATCH")) { const long v = atol(e); if (v >= 0) return (size_t)v; }
    return keys_registered ? 4 : 10;
}

static int kernel_for(const hs_ed25519_ctx *ctx, size_t n) {
    const int f = ctx ? ctx->kernel : HS_K_AUTO;
    if (f == HS_K_AUTO || f == HS_K_CACHED_C || f == HS_K_OCT || f == HS_K_OCT_SPLIT) return n <= g_quad_max_batch.load() ? HS_K_FUSED : HS_K_LANE;
    return f == HS_K_FUSED_PAIRS ? HS_K_FUSED : f;      // the pairs form only exists for verify_device_batches
}
extern "C" const char *hs_ed25519_dominant_kernel_for(size_t n) {
    const size_t simds = g_init.load() && !g_devs.empty() && g_devs[0].cus ? (size_t)g_devs[0].cus * 4 : 1024;
    if (n <= simds * 8 && n <= g_oct_max_batch.load())
        return oct_split_for(n, simds, g_init.load() ? g_oct_pick[0].load(std::memory_order_relaxed) : -1) ? "k_verify_fused8h" : "k_verify_fused8";
    if (n <= g_quad_max_batch.load()) return n <= simds * 16 ? "k_verify_fused" : "k_verify_fused3";
    const size_t tail = n % (simds * 64);       // a sliver past whole rounds of wavefronts rides in the stage-2 launch (enqueue_cold)
    return n > simds * 64 && tail && tail <= g_tail_max.load() ? "k_verify_tail" : "k_verify";
}
extern "C" const char *hs_ed25519_dominant_kernel(void) { return hs_ed25519_dominant_kernel_for(1024); }
// (for `count` batches of 1,024 signatures each: the lane form also wants 8,192 signatures in the call)
extern "C" const char *hs_ed25519_batches_kernel_for(const hs_ed25519_ctx *ctx, size_t count) {
    const int k = ctx ? ctx->kernel : HS_K_AUTO;
    if (k == HS_K_AUTO && count == 1) return hs_ed25519_dominant_kernel_for(1024);      // a list of one takes the single-batch kernels
    if (k == HS_K_LANE || (k == HS_K_AUTO && count >= g_lane_min_batches.load() && count * 1024 >= kLaneMinTotal)) return "k_verify_batches";
    return k == HS_K_FUSED_PAIRS || (k == HS_K_AUTO && count >= g_pairs_min_batches.load()) ? "k_verify_fused_pairs_batches" : "k_verify_fused_batches";
}
// what a LONE call of n signatures through the key cache runs (a context marked concurrent switches to k_cached_chain2 earlier)
extern "C" const char *hs_ed25519_cached_kernel_for(size_t n) {
    return n >= cached_c_min_lone() ? "k_cached_chain2" : "k_verify_cached_lane2r";
}
static bool cached_c_form(const hs_ed25519_ctx *ctx, size_t n) {
    if (ctx->kernel == HS_K_CACHED_C) return true;
    return hs_k_auto_like(ctx->kernel) && n >= (ctx->several_in_flight ? g_cached_c_min.load() : cached_c_min_lone());
}

// One kernel launch.  With events the dispatch itself is time-stamped (hipExtLaunchKernel's
// start/stop events) instead of bracketing it with marker packets: at 16 concurrent queues two
// markers per step cost 14 % of small-batch throughput, dispatch timestamps ~3 %.
// (hipExtAnyOrderLaunch was tried for back-to-back independent batches on one stream: no
// effect on gfx950, as hip_ext.h warns for GFX9xx.)
static hipError_t launch1(const void *fn, dim3 grid, dim3 block, void **args, hipStream_t st, hipEvent_t start, hipEvent_t stop) {
    if (start || stop) return hipExtLaunchKernel(fn, grid, block, args, 0, st, start, stop, 0);
    return hipLaunchKernel(fn, grid, block, args, 0, st);
}

// enqueue the cold path over n signatures whose messages come from `msgs`
int hs::enqueue_cold(hs_ed25519_ctx *ctx, msg_src msgs, const uint8_t *d_pks, const uint8_t *d_sigs, size_t n,
                     uint32_t mode, uint8_t *d_bitmap, uint32_t *d_nreject, void *stream, const u32 *d_skip) {
    if (!d_pks || !d_sigs || !d_bitmap) return fail(HS_ED25519_E_ARG, "verify_device: null pointer");
    if (misaligned(d_pks, 16) || misaligned(d_sigs, 16) || misaligned(d_bitmap, 8) || misaligned(d_nreject, 4))
        return fail(HS_ED25519_E_ARG, "verify_device: misaligned pointer");
    if (mode > 1 || n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "verify_device: bad mode / n");
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const int kern = kernel_for(ctx, n);
    if (kern != HS_K_FUSED) { int rc = ctx_reserve(ctx, n, ctx->staging); if (rc) return rc; }
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const u32 np = (u32)round_up(n, 64);
    workspace ws = workspace_at(ctx->d_ws, np);
    const u32 nn = (u32)n;
    const bool rec = ctx->ev_slots && (ctx->ev_seq++ % ctx->ev_every) == 0;
    const size_t slot = rec ? ctx->ev_calls++ % ctx->ev_slots : 0;
    hipEvent_t *ev = rec ? &ctx->ev[3 * slot] : nullptr;
    const u32 *p_pk = (const u32 *)d_pks, *p_sg = (const u32 *)d_sigs;
    u32 nn_ = nn, mode_ = mode;
    unsigned long long *p_bm = (unsigned long long *)d_bitmap;
    u32 *p_nr = d_nreject;
    const u32 *p_skip = d_skip;
    void *a1[] = { &ws, &msgs, &p_pk, &p_sg, &nn_, &mode_, &p_skip };
    void *a2[] = { &ws, &nn_, &mode_, &p_bm, &p_nr, &p_skip };
    void *a3[] = { &msgs, &p_pk, &p_sg, &nn_, &mode_, &p_bm, &p_nr, &p_skip };
    // hipLaunchKernel returns its own status (hipGetLastError may carry another library's stale error)
    if (ev) ctx->ev_mid[slot] = kern != HS_K_FUSED;
    if (kern == HS_K_FUSED && ctx->kernel == HS_K_FUSED_PAIRS && !d_skip) {
        // the pairs form forced on a single batch (A/B measurements, tools/r2_sweeps.py): a one-entry batch list
        fused_batch_list L;
        L.count = 1;
        fused_batch &f = L.b[0];
        f.msgs = msgs; f.pks = p_pk; f.sigs = p_sg; f.bitmap16 = (unsigned short *)d_bitmap; f.nreject = d_nreject;
        f.n = nn; f.mode = mode; f.first_block = 0; f.pad = 0;
        void *a[] = { &L };
        HS_HIP(launch1((const void *)k_verify_fused_pairs_batches, dim3(np / 32), dim3(64), a, st, ev ? ev[0] : nullptr, ev ? ev[2] : nullptr),
               "launch k_verify_fused_pairs_batches");
    } else if (kern == HS_K_FUSED) {
        // 4-bit windows while the batch is at most one wavefront per SIMD (40,960 B of LDS per wavefront), 3-bit above.
        // "Per SIMD" = of the CUs this context's stream may use: a tenant's wide context or a narrow pool context is confined to
        // a slice of the chip, and a form tuned for one workgroup per SIMD would be oversubscribed there.
        const size_t simds = (size_t)(ctx->cus ? ctx->cus : g_devs[ctx->slot].cus) * 4;
        // eight lanes per signature: still one wavefront per SIMD at most - for a call that has the chip to itself.  With
        // several whole-chip calls at once (the slots of a round ring, hs_ed25519_ctx_set_concurrent) twice the wavefronts
        // per signature cost throughput (41 against 53 M signatures/s from four ring slots): four lanes per signature then.
        const int ksel = ctx->kernel == HS_K_CACHED_C ? HS_K_AUTO : ctx->kernel;      // a key-cache selector leaves cold calls alone
        const bool oct_forced = ksel == HS_K_OCT || ksel == HS_K_OCT_SPLIT;
        if (n <= simds * 8 && (oct_forced || (ksel == HS_K_AUTO && !ctx->several_in_flight && g_oct_max_batch.load() >= n))) {
            const bool split = oct_forced ? ksel == HS_K_OCT_SPLIT
                                          : oct_split_for(n, simds, g_oct_pick[ctx->slot].load(std::memory_order_relaxed), simds >= (size_t)g_devs[ctx->slot].cus * 4);
            if (split)
                HS_HIP(launch1((const void *)k_verify_fused8h, dim3(np / 8), dim3(128), a3, st, ev ? ev[0] : nullptr, ev ? ev[2] : nullptr),
                       "launch k_verify_fused8h");
            else
                HS_HIP(launch1((const void *)k_verify_fused8, dim3(np / 8), dim3(64), a3, st, ev ? ev[0] : nullptr, ev ? ev[2] : nullptr),
                       "launch k_verify_fused8");
        }
        else {
            const bool w4 = n <= simds * 16;
            HS_HIP(launch1(w4 ? (const void *)k_verify_fused : (const void *)k_verify_fused3, dim3(np / 16), dim3(64), a3, st,
                           ev ? ev[0] : nullptr, ev ? ev[2] : nullptr), "launch k_verify_fused");
        }
    } else {
        // One lane per signature: a launch is ceil(n / 64) wavefronts for the chip's 1,024 SIMDs, and its time is that of
        // the SIMD with most wavefronts - 66,667 signatures (the QC of N = 100,000) are 1,042 wavefronts, eighteen SIMDs
        // carry two, and the call took 1.49 ms where 65,536 signatures take 0.85.  When such a sliver
        // is left over it goes through the four-lane fused form instead (16 signatures per wavefront, a third of the one-lane
        // wavefront's time) - as the first workgroups of the stage-2 launch itself (k_verify_tail): 0.96 ms.  (It first ran as
        // k_verify_fused on a side stream between two events: 0.87 ms when the side stream had a hardware queue of its own,
        // 1.26 ms when the runtime had mapped it onto the main stream's; a side stream of higher priority: 1.2 ms every
        // time - profiles/r3_concurrent_ab.txt.)
        // Not behind an RLC stage (d_skip: the kernels here are its rare fallback and return at once when it accepted), and
        // not when several whole-chip calls run at once (the sliver's wavefronts then only compete with the other calls').
        const size_t round_sigs = (size_t)(ctx->cus ? ctx->cus : g_devs[ctx->slot].cus) * 4 * 64;
        const size_t tail = round_sigs ? n % round_sigs : 0;
        if ((hs_k_auto_like(ctx->kernel) || ctx->kernel == HS_K_CACHED_C) && !d_skip && !ctx->several_in_flight && n > round_sigs && tail && tail <= g_tail_max.load()) {
            const size_t main_n = n - tail;                      // a multiple of 64: the tail's bitmap starts on a word
            msg_src tmsgs = msgs;
            tmsgs.digests = msgs.digests + (size_t)msgs.stride_words * main_n;
            const u32 *t_pk = p_pk + (size_t)8 * main_n, *t_sg = p_sg + (size_t)16 * main_n;
            unsigned short *t_bm = (unsigned short *)(p_bm + main_n / 64);
            u32 tn = (u32)tail, tblocks = (u32)(round_up(tail, 64) / 16);
            nn_ = (u32)main_n;
            void *at[] = { &ws, &nn_, &mode_, &p_bm, &p_nr, &tmsgs, &t_pk, &t_sg, &tn, &t_bm, &tblocks };
            HS_HIP(launch1((const void *)k_prepare, dim3((2 * nn_ + 63) / 64), dim3(64), a1, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr),
                   "launch k_prepare");
            HS_HIP(launch1((const void *)k_verify_tail, dim3(nn_ / 64 + tblocks), dim3(64), at, st, nullptr, ev ? ev[2] : nullptr),
                   "launch k_verify_tail");
            return 0;
        }
        HS_HIP(launch1((const void *)k_prepare, dim3((2 * nn + 63) / 64), dim3(64), a1, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr),
               "launch k_prepare");
        HS_HIP(launch1((const void *)k_verify, dim3(np / 64), dim3(64), a2, st, nullptr, ev ? ev[2] : nullptr), "launch k_verify");
    }
    return 0;
}
int hs::make_digest_src(msg_src &m, const uint8_t *d_digests, size_t digest_stride) {
    if (!d_digests) return fail(HS_ED25519_E_ARG, "verify_device: null digests");
    if (misaligned(d_digests, 16)) return fail(HS_ED25519_E_ARG, "verify_device: misaligned digests");
    if (digest_stride != 0 && digest_stride != 32) return fail(HS_ED25519_E_ARG, "verify_device: bad digest_stride");
    m.digests = (const u32 *)d_digests; m.stride_words = (u32)(digest_stride / 4);
    return 0;
}
// timeout certificate: build the n per-vote digests into the context's buffer (k_tc_digests, same stream)
static int make_tc_src(hs_ed25519_ctx *ctx, msg_src &m, uint64_t round, const uint64_t *d_hq, size_t n, void *stream) {
    if (!d_hq) return fail(HS_ED25519_E_ARG, "verify_device_tc: null high_qc_rounds");
    if (misaligned(d_hq, 8)) return fail(HS_ED25519_E_ARG, "verify_device_tc: misaligned high_qc_rounds");
    if (n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "verify_device_tc: bad n");
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    if (n > ctx->tc_cap) {
        const size_t cap = round_up(n < 1024 ? 1024 : n, 64);
        hipStream_t own = ctx->stream;
        HS_HIP(hipStreamSynchronize(stream ? (hipStream_t)stream : own), "stream sync");    // the old buffer may be in use
        if (ctx->d_tc) (void)hipFree(ctx->d_tc);
        ctx->d_tc = nullptr; ctx->tc_cap = 0;
        HS_HIP(hipMalloc((void **)&ctx->d_tc, cap * 32), "hipMalloc TC digests");
        ctx->tc_cap = cap;
    }
    unsigned long long r = round;
    const unsigned long long *p_hq = (const unsigned long long *)d_hq;
    u32 nn = (u32)n;
    u32 *p_out = ctx->d_tc;
    void *a[] = { &r, &p_hq, &nn, &p_out };
    HS_HIP(hipLaunchKernel((const void *)k_tc_digests, dim3((nn + 63) / 64), dim3(64), a, 0,
                           stream ? (hipStream_t)stream : ctx->stream), "launch k_tc_digests");
    m.digests = ctx->d_tc; m.stride_words = 8;
    return 0;
}

extern "C" int hs_ed25519_verify_device(hs_ed25519_ctx *ctx, const uint8_t *d_digests, size_t digest_stride,
                                        const uint8_t *d_pks, const uint8_t *d_sigs, size_t n, uint32_t mode,
                                        uint8_t *d_bitmap, uint32_t *d_nreject, void *stream) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx) return fail(HS_ED25519_E_ARG, "verify_device: null context");
    if (mode > HS_ED25519_BATCH_RLC) return fail(HS_ED25519_E_ARG, "verify_device: bad mode");
    if (n == 0) return 0;
    msg_src m;
    int rc = make_digest_src(m, d_digests, digest_stride);
    if (rc) return rc;
    if (mode == HS_ED25519_BATCH_RLC) {     // dalek's verify_batch: one random linear combination, per-signature fallback
        u32 *d_ok = nullptr;
        rc = enqueue_rlc(ctx, nullptr, m, d_pks, nullptr, d_sigs, n, &d_ok, stream);
        if (rc) return rc;
        return enqueue_cold(ctx, m, d_pks, d_sigs, n, HS_ED25519_BATCH, d_bitmap, d_nreject, stream, d_ok);
    }
    return enqueue_cold(ctx, m, d_pks, d_sigs, n, mode, d_bitmap, d_nreject, stream);
}
// Every entry of a batch list is checked BEFORE the first launch: with more than 16 entries the list is cut into several
// launches, and an error found in a later chunk must not leave earlier chunks running on the caller's buffers.
static int check_batch_list(const hs_ed25519_batch *batches, size_t count, size_t key_align, const char *who) {
    static thread_local char msg[160];
    for (size_t i = 0; i < count; i++) {
        const hs_ed25519_batch &b = batches[i];
        if (b.n == 0) continue;
        const char *what = nullptr;
        if (b.mode > 1 || b.n > 32768) what = "mode must be BATCH / STRICT and n <= 32,768";
        else if (!b.d_pks || !b.d_sigs || !b.d_bitmap || !b.d_digests) what = "null pointer";
        else if (misaligned(b.d_pks, key_align) || misaligned(b.d_sigs, 16) || misaligned(b.d_bitmap, 8) || misaligned(b.d_nreject, 4) ||
                 misaligned(b.d_digests, 16)) what = "misaligned pointer";
        else if (b.digest_stride != 0 && b.digest_stride != 32) what = "digest_stride must be 0 or 32";
        if (what) { snprintf(msg, sizeof msg, "%s: entry %zu: %s", who, i, what); return hs::hs_fail(HS_ED25519_E_ARG, msg); }
    }
    return 0;
}
// the one-lane form of hs_ed25519_verify_device_batches: two launches per 16 batches, workspace slices from the context
static int enqueue_lane_batches(hs_ed25519_ctx *ctx, const hs_ed25519_batch *batches, size_t count, hipStream_t st) {
    size_t total = 0;
    for (size_t i = 0; i < count; i++) total += round_up(batches[i].n, 64);
    if (total == 0) return 0;
    int rc = ctx_reserve(ctx, total, ctx->staging);
    if (rc) return rc;
    size_t done = 0, ws_off = 0;            // ws_off: padded signatures already given a workspace slice
    while (done < count) {
        lane_batch_list L;
        L.count = 0;
        u32 pw = 0, vw = 0;
        for (; done < count && L.count < HS_MAX_LANE_BATCHES; done++) {
            const hs_ed25519_batch &b = batches[done];
            if (b.n == 0) continue;
            lane_batch &f = L.b[L.count++];
            rc = make_digest_src(f.msgs, b.d_digests, b.digest_stride);
            if (rc) return rc;
            const u32 np = (u32)round_up(b.n, 64);
            f.ws = workspace_at(ctx->d_ws + workspace_words(ws_off), np);
            ws_off += np;
            f.pks = (const u32 *)b.d_pks; f.sigs = (const u32 *)b.d_sigs; f.bitmap = (unsigned long long *)b.d_bitmap;
            f.nreject = b.d_nreject; f.n = (u32)b.n; f.mode = b.mode; f.first_p = pw; f.first_v = vw;
            pw += 2 * (np / 64); vw += np / 64;          // A blocks, then R blocks
        }
        if (!L.count) continue;
        const bool rec = ctx->ev_slots && (ctx->ev_seq++ % ctx->ev_every) == 0;
        const size_t slot = rec ? ctx->ev_calls++ % ctx->ev_slots : 0;
        hipEvent_t *ev = rec ? &ctx->ev[3 * slot] : nullptr;
        if (ev) ctx->ev_mid[slot] = 1;
        void *a[] = { &L };
        HS_HIP(launch1((const void *)k_prepare_batches, dim3(pw), dim3(64), a, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr), "launch k_prepare_batches");
        HS_HIP(launch1((const void *)k_verify_batches, dim3(vw), dim3(64), a, st, nullptr, ev ? ev[2] : nullptr), "launch k_verify_batches");
    }
    return 0;
}
// Several independent small batches (each <= 32,768 signatures, BATCH or STRICT) in one launch per 16 of them.
extern "C" int hs_ed25519_verify_device_batches(hs_ed25519_ctx *ctx, const hs_ed25519_batch *batches, size_t count, void *stream) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx || (!batches && count)) return fail(HS_ED25519_E_ARG, "verify_device_batches: null context / list");
    { const int rc = check_batch_list(batches, count, 16, "verify_device_batches"); if (rc) return rc; }
    if (count == 1 && hs_k_auto_like(ctx->kernel) && batches[0].n) {
        // a list of one is a single batch: it gets the single-batch kernels (the eight-lane latency form up to 8,192 signatures)
        msg_src m;
        const int rc = make_digest_src(m, batches[0].d_digests, batches[0].digest_stride);
        if (rc) return rc;
        return enqueue_cold(ctx, m, batches[0].d_pks, batches[0].d_sigs, batches[0].n, batches[0].mode, batches[0].d_bitmap, batches[0].d_nreject, stream);
    }
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // pairs form (two signatures per quad, 11 % less work, 1.85x a wavefront's latency) once the caller hands over
    // enough batches per launch that the shorter form's extra wavefronts would only queue; HS_K_FUSED / HS_K_FUSED_PAIRS
    // on the context force either
    bool lane = ctx->kernel == HS_K_LANE;
    if (hs_k_auto_like(ctx->kernel) && count >= g_lane_min_batches.load()) {
        size_t total = 0;
        for (size_t i = 0; i < count; i++) total += batches[i].n;
        lane = total >= kLaneMinTotal;
    }
    if (lane) return enqueue_lane_batches(ctx, batches, count, st);
    const bool pairs = ctx->kernel == HS_K_FUSED_PAIRS || (hs_k_auto_like(ctx->kernel) && count >= g_pairs_min_batches.load());
    const u32 per_block = pairs ? 32 : 16;
    size_t done = 0;
    while (done < count) {
        fused_batch_list L;
        L.count = 0;
        u32 blocks = 0;
        for (; done < count && L.count < HS_MAX_BATCHES_PER_LAUNCH; done++) {
            const hs_ed25519_batch &b = batches[done];
            if (b.n == 0) continue;
            fused_batch &f = L.b[L.count++];
            int rc = make_digest_src(f.msgs, b.d_digests, b.digest_stride);
            if (rc) return rc;
            f.pks = (const u32 *)b.d_pks; f.sigs = (const u32 *)b.d_sigs; f.bitmap16 = (unsigned short *)b.d_bitmap;
            f.nreject = b.d_nreject; f.n = (u32)b.n; f.mode = b.mode; f.first_block = blocks; f.pad = 0;
            blocks += (u32)(round_up(b.n, 64) / per_block);
        }
        if (!L.count) continue;
        const bool rec = ctx->ev_slots && (ctx->ev_seq++ % ctx->ev_every) == 0;
        const size_t slot = rec ? ctx->ev_calls++ % ctx->ev_slots : 0;
        hipEvent_t *ev = rec ? &ctx->ev[3 * slot] : nullptr;
        if (ev) ctx->ev_mid[slot] = 0;
        void *a[] = { &L };
        HS_HIP(launch1(pairs ? (const void *)k_verify_fused_pairs_batches : (const void *)k_verify_fused_batches, dim3(blocks), dim3(64), a,
                       st, ev ? ev[0] : nullptr, ev ? ev[2] : nullptr),
               pairs ? "launch k_verify_fused_pairs_batches" : "launch k_verify_fused_batches");
    }
    return 0;
}
// The same for the key-cache path: batches[i].d_pks holds the uint32 key indices of batch i.
extern "C" int hs_ed25519_verify_device_batches_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, const hs_ed25519_batch *batches,
                                                       size_t count, void *stream) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx || !ks || (!batches && count)) return fail(HS_ED25519_E_ARG, "verify_device_batches_cached: null context / key set / list");
    if (ks->slot != ctx->slot) return fail(HS_ED25519_E_ARG, "verify_device_batches_cached: the key set lives on another device slot than the context");
    { const int rc = check_batch_list(batches, count, 4, "verify_device_batches_cached"); if (rc) return rc; }
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    keyset_view kv; kv.tab = ks->d_tab; kv.pk = ks->d_pk; kv.flag = ks->d_flag;
    u32 nkeys = (u32)ks->nkeys;
    const u32 *p_bt = g_devs[ctx->slot].base_tab16;
    size_t done = 0;
    if (ctx->kernel == HS_K_CACHED_C) {
        // by selector only: chains + the batched finish over the list (k_cached_chain2_batches, k_cached_finish_batches), every
        // batch with its own slice of the context's workspace - for lists of small batches the finish launch is a tail as long
        // as the chains themselves (185 against 269 M verifies/s at four 1,024-signature batches per launch)
        size_t total = 0;
        for (size_t i = 0; i < count; i++) total += round_up(batches[i].n, 64);
        if (total == 0) return 0;
        int rc = ctx_reserve(ctx, total, ctx->staging);      // HS_CFIN_ROWS words per signature of workspace_words' 643
        if (rc) return rc;
        size_t ws_off = 0;
        while (done < count) {
            cachedc_batch_list L;
            L.count = 0; L.m = 4;
            u32 blocks = 0, fblocks = 0;
            for (; done < count && L.count < HS_MAX_BATCHES_PER_LAUNCH; done++) {
                const hs_ed25519_batch &b = batches[done];
                if (b.n == 0) continue;
                cachedc_batch &f = L.b[L.count++];
                rc = make_digest_src(f.msgs, b.d_digests, b.digest_stride);
                if (rc) return rc;
                f.key_idx = (const u32 *)b.d_pks; f.sigs = (const u32 *)b.d_sigs; f.bitmap = (unsigned long long *)b.d_bitmap;
                f.nreject = b.d_nreject; f.n = (u32)b.n; f.mode = b.mode;
                f.np = (u32)round_up(b.n, 64); f.nl = (u32)round_up((b.n + L.m - 1) / L.m, 64);
                f.fin = ctx->d_ws + (size_t)HS_CFIN_ROWS * ws_off;
                ws_off += f.np;
                f.first_block = blocks; f.first_f = fblocks;
                blocks += f.np / 64; fblocks += f.nl / 64;
            }
            if (!L.count) continue;
            const bool rec = ctx->ev_slots && (ctx->ev_seq++ % ctx->ev_every) == 0;
            const size_t slot = rec ? ctx->ev_calls++ % ctx->ev_slots : 0;
            hipEvent_t *ev = rec ? &ctx->ev[3 * slot] : nullptr;
            if (ev) ctx->ev_mid[slot] = 1;
            void *ac[] = { &kv, &nkeys, &p_bt, &L };
            void *af[] = { &L };
            HS_HIP(launch1((const void *)k_cached_chain2_batches, dim3(blocks), dim3(128), ac, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr),
                   "launch k_cached_chain2_batches");
            HS_HIP(launch1((const void *)k_cached_finish_batches, dim3(fblocks), dim3(64), af, st, nullptr, ev ? ev[2] : nullptr),
                   "launch k_cached_finish_batches");
        }
        return 0;
    }
    while (done < count) {
        cached_batch_list L;
        L.count = 0;
        u32 blocks = 0;
        for (; done < count && L.count < HS_MAX_BATCHES_PER_LAUNCH; done++) {
            const hs_ed25519_batch &b = batches[done];
            if (b.n == 0) continue;
            cached_batch &f = L.b[L.count++];
            int rc = make_digest_src(f.msgs, b.d_digests, b.digest_stride);
            if (rc) return rc;
            f.key_idx = (const u32 *)b.d_pks; f.sigs = (const u32 *)b.d_sigs; f.bitmap32 = (u32 *)b.d_bitmap;
            f.nreject = b.d_nreject; f.n = (u32)b.n; f.mode = b.mode; f.first_block = blocks; f.pad = 0;
            blocks += (u32)(round_up(b.n, 64) / 64);
        }
        if (!L.count) continue;
        const bool rec = ctx->ev_slots && (ctx->ev_seq++ % ctx->ev_every) == 0;
        const size_t slot = rec ? ctx->ev_calls++ % ctx->ev_slots : 0;
        hipEvent_t *ev = rec ? &ctx->ev[3 * slot] : nullptr;
        if (ev) ctx->ev_mid[slot] = 0;
        void *a[] = { &kv, &nkeys, &p_bt, &L };
        HS_HIP(launch1((const void *)k_verify_cached_lane2r_batches, dim3(blocks), dim3(192), a, st, ev ? ev[0] : nullptr, ev ? ev[2] : nullptr),
               "launch k_verify_cached_lane2r_batches");
    }
    return 0;
}
// TC::verify shape with the per-vote digests built on the device (strict semantics)
extern "C" int hs_ed25519_verify_device_tc(hs_ed25519_ctx *ctx, uint64_t round, const uint64_t *d_high_qc_rounds,
                                           const uint8_t *d_pks, const uint8_t *d_sigs, size_t n, uint8_t *d_bitmap,
                                           uint32_t *d_nreject, void *stream) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx) return fail(HS_ED25519_E_ARG, "verify_device_tc: null context");
    if (n == 0) return 0;
    msg_src m;
    int rc = make_tc_src(ctx, m, round, d_high_qc_rounds, n, stream);
    if (rc) return rc;
    return enqueue_cold(ctx, m, d_pks, d_sigs, n, HS_ED25519_STRICT, d_bitmap, d_nreject, stream);
}

extern "C" int hs_ed25519_ctx_set_profiling(hs_ed25519_ctx *ctx, size_t slots) {
    if (!ctx) return fail(HS_ED25519_E_ARG, "set_profiling: null context");
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    ctx->ev.clear(); ctx->ev_slots = 0; ctx->ev_calls = 0; ctx->ev_seq = 0;
    ctx->ev_mid.assign(slots, 0);
    for (size_t i = 0; i < 3 * slots; i++) {
        hipEvent_t e;
        HS_HIP(hipEventCreate(&e), "hipEventCreate");
        ctx->ev.push_back(e);
    }
    ctx->ev_slots = slots;
    return 0;
}
// Sample only every `every`-th call (event markers are queue packets: at 16 concurrent queues
// three of them per step cost ~14 % of small-batch throughput).
extern "C" int hs_ed25519_ctx_set_profiling_stride(hs_ed25519_ctx *ctx, size_t every) {
    if (!ctx || every == 0) return fail(HS_ED25519_E_ARG, "set_profiling_stride: bad arguments");
    ctx->ev_every = every; ctx->ev_seq = 0;
    return 0;
}
extern "C" int hs_ed25519_ctx_kernel_times(hs_ed25519_ctx *ctx, double *prepare_ms_sum, double *verify_ms_sum,
                                           size_t *launches) {
    if (!ctx || !prepare_ms_sum || !verify_ms_sum || !launches) return fail(HS_ED25519_E_ARG, "kernel_times: null");
    const size_t cnt = ctx->ev_calls < ctx->ev_slots ? ctx->ev_calls : ctx->ev_slots;
    double a = 0, b = 0;
    for (size_t i = 0; i < cnt; i++) {
        float t0 = 0, t1 = 0;
        if (ctx->ev_mid[i]) {
            HS_HIP(hipEventElapsedTime(&t0, ctx->ev[3 * i], ctx->ev[3 * i + 1]), "hipEventElapsedTime");
            HS_HIP(hipEventElapsedTime(&t1, ctx->ev[3 * i + 1], ctx->ev[3 * i + 2]), "hipEventElapsedTime");
        } else {
            HS_HIP(hipEventElapsedTime(&t1, ctx->ev[3 * i], ctx->ev[3 * i + 2]), "hipEventElapsedTime");
        }
        a += t0; b += t1;
    }
    *prepare_ms_sum = a; *verify_ms_sum = b; *launches = cnt;
    return 0;
}

// ---- key sets --------------------------------------------------------------------------------
// tables of `nkeys` keys: projective multiples into a scratch buffer (chunks of kBuildChunk keys, 2.7 GB), then one
// inversion per entry into the affine table
static const size_t kBuildChunk = 4096;
static int build_tables(u32 *d_tab, const u32 *d_axy, size_t nkeys, bool negate, hipStream_t st) {
    const size_t chunk = nkeys < kBuildChunk ? nkeys : kBuildChunk;
    u32 *d_proj = nullptr;
    HS_HIP(hipMalloc((void **)&d_proj, chunk * (size_t)HS_KS_KEY_PROJ_WORDS * 4), "hipMalloc table scratch");
    int rc = 0;
    for (size_t c0 = 0; c0 < nkeys && !rc; c0 += chunk) {
        const size_t nc = nkeys - c0 < chunk ? nkeys - c0 : chunk;
        const u32 nk = (u32)nc, pad = (u32)round_up(nc, 16), neg = negate ? 1u : 0u;
        const u32 *p_axy = d_axy + (size_t)20 * c0;
        void *args[] = { &d_proj, &p_axy, (void *)&nk, (void *)&pad, (void *)&neg };
        hipError_t e = hipLaunchKernel((const void *)k_keyset_build, dim3(pad / 16 * HS_KS_WINDOWS), dim3(64), args, 0, st);
        if (e == hipSuccess) {
            const u32 *p_proj = d_proj; u32 *p_tab = d_tab + c0 * (size_t)HS_KS_KEY_WORDS;
            static_assert((HS_KS_WINDOWS * HS_KS_ENTRIES) % HS_KS_AFFINE_GROUP == 0, "a group of entries must not span two keys");
            unsigned long long groups = (unsigned long long)nc * HS_KS_WINDOWS * HS_KS_ENTRIES / HS_KS_AFFINE_GROUP;
            void *a2[] = { &p_proj, &p_tab, &groups };
            e = hipLaunchKernel((const void *)k_keyset_affine, dim3((unsigned)((groups + 63) / 64)), dim3(64), a2, 0, st);
        }
        if (e != hipSuccess) rc = fail(HS_ED25519_E_DEVICE, "launch k_keyset_build / k_keyset_affine", e);
    }
    const hipError_t es = hipStreamSynchronize(st);          // the scratch is freed below
    (void)hipFree(d_proj);
    if (!rc && es != hipSuccess) rc = fail(HS_ED25519_E_DEVICE, "table build", es);
    return rc;
}
static int ensure_base_table(int slot) {        // the tables are built by hs_ed25519_init_devices; this only checks
    if (!g_init.load() || slot < 0 || slot >= (int)g_devs.size()) return fail(HS_ED25519_E_ARG, "no such device slot");
    if (!g_devs[slot].base_tab || !g_devs[slot].base_tab16) return fail(HS_ED25519_E_DEVICE, "the base point's tables are missing");
    return 0;
}
const u32 *hs::dev_base_table(int slot) {
    if (ensure_base_table(slot)) return nullptr;
    return g_devs[slot].base_tab;
}

extern "C" size_t hs_ed25519_keyset_bytes(size_t nkeys) { return nkeys * ((size_t)HS_KS_KEY_WORDS * 4 + 32 + 4 + 128); }

extern "C" int hs_ed25519_keyset_create_on(hs_ed25519_keyset **out, const uint8_t *pks, size_t nkeys, int slot) {
    if (!out || !pks || nkeys == 0 || nkeys > 0x0fffffffu) return fail(HS_ED25519_E_ARG, "keyset_create: bad arguments");
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (slot < 0 || slot >= (int)g_devs.size()) return fail(HS_ED25519_E_ARG, "keyset_create: no such device slot");
    int rc = ensure_base_table(slot);
    if (rc) return rc;
    HS_HIP(hipSetDevice(g_devs[slot].ordinal), "hipSetDevice");
    hs_ed25519_keyset *ks = new (std::nothrow) hs_ed25519_keyset();
    if (!ks) return fail(HS_ED25519_E_NOMEM, "keyset_create: out of memory");
    ks->slot = slot;
    // Registration runs on a wide pool context's stream, not on the legacy stream: a legacy-stream operation is an error
    // (hipErrorStreamCaptureImplicit) while another thread captures a round ring's hipGraph, and a node registers the next epoch's
    // committee while rounds stream (round 5: the thread-sanitizer run of tests/cpp/stress_host.cpp hit it twice in 48,000 calls).
    hs_ed25519_ctx *pc = nullptr;
    rc = pool_get(&pc, slot, (size_t)-1);
    if (rc) { delete ks; return rc; }
    struct putter { hs_ed25519_ctx *c; ~putter() { hs::pool_put(c); } } pool_guard{pc};
    hipStream_t st = pc->stream;
    u32 *d_in = nullptr, *d_axy = nullptr;
    hipError_t e = hipSuccess;
    do {
        if ((e = hipMalloc((void **)&ks->d_tab, nkeys * (size_t)HS_KS_KEY_WORDS * 4)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&ks->d_pk, nkeys * 32)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&ks->d_flag, nkeys * 4)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&ks->d_niels, nkeys * 128)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&d_in, nkeys * 32)) != hipSuccess) break;
        if ((e = hipMalloc((void **)&d_axy, nkeys * 80)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(d_in, pks, nkeys * 32, hipMemcpyHostToDevice, st)) != hipSuccess) break;
        const u32 nk = (u32)nkeys;
        void *a[] = { &d_in, (void *)&nk, &d_axy, &ks->d_pk, &ks->d_flag, &ks->d_niels };
        if ((e = hipLaunchKernel((const void *)k_keyset_decompress, dim3((nk + 63) / 64), dim3(64), a, 0, st)) != hipSuccess) break;
        if (build_tables(ks->d_tab, d_axy, nkeys, true, st)) { e = hipErrorUnknown; break; }
        e = hipStreamSynchronize(st);
    } while (0);
    if (d_in) (void)hipFree(d_in);
    if (d_axy) (void)hipFree(d_axy);
    if (e != hipSuccess) {
        if (ks->d_tab) (void)hipFree(ks->d_tab);
        if (ks->d_pk) (void)hipFree(ks->d_pk);
        if (ks->d_flag) (void)hipFree(ks->d_flag);
        if (ks->d_niels) (void)hipFree(ks->d_niels);
        delete ks;
        return fail(e == hipErrorOutOfMemory ? HS_ED25519_E_NOMEM : HS_ED25519_E_DEVICE, "keyset_create", e);
    }
    ks->nkeys = nkeys;
    *out = ks;
    return 0;
}
extern "C" int hs_ed25519_keyset_create(hs_ed25519_keyset **out, const uint8_t *pks, size_t nkeys) {
    return hs_ed25519_keyset_create_on(out, pks, nkeys, 0);
}
extern "C" void hs_ed25519_keyset_destroy(hs_ed25519_keyset *ks) {
    if (!ks) return;
    // No hipDeviceSynchronize here (until round 5 there was one, return value ignored): a device-wide synchronisation marks
    // every stream that is capturing on ANY thread invalidated, and a node drops the last epoch's committee while its rounds
    // stream through a graph-mode ring (tests/test_gpu_ring.py, registration while capturing: one run in two failed).  What
    // must hold is that nothing still reads the tables: the host API counts its uncollected work on the set (engine.h, `users`).
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(10);
    while (ks->users.load(std::memory_order_acquire) > 0 && std::chrono::steady_clock::now() < deadline)
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (ks->users.load(std::memory_order_acquire) > 0) {      // leak rather than free under a running kernel; the struct stays too (late releases touch it)
        (void)fail(HS_ED25519_E_ARG, "keyset_destroy: work is still in flight on the key set after 10 s (tickets not collected, a round ring "
                                     "not destroyed?) - its tables are leaked, not freed");
        return;
    }
    const int ord = hs::dev_ordinal(ks->slot);
    if (ord >= 0) (void)hipSetDevice(ord);
    if (ks->d_tab) (void)hipFree(ks->d_tab);
    if (ks->d_pk) (void)hipFree(ks->d_pk);
    if (ks->d_flag) (void)hipFree(ks->d_flag);
    if (ks->d_niels) (void)hipFree(ks->d_niels);
    delete ks;
}
extern "C" size_t hs_ed25519_keyset_size(const hs_ed25519_keyset *ks) { return ks ? ks->nkeys : 0; }

int hs::enqueue_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, msg_src msgs, const uint32_t *d_key_idx,
                       const uint8_t *d_sigs, size_t n, uint32_t mode, uint8_t *d_bitmap, uint32_t *d_nreject, void *stream,
                       const u32 *d_skip) {
    if (!d_key_idx || !d_sigs || !d_bitmap) return fail(HS_ED25519_E_ARG, "verify_device_cached: null pointer");
    if (misaligned(d_key_idx, 4) || misaligned(d_sigs, 16) || misaligned(d_bitmap, 8) || misaligned(d_nreject, 4))
        return fail(HS_ED25519_E_ARG, "verify_device_cached: misaligned pointer");
    if (mode > 1 || n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "verify_device_cached: bad mode / n");
    if (ks->slot != ctx->slot) return fail(HS_ED25519_E_ARG, "verify_device_cached: the key set lives on another device slot than the context");
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const u32 np = (u32)round_up(n, 64);
    keyset_view kv; kv.tab = ks->d_tab; kv.pk = ks->d_pk; kv.flag = ks->d_flag;
    u32 nkeys = (u32)ks->nkeys, nn = (u32)n, mode_ = mode;
    const u32 *p_bt = g_devs[ctx->slot].base_tab16, *p_ki = d_key_idx, *p_sg = (const u32 *)d_sigs;
    unsigned short *p_bm = (unsigned short *)d_bitmap;
    u32 *p_nr = d_nreject;
    const u32 *p_skip = d_skip;
    void *a[] = { &kv, &nkeys, &p_bt, &msgs, &p_ki, &p_sg, &nn, &mode_, &p_bm, &p_nr, &p_skip };
    const bool rec = ctx->ev_slots && (ctx->ev_seq++ % ctx->ev_every) == 0;
    const size_t slot = rec ? ctx->ev_calls++ % ctx->ev_slots : 0;
    hipEvent_t *ev = rec ? &ctx->ev[3 * slot] : nullptr;
    if (ev) ctx->ev_mid[slot] = 0;
    hipEvent_t e0 = ev ? ev[0] : nullptr, e2 = ev ? ev[2] : nullptr;
    if (cached_c_form(ctx, n)) {
        // chains, then HS_CFIN_M signatures per finish lane (verify_cached.h): R' and the flags pass through the context's workspace
        int rc = ctx_reserve(ctx, n, ctx->staging);
        if (rc) return rc;
        u32 *fin = ctx->d_ws;
        // signatures per finish lane: 4 for a call that is alone (the tail is latency: 0.290 against 0.337 ms at 66,667), 8 among
        // others (414 against 410 M signatures/s with four in flight; 16 from 131,072 bought 1 % - 441 against 436 M - and was
        // retired in round 5 with the other sub-5 % switches)
        u32 m = ctx->several_in_flight ? 8u : 4u;
        u32 nl = (u32)round_up((n + m - 1) / m, 64), np_ = np;
        unsigned long long *p_bm64 = (unsigned long long *)d_bitmap;
        void *ac[] = { &kv, &nkeys, &p_bt, &msgs, &p_ki, &p_sg, &nn, &mode_, &fin, &np_, &p_skip };
        void *af[] = { &fin, &np_, &nl, &m, &nn, &p_sg, &mode_, &p_bm64, &p_nr, &p_skip };
        if (ev) ctx->ev_mid[slot] = 1;
        HS_HIP(launch1((const void *)k_cached_chain2, dim3(np / 64), dim3(128), ac, st, e0, ev ? ev[1] : nullptr), "launch k_cached_chain2");
        HS_HIP(launch1((const void *)k_cached_finish, dim3(nl / 64), dim3(64), af, st, nullptr, e2), "launch k_cached_finish");
        return 0;
    }
    HS_HIP(launch1((const void *)k_verify_cached_lane2r, dim3(np / 64), dim3(192), a, st, e0, e2), "launch k_verify_cached_lane2r");
    return 0;
}
extern "C" int hs_ed25519_verify_device_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, const uint8_t *d_digests,
                                               size_t digest_stride, const uint32_t *d_key_idx, const uint8_t *d_sigs,
                                               size_t n, uint32_t mode, uint8_t *d_bitmap, uint32_t *d_nreject, void *stream) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx || !ks) return fail(HS_ED25519_E_ARG, "verify_device_cached: null context / key set");
    if (mode > HS_ED25519_BATCH_RLC) return fail(HS_ED25519_E_ARG, "verify_device_cached: bad mode");
    if (n == 0) return 0;
    msg_src m;
    int rc = make_digest_src(m, d_digests, digest_stride);
    if (rc) return rc;
    if (mode == HS_ED25519_BATCH_RLC) {     // A served from the key set's Niels records; fallback = the cached per-signature kernels
        u32 *d_ok = nullptr;
        rc = enqueue_rlc(ctx, ks, m, nullptr, d_key_idx, d_sigs, n, &d_ok, stream);
        if (rc) return rc;
        return enqueue_cached(ctx, ks, m, d_key_idx, d_sigs, n, HS_ED25519_BATCH, d_bitmap, d_nreject, stream, d_ok);
    }
    return enqueue_cached(ctx, ks, m, d_key_idx, d_sigs, n, mode, d_bitmap, d_nreject, stream);
}
extern "C" int hs_ed25519_verify_device_tc_cached(hs_ed25519_ctx *ctx, const hs_ed25519_keyset *ks, uint64_t round,
                                                  const uint64_t *d_high_qc_rounds, const uint32_t *d_key_idx,
                                                  const uint8_t *d_sigs, size_t n, uint8_t *d_bitmap, uint32_t *d_nreject,
                                                  void *stream) {
    if (!g_init.load()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx || !ks) return fail(HS_ED25519_E_ARG, "verify_device_tc_cached: null context / key set");
    if (n == 0) return 0;
    msg_src m;
    int rc = make_tc_src(ctx, m, round, d_high_qc_rounds, n, stream);
    if (rc) return rc;
    return enqueue_cached(ctx, ks, m, d_key_idx, d_sigs, n, HS_ED25519_STRICT, d_bitmap, d_nreject, stream);
}

