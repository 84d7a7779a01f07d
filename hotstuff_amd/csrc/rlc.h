// rlc.h - the batch random-linear-combination check: what ed25519-dalek 1.0.1 `verify_batch`
// computes when the reference calls it at /root/reference/crypto/src/lib.rs:218 (reached from
// QC::verify, consensus/src/messages.rs:197):
//
//      -[sum z_i s_i mod L] B  +  sum [z_i] R_i  +  sum [z_i h_i mod L] A_i   ==   O
//
// with 127-bit z_i (dalek: 128-bit), h_i = SHA-512(R_i || A_i || M) mod L, evaluated as ONE multi-scalar
// multiplication over 2n + 1 points (dalek: Straus below 190 points, Pippenger above; here
// Pippenger with signed 8-bit windows throughout, shaped for the GPU):
//
//   stage 1  rlc_point / rlc_scalars   one lane per point: decompress to an affine Niels record
//            (128 B, one cache line); the R lane also hashes, draws z_i, forms z_i h_i and z_i s_i
//            mod L, recodes both scalars into 32 signed radix-256 digits (one byte each) and adds
//            z_i s_i into eight lazy 64-bit limb sums.
//   stage 2  (kernel only) counting sort of the points of every window by |digit| -> per-bucket runs
//   stage 3  rlc_accum_lane            one lane per chunk of at most C consecutive points of one bucket's run
//            (a bucket gets ceil(len / C) lanes, so skewed buckets - the 5-bit top window of the 253-bit
//            scalars - cost lanes, not time): mixed additions
//   stage 4  rlc_bucket_fold / the wavefront-shuffle form in hs_rlc.hip: chunk partials -> bucket sums
//            -> sum_j j * bucket_j  (suffix scan + reduction across the 64 lanes of a wavefront);
//            rlc_b_point: [-(sum z_i s_i)]B from the base point's window table, beside it
//   stage 5  rlc_final_quad            Horner over the 32 window sums (248 doublings, four lanes per
//            point), + the [b]B point, identity test.
//
// All of it is host+device so that tests/emu runs the same arithmetic on the CPU against oracle/.
// z_i: dalek draws them from a merlin transcript finalised with thread_rng(); here
// z_i = SHA-512(key || h_i || s_i || le64(i))[..16] mod 2^127 with a 32-byte process secret `key` - a uniform
// coefficient of the same strength, unpredictable to whoever assembled the batch, and reproducible under a pinned key.
#pragma once
#include "verify_cached.h"

namespace hs {

#define HS_RLC_WINDOWS 32
#define HS_RLC_BUCKETS 128
#define HS_RLC_REC_WORDS 32            // ypx[10] ymx[10] xy2d[10] flags pad  = 128 bytes
#define HS_RLC_ID_CACHED 0x40000000u   // sorted entry: the record lives in the key set's Niels array
#define HS_RLC_ID_NEG 0x80000000u      // sorted entry: subtract the point

struct rlc_key { u32 w[8]; };

// One call's buffers.  Points: p in [0, np) = R_p, p in [np, 2 np) = A_(p - np); record 2 np is the
// identity (what idle lanes add).
struct rlc_view {
    u32 *rec;                      // [2 np + 1][32]
    uint8_t *dig;                  // [32][2 np]   recoded scalar bytes, 128 = digit 0
    unsigned long long *csum;      // [8]  lazy limb sums of z_i s_i mod L
    u32 *failed;                   // != 0: a signature of the batch fails the format / curve checks
    u32 np;
};

// coefficient of signature `index`: the low 127 bits of the keyed hash (z < 2^127 keeps the signed radix-256
// recoding inside 16 windows: no carry window that would put half of all R points into one bucket); z[4..7] = 0
HS_HD void rlc_coeff(u32 z[8], const rlc_key &key, const u32 k[8], const u32 s[8], u32 index) {
    u32 in[26];
#pragma unroll
    for (int i = 0; i < 8; i++) { in[i] = key.w[i]; in[8 + i] = k[i]; in[16 + i] = s[i]; }
    in[24] = index; in[25] = 0;
    u64 w[16], hh[8];
#pragma unroll
    for (int i = 0; i < 13; i++) w[i] = ((u64)bswap32(in[2 * i]) << 32) | bswap32(in[2 * i + 1]);
    w[13] = 0x8000000000000000ull; w[14] = 0; w[15] = 104 * 8;
    sha512_oneblock(hh, w);
    z[0] = bswap32((u32)(hh[0] >> 32)); z[1] = bswap32((u32)hh[0]);
    z[2] = bswap32((u32)(hh[1] >> 32)); z[3] = bswap32((u32)hh[1]) & 0x7fffffffu;
    z[4] = z[5] = z[6] = z[7] = 0;
}

HS_HD void rlc_store_record(u32 *rec, const ge_p3 &p, u32 flags) {
    fe ypx, ymx, xy2d;
    fe_add(ypx, p.Y, p.X); fe_carry(ypx);
    fe_sub(ymx, p.Y, p.X); fe_carry(ymx);
    fe_mul(xy2d, p.T, fe_d2());
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    uint4 *q = reinterpret_cast<uint4 *>(rec);
    q[0] = make_uint4(ypx.v[0], ypx.v[1], ypx.v[2], ypx.v[3]);
    q[1] = make_uint4(ypx.v[4], ypx.v[5], ypx.v[6], ypx.v[7]);
    q[2] = make_uint4(ypx.v[8], ypx.v[9], ymx.v[0], ymx.v[1]);
    q[3] = make_uint4(ymx.v[2], ymx.v[3], ymx.v[4], ymx.v[5]);
    q[4] = make_uint4(ymx.v[6], ymx.v[7], ymx.v[8], ymx.v[9]);
    q[5] = make_uint4(xy2d.v[0], xy2d.v[1], xy2d.v[2], xy2d.v[3]);
    q[6] = make_uint4(xy2d.v[4], xy2d.v[5], xy2d.v[6], xy2d.v[7]);
    q[7] = make_uint4(xy2d.v[8], xy2d.v[9], flags, 0u);
#else
    for (int j = 0; j < 10; j++) { rec[j] = ypx.v[j]; rec[10 + j] = ymx.v[j]; rec[20 + j] = xy2d.v[j]; }
    rec[30] = flags; rec[31] = 0;
#endif
}
HS_HD void rlc_load_record(ge_niels &q, const u32 *rec) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const uint4 *r = reinterpret_cast<const uint4 *>(rec);
    u32 w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = r[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#else
    const u32 *w = rec;
#endif
#pragma unroll
    for (int j = 0; j < 10; j++) { q.ypx.v[j] = w[j]; q.ymx.v[j] = w[10 + j]; q.xy2d.v[j] = w[20 + j]; }
}
HS_HD void rlc_store_identity_record(u32 *rec) {
    for (int j = 0; j < 32; j++) rec[j] = 0;
    rec[0] = 1; rec[10] = 1; rec[30] = HS_F_ONCURVE;
}

// stage 1a: the point of task `p` (enc = its 32 encoding bytes): decompress, write the record.
// Returns false when the encoding is not on the curve.
HS_HD bool rlc_point(const rlc_view &v, u32 p, const u32 enc[8]) {
    ge_p3 pt;
    const bool ok = ge_decompress(pt, enc);
    rlc_store_record(v.rec + (size_t)HS_RLC_REC_WORDS * p, pt, ok ? HS_F_ONCURVE : 0);
    return ok;
}

HS_HD void rlc_store_digits(uint8_t *dig, size_t stride, const u32 r[8]) {
#pragma unroll
    for (int w = 0; w < HS_RLC_WINDOWS; w++) dig[(size_t)w * stride] = (uint8_t)(r[w >> 2] >> (8 * (w & 3)));
}
HS_HD void rlc_add_csum(unsigned long long *csum, const unsigned long long c[8]) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        atomicAdd(csum + j, c[j]);
#else
        csum[j] += c[j];
#endif
    }
}
// stage 1b: the scalars of signature i (pk / sig / msg: its bytes as little-endian words).
// c = z_i s_i mod L (zero when s >= L), to be added into v.csum by the caller - the kernel first sums it across
// the wavefront: 8 atomics per SIGNATURE on the same eight addresses serialised k_rlc_prepare (0.27 ms of atomics at
// 66,667 signatures).  Returns false when s >= L.
// zi: the index the coefficient is drawn for - the signature's position in the WHOLE call (a shard of a sharded combination
// passes its offset + i, so that equal tuples in different shards still get different coefficients)
HS_HD bool rlc_scalars(const rlc_view &v, u32 i, const rlc_key &key, const u32 *pk, const u32 *sig, const u32 *msg, u32 c[8], u32 zi) {
    u32 in[24], h[16], k[8], s[8], z[8], a[8], r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { in[j] = sig[j]; in[8 + j] = pk[j]; in[16 + j] = msg[j]; s[j] = sig[8 + j]; }
    sha512_96(h, in);
    sc_reduce512(k, h);
    const bool canon = sc_is_canonical(s);
    rlc_coeff(z, key, k, s, zi);
    sc_mul(a, z, k);                       // z_i h_i mod L  (dalek: Scalar multiplication)
    sc_mul(c, z, s);                       // z_i s_i mod L
    const size_t stride = 2 * (size_t)v.np;
    sc_recode256(r, z);
    rlc_store_digits(v.dig + i, stride, r);
    sc_recode256(r, a);
    rlc_store_digits(v.dig + v.np + i, stride, r);
    if (!canon) {
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = 0;
    }
    return canon;
}

HS_HD bool rlc_scalars(const rlc_view &v, u32 i, const rlc_key &key, const u32 *pk, const u32 *sig, const u32 *msg, u32 c[8]) {
    return rlc_scalars(v, i, key, pk, sig, msg, c, i);
}

// stage 3: one lane's run [lo, hi) of a window's sorted list, as a loop of `trip` steps shared by
// the wavefront (idle lanes add the identity record).  ksn: the key set's Niels array or null.
HS_HD void rlc_accum_lane(ge_p3 &acc, const u32 *sorted, u32 lo, u32 hi, u32 trip, const u32 *rec, const u32 *ksn, u32 id_identity) {
    ge_identity(acc);
    if (trip == 0) return;
    // software-pipelined: the entry and the 128-byte record of step t + 1 are in flight while step t's addition
    // computes (two dependent loads per step, ~3 us of latency a two-wavefronts-per-SIMD kernel cannot hide otherwise)
    u32 e = lo < hi ? sorted[lo] : id_identity;
    ge_niels q;
    rlc_load_record(q, (e & HS_RLC_ID_CACHED) ? ksn + (size_t)HS_RLC_REC_WORDS * (e & 0x3fffffffu) : rec + (size_t)HS_RLC_REC_WORDS * (e & 0x3fffffffu));
#pragma nounroll
    for (u32 t = 0; t < trip; t++) {
        const u32 pos = lo + t + 1;
        const u32 e_next = (t + 1 < trip && pos < hi) ? sorted[pos] : id_identity;
        ge_niels q_next;
        rlc_load_record(q_next, (e_next & HS_RLC_ID_CACHED) ? ksn + (size_t)HS_RLC_REC_WORDS * (e_next & 0x3fffffffu)
                                                            : rec + (size_t)HS_RLC_REC_WORDS * (e_next & 0x3fffffffu));
        ge_p1p1 s;
        ge_madd(s, acc, q, (e & HS_RLC_ID_NEG) != 0);
        ge_p1p1_to_p3(acc, s, true);
        q = q_next; e = e_next;
    }
}

// b = -(sum) mod L from the eight lazy limb sums
HS_HD void rlc_b_scalar(u32 b[8], const unsigned long long *csum) {
    u32 x[16];
    u64 carry = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u64 lo = j < 8 ? (csum[j] & 0xffffffffull) : 0, hi = (j >= 1 && j <= 8) ? (csum[j - 1] >> 32) : 0;
        const u64 t = lo + hi + carry;
        x[j] = (u32)t; carry = t >> 32;
    }
    u32 r[8];
    sc_reduce512(r, x);
    const u32 Lw[8] = HS_L_WORDS;
    u32 nz = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) nz |= r[j];
    u64 borrow = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u64 d = (u64)Lw[j] - r[j] - borrow;
        b[j] = nz ? (u32)d : 0u;
        borrow = (d >> 63) & 1;
    }
}

// stage 5: Horner over the window sums + the [b]B point (slots written by rlc_store_wsum), identity test.
// Quad form: one coordinate per lane.  Returns the per-lane flags; the caller ANDs the quad.
HS_Q qv rlc_load_point_quad(const u32 *pt40) {
#if defined(__HIPCC__)
    qv r;
    const u32 *p = pt40 + 10 * q_lane();
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = p[i];
    return r;
#else
    qv r; for (int l = 0; l < 4; l++) for (int i = 0; i < 10; i++) r.l[l].v[i] = pt40[10 * l + i]; return r;
#endif
}
// window sums as handed from stage 4 to stage 5: slot w < 32 = W_w, slot 32 = [b]B; each slot holds the point
// twice, extended (X, Y, Z, T) in words 0-39 and cached (Y-X, Y+X, 2dT, 2Z) in words 40-79
#define HS_RLC_WSUM_WORDS 80
HS_HD void rlc_store_wsum(u32 *slot, const ge_p3 &p) {
    fe ymx, ypx, t2d, z2;
    fe_sub(ymx, p.Y, p.X); fe_carry(ymx);
    fe_add(ypx, p.Y, p.X); fe_carry(ypx);
    fe_mul(t2d, p.T, fe_d2());
    fe_add(z2, p.Z, p.Z); fe_carry(z2);
    for (int i = 0; i < 10; i++) {
        slot[i] = p.X.v[i]; slot[10 + i] = p.Y.v[i]; slot[20 + i] = p.Z.v[i]; slot[30 + i] = p.T.v[i];
        slot[40 + i] = ymx.v[i]; slot[50 + i] = ypx.v[i]; slot[60 + i] = t2d.v[i]; slot[70 + i] = z2.v[i];
    }
}
// [b]B, b = -(sum z_i s_i) mod L, one lane, from the base point's window table (verify_cached.h layout)
HS_HD void rlc_b_point(ge_p3 &bb, const u32 *btab, const unsigned long long *csum) {
    u32 b[8], bw[8];
    rlc_b_scalar(b, csum);
    sc_recode256(bw, b);
    cached_chain(bb, btab, bw);
}
// the same point on a quad (one coordinate per lane): a single lane would run 256 field multiplications
// back to back, the slowest thing in stage 4
HS_Q qv rlc_b_point_quad(const u32 *btab, const unsigned long long *csum) {
    u32 b[8], bw[8];
    rlc_b_scalar(b, csum);
    sc_recode256(bw, b);
    qv acc = q_lanes(fe_small(0), fe_small(1), fe_small(1), fe_small(0));
#pragma nounroll
    for (int w = HS_KS_WINDOWS - 1; w >= 0; w--) {
        const int ds = sc_pop_digit8(bw);
        const int is = ds < 0 ? -ds : ds;
        const qv eb = q_load_entry(btab + ((size_t)w * HS_KS_ENTRIES + is) * HS_KS_ENTRY_WORDS);
        acc = q_add_cached(acc, q_cached_signed(eb, ds < 0));
    }
    return acc;
}
// slot = extended (X, Y, Z, T) then cached (Y-X, Y+X, 2dT, 2Z), from a quad
HS_Q void rlc_store_wsum_quad(u32 *slot, const qv &p) {
    const qv c = q_to_cached(p);
#if defined(__HIPCC__)
    const u32 l = q_lane();
#pragma unroll
    for (int i = 0; i < 10; i++) { slot[10 * l + i] = p.v[i]; slot[40 + 10 * l + i] = c.v[i]; }
#else
    for (int l = 0; l < 4; l++) for (int i = 0; i < 10; i++) { slot[10 * l + i] = p.l[l].v[i]; slot[40 + 10 * l + i] = c.l[l].v[i]; }
#endif
}
// Horner over the 32 window sums + the [b]B point: the value of the whole combination, one coordinate per lane
HS_Q qv rlc_final_point_quad(const u32 *wsum) {
    qv acc = rlc_load_point_quad(wsum + (size_t)HS_RLC_WSUM_WORDS * (HS_RLC_WINDOWS - 1));
#pragma nounroll
    for (int w = HS_RLC_WINDOWS - 2; w >= 0; w--) {
#pragma nounroll
        for (int d = 0; d < 8; d++) acc = q_dbl(acc);
        acc = q_add_cached(acc, rlc_load_point_quad(wsum + (size_t)HS_RLC_WSUM_WORDS * w + 40));
    }
    return q_add_cached(acc, rlc_load_point_quad(wsum + (size_t)HS_RLC_WSUM_WORDS * HS_RLC_WINDOWS + 40));
}
// identity: X == 0 (lane 0), Y == Z (lane 1), Z != 0 (lane 2; fail closed on a zeroed accumulator)
HS_Q qflags rlc_identity_quad(const qv &acc) {
    const qv z = q_perm<2, 2, 2, 2>(acc);
    qflags f;
#if defined(__HIPCC__)
    const u32 lane = q_lane();
    f.f = lane == 0 ? fe_iszero(acc) : (lane == 1 ? fe_eq(q_carry(acc), q_carry(z)) : (lane == 2 ? !fe_iszero(acc) : true));
#else
    f.f[0] = fe_iszero(acc.l[0]);
    { fe y = acc.l[1], zz = z.l[1]; fe_carry(y); fe_carry(zz); f.f[1] = fe_eq(y, zz); }
    f.f[2] = !fe_iszero(acc.l[2]); f.f[3] = true;
#endif
    return f;
}
HS_Q qflags rlc_final_quad(const u32 *wsum) { return rlc_identity_quad(rlc_final_point_quad(wsum)); }

// One combination cut over several GPUs (SURVEY.md 8e "batch-MSM mode"): every device reduces ITS shard of the signatures
// to one partial point under the same coefficient key - a record of HS_RLC_PART_WORDS words: the point in the window-sum slot
// format (extended + cached, 80 words) and the shard's format / curve failure word - and whoever holds all of them adds them up
// and tests the identity ONCE, which is exactly dalek's single verify_batch over the whole batch.
#define HS_RLC_PART_WORDS 96           // 80 point words | failed | pad  = 384 bytes
HS_Q qflags rlc_join_quad(const u32 *parts, u32 count) {
    qv acc = rlc_load_point_quad(parts);
#pragma nounroll
    for (u32 s = 1; s < count; s++) acc = q_add_cached(acc, rlc_load_point_quad(parts + (size_t)HS_RLC_PART_WORDS * s + 40));
    return rlc_identity_quad(acc);
}

// stage 4, serial form (tests/emu and the definition the wavefront-shuffle kernel must match):
// wsum = sum_{j=1..128} j * bucket_j by the running-sum trick
HS_HD void rlc_bucket_fold(ge_p3 &wsum, const ge_p3 *bucket /* [128], bucket[j-1] = sum of the points with |digit| = j */) {
    ge_p3 run, tot;
    ge_identity(run); ge_identity(tot);
    for (int j = HS_RLC_BUCKETS - 1; j >= 0; j--) {
        ge_add_p3(run, run, bucket[j]);
        ge_add_p3(tot, tot, run);
    }
    wsum = tot;
}

}  // namespace hs
