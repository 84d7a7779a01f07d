// verify_cached.h - verification against a registered key set (committee key cache).
//
// In the reference every verified key belongs to the committee, which is static per epoch
// (consensus/src/config.rs:39-43; Block/Vote/QC/Timeout/TC::verify all check
// committee.stake(author) > 0 before the signature, consensus/src/messages.rs:57-61,
// 138-142, 182-194, 252-256, 292-304).  So the variable-base half of
//      [k](-A) + [s]B == R
// can be tabled per key once per epoch.  With signed 8-bit windows
//      [k](-A) = sum_w T_A[w][k_w],   T_A[w][j] = j * 2^(8w) * (-A),   w < 32, j <= 128
// and the base point's half on 16-bit windows (further down) a verification is 32 + 16 table additions: no
// doublings, no decompression of A.  R' = sum is
// then compared with R in one of two ways.  (a) Without decompressing R: invert Z once, compare
// y' with the encoded y and the parity of x' with the sign bit (equivalent to dalek's "decompress
// R, compare points": if y' = y_R the encoding is on the curve and decodes to (+-x', y'); the sign
// bit picks x' unless x' = 0, where dalek accepts either sign).  (b) Against an R that another
// wavefront of the workgroup decompressed meanwhile, as points: X == x_R Z, Y == y_R Z, Z != 0 -
// the form the default kernels use, because it takes the 254-squaring chain off the addition chains.
//
// Cost: 48 additions + 1 inversion or square root, against 252 doublings + 80 additions + 2
// square roots for the cold path.  Price: 129 * 32 entries of 128 bytes = 528,384 bytes of HBM per key
// (5.3 GB for 10,000 validators, 53 GB for 100,000 - MI355X has 288 GB).
//
// Table entries are AFFINE Niels points (y-x, y+x, 2dxy) padded to 128 bytes = exactly one cache line per
// lookup (round 1 stored projective (Y-X, Y+X, 2dT, 2Z), 160 bytes straddling two or three lines: an
// addition then cost 8 multiplications instead of 7, and a gather twice the lines).  Registration builds
// the projective multiples with the quad kernel's formulas and normalises every entry with an inversion of
// its own (one lane per entry; 4,128 inversions per key, ~4 us of chip time per key).  The quad forms read
// the fourth coordinate, 2Z = 2, from a constant; entry 0 of every window is the identity.
// Same accept set as verify_lane / verify_quad (tests run all 70 edge vectors through it).
#pragma once
#include "verify_quad.h"

namespace hs {

#define HS_KS_WINDOWS 32
#define HS_KS_ENTRIES 129
#define HS_KS_ENTRY_WORDS 32                                   // (y-x)[10] (y+x)[10] (2dxy)[10] pad[2]: 128 bytes
#define HS_KS_KEY_WORDS (HS_KS_WINDOWS * HS_KS_ENTRIES * HS_KS_ENTRY_WORDS)
#define HS_KS_PROJ_WORDS 40                                    // registration scratch: projective (Y-X, Y+X, 2dT, 2Z), 4 lanes x 10 limbs
#define HS_KS_KEY_PROJ_WORDS (HS_KS_WINDOWS * HS_KS_ENTRIES * HS_KS_PROJ_WORDS)

// The base point's half of the sum uses 16-bit windows: B is one fixed point, so its table may be as large as pays -
//      [s]B = sum_w T_B16[w][s_w],   T_B16[w][j] = j * 2^(16w) * B,   w < 16, j <= 32,768
// 16 additions instead of 32 for 16 x 32,769 x 128 bytes = 67 MB per GPU (MALL-resident; the per-key tables cannot
// grow the same way: 528 KB per key already).  A verification against a registered key is then 32 + 16 = 48 table
// additions.  The table is built on the device from the 8-bit one: j = 256 hi + lo with lo in [-127, 128], so
// T_B16[w][j] = T_B[2w + 1][hi] +- T_B[2w][|lo|] - one addition per entry, then the shared-inversion normalisation.
#define HS_B16_WINDOWS 16
#define HS_B16_ENTRIES 32769
#define HS_B16_WORDS ((size_t)HS_B16_WINDOWS * HS_B16_ENTRIES * HS_KS_ENTRY_WORDS)
static_assert((HS_B16_WINDOWS * HS_B16_ENTRIES) % 8 == 0, "the 16-bit base table is normalised in groups of eight entries");

// per-key header written by the registration pass
struct keyset_view {
    const u32 *tab;      // [key][window][entry][32 words]
    const u32 *pk;       // [key][8]   the 32 key bytes (hash input)
    const u32 *flag;     // [key]      HS_F_ONCURVE | HS_F_SMALL
};

#if defined(__HIPCC__)
HS_Q bool q_quad_all(const qflags &f) {       // all four lanes set, uniform over the quad
    const int v = f.f ? 1 : 0;
    const int a = __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true), b = __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true);
    const int c = __builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true), d = __builtin_amdgcn_mov_dpp(v, 0xff, 0xf, 0xf, true);
    return a && b && c && d;
}
HS_Q void q_store_entry(u32 *entry, const qv &e) {      // a PROJECTIVE entry of the registration scratch (40 words)
    uint2 *p = reinterpret_cast<uint2 *>(entry + 10 * q_lane());
#pragma unroll
    for (int i = 0; i < 5; i++) p[i] = make_uint2(e.v[2 * i], e.v[2 * i + 1]);
}
#else
HS_Q bool q_quad_all(const qflags &f) { return f.f[0] && f.f[1] && f.f[2] && f.f[3]; }
HS_Q void q_store_entry(u32 *entry, const qv &e) {
    for (int l = 0; l < 4; l++) for (int i = 0; i < 10; i++) entry[10 * l + i] = e.l[l].v[i];
}
#endif

// ---- registration: one quad per (key, window), then one lane per entry ------------------------
// ax, ay: the decompressed key (affine); negate: tables of -A (keys) or +B (base point).
// Writes entries 0..128 of window `w` into the PROJECTIVE scratch: j * 2^(8w) * (+-A) as (Y-X, Y+X, 2dT, 2Z).
HS_Q void keyset_build_window(u32 *win /* [129][40] */, const fe &ax, const fe &ay, bool negate, int w) {
    fe x = ax;
    if (negate) fe_neg(x, ax);
    qv p = q_mul(q_lanes(x, ay, fe_small(1), x), q_lanes(fe_small(1), fe_small(1), fe_small(1), ay));
#pragma nounroll
    for (int d = 0; d < 8 * w; d++) p = q_dbl(p);
    q_store_entry(win, q_lanes(fe_small(1), fe_small(1), fe_small(0), fe_small(2)));
    const qv e1 = q_to_cached(p);
    q_store_entry(win + HS_KS_PROJ_WORDS, e1);
#pragma nounroll
    for (int j = 2; j < HS_KS_ENTRIES; j++) {
        p = q_add_cached(p, e1);
        q_store_entry(win + (size_t)j * HS_KS_PROJ_WORDS, q_to_cached(p));
    }
}
// projective (A, B, C, D) = (Y-X, Y+X, 2dT, 2Z)  ->  affine (y-x, y+x, 2dxy) = (A, B, C) * 2 / D   (2dxy = 2dT / Z)
HS_HD void keyset_affine_entry(u32 *out /* 32 words */, const u32 *in /* 40 words */) {
    fe a, b, c, d, inv;
#pragma unroll
    for (int i = 0; i < 10; i++) { a.v[i] = in[i]; b.v[i] = in[10 + i]; c.v[i] = in[20 + i]; d.v[i] = in[30 + i]; }
    fe_invert(inv, d);
    fe_add(inv, inv, inv);                  // 2 / D  (2x: the first operand of the multiplications below)
    fe_mul(a, inv, a); fe_mul(b, inv, b); fe_mul(c, inv, c);
#pragma unroll
    for (int i = 0; i < 10; i++) { out[i] = a.v[i]; out[10 + i] = b.v[i]; out[20 + i] = c.v[i]; }
    out[30] = 0; out[31] = 0;
}
// The same for HS_KS_AFFINE_GROUP consecutive entries with ONE inversion (Montgomery's trick): prefix products of the
// D's, invert the last, peel the individual inverses off backwards.  3 + 5 multiplications per entry and an eighth of
// an inversion (~ 265 multiplications) instead of a whole one: registration of 10,000 keys went from 44 to ~ 10 ms.
// 4,128 entries per key = 516 groups, so a group never spans two keys (a key that is not on the curve - never read
// by the verification kernels - cannot spoil a neighbour's entries through a zero D).
#define HS_KS_AFFINE_GROUP 8
HS_HD void keyset_affine_group(u32 *out /* G x 32 words */, const u32 *in /* G x 40 words */) {
    fe pre[HS_KS_AFFINE_GROUP - 1];          // pre[i] = D_0 D_1 ... D_{i+1}
    {
        fe d0, d1;
#pragma unroll
        for (int i = 0; i < 10; i++) { d0.v[i] = in[30 + i]; d1.v[i] = in[HS_KS_PROJ_WORDS + 30 + i]; }
        fe_mul(pre[0], d0, d1);
    }
#pragma unroll
    for (int j = 2; j < HS_KS_AFFINE_GROUP; j++) {
        fe d;
#pragma unroll
        for (int i = 0; i < 10; i++) d.v[i] = in[j * HS_KS_PROJ_WORDS + 30 + i];
        fe_mul(pre[j - 1], pre[j - 2], d);
    }
    fe inv; fe_invert(inv, pre[HS_KS_AFFINE_GROUP - 2]);        // 1 / (D_0 ... D_{G-1})
#pragma unroll
    for (int j = HS_KS_AFFINE_GROUP - 1; j >= 0; j--) {
        const u32 *e = in + j * HS_KS_PROJ_WORDS;
        fe dinv;                                                // 1 / D_j
        if (j > 0) {
            fe d, prev;
#pragma unroll
            for (int i = 0; i < 10; i++) { d.v[i] = e[30 + i]; prev.v[i] = j >= 2 ? pre[j - 2].v[i] : in[30 + i]; }
            fe_mul(dinv, inv, prev);
            fe_mul(inv, inv, d);
        } else dinv = inv;
        fe_add(dinv, dinv, dinv);                               // 2 / D_j (2x: first operand below)
        fe a, b, c;
#pragma unroll
        for (int i = 0; i < 10; i++) { a.v[i] = e[i]; b.v[i] = e[10 + i]; c.v[i] = e[20 + i]; }
        fe_mul(a, dinv, a); fe_mul(b, dinv, b); fe_mul(c, dinv, c);
        u32 *o = out + j * HS_KS_ENTRY_WORDS;
#pragma unroll
        for (int i = 0; i < 10; i++) { o[i] = a.v[i]; o[10 + i] = b.v[i]; o[20 + i] = c.v[i]; }
        o[30] = 0; o[31] = 0;
    }
}

// ---- one lane per signature (throughput form, large batches) -------------------------------
// Same tables, read whole (one 128-byte line per lookup and lane).  Per signature:
// 64 serial table additions (7 field multiplications each) + one inversion, against 4x-replicated
// work in the quad form - ~715 field multiplications per verification.
struct tab_entry { fe ymx, ypx, t2d; };          // affine (y-x, y+x, 2dxy) as stored

HS_HD void load_tab_entry(tab_entry &e, const u32 *p) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const uint4 *q = reinterpret_cast<const uint4 *>(p);       // the whole 128-byte line
    u32 w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#else
    const u32 *w = p;
#endif
#pragma unroll
    for (int i = 0; i < 10; i++) { e.ymx.v[i] = w[i]; e.ypx.v[i] = w[10 + i]; e.t2d.v[i] = w[20 + i]; }
}
// r = p +- e  (extended += stored affine entry); 7 multiplications, result reduced
HS_HD void ge_add_tab(ge_p3 &r, const ge_p3 &p, const tab_entry &e, bool neg) {
    fe qa, qb, a, b, c, d, t;
    fe_select(qa, e.ymx, e.ypx, neg);
    fe_select(qb, e.ypx, e.ymx, neg);
    fe_sub(t, p.Y, p.X); fe_mul(a, t, qa);        // A = (Y1-X1)(Y2-+X2)
    fe_add(t, p.Y, p.X); fe_mul(b, t, qb);        // B = (Y1+X1)(Y2+-X2)
    fe_mul(c, p.T, e.t2d);                        // C = T1 2dT2
    fe_add(d, p.Z, p.Z); fe_carry(d);             // D = 2 Z1   (Z2 = 1)
    fe E, H, zs, zd, F, G;
    fe_sub(E, b, a);                              // 3x
    fe_add(H, b, a);                              // 2x
    fe_add(zs, d, c);                             // 2x
    fe_sub(zd, d, c);                             // 3x
    fe_select(F, zd, zs, neg);                    // F = D -+ C
    fe_select(G, zs, zd, neg);                    // G = D +- C
    fe_mul(r.X, E, F); fe_mul(r.Y, G, H); fe_mul(r.Z, F, G); fe_mul(r.T, E, H);
}
// The same addition in two halves, so that a chain can fetch its NEXT entry between them: the first half is the only
// one that reads the entry (three multiplications), the second (four multiplications, ~600 instructions) then covers
// the gather's latency - a table line is a first touch in a multi-GB table, 1-2 us away.  Left to the compiler the ten
// loads of an entry sit ~50 instructions ahead of their first use.
struct add_tab_half { fe a, b, c; };
HS_HD void ge_add_tab_first(add_tab_half &h, const ge_p3 &p, const tab_entry &e, bool neg) {
    fe qa, qb, t;
    fe_select(qa, e.ymx, e.ypx, neg);
    fe_select(qb, e.ypx, e.ymx, neg);
    fe_sub(t, p.Y, p.X); fe_mul(h.a, t, qa);
    fe_add(t, p.Y, p.X); fe_mul(h.b, t, qb);
    fe_mul(h.c, p.T, e.t2d);
}
HS_HD void ge_add_tab_second(ge_p3 &r, const ge_p3 &p, const add_tab_half &h, bool neg) {
    fe d, E, H, zs, zd, F, G;
    fe_add(d, p.Z, p.Z); fe_carry(d);
    fe_sub(E, h.b, h.a);
    fe_add(H, h.b, h.a);
    fe_add(zs, d, h.c);
    fe_sub(zd, d, h.c);
    fe_select(F, zd, zs, neg);
    fe_select(G, zs, zd, neg);
    fe_mul(r.X, E, F); fe_mul(r.Y, G, H); fe_mul(r.Z, F, G); fe_mul(r.T, E, H);
}
// sum_w T[w][d_w] over the 32 windows of one table and one radix-256 recoded scalar
HS_HD void cached_chain(ge_p3 &acc, const u32 *tab, u32 dw[8]) {
    ge_identity(acc);
    tab_entry e;
    int d = sc_pop_digit8(dw);
    load_tab_entry(e, tab + ((size_t)(HS_KS_WINDOWS - 1) * HS_KS_ENTRIES + (d < 0 ? -d : d)) * HS_KS_ENTRY_WORDS);
#pragma nounroll
    for (int w = HS_KS_WINDOWS - 1; w >= 0; w--) {
        const bool neg = d < 0;
        add_tab_half h;
        ge_add_tab_first(h, acc, e, neg);
        if (w > 0) {            // the next window's entry, fetched behind the second half
            d = sc_pop_digit8(dw);
            load_tab_entry(e, tab + ((size_t)(w - 1) * HS_KS_ENTRIES + (d < 0 ? -d : d)) * HS_KS_ENTRY_WORDS);
        }
        ge_add_tab_second(acc, acc, h, neg);
    }
}
// Entry j of window w of the 16-bit base table, PROJECTIVE (Y-X, Y+X, 2dT, 2Z; 40 words - what keyset_affine_group
// normalises), from the 8-bit table t8 of +B: the stored affine entry (y-x, y+x, 2dxy) of 2^(8(2w+1)) hi B becomes the
// extended point (2x, 2y, 2, 2xy) - 2xy = 2dxy / d - and the entry of 2^(16w) |lo| B is added to or subtracted from it.
HS_HD void base16_entry_projective(u32 *out /* 40 words */, const u32 *t8, u32 w, u32 j) {
    u32 hi = j >> 8;
    int lo = (int)(j & 255u);
    if (lo > 128) { lo -= 256; hi += 1; }           // j <= 32,768, so hi <= 128 either way
    tab_entry eh, el;
    load_tab_entry(eh, t8 + ((size_t)(2 * w + 1) * HS_KS_ENTRIES + hi) * HS_KS_ENTRY_WORDS);
    load_tab_entry(el, t8 + ((size_t)(2 * w) * HS_KS_ENTRIES + (u32)(lo < 0 ? -lo : lo)) * HS_KS_ENTRY_WORDS);
    ge_p3 p, r;
    fe_sub(p.X, eh.ypx, eh.ymx); fe_carry(p.X);
    fe_add(p.Y, eh.ypx, eh.ymx); fe_carry(p.Y);
    p.Z = fe_small(2);
    const u32 dinv[10] = FE26_DINV;
    fe_mul(p.T, eh.t2d, fe_const(dinv));
    ge_add_tab(r, p, el, lo < 0);
    ge_cached c;
    ge_to_cached(c, r);
    fe z2; fe_add(z2, c.Z, c.Z); fe_carry(z2);
#pragma unroll
    for (int i = 0; i < 10; i++) { out[i] = c.YmX.v[i]; out[10 + i] = c.YpX.v[i]; out[20 + i] = c.T2d.v[i]; out[30 + i] = z2.v[i]; }
}
// p + q for two extended points (q converted to cached form on the fly)
HS_HD void ge_add_p3(ge_p3 &r, const ge_p3 &p, const ge_p3 &q) {
    ge_cached c;
    ge_to_cached(c, q);
    ge_p1p1 t;
    ge_add(t, p, c, false);
    ge_p1p1_to_p3(r, t, true);
}
// The default two-lane form: the 32 + 16 additions of one signature split EVENLY, 24 per lane, returning R' for a
// projective comparison against an R decompressed elsewhere.  The even lane sums the key's windows 31..8; the odd
// lane the key's windows 7..0 and then the base point's sixteen 16-bit windows.  One instruction stream: every step
// is "pop a digit, gather one 128-byte entry, add" with a per-lane table pointer, window stride and digit source.
// kw: k recoded in radix 256, sw: s recoded in radix 65536 (both uniform over the pair).
struct lane2_split { u32 d8[8], d16[8]; };
// this lane's digit words: even = k as is (its top 24 digits are used), odd = k shifted up by 24 digits (its low 8
// digits first) and s
HS_HD void lane2_split_init(lane2_split &ls, const u32 kw[8], const u32 sw[8], bool odd) {
#pragma unroll
    for (int i = 0; i < 8; i++) { ls.d8[i] = odd ? (i >= 6 ? kw[i - 6] : 0u) : kw[i]; ls.d16[i] = sw[i]; }
}
// step t = 0..23 of a lane's chain: the entry to add and its sign
HS_HD const u32 *lane2_split_step(lane2_split &ls, int t, bool odd, const u32 *atab, const u32 *b16, bool &neg) {
    const int d8 = sc_pop_digit8(ls.d8);
    int d = d8;
    size_t row = (size_t)((odd ? 7 : 31) - t) * HS_KS_ENTRIES;       // the key's window 31 - t (even) or 7 - t (odd, t < 8)
    const u32 *tab = atab;
    if (t >= 8) {                                                     // uniform over the wavefront
        const int d16 = sc_pop_digit16(ls.d16);
        if (odd) { d = d16; row = (size_t)(23 - t) * HS_B16_ENTRIES; tab = b16; }
    }
    neg = d < 0;
    return tab + (row + (size_t)(d < 0 ? -d : d)) * HS_KS_ENTRY_WORDS;
}
HS_HD void lane2_split_chain(ge_p3 &mine, const u32 *atab, const u32 *b16, const u32 kw[8], const u32 sw[8], bool odd) {
    lane2_split ls;
    lane2_split_init(ls, kw, sw, odd);
    ge_identity(mine);
    tab_entry e;
    bool neg;
    load_tab_entry(e, lane2_split_step(ls, 0, odd, atab, b16, neg));
#pragma nounroll
    for (int t = 0; t < 24; t++) {
        add_tab_half h;
        ge_add_tab_first(h, mine, e, neg);
        bool neg_next = false;
        if (t + 1 < 24) load_tab_entry(e, lane2_split_step(ls, t + 1, odd, atab, b16, neg_next));   // fetched behind the second half
        ge_add_tab_second(mine, mine, h, neg);
        neg = neg_next;
    }
}
#if defined(__HIPCC__)
HS_Q void cached_lane2_chain(ge_p3 &acc, const u32 *atab, const u32 *b16, u32 kw[8], u32 sw[8]) {
    ge_p3 mine, other;
    lane2_split_chain(mine, atab, b16, kw, sw, (threadIdx.x & 1u) != 0);
    other.X = q_perm<1, 0, 3, 2>(mine.X); other.Y = q_perm<1, 0, 3, 2>(mine.Y);
    other.Z = q_perm<1, 0, 3, 2>(mine.Z); other.T = q_perm<1, 0, 3, 2>(mine.T);
    ge_add_p3(acc, mine, other);
}
#endif
// ---- finishing by compression, one inversion per HS_CFIN_M signatures ("c" forms) ----------------------------------
// cached_finish above inverts Z per signature; the default latency kernel avoids that by having a third wavefront
// decompress the R's beside the chains - 254 squarings either way, a third of a key-cache verification's instructions.
// An inversion, unlike a square root, batches (Montgomery): a lane that finishes M signatures one after the other needs
// ONE inversion and 4 multiplications per signature.  Lanes of a wavefront share their instruction stream, so the batch
// must be sequential work of ONE lane - hence a second, tiny kernel: the chain kernel leaves every signature's R' =
// (X : Y : Z) and its format / key flags in a structure-of-arrays workspace, and a finish lane walks HS_CFIN_M of them:
//      pre_j = Z_0 ... Z_j  (stored);   inv = 1 / pre_{M-1};   backwards:  1/Z_j = inv pre_{j-1},  inv = inv Z_j
// then x' = X / Z, y' = Y / Z against R's 32 bytes exactly as cached_finish does (y' == y_R mod p and the parity of x' ==
// the sign bit, or x' == 0: equivalent to "decompress R leniently, compare as points"), Z != 0 or the signature is
// rejected and its Z replaced by 1 in the products (fail closed, and a faulted accumulator cannot spoil its neighbours).
// STRICT's [8]R = O test is made on R's encoded y (ge_is_small_order_affine): an accepted signature has R' == R.
// Per signature: 5 multiplications + 1/16 inversion ~ 45 wave-instructions, against ~ 500 for a square root per signature.
// Workspace rows (stride np = n rounded up to 64): X[10] Y[10] Z[10] flags[1] pre[10].
#define HS_CFIN_M 16          // the most signatures a finish lane takes (its accept bits are a 16-bit mask); the kernels take m <= HS_CFIN_M as an argument
#define HS_CFIN_ROWS 41
HS_HD void cfin_store(u32 *fin, u32 np, u32 i, const ge_p3 &acc, bool ok) {
    fe x = acc.X, y = acc.Y, z = acc.Z;
    fe_carry(x); fe_carry(y); fe_carry(z);
#pragma unroll
    for (int j = 0; j < 10; j++) { fin[(size_t)j * np + i] = x.v[j]; fin[(size_t)(10 + j) * np + i] = y.v[j]; fin[(size_t)(20 + j) * np + i] = z.v[j]; }
    fin[(size_t)30 * np + i] = ok ? 1u : 0u;
}
HS_HD void cfin_load(fe &r, const u32 *fin, u32 np, u32 row0, u32 i) {
#pragma unroll
    for (int j = 0; j < 10; j++) r.v[j] = fin[(size_t)(row0 + j) * np + i];
}
// One finish lane: its signatures are first, first + step, ... (at most HS_CFIN_M of them, those below n).  sigs: the batch's
// packed signatures (R's encoding = the first 8 words of each).  Returns bit j = signature first + j step accepted.
// Control flow is uniform: a lane past its last signature works on signature `first` again (or 0) and drops the result.
// m: signatures per finish lane (1 ... HS_CFIN_M; uniform over the launch).  Fewer = a shorter tail (the lane's own 1,100
// instructions per signature; the inversion's 29 K stay) for more finish wavefronts: same box, 66,667 signatures alone
// 0.337 / 0.305 / 0.290 ms at m = 16 / 8 / 4, four in flight 410 / 414 / 410 M signatures/s; 131,072: 441 / 436 / 430 M.
HS_HD u32 cfin_lane(u32 *fin, u32 np, u32 first, u32 step, u32 n, const u32 *sigs, bool strict, int m = HS_CFIN_M) {
    fe pre = fe_small(1);
#pragma nounroll
    for (int j = 0; j < m; j++) {
        const u32 s = first + (u32)j * step;
        const bool valid = s < n;
        const u32 i = valid ? s : (first < n ? first : 0u);
        fe z; cfin_load(z, fin, np, 20, i);
        const bool use = valid && !fe_iszero(z);
        fe zz; fe_select(zz, z, fe_small(1), !use);
        fe_mul(pre, pre, zz);
        if (valid) {
#pragma unroll
            for (int t = 0; t < 10; t++) fin[(size_t)(31 + t) * np + s] = pre.v[t];
        }
    }
    fe inv; fe_invert(inv, pre);
    u32 bits = 0;
#pragma nounroll
    for (int j = m - 1; j >= 0; j--) {
        const u32 s = first + (u32)j * step;
        const bool valid = s < n;
        const u32 i = valid ? s : (first < n ? first : 0u);
        fe x, y, z, prev;
        cfin_load(z, fin, np, 20, i);
        const bool znz = !fe_iszero(z);
        const bool use = valid && znz;
        if (j > 0 && valid) cfin_load(prev, fin, np, 31, s - step); else prev = fe_small(1);      // valid at j implies valid at j - 1
        fe zinv; fe_mul(zinv, inv, prev);
        fe zz; fe_select(zz, z, fe_small(1), !use);
        fe_mul(inv, inv, zz);
        cfin_load(x, fin, np, 0, i); cfin_load(y, fin, np, 10, i);
        fe_mul(x, x, zinv); fe_mul(y, y, zinv);
        u32 renc[8];
#pragma unroll
        for (int t = 0; t < 8; t++) renc[t] = sigs[(size_t)16 * i + t];
        u32 xw[8], yw[8], ew[8];
        fe_to_words(xw, x); fe_to_words(yw, y);
        fe ye; fe_from_words(ye, renc);
        fe_to_words(ew, ye);
        u32 dy = 0, nz = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) { dy |= yw[t] ^ ew[t]; nz |= xw[t]; }
        bool ok = fin[(size_t)30 * np + i] != 0 && znz && dy == 0 && ((xw[0] & 1) == (renc[7] >> 31) || nz == 0);
        if (strict) ok = ok && !ge_is_small_order_affine(ye);
        if (ok && valid) bits |= 1u << j;
    }
    return bits;
}

// finishing against a decompressed R (one-lane forms): flags of R as in cached_quad_match_point
HS_HD bool cached_lane_match_point(const ge_p3 &acc, const fe &rx, const fe &ry, u32 r_flags, bool strict) {
    bool ok = (r_flags & HS_F_ONCURVE) != 0;
    if (strict) ok = ok && !(r_flags & HS_F_SMALL);
    return ok && ge_match_point(acc, rx, ry);
}

}  // namespace hs
