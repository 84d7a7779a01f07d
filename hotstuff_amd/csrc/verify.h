// verify.h - the two per-lane stages of one Ed25519 verification (host+device).
//
//   stage 1  prepare_point : one lane per (signature, point) pair; decompress A or R,
//                            small-order flag; the R lane also hashes
//                            k = SHA-512(R||A||M) mod L, range-checks s and recodes both scalars.
//   stage 2  verify_lane   : one lane per signature; [k](-A) + [s]B by a joint
//                            fixed-window double-scalar multiplication (signed radix 16 on -A,
//                            signed radix 65536 on B from the 32,769-entry table of verify_cached.h)
//                            - lane-uniform control flow, table entries chosen per lane -
//                            then the projective comparison with R.
//
// Semantics restated from ed25519-dalek 1.0.1 as reached from the reference:
//   mode HS_MODE_STRICT  = PublicKey::verify_strict   (crypto/src/lib.rs:200-204)
//   mode HS_MODE_BATCH   = the per-signature cofactorless equation that
//                          dalek::verify_batch sums  (crypto/src/lib.rs:206-219);
//                          a batch verdict is the AND over its signatures
//                          (exact wherever dalek is deterministic; on torsion-carrying inputs,
//                          where dalek's answer depends on its random z_i, this per-signature
//                          verdict is returned - DESIGN.md section 3).
#pragma once
#include "ge25519.h"
#include "sha512_sc.h"
#include "halfscalar.h"

namespace hs {

enum : u32 { HS_MODE_BATCH = 0, HS_MODE_STRICT = 1 };
enum : u32 { HS_F_ONCURVE = 1, HS_F_SMALL = 2, HS_F_SCANON = 4 };

// Workspace between the stages, structure-of-arrays with row stride `np` (n padded to
// 64) so that a wavefront's 64 lanes touch 64 consecutive dwords of every row:
//   pt   [2][20][np] u32   x,y limbs of A (slot 0) and R (slot 1)
//   sc   [24][np]    u32   the three scalars of the half-size form (halfscalar.h), recoded: |u| signed radix 16 (rows 0-7),
//                          v signed radix 16 (rows 8-15), w = v s mod L signed radix 65536 (rows 16-23)
//   flag [2][np]     u32   HS_F_* bits per point
//   meta [np]        u32   bits 0-7: windows the signature's u and v need (1 ... 64); bit 8: u is negative
//   tab  [np][2][9][32] u32 stage 2's per-signature tables j * P, j = 0..8, for P = (+-)A and P = -R, cached form
//                          (Y+X, Y-X, Z, 2dT) with every coordinate packed to 255 bits: one 128-byte entry = one cache
//                          line after the other; a lane reads the entry its digit picks with eight 16-byte loads (the
//                          compiler's private-memory layout of the same array interleaves the lanes' dwords, so that every
//                          dword of a lookup touched up to 64 cache lines; round 2's 160-byte entries straddled two lines)
struct workspace {
    u32 *pt, *sc, *flag, *meta, *tab;
    u32 np;
};
#define HS_WS_SC_ROWS 24
#define HS_WS_ENTRY_WORDS 32
#define HS_WS_TAB1_WORDS (9 * HS_WS_ENTRY_WORDS)
#define HS_WS_TAB_WORDS (2 * HS_WS_TAB1_WORDS)
HS_HD size_t workspace_words(size_t np) { return (size_t)np * (2 * 20 + HS_WS_SC_ROWS + 2 + 1 + HS_WS_TAB_WORDS); }
HS_HD workspace workspace_at(u32 *base, size_t np) {
    workspace ws;
    ws.np = (u32)np; ws.pt = base; ws.sc = ws.pt + (size_t)40 * np; ws.flag = ws.sc + (size_t)HS_WS_SC_ROWS * np;
    ws.meta = ws.flag + (size_t)2 * np; ws.tab = ws.meta + np;
    return ws;
}
// A table entry in the workspace: the four coordinates of the cached form packed to 255 bits each, 4 x 32 bytes = ONE
// 128-byte line (ten 32-bit limbs per coordinate are 160 bytes, which straddle two lines: with two tables per signature
// the lookups' lines, not the arithmetic, bounded the one-lane kernel - profiles/r3_pmc_traffic.json).
// fe_pack255: any lazily reduced element (limbs < 2^28, value < 2^256) -> the same residue as an integer below 2^255.
// The limbs are SUMMED at their bit offsets (0, 26, 51, 77, 102, 128, 153, 179, 204, 230) rather than OR-ed, so limb
// slack is absorbed arithmetically, and bit 255 is folded back as 19.
HS_HD void fe_pack255(u32 w[8], const fe &a) {
    u64 acc = (u64)a.v[0] + ((u64)a.v[1] << 26);
    w[0] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[2] << 19;
    w[1] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[3] << 13;
    w[2] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[4] << 6;
    w[3] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[5] + ((u64)a.v[6] << 25);
    w[4] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[7] << 19;
    w[5] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[8] << 12;
    w[6] = (u32)acc; acc >>= 32;
    acc += (u64)a.v[9] << 6;
    HS_CHECK((acc >> 32) == 0, "fe_pack255: value does not fit 256 bits");
    const u32 top = (u32)acc;
    w[7] = top & 0x7fffffffu;
    u64 c = 19u * (top >> 31);
#pragma unroll
    for (int i = 0; i < 8; i++) { c += w[i]; w[i] = (u32)c; c >>= 32; }
    HS_CHECK((w[7] >> 31) == 0, "fe_pack255: fold did not land below 2^255");
}
HS_HD void ws_store_cached(u32 *p, const ge_cached &c) {
    u32 w[32];
    fe_pack255(w, c.YpX); fe_pack255(w + 8, c.YmX); fe_pack255(w + 16, c.Z); fe_pack255(w + 24, c.T2d);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
#else
    for (int i = 0; i < 32; i++) p[i] = w[i];
#endif
}
// the 32 packed words of an entry (eight 16-byte loads), and their unpacking - separate so that a lookup can be REQUESTED a
// window ahead of its use (verify_lane)
HS_HD void ws_fetch_entry(u32 w[32], const u32 *p) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#else
    for (int i = 0; i < 32; i++) w[i] = p[i];
#endif
}
HS_HD void ws_unpack_entry(ge_cached &c, const u32 w[32]) {
    fe_from_words(c.YpX, w); fe_from_words(c.YmX, w + 8); fe_from_words(c.Z, w + 16); fe_from_words(c.T2d, w + 24);
}
HS_HD void ws_load_cached(ge_cached &c, const u32 *p) { u32 w[32]; ws_fetch_entry(w, p); ws_unpack_entry(c, w); }

// ---- stage 1 ------------------------------------------------------------------------------
// task in [0, n): A of signature `task`; task in [n, 2n): R of signature task - n.
// pk / sig / msg point at this signature's 32 / 64 / 32 bytes as little-endian words.
HS_HD void prepare_point(const workspace &ws, u32 i, bool is_r, const u32 *pk, const u32 *sig,
                         const u32 *msg, u32 mode) {
    u32 enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) enc[j] = is_r ? sig[j] : pk[j];
    ge_p3 p;
    u32 fl = ge_decompress(p, enc) ? HS_F_ONCURVE : 0;
    if (mode & HS_MODE_STRICT) fl |= ge_is_small_order_affine(p.Y) ? HS_F_SMALL : 0;
    const u32 slot = is_r ? 1u : 0u;
    u32 *row = ws.pt + (size_t)slot * 20 * ws.np + i;
#pragma unroll
    for (int j = 0; j < 10; j++) { row[(size_t)j * ws.np] = p.X.v[j]; row[(size_t)(10 + j) * ws.np] = p.Y.v[j]; }
    if (is_r) {
        u32 in[24], h[16], k[8], s[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { in[j] = sig[j]; in[8 + j] = pk[j]; in[16 + j] = msg[j]; s[j] = sig[8 + j]; }
        sha512_96(h, in);
        sc_reduce512(k, h);
        fl |= sc_is_canonical(s) ? HS_F_SCANON : 0;
        // the half-size form of  [s]B + [k](-A) == R :   [v s]B + [u](-A) + [v](-R) == O,  u = v k (mod 8L), v odd
        half_pair hp;
        half_scalar(hp, k);
        u32 w[8], ru[8], rv[8], rw[8];
        sc_mul(w, hp.v, s);      // meaningless when s >= L; the lane is rejected by HS_F_SCANON
        sc_recode16(ru, hp.u);
        sc_recode16(rv, hp.v);
        sc_recode65536(rw, w);
        int nw = sc_recode16_len(ru);
        const int nv = sc_recode16_len(rv);
        nw = nw > nv ? nw : nv;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            ws.sc[(size_t)j * ws.np + i] = ru[j]; ws.sc[(size_t)(8 + j) * ws.np + i] = rv[j]; ws.sc[(size_t)(16 + j) * ws.np + i] = rw[j];
        }
        ws.meta[i] = (u32)nw | (hp.u_neg << 8);
    }
    ws.flag[(size_t)slot * ws.np + i] = fl;
}

// ---- stage 2 ------------------------------------------------------------------------------
// take the top nibble of a 256-bit recoded scalar as a signed digit and shift left by 4
HS_HD int sc_pop_digit(u32 w[8]) {
    const int d = (int)(w[7] >> 28) - 8;
#pragma unroll
    for (int j = 7; j > 0; j--) w[j] = (w[j] << 4) | (w[j - 1] >> 28);
    w[0] <<= 4;
    return d;
}

// take the top byte of a 256-bit radix-256 recoded scalar as a signed digit and shift left by 8
HS_HD int sc_pop_digit8(u32 w[8]) {
    const int d = (int)(w[7] >> 24) - 128;
#pragma unroll
    for (int j = 7; j > 0; j--) w[j] = (w[j] << 8) | (w[j - 1] >> 24);
    w[0] <<= 8;
    return d;
}
// take the top half-word of a 256-bit radix-65536 recoded scalar as a signed digit and shift left by 16
HS_HD int sc_pop_digit16(u32 w[8]) {
    const int d = (int)(w[7] >> 16) - 32768;
#pragma unroll
    for (int j = 7; j > 0; j--) w[j] = (w[j] << 16) | (w[j - 1] >> 16);
    w[0] <<= 16;
    return d;
}

// acc == (rx, ry) as points:  X == x_R Z  and  Y == y_R Z  and  Z != 0.
// The complete formulas never produce Z = 0, so the last test changes no verdict; it makes the
// comparison FAIL CLOSED: an all-zero accumulator (what an arithmetic fault produces - it is
// absorbing under these formulas, and 0 == x_R * 0) satisfies both equalities for every signature.
HS_HD bool ge_match_point(ge_p3 acc, const fe &rx, const fe &ry) {
    fe a, b;
    fe_mul(a, rx, acc.Z);
    fe_mul(b, ry, acc.Z);
    fe_carry(acc.X); fe_carry(acc.Y); fe_carry(acc.Z);
    return fe_eq(a, acc.X) && fe_eq(b, acc.Y) && !fe_iszero(acc.Z);
}

// one stored table entry (y-x, y+x, 2dxy; 32 words = one 128-byte line, verify_cached.h) as an affine Niels point
HS_HD void ge_load_niels(ge_niels &n, const u32 *entry) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const uint4 *q = reinterpret_cast<const uint4 *>(entry);
    u32 w[32];
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#else
    const u32 *w = entry;
#endif
#pragma unroll
    for (int i = 0; i < 10; i++) { n.ymx.v[i] = w[i]; n.ypx.v[i] = w[10 + i]; n.xy2d.v[i] = w[20 + i]; }
}
// the largest value of x over the lanes of the wavefront (x itself on the host, where a "wavefront" is one lane)
HS_HD u32 wave_max_u32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const u32 y = (u32)__shfl_xor((int)x, m, 64); x = y > x ? y : x; }
    return (u32)__builtin_amdgcn_readfirstlane((int)x);
#else
    return x;
#endif
}
// j * P, j = 0..8, cached form, into nine consecutive 160-byte entries
HS_HD void ws_build_table(u32 *tab, ge_p3 p) {
    ge_cached a1, c;
    ge_cached_identity(c);
    ws_store_cached(tab, c);
    ge_to_cached(a1, p);
    ws_store_cached(tab + HS_WS_ENTRY_WORDS, a1);
#pragma nounroll
    for (int j = 2; j <= 8; j++) {
        ge_p1p1 t;
        ge_add(t, p, a1, false);
        ge_p1p1_to_p3(p, t, true);
        ge_to_cached(c, p);
        ws_store_cached(tab + HS_WS_ENTRY_WORDS * j, c);
    }
}
// Stage 2 in the half-size form (halfscalar.h):  [w]B + [u](-A) + [v](-R) == O  with |u|, v ~ 2^128 - about 33 windows of
// four doublings instead of 64, two table additions per window (the signature's tables of (+-)A and -R), and the
// sixteen additions of [w]B from the base point's sixteen 16-bit-window tables (b16: T[i][j] = j 2^(16 i) B, 32,769
// entries of 128 bytes per window; absolute multiples, added after the last doubling).
// Control flow is uniform over the wavefront: every lane runs the window count of the lane that needs most.
HS_HD bool verify_lane(const workspace &ws, u32 i, u32 mode, const u32 *b16) {
    const u32 fa = ws.flag[i], fr = ws.flag[ws.np + i];
    bool ok = (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE) && (fr & HS_F_SCANON);
    if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);
    const u32 meta = ws.meta[i];
    const bool u_neg = (meta >> 8) & 1;

    // per-lane tables j * P, j = 0..8, in this signature's slice of the workspace (indexed per lane below):
    // P = -A (or +A when u < 0: the digits are those of |u|) and P = -R
    u32 *atab = ws.tab + (size_t)i * HS_WS_TAB_WORDS, *rtab = atab + HS_WS_TAB1_WORDS;
    {
        const u32 *ra = ws.pt + i, *rr = ws.pt + (size_t)20 * ws.np + i;
        ge_p3 p;
        fe x, nx;
#pragma unroll
        for (int j = 0; j < 10; j++) { x.v[j] = ra[(size_t)j * ws.np]; p.Y.v[j] = ra[(size_t)(10 + j) * ws.np]; }
        fe_neg(nx, x);
        fe_select(p.X, nx, x, u_neg);
        fe_1(p.Z);
        fe_mul(p.T, p.X, p.Y);
        ws_build_table(atab, p);
#pragma unroll
        for (int j = 0; j < 10; j++) { x.v[j] = rr[(size_t)j * ws.np]; p.Y.v[j] = rr[(size_t)(10 + j) * ws.np]; }
        fe_neg(p.X, x);
        fe_1(p.Z);
        fe_mul(p.T, p.X, p.Y);
        ws_build_table(rtab, p);
    }

#ifdef HS_FORCE_NWIN
    const int nwin = HS_FORCE_NWIN;          // timing experiments only (wrong verdicts for scalars that need more windows)
#else
    const int nwin = (int)wave_max_u32(meta & 0xffu);
#endif
    const u32 *scu = ws.sc + i, *scv = ws.sc + (size_t)8 * ws.np + i, *scw = ws.sc + (size_t)16 * ws.np + i;
    ge_p3 acc;
    ge_identity(acc);
#ifdef HS_LANE_PREFETCH      // measured equal (103.3 against 103.3 M verifies/s, 239 against 180 VGPRs): the kernel is VALU-issue bound; kept for A/B
    // The two entries a window adds are requested ONE WINDOW AHEAD (their digits are known from the start) and consumed
    // behind the next four doublings: the lookups go to L2 / MALL, and with two wavefronts per SIMD there is little else to
    // cover them with (SQ_WAIT_ANY was 16 % of the wavefronts' cycles, profiles/r3_pmc_sq.json).
    u32 wa[32], wr[32];
    int du, dv;
    {
        const int w = nwin - 1;
        const u32 sh = 4u * ((u32)w & 7u);
        du = (int)((scu[(size_t)(w >> 3) * ws.np] >> sh) & 15u) - 8;
        dv = (int)((scv[(size_t)(w >> 3) * ws.np] >> sh) & 15u) - 8;
        ws_fetch_entry(wa, atab + HS_WS_ENTRY_WORDS * (du < 0 ? -du : du));
        ws_fetch_entry(wr, rtab + HS_WS_ENTRY_WORDS * (dv < 0 ? -dv : dv));
    }
#pragma nounroll
    for (int w = nwin - 1; w >= 0; w--) {
        ge_p1p1 t;
        const int nd = w == nwin - 1 ? 0 : 4;       // the accumulator is still the identity in the first window
#pragma nounroll
        for (int d = 0; d < nd; d++) {
            ge_dbl(t, acc);
            ge_p1p1_to_p3(acc, t, d == 3);
        }
        ge_cached ea, er;
        ws_unpack_entry(ea, wa);
        ws_unpack_entry(er, wr);
        const bool na = du < 0, nr = dv < 0;
        if (w > 0) {        // next window's digits (one dword of each recoded scalar: wavefront-uniform row, coalesced) and entries
            const u32 sh = 4u * ((u32)(w - 1) & 7u);
            du = (int)((scu[(size_t)((w - 1) >> 3) * ws.np] >> sh) & 15u) - 8;
            dv = (int)((scv[(size_t)((w - 1) >> 3) * ws.np] >> sh) & 15u) - 8;
            ws_fetch_entry(wa, atab + HS_WS_ENTRY_WORDS * (du < 0 ? -du : du));
            ws_fetch_entry(wr, rtab + HS_WS_ENTRY_WORDS * (dv < 0 ? -dv : dv));
        }
        ge_add(t, acc, ea, na);
        ge_p1p1_to_p3(acc, t, true);
        ge_add(t, acc, er, nr);
        ge_p1p1_to_p3(acc, t, w == 0);
    }
#else
#pragma nounroll
    for (int w = nwin - 1; w >= 0; w--) {
        ge_p1p1 t;
        const int nd = w == nwin - 1 ? 0 : 4;       // the accumulator is still the identity in the first window
#pragma nounroll
        for (int d = 0; d < nd; d++) {
            ge_dbl(t, acc);
            ge_p1p1_to_p3(acc, t, d == 3);
        }
        // the digits of this window: one dword of each recoded scalar (wavefront-uniform row, coalesced)
        const u32 sh = 4u * ((u32)w & 7u);
        const int du = (int)((scu[(size_t)(w >> 3) * ws.np] >> sh) & 15u) - 8;
        const int dv = (int)((scv[(size_t)(w >> 3) * ws.np] >> sh) & 15u) - 8;
        ge_cached e;
        ws_load_cached(e, atab + HS_WS_ENTRY_WORDS * (du < 0 ? -du : du));
        ge_add(t, acc, e, du < 0);
        ge_p1p1_to_p3(acc, t, true);
        ws_load_cached(e, rtab + HS_WS_ENTRY_WORDS * (dv < 0 ? -dv : dv));
        ge_add(t, acc, e, dv < 0);
        ge_p1p1_to_p3(acc, t, w == 0);
    }
#endif
    // [w]B: sixteen entries of the base point's tables, T[j][d_j] = d_j 2^(16 j) B - absolute multiples, so they are added
    // once the doublings are over
#pragma nounroll
    for (int j = 0; j < 16; j++) {
        const int dw = (int)((scw[(size_t)(j >> 1) * ws.np] >> (16u * ((u32)j & 1u))) & 0xffffu) - 32768;
        ge_niels eb;
        ge_load_niels(eb, b16 + ((size_t)j * 32769 + (size_t)(dw < 0 ? -dw : dw)) * 32);
        ge_p1p1 t;
        ge_madd(t, acc, eb, dw < 0);
        ge_p1p1_to_p3(acc, t, j < 15);
    }
    // [v]([s]B + [k](-A) - R) == O  <=>  the equation itself (v odd, halfscalar.h).  Z != 0: fail closed on a zeroed accumulator
    fe_carry(acc.X); fe_carry(acc.Y); fe_carry(acc.Z);
    return ok && fe_iszero(acc.X) && fe_eq(acc.Y, acc.Z) && !fe_iszero(acc.Z);
}

// ---- the base point ------------------------------------------------------------------------
HS_HD void ge_basepoint(ge_p3 &b) {
    const u32 bx[10] = FE26_BX, by[10] = FE26_BY, bt[10] = FE26_BT;
    b.X = fe_const(bx); b.Y = fe_const(by); fe_1(b.Z); b.T = fe_const(bt);
}
}  // namespace hs
