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

namespace hs {

enum : u32 { HS_MODE_BATCH = 0, HS_MODE_STRICT = 1 };
enum : u32 { HS_F_ONCURVE = 1, HS_F_SMALL = 2, HS_F_SCANON = 4 };

// Workspace between the stages, structure-of-arrays with row stride `np` (n padded to
// 64) so that a wavefront's 64 lanes touch 64 consecutive dwords of every row:
//   pt   [2][20][np] u32   x,y limbs of A (slot 0) and R (slot 1)
//   sc   [16][np]    u32   k recoded signed radix-16 (rows 0-7), s recoded signed radix-256 (rows 8-15)
//   flag [2][np]     u32   HS_F_* bits per point
//   tab  [np][9][40] u32   stage 2's per-signature table j * (-A), j = 0..8, cached form (Y+X, Y-X, Z, 2dT), one
//                          160-byte entry after the other: a lane reads the entry its digit picks with ten 16-byte
//                          loads (the compiler's private-memory layout of the same array interleaves the lanes' dwords,
//                          so that every one of the 40 dwords of a lookup touched up to 64 cache lines)
struct workspace {
    u32 *pt, *sc, *flag, *tab;
    u32 np;
};
#define HS_WS_TAB_WORDS (9 * 40)
HS_HD size_t workspace_words(size_t np) { return (size_t)np * (2 * 20 + 16 + 2 + HS_WS_TAB_WORDS); }
HS_HD workspace workspace_at(u32 *base, size_t np) {
    workspace ws;
    ws.np = (u32)np; ws.pt = base; ws.sc = ws.pt + (size_t)40 * np; ws.flag = ws.sc + (size_t)16 * np; ws.tab = ws.flag + (size_t)2 * np;
    return ws;
}
HS_HD void ws_store_cached(u32 *p, const ge_cached &c) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    uint4 *q = reinterpret_cast<uint4 *>(p);
    u32 w[40];
#pragma unroll
    for (int i = 0; i < 10; i++) { w[i] = c.YpX.v[i]; w[10 + i] = c.YmX.v[i]; w[20 + i] = c.Z.v[i]; w[30 + i] = c.T2d.v[i]; }
#pragma unroll
    for (int i = 0; i < 10; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
#else
    for (int i = 0; i < 10; i++) { p[i] = c.YpX.v[i]; p[10 + i] = c.YmX.v[i]; p[20 + i] = c.Z.v[i]; p[30 + i] = c.T2d.v[i]; }
#endif
}
HS_HD void ws_load_cached(ge_cached &c, const u32 *p) {
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    u32 w[40];
#pragma unroll
    for (int i = 0; i < 10; i++) { const uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#else
    const u32 *w = p;
#endif
#pragma unroll
    for (int i = 0; i < 10; i++) { c.YpX.v[i] = w[i]; c.YmX.v[i] = w[10 + i]; c.Z.v[i] = w[20 + i]; c.T2d.v[i] = w[30 + i]; }
}

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
        u32 in[24], h[16], k[8], s[8], rk[8], rs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { in[j] = sig[j]; in[8 + j] = pk[j]; in[16 + j] = msg[j]; s[j] = sig[8 + j]; }
        sha512_96(h, in);
        sc_reduce512(k, h);
        fl |= sc_is_canonical(s) ? HS_F_SCANON : 0;
        sc_recode16(rk, k);
        sc_recode256(rs, s);     // meaningless when s >= L; the lane is rejected by HS_F_SCANON
#pragma unroll
        for (int j = 0; j < 8; j++) { ws.sc[(size_t)j * ws.np + i] = rk[j]; ws.sc[(size_t)(8 + j) * ws.np + i] = rs[j]; }
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
// b16: window 0 of the base point's 16-bit table (verify_cached.h): entry j = j B, j <= 32,768, 128 bytes each,
// L2 / MALL-resident - one addition per 16 doublings for [s]B.
HS_HD bool verify_lane(const workspace &ws, u32 i, u32 mode, const u32 *b16) {
    const u32 fa = ws.flag[i], fr = ws.flag[ws.np + i];
    bool ok = (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE) && (fr & HS_F_SCANON);
    if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);

    ge_p3 p;                       // -A
    fe rx, ry;
    {
        const u32 *ra = ws.pt + i, *rr = ws.pt + (size_t)20 * ws.np + i;
        fe ax;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            ax.v[j] = ra[(size_t)j * ws.np]; p.Y.v[j] = ra[(size_t)(10 + j) * ws.np];
            rx.v[j] = rr[(size_t)j * ws.np]; ry.v[j] = rr[(size_t)(10 + j) * ws.np];
        }
        fe_neg(p.X, ax);
        fe_1(p.Z);
        fe_mul(p.T, p.X, p.Y);
    }
    u32 kw[8], sw[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { kw[j] = ws.sc[(size_t)j * ws.np + i]; sw[j] = ws.sc[(size_t)(8 + j) * ws.np + i]; }
    {   // stage 1 stored s + 0x80..80 (radix 256); 16-bit windows want s + 0x8000..8000: subtract 0x0080..0080
        u64 borrow = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) { const u64 t = (u64)sw[j] - 0x00800080u - borrow; sw[j] = (u32)t; borrow = (t >> 32) & 1; }
    }

    // per-lane table j * (-A), j = 0..8, in this signature's slice of the workspace (indexed per lane below)
    u32 *atab = ws.tab + (size_t)i * HS_WS_TAB_WORDS;
    {
        ge_cached a1, c;
        ge_cached_identity(c);
        ws_store_cached(atab, c);
        ge_to_cached(a1, p);
        ws_store_cached(atab + 40, a1);
#pragma nounroll
        for (int j = 2; j <= 8; j++) {
            ge_p1p1 t;
            ge_add(t, p, a1, false);
            ge_p1p1_to_p3(p, t, true);
            ge_to_cached(c, p);
            ws_store_cached(atab + 40 * j, c);
        }
    }

    ge_p3 acc;
    ge_identity(acc);
#pragma nounroll
    for (int w = 0; w < 64; w++) {
        ge_p1p1 t;
        const int nd = w ? 4 : 0;       // the accumulator is still the identity in the first window
#pragma nounroll
        for (int d = 0; d < nd; d++) {
            ge_dbl(t, acc);
            ge_p1p1_to_p3(acc, t, d == 3);
        }
        const int dk = sc_pop_digit(kw);
        const int ik = dk < 0 ? -dk : dk;
        ge_cached ea;
        ws_load_cached(ea, atab + 40 * ik);
        ge_add(t, acc, ea, dk < 0);
        if ((w & 3) == 3) {     // [s]B: one 16-bit window per four 4-bit windows of k
            ge_p1p1_to_p3(acc, t, true);
            const int ds = sc_pop_digit16(sw);
            ge_niels eb;
            ge_load_niels(eb, b16 + (size_t)(ds < 0 ? -ds : ds) * 32);
            ge_madd(t, acc, eb, ds < 0);
        }
        ge_p1p1_to_p3(acc, t, false);
    }
    return ok && ge_match_point(acc, rx, ry);
}

// ---- base-point table (computed once on the host at init) ----------------------------------
HS_HD void ge_p3_to_niels(ge_niels &n, const ge_p3 &p) {
    fe zi, x, y, xy;
    fe_invert(zi, p.Z);
    fe_mul(x, p.X, zi); fe_mul(y, p.Y, zi);
    fe_add(n.ypx, y, x); fe_carry(n.ypx);
    fe_sub(n.ymx, y, x); fe_carry(n.ymx);
    fe_mul(xy, x, y); fe_mul(n.xy2d, xy, fe_d2());
}
HS_HD void ge_basepoint(ge_p3 &b) {
    const u32 bx[10] = FE26_BX, by[10] = FE26_BY, bt[10] = FE26_BT;
    b.X = fe_const(bx); b.Y = fe_const(by); fe_1(b.Z); b.T = fe_const(bt);
}
}  // namespace hs
