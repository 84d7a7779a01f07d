// ge25519.h - Edwards25519 point arithmetic on fe26 limbs (host+device).
//
// Replaces curve25519-dalek's EdwardsPoint / CompressedEdwardsY as used by
// ed25519-dalek 1.0.1 under /root/reference/crypto/src/lib.rs:200-219:
// decompress (lenient, exactly dalek's rules - see oracle/ed25519_ref.py),
// is_small_order, unified addition, doubling.  Formulas: extended twisted
// Edwards, a = -1 (Hisil-Wong-Carter-Dawson 2008); the unified addition is
// complete on this curve, so identity / equal / opposite operands need no
// branches - which is what keeps a 64-lane wavefront convergent.
#pragma once
#include "fe26.h"
#include "curve_consts26.h"

namespace hs {

struct ge_p3 { fe X, Y, Z, T; };
struct ge_p1p1 { fe X, Y, Z, T; };          // "completed": (X:Z), (Y:T)
struct ge_cached { fe YpX, YmX, Z, T2d; };  // projective Niels
struct ge_niels { fe ypx, ymx, xy2d; };     // affine Niels (Z = 1)

HS_HD fe fe_const(const u32 (&c)[10]) { fe r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = c[i];
    return r; }
HS_HD fe fe_d()      { const u32 c[10] = FE26_D; return fe_const(c); }
HS_HD fe fe_d2()     { const u32 c[10] = FE26_D2; return fe_const(c); }
HS_HD fe fe_sqrtm1() { const u32 c[10] = FE26_SQRTM1; return fe_const(c); }

HS_HD void fe_select(fe &r, const fe &a, const fe &b, bool pick_b) {
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
}

HS_HD void ge_identity(ge_p3 &r) { fe_0(r.X); fe_1(r.Y); fe_1(r.Z); fe_0(r.T); }

// completed -> extended.  Operand bounds (fe26.h): from ge_dbl X 3x, Y 1x, Z 1x, T 4x;
// from ge_add/ge_madd X 3x, Y 2x, Z and T 2x or 3x.  T is always the first (<=4x) operand.
// Only X and Z ever stand second: a multiplication pre-scales its SECOND operand by 19 (nine v_mul_lo_u32) and the odd limbs
// of its first by 2, and the compiler shares those between the products of one call - two pre-scaled operands instead of three.
HS_HD void ge_p1p1_to_p3(ge_p3 &r, const ge_p1p1 &p, bool want_t) {
    fe x, y, z;
    fe_mul(x, p.T, p.X);
    fe_mul(y, p.Y, p.Z);
    fe_mul(z, p.T, p.Z);
    if (want_t) fe_mul(r.T, p.Y, p.X);
    r.X = x; r.Y = y; r.Z = z;
}

// r = 2p (T of p unused).  4 squarings.
HS_HD void ge_dbl(ge_p1p1 &r, const ge_p3 &p) {
    fe xx, yy, zz, a, aa;
    fe_sq(xx, p.X);
    fe_sq(yy, p.Y);
    fe_sq(zz, p.Z);
    fe_add(a, p.X, p.Y);
    fe_sq(aa, a);
    fe_add(r.Y, yy, xx); fe_carry(r.Y);      // Y3 = YY + XX          (1x)
    fe_sub(r.Z, yy, xx); fe_carry(r.Z);      // Z3 = YY - XX          (1x)
    fe_sub(r.X, aa, r.Y);                    // X3 = (X+Y)^2 - Y3     (3x)
    fe_add(zz, zz, zz);
    fe_sub(r.T, zz, r.Z);                    // T3 = 2ZZ - Z3         (4x)
}

// r = p + q (q cached), or p - q when neg.  4 multiplications.
HS_HD void ge_add(ge_p1p1 &r, const ge_p3 &p, const ge_cached &q, bool neg) {
    fe qa, qb, a, b, c, d, t;
    fe_select(qa, q.YpX, q.YmX, neg);
    fe_select(qb, q.YmX, q.YpX, neg);
    fe_add(t, p.Y, p.X); fe_mul(a, t, qa);
    fe_sub(t, p.Y, p.X); fe_mul(b, t, qb);
    fe_mul(c, p.T, q.T2d);
    fe_mul(d, p.Z, q.Z); fe_add(d, d, d); fe_carry(d);   // keep Z3/T3 <= 3x for either sign
    fe_sub(r.X, a, b);                       // 3x
    fe_add(r.Y, a, b);                       // 2x
    fe zs, zd;
    fe_add(zs, d, c);                        // 2x
    fe_sub(zd, d, c);                        // 3x
    fe_select(r.Z, zs, zd, neg);
    fe_select(r.T, zd, zs, neg);
}
// mixed addition with an affine Niels point.  3 multiplications.
HS_HD void ge_madd(ge_p1p1 &r, const ge_p3 &p, const ge_niels &q, bool neg) {
    fe qa, qb, a, b, c, d, t;
    fe_select(qa, q.ypx, q.ymx, neg);
    fe_select(qb, q.ymx, q.ypx, neg);
    fe_add(t, p.Y, p.X); fe_mul(a, t, qa);
    fe_sub(t, p.Y, p.X); fe_mul(b, t, qb);
    fe_mul(c, p.T, q.xy2d);
    fe_add(d, p.Z, p.Z); fe_carry(d);
    fe_sub(r.X, a, b);
    fe_add(r.Y, a, b);
    fe zs, zd;
    fe_add(zs, d, c);
    fe_sub(zd, d, c);
    fe_select(r.Z, zs, zd, neg);
    fe_select(r.T, zd, zs, neg);
}

HS_HD void ge_to_cached(ge_cached &c, const ge_p3 &p) {
    fe_add(c.YpX, p.Y, p.X); fe_carry(c.YpX);
    fe_sub(c.YmX, p.Y, p.X); fe_carry(c.YmX);
    c.Z = p.Z;
    fe_mul(c.T2d, p.T, fe_d2());
}
HS_HD void ge_cached_identity(ge_cached &c) { fe_1(c.YpX); fe_1(c.YmX); fe_1(c.Z); fe_0(c.T2d); }

// identity test for a projective point (X : Y : Z): X == 0 and Y == Z
HS_HD bool ge_is_identity(const ge_p3 &p) {
    fe y = p.Y; fe_carry(y);
    fe z = p.Z; fe_carry(z);
    return fe_iszero(p.X) && fe_eq(y, z);
}
// [8]P == O  (dalek EdwardsPoint::is_small_order)
HS_HD bool ge_is_small_order(const ge_p3 &p) {
    ge_p3 q = p;
    ge_p1p1 t;
#pragma nounroll
    for (int i = 0; i < 3; i++) { ge_dbl(t, q); ge_p1p1_to_p3(q, t, false); }
    return ge_is_identity(q);
}

// The same test for an AFFINE point on the curve (what ge_decompress returns: Z = 1), without the three doublings.
// The 8-torsion subgroup of Edwards25519 is cyclic of order 8: (0, 1), (0, -1), (+-sqrt(-1), 0) and the four order-8
// points (+-x8, +-y8) - every sign combination, since adding (0, -1) maps (x, y) to (-x, -y).  So a curve point is of
// small order iff its canonical y is one of 0, 1, -1, y8, -y8 (each of those y has only small-order x's).  About a
// hundred plain instructions instead of 12 squarings + 9 multiplications (2 % of a cold verification); same verdict
// on every input, checked against ge_is_small_order in tests/test_emu_device_math.py.
HS_HD bool ge_is_small_order_affine(const fe &y) {
    u32 w[8];
    fe_to_words(w, y);
    const u32 ya[8] = ED25519_Y8A_WORDS, yb[8] = ED25519_Y8B_WORDS;
    u32 d0 = 0, d1 = w[0] ^ 1u, dm = (w[0] ^ 0xffffffecu) | (w[7] ^ 0x7fffffffu), da = 0, db = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d0 |= w[i]; da |= w[i] ^ ya[i]; db |= w[i] ^ yb[i];
        if (i > 0) d1 |= w[i];
        if (i > 0 && i < 7) dm |= ~w[i];
    }
    return d0 == 0 || d1 == 0 || dm == 0 || da == 0 || db == 0;
}

// CompressedEdwardsY::decompress.  w = the 32 encoding bytes as 8 LE words.
// Lenient as dalek is: y taken mod 2^255 without a canonical check, "negative
// zero" accepted.  Returns false when the point is not on the curve.
// On success r = (x, y, 1, xy).
HS_HD bool ge_decompress(ge_p3 &r, const u32 w[8]) {
    fe y, yy, u, v, v3, x, vxx, one, t;
    fe_from_words(y, w);
    fe_1(one);
    fe_sq(yy, y);
    fe_sub(u, yy, one); fe_carry(u);                 // u = y^2 - 1
    fe_mul(v, yy, fe_d()); fe_add(v, v, one); fe_carry(v);   // v = d y^2 + 1
    fe_sq(v3, v); fe_mul(v3, v3, v);                 // v^3
    fe_sq(x, v3); fe_mul(x, x, v); fe_mul(x, x, u);  // u v^7
    fe_pow22523(x, x);
    fe_mul(x, x, v3); fe_mul(x, x, u);               // u v^3 (u v^7)^((p-5)/8)
    fe_sq(vxx, x); fe_mul(vxx, vxx, v);
    fe_sub(t, vxx, u);
    const bool ok_plain = fe_iszero(t);              // v x^2 == u
    fe_add(t, vxx, u);
    const bool ok_flip = fe_iszero(t);               // v x^2 == -u
    fe xi; fe_mul(xi, x, fe_sqrtm1());
    fe_select(x, x, xi, ok_flip && !ok_plain);
    // dalek: take the non-negative root, then negate iff the sign bit is set
    const bool sign = (w[7] >> 31) != 0;
    fe nx; fe_neg(nx, x);
    fe_select(x, x, nx, fe_isneg(x) != sign);
    r.X = x; r.Y = y; fe_1(r.Z); fe_mul(r.T, x, y);
    return ok_plain || ok_flip;
}

}  // namespace hs
