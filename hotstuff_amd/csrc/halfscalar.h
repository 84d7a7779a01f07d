// halfscalar.h - half-size scalars for the cold verification path (host+device).
//
// dalek's strict / per-signature check is the exact point equation  D := [s]B + [k](-A) - R == O
// (/root/reference/crypto/src/lib.rs:203 -> ed25519-dalek 1.0.1 verify_strict; k = SHA-512(R||A||M) mod L).  Evaluated as
// written it costs 252 point doublings, 56 % of a cold verification, all on one serial chain.  Antipa et al. (SAC 2005)
// and Pornin (2020, "Optimized lattice basis reduction in dimension 2, and fast Schnorr and EdDSA signature
// verification") halve the scalars: take a SHORT vector (u, v) of the lattice
//
//          { (u, v) :  u = v k  (mod 8L) },        |u|, |v| ~ sqrt(8L) ~ 2^128,
//
// and check   [v s mod L]B + [u](-A) + [v](-R) == O   instead: u = v k modulo the ORDER OF THE WHOLE GROUP (8L, not L), so
// [u]A = [v][k]A for every curve point A, torsion component included, and the left side is exactly [v]D.  D has order
// dividing 8L; with v ODD and 0 < v < L,  [v]D = O  <=>  D = O  - the same verdict as dalek's on EVERY input, including
// the mixed-order keys and R's of SURVEY.md 8a class (iii) (tests: all edge and torsion-fuzz vectors, CPU and GPU).
// Cost: ~130 doublings instead of 252 for two more table additions per window; the B part needs no doublings at all
// (the base point's sixteen 16-bit-window tables are absolute).
//
// The short vector comes from a truncated extended Euclid on (8L, k), Lehmer style: the quotient steps run on the leading
// 63 bits (64-bit arithmetic, the quotient from one double-precision division) and are applied to the full numbers as one
// 2x2 matrix with 32-bit entries (multi-word multiply-adds, which this chip does well).  NOTHING here needs to be exact for the verdict to be right: every step is a
// unimodular row operation applied to remainders and cofactors alike, so (a, ta), (b, tb) always stay a basis of the
// lattice (hence ta, tb are never both even); precision only decides how short the result is.  And the result is CHECKED
// before use (half_scalar_ok: v odd, u = v k mod L and mod 8) - a pair that fails falls back to (k, 1), i.e. to the plain
// equation.  The window loop's length follows the scalars it is given, so long pairs (special k such as L - 1, for which
// no short odd-v vector exists) cost time, never correctness.
#pragma once
#include "sha512_sc.h"

namespace hs {

HS_HD int hs_clz32(u32 x) {      // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clz((int)x);
#else
    return __builtin_clz(x);
#endif
}
HS_HD int hs_clz64(u64 x) {      // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
// does any lane of the wavefront want another round?  (on the host a "wavefront" is the one value)
HS_HD bool wave_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ballot(x) != 0ull;
#else
    return x;
#endif
}
// bit length of an N-word little-endian number (0 for 0); lane-uniform select chain
template <int N>
HS_HD int mp_bitlen(const u32 *a) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < N; i++) n = a[i] ? 32 * i + 32 - hs_clz32(a[i]) : n;
    return n;
}
// bits [e, e + 64) of a 256-bit number, e in [0, 255]
HS_HD u64 mp_window64(const u32 a[8], int e) {
    const int wi = e >> 5, bit = e & 31;
    u32 x0 = 0, x1 = 0, x2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x0 = wi == i ? a[i] : x0;
        x1 = wi + 1 == i ? a[i] : x1;
        x2 = wi + 2 == i ? a[i] : x2;
    }
    const u32 lo = (u32)((((u64)x1 << 32) | x0) >> bit), hi = (u32)((((u64)x2 << 32) | x1) >> bit);
    return ((u64)hi << 32) | lo;
}
// r = p x - q y  modulo 2^(32 W)  (two's complement; exact whenever the true value fits W words)
template <int W>
HS_HD void mp_lin2(u32 *r, u32 p, const u32 *x, u32 q, const u32 *y) {
    u64 cp = 0, cq = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < W; i++) {
        cp += (u64)p * x[i];
        cq += (u64)q * y[i];
        const u64 d = (u64)(u32)cp - (u32)cq - borrow;
        r[i] = (u32)d;
        borrow = (d >> 63) & 1;
        cp >>= 32; cq >>= 32;
    }
}
template <int W>
HS_HD void mp_neg_if(u32 *r, bool neg) {       // two's complement negation under a predicate
    const u32 m = neg ? 0xffffffffu : 0u;
    u64 c = neg ? 1 : 0;
#pragma unroll
    for (int i = 0; i < W; i++) { c += (u64)(r[i] ^ m); r[i] = (u32)c; c >>= 32; }
}
template <int W>
HS_HD void mp_addsub(u32 *r, const u32 *x, const u32 *y, bool sub) {      // r = x + y or x - y, modulo 2^(32 W)
    const u32 m = sub ? 0xffffffffu : 0u;
    u64 c = sub ? 1 : 0;
#pragma unroll
    for (int i = 0; i < W; i++) { c += (u64)x[i] + (y[i] ^ m); r[i] = (u32)c; c >>= 32; }
}
template <int W>
HS_HD bool mp_less(const u32 *x, const u32 *y) {      // unsigned x < y
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < W; i++) { const u64 d = (u64)x[i] - y[i] - borrow; borrow = (d >> 63) & 1; }
    return borrow != 0;
}
template <int W>
HS_HD void mp_swap_if(u32 *x, u32 *y, bool sw) {
#pragma unroll
    for (int i = 0; i < W; i++) { const u32 a = x[i], b = y[i]; x[i] = sw ? b : a; y[i] = sw ? a : b; }
}

#define HS_HALF_TARGET_BITS 130      // stop the reduction once the larger remainder has at most this many bits
#define HS_HALF_COF_WORDS 6          // cofactors: two's complement, 192 bits (they stay below ~2^140)
#define HS_8L_WORDS { 0xe7ae9f68u, 0xc09318d2u, 0x17bce6b2u, 0xa6f7cef5u, 0u, 0u, 0u, 0x80000000u }      // 8 L

struct half_pair {
    u32 u[8];        // |u|
    u32 v[8];        // v > 0, odd
    u32 u_neg;       // 1: u < 0
};

// (a, ta), (b, tb): a basis of { (x, t) : x = t k (mod 8L) } with a >= b >= 0 reduced until bitlen(a) <= target
// (or until no further step is possible at this precision - then b is already far shorter than a).
HS_HD void half_reduce(u32 a[9], u32 ta[HS_HALF_COF_WORDS], u32 b[9], u32 tb[HS_HALF_COF_WORDS], const u32 k[8]) {
    const u32 n8[8] = HS_8L_WORDS;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = n8[i]; b[i] = k[i]; }
    a[8] = b[8] = 0;
#pragma unroll
    for (int i = 0; i < HS_HALF_COF_WORDS; i++) { ta[i] = 0; tb[i] = 0; }
    tb[0] = 1;
    bool go = true;
    // both loops run as long as ANY lane of the wavefront still steps (uniform control flow; a finished lane idles through
    // identity steps): typically 5 outer rounds of ~25-35 quotient steps, against the 8 x 44 of the bounds
#pragma nounroll
    for (int outer = 0; outer < 8; outer++) {
        const int n = mp_bitlen<8>(a);
        bool bzero = true;
#pragma unroll
        for (int i = 0; i < 8; i++) bzero = bzero && b[i] == 0;
        go = go && !bzero && n > HS_HALF_TARGET_BITS;
        if (!wave_any(go)) break;
        // the leading 63 bits of a, and the bits of b at the same position
        const int e = n > 63 ? n - 63 : 0;
        u64 ah = mp_window64(a, e), bh = mp_window64(b, e);
        // Euclid steps on (ah, bh): row a = (p, -q), row b = (-r, s) up to a common sign; a step replaces
        // (row a, row b) by (row b, row a - quot row b).  The quotient comes from one double-precision division (both
        // operands below 2^63, the quotient below 2^32: the estimate is off by one at most, and a remainder that comes
        // out negative or >= bh says which way).  Masks instead of selects throughout: a lone wavefront pays 20+ cycles
        // for a v_cndmask on a freshly written condition (profiles/r1_ubench_valu.txt), a handful for and / xor.
        u32 p = 1, q = 0, r = 0, s = 1;
        bool run = go;
        int steps = 0;
#pragma nounroll
        for (int it = 0; it < 40; it++) {
            run = run && bh != 0 && (64 - hs_clz64(ah | 1)) + e > HS_HALF_TARGET_BITS;
            if (!wave_any(run)) break;
            const u64 bsafe = bh | (u64)(bh == 0);
            u64 quot = (u64)((double)ah / (double)bsafe);
            const u64 prod = quot * bsafe;                          // <= ah + bsafe < 2^64
            u64 rem = ah - prod;
            const u64 neg = (u64)((int64_t)rem >> 63);              // estimate one too large
            rem += bsafe & neg; quot += neg;
            const u64 ge = ~(u64)((int64_t)(rem - bsafe) >> 63);    // estimate one too small
            rem -= bsafe & ge; quot -= ge;
            const u64 np = (u64)p + (quot & 0xffffffffu) * r, nq = (u64)q + (quot & 0xffffffffu) * s;
            run = run && ((quot | np | nq) >> 32) == 0;
            const u64 m = run ? ~(u64)0 : (u64)0;
            const u32 m32 = (u32)m;
            const u64 xa = (ah ^ bh) & m, xb = (bh ^ rem) & m;
            ah ^= xa; bh ^= xb;
            const u32 xp = (p ^ r) & m32, xq = (q ^ s) & m32, xr = (r ^ (u32)np) & m32, xs = (s ^ (u32)nq) & m32;
            p ^= xp; q ^= xq; r ^= xr; s ^= xs;
            steps += (int)(m32 & 1u);
        }
        go = go && steps > 0;
        if (!go) { p = 1; q = 0; r = 0; s = 1; }      // identity for the lanes that are done
        // apply the matrix to the remainders and to the cofactors:  a' = +-(p a - q b),  b' = +-(s b - r a)
        u32 na[9], nb[9], nta[HS_HALF_COF_WORDS], ntb[HS_HALF_COF_WORDS];
        mp_lin2<9>(na, p, a, q, b);
        mp_lin2<9>(nb, s, b, r, a);
        mp_lin2<HS_HALF_COF_WORDS>(nta, p, ta, q, tb);
        mp_lin2<HS_HALF_COF_WORDS>(ntb, s, tb, r, ta);
        // remainders are kept non-negative (the rows' common sign alternates with every step, and beyond the precision of the
        // leading parts a difference may come out negative anyway); the cofactor follows its remainder's sign, and the
        // ordering a >= b is re-established by value.
        const bool nega = (na[8] >> 31) != 0, negb = (nb[8] >> 31) != 0;
        mp_neg_if<9>(na, nega); mp_neg_if<HS_HALF_COF_WORDS>(nta, nega);
        mp_neg_if<9>(nb, negb); mp_neg_if<HS_HALF_COF_WORDS>(ntb, negb);
        const bool sw = mp_less<9>(na, nb);
        mp_swap_if<9>(na, nb, sw); mp_swap_if<HS_HALF_COF_WORDS>(nta, ntb, sw);
#pragma unroll
        for (int i = 0; i < 9; i++) { a[i] = na[i]; b[i] = nb[i]; }
#pragma unroll
        for (int i = 0; i < HS_HALF_COF_WORDS; i++) { ta[i] = nta[i]; tb[i] = ntb[i]; }
    }
}

// |x| of a two's complement W-word number -> 8 words, and its sign
template <int W>
HS_HD bool mp_abs8(u32 out[8], const u32 *x) {
    u32 t[W];
#pragma unroll
    for (int i = 0; i < W; i++) t[i] = x[i];
    const bool neg = (t[W - 1] >> 31) != 0;
    mp_neg_if<W>(t, neg);
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = i < W ? t[i] : 0;
    return neg;
}

// The pair must be usable before it is trusted: v odd and 0 < v < L (so gcd(v, 8L) = 1: [v]D = O <=> D = O), |u| < 2^253
// (what the recodings take), and u = v k modulo L and modulo 8 (together: modulo 8L).
// v < L is not implied by the rest: (u, v) = (0, L) satisfies every congruence whenever k = 0 (mod 8) (L is odd, L k = 0 mod L,
// L k = 5 k = 0 mod 8) and would turn the check into [L]R == O - accept for every torsion-free R.  It is where Euclid on
// (8L, k) ends for such k; half_reduce stops 120 bits earlier and its cofactors hold 192 bits, so it never produced it, but this
// predicate is the safety net and must not lean on that.
static_assert(HS_HALF_COF_WORDS <= 7, "cofactors wider than 224 bits could reach L: keep v < L provable");
HS_HD bool half_scalar_ok(const half_pair &h, const u32 k[8]) {
    const u32 Lw[8] = HS_L_WORDS;
    bool ok = (h.v[0] & 1) != 0 && mp_less<8>(h.v, Lw) && (h.u[7] >> 29) == 0;
    u32 vk[8];
    sc_mul(vk, h.v, k);                                  // v k mod L
    // u mod L: |u| < 2^253 < 2 L ... reduce by one conditional subtraction, then negate modulo L when u < 0
    u32 um[8], t[8];
    mp_addsub<8>(t, h.u, Lw, true);
    const bool lt = mp_less<8>(h.u, Lw);
#pragma unroll
    for (int i = 0; i < 8; i++) um[i] = lt ? h.u[i] : t[i];
    bool zero = true;
#pragma unroll
    for (int i = 0; i < 8; i++) zero = zero && um[i] == 0;
    mp_addsub<8>(t, Lw, um, true);                       // L - um
#pragma unroll
    for (int i = 0; i < 8; i++) um[i] = (h.u_neg && !zero) ? t[i] : um[i];
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= um[i] ^ vk[i];
    ok = ok && d == 0;
    const u32 u8 = h.u_neg ? (0u - h.u[0]) & 7u : h.u[0] & 7u;
    ok = ok && (((h.v[0] & 7u) * (k[0] & 7u)) & 7u) == u8;
    return ok;
}

// (u, v) for the scalar k < L:  u = v k (mod 8L), v odd and positive, both as short as the lattice allows (~2^128 for all
// but a negligible fraction of k; up to 2^253 for the few k that have no short odd-v vector).  Never fails: the plain
// pair (k, 1) is the fallback.
HS_HD void half_scalar(half_pair &out, const u32 k[8]) {
    u32 a[9], b[9], ta[HS_HALF_COF_WORDS], tb[HS_HALF_COF_WORDS];
    half_reduce(a, ta, b, tb, k);
    // candidates: b, a, a - b, a + b (with their cofactors); the shortest one whose cofactor is odd
    u32 cu[4][9], ct[4][HS_HALF_COF_WORDS];
#pragma unroll
    for (int i = 0; i < 9; i++) { cu[0][i] = b[i]; cu[1][i] = a[i]; }
#pragma unroll
    for (int i = 0; i < HS_HALF_COF_WORDS; i++) { ct[0][i] = tb[i]; ct[1][i] = ta[i]; }
    mp_addsub<9>(cu[2], a, b, true);  mp_addsub<HS_HALF_COF_WORDS>(ct[2], ta, tb, true);
    mp_addsub<9>(cu[3], a, b, false); mp_addsub<HS_HALF_COF_WORDS>(ct[3], ta, tb, false);
    int best = 1 << 20;
    half_pair h;
#pragma unroll
    for (int i = 0; i < 8; i++) { h.u[i] = k[i]; h.v[i] = 0; }
    h.v[0] = 1; h.u_neg = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        u32 uu[8], vv[8];
        const bool toolong = cu[c][8] != 0;                                  // only a + b can reach the ninth word
        const bool vneg = mp_abs8<HS_HALF_COF_WORDS>(vv, ct[c]);
#pragma unroll
        for (int i = 0; i < 8; i++) uu[i] = cu[c][i];
        const int lu = mp_bitlen<8>(uu), lv = mp_bitlen<8>(vv);
        const int len = lu > lv ? lu : lv;
        const bool take = !toolong && (vv[0] & 1) && len < best;
        best = take ? len : best;
#pragma unroll
        for (int i = 0; i < 8; i++) { h.u[i] = take ? uu[i] : h.u[i]; h.v[i] = take ? vv[i] : h.v[i]; }
        h.u_neg = take ? (vneg ? 1u : 0u) : h.u_neg;                         // remainders are >= 0: u takes -sign(v) after v is made positive
    }
    // the fallback pair, for whatever reason the candidate does not check out
    const bool ok = half_scalar_ok(h, k);
#pragma unroll
    for (int i = 0; i < 8; i++) { out.u[i] = ok ? h.u[i] : k[i]; out.v[i] = ok ? h.v[i] : (i == 0 ? 1u : 0u); }
    out.u_neg = ok ? h.u_neg : 0u;
}

// number of signed radix-16 digits the recoded scalar r = x + 0x88...8 (sc_recode16) really has: the digits above are 0
HS_HD int sc_recode16_len(const u32 r[8]) {
    int n = 1;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 d = r[i] ^ 0x88888888u;
        n = d ? 8 * i + 8 - (hs_clz32(d) >> 2) : n;
    }
    return n;
}

// Signed radix-8 recoding for x < 2^253 (the quad kernels' 3-bit-window form): r = x + 0o44...4 (85 digits, < 2^255);
// digit_i = bits [3i, 3i + 3) of r - 4 in [-4, 3] and sum digit_i 8^i = x.
#define HS_RADIX8_DIGITS 85
#define HS_BIAS8_WORDS { 0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x49249249u, 0x92492492u, 0x24924924u, 0x49249249u }
HS_HD void sc_recode8(u32 r[8], const u32 x[8]) {
    const u32 bias[8] = HS_BIAS8_WORDS;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)x[i] + bias[i] + carry;
        r[i] = (u32)t;
        carry = t >> 32;
    }
}
// digits the radix-8 recoding really has (the ones above are 0)
HS_HD int sc_recode8_len(const u32 r[8]) {
    const u32 bias[8] = HS_BIAS8_WORDS;
    int top = 0;      // position of the highest bit in which r differs from the bias
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u32 d = r[i] ^ bias[i];
        top = d ? 32 * i + 31 - hs_clz32(d) : top;
    }
    return top / 3 + 1;
}
// the recoded scalar is kept left-aligned for popping: digit 84 at bits [253, 256)
HS_HD void sc_align8(u32 r[8]) {
#pragma unroll
    for (int j = 7; j > 0; j--) r[j] = (r[j] << 1) | (r[j - 1] >> 31);
    r[0] <<= 1;
}
// take the top three bits as a signed digit and shift left by 3
HS_HD int sc_pop_digit3(u32 w[8]) {
    const int d = (int)(w[7] >> 29) - 4;
#pragma unroll
    for (int j = 7; j > 0; j--) w[j] = (w[j] << 3) | (w[j - 1] >> 29);
    w[0] <<= 3;
    return d;
}

}  // namespace hs
