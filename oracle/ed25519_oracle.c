/* oracle/ed25519_oracle.c - CPU restatement of the Ed25519 verification path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (hotstuff_amd/, libhs_ed25519.so) never links, imports or calls it.
 *
 * The reference's arithmetic is in the un-vendored crate ed25519-dalek 1.0.1
 * (crypto/Cargo.toml:10) -> curve25519-dalek 3.x (u64 backend: 5x51-bit limbs,
 * u128 products), sha2 0.9.  This file restates that published algorithm in
 * plain C with the same shape (51-bit limbs, unsigned __int128; Straus with
 * width-5 NAF below 190 points, Pippenger at or above), anchored on the
 * reference's call sites:
 *   crypto/src/lib.rs:185-191  Signature::new          -> hs_oracle_sign
 *   crypto/src/lib.rs:200-204  Signature::verify       -> hs_oracle_verify_strict
 *   crypto/src/lib.rs:206-219  Signature::verify_batch -> hs_oracle_verify_batch
 *   consensus/src/messages.rs:197 (QC call site), 307-313 (TC::verify's loop) -> hs_oracle_verify_tc
 * Parity status: pinned on RFC 8032 vectors + OpenSSL-generated fixtures
 * (tests/golden) and on oracle/ed25519_ref.py (big-int); the edge classes are
 * "parity unpinned" by the reference itself - it has no vectors.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include "sha512_consts.h"
#include "curve_consts51.h"

typedef unsigned __int128 u128;
typedef uint64_t fe[5];
#define MASK51 ((1ULL << 51) - 1)

/* ------------------------------------------------------------------ SHA-512 */
static const uint64_t K512[80] = SHA512_K_INIT;
static const uint64_t H512[8] = SHA512_H0_INIT;
typedef struct { uint64_t h[8]; uint8_t buf[128]; size_t fill; uint64_t total; } sha512_ctx;

static inline uint64_t ror64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
static inline uint64_t be64(const uint8_t *p) {
    uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | p[i]; return v;
}
static void sha512_block(uint64_t h[8], const uint8_t *blk) {
    uint64_t w[80], a[8];
    for (int i = 0; i < 16; i++) w[i] = be64(blk + 8 * i);
    for (int i = 16; i < 80; i++) {
        uint64_t s0 = ror64(w[i-15], 1) ^ ror64(w[i-15], 8) ^ (w[i-15] >> 7);
        uint64_t s1 = ror64(w[i-2], 19) ^ ror64(w[i-2], 61) ^ (w[i-2] >> 6);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    memcpy(a, h, sizeof a);
    for (int i = 0; i < 80; i++) {
        uint64_t S1 = ror64(a[4], 14) ^ ror64(a[4], 18) ^ ror64(a[4], 41);
        uint64_t ch = (a[4] & a[5]) ^ (~a[4] & a[6]);
        uint64_t t1 = a[7] + S1 + ch + K512[i] + w[i];
        uint64_t S0 = ror64(a[0], 28) ^ ror64(a[0], 34) ^ ror64(a[0], 39);
        uint64_t mj = (a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2]);
        uint64_t t2 = S0 + mj;
        a[7] = a[6]; a[6] = a[5]; a[5] = a[4]; a[4] = a[3] + t1;
        a[3] = a[2]; a[2] = a[1]; a[1] = a[0]; a[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += a[i];
}
static void sha512_init(sha512_ctx *c) { memcpy(c->h, H512, sizeof H512); c->fill = 0; c->total = 0; }
static void sha512_update(sha512_ctx *c, const uint8_t *p, size_t n) {
    c->total += n;
    while (n) {
        size_t k = 128 - c->fill; if (k > n) k = n;
        memcpy(c->buf + c->fill, p, k); c->fill += k; p += k; n -= k;
        if (c->fill == 128) { sha512_block(c->h, c->buf); c->fill = 0; }
    }
}
static void sha512_final(sha512_ctx *c, uint8_t out[64]) {
    uint64_t bits = c->total * 8;
    uint8_t pad = 0x80;
    sha512_update(c, &pad, 1);
    pad = 0;
    while (c->fill != 112) sha512_update(c, &pad, 1);
    uint8_t len[16] = {0};
    for (int i = 0; i < 8; i++) len[15 - i] = (uint8_t)(bits >> (8 * i));
    sha512_update(c, len, 16);
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) out[8*i + j] = (uint8_t)(c->h[i] >> (56 - 8*j));
}
void hs_oracle_sha512(const uint8_t *m, size_t n, uint8_t out[64]) {
    sha512_ctx c; sha512_init(&c); sha512_update(&c, m, n); sha512_final(&c, out);
}

/* ------------------------------------------------------------------ GF(2^255-19), 5x51 */
static const fe FE_D = FE51_D, FE_D2 = FE51_D2, FE_SQRTM1 = FE51_SQRTM1;
static inline void fe_copy(fe r, const fe a) { memcpy(r, a, sizeof(fe)); }
static inline void fe_0(fe r) { memset(r, 0, sizeof(fe)); }
static inline void fe_1(fe r) { fe_0(r); r[0] = 1; }
static inline uint64_t le64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

static void fe_frombytes(fe r, const uint8_t s[32]) {   /* bit 255 ignored, no canonical check */
    r[0] = le64(s) & MASK51;
    r[1] = (le64(s + 6) >> 3) & MASK51;
    r[2] = (le64(s + 12) >> 6) & MASK51;
    r[3] = (le64(s + 19) >> 1) & MASK51;
    r[4] = (le64(s + 24) >> 12) & MASK51;
}
static inline void fe_carry(fe r) {
    uint64_t c;
    c = r[0] >> 51; r[0] &= MASK51; r[1] += c;
    c = r[1] >> 51; r[1] &= MASK51; r[2] += c;
    c = r[2] >> 51; r[2] &= MASK51; r[3] += c;
    c = r[3] >> 51; r[3] &= MASK51; r[4] += c;
    c = r[4] >> 51; r[4] &= MASK51; r[0] += 19 * c;
    c = r[0] >> 51; r[0] &= MASK51; r[1] += c;
}
static void fe_tobytes(uint8_t s[32], const fe a) {     /* canonical encoding */
    fe t; fe_copy(t, a); fe_carry(t); fe_carry(t);
    /* t < 2^255 + small; subtract p if t >= p */
    uint64_t q = (t[0] + 19) >> 51;
    q = (t[1] + q) >> 51; q = (t[2] + q) >> 51; q = (t[3] + q) >> 51; q = (t[4] + q) >> 51;
    t[0] += 19 * q;
    uint64_t c;
    c = t[0] >> 51; t[0] &= MASK51; t[1] += c;
    c = t[1] >> 51; t[1] &= MASK51; t[2] += c;
    c = t[2] >> 51; t[2] &= MASK51; t[3] += c;
    c = t[3] >> 51; t[3] &= MASK51; t[4] += c;
    t[4] &= MASK51;
    uint64_t w0 = t[0] | (t[1] << 51);
    uint64_t w1 = (t[1] >> 13) | (t[2] << 38);
    uint64_t w2 = (t[2] >> 26) | (t[3] << 25);
    uint64_t w3 = (t[3] >> 39) | (t[4] << 12);
    memcpy(s, &w0, 8); memcpy(s + 8, &w1, 8); memcpy(s + 16, &w2, 8); memcpy(s + 24, &w3, 8);
}
static inline void fe_add(fe r, const fe a, const fe b) { for (int i = 0; i < 5; i++) r[i] = a[i] + b[i]; }
static inline void fe_sub(fe r, const fe a, const fe b) {   /* a + 4p - b, then carry */
    r[0] = a[0] + 0x1fffffffffffb4ULL - b[0];
    for (int i = 1; i < 5; i++) r[i] = a[i] + 0x1ffffffffffffcULL - b[i];
    fe_carry(r);
}
static inline void fe_neg(fe r, const fe a) { fe z; fe_0(z); fe_sub(r, z, a); }
static void fe_mul(fe r, const fe a, const fe b) {
    uint64_t b1 = 19 * b[1], b2 = 19 * b[2], b3 = 19 * b[3], b4 = 19 * b[4];
    u128 t0 = (u128)a[0]*b[0] + (u128)a[1]*b4 + (u128)a[2]*b3 + (u128)a[3]*b2 + (u128)a[4]*b1;
    u128 t1 = (u128)a[0]*b[1] + (u128)a[1]*b[0] + (u128)a[2]*b4 + (u128)a[3]*b3 + (u128)a[4]*b2;
    u128 t2 = (u128)a[0]*b[2] + (u128)a[1]*b[1] + (u128)a[2]*b[0] + (u128)a[3]*b4 + (u128)a[4]*b3;
    u128 t3 = (u128)a[0]*b[3] + (u128)a[1]*b[2] + (u128)a[2]*b[1] + (u128)a[3]*b[0] + (u128)a[4]*b4;
    u128 t4 = (u128)a[0]*b[4] + (u128)a[1]*b[3] + (u128)a[2]*b[2] + (u128)a[3]*b[1] + (u128)a[4]*b[0];
    t1 += (uint64_t)(t0 >> 51); r[0] = (uint64_t)t0 & MASK51;
    t2 += (uint64_t)(t1 >> 51); r[1] = (uint64_t)t1 & MASK51;
    t3 += (uint64_t)(t2 >> 51); r[2] = (uint64_t)t2 & MASK51;
    t4 += (uint64_t)(t3 >> 51); r[3] = (uint64_t)t3 & MASK51;
    uint64_t c = (uint64_t)(t4 >> 51); r[4] = (uint64_t)t4 & MASK51;
    r[0] += 19 * c; c = r[0] >> 51; r[0] &= MASK51; r[1] += c;
}
static void fe_sq(fe r, const fe a) {
    uint64_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
    uint64_t d0 = 2 * a0, d1 = 2 * a1, d2 = 2 * a2, a3_19 = 19 * a3, a4_19 = 19 * a4;
    u128 t0 = (u128)a0*a0 + (u128)d1*a4_19 + (u128)d2*a3_19;
    u128 t1 = (u128)d0*a1 + (u128)d2*a4_19 + (u128)a3*a3_19;
    u128 t2 = (u128)d0*a2 + (u128)a1*a1 + (u128)(2*a3)*a4_19;
    u128 t3 = (u128)d0*a3 + (u128)d1*a2 + (u128)a4*a4_19;
    u128 t4 = (u128)d0*a4 + (u128)d1*a3 + (u128)a2*a2;
    t1 += (uint64_t)(t0 >> 51); r[0] = (uint64_t)t0 & MASK51;
    t2 += (uint64_t)(t1 >> 51); r[1] = (uint64_t)t1 & MASK51;
    t3 += (uint64_t)(t2 >> 51); r[2] = (uint64_t)t2 & MASK51;
    t4 += (uint64_t)(t3 >> 51); r[3] = (uint64_t)t3 & MASK51;
    uint64_t c = (uint64_t)(t4 >> 51); r[4] = (uint64_t)t4 & MASK51;
    r[0] += 19 * c; c = r[0] >> 51; r[0] &= MASK51; r[1] += c;
}
static void fe_sqn(fe r, const fe a, int n) { fe_sq(r, a); for (int i = 1; i < n; i++) fe_sq(r, r); }
/* z^(2^250-1) ladder shared by invert and pow22523 */
static void fe_pow250(fe r, fe z11, const fe z) {
    fe z2, z9, t, z5, z10, z20, z50, z100;
    fe_sq(z2, z); fe_sqn(t, z2, 2); fe_mul(z9, t, z); fe_mul(z11, z9, z2);
    fe_sq(t, z11); fe_mul(z5, t, z9);                 /* 2^5 - 1 */
    fe_sqn(t, z5, 5); fe_mul(z10, t, z5);             /* 2^10 - 1 */
    fe_sqn(t, z10, 10); fe_mul(z20, t, z10);
    fe_sqn(t, z20, 20); fe_mul(t, t, z20);            /* 2^40 - 1 */
    fe_sqn(t, t, 10); fe_mul(z50, t, z10);
    fe_sqn(t, z50, 50); fe_mul(z100, t, z50);
    fe_sqn(t, z100, 100); fe_mul(t, t, z100);         /* 2^200 - 1 */
    fe_sqn(t, t, 50); fe_mul(r, t, z50);              /* 2^250 - 1 */
}
static void fe_invert(fe r, const fe z) { fe t, z11; fe_pow250(t, z11, z); fe_sqn(t, t, 5); fe_mul(r, t, z11); }
static void fe_pow22523(fe r, const fe z) { fe t, z11; fe_pow250(t, z11, z); fe_sqn(t, t, 2); fe_mul(r, t, z); }
static int fe_iszero(const fe a) { uint8_t s[32]; fe_tobytes(s, a); uint8_t d = 0; for (int i = 0; i < 32; i++) d |= s[i]; return d == 0; }
static int fe_isneg(const fe a) { uint8_t s[32]; fe_tobytes(s, a); return s[0] & 1; }
static int fe_eq(const fe a, const fe b) { fe t; fe_sub(t, a, b); return fe_iszero(t); }

/* ------------------------------------------------------------------ points */
typedef struct { fe X, Y, Z, T; } ge;                  /* extended */
typedef struct { fe YpX, YmX, Z, T2d; } ge_cached;     /* "ProjectiveNiels" */
static const ge GE_B = { FE51_BX, FE51_BY, {1, 0, 0, 0, 0}, FE51_BT };

static void ge_identity(ge *r) { fe_0(r->X); fe_1(r->Y); fe_1(r->Z); fe_0(r->T); }
static void ge_to_cached(ge_cached *c, const ge *p) {
    fe_add(c->YpX, p->Y, p->X); fe_carry(c->YpX); fe_sub(c->YmX, p->Y, p->X);
    fe_copy(c->Z, p->Z); fe_mul(c->T2d, p->T, FE_D2);
}
static void ge_add_cached(ge *r, const ge *p, const ge_cached *q, int negate) {
    fe a, b, c, d, e, f, g, h;
    fe_add(a, p->Y, p->X); fe_sub(b, p->Y, p->X);
    if (!negate) { fe_mul(a, a, q->YpX); fe_mul(b, b, q->YmX); }
    else         { fe_mul(a, a, q->YmX); fe_mul(b, b, q->YpX); }
    fe_mul(c, p->T, q->T2d); fe_mul(d, p->Z, q->Z); fe_add(d, d, d);
    fe_sub(e, a, b); fe_add(h, a, b);
    if (!negate) { fe_sub(f, d, c); fe_add(g, d, c); } else { fe_add(f, d, c); fe_sub(g, d, c); }
    fe_carry(g); fe_carry(h); fe_carry(f);
    fe_mul(r->X, e, f); fe_mul(r->Y, g, h); fe_mul(r->Z, f, g); fe_mul(r->T, e, h);
}
static void ge_add(ge *r, const ge *p, const ge *q) { ge_cached c; ge_to_cached(&c, q); ge_add_cached(r, p, &c, 0); }
static void ge_dbl(ge *r, const ge *p) {
    fe a, b, c, e, g, f, h, t;
    fe_sq(a, p->X); fe_sq(b, p->Y); fe_sq(c, p->Z); fe_add(c, c, c);
    fe_add(t, p->X, p->Y); fe_sq(t, t);
    fe_add(h, a, b); fe_carry(h);          /* H' = A + B (sign folded below) */
    fe_sub(e, h, t);                       /* -E */
    fe_sub(g, a, b);                       /* -G */
    fe_add(f, c, g); fe_carry(f);          /* -F = C - G ... with G = B - A */
    /* (-E)(-F), (-G)(-H'), (-F)(-G), (-E)(-H'): all sign pairs cancel */
    fe_mul(r->X, e, f); fe_mul(r->Y, g, h); fe_mul(r->Z, f, g); fe_mul(r->T, e, h);
}
static void ge_neg(ge *r, const ge *p) { fe_neg(r->X, p->X); fe_copy(r->Y, p->Y); fe_copy(r->Z, p->Z); fe_neg(r->T, p->T); }
static int ge_eq(const ge *p, const ge *q) {
    fe a, b; fe_mul(a, p->X, q->Z); fe_mul(b, q->X, p->Z); if (!fe_eq(a, b)) return 0;
    fe_mul(a, p->Y, q->Z); fe_mul(b, q->Y, p->Z); return fe_eq(a, b);
}
static int ge_is_identity(const ge *p) { return fe_iszero(p->X) && fe_eq(p->Y, p->Z); }
static int ge_is_small_order(const ge *p) { ge t; ge_dbl(&t, p); ge_dbl(&t, &t); ge_dbl(&t, &t); return ge_is_identity(&t); }
/* CompressedEdwardsY::decompress - lenient exactly as documented in ed25519_ref.py */
static int ge_decompress(ge *r, const uint8_t s[32]) {
    fe u, v, v3, x, vxx, chk;
    fe_frombytes(r->Y, s); fe_1(r->Z);
    fe_sq(u, r->Y); fe_mul(v, u, FE_D);
    fe one; fe_1(one); fe_sub(u, u, one); fe_add(v, v, one); fe_carry(v);
    fe_sq(v3, v); fe_mul(v3, v3, v);                   /* v^3 */
    fe_sq(x, v3); fe_mul(x, x, v); fe_mul(x, x, u);    /* u v^7 */
    fe_pow22523(x, x); fe_mul(x, x, v3); fe_mul(x, x, u);
    fe_sq(vxx, x); fe_mul(vxx, vxx, v);
    fe_sub(chk, vxx, u);
    if (!fe_iszero(chk)) {
        fe_add(chk, vxx, u);
        if (!fe_iszero(chk)) return 0;
        fe_mul(x, x, FE_SQRTM1);
    }
    if (fe_isneg(x) != (s[31] >> 7)) fe_neg(x, x);
    fe_copy(r->X, x); fe_mul(r->T, r->X, r->Y);
    return 1;
}
static void ge_compress(uint8_t s[32], const ge *p) {
    fe zi, x, y; fe_invert(zi, p->Z); fe_mul(x, p->X, zi); fe_mul(y, p->Y, zi);
    fe_tobytes(s, y); s[31] ^= (uint8_t)(fe_isneg(x) << 7);
}

/* ------------------------------------------------------------------ scalars mod L (4x64) */
static const uint64_t SC_L[4] = { 0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL };
static const uint64_t SC_C[2] = { 0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL };   /* L - 2^252 */
typedef struct { uint64_t v[4]; } sc;

static int sc_geq_L(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > SC_L[i]) return 1; if (a[i] < SC_L[i]) return 0; }
    return 1;
}
/* r = x mod L for x of nl 64-bit limbs (nl <= 8) via 2^252 = -c folding */
static void sc_reduce_limbs(sc *r, const uint64_t *x, int nl) {
    uint64_t cur[9]; int n = nl; memcpy(cur, x, 8 * nl);
    /* acc = sum (+/-) lo parts, signed 5-limb accumulator */
    __int128 acc[5] = {0, 0, 0, 0, 0};
    int sign = 1;
    for (;;) {
        uint64_t lo[4] = {0, 0, 0, 0}, hi[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4 && i < n; i++) lo[i] = cur[i];
        lo[3] &= 0x0fffffffffffffffULL;
        int nh = 0;
        for (int i = 3; i < n; i++) {            /* hi = cur >> 252 */
            uint64_t w = cur[i] >> 60; if (i + 1 < n) w |= cur[i + 1] << 4;
            hi[nh++] = w;
        }
        for (int i = 0; i < 4; i++) acc[i] += sign * (__int128)lo[i];
        int z = 1; for (int i = 0; i < nh; i++) if (hi[i]) z = 0;
        if (z) break;
        /* cur = c * hi */
        uint64_t prod[9] = {0};
        for (int i = 0; i < nh; i++) {
            u128 carry = 0;
            for (int j = 0; j < 2; j++) {
                u128 t = (u128)hi[i] * SC_C[j] + prod[i + j] + carry;
                prod[i + j] = (uint64_t)t; carry = t >> 64;
            }
            int k = i + 2;
            while (carry) { u128 t = (u128)prod[k] + carry; prod[k] = (uint64_t)t; carry = t >> 64; k++; }
        }
        n = nh + 2; memcpy(cur, prod, 8 * n);
        while (n > 1 && cur[n - 1] == 0) n--;
        sign = -sign;
    }
    /* acc in (-2^254, 2^254): add 4L then normalise, then subtract L while >= L */
    for (int i = 0; i < 4; i++) acc[i] += (__int128)4 * SC_L[i];
    uint64_t out[5]; __int128 carry = 0;
    for (int i = 0; i < 5; i++) { __int128 t = acc[i] + carry; out[i] = (uint64_t)t; carry = t >> 64; }
    for (int it = 0; it < 8; it++) {
        if (out[4] == 0 && !sc_geq_L(out)) break;
        unsigned __int128 b = 0;
        for (int i = 0; i < 5; i++) {
            uint64_t li = i < 4 ? SC_L[i] : 0;
            u128 t = (u128)out[i] - li - (uint64_t)b; out[i] = (uint64_t)t; b = (t >> 64) & 1;
        }
    }
    memcpy(r->v, out, 32);
}
static void sc_from_bytes64(sc *r, const uint8_t h[64]) { uint64_t x[8]; memcpy(x, h, 64); sc_reduce_limbs(r, x, 8); }
static void sc_from_bytes32(sc *r, const uint8_t s[32]) { memcpy(r->v, s, 32); }   /* caller checked < L */
static void sc_mul(sc *r, const sc *a, const sc *b) {
    uint64_t p[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 carry = 0;
        for (int j = 0; j < 4; j++) { u128 t = (u128)a->v[i] * b->v[j] + p[i + j] + carry; p[i + j] = (uint64_t)t; carry = t >> 64; }
        p[i + 4] = (uint64_t)carry;
    }
    sc_reduce_limbs(r, p, 8);
}
static void sc_add(sc *r, const sc *a, const sc *b) {
    uint64_t t[5]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t[i] = (uint64_t)c; c >>= 64; }
    t[4] = (uint64_t)c; sc_reduce_limbs(r, t, 5);
}
static void sc_neg(sc *r, const sc *a) {
    int z = !(a->v[0] | a->v[1] | a->v[2] | a->v[3]);
    if (z) { *r = *a; return; }
    u128 b = 0;
    for (int i = 0; i < 4; i++) { u128 t = (u128)SC_L[i] - a->v[i] - (uint64_t)b; r->v[i] = (uint64_t)t; b = (t >> 64) & 1; }
}
static int sc_is_canonical(const uint8_t s[32]) { uint64_t v[4]; memcpy(v, s, 32); return !sc_geq_L(v); }

/* ------------------------------------------------------------------ scalar mults */
/* width-w non-adjacent form, digits in (-2^(w-1), 2^(w-1)), odd or zero (dalek Scalar::non_adjacent_form) */
static void sc_naf(int8_t naf[256], const sc *k, int w) {
    uint64_t x[5]; memcpy(x, k->v, 32); x[4] = 0;
    memset(naf, 0, 256);
    int width = 1 << w, pos = 0, carry = 0;
    while (pos < 256) {
        int idx = pos / 64, bit = pos % 64;
        uint64_t buf = bit < 64 - w ? x[idx] >> bit : (x[idx] >> bit) | (x[idx + 1] << (64 - bit));
        int window = carry + (int)(buf & (uint64_t)(width - 1));
        if ((window & 1) == 0) { pos += 1; continue; }
        if (window < width / 2) { carry = 0; naf[pos] = (int8_t)window; }
        else { carry = 1; naf[pos] = (int8_t)(window - width); }
        pos += w;
    }
}
static void ge_odd_table(ge_cached tab[8], const ge *p) {   /* [1,3,5,...,15]P */
    ge p2, cur = *p; ge_dbl(&p2, p); ge_cached c2; ge_to_cached(&c2, &p2);
    ge_to_cached(&tab[0], &cur);
    for (int i = 1; i < 8; i++) { ge t; ge_add_cached(&t, &cur, &c2, 0); cur = t; ge_to_cached(&tab[i], &cur); }
}
/* Straus with NAF-5: sum k_i P_i  (dalek straus.rs vartime path) */
static void ge_straus(ge *r, const sc *k, const ge *p, size_t n) {
    int8_t (*naf)[256] = malloc(n * 256);
    ge_cached (*tab)[8] = malloc(n * sizeof(ge_cached[8]));
    for (size_t i = 0; i < n; i++) { sc_naf(naf[i], &k[i], 5); ge_odd_table(tab[i], &p[i]); }
    ge acc; ge_identity(&acc);
    for (int b = 255; b >= 0; b--) {
        ge t; ge_dbl(&t, &acc); acc = t;
        for (size_t i = 0; i < n; i++) {
            int d = naf[i][b];
            if (d > 0) { ge_add_cached(&t, &acc, &tab[i][d / 2], 0); acc = t; }
            else if (d < 0) { ge_add_cached(&t, &acc, &tab[i][(-d) / 2], 1); acc = t; }
        }
    }
    *r = acc; free(naf); free(tab);
}
/* signed radix-2^w digits (dalek Scalar::to_radix_2w), count = ceil(256/w) (+1 when w divides) */
static int sc_radix2w(int8_t *dig, const sc *k, int w) {
    int count = (256 + w - 1) / w; int radix = 1 << w; int carry = 0;
    uint64_t x[5]; memcpy(x, k->v, 32); x[4] = 0;
    for (int i = 0; i < count; i++) {
        int pos = i * w, idx = pos / 64, bit = pos % 64;
        uint64_t buf = bit < 64 - w ? x[idx] >> bit : (x[idx] >> bit) | (bit ? x[idx + 1] << (64 - bit) : 0);
        int coef = carry + (int)(buf & (uint64_t)(radix - 1));
        carry = (coef + radix / 2) >> w;
        dig[i] = (int8_t)(coef - (carry << w));
    }
    if (w == 8) { dig[count] = (int8_t)carry; return count + 1; }
    dig[count - 1] += (int8_t)(carry << w);
    return count;
}
/* Pippenger bucket method (dalek pippenger.rs): w = 6 / 7 / 8 for n < 500 / < 800 / >= 800 */
static void ge_pippenger(ge *r, const sc *k, const ge *p, size_t n) {
    int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
    int nb = 1 << (w - 1);
    int8_t (*dig)[48] = malloc(n * 48);
    int cols = 0;
    for (size_t i = 0; i < n; i++) cols = sc_radix2w(dig[i], &k[i], w);
    ge_cached *pc = malloc(n * sizeof(ge_cached));
    for (size_t i = 0; i < n; i++) ge_to_cached(&pc[i], &p[i]);
    ge *bucket = malloc(nb * sizeof(ge));
    ge total; ge_identity(&total);
    for (int c = cols - 1; c >= 0; c--) {
        for (int j = 0; j < nb; j++) ge_identity(&bucket[j]);
        for (size_t i = 0; i < n; i++) {
            int d = dig[i][c]; ge t;
            if (d > 0) { ge_add_cached(&t, &bucket[d - 1], &pc[i], 0); bucket[d - 1] = t; }
            else if (d < 0) { ge_add_cached(&t, &bucket[-d - 1], &pc[i], 1); bucket[-d - 1] = t; }
        }
        ge run = bucket[nb - 1], sum = bucket[nb - 1], t;
        for (int j = nb - 2; j >= 0; j--) { ge_add(&t, &run, &bucket[j]); run = t; ge_add(&t, &sum, &run); sum = t; }
        for (int j = 0; j < w; j++) { ge_dbl(&t, &total); total = t; }
        ge_add(&t, &total, &sum); total = t;
    }
    *r = total; free(dig); free(pc); free(bucket);
}
static void ge_msm(ge *r, const sc *k, const ge *p, size_t n) {   /* dalek: Straus < 190 <= Pippenger */
    if (n < 190) ge_straus(r, k, p, n); else ge_pippenger(r, k, p, n);
}
static void ge_scalarmult(ge *r, const sc *k, const ge *p) { ge_straus(r, k, p, 1); }

/* ------------------------------------------------------------------ sign / verify */
static void expand_seed(const uint8_t seed[32], sc *a, uint8_t prefix[32]) {
    uint8_t h[64]; hs_oracle_sha512(seed, 32, h);
    h[0] &= 248; h[31] &= 63; h[31] |= 64;
    memcpy(a->v, h, 32); memcpy(prefix, h + 32, 32);     /* a is NOT reduced; used as an integer */
}
void hs_oracle_public_key(const uint8_t seed[32], uint8_t pk[32]) {
    sc a; uint8_t prefix[32]; expand_seed(seed, &a, prefix);
    ge A; ge_scalarmult(&A, &a, &GE_B); ge_compress(pk, &A);
}
/* RFC 8032 5.1.6 == dalek Keypair::sign (crypto/src/lib.rs:185-191) */
void hs_oracle_sign(const uint8_t seed[32], const uint8_t *msg, size_t mlen, uint8_t sig[64]) {
    sc a, r, k, s; uint8_t prefix[32], h[64], pk[32];
    expand_seed(seed, &a, prefix);
    ge A, R; ge_scalarmult(&A, &a, &GE_B); ge_compress(pk, &A);
    sha512_ctx c; sha512_init(&c); sha512_update(&c, prefix, 32); sha512_update(&c, msg, mlen); sha512_final(&c, h);
    sc_from_bytes64(&r, h);
    ge_scalarmult(&R, &r, &GE_B); ge_compress(sig, &R);
    sha512_init(&c); sha512_update(&c, sig, 32); sha512_update(&c, pk, 32); sha512_update(&c, msg, mlen); sha512_final(&c, h);
    sc_from_bytes64(&k, h);
    sc ar; uint64_t a5[4]; memcpy(a5, a.v, 32); sc_reduce_limbs(&ar, a5, 4);
    sc_mul(&s, &k, &ar); sc_add(&s, &s, &r);
    memcpy(sig + 32, s.v, 32);
}
static void hram(sc *k, const uint8_t R[32], const uint8_t A[32], const uint8_t *msg, size_t mlen) {
    uint8_t h[64]; sha512_ctx c; sha512_init(&c);
    sha512_update(&c, R, 32); sha512_update(&c, A, 32); sha512_update(&c, msg, mlen); sha512_final(&c, h);
    sc_from_bytes64(k, h);
}
/* Signature::from_bytes + check_scalar (lib.rs:201,215) */
static int sig_format_ok(const uint8_t sig[64]) { return !(sig[63] & 0xE0) && sc_is_canonical(sig + 32); }

/* mode 1 = strict (verify_strict), mode 0 = per-signature cofactorless verdict */
static int verify_one(const uint8_t *msg, size_t mlen, const uint8_t pk[32], const uint8_t sig[64], int strict) {
    if (!sig_format_ok(sig)) return 0;
    ge A, R, mA, Rp;
    if (!ge_decompress(&A, pk)) return 0;
    if (!ge_decompress(&R, sig)) return 0;
    if (strict && (ge_is_small_order(&R) || ge_is_small_order(&A))) return 0;
    sc ks[2]; ge ps[2];
    hram(&ks[0], sig, pk, msg, mlen); sc_from_bytes32(&ks[1], sig + 32);
    ge_neg(&mA, &A); ps[0] = mA; ps[1] = GE_B;
    ge_straus(&Rp, ks, ps, 2);          /* vartime_double_scalar_mul_basepoint */
    return ge_eq(&Rp, &R);
}
int hs_oracle_verify_strict(const uint8_t *msg, size_t mlen, const uint8_t pk[32], const uint8_t sig[64]) {
    return verify_one(msg, mlen, pk, sig, 1);
}
int hs_oracle_verify_cofactorless(const uint8_t *msg, size_t mlen, const uint8_t pk[32], const uint8_t sig[64]) {
    return verify_one(msg, mlen, pk, sig, 0);
}

/* splitmix-style expander for the z_i (dalek: merlin transcript + thread_rng; the
 * distribution - uniform 128-bit - is what matters for the accept/reject law) */
static uint64_t mix64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31);
}
/* dalek::verify_batch through Signature::verify_batch (lib.rs:206-219).
 * msgs: n messages of mlen bytes, stride mstride (0 = one shared message). */
int hs_oracle_verify_batch(const uint8_t *msgs, size_t mlen, size_t mstride,
                           const uint8_t *pks, const uint8_t *sigs, size_t n, uint64_t entropy) {
    size_t m = 2 * n + 1;
    sc *k = malloc(m * sizeof(sc)); ge *p = malloc(m * sizeof(ge));
    int ok = 1;
    /* lib.rs:213-217: per-pair from_bytes checks in order, early return */
    for (size_t i = 0; i < n && ok; i++) {
        if (!sig_format_ok(sigs + 64 * i)) ok = 0;
        else if (!ge_decompress(&p[1 + n + i], pks + 32 * i)) ok = 0;
    }
    sc bsum; memset(&bsum, 0, sizeof bsum);
    uint8_t seedh[64]; hs_oracle_sha512(sigs, 64 * n, seedh);
    uint64_t st; memcpy(&st, seedh, 8); st ^= entropy;
    for (size_t i = 0; i < n && ok; i++) {
        sc h, s, z, t;
        hram(&h, sigs + 64 * i, pks + 32 * i, msgs + mstride * i, mlen);
        sc_from_bytes32(&s, sigs + 64 * i + 32);
        z.v[0] = mix64(&st); z.v[1] = mix64(&st); z.v[2] = 0; z.v[3] = 0;
        sc_mul(&t, &z, &s); sc_add(&bsum, &bsum, &t);
        k[1 + i] = z; sc_mul(&k[1 + n + i], &z, &h);
        if (!ge_decompress(&p[1 + i], sigs + 64 * i)) ok = 0;   /* optional_multiscalar_mul -> None */
    }
    if (ok) {
        sc_neg(&k[0], &bsum); p[0] = GE_B;
        ge id; ge_msm(&id, k, p, m);
        ok = ge_is_identity(&id);
    }
    free(k); free(p);
    return ok;
}

/* ------------------------------------------------------------------ bulk (threaded) helpers */
typedef struct {
    const uint8_t *msgs; size_t mlen, mstride; const uint8_t *pks, *sigs; uint8_t *out; size_t lo, hi; int mode;
    const uint8_t *seeds; uint8_t *pk_out, *sig_out;
} job_t;
static void *verify_job(void *arg) {
    job_t *j = arg;
    for (size_t i = j->lo; i < j->hi; i++)
        j->out[i] = (uint8_t)verify_one(j->msgs + j->mstride * i, j->mlen, j->pks + 32 * i, j->sigs + 64 * i, j->mode);
    return NULL;
}
static void *sign_job(void *arg) {
    job_t *j = arg;
    for (size_t i = j->lo; i < j->hi; i++) {
        hs_oracle_public_key(j->seeds + 32 * i, j->pk_out + 32 * i);
        hs_oracle_sign(j->seeds + 32 * i, j->msgs + j->mstride * i, j->mlen, j->sig_out + 64 * i);
    }
    return NULL;
}
static void run_jobs(job_t *proto, size_t n, int nthreads, void *(*fn)(void *)) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = malloc(nthreads * sizeof *th); job_t *jobs = malloc(nthreads * sizeof *jobs);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = *proto; jobs[t].lo = n * t / nthreads; jobs[t].hi = n * (t + 1) / nthreads;
        if (nthreads == 1) fn(&jobs[t]); else pthread_create(&th[t], NULL, fn, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
/* out[i] = verdict (1/0) per pair; mode 1 strict, 0 cofactorless */
void hs_oracle_verify_many(const uint8_t *msgs, size_t mlen, size_t mstride, const uint8_t *pks,
                           const uint8_t *sigs, size_t n, int mode, uint8_t *out, int nthreads) {
    job_t j = {0}; j.msgs = msgs; j.mlen = mlen; j.mstride = mstride; j.pks = pks; j.sigs = sigs; j.out = out; j.mode = mode;
    run_jobs(&j, n, nthreads, verify_job);
}
void hs_oracle_sign_many(const uint8_t *seeds, const uint8_t *msgs, size_t mlen, size_t mstride, size_t n,
                         uint8_t *pk_out, uint8_t *sig_out, int nthreads) {
    job_t j = {0}; j.seeds = seeds; j.msgs = msgs; j.mlen = mlen; j.mstride = mstride; j.pk_out = pk_out; j.sig_out = sig_out;
    run_jobs(&j, n, nthreads, sign_job);
}
/* TC::verify's signature loop as the reference runs it (consensus/src/messages.rs:307-313): per vote
 *   digest = SHA-512(round.to_le_bytes() || high_qc_round.to_le_bytes())[..32];  signature.verify(&digest, author)  (= verify_strict)
 * out[i] = verdict per vote.  bench.py times it beside hs_ed25519_verify_tc. */
typedef struct { job_t j; uint64_t round; const uint64_t *hq; } tjob_t;
static void *tc_job(void *arg) {
    tjob_t *t = arg; job_t *j = &t->j;
    for (size_t i = j->lo; i < j->hi; i++) {
        uint8_t m[16], h[64];
        for (int b = 0; b < 8; b++) { m[b] = (uint8_t)(t->round >> (8 * b)); m[8 + b] = (uint8_t)(t->hq[i] >> (8 * b)); }
        hs_oracle_sha512(m, 16, h);
        j->out[i] = (uint8_t)verify_one(h, 32, j->pks + 32 * i, j->sigs + 64 * i, 1);
    }
    return NULL;
}
void hs_oracle_verify_tc(uint64_t round, const uint64_t *hq_rounds, const uint8_t *pks, const uint8_t *sigs, size_t n,
                         uint8_t *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = malloc(nthreads * sizeof *th); tjob_t *jobs = calloc(nthreads, sizeof *jobs);
    for (int t = 0; t < nthreads; t++) {
        job_t *j = &jobs[t].j; j->pks = pks; j->sigs = sigs; j->out = out;
        j->lo = n * t / nthreads; j->hi = n * (t + 1) / nthreads; jobs[t].round = round; jobs[t].hq = hq_rounds;
        if (nthreads == 1) tc_job(&jobs[t]); else pthread_create(&th[t], NULL, tc_job, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
/* batch verify of n pairs split into nthreads independent sub-batches (all must pass) */
typedef struct { job_t j; int ok; uint64_t entropy; } bjob_t;
static void *batch_job(void *arg) {
    bjob_t *b = arg; job_t *j = &b->j;
    b->ok = hs_oracle_verify_batch(j->msgs + j->mstride * j->lo, j->mlen, j->mstride, j->pks + 32 * j->lo,
                                   j->sigs + 64 * j->lo, j->hi - j->lo, b->entropy);
    return NULL;
}
int hs_oracle_verify_batch_mt(const uint8_t *msgs, size_t mlen, size_t mstride, const uint8_t *pks,
                              const uint8_t *sigs, size_t n, uint64_t entropy, int nthreads) {
    if (nthreads <= 1 || n < 2 * (size_t)nthreads) return hs_oracle_verify_batch(msgs, mlen, mstride, pks, sigs, n, entropy);
    pthread_t *th = malloc(nthreads * sizeof *th); bjob_t *jobs = calloc(nthreads, sizeof *jobs);
    for (int t = 0; t < nthreads; t++) {
        job_t *j = &jobs[t].j; j->msgs = msgs; j->mlen = mlen; j->mstride = mstride; j->pks = pks; j->sigs = sigs;
        j->lo = n * t / nthreads; j->hi = n * (t + 1) / nthreads; jobs[t].entropy = entropy + t;
        pthread_create(&th[t], NULL, batch_job, &jobs[t]);
    }
    int ok = 1;
    for (int t = 0; t < nthreads; t++) { pthread_join(th[t], NULL); ok &= jobs[t].ok; }
    free(th); free(jobs); return ok;
}

/* ------------------------------------------------------------------ low-level probes for unit tests */
void hs_oracle_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t r[32]) { fe x, y, z; fe_frombytes(x, a); fe_frombytes(y, b); fe_mul(z, x, y); fe_tobytes(r, z); }
void hs_oracle_fe_invert(const uint8_t a[32], uint8_t r[32]) { fe x, z; fe_frombytes(x, a); fe_invert(z, x); fe_tobytes(r, z); }
void hs_oracle_fe_pow22523(const uint8_t a[32], uint8_t r[32]) { fe x, z; fe_frombytes(x, a); fe_pow22523(z, x); fe_tobytes(r, z); }
int hs_oracle_decompress(const uint8_t s[32], uint8_t x[32], uint8_t y[32]) {
    ge p; if (!ge_decompress(&p, s)) return 0; fe_tobytes(x, p.X); fe_tobytes(y, p.Y); return 1;
}
void hs_oracle_sc_reduce64(const uint8_t h[64], uint8_t r[32]) { sc s; sc_from_bytes64(&s, h); memcpy(r, s.v, 32); }
void hs_oracle_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t r[32]) { sc x, y, z; memcpy(x.v, a, 32); memcpy(y.v, b, 32); sc_mul(&z, &x, &y); memcpy(r, z.v, 32); }
/* r = compress([k]P + [s]B) with P = decompress(p); returns 0 if P is off-curve */
int hs_oracle_double_scalarmult(const uint8_t k[32], const uint8_t p[32], const uint8_t s[32], uint8_t r[32]) {
    ge P, R; if (!ge_decompress(&P, p)) return 0;
    sc ks[2]; ge ps[2]; memcpy(ks[0].v, k, 32); memcpy(ks[1].v, s, 32); ps[0] = P; ps[1] = GE_B;
    ge_straus(&R, ks, ps, 2); ge_compress(r, &R); return 1;
}
/* r = compress(sum k_i P_i) through the dalek size switch (Straus < 190 <= Pippenger) */
int hs_oracle_msm(const uint8_t *ks, const uint8_t *ps, size_t n, uint8_t r[32]) {
    sc *k = malloc(n * sizeof(sc) + 1); ge *p = malloc(n * sizeof(ge) + 1); int ok = 1;
    for (size_t i = 0; i < n; i++) { memcpy(k[i].v, ks + 32 * i, 32); if (!ge_decompress(&p[i], ps + 32 * i)) ok = 0; }
    if (ok) { ge R; ge_msm(&R, k, p, n); ge_compress(r, &R); }
    free(k); free(p); return ok;
}
