/* tests/craft/craft.c - adversarial Ed25519 tuples by construction, at the rate a 10^7 differential needs.
 *
 * TEST INFRASTRUCTURE (like oracle/, whose arithmetic it reuses by including the source): only tests/ and
 * tools/differential.py load libhs_craft.so.  Nothing here verifies for the product, and nothing here decides a verdict the
 * tests trust: verdicts come from the oracle's own entry points (hs_oracle_verify_strict / _cofactorless through
 * craft_verify_both_many, which is checked against those two functions on a sample by every run of the differential).
 *
 * Why a signer and not byte flips: a flipped byte lands in "off the curve" or "wrong equation" - the two
 * classes every implementation gets right.  The classes that separate implementations of the reference's predicates
 * (/root/reference/crypto/src/lib.rs:200-219 -> ed25519-dalek 1.0.1 verify_strict / verify_batch) need signatures whose
 * EQUATION is crafted:
 *   A' = [a]B + T_i, R' = [r]B + T_j with T an 8-torsion point, both hashed as encoded, s = r + H(R' || A' || M) a:
 *       [s]B - [k]A' = R' - T_j - [k]T_i, so the cofactorless equation holds iff T_j + [k]T_i = O - a verdict that depends on
 *       k mod 8 and on the torsion components surviving every scalar transformation on the way (the half-size form multiplies
 *       the whole equation by v: an even v, or a congruence taken mod L instead of mod 8L, accepts where dalek rejects);
 *   a = 0 (A' a pure torsion point): what verify_strict's small-order rejection is for;
 *   the alternative encodings dalek's lenient decompression accepts: y + p for y < 19, and "-0" (x = 0 with the sign bit set);
 *   s outside [0, L): s + L (the same scalar mod L - an implementation that reduces before it checks accepts), L - 1, L, L + 1,
 *       2^252, high bits set.
 */
#include "../../oracle/ed25519_oracle.c"

/* spec word of one tuple */
#define CR_A_TORS(s)   ((s) & 7u)              /* A' = [a]B + T_i                                   */
#define CR_R_TORS(s)   (((s) >> 3) & 7u)       /* R' = [r]B + T_j                                   */
#define CR_ZERO_A      (1u << 6)               /* a = 0: A' is the torsion point alone              */
#define CR_ZERO_R      (1u << 7)               /* r = 0: R' is the torsion point alone              */
#define CR_A_ENC(s)    (((s) >> 8) & 3u)       /* bit 0: y + p when y < 19; bit 1: "-0" when x = 0  */
#define CR_R_ENC(s)    (((s) >> 10) & 3u)
#define CR_S_MUT(s)    (((s) >> 12) & 15u)     /* 0 none, 1 s + L, 2 L - 1, 3 L, 4 L + 1, 5 2^252, 6 s | 2^253, 7 s | 2^255, 8 zero, 9 s + 1, 10 s ^ 2^127 */

/* the 8-torsion subgroup: T[j] = [j]T[1], T[1] of order 8 (y = 0x7a03ac92...c7; the encoding every torsion table lists) */
static const uint8_t TORSION8[32] = { 0xc7, 0x17, 0x6a, 0x70, 0x3d, 0x4d, 0xd8, 0x4f, 0xba, 0x3c, 0x0b, 0x76, 0x0d, 0x10, 0x67, 0x0f,
                                      0x2a, 0x20, 0x53, 0xfa, 0x2c, 0x39, 0xcc, 0xc6, 0x4e, 0xc7, 0xfd, 0x77, 0x92, 0xac, 0x03, 0x7a };
static ge TORS[8];
static pthread_once_t tors_once = PTHREAD_ONCE_INIT;
static void tors_init(void) {
    ge_identity(&TORS[0]);
    if (!ge_decompress(&TORS[1], TORSION8)) abort();
    for (int j = 2; j < 8; j++) ge_add(&TORS[j], &TORS[j - 1], &TORS[1]);
    ge t; ge_add(&t, &TORS[7], &TORS[1]);
    if (!ge_is_identity(&t) || ge_is_identity(&TORS[4])) abort();          /* order exactly 8 */
}

/* canonical encoding, then the alternative encodings where the point has one */
static void encode_variant(uint8_t out[32], const ge *p, unsigned enc) {
    ge_compress(out, p);
    const uint8_t sign = out[31] & 0x80;
    uint8_t y[32]; memcpy(y, out, 32); y[31] &= 0x7f;
    int small = y[0] < 19, is_one = y[0] == 1, is_m1 = y[0] == 0xec && y[31] == 0x7f;
    for (int i = 1; i < 32; i++) { if (y[i]) small = is_one = 0; if (y[i] != (i == 31 ? 0x7f : 0xff)) is_m1 = 0; }
    if ((enc & 1) && small) {                  /* y + p = 2^255 - 19 + y */
        y[0] = (uint8_t)(0xed + y[0]);
        for (int i = 1; i < 31; i++) y[i] = 0xff;
        y[31] = 0x7f;
    }
    memcpy(out, y, 32);
    out[31] |= sign;
    if ((enc & 2) && (is_one || is_m1)) out[31] ^= 0x80;      /* x = 0: the other sign bit, "-0" */
}
static void add_L(uint64_t v[4]) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)v[i] + SC_L[i]; v[i] = (uint64_t)c; c >>= 64; } }

/* one tuple: key bytes (as hashed and as handed to the verifier) and signature */
static void craft_one(const uint8_t seed[32], const uint8_t msg[32], uint32_t spec, uint8_t pk[32], uint8_t sig[64]) {
    pthread_once(&tors_once, tors_init);
    sc a, r, k, s, ar; uint8_t prefix[32], h[64];
    expand_seed(seed, &a, prefix);
    if (spec & CR_ZERO_A) memset(&a, 0, sizeof a);
    ge A, R, t;
    ge_scalarmult(&t, &a, &GE_B); ge_add(&A, &t, &TORS[CR_A_TORS(spec)]);
    encode_variant(pk, &A, CR_A_ENC(spec));
    sha512_ctx c; sha512_init(&c); sha512_update(&c, prefix, 32); sha512_update(&c, msg, 32); sha512_final(&c, h);
    sc_from_bytes64(&r, h);
    if (spec & CR_ZERO_R) memset(&r, 0, sizeof r);
    ge_scalarmult(&t, &r, &GE_B); ge_add(&R, &t, &TORS[CR_R_TORS(spec)]);
    encode_variant(sig, &R, CR_R_ENC(spec));
    hram(&k, sig, pk, msg, 32);
    uint64_t a4[4]; memcpy(a4, a.v, 32); sc_reduce_limbs(&ar, a4, 4);
    sc_mul(&s, &k, &ar); sc_add(&s, &s, &r);
    uint64_t v[4]; memcpy(v, s.v, 32);
    switch (CR_S_MUT(spec)) {
    case 1: add_L(v); break;
    case 2: memcpy(v, SC_L, 32); v[0] -= 1; break;
    case 3: memcpy(v, SC_L, 32); break;
    case 4: memcpy(v, SC_L, 32); v[0] += 1; break;
    case 5: v[0] = v[1] = v[2] = 0; v[3] = 1ULL << 60; break;
    case 6: v[3] |= 1ULL << 61; break;
    case 7: v[3] |= 1ULL << 63; break;
    case 8: memset(v, 0, 32); break;
    case 9: { u128 cc = 1; for (int i = 0; i < 4; i++) { cc += v[i]; v[i] = (uint64_t)cc; cc >>= 64; } } break;
    case 10: v[1] ^= 1ULL << 63; break;
    default: break;
    }
    memcpy(sig + 32, v, 32);
}

typedef struct { const uint8_t *seeds, *msgs; const uint32_t *spec; uint8_t *pk, *sig; size_t lo, hi; } cjob;
static void *craft_job(void *arg) {
    cjob *j = arg;
    for (size_t i = j->lo; i < j->hi; i++) craft_one(j->seeds + 32 * i, j->msgs + 32 * i, j->spec[i], j->pk + 32 * i, j->sig + 64 * i);
    return NULL;
}
/* seeds, msgs: n x 32; spec: n words; out: pk n x 32, sig n x 64 */
void hs_craft_many(const uint8_t *seeds, const uint8_t *msgs, const uint32_t *spec, size_t n, uint8_t *pk, uint8_t *sig, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = malloc(nthreads * sizeof *th); cjob *jobs = malloc(nthreads * sizeof *jobs);
    for (int t = 0; t < nthreads; t++) {
        cjob j = { seeds, msgs, spec, pk, sig, n * t / nthreads, n * (t + 1) / nthreads };
        jobs[t] = j;
        pthread_create(&th[t], NULL, craft_job, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* Both labels of one tuple in ONE pass over the oracle's own pieces, in verify_one's order (the 10^7 run would otherwise pay the
 * double-scalar multiplication twice).  bit 0 = cofactorless (BATCH), bit 1 = strict.  strict = cofactorless and neither point
 * of small order: exactly how verify_one(strict = 1) differs from verify_one(strict = 0).  tools/differential.py checks this
 * function against hs_oracle_verify_strict / hs_oracle_verify_cofactorless on a sample of every chunk. */
static uint8_t verify_both(const uint8_t *msg, const uint8_t pk[32], const uint8_t sig[64]) {
    if (!sig_format_ok(sig)) return 0;
    ge A, R, mA, Rp;
    if (!ge_decompress(&A, pk)) return 0;
    if (!ge_decompress(&R, sig)) return 0;
    const int small = ge_is_small_order(&R) || ge_is_small_order(&A);
    sc ks[2]; ge ps[2];
    hram(&ks[0], sig, pk, msg, 32); sc_from_bytes32(&ks[1], sig + 32);
    ge_neg(&mA, &A); ps[0] = mA; ps[1] = GE_B;
    ge_straus(&Rp, ks, ps, 2);
    const int eq = ge_eq(&Rp, &R);
    return (uint8_t)((eq ? 1 : 0) | ((eq && !small) ? 2 : 0));
}
typedef struct { const uint8_t *msgs, *pk, *sig; uint8_t *out; size_t lo, hi; } vjob;
static void *both_job(void *arg) {
    vjob *j = arg;
    for (size_t i = j->lo; i < j->hi; i++) j->out[i] = verify_both(j->msgs + 32 * i, j->pk + 32 * i, j->sig + 64 * i);
    return NULL;
}
void hs_craft_verify_both_many(const uint8_t *msgs, const uint8_t *pk, const uint8_t *sig, size_t n, uint8_t *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = malloc(nthreads * sizeof *th); vjob *jobs = malloc(nthreads * sizeof *jobs);
    for (int t = 0; t < nthreads; t++) {
        vjob j = { msgs, pk, sig, out, n * t / nthreads, n * (t + 1) / nthreads };
        jobs[t] = j;
        pthread_create(&th[t], NULL, both_job, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}
/* [8]P == O for an encoded point (class bookkeeping of the reports); -1 when it does not decode */
int hs_craft_is_small_order(const uint8_t p[32]) { ge P; if (!ge_decompress(&P, p)) return -1; return ge_is_small_order(&P); }
