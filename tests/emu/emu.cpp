// tests/emu/emu.cpp - TEST INFRASTRUCTURE.  Compiles the product's host+device headers
// (hotstuff_amd/csrc/*.h) with g++ and runs the two lane stages serially on the CPU, so the
// exact device arithmetic can be checked against oracle/ without a GPU.  Never shipped and
// never used by the product path (which has no CPU fallback).
#include <cstring>
#include <vector>
#include "../../hotstuff_amd/csrc/verify.h"
#include "../../hotstuff_amd/csrc/verify_quad.h"
#include "../../hotstuff_amd/csrc/verify_cached.h"
#include "retired_cores.h"      // cores of kernels that left the product: three more organisations of the same primitives
#include "../../hotstuff_amd/csrc/rlc.h"
#include "../../hotstuff_amd/csrc/wire.h"
using namespace hs;

static void load_words(u32 *w, const uint8_t *b, int n) { memcpy(w, b, 4 * n); }

const uint32_t *emu_base16_fwd(void);
template <int W>
static void verify_quad_w(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *out) {
    const u32 *btab = emu_base16_fwd();        // the base point's 16-bit tables
    for (size_t i = 0; i < n; i++) {
        u32 pk[8], sig[16], msg[8];
        load_words(pk, pks + 32 * i, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        ge_p3 A, R;
        u32 fa = ge_decompress(A, pk) ? HS_F_ONCURVE : 0, fr = ge_decompress(R, sig) ? HS_F_ONCURVE : 0;
        if (mode & HS_MODE_STRICT) { fa |= ge_is_small_order_affine(A.Y) ? HS_F_SMALL : 0; fr |= ge_is_small_order_affine(R.Y) ? HS_F_SMALL : 0; }
        half_scalars hs;
        bool ok = prepare_scalars_half<W>(hs, pk, sig, msg);
        ok = ok && (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE);
        if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);
        qtab atab, rtab; qflags f = verify_quad_core<W>(A.X, A.Y, R.X, R.Y, hs, (int)(hs.meta & 0xffu), btab, atab, rtab);
        out[i] = (ok && f.f[0] && f.f[1] && f.f[2] && f.f[3]) ? 1 : 0;
    }
}
extern "C" {
void emu_fe_mul(const uint8_t *a, const uint8_t *b, uint8_t *r) {
    u32 wa[8], wb[8], wr[8]; load_words(wa, a, 8); load_words(wb, b, 8);
    fe x, y, z; fe_from_words(x, wa); fe_from_words(y, wb); fe_mul(z, x, y); fe_to_words(wr, z); memcpy(r, wr, 32);
}
// loose-operand multiplication: (a1 + a2) * (b1 + 2p - b2)  exercises the 2x / 3x bounds
void emu_fe_mul_loose(const uint8_t *a1, const uint8_t *a2, const uint8_t *b1, const uint8_t *b2, uint8_t *r) {
    u32 w[8], wr[8]; fe x1, x2, y1, y2, f, g, z;
    load_words(w, a1, 8); fe_from_words(x1, w); load_words(w, a2, 8); fe_from_words(x2, w);
    load_words(w, b1, 8); fe_from_words(y1, w); load_words(w, b2, 8); fe_from_words(y2, w);
    fe_add(f, x1, x2); fe_sub(g, y1, y2); fe_mul(z, f, g); fe_to_words(wr, z); memcpy(r, wr, 32);
}
void emu_fe_sq_loose(const uint8_t *a1, const uint8_t *a2, uint8_t *r) {   // (a1 + 2p - a2)^2
    u32 w[8], wr[8]; fe x1, x2, f, z;
    load_words(w, a1, 8); fe_from_words(x1, w); load_words(w, a2, 8); fe_from_words(x2, w);
    fe_sub(f, x1, x2); fe_sq(z, f); fe_to_words(wr, z); memcpy(r, wr, 32);
}
void emu_fe_sq(const uint8_t *a, uint8_t *r) {
    u32 wa[8], wr[8]; load_words(wa, a, 8); fe x, z; fe_from_words(x, wa); fe_sq(z, x); fe_to_words(wr, z); memcpy(r, wr, 32);
}
void emu_fe_invert(const uint8_t *a, uint8_t *r) {
    u32 wa[8], wr[8]; load_words(wa, a, 8); fe x, z; fe_from_words(x, wa); fe_invert(z, x); fe_to_words(wr, z); memcpy(r, wr, 32);
}
void emu_fe_roundtrip(const uint8_t *a, uint8_t *r) {
    u32 wa[8], wr[8]; load_words(wa, a, 8); fe x; fe_from_words(x, wa); fe_to_words(wr, x); memcpy(r, wr, 32);
}
// deliberately breaks a documented bound (second multiplication operand far above 3x): the
// HS_BOUND_CHECK build must abort here, the plain build silently computes garbage
void emu_violate_bounds(uint8_t *r) {
    fe x; for (int i = 0; i < 10; i++) x.v[i] = 0x03ffffff;
    fe y = x;
    for (int k = 0; k < 4; k++) fe_add(y, y, y);      // 16x
    fe z; fe_mul(z, x, y);
    u32 w[8]; fe_to_words(w, z); memcpy(r, w, 32);
}
void emu_sha512_96(const uint8_t *in, uint8_t *out) { u32 wi[24], wo[16]; load_words(wi, in, 24); sha512_96(wo, wi); memcpy(out, wo, 64); }
void emu_tc_vote_digest(uint64_t round, uint64_t hq, uint8_t *out) { u32 w[8]; tc_vote_digest(w, round, hq); memcpy(out, w, 32); }
void emu_sc_reduce512(const uint8_t *in, uint8_t *out) { u32 wi[16], wo[8]; load_words(wi, in, 16); sc_reduce512(wo, wi); memcpy(out, wo, 32); }
void emu_sc_mul(const uint8_t *a, const uint8_t *b, uint8_t *out) { u32 wa[8], wb[8], wo[8]; load_words(wa, a, 8); load_words(wb, b, 8); sc_mul(wo, wa, wb); memcpy(out, wo, 32); }
// halfscalar.h: (|u|, u < 0, v) for k; returns 1 when the pair passed its own check (0: the (k, 1) fallback was taken)
int emu_half_scalar(const uint8_t *k32, uint8_t *u32_out, int *u_neg, uint8_t *v32_out) {
    u32 k[8]; load_words(k, k32, 8);
    half_pair h; half_scalar(h, k);
    memcpy(u32_out, h.u, 32); memcpy(v32_out, h.v, 32); *u_neg = (int)h.u_neg;
    bool same = h.u_neg == 0; for (int i = 0; i < 8; i++) same = same && h.u[i] == k[i] && h.v[i] == (i == 0 ? 1u : 0u);
    return same ? 0 : 1;
}
// the predicate alone on a CHOSEN pair: 1 = half_scalar_ok would let (u, v) be used for k
int emu_half_scalar_ok(const uint8_t *k32, const uint8_t *absu32, int u_neg, const uint8_t *v32) {
    u32 k[8]; load_words(k, k32, 8);
    half_pair h; load_words(h.u, absu32, 8); load_words(h.v, v32, 8); h.u_neg = u_neg ? 1u : 0u;
    return half_scalar_ok(h, k) ? 1 : 0;
}
int emu_recode16_len(const uint8_t *x32) { u32 x[8], r[8]; load_words(x, x32, 8); sc_recode16(r, x); return sc_recode16_len(r); }
int emu_sc_is_canonical(const uint8_t *s) { u32 w[8]; load_words(w, s, 8); return sc_is_canonical(w); }
void emu_sc_digits8(const uint8_t *s, int16_t *digits) {
    u32 w[8], r[8]; load_words(w, s, 8); sc_recode256(r, w);
    for (int i = 0; i < 32; i++) digits[i] = (int16_t)sc_pop_digit8(r);
}
// digits of the signed radix-16 recoding, most significant first
void emu_sc_digits(const uint8_t *s, int8_t *digits) {
    u32 w[8], r[8]; load_words(w, s, 8); sc_recode16(r, w);
    for (int i = 0; i < 64; i++) digits[i] = (int8_t)sc_pop_digit(r);
}
int emu_decompress(const uint8_t *s, uint8_t *x, uint8_t *y, int *small) {
    u32 w[8], wx[8], wy[8]; load_words(w, s, 8); ge_p3 p; bool ok = ge_decompress(p, w);
    fe_to_words(wx, p.X); fe_to_words(wy, p.Y); memcpy(x, wx, 32); memcpy(y, wy, 32);
    // bit 0: the canonical-y comparison the kernels use after a decompression; bit 1: [8]P == O by three doublings
    *small = ok ? (int)ge_is_small_order_affine(p.Y) | ((int)ge_is_small_order(p) << 1) : 0;
    return ok;
}
const uint32_t *emu_base16(void);
// out[i] = per-signature verdict.  msg_stride = 0 (shared digest) or 32.
void emu_verify(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n,
                uint32_t mode, uint8_t *out) {
    if (!n) return;
    const u32 np = (u32)((n + 63) / 64 * 64);
    std::vector<u32> buf(workspace_words(np));
    workspace ws = workspace_at(buf.data(), np);
    const u32 *btab = emu_base16();
    for (size_t t = 0; t < 2 * n; t++) {
        const bool is_r = t >= n; const size_t i = is_r ? t - n : t;
        u32 pk[8], sig[16], msg[8];
        load_words(pk, pks + 32 * i, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        prepare_point(ws, (u32)i, is_r, pk, sig, msg, mode);
    }
    for (size_t i = 0; i < n; i++) out[i] = verify_lane(ws, (u32)i, mode, btab) ? 1 : 0;
}
const uint32_t *emu_base16(void);
// the quad (4 lanes per signature) core (verify_quad_core) behind the scalar half of stage 1 (prepare_scalars) and a
// plain decompression of A and R - what verify_quad_fused does on the device with DPP broadcasts
// 4-bit windows (the latency kernel k_verify_fused) and 3-bit windows (k_verify_fused3, the batch-list kernels)
void emu_verify_quad(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *out) {
    verify_quad_w<4>(msgs, msg_stride, pks, sigs, n, mode, out);
}
void emu_verify_quad3(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *out) {
    verify_quad_w<3>(msgs, msg_stride, pks, sigs, n, mode, out);
}
// the eight-lane latency form (verify_oct_fused): the two roles' chains and the join, run one after the other
void emu_verify_oct(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *out) {
    const u32 *btab = emu_base16();
    for (size_t i = 0; i < n; i++) {
        u32 pk[8], sig[16], msg[8];
        load_words(pk, pks + 32 * i, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        ge_p3 A, R;
        u32 fa = ge_decompress(A, pk) ? HS_F_ONCURVE : 0, fr = ge_decompress(R, sig) ? HS_F_ONCURVE : 0;
        if (mode & HS_MODE_STRICT) { fa |= ge_is_small_order_affine(A.Y) ? HS_F_SMALL : 0; fr |= ge_is_small_order_affine(R.Y) ? HS_F_SMALL : 0; }
        half_scalars hs;
        bool ok = prepare_scalars_half<4>(hs, pk, sig, msg);
        ok = ok && (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE);
        if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);
        const int nwin = (int)(hs.meta & 0xffu);
        qv part[2];
        for (int role = 0; role < 2; role++) {
            const ge_p3 &P = role ? R : A;
            fe nx, sx; fe_neg(nx, P.X);
            fe_select(sx, nx, P.X, !role && ((hs.meta >> 8) & 1) != 0);
            u32 rd[8], rwl[8];
            quad_half_operands(rd, rwl, hs, role != 0);
            qtab tab;
            part[role] = quad_half_chain(sx, P.Y, rd, rwl, role ? 8u : 0u, nwin, btab, tab);
        }
        const qflags f0 = quad_half_join(part[0], part[1]), f1 = quad_half_join(part[1], part[0]);
        bool all = ok;
        for (int l = 0; l < 4; l++) all = all && f0.f[l] && f1.f[l];
        out[i] = all ? 1 : 0;
    }
}
// the same as the split form computes it (k_verify_fused8h): the table is ALWAYS j (-P), built before the scalars are known, and
// the sign of u is applied at the lookup (quad_half_walk's flip)
void emu_verify_oct_split(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *out) {
    const u32 *btab = emu_base16();
    for (size_t i = 0; i < n; i++) {
        u32 pk[8], sig[16], msg[8];
        load_words(pk, pks + 32 * i, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        ge_p3 A, R;
        u32 fa = ge_decompress(A, pk) ? HS_F_ONCURVE : 0, fr = ge_decompress(R, sig) ? HS_F_ONCURVE : 0;
        if (mode & HS_MODE_STRICT) { fa |= ge_is_small_order_affine(A.Y) ? HS_F_SMALL : 0; fr |= ge_is_small_order_affine(R.Y) ? HS_F_SMALL : 0; }
        qtab tabs[2];
        for (int role = 0; role < 2; role++) {          // before the "barrier": points and tables only
            const ge_p3 &P = role ? R : A;
            fe nx; fe_neg(nx, P.X);
            qtab_build<4>(tabs[role], nx, P.Y);
        }
        half_scalars hs;
        bool ok = prepare_scalars_half<4>(hs, pk, sig, msg);
        ok = ok && (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE);
        if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);
        const int nwin = (int)(hs.meta & 0xffu);
        qv part[2];
        for (int role = 0; role < 2; role++) {
            u32 rd[8], rwl[8];
            quad_half_operands(rd, rwl, hs, role != 0);
            part[role] = quad_half_walk(rd, rwl, role ? 8u : 0u, nwin, !role && ((hs.meta >> 8) & 1) != 0, btab, tabs[role]);
        }
        const qflags f0 = quad_half_join(part[0], part[1]), f1 = quad_half_join(part[1], part[0]);
        bool all = ok;
        for (int l = 0; l < 4; l++) all = all && f0.f[l] && f1.f[l];
        out[i] = all ? 1 : 0;
    }
}
// radix-8 recoding: digits most significant first (85 of them) and the count the kernels use
int emu_sc_digits3(const uint8_t *x32, int8_t *digits) {
    u32 x[8], r[8]; load_words(x, x32, 8); sc_recode8(r, x);
    const int n = sc_recode8_len(r);
    sc_align8(r);
    for (int i = 0; i < HS_RADIX8_DIGITS; i++) digits[i] = (int8_t)sc_pop_digit3(r);
    return n;
}
// Every finishing form must reject an all-zero accumulator (what an arithmetic fault leaves behind:
// it is absorbing under the addition formulas and satisfies X == x*Z, Y == y*Z trivially).
// Bit i of the result is set when form i ACCEPTED it: 0 lane, 1 quad point match, 2 quad
// encoding match (inversion), 3 one-lane cached finish, 4 the batched finish (cfin_lane; bit 5: it spoilt a genuine
// neighbour).  Batch mode, R = (sqrt(-1), 0) / y = 0.
uint32_t emu_zero_accumulator_accepts(void) {
    fe zero, one; fe_0(zero); fe_1(one);
    ge_p3 acc; acc.X = acc.Y = acc.Z = acc.T = zero;
    u32 out = 0;
    // any point will do for the comparisons; use the base point as "R"
    ge_p3 b; ge_basepoint(b);
    if (ge_match_point(acc, b.X, b.Y)) out |= 1;
    const qv q = q_lanes(zero, zero, zero, zero);
    const qflags f = q_match_point(q, b.X, b.Y);
    if (f.f[0] && f.f[1] && f.f[2] && f.f[3]) out |= 2;
    u32 renc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                 // encoding of y = 0, sign 0
    if (q_quad_all(q_match_encoding(q_mul(q, q_invert(q_perm<2, 2, 2, 2>(q))), renc))) out |= 4;
    if (cached_finish(acc, renc, false)) out |= 8;
    {   // the batched finish: a zeroed accumulator among genuine neighbours (which must still be accepted: bit 16 set = they were not)
        const u32 np = 64;
        std::vector<u32> fin((size_t)HS_CFIN_ROWS * np, 0), sg(16 * 3, 0);
        ge_p3 p; ge_p1p1 t; ge_dbl(t, b); ge_p1p1_to_p3(p, t, true);          // 2B with Z != 1
        fe zi, ax, ay; fe_invert(zi, p.Z); fe_mul(ax, p.X, zi); fe_mul(ay, p.Y, zi);
        u32 good[8]; fe_to_words(good, ay); { u32 xw[8]; fe_to_words(xw, ax); good[7] |= (xw[0] & 1) << 31; }
        cfin_store(fin.data(), np, 0, p, true); cfin_store(fin.data(), np, 1, acc, true); cfin_store(fin.data(), np, 2, p, true);
        for (int i = 0; i < 8; i++) { sg[i] = good[i]; sg[16 + i] = renc[i]; sg[32 + i] = good[i]; }
        const u32 bits = cfin_lane(fin.data(), np, 0, 1, 3, sg.data(), false);     // one lane, its three signatures in a row
        if (bits & 2) out |= 16;
        if ((bits & 5) != 5) out |= 32;
    }
    return out;
}
// sanity of the same four forms on a genuine point: acc = 3B scaled by Z = 5, R = 3B
uint32_t emu_genuine_point_accepts(void) {
    ge_p3 b, p; ge_basepoint(b);
    ge_p1p1 t; ge_cached c; ge_dbl(t, b); ge_p1p1_to_p3(p, t, true); ge_to_cached(c, b); ge_add(t, p, c, false); ge_p1p1_to_p3(p, t, true);
    fe zi, ax, ay; fe_invert(zi, p.Z); fe_mul(ax, p.X, zi); fe_mul(ay, p.Y, zi);
    u32 out = 0;
    if (ge_match_point(p, ax, ay)) out |= 1;
    const qv q = q_lanes(p.X, p.Y, p.Z, p.T);
    const qflags f = q_match_point(q, ax, ay);
    if (f.f[0] && f.f[1] && f.f[2] && f.f[3]) out |= 2;
    u32 renc[8]; fe_to_words(renc, ay); { u32 xw[8]; fe_to_words(xw, ax); renc[7] |= (xw[0] & 1) << 31; }
    if (q_quad_all(q_match_encoding(q_mul(q, q_invert(q_perm<2, 2, 2, 2>(q))), renc))) out |= 4;
    if (cached_finish(p, renc, false)) out |= 8;
    {
        const u32 np = 64;
        std::vector<u32> fin((size_t)HS_CFIN_ROWS * np, 0), sg(16, 0);
        cfin_store(fin.data(), np, 0, p, true);
        for (int i = 0; i < 8; i++) sg[i] = renc[i];
        if (cfin_lane(fin.data(), np, 0, 64, 1, sg.data(), false) == 1u) out |= 16;
    }
    return out;
}
// ---- committee key cache (verify_cached.h) ----
// The device normalises the table entries in groups of eight sharing one inversion (keyset_affine_group, one lane per
// group).  516 inversions per key serially on the CPU would still make this harness the slowest thing in the suite, so
// the emulation normalises a whole window with one inversion (129 entries) - and emu_affine_forms_agree checks that
// this form, the device's group form and the plain per-entry form (keyset_affine_entry) give the same field elements.
static void affine_window_batch(u32 *dst /* [129][32] */, const u32 *proj /* [129][40] */) {
    fe d[HS_KS_ENTRIES], pre[HS_KS_ENTRIES];
    for (int j = 0; j < HS_KS_ENTRIES; j++) for (int i = 0; i < 10; i++) d[j].v[i] = proj[(size_t)j * HS_KS_PROJ_WORDS + 30 + i];
    pre[0] = d[0];
    for (int j = 1; j < HS_KS_ENTRIES; j++) fe_mul(pre[j], pre[j - 1], d[j]);
    fe inv; fe_invert(inv, pre[HS_KS_ENTRIES - 1]);
    for (int j = HS_KS_ENTRIES - 1; j >= 0; j--) {
        fe dj_inv;
        if (j) { fe_mul(dj_inv, inv, pre[j - 1]); fe_mul(inv, inv, d[j]); } else dj_inv = inv;
        fe two; fe_add(two, dj_inv, dj_inv);
        const u32 *in = proj + (size_t)j * HS_KS_PROJ_WORDS; u32 *out = dst + (size_t)j * HS_KS_ENTRY_WORDS;
        fe a, b, c;
        for (int i = 0; i < 10; i++) { a.v[i] = in[i]; b.v[i] = in[10 + i]; c.v[i] = in[20 + i]; }
        fe_mul(a, two, a); fe_mul(b, two, b); fe_mul(c, two, c);
        for (int i = 0; i < 10; i++) { out[i] = a.v[i]; out[10 + i] = b.v[i]; out[20 + i] = c.v[i]; }
        out[30] = out[31] = 0;
    }
}
// 1 when the per-entry device form and the batch form above agree (as field elements) on every entry of a window
int emu_affine_forms_agree(const uint8_t *pk32, int win) {
    u32 w[8]; load_words(w, pk32, 8); ge_p3 p; if (!ge_decompress(p, w)) return -1;
    std::vector<u32> proj((size_t)HS_KS_ENTRIES * HS_KS_PROJ_WORDS), x((size_t)HS_KS_ENTRIES * HS_KS_ENTRY_WORDS), y(x.size());
    keyset_build_window(proj.data(), p.X, p.Y, true, win);
    affine_window_batch(x.data(), proj.data());
    for (int j = 0; j < HS_KS_ENTRIES; j++) keyset_affine_entry(y.data() + (size_t)j * HS_KS_ENTRY_WORDS, proj.data() + (size_t)j * HS_KS_PROJ_WORDS);
    // ... and the device's form: groups of HS_KS_AFFINE_GROUP entries sharing one inversion (the 129th entry of this
    // window is covered as the first of a group made of it and seven copies of entry 0)
    {
        std::vector<u32> z(x.size());
        for (int g = 0; g + HS_KS_AFFINE_GROUP <= HS_KS_ENTRIES; g += HS_KS_AFFINE_GROUP)
            keyset_affine_group(z.data() + (size_t)g * HS_KS_ENTRY_WORDS, proj.data() + (size_t)g * HS_KS_PROJ_WORDS);
        std::vector<u32> tail_in((size_t)HS_KS_AFFINE_GROUP * HS_KS_PROJ_WORDS), tail_out((size_t)HS_KS_AFFINE_GROUP * HS_KS_ENTRY_WORDS);
        for (int t = 0; t < HS_KS_AFFINE_GROUP; t++)
            memcpy(tail_in.data() + (size_t)t * HS_KS_PROJ_WORDS, proj.data() + (size_t)(t == 0 ? HS_KS_ENTRIES - 1 : 0) * HS_KS_PROJ_WORDS, HS_KS_PROJ_WORDS * 4);
        keyset_affine_group(tail_out.data(), tail_in.data());
        memcpy(z.data() + (size_t)(HS_KS_ENTRIES - 1) * HS_KS_ENTRY_WORDS, tail_out.data(), HS_KS_ENTRY_WORDS * 4);
        for (int j = 0; j < HS_KS_ENTRIES; j++) for (int c = 0; c < 3; c++) {
            fe f, g; u32 fw[8], gw[8];
            for (int i = 0; i < 10; i++) { f.v[i] = z[(size_t)j * HS_KS_ENTRY_WORDS + 10 * c + i]; g.v[i] = y[(size_t)j * HS_KS_ENTRY_WORDS + 10 * c + i]; }
            fe_to_words(fw, f); fe_to_words(gw, g);
            if (memcmp(fw, gw, 32)) return 0;
        }
    }
    for (int j = 0; j < HS_KS_ENTRIES; j++) for (int c = 0; c < 3; c++) {
        fe f, g; u32 fw[8], gw[8];
        for (int i = 0; i < 10; i++) { f.v[i] = x[(size_t)j * HS_KS_ENTRY_WORDS + 10 * c + i]; g.v[i] = y[(size_t)j * HS_KS_ENTRY_WORDS + 10 * c + i]; }
        fe_to_words(fw, f); fe_to_words(gw, g);
        if (memcmp(fw, gw, 32)) return 0;
    }
    return 1;
}
size_t emu_keyset_words(size_t nkeys) { return nkeys * (size_t)HS_KS_KEY_WORDS; }
// tab: nkeys * HS_KS_KEY_WORDS words; flag: nkeys words.  Key `nkeys` is not stored here: the
// base point's own table is built by emu_keyset_build_base.
void emu_keyset_build(const uint8_t *pks, size_t nkeys, uint32_t *tab, uint32_t *flag) {
    for (size_t k = 0; k < nkeys; k++) {
        u32 w[8]; load_words(w, pks + 32 * k, 8);
        ge_p3 p; u32 fl = ge_decompress(p, w) ? HS_F_ONCURVE : 0;
        fl |= ge_is_small_order_affine(p.Y) ? HS_F_SMALL : 0;
        flag[k] = fl;
        for (int win = 0; win < HS_KS_WINDOWS; win++) {
            std::vector<u32> proj((size_t)HS_KS_ENTRIES * HS_KS_PROJ_WORDS);
            keyset_build_window(proj.data(), p.X, p.Y, true, win);
            u32 *dst = tab + k * (size_t)HS_KS_KEY_WORDS + (size_t)win * HS_KS_ENTRIES * HS_KS_ENTRY_WORDS;
            affine_window_batch(dst, proj.data());
        }
    }
}
void emu_keyset_build_base(uint32_t *tab) {
    ge_p3 b; ge_basepoint(b);
    for (int win = 0; win < HS_KS_WINDOWS; win++) {
        std::vector<u32> proj((size_t)HS_KS_ENTRIES * HS_KS_PROJ_WORDS);
        keyset_build_window(proj.data(), b.X, b.Y, false, win);
        u32 *dst = tab + (size_t)win * HS_KS_ENTRIES * HS_KS_ENTRY_WORDS;
        affine_window_batch(dst, proj.data());
    }
}
// The base point's 16-bit table from its 8-bit one (what k_base16_build + k_keyset_affine do on the device).
size_t emu_base16_words(void) { return HS_B16_WORDS; }
void emu_base16_build(const uint32_t *t8, uint32_t *b16) {
    const size_t entries = (size_t)HS_B16_WINDOWS * HS_B16_ENTRIES;
    u32 proj[HS_KS_AFFINE_GROUP * HS_KS_PROJ_WORDS];
    for (size_t g = 0; g < entries / HS_KS_AFFINE_GROUP; g++) {
        for (int t = 0; t < HS_KS_AFFINE_GROUP; t++) {
            const size_t e = g * HS_KS_AFFINE_GROUP + t;
            base16_entry_projective(proj + (size_t)t * HS_KS_PROJ_WORDS, t8, (u32)(e / HS_B16_ENTRIES), (u32)(e % HS_B16_ENTRIES));
        }
        keyset_affine_group(b16 + g * (HS_KS_AFFINE_GROUP * HS_KS_ENTRY_WORDS), proj);
    }
}
static const uint32_t *g_b16 = nullptr;
void emu_set_base16(const uint32_t *b16) { g_b16 = b16; }
void emu_keyset_build_base(uint32_t *tab);
// the table the verification paths below read: the one a test handed over, or one built here on first use
const uint32_t *emu_base16(void);
}
const uint32_t *emu_base16_fwd(void) { return emu_base16(); }
extern "C" {
const uint32_t *emu_base16(void) {
    if (!g_b16) {
        static std::vector<u32> t8((size_t)HS_KS_KEY_WORDS), b16(HS_B16_WORDS);
        emu_keyset_build_base(t8.data());
        emu_base16_build(t8.data(), b16.data());
        g_b16 = b16.data();
    }
    return g_b16;
}
// canonical (y-x, y+x, 2dxy) of a stored 128-byte entry as 96 bytes
void emu_entry_canonical(const uint32_t *entry, uint8_t *out96) {
    for (int c = 0; c < 3; c++) { fe f; for (int i = 0; i < 10; i++) f.v[i] = entry[10 * c + i]; u32 w[8]; fe_to_words(w, f); memcpy(out96 + 32 * c, w, 32); }
}
// signed radix-65536 digits, most significant first
void emu_sc_digits16(const uint8_t *s, int32_t *digits) {
    u32 w[8], r[8]; load_words(w, s, 8); sc_recode65536(r, w);
    for (int i = 0; i < 16; i++) digits[i] = sc_pop_digit16(r);
}
// The two-lane split of a key-cache verification (lane2_split_*): for lane `odd` and step t = 0..23 write
// (table: 0 = the key's, 1 = the base point's 16-bit table; window; signed digit) as three int32 each - recovered from
// the entry pointer the step returns, exactly as the kernel would dereference it.
void emu_lane2_split_trace(const uint8_t *k32, const uint8_t *s32, int odd, int32_t *out /* 24 x 3 */) {
    u32 k[8], sc[8], kw[8], sw[8];
    load_words(k, k32, 8); load_words(sc, s32, 8);
    sc_recode256(kw, k); sc_recode65536(sw, sc);
    const uintptr_t A = 0x1000000000ull, Bq = 0x9000000000ull;       // two far-apart fake table bases: only pointer arithmetic happens
    lane2_split ls;
    lane2_split_init(ls, kw, sw, odd != 0);
    for (int t = 0; t < 24; t++) {
        bool neg;
        const u32 *p = lane2_split_step(ls, t, odd != 0, (const u32 *)A, (const u32 *)Bq, neg);
        const uintptr_t a = (uintptr_t)p;
        const bool base = a >= Bq;
        const uint64_t idx = (a - (base ? Bq : A)) / (4 * HS_KS_ENTRY_WORDS);
        const uint64_t per = base ? HS_B16_ENTRIES : HS_KS_ENTRIES;
        out[3 * t] = base ? 1 : 0; out[3 * t + 1] = (int32_t)(idx / per); out[3 * t + 2] = (int32_t)(idx % per) * (neg ? -1 : 1);
    }
}
// lane_form = 1: the one-lane-per-signature core (verify_cached_lane), 0: the quad core
static int g_cached_lane_form = 0;
void emu_cached_set_lane_form(int v) { g_cached_lane_form = v; }
// lane_form 4: the "c" forms (k_cached_chain2 + k_cached_finish): two-lane chains leave R' and the flags in the workspace,
// finish lanes walk HS_CFIN_M signatures each with one inversion (cfin_store / cfin_lane, verify_cached.h)
static int g_cfin_m = HS_CFIN_M;
static void emu_verify_cached_c(const uint32_t *tab, const uint32_t *flag, const uint8_t *pks, size_t nkeys,
                                const uint8_t *msgs, size_t msg_stride, const uint32_t *key_idx, const uint8_t *sigs, size_t n,
                                uint32_t mode, uint8_t *out) {
    if (!n) return;
    const u32 np = (u32)((n + 63) / 64 * 64);
    std::vector<u32> fin((size_t)HS_CFIN_ROWS * np, 0), sw32(16 * n);
    memcpy(sw32.data(), sigs, 64 * n);
    (void)emu_base16();
    const bool strict = (mode & HS_MODE_STRICT) != 0;
    for (size_t i = 0; i < n; i++) {
        const u32 ki_raw = key_idx[i];
        const bool known = ki_raw < nkeys;
        const u32 ki = known ? ki_raw : 0;
        u32 pk[8], sig[16], msg[8], kw[8], sw[8];
        load_words(pk, pks + 32 * ki, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        bool ok = prepare_scalars(kw, sw, pk, sig, msg, true, true);
        const u32 fl = flag[ki];
        ok = ok && known && (fl & HS_F_ONCURVE);
        if (strict) ok = ok && !(fl & HS_F_SMALL);
        ge_p3 pa, pb, acc;
        lane2_split_chain(pa, tab + ki * (size_t)HS_KS_KEY_WORDS, g_b16, kw, sw, false);
        lane2_split_chain(pb, tab + ki * (size_t)HS_KS_KEY_WORDS, g_b16, kw, sw, true);
        ge_add_p3(acc, pa, pb);
        cfin_store(fin.data(), np, (u32)i, acc, ok);
    }
    const u32 m = (u32)g_cfin_m;          // signatures per finish lane: 16 (form 4) or 4 (form 5), as the kernels take it
    const u32 nl = (u32)(((n + m - 1) / m + 63) / 64 * 64);
    for (u32 L = 0; L < nl; L++) {
        const u32 bits = cfin_lane(fin.data(), np, L, nl, (u32)n, sw32.data(), strict, (int)m);
        for (u32 j = 0; j < m; j++) { const size_t sidx = L + (size_t)j * nl; if (sidx < n) out[sidx] = (bits >> j) & 1; }
    }
}
void emu_verify_cached(const uint32_t *tab, const uint32_t *flag, const uint8_t *pks, size_t nkeys, const uint32_t *btab,
                       const uint8_t *msgs, size_t msg_stride, const uint32_t *key_idx, const uint8_t *sigs, size_t n,
                       uint32_t mode, uint8_t *out) {
    if (g_cached_lane_form == 4 || g_cached_lane_form == 5) { g_cfin_m = g_cached_lane_form == 4 ? HS_CFIN_M : 4; emu_verify_cached_c(tab, flag, pks, nkeys, msgs, msg_stride, key_idx, sigs, n, mode, out); return; }
    for (size_t i = 0; i < n; i++) {
        const u32 ki = key_idx[i];
        if (ki >= nkeys) { out[i] = 0; continue; }
        u32 pk[8], sig[16], msg[8], kw[8], sw[8];
        load_words(pk, pks + 32 * ki, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        // forms 1 and 3 are the shipped kernels: s in radix 65536 against the base point's 16-bit table (emu_set_base16)
        const bool s16 = g_cached_lane_form == 1 || g_cached_lane_form == 3;
        bool ok = prepare_scalars(kw, sw, pk, sig, msg, true, s16);
        if (s16) (void)emu_base16();
        const u32 fl = flag[ki];
        ok = ok && (fl & HS_F_ONCURVE);
        if (mode & HS_MODE_STRICT) ok = ok && !(fl & HS_F_SMALL);
        const bool strict = (mode & HS_MODE_STRICT) != 0;
        bool curve;
        if (g_cached_lane_form == 1) {
            curve = verify_cached_lane(tab + ki * (size_t)HS_KS_KEY_WORDS, g_b16, kw, sw, sig, strict);
        } else if (g_cached_lane_form == 3) {      // one-lane chains + R decompressed separately (k_verify_cached_lane2r)
            ge_p3 R; u32 rf = ge_decompress(R, sig) ? HS_F_ONCURVE : 0;
            if (strict) rf |= ge_is_small_order_affine(R.Y) ? HS_F_SMALL : 0;
            ge_p3 pa, pb, acc;      // the pair's two lanes: 24 additions each (lane2_split_chain), then the join
            lane2_split_chain(pa, tab + ki * (size_t)HS_KS_KEY_WORDS, g_b16, kw, sw, false);
            lane2_split_chain(pb, tab + ki * (size_t)HS_KS_KEY_WORDS, g_b16, kw, sw, true);
            ge_add_p3(acc, pa, pb);
            curve = cached_lane_match_point(acc, R.X, R.Y, rf, strict);
        } else if (g_cached_lane_form == 2) {      // chain + R decompressed separately (the 5-wavefront kernel)
            ge_p3 R; u32 rf = ge_decompress(R, sig) ? HS_F_ONCURVE : 0;
            if (strict) rf |= ge_is_small_order_affine(R.Y) ? HS_F_SMALL : 0;
            const qv acc = cached_quad_chain(tab + ki * (size_t)HS_KS_KEY_WORDS, btab, kw, sw);
            curve = cached_quad_match_point(acc, R.X, R.Y, rf, strict);
        } else {
            curve = verify_cached_core(tab + ki * (size_t)HS_KS_KEY_WORDS, btab, kw, sw, sig, strict);
        }
        out[i] = (ok && curve) ? 1 : 0;
    }
}

// ---- batch random-linear-combination check (rlc.h): every stage on the CPU -----------------------
// Same functions the kernels of hs_rlc.hip call, with the counting sort and the bucket fold done
// serially.  Returns 1 when the combination is the identity, 0 otherwise (incl. format failures).
// segments: how many lanes share a bucket's run (exercises the partial-sum path).
// partial != null: a shard of a combination cut over several devices (hs_shard.hip): coefficients drawn for positions
// index_base + i, the result handed over as a partial record (HS_RLC_PART_WORDS words) instead of tested; returns 1 then
static int emu_rlc_core(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n,
                        const uint8_t *key32, int segments, u32 index_base, u32 *partial) {
    if (!n) return 1;
    const u32 np = (u32)((n + 63) / 64 * 64), NP = 2 * np;
    std::vector<u32> rec((size_t)(NP + 1) * HS_RLC_REC_WORDS, 0);
    std::vector<uint8_t> dig((size_t)HS_RLC_WINDOWS * NP, 128);
    unsigned long long csum[8] = {0};
    u32 fail = 0;
    rlc_view v; v.rec = rec.data(); v.dig = dig.data(); v.csum = csum; v.failed = &fail; v.np = np;
    rlc_key key; memcpy(key.w, key32, 32);
    for (size_t i = 0; i < n; i++) {
        u32 pk[8], sig[16], msg[8];
        load_words(pk, pks + 32 * i, 8); load_words(sig, sigs + 64 * i, 16); load_words(msg, msgs + msg_stride * i, 8);
        if (!rlc_point(v, (u32)i, sig)) fail = 1;
        if (!rlc_point(v, np + (u32)i, pk)) fail = 1;
        u32 c[8];
        if (!rlc_scalars(v, (u32)i, key, pk, sig, msg, c, index_base + (u32)i)) fail = 1;
        unsigned long long c64[8];
        for (int j = 0; j < 8; j++) c64[j] = c[j];
        rlc_add_csum(csum, c64);
    }
    rlc_store_identity_record(rec.data() + (size_t)NP * HS_RLC_REC_WORDS);
    if (fail && !partial) return 0;
    static std::vector<u32> btab;
    if (btab.empty()) { btab.resize(HS_KS_KEY_WORDS); emu_keyset_build_base(btab.data()); }
    std::vector<u32> wsum((size_t)(HS_RLC_WINDOWS + 1) * HS_RLC_WSUM_WORDS);
    for (int w = 0; w < HS_RLC_WINDOWS; w++) {
        std::vector<u32> sorted; std::vector<u32> off(HS_RLC_BUCKETS + 1, 0);
        std::vector<std::vector<u32>> lists(HS_RLC_BUCKETS);
        for (u32 p = 0; p < NP; p++) {
            const int d = (int)dig[(size_t)w * NP + p] - 128;
            if (d) lists[(d < 0 ? -d : d) - 1].push_back(p | (d < 0 ? HS_RLC_ID_NEG : 0u));
        }
        for (int j = 0; j < HS_RLC_BUCKETS; j++) { off[j] = (u32)sorted.size(); sorted.insert(sorted.end(), lists[j].begin(), lists[j].end()); }
        off[HS_RLC_BUCKETS] = (u32)sorted.size();
        sorted.push_back(0);
        ge_p3 bucket[HS_RLC_BUCKETS];
        for (int j = 0; j < HS_RLC_BUCKETS; j++) {
            ge_identity(bucket[j]);
            const u32 lo = off[j], len = off[j + 1] - off[j];
            for (int sgm = 0; sgm < segments; sgm++) {
                const u32 a = lo + (u32)((u64)len * sgm / segments), b = lo + (u32)((u64)len * (sgm + 1) / segments);
                ge_p3 part;
                rlc_accum_lane(part, sorted.data(), a, b, (b - a) + (sgm & 1), rec.data(), nullptr, NP);   // odd segments run one idle step
                ge_add_p3(bucket[j], bucket[j], part);
            }
        }
        ge_p3 W;
        rlc_bucket_fold(W, bucket);
        rlc_store_wsum(wsum.data() + (size_t)HS_RLC_WSUM_WORDS * w, W);
    }
    rlc_store_wsum_quad(wsum.data() + (size_t)HS_RLC_WSUM_WORDS * HS_RLC_WINDOWS, rlc_b_point_quad(btab.data(), csum));
    {   // the one-lane form of the same point must agree (projectively): compare b B + (-(b B)) == O via the final test is overkill; check affine x, y
        ge_p3 bb; rlc_b_point(bb, btab.data(), csum);
        const u32 *q = wsum.data() + (size_t)HS_RLC_WSUM_WORDS * HS_RLC_WINDOWS;
        fe qx, qy, qz; for (int i = 0; i < 10; i++) { qx.v[i] = q[i]; qy.v[i] = q[10 + i]; qz.v[i] = q[20 + i]; }
        fe l1, r1, l2, r2; fe_mul(l1, bb.X, qz); fe_mul(r1, qx, bb.Z); fe_mul(l2, bb.Y, qz); fe_mul(r2, qy, bb.Z);
        if (!fe_eq(l1, r1) || !fe_eq(l2, r2)) return -1;
    }
    if (partial) {
        rlc_store_wsum_quad(partial, rlc_final_point_quad(wsum.data()));
        partial[80] = fail;
        return 1;
    }
    const qflags f = rlc_final_quad(wsum.data());
    return f.f[0] && f.f[1] && f.f[2] && f.f[3];
}
int emu_rlc_verify(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n,
                   const uint8_t *key32, int segments) {
    return emu_rlc_core(msgs, msg_stride, pks, sigs, n, key32, segments, 0, nullptr);
}
// one combination cut over `nshards` devices (hs_shard.hip's plan: contiguous shards of ceil(n / k) rounded up to 64): the
// partial records joined and the identity tested once - must equal emu_rlc_verify over the whole batch
int emu_rlc_verify_sharded(const uint8_t *msgs, size_t msg_stride, const uint8_t *pks, const uint8_t *sigs, size_t n,
                           const uint8_t *key32, int nshards) {
    if (!n) return 1;
    const size_t S = ((n + nshards - 1) / nshards + 63) / 64 * 64;
    std::vector<u32> parts;
    u32 count = 0, failed = 0;
    for (int s = 0; s < nshards; s++) {
        const size_t lo = (size_t)s * S < n ? (size_t)s * S : n, hi = lo + S < n ? lo + S : n;
        if (hi == lo) continue;
        parts.resize((size_t)(count + 1) * HS_RLC_PART_WORDS, 0);
        u32 *p = parts.data() + (size_t)count * HS_RLC_PART_WORDS;
        if (emu_rlc_core(msgs + msg_stride * lo, msg_stride, pks + 32 * lo, sigs + 64 * lo, hi - lo, key32, 1 + s % 3, (u32)lo, p) != 1) return -1;
        failed |= p[80];
        count++;
    }
    const qflags f = rlc_join_quad(parts.data(), count);
    return (f.f[0] && f.f[1] && f.f[2] && f.f[3] && !failed) ? 1 : 0;
}
void emu_rlc_coeff(const uint8_t *key32, const uint8_t *k32, const uint8_t *s32, uint32_t index, uint8_t *z16) {
    rlc_key key; memcpy(key.w, key32, 32);
    u32 k[8], s[8], z[8]; load_words(k, k32, 8); load_words(s, s32, 8);
    rlc_coeff(z, key, k, s, index);
    memcpy(z16, z, 16);
}

// ---- wire decode and signing (wire.h) ----
int emu_wire_decode(const uint8_t *frame116, uint8_t *pk32, uint8_t *sig64) {
    u32 f[29], pk[8], sig[16]; memcpy(f, frame116, 116);
    for (int i = 0; i < 8; i++) pk[i] = 0;
    const bool ok = wire_decode_vote(pk, sig, f);
    memcpy(pk32, pk, 32); memcpy(sig64, sig, 64);
    return ok;
}
static std::vector<u32> &emu_btab() { static std::vector<u32> t; if (t.empty()) { t.resize(HS_KS_KEY_WORDS); emu_keyset_build_base(t.data()); } return t; }
void emu_sign(const uint8_t *secret64, const uint8_t *msg32, uint8_t *sig64) {
    u32 sk[16], m[8], sig[16]; memcpy(sk, secret64, 64); memcpy(m, msg32, 32);
    sign_one(sig, sk, m, emu_btab().data());
    memcpy(sig64, sig, 64);
}
void emu_public_key(const uint8_t *seed32, uint8_t *pk32) {
    u32 sd[8], pk[8]; memcpy(sd, seed32, 32);
    public_key_of(pk, sd, emu_btab().data());
    memcpy(pk32, pk, 32);
}
}
