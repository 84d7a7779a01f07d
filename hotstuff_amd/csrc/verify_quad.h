// verify_quad.h - stage 2 with FOUR lanes per signature (a "quad").
//
// The extended coordinates (X, Y, Z, T) of the running point live one per lane in lanes
// 4q..4q+3 of a wavefront.  The HWCD formulas split into two rounds of four independent
// field multiplications, so
//      doubling  = 1 vector squaring + 1 vector multiplication
//      addition  = 2 vector multiplications
// and the operand shuffles between the rounds are DPP quad_perm moves, which on CDNA4 stay
// inside the VALU (no LDS, no ds_swizzle).  Compared with one lane per signature this gives
// 4x the wavefronts per launch (a 1,024-signature batch becomes 64 wavefronts instead of
// 16), ~3x lower latency per verification and a quarter of the per-lane register state.
//
// The code is written once over a tiny vector abstraction (qv + q* primitives): on the GPU a
// qv is ONE lane's field element and the primitives are DPP moves / lane selects; under a
// plain C++ compiler a qv is all four lanes and the primitives are loops, which is how
// tests/emu validates exactly this algorithm on the CPU.
//
// Same semantics as verify_lane (verify.h); replaces dalek's
// vartime_double_scalar_mul_basepoint + point comparison (crypto/src/lib.rs:203, :218).
#pragma once
#include "verify.h"

namespace hs {

#if defined(__HIPCC__)
// ------------------------------------------------------------------ device: qv = one lane
#define HS_Q __device__ __forceinline__
typedef fe qv;
HS_Q u32 q_lane() { return threadIdx.x & 3u; }
template <int P0, int P1, int P2, int P3>
HS_Q qv q_perm(const qv &a) {       // lane i <- lane P_i of the same quad
    qv r;
#pragma unroll
    for (int i = 0; i < 10; i++)
        r.v[i] = (u32)__builtin_amdgcn_mov_dpp((int)a.v[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
    return r;
}
template <int MASK>
HS_Q qv q_sel(const qv &a, const qv &b) {   // lane i takes b iff bit i of MASK
#ifdef HS_QSEL_CNDMASK
    const bool pick = (MASK >> q_lane()) & 1;
    qv r; fe_select(r, a, b, pick); return r;
#else
    // a mask in a register, (b & m) | (a & ~m) = one v_bfi_b32 per limb; the compiler's own form is a v_cndmask_b32_e64
    // on an SGPR-pair mask (4.9 cycles against ~3, as for q_comb's `keep`).  Measured A/B (-DHS_QSEL_CNDMASK): throughput
    // equal within noise, a lone wavefront's chain 1 % shorter (cold QC p50 0.383 against 0.387 ms)
    u32 m = ((MASK >> q_lane()) & 1) ? 0xffffffffu : 0u;
    asm("" : "+v"(m));
    qv r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (b.v[i] & m) | (a.v[i] & ~m);
    return r;
#endif
}
HS_Q qv q_mul(const qv &f, const qv &g) { qv r; fe_mul(r, f, g); return r; }
HS_Q qv q_sq(const qv &f) { qv r; fe_sq(r, f); return r; }
HS_Q qv q_add(const qv &a, const qv &b) { qv r; fe_add(r, a, b); return r; }
HS_Q qv q_sub(const qv &a, const qv &b) { qv r; fe_sub(r, a, b); return r; }
HS_Q qv q_carry(const qv &a) { qv r = a; fe_carry(r); return r; }
HS_Q qv q_pick(const qv &a, const qv &b, bool pick_b) { qv r; fe_select(r, a, b, pick_b); return r; }
// The workhorse between two multiplication rounds: lane i <- a[PA_i] (+|-) b[PB_i], where the lanes
// in NEG subtract (a + 2p - b; b reduced), the lanes in ZERO take a[PA_i] alone and the others add.
// Every lane runs the same three instructions per limb - the sign is data: b ^ mask + (mask & (2p + 1))
// is 2p - b (mod 2^32) when mask is all ones and b when it is zero - instead of computing every
// combination in every lane and selecting (which cost the doubling 250 of its 503 instructions).
template <int PA, int PB, int NEG, int ZERO>
HS_Q qv q_comb(const qv &a, const qv &b) {
    const u32 lane = q_lane();
    u32 neg = ((NEG >> lane) & 1) ? 0xffffffffu : 0u;
    u32 keep = ((ZERO >> lane) & 1) ? 0u : 0xffffffffu;
    // keep it a mask in a register: known to be 0 / ~0 per lane, the compiler turns `qb & keep` into a v_cndmask_b32_e64
    // on an SGPR-pair mask (4.9 cycles per wave64 instruction at two wavefronts per SIMD against 3.0 for the v_and_b32)
    if (ZERO) asm("" : "+v"(keep));
    // NEVER write this as `pa + bias - qb`: the compiler folds the DPP move into the subtraction as
    // v_subrev_u32_dpp, and on gfx950 that instruction permutes its OTHER operand (measured:
    // tools/dpp_subrev_check.hip, profiles/r1_dpp_subrev.txt) - silently wrong lanes.  With every lane
    // negating, the mask is kept opaque so that the xor/add form cannot be turned back into a subtraction;
    // __graft_entry__.build_engine() refuses a code object that contains a v_sub*rev*_dpp.
    if (NEG == 0xf) asm volatile("" : "+v"(neg));
    qv r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const u32 bias = i == 0 ? 0x7ffffdau : ((i & 1) ? 0x3fffffeu : 0x7fffffeu);
        const u32 pa = (u32)__builtin_amdgcn_mov_dpp((int)a.v[i], PA, 0xf, 0xf, true);
        u32 qb = (u32)__builtin_amdgcn_mov_dpp((int)b.v[i], PB, 0xf, 0xf, true);
        if (ZERO) qb &= keep;
        if (NEG) r.v[i] = pa + (qb ^ neg) + (neg & (bias + 1));
        else r.v[i] = pa + qb;
    }
    return r;
}
// a cached entry (Y-X, Y+X, 2dT, 2Z) or, where `negate` (uniform over the quad), its negative
// (Y+X, Y-X, 2p - 2dT, 2Z): swap lanes 0/1 and negate lane 2, again with the sign as data
HS_Q qv q_cached_signed(const qv &e, bool negate) {
    u32 m = negate ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));              // keep it a mask: (sw & m) | (e & ~m) is one v_bfi_b32, a VCC select costs 4x
    const u32 n2 = q_lane() == 2 ? m : 0u;
    qv r;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const u32 bias = i == 0 ? 0x7ffffdau : ((i & 1) ? 0x3fffffeu : 0x7fffffeu);
        const u32 sw = (u32)__builtin_amdgcn_mov_dpp((int)e.v[i], 1 | (0 << 2) | (2 << 4) | (3 << 6), 0xf, 0xf, true);
        const u32 x = (sw & m) | (e.v[i] & ~m);
        r.v[i] = (x ^ n2) + (n2 & (bias + 1));
    }
    return r;
}
HS_Q qv q_lanes(const fe &c0, const fe &c1, const fe &c2, const fe &c3) {   // per-lane constant
    const u32 l = q_lane();
    qv r; fe_select(r, c0, c1, l == 1); fe_select(r, r, c2, l == 2); fe_select(r, r, c3, l == 3); return r;
}
// entry `idx` of a table laid out [idx][lane] (LDS or global)
HS_Q qv q_load(const fe (*tab)[4], int idx) { return tab[idx][q_lane()]; }
// per-lane equality of canonical values -> one flag per lane
struct qflags { bool f; };
HS_Q qflags q_eq(const qv &a, const qv &b_reduced) { qflags r; r.f = fe_eq(a, b_reduced); return r; }
// lanes 0, 1: a == b; lanes 2, 3: b != 0
HS_Q qflags q_eq_nz(const qv &a, const qv &b_reduced) {
    qflags r; const bool eq = fe_eq(a, b_reduced), nz = !fe_iszero(b_reduced);
    r.f = q_lane() < 2 ? eq : nz; return r;
}
#else
// ------------------------------------------------------------------ host: qv = four lanes
#define HS_Q inline
struct qv { fe l[4]; };
template <int P0, int P1, int P2, int P3>
HS_Q qv q_perm(const qv &a) { qv r; r.l[0] = a.l[P0]; r.l[1] = a.l[P1]; r.l[2] = a.l[P2]; r.l[3] = a.l[P3]; return r; }
template <int MASK>
HS_Q qv q_sel(const qv &a, const qv &b) { qv r; for (int i = 0; i < 4; i++) r.l[i] = ((MASK >> i) & 1) ? b.l[i] : a.l[i]; return r; }
HS_Q qv q_mul(const qv &f, const qv &g) { qv r; for (int i = 0; i < 4; i++) fe_mul(r.l[i], f.l[i], g.l[i]); return r; }
HS_Q qv q_sq(const qv &f) { qv r; for (int i = 0; i < 4; i++) fe_sq(r.l[i], f.l[i]); return r; }
HS_Q qv q_add(const qv &a, const qv &b) { qv r; for (int i = 0; i < 4; i++) fe_add(r.l[i], a.l[i], b.l[i]); return r; }
HS_Q qv q_sub(const qv &a, const qv &b) { qv r; for (int i = 0; i < 4; i++) fe_sub(r.l[i], a.l[i], b.l[i]); return r; }
HS_Q qv q_carry(const qv &a) { qv r = a; for (int i = 0; i < 4; i++) fe_carry(r.l[i]); return r; }
HS_Q qv q_pick(const qv &a, const qv &b, bool pick_b) { return pick_b ? b : a; }
template <int PA, int PB, int NEG, int ZERO>
HS_Q qv q_comb(const qv &a, const qv &b) {
    qv r;
    for (int i = 0; i < 4; i++) {
        const fe &x = a.l[(PA >> (2 * i)) & 3], &y = b.l[(PB >> (2 * i)) & 3];
        if ((ZERO >> i) & 1) r.l[i] = x;
        else if ((NEG >> i) & 1) fe_sub(r.l[i], x, y);
        else fe_add(r.l[i], x, y);
    }
    return r;
}
HS_Q qv q_cached_signed(const qv &e, bool negate) {
    if (!negate) return e;
    qv r; fe z; fe_0(z);
    r.l[0] = e.l[1]; r.l[1] = e.l[0]; fe_sub(r.l[2], z, e.l[2]); r.l[3] = e.l[3];
    return r;
}
HS_Q qv q_lanes(const fe &c0, const fe &c1, const fe &c2, const fe &c3) { qv r; r.l[0] = c0; r.l[1] = c1; r.l[2] = c2; r.l[3] = c3; return r; }
HS_Q qv q_load(const fe (*tab)[4], int idx) { qv r; for (int i = 0; i < 4; i++) r.l[i] = tab[idx][i]; return r; }
struct qflags { bool f[4]; };
HS_Q qflags q_eq(const qv &a, const qv &b_reduced) { qflags r; for (int i = 0; i < 4; i++) r.f[i] = fe_eq(a.l[i], b_reduced.l[i]); return r; }
HS_Q qflags q_eq_nz(const qv &a, const qv &b_reduced) {
    qflags r; for (int i = 0; i < 4; i++) r.f[i] = i < 2 ? fe_eq(a.l[i], b_reduced.l[i]) : !fe_iszero(b_reduced.l[i]); return r;
}
#endif

HS_HD fe fe_small(u32 c) { fe r; fe_0(r); r.v[0] = c; return r; }

// A stored table entry: the AFFINE Niels point (y-x, y+x, 2dxy) in 32 words = 128 bytes = one cache line
// (verify_cached.h builds the tables).  As a quad operand it is (y-x, y+x, 2dxy, 2): lanes 0-2 read their ten
// limbs, lane 3 takes the constant 2Z = 2.
#define HS_ENTRY_WORDS 32
#if defined(__HIPCC__)
HS_Q qv q_load_entry(const u32 *entry) {
    const u32 l = q_lane();
    const uint2 *p = reinterpret_cast<const uint2 *>(entry + 10 * (l < 3 ? l : 2));
    u32 m = l < 3 ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));              // a mask (v_and / v_bfi), not a VCC select
    qv r;
#pragma unroll
    for (int i = 0; i < 5; i++) { const uint2 v = p[i]; r.v[2 * i] = v.x & m; r.v[2 * i + 1] = v.y & m; }
    r.v[0] |= 2u & ~m;
    return r;
}
#else
HS_Q qv q_load_entry(const u32 *entry) {
    qv r; for (int l = 0; l < 3; l++) for (int i = 0; i < 10; i++) r.l[l].v[i] = entry[10 * l + i];
    r.l[3] = fe_small(2); return r;
}
#endif

// The per-signature table j*(-A), j = 1..8 (digit 0 takes the identity from a constant).  On the
// GPU it lives in LDS, one 20,480-byte slab per wavefront laid out [entry][limb][lane]: whatever
// entry each quad picks, the 32 lanes of a half-wave hit 32 different banks (bank = lane mod 32),
// so lookups are conflict-free - and the table never touches HBM (in private memory it was
// 368 B x 4 lanes of write-back traffic per signature, profiles/r1_pmc_traffic.json).
// Eight entries, not nine: 8 x 20,480 B is exactly a CU's 160 KB of LDS, i.e. two wavefronts per
// SIMD - which is also what the kernel's ~250 VGPRs allow; with the identity stored as a ninth
// entry (23,040 B) only six wavefronts fit a CU.
#define HS_QTAB_ENTRIES 8
// W-bit signed windows need 2^(W-1) entries.  W = 4: 8 entries, and with the two tables of the half-size form 40,960 B of
// LDS per wavefront - one wavefront per SIMD (the latency kernel: a single batch of up to 16,384 signatures is at most one
// wavefront per SIMD anyway).  W = 3: 4 entries, 20,480 B for both tables - two wavefronts per SIMD, which is what the
// kernels that put several batches into one launch live on; 45 windows of three doublings instead of 34 of four cost
// 3 - 4 % more instructions per verification (14 more additions, 8 fewer table entries to build).
#define HS_QTAB_ENTRIES_OF(W) (1 << ((W) - 1))
#if defined(__HIPCC__)
#define HS_QTAB_WORDS_OF(W) (HS_QTAB_ENTRIES_OF(W) * 10 * 64)
#define HS_QTAB_WORDS HS_QTAB_WORDS_OF(4)
struct qtab { u32 *lds; };      // this wavefront's slab
HS_Q void qtab_store(const qtab &t, int j, const qv &e) {
    u32 *p = t.lds + (size_t)j * 640 + (threadIdx.x & 63u);
#pragma unroll
    for (int i = 0; i < 10; i++) p[i * 64] = e.v[i];
}
HS_Q qv qtab_load(const qtab &t, int j) {
    const u32 *p = t.lds + (size_t)j * 640 + (threadIdx.x & 63u);
    qv r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = p[i * 64];
    return r;
}
#else
struct qtab { qv e[HS_QTAB_ENTRIES]; };
HS_Q void qtab_store(qtab &t, int j, const qv &e) { t.e[j] = e; }
HS_Q qv qtab_load(const qtab &t, int j) { return t.e[j]; }
#endif
// |digit| * (-A) in cached form: table entry |digit| - 1, or the identity (1, 1, 0, 2) for digit 0.
// The choice is per quad, so it is a mask, not a branch: limbs above the lowest are e & m, the
// lowest is e0 & m | id0 & ~m.
HS_Q qv qtab_pick(const qtab &t, int mag) {
    const qv e = qtab_load(t, mag ? mag - 1 : 0);
#if defined(__HIPCC__)
    u32 m = mag ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));              // keep it a mask (v_and / v_bfi), not a VCC select
    const u32 l = q_lane();
    const u32 id0 = l == 2 ? 0u : (l == 3 ? 2u : 1u);
    qv r;
    r.v[0] = (e.v[0] & m) | (id0 & ~m);
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = e.v[i] & m;
    return r;
#else
    return mag ? e : q_lanes(fe_small(1), fe_small(1), fe_small(0), fe_small(2));
#endif
}

#define HS_QP(p0, p1, p2, p3) ((p0) | ((p1) << 2) | ((p2) << 4) | ((p3) << 6))
// (X, Y, Z, T) -> (Y-X, Y+X, T, Z): the left operands of an addition's first round    (<= 3x)
HS_Q qv q_add_lhs(const qv &p) {
    return q_comb<HS_QP(1, 1, 3, 2), HS_QP(0, 0, 0, 0), 0b0001, 0b1100>(p, p);
}
// cached form (Y-X, Y+X, 2dT, 2Z) of an extended point: one vector multiplication by constants
HS_Q qv q_to_cached(const qv &p) {
    const qv k = q_lanes(fe_small(1), fe_small(1), fe_d2(), fe_small(2));
    return q_mul(q_add_lhs(p), k);
}
// negate a cached entry: swap lanes 0/1, negate lane 2
HS_Q qv q_cached_neg(const qv &e) { return q_cached_signed(e, true); }
// p + e  (e cached, already sign-adjusted) -> extended
HS_Q qv q_add_cached(const qv &p, const qv &e) {
    const qv m = q_mul(q_add_lhs(p), e);              // (A, B, C, D), reduced
    // E = B - A, H = B + A, F = D - C, G = D + C
    const qv l = q_comb<HS_QP(1, 3, 3, 1), HS_QP(0, 2, 2, 0), 0b1101, 0>(m, m);   // (E, G, F, E)   <= 3x
    const qv r = q_comb<HS_QP(3, 1, 3, 1), HS_QP(2, 0, 2, 0), 0b0001, 0>(m, m);   // (F, H, G, H)   <= 3x
    return q_mul(l, r);                                                             // (EF, GH, FG, EH)
}
// 2p -> extended (input T unused)
HS_Q qv q_dbl(const qv &p) {
    const qv a = q_comb<HS_QP(0, 1, 2, 0), HS_QP(0, 0, 0, 1), 0, 0b0111>(p, p);    // (X, Y, Z, X+Y)
    const qv s = q_sq(a);                                                           // (XX, YY, ZZ, AA), reduced
    // t = (S, D, 2ZZ, AA) with S = YY + XX, D = YY - XX; one carry for all four
    const qv t = q_carry(q_comb<HS_QP(1, 1, 2, 3), HS_QP(0, 0, 2, 0), 0b0010, 0b1000>(s, s));
    // w = (E, F, F, E) with E = AA - S, F = 2ZZ - D                                  (3x)
    const qv w = q_comb<HS_QP(3, 2, 2, 3), HS_QP(0, 1, 1, 0), 0b1111, 0>(t, t);
    const qv l = q_sel<0b0010>(w, t);                                               // (E, D, F, E)
    const qv r = q_sel<0b0001>(q_perm<0, 0, 1, 0>(t), q_perm<1, 1, 1, 1>(w));       // (F, S, D, S)
    return q_mul(l, r);                                                             // (EF, DS, FD, ES)
}

// The recoded scalars of the half-size form (halfscalar.h) for one signature: |u| and v in signed radix 16, w = v s mod L
// in signed radix 65536, how many windows u and v need, and the sign of u.
#ifdef HS_FORCE_NWIN
#define HS_NWIN_OF(meta) (HS_FORCE_NWIN)     // timing experiments only
#else
#define HS_NWIN_OF(meta) ((int)wave_max_u32((meta) & 0xffu))
#endif
// The recoded scalars of the half-size form (halfscalar.h) for one signature: |u| and v in signed radix 2^W (W = 4: 64 digits;
// W = 3: 85 digits, kept left-aligned), w = v s mod L in signed radix 65536, how many windows u and v need, the sign of u.
struct half_scalars { u32 ru[8], rv[8], rw[8]; u32 meta; };      // meta: bits 0-7 windows (1 ... 85), bit 8: u < 0

// j * P, cached, j = 1 ... 2^(W-1) at index j - 1, for the affine point (x, y)
template <int W>
HS_Q void qtab_build(qtab &tab, const fe &x, const fe &y) {
    qv p = q_mul(q_lanes(x, y, fe_small(1), x), q_lanes(fe_small(1), fe_small(1), fe_small(1), y));       // (x, y, 1, xy)
    const qv a1 = q_to_cached(p);
    qtab_store(tab, 0, a1);
#pragma nounroll
    for (int j = 2; j <= HS_QTAB_ENTRIES_OF(W); j++) {
        p = q_add_cached(p, a1);
        qtab_store(tab, j - 1, q_to_cached(p));
    }
}
// acc == O as a point: X == 0 (lane 0), Y == Z (lane 1), Z != 0 (lanes 2, 3; fail closed on a zeroed accumulator, see
// ge_match_point in verify.h).  The caller ANDs the four flags.
HS_Q qflags q_is_identity(const qv &acc) {
    const qv z = q_perm<2, 2, 2, 2>(acc);
    const qv zero = q_lanes(fe_small(0), fe_small(0), fe_small(0), fe_small(0));
    const qv rhs = q_sel<0b0001>(z, zero);                // (0, Z, Z, Z)
    const qv lhs = q_carry(q_sel<0b1100>(acc, z));        // (X, Y, Z, Z), reduced
    return q_eq_nz(rhs, lhs);
}
template <int W> HS_HD int sc_pop_window(u32 w[8]) { return W == 4 ? sc_pop_digit(w) : sc_pop_digit3(w); }

// The quad stage-2 core in the half-size form (halfscalar.h):  [w]B + [u](-A) + [v](-R) == O.  ax, ay, rx, ry (affine A
// and R) and the recoded scalars are uniform across the four lanes of a quad; `nwin` (windows of W doublings) is uniform
// over the wavefront.  atab / rtab: the per-signature tables j * (+-A) and j * (-R), j = 1 ... 2^(W-1), in LDS.
// b16: the base point's sixteen 16-bit-window tables (verify_cached.h): T[i][j] = j 2^(16 i) B, 32,769 affine Niels entries
// of 128 bytes per window - [w]B costs 16 additions and no doublings; they come after the window loop (absolute multiples).
// Returns, per lane, whether this lane's part of the identity test passed; the caller ANDs the four lanes of the quad
// with the signature-level flags.
template <int W>
HS_Q qflags verify_quad_core(const fe &ax, const fe &ay, const fe &rx, const fe &ry, half_scalars &hs, int nwin,
                             const u32 *b16, qtab &atab, qtab &rtab) {
    {
        // (+-)A: -A when u > 0 (the digits are those of |u|), and -R
        fe nax, sx, nrx;
        fe_neg(nax, ax);
        fe_select(sx, nax, ax, ((hs.meta >> 8) & 1) != 0);
        qtab_build<W>(atab, sx, ay);
        fe_neg(nrx, rx);
        qtab_build<W>(rtab, nrx, ry);
    }
    // the recoded scalars are popped from the top: drop the windows above the wavefront's count (all zero digits)
#pragma nounroll
    for (int t = nwin; t < (W == 4 ? 64 : HS_RADIX8_DIGITS); t++) { (void)sc_pop_window<W>(hs.ru); (void)sc_pop_window<W>(hs.rv); }
    qv acc = q_lanes(fe_small(0), fe_small(1), fe_small(1), fe_small(0));
    // The entries a window adds are FETCHED ONE WINDOW AHEAD: a lone wavefront (the latency regime) has nothing else to
    // cover an LDS or memory round trip with, and fetched at the point of use they kept it waiting 37 % of its cycles
    // (profiles/r3_pmc_sq.json).  The digits are known from the start, so window w - 1's two entries are requested right
    // after window w's additions have been issued and are consumed behind the next W doublings.
    int du = sc_pop_window<W>(hs.ru), dv = sc_pop_window<W>(hs.rv);
    qv ea = qtab_pick(atab, du < 0 ? -du : du), er = qtab_pick(rtab, dv < 0 ? -dv : dv);
#pragma nounroll
    for (int w = nwin - 1; w >= 0; w--) {
        // the accumulator is still the identity in the very first window: its doublings are skipped
        const int nd = w == nwin - 1 ? 0 : W;
#pragma nounroll
        for (int d = 0; d < nd; d++) acc = q_dbl(acc);
        const qv sa = q_cached_signed(ea, du < 0), sr = q_cached_signed(er, dv < 0);
        if (w > 0) {
            du = sc_pop_window<W>(hs.ru); dv = sc_pop_window<W>(hs.rv);
            ea = qtab_pick(atab, du < 0 ? -du : du); er = qtab_pick(rtab, dv < 0 ? -dv : dv);
        }
        acc = q_add_cached(acc, sa);
        acc = q_add_cached(acc, sr);
    }
    // [w]B: the same one-ahead fetch for the sixteen gathers from the base point's tables (these come from L2 / MALL / HBM)
    int ds = sc_pop_digit16(hs.rw);
    qv eb = q_load_entry(b16 + ((size_t)15 * 32769 + (size_t)(ds < 0 ? -ds : ds)) * HS_ENTRY_WORDS);
#pragma nounroll
    for (int j = 15; j >= 0; j--) {
        const qv sb = q_cached_signed(eb, ds < 0);
        if (j > 0) {
            ds = sc_pop_digit16(hs.rw);
            eb = q_load_entry(b16 + ((size_t)(j - 1) * 32769 + (size_t)(ds < 0 ? -ds : ds)) * HS_ENTRY_WORDS);
        }
        acc = q_add_cached(acc, sb);
    }
    return q_is_identity(acc);
}

// ---- eight lanes per signature: one quad per per-signature table (the latency form for n <= 8,192) ----------------------
// In verify_quad_core a quad runs TWO additions per window (one from each table), builds both tables and adds all sixteen
// base-point entries.  With a batch this small half the chip's SIMDs idle anyway, so the signature gets a second quad:
// quad 0 (role 0) evaluates  [u](+-A) + sum_{j < 8} T[j][w_j],  quad 1 (role 1)  [v](-R) + sum_{j >= 8} T[j][w_j]  - the same
// instruction stream on different data: ONE addition per window, ONE table of eight entries (20,480 B of LDS per wavefront
// again), eight base-point additions - and the two partial points are added at the end (quad_half_join).  33 windows of
// (4 doublings + 1 addition) + 7 + 8 + 1.5 additions on the chain instead of 33 x (4 + 2) + 14 + 16.
// (x, y): this role's affine point, sign-adjusted; rd: this role's scalar recoded radix 16 (popped from the top); rwl: this
// role's eight radix-65536 digits of w, left-aligned (popped from the top); jbase: 0 or 8.
// The walk over a table that is ALREADY BUILT: `flip` negates every table entry picked (uniform over the quad), so that a table of
// j (-A) serves u < 0 as well - which is what lets the table be built before the scalars are known (the split form below).
HS_Q qv quad_half_walk(u32 rd[8], u32 rwl[8], u32 jbase, int nwin, bool flip, const u32 *b16, qtab &tab) {
#pragma nounroll
    for (int t = nwin; t < 64; t++) (void)sc_pop_digit(rd);
    qv acc = q_lanes(fe_small(0), fe_small(1), fe_small(1), fe_small(0));
    int d = sc_pop_digit(rd);
    qv e = qtab_pick(tab, d < 0 ? -d : d);
#pragma nounroll
    for (int w = nwin - 1; w >= 0; w--) {
        const int nd = w == nwin - 1 ? 0 : 4;
#pragma nounroll
        for (int k = 0; k < nd; k++) acc = q_dbl(acc);
        const qv se = q_cached_signed(e, (d < 0) != flip);
        if (w > 0) { d = sc_pop_digit(rd); e = qtab_pick(tab, d < 0 ? -d : d); }
        acc = q_add_cached(acc, se);
    }
    int ds = sc_pop_digit16(rwl);
    qv eb = q_load_entry(b16 + ((size_t)(jbase + 7) * 32769 + (size_t)(ds < 0 ? -ds : ds)) * HS_ENTRY_WORDS);
#pragma nounroll
    for (int j = 7; j >= 0; j--) {
        const qv sb = q_cached_signed(eb, ds < 0);
        if (j > 0) {
            ds = sc_pop_digit16(rwl);
            eb = q_load_entry(b16 + ((size_t)(jbase + (u32)j - 1) * 32769 + (size_t)(ds < 0 ? -ds : ds)) * HS_ENTRY_WORDS);
        }
        acc = q_add_cached(acc, sb);
    }
    return acc;
}
HS_Q qv quad_half_chain(const fe &x, const fe &y, u32 rd[8], u32 rwl[8], u32 jbase, int nwin, const u32 *b16, qtab &tab) {
    qtab_build<4>(tab, x, y);
    return quad_half_walk(rd, rwl, jbase, nwin, false, b16, tab);
}
// the two roles' partial points added, and the identity test
HS_Q qflags quad_half_join(const qv &mine, const qv &other) { return q_is_identity(q_add_cached(mine, q_to_cached(other))); }
// this role's operands out of the signature's recoded scalars
HS_HD void quad_half_operands(u32 rd[8], u32 rwl[8], const half_scalars &hs, bool role1) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
        rd[t] = role1 ? hs.rv[t] : hs.ru[t];
        rwl[t] = role1 ? hs.rw[t] : (t >= 4 ? hs.rw[t - 4] : 0u);       // digits 15..8, or digits 7..0 moved to the top
    }
}

// The scalar half of stage 1 (hash, range check, recodings) - cheap, every lane of the quad
// computes it redundantly so no lane diverges.  Returns s < L.
// k is recoded in signed radix 16 (cold path: 4-bit windows on -A) or, with k_radix256, in
// signed radix 256 (cached path: 8-bit windows on both points).
// s is recoded in signed radix 256 or, with s_radix65536, in signed radix 65536 (the base point's 16-bit table).
HS_HD bool prepare_scalars(u32 kw[8], u32 sw[8], const u32 *pk, const u32 *sig, const u32 *msg, bool k_radix256 = false,
                           bool s_radix65536 = false) {
    u32 in[24], h[16], k[8], s[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { in[j] = sig[j]; in[8 + j] = pk[j]; in[16 + j] = msg[j]; s[j] = sig[8 + j]; }
    sha512_96(h, in);
    sc_reduce512(k, h);
    if (k_radix256) sc_recode256(kw, k); else sc_recode16(kw, k);
    if (s_radix65536) sc_recode65536(sw, s); else sc_recode256(sw, s);
    return sc_is_canonical(s);
}

// The same for the cold path's half-size form (halfscalar.h): hash, range check, the short pair (u, v) for k, w = v s mod L,
// the three recodings, the window count.  Returns s < L.
template <int W>
HS_HD bool prepare_scalars_half(half_scalars &out, const u32 *pk, const u32 *sig, const u32 *msg) {
    u32 in[24], h[16], k[8], s[8], w[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { in[j] = sig[j]; in[8 + j] = pk[j]; in[16 + j] = msg[j]; s[j] = sig[8 + j]; }
    sha512_96(h, in);
    sc_reduce512(k, h);
    half_pair hp;
    half_scalar(hp, k);
    sc_mul(w, hp.v, s);
    int nu, nv;
    if (W == 4) {
        sc_recode16(out.ru, hp.u);
        sc_recode16(out.rv, hp.v);
        nu = sc_recode16_len(out.ru); nv = sc_recode16_len(out.rv);
    } else {
        sc_recode8(out.ru, hp.u);
        sc_recode8(out.rv, hp.v);
        nu = sc_recode8_len(out.ru); nv = sc_recode8_len(out.rv);
        sc_align8(out.ru); sc_align8(out.rv);
    }
    sc_recode65536(out.rw, w);
    out.meta = (u32)(nu > nv ? nu : nv) | (hp.u_neg << 8);
    return sc_is_canonical(s);
}

#if defined(__HIPCC__)
// Fused stage 1 + stage 2 for one quad: lane 1 decompresses R, the other lanes decompress A
// (one shared instruction stream), everybody hashes, DPP broadcasts hand A to the whole
// quad, then the core.  One launch per batch and no workspace traffic.
template <int W>
HS_Q bool verify_quad_fused(const u32 *pk, const u32 *sig, const u32 *msg, u32 mode, const u32 *b16, qtab &atab, qtab &rtab) {
    const u32 lane = q_lane();
    u32 enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) enc[j] = lane == 1 ? sig[j] : pk[j];
    ge_p3 pt;
    u32 fl = ge_decompress(pt, enc) ? HS_F_ONCURVE : 0;
    if (mode & HS_MODE_STRICT) fl |= ge_is_small_order_affine(pt.Y) ? HS_F_SMALL : 0;
    const u32 fa = (u32)__builtin_amdgcn_mov_dpp((int)fl, 0x00, 0xf, 0xf, true);   // quad_perm [0,0,0,0]
    const u32 fr = (u32)__builtin_amdgcn_mov_dpp((int)fl, 0x55, 0xf, 0xf, true);   // quad_perm [1,1,1,1]
    half_scalars hs;
    bool ok = prepare_scalars_half<W>(hs, pk, sig, msg);
    ok = ok && (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE);
    if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);
    const fe ax = q_perm<0, 0, 0, 0>(pt.X), ay = q_perm<0, 0, 0, 0>(pt.Y);
    const fe rx = q_perm<1, 1, 1, 1>(pt.X), ry = q_perm<1, 1, 1, 1>(pt.Y);
    const int nwin = HS_NWIN_OF(hs.meta);
    const qflags f = verify_quad_core<W>(ax, ay, rx, ry, hs, nwin, b16, atab, rtab);
    return f.f && ok;
}

// The eight-lane latency form (quad_half_chain): lanes 0-3 of an octet take role 0 (A), lanes 4-7 role 1 (R).  Every lane
// decompresses its role's point and computes the scalars; the partial points cross the octet with one lane exchange.
HS_Q bool verify_oct_fused(const u32 *pk, const u32 *sig, const u32 *msg, u32 mode, const u32 *b16, qtab &tab) {
    const bool role1 = (threadIdx.x & 4u) != 0;
    u32 enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) enc[j] = role1 ? sig[j] : pk[j];
    ge_p3 pt;
    u32 fl = ge_decompress(pt, enc) ? HS_F_ONCURVE : 0;
    if (mode & HS_MODE_STRICT) fl |= ge_is_small_order_affine(pt.Y) ? HS_F_SMALL : 0;
    const u32 fo = (u32)__shfl_xor((int)fl, 4, 64);                  // the other role's point flags
    half_scalars hs;
    bool ok = prepare_scalars_half<4>(hs, pk, sig, msg);
    ok = ok && (fl & HS_F_ONCURVE) && (fo & HS_F_ONCURVE);
    if (mode & HS_MODE_STRICT) ok = ok && !(fl & HS_F_SMALL) && !(fo & HS_F_SMALL);
    // role 0: -A, or +A when u < 0 (the digits are those of |u|); role 1: -R
    fe nx, sx;
    fe_neg(nx, pt.X);
    fe_select(sx, nx, pt.X, !role1 && ((hs.meta >> 8) & 1) != 0);
    u32 rd[8], rwl[8];
    quad_half_operands(rd, rwl, hs, role1);
    const int nwin = HS_NWIN_OF(hs.meta);
    const qv mine = quad_half_chain(sx, pt.Y, rd, rwl, role1 ? 8u : 0u, nwin, b16, tab);
    qv other;
#pragma unroll
    for (int i = 0; i < 10; i++) other.v[i] = (u32)__shfl_xor((int)mine.v[i], 4, 64);
    const qflags f = quad_half_join(mine, other);
    return f.f && ok;
}

// The eight-lane form SPLIT over two wavefronts of one workgroup (k_verify_fused8h).  On verify_oct_fused's chain the hash
// (13 us on a lone wavefront) and the lattice reduction (38 us) come before the square root (70 us) and the table (13 us)
// although neither needs the other: the table of j (-P) depends on the point alone once the sign of u is applied at the
// lookup (quad_half_walk's `flip`), and the scalars depend on the bytes alone.  So a HELPER wavefront computes this octet-
// wavefront's scalars - lane for lane what the main wavefront's lanes would have computed - while the main one decompresses and
// builds its tables; they meet at one barrier, the scalars cross in LDS (17 words per lane), and the main wavefront walks.
// A lone call has the SIMDs for it: 6,667 signatures are 834 + 834 wavefronts on 1,024 SIMDs, and two wavefronts on one SIMD
// barely slow each other (profiles/r1_ubench_valu.txt: 11.4 cycles per multiply-add alone, 5.9 per SIMD for two).
#define HS_OCT_SC_WORDS 17
// helper wavefront: the scalars of lane `lane`'s signature and role -> LDS
HS_Q void oct_split_scalars(u32 *s_sc, u32 lane, const u32 *pk, const u32 *sig, const u32 *msg) {
    half_scalars hs;
    const bool ok = prepare_scalars_half<4>(hs, pk, sig, msg);
    u32 rd[8], rwl[8];
    quad_half_operands(rd, rwl, hs, (lane & 4u) != 0);
    const u32 nwin = (u32)HS_NWIN_OF(hs.meta);
#pragma unroll
    for (int t = 0; t < 8; t++) { s_sc[t * 64 + lane] = rd[t]; s_sc[(8 + t) * 64 + lane] = rwl[t]; }
    s_sc[16 * 64 + lane] = nwin | (hs.meta & 0x100u) | (ok ? 0x200u : 0u);
}
// main wavefront, before the barrier: this role's point and its table j (-P), j = 1 ... 8; returns the point's flags
HS_Q u32 oct_split_point(const u32 *pk, const u32 *sig, u32 mode, qtab &tab) {
    const bool role1 = (threadIdx.x & 4u) != 0;
    u32 enc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) enc[j] = role1 ? sig[j] : pk[j];
    ge_p3 pt;
    u32 fl = ge_decompress(pt, enc) ? HS_F_ONCURVE : 0;
    if (mode & HS_MODE_STRICT) fl |= ge_is_small_order_affine(pt.Y) ? HS_F_SMALL : 0;
    fe nx;
    fe_neg(nx, pt.X);
    qtab_build<4>(tab, nx, pt.Y);
    return fl;
}
// main wavefront, after the barrier
HS_Q bool oct_split_walk(const u32 *s_sc, u32 fl, u32 mode, const u32 *b16, qtab &tab) {
    const u32 lane = threadIdx.x & 63u;
    const bool role1 = (lane & 4u) != 0;
    u32 rd[8], rwl[8];
#pragma unroll
    for (int t = 0; t < 8; t++) { rd[t] = s_sc[t * 64 + lane]; rwl[t] = s_sc[(8 + t) * 64 + lane]; }
    const u32 meta = s_sc[16 * 64 + lane];
    const u32 fo = (u32)__shfl_xor((int)fl, 4, 64);                  // the other role's point flags
    bool ok = (meta & 0x200u) != 0 && (fl & HS_F_ONCURVE) && (fo & HS_F_ONCURVE);
    if (mode & HS_MODE_STRICT) ok = ok && !(fl & HS_F_SMALL) && !(fo & HS_F_SMALL);
    const int nwin = (int)(meta & 0xffu);                            // the helper wavefront's maximum: uniform
    const qv mine = quad_half_walk(rd, rwl, role1 ? 8u : 0u, nwin, !role1 && (meta & 0x100u) != 0, b16, tab);
    qv other;
#pragma unroll
    for (int i = 0; i < 10; i++) other.v[i] = (u32)__shfl_xor((int)mine.v[i], 4, 64);
    const qflags f = quad_half_join(mine, other);
    return f.f && ok;
}

// Throughput form: TWO signatures per quad, one after the other.  In verify_quad_fused lanes 2, 3 of every quad only
// shadow lane 0 through the decompression - a 251-squaring chain that is a fifth of the kernel's multiply-adds.  Here
// the four lanes decompress A, R of the quad's first signature and A, R of its second at once (one instruction stream,
// all lanes useful), then the quad runs the window loop twice: 11 % fewer multiply-adds per verification, 1.85x the
// latency of a wavefront (which now carries 32 signatures).  Used where the caller hands over enough batches per
// launch to fill the chip anyway (hs_ed25519_verify_device_batches).
struct quad_pair_points { fe X, Y; u32 fl; half_scalars hs; };
// lane 0: A of signature j0, lane 1: R of j0, lane 2: A of j1, lane 3: R of j1; lanes 0, 1 also hash signature j0 and
// lanes 2, 3 signature j1 (one SHA-512 pass for both; HS_F_SCANON in fl = s < L)
template <int W>
HS_Q quad_pair_points quad_pair_decompress(const u32 *pks, const u32 *sigs, msg_src msgs, u32 j0, u32 j1, u32 mode) {
    const u32 lane = q_lane();
    const u32 j = (lane & 2) ? j1 : j0;
    const u32 *pk = pks + (size_t)8 * j, *sig = sigs + (size_t)16 * j;
    u32 enc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) enc[t] = (lane & 1) ? sig[t] : pk[t];
    ge_p3 pt;
    quad_pair_points r;
    r.fl = ge_decompress(pt, enc) ? HS_F_ONCURVE : 0;
    if (mode & HS_MODE_STRICT) r.fl |= ge_is_small_order_affine(pt.Y) ? HS_F_SMALL : 0;
    r.X = pt.X; r.Y = pt.Y;
    u32 msg[8];
    load_msg(msg, msgs, j);
    if (prepare_scalars_half<W>(r.hs, pk, sig, msg)) r.fl |= HS_F_SCANON;
    return r;
}
// pass h (0 / 1, uniform over the wavefront): the quad's first / second signature
template <int W>
HS_Q bool quad_pair_pass(const quad_pair_points &pp, u32 h, u32 mode, const u32 *b16, qtab &atab, qtab &rtab) {
    fe sx = pp.X, sy = pp.Y;
    u32 sf = pp.fl;
    half_scalars hs;
    if (h) {        // bring lanes 2, 3 down to lanes 0, 1
        sx = q_perm<2, 3, 2, 3>(pp.X); sy = q_perm<2, 3, 2, 3>(pp.Y);
        sf = (u32)__builtin_amdgcn_mov_dpp((int)pp.fl, HS_QP(2, 3, 2, 3), 0xf, 0xf, true);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            hs.ru[t] = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.ru[t], HS_QP(2, 2, 2, 2), 0xf, 0xf, true);
            hs.rv[t] = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.rv[t], HS_QP(2, 2, 2, 2), 0xf, 0xf, true);
            hs.rw[t] = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.rw[t], HS_QP(2, 2, 2, 2), 0xf, 0xf, true);
        }
        hs.meta = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.meta, HS_QP(2, 2, 2, 2), 0xf, 0xf, true);
    } else {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            hs.ru[t] = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.ru[t], HS_QP(0, 0, 0, 0), 0xf, 0xf, true);
            hs.rv[t] = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.rv[t], HS_QP(0, 0, 0, 0), 0xf, 0xf, true);
            hs.rw[t] = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.rw[t], HS_QP(0, 0, 0, 0), 0xf, 0xf, true);
        }
        hs.meta = (u32)__builtin_amdgcn_mov_dpp((int)pp.hs.meta, HS_QP(0, 0, 0, 0), 0xf, 0xf, true);
    }
    const u32 fa = (u32)__builtin_amdgcn_mov_dpp((int)sf, 0x00, 0xf, 0xf, true);
    const u32 fr = (u32)__builtin_amdgcn_mov_dpp((int)sf, 0x55, 0xf, 0xf, true);
    bool ok = (fa & HS_F_SCANON) != 0;
    ok = ok && (fa & HS_F_ONCURVE) && (fr & HS_F_ONCURVE);
    if (mode & HS_MODE_STRICT) ok = ok && !(fa & HS_F_SMALL) && !(fr & HS_F_SMALL);
    const fe ax = q_perm<0, 0, 0, 0>(sx), ay = q_perm<0, 0, 0, 0>(sy);
    const fe rx = q_perm<1, 1, 1, 1>(sx), ry = q_perm<1, 1, 1, 1>(sy);
    const int nwin = HS_NWIN_OF(hs.meta);
    const qflags f = verify_quad_core<W>(ax, ay, rx, ry, hs, nwin, b16, atab, rtab);
    return f.f && ok;
}
#endif

}  // namespace hs
