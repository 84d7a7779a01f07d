// retired_cores.h - TEST INFRASTRUCTURE: the arithmetic cores of key-cache kernels that left the product
// (k_verify_cached: one quad per signature finishing by inversion; k_verify_cached_5w: the same with R decompressed
// elsewhere; k_verify_cached_lane: one lane per signature), moved here from hotstuff_amd/csrc/verify_cached.h and
// verify_quad.h when the last of those kernels went (round 5).  The emulation (emu.cpp, forms 0 / 1 / 2) keeps running them
// because they are three more organisations of the SAME field and point primitives the product's kernels use - a
// disagreement between any two forms on any vector is a bug in a primitive.  Host twins only (the emulation is compiled
// by g++); nothing under hotstuff_amd/ includes this file.
#pragma once
namespace hs {
HS_Q qv q_invert(const qv &z) { qv r; for (int i = 0; i < 4; i++) fe_invert(r.l[i], z.l[i]); return r; }
HS_Q bool q_quad_and01(const qflags &f) { return f.f[0] && f.f[1]; }
HS_Q qflags q_match_encoding(const qv &affine, const u32 enc[8]) {
    u32 wx[8], wy[8], yw[8];
    fe_to_words(wx, affine.l[0]); fe_to_words(wy, affine.l[1]);
    fe ye; fe_from_words(ye, enc); fe_to_words(yw, ye);
    u32 dy = 0, nz = 0;
    for (int i = 0; i < 8; i++) { dy |= wy[i] ^ yw[i]; nz |= wx[i]; }
    qflags r; r.f[0] = (wx[0] & 1) == (enc[7] >> 31) || nz == 0; r.f[1] = dy == 0; r.f[2] = !fe_iszero(affine.l[2]); r.f[3] = true;
    return r;
}
HS_Q qflags q_identity_lanes(const qv &p) {
    qflags r; r.f[0] = fe_iszero(p.l[0]);
    fe y = p.l[1], z = p.l[2]; fe_carry(y); fe_carry(z);
    r.f[1] = fe_eq(y, z); r.f[2] = r.f[3] = true; return r;
}
// acc == (rx, ry) as points: X == x_R Z (lane 0), Y == y_R Z (lane 1), and Z != 0 (lanes 2, 3) -
// see ge_match_point (verify.h) for why the last test is there.  The caller ANDs the four flags.
HS_Q qflags q_match_point(const qv &acc, const fe &rx, const fe &ry) {
    const qv z = q_perm<2, 2, 2, 2>(acc);
    const qv rhs = q_mul(q_lanes(rx, ry, fe_small(1), fe_small(1)), z);
    const qv lhs = q_carry(q_sel<0b1100>(acc, z));
    return q_eq_nz(rhs, lhs);
}

// ---- verification: one quad per signature -------------------------------------------------
// atab: this key's table; btab: the base point's table (same layout, +B).
// kw, sw: radix-256 recodings of k and s.  renc: the 8 words of R's encoding.
// strict: also reject when [8]R' = O.   Returns the quad-uniform verdict of the curve part.
HS_Q qv cached_quad_chain(const u32 *atab, const u32 *btab, u32 kw[8], u32 sw[8]) {
    qv acc = q_lanes(fe_small(0), fe_small(1), fe_small(1), fe_small(0));
#pragma nounroll
    for (int w = HS_KS_WINDOWS - 1; w >= 0; w--) {
        const int dk = sc_pop_digit8(kw), ds = sc_pop_digit8(sw);
        const int ik = dk < 0 ? -dk : dk, is = ds < 0 ? -ds : ds;
        const qv ea = q_load_entry(atab + ((size_t)w * HS_KS_ENTRIES + ik) * HS_KS_ENTRY_WORDS);
        const qv eb = q_load_entry(btab + ((size_t)w * HS_KS_ENTRIES + is) * HS_KS_ENTRY_WORDS);
        acc = q_add_cached(acc, q_cached_signed(ea, dk < 0));
        acc = q_add_cached(acc, q_cached_signed(eb, ds < 0));
    }
    return acc;
}
HS_Q bool verify_cached_core(const u32 *atab, const u32 *btab, u32 kw[8], u32 sw[8], const u32 renc[8], bool strict) {
    const qv acc = cached_quad_chain(atab, btab, kw, sw);
    bool ok = true;
    if (strict) {       // [8]R' == O ?  (R' == R whenever the signature is accepted)
        qv t = q_dbl(q_dbl(q_dbl(acc)));
        ok = !q_quad_and01(q_identity_lanes(t));
    }
    const qv zinv = q_invert(q_perm<2, 2, 2, 2>(acc));
    const qv aff = q_mul(acc, zinv);                        // (x', y', 1, t')
    return ok && q_quad_all(q_match_encoding(aff, renc));
}
// The other way to finish: R was decompressed elsewhere (concurrently, by another wavefront of
// the workgroup) to affine (rx, ry); compare as points, X == x_R Z and Y == y_R Z - two
// multiplications instead of an inversion on the critical path.  r_flags: HS_F_* of R.
HS_Q bool cached_quad_match_point(const qv &acc, const fe &rx, const fe &ry, u32 r_flags, bool strict) {
    const qflags f = q_match_point(acc, rx, ry);
    bool ok = (r_flags & HS_F_ONCURVE) != 0;
    if (strict) ok = ok && !(r_flags & HS_F_SMALL);
    return ok && q_quad_all(f);
}

// sum_w T_B16[w][d_w] over the 16 windows of the base point's 16-bit table and a radix-65536 recoded scalar
HS_HD void cached_chain16(ge_p3 &acc, const u32 *b16, u32 dw[8]) {
    ge_identity(acc);
    tab_entry e;
    int d = sc_pop_digit16(dw);
    load_tab_entry(e, b16 + ((size_t)(HS_B16_WINDOWS - 1) * HS_B16_ENTRIES + (d < 0 ? -d : d)) * HS_KS_ENTRY_WORDS);
#pragma nounroll
    for (int w = HS_B16_WINDOWS - 1; w >= 0; w--) {
        const bool neg = d < 0;
        add_tab_half h;
        ge_add_tab_first(h, acc, e, neg);
        if (w > 0) {
            d = sc_pop_digit16(dw);
            load_tab_entry(e, b16 + ((size_t)(w - 1) * HS_B16_ENTRIES + (d < 0 ? -d : d)) * HS_KS_ENTRY_WORDS);
        }
        ge_add_tab_second(acc, acc, h, neg);
    }
}
// The check of R' = (X:Y:Z:T) against R's encoding (and the strict small-order test).
HS_HD bool cached_finish(const ge_p3 &acc, const u32 renc[8], bool strict) {
    bool ok = true;
    if (strict) ok = !ge_is_small_order(acc);      // [8]R' == O ?  (R' == R whenever accepted)
    fe zi, x, y;
    fe_invert(zi, acc.Z);
    fe_mul(x, acc.X, zi);
    fe_mul(y, acc.Y, zi);
    u32 xw[8], yw[8], ew[8];
    fe_to_words(xw, x);
    fe_to_words(yw, y);
    fe ye; fe_from_words(ye, renc);
    fe_to_words(ew, ye);
    u32 dy = 0, nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { dy |= yw[i] ^ ew[i]; nz |= xw[i]; }
    // Z != 0: fail closed on a faulted (all-zero) accumulator, whose "affine" image is (0, 0) - ge_match_point, verify.h
    return ok && dy == 0 && ((xw[0] & 1) == (renc[7] >> 31) || nz == 0) && !fe_iszero(acc.Z);
}
// The curve part of one cached verification on ONE lane (both chains serially): the plain form the split forms below are
// held against in tests/emu (its kernel, k_verify_cached_lane, left the product in round 5 - it had no size range left).
// b16: the base point's 16-bit table; sw: s recoded in radix 65536 (kw: k in radix 256) - 32 + 16 additions.
HS_HD bool verify_cached_lane(const u32 *atab, const u32 *b16, u32 kw[8], u32 sw[8], const u32 renc[8], bool strict) {
    ge_p3 pa, pb, acc;
    cached_chain(pa, atab, kw);
    cached_chain16(pb, b16, sw);
    ge_add_p3(acc, pa, pb);
    return cached_finish(acc, renc, strict);
}
}  // namespace hs
