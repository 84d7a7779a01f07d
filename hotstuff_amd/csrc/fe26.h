// fe26.h - GF(2^255-19) for gfx950 lanes: ten unsigned limbs, radix 2^25.5.
//
// Why this radix on CDNA4: the VALU multiplies 32x32->64 with a 64-bit
// accumulate in ONE instruction (v_mad_u64_u32).  With 26/25-bit limbs every
// column of the schoolbook product - including the x19 wrap and the x2 of the
// odd*odd terms - fits a 64-bit accumulator, so a field multiplication is a
// straight run of 100 v_mad_u64_u32 with no carry handling until the end.
// (51-bit limbs, which the CPU restatement in oracle/ uses, would need
// 64x64->128 products = four such instructions each plus 128-bit adds.)
//
// Replaces curve25519-dalek's FieldElement (pulled in through
// ed25519-dalek 1.0.1, /root/reference/crypto/Cargo.toml:10) on the path
// crypto/src/lib.rs:200-219.  All code here is host+device so tests can run
// the identical arithmetic on the CPU (tests/emu) against oracle/.
//
// Limb bounds ("k x" = limb i below k * 2^(26 - (i & 1))):
//   reduced : output of mul / sq / carry            (1x, tiny slack on limb 1)
//   add     : reduced + reduced            -> 2x
//   sub     : a + 2p - b, b reduced        -> a + 2x   (no carry inside)
//   mul/sq  : operands up to 4x and 3x (second operand: 19*g must fit 32 bits)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HS_HD __host__ __device__ __forceinline__
#else
#define HS_HD inline
#endif

// HS_BOUND_CHECK (host builds of tests/emu only): every field operation asserts the
// preconditions documented above and that no 32- / 64-bit intermediate wraps.
#if defined(HS_BOUND_CHECK) && !defined(__HIPCC__)
#include <cstdio>
#include <cstdlib>
#define HS_CHECK(cond, what) do { if (!(cond)) { std::fprintf(stderr, "HS_BOUND_CHECK failed: %s (%s:%d)\n", what, __FILE__, __LINE__); std::abort(); } } while (0)
#else
#define HS_CHECK(cond, what) do { } while (0)
#endif

namespace hs {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe { u32 v[10]; };

#define HS_M26 0x3ffffffu
#define HS_M25 0x1ffffffu

HS_HD void fe_0(fe &r) {
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = 0;
}
HS_HD void fe_1(fe &r) { fe_0(r); r.v[0] = 1; }

HS_HD void fe_add(fe &r, const fe &a, const fe &b) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        HS_CHECK((u64)a.v[i] + b.v[i] < (1ull << 32), "fe_add overflows a limb");
        r.v[i] = a.v[i] + b.v[i];
    }
}
// r = a - b + 2p; b must be reduced.
HS_HD void fe_sub(fe &r, const fe &a, const fe &b) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const u32 bias = i == 0 ? 0x7ffffdau : ((i & 1) ? 0x3fffffeu : 0x7fffffeu);
        HS_CHECK(b.v[i] <= bias, "fe_sub: subtrahend limb exceeds the 2p bias (not reduced)");
        HS_CHECK((u64)a.v[i] + bias < (1ull << 32), "fe_sub overflows a limb");
        r.v[i] = a.v[i] + bias - b.v[i];
    }
}
// 32-bit carry pass: any limbs < 2^31 -> reduced.
HS_HD void fe_carry(fe &r) {
    u32 c;
#pragma unroll
    for (int i = 0; i < 10; i++) HS_CHECK(r.v[i] < 0xf0000000u, "fe_carry: limb too large for a 32-bit carry pass");
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (i & 1) { c = r.v[i] >> 25; r.v[i] &= HS_M25; }
        else       { c = r.v[i] >> 26; r.v[i] &= HS_M26; }
        r.v[i + 1] += c;
    }
    c = r.v[9] >> 25; r.v[9] &= HS_M25; r.v[0] += 19 * c;
    c = r.v[0] >> 26; r.v[0] &= HS_M26; r.v[1] += c;
}
HS_HD void fe_neg(fe &r, const fe &a) { fe z; fe_0(z); fe_sub(r, z, a); fe_carry(r); }

// the wrap of the last carry (c < 2^39) into limbs 0..2, shared by both forms
HS_HD void fe_fold_top(fe &r, u64 c) {
    u64 t = (u64)r.v[0] + 19 * c;
    r.v[0] = (u32)t & HS_M26;
    t = (t >> 26) + r.v[1];                    // t >> 26 < 2^18
    r.v[1] = (u32)t & HS_M25;
    r.v[2] += (u32)(t >> 25);
}

// carry ten 64-bit columns down to a reduced element
HS_HD void fe_reduce_cols(fe &r, u64 h[10]) {
    u64 c;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (i & 1) { c = h[i] >> 25; r.v[i] = (u32)h[i] & HS_M25; }
        else       { c = h[i] >> 26; r.v[i] = (u32)h[i] & HS_M26; }
        h[i + 1] += c;
    }
    c = h[9] >> 25; r.v[9] = (u32)h[9] & HS_M25;
    fe_fold_top(r, c);                         // c < 2^39
}

// The carry out of column k enters column k + 1 as the ADDEND of that column's first multiply-add instead of through
// a 64-bit add afterwards: v_mad_u64_u32 adds a 64-bit operand for free and a fully dependent chain of them issues as
// fast as independent ones (profiles/r1_ubench_valu.txt: 10.4 / 5.5 cycles against 11.4 / 5.9), while the
// v_lshl_add_u64 it replaces costs 6.6 / 5.3 cycles per column.  Same integers as fe_reduce_cols, bit for bit.
// Measured (profiles/r2_sweeps.md section 4): +2 % on the headline, +3.5 % on the key-cache batches, -3.5 % QC latency.
#ifndef HS_FE_SERIAL_CARRY
#define HS_FE_SERIAL_CARRY 1
#endif
// One column of a product: acc += sum a[t] * b[t], the running 64-bit value being each multiply-add's own addend.
// On the device the whole column is ONE assembly statement: left to itself the compiler re-associates the sum, starts
// every column at zero and adds the carry afterwards with a v_lshl_add_u64 of its own (ten per multiplication), and
// between separate assembly statements it must assume a forwarding hazard and pads every dependent pair with an s_nop.
// A single accumulator instead of ten also frees ~18 registers (k_verify_cached_lane2r: 242 -> 166 VGPRs).
#if defined(__HIP_DEVICE_COMPILE__)
#define HS_MAD_LINE(a, b) "v_mad_u64_u32 %0, vcc, %" #a ", %" #b ", %0\n\t"
HS_HD void fe_col5(u64 &acc, const u32 *a, const u32 *b) {
    asm(HS_MAD_LINE(1, 6) HS_MAD_LINE(2, 7) HS_MAD_LINE(3, 8) HS_MAD_LINE(4, 9) HS_MAD_LINE(5, 10)
        : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]) : "vcc");
}
HS_HD void fe_col6(u64 &acc, const u32 *a, const u32 *b) {
    asm(HS_MAD_LINE(1, 7) HS_MAD_LINE(2, 8) HS_MAD_LINE(3, 9) HS_MAD_LINE(4, 10) HS_MAD_LINE(5, 11) HS_MAD_LINE(6, 12)
        : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]),
                      "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]) : "vcc");
}
HS_HD void fe_col10(u64 &acc, const u32 *a, const u32 *b) {
    asm(HS_MAD_LINE(1, 11) HS_MAD_LINE(2, 12) HS_MAD_LINE(3, 13) HS_MAD_LINE(4, 14) HS_MAD_LINE(5, 15)
        HS_MAD_LINE(6, 16) HS_MAD_LINE(7, 17) HS_MAD_LINE(8, 18) HS_MAD_LINE(9, 19) HS_MAD_LINE(10, 20)
        : "+v"(acc) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
                      "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(b[8]), "v"(b[9]) : "vcc");
}
#endif
template <int N>
HS_HD void fe_col(u64 &acc, const u32 (&a)[N], const u32 (&b)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(N == 5 || N == 6 || N == 10, "column lengths of fe_sq (5, 6) and fe_mul (10)");
    if (N == 5) fe_col5(acc, a, b); else if (N == 6) fe_col6(acc, a, b); else fe_col10(acc, a, b);
#else
    for (int t = 0; t < N; t++) {
        HS_CHECK(acc + (u64)a[t] * b[t] >= acc, "field multiplication: 64-bit column accumulator wrapped");
        acc += (u64)a[t] * b[t];
    }
#endif
}

HS_HD void fe_mul(fe &r, const fe &f, const fe &g) {
    u32 g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        HS_CHECK((u64)19 * g.v[i] < (1ull << 32), "fe_mul: 19*g does not fit 32 bits (second operand above ~3.3x)");
        HS_CHECK((u64)2 * f.v[i] < (1ull << 32), "fe_mul: 2*f does not fit 32 bits");
        g19[i] = 19u * g.v[i]; f2[i] = 2u * f.v[i];
    }
#if HS_FE_SERIAL_CARRY
    u64 acc = 0;                               // carry into column k
    fe out;                                    // r may alias f or g
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u32 ca[10], cb[10];
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrap = i > k;
            ca[i] = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            cb[i] = wrap ? g19[j] : g.v[j];
        }
        fe_col<10>(acc, ca, cb);
        HS_CHECK(acc < (1ull << 63), "fe_mul: column too large for the carry chain");
        if (k & 1) { out.v[k] = (u32)acc & HS_M25; acc >>= 25; }
        else       { out.v[k] = (u32)acc & HS_M26; acc >>= 26; }
    }
    fe_fold_top(out, acc);
    r = out;
#else
    u64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            const bool wrap = i > k;
            const u32 fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            const u32 gj = wrap ? g19[j] : g.v[j];
            HS_CHECK(acc + (u64)fi * gj >= acc, "fe_mul: 64-bit column accumulator wrapped");
            acc += (u64)fi * gj;
        }
        HS_CHECK(acc < (1ull << 63), "fe_mul: column too large for the carry chain");
        h[k] = acc;
    }
    fe_reduce_cols(r, h);
#endif
}

// the operands of term t of column k of a squaring (i <= j, i + j = k mod 10):
//   diagonal  f_i^2 * (2 if i odd) * (19 if wrapped);  off-diagonal  2 f_i * f_j * (2 if both odd) * (19 if wrapped)
template <int K>
HS_HD void fe_sq_col(u64 &acc, const fe &f, const u32 (&f2)[10], const u32 (&f19)[10], const u32 (&f38)[10]) {
    constexpr int N = (K & 1) ? 5 : 6;
    u32 a[N], b[N];
    int t = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const int j = (K - i + 10) % 10;
        if (i > j) continue;
        const bool wrap = (i + j) >= 10;
        const bool oo = (i & 1) && (j & 1);
        if (i == j) { a[t] = oo ? f2[i] : f.v[i]; b[t] = wrap ? f19[i] : f.v[i]; }
        else { a[t] = f2[i]; b[t] = wrap ? (oo ? f38[j] : f19[j]) : (oo ? f2[j] : f.v[j]); }    // each factor stays below 2^32
        t++;
    }
    fe_col<N>(acc, a, b);
}

HS_HD void fe_sq(fe &r, const fe &f) {
    // column k = sum_{i<=j, i+j = k (mod 10)} c_ij f_i f_j, c = 2 off-diagonal,
    // x2 more when both odd, x19 when i+j >= 10.
    u32 f2[10], f19[10], f38[10];
#pragma unroll
    for (int i = 0; i < 10; i++) {
        HS_CHECK((u64)19 * f.v[i] < (1ull << 32), "fe_sq: 19*f does not fit 32 bits");
        HS_CHECK(!(i & 1) || (u64)38 * f.v[i] < (1ull << 32), "fe_sq: 38*f (odd limb) does not fit 32 bits");
        f2[i] = 2u * f.v[i]; f19[i] = 19u * f.v[i];
#if defined(__HIP_DEVICE_COMPILE__)
        // 38 f as a shift of 19 f: left to itself the compiler multiplies again (v_mul_lo_u32 issues at 4.8 cycles per
        // wavefront instruction, a plain shift at 2.5 - profiles/r1_ubench_valu.txt); only limbs 5, 7, 9 are ever used
        if (i & 1) asm("v_lshlrev_b32 %0, 1, %1" : "=v"(f38[i]) : "v"(f19[i])); else f38[i] = 0;
#else
        f38[i] = 38u * f.v[i];
#endif
    }
#if HS_FE_SERIAL_CARRY
    u64 acc = 0;                               // carry into the next column
    fe out;                                    // r may alias f
#define HS_SQ_COL(K, M, S) fe_sq_col<K>(acc, f, f2, f19, f38); HS_CHECK(acc < (1ull << 63), "fe_sq: column too large for the carry chain"); \
    out.v[K] = (u32)acc & M; acc >>= S;
    HS_SQ_COL(0, HS_M26, 26) HS_SQ_COL(1, HS_M25, 25) HS_SQ_COL(2, HS_M26, 26) HS_SQ_COL(3, HS_M25, 25) HS_SQ_COL(4, HS_M26, 26)
    HS_SQ_COL(5, HS_M25, 25) HS_SQ_COL(6, HS_M26, 26) HS_SQ_COL(7, HS_M25, 25) HS_SQ_COL(8, HS_M26, 26) HS_SQ_COL(9, HS_M25, 25)
#undef HS_SQ_COL
    fe_fold_top(out, acc);
    r = out;
#else
    u64 h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        u64 acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool oo = (i & 1) && (j & 1);
            const u32 a = i == j ? (oo ? f2[i] : f.v[i]) : f2[i];
            const u32 b = i == j ? (wrap ? f19[i] : f.v[i]) : (wrap ? (oo ? f38[j] : f19[j]) : (oo ? f2[j] : f.v[j]));
            HS_CHECK(acc + (u64)a * b >= acc, "fe_sq: 64-bit column accumulator wrapped");
            acc += (u64)a * b;
        }
        HS_CHECK(acc < (1ull << 63), "fe_sq: column too large for the carry chain");
        h[k] = acc;
    }
    fe_reduce_cols(r, h);
#endif
}

// r = a^(2^N).  Short runs are straight-line; long runs are a real loop (a 1- or 2-trip
// `nounroll` loop sends hipcc's register allocation from ~130 to ~300 VGPRs - measured).
template <int N>
HS_HD void fe_sqn(fe &r, const fe &a) {
    fe_sq(r, a);
    if (N <= 5) {
#pragma unroll
        for (int i = 1; i < N; i++) fe_sq(r, r);
    } else {
#pragma nounroll
        for (int i = 1; i < N; i++) fe_sq(r, r);
    }
}

// z^(2^250 - 1) and z^11: shared head of the inversion and sqrt-ratio ladders
HS_HD void fe_pow250(fe &r, fe &z11, const fe &z) {
    fe z2, z9, t, z5, z10, z20, z50, z100;
    fe_sq(z2, z); fe_sqn<2>(t, z2); fe_mul(z9, t, z); fe_mul(z11, z9, z2);
    fe_sq(t, z11); fe_mul(z5, t, z9);
    fe_sqn<5>(t, z5); fe_mul(z10, t, z5);
    fe_sqn<10>(t, z10); fe_mul(z20, t, z10);
    fe_sqn<20>(t, z20); fe_mul(t, t, z20);
    fe_sqn<10>(t, t); fe_mul(z50, t, z10);
    fe_sqn<50>(t, z50); fe_mul(z100, t, z50);
    fe_sqn<100>(t, z100); fe_mul(t, t, z100);
    fe_sqn<50>(t, t); fe_mul(r, t, z50);
}
HS_HD void fe_invert(fe &r, const fe &z) {
    fe t, z11; fe_pow250(t, z11, z); fe_sqn<5>(t, t); fe_mul(r, t, z11);
}
HS_HD void fe_pow22523(fe &r, const fe &z) {
    fe t, z11; fe_pow250(t, z11, z); fe_sqn<2>(t, t); fe_mul(r, t, z);
}

// little-endian 32 bytes (as 8 words) -> reduced limbs; bit 255 dropped, no canonical check
HS_HD void fe_from_words(fe &r, const u32 w[8]) {
    r.v[0] = w[0] & HS_M26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & HS_M25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & HS_M26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & HS_M25;
    r.v[4] = (w[3] >> 6) & HS_M26;
    r.v[5] = w[4] & HS_M25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & HS_M26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & HS_M25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & HS_M26;
    r.v[9] = (w[7] >> 6) & HS_M25;
}
// canonical value as 8 little-endian words (input: any limbs < 2^31)
HS_HD void fe_to_words(u32 w[8], const fe &a) {
    fe t = a;
    fe_carry(t); fe_carry(t);
    // q = 1 iff t >= p
    u32 q = (t.v[0] + 19) >> 26;
#pragma unroll
    for (int i = 1; i < 10; i++) q = (t.v[i] + q) >> ((i & 1) ? 25 : 26);
    t.v[0] += 19 * q;
    u32 c;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (i & 1) { c = t.v[i] >> 25; t.v[i] &= HS_M25; }
        else       { c = t.v[i] >> 26; t.v[i] &= HS_M26; }
        t.v[i + 1] += c;
    }
    t.v[9] &= HS_M25;
    w[0] = t.v[0] | (t.v[1] << 26);
    w[1] = (t.v[1] >> 6) | (t.v[2] << 19);
    w[2] = (t.v[2] >> 13) | (t.v[3] << 13);
    w[3] = (t.v[3] >> 19) | (t.v[4] << 6);
    w[4] = t.v[5] | (t.v[6] << 25);
    w[5] = (t.v[6] >> 7) | (t.v[7] << 19);
    w[6] = (t.v[7] >> 13) | (t.v[8] << 12);
    w[7] = (t.v[8] >> 20) | (t.v[9] << 6);
}
HS_HD bool fe_iszero(const fe &a) {
    u32 w[8]; fe_to_words(w, a);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d |= w[i];
    return d == 0;
}
HS_HD bool fe_isneg(const fe &a) { u32 w[8]; fe_to_words(w, a); return w[0] & 1; }
// a == b (mod p); b reduced
HS_HD bool fe_eq(const fe &a, const fe &b) { fe t; fe_sub(t, a, b); return fe_iszero(t); }

}  // namespace hs
