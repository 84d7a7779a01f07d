// sha512_sc.h - k = SHA-512(R || A || M) mod L for 32-byte R, A, M, and the
// scalar-side checks of the verification path.
//
// Replaces, on crypto/src/lib.rs:200-219 of the reference, what ed25519-dalek
// 1.0.1 does with sha2::Sha512 + Scalar::from_hash (hram), check_scalar
// (s < L) and the scalar recoding that feeds the scalar multiplication.
// The reference's Digest is always 32 bytes (crypto/src/lib.rs:21-22), so the
// hash input is exactly 96 bytes = one padded SHA-512 block.
#pragma once
#include "fe26.h"
#include "sha512_consts.h"

#if defined(__HIPCC__)
#define HS_TABLE static __device__ __constant__ const
#else
#define HS_TABLE static const
#endif

namespace hs {

HS_TABLE u64 SHA512_K[80] = HS_SHA512_K_INIT;

// 64-bit rotate right by a constant.  (Two v_alignbit_b32 per rotate instead of the two 64-bit shifts
// and two ORs the compiler emits were measured and lost: 937 vs 1,097 instructions per 16 rounds, but
// no faster on a lone wavefront, where a VOP3 funnel shift issues at 8.7 cycles against 5.7.)
template <int N>
HS_HD u64 ror64(u64 x) { return (x >> N) | (x << (64 - N)); }
HS_HD u32 bswap32(u32 x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }

// one SHA-512 compression of a single padded block w[16] from the initial state
// Sixteen rounds on the state st[8] = (a..h).  SCHED: extend the message schedule in place first
// (rounds 16-79).  A template rather than `if (r)` inside the loop: with the test at run time the
// compiler kept a branch per round and 478 register moves per 16 rounds to merge the two paths
// (28 % of the loop: 1,683 instructions per 16 rounds, now 1,097); sixteen rounds are two full
// rotations of the eight state words, so straight code needs no moves at all.  Measured on one box:
// fused kernel 0.486 -> 0.455 ms per 1,024-signature batch.
template <bool SCHED>
HS_HD void sha512_rounds16(u64 st[8], u64 w[16], int r) {
    u64 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (SCHED) {
            const u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            const u64 s0 = ror64<1>(w15) ^ ror64<8>(w15) ^ (w15 >> 7);
            const u64 s1 = ror64<19>(w2) ^ ror64<61>(w2) ^ (w2 >> 6);
            w[i] += s0 + w[(i + 9) & 15] + s1;
        }
        const u64 S1 = ror64<14>(e) ^ ror64<18>(e) ^ ror64<41>(e);
        const u64 ch = (e & f) ^ (~e & g);
        const u64 t1 = h + S1 + ch + SHA512_K[r + i] + w[i];
        const u64 S0 = ror64<28>(a) ^ ror64<34>(a) ^ ror64<39>(a);
        const u64 mj = (a & b) ^ (a & c) ^ (b & c);
        const u64 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] = a; st[1] = b; st[2] = c; st[3] = d; st[4] = e; st[5] = f; st[6] = g; st[7] = h;
}
HS_HD void sha512_oneblock(u64 hh[8], u64 w[16]) {
    const u64 H0[8] = HS_SHA512_H0_INIT;
    u64 st[8];
#pragma unroll
    for (int i = 0; i < 8; i++) st[i] = H0[i];
    sha512_rounds16<false>(st, w, 0);
#pragma nounroll
    for (int r = 16; r < 80; r += 16) sha512_rounds16<true>(st, w, r);
#pragma unroll
    for (int i = 0; i < 8; i++) hh[i] = st[i] + H0[i];
}

// in: 24 little-endian words (R, A, M).  out: the 64-byte digest as a
// little-endian 512-bit integer in 16 words.
HS_HD void sha512_96(u32 out[16], const u32 in[24]) {
    u64 w[16], hh[8];
#pragma unroll
    for (int i = 0; i < 12; i++) w[i] = ((u64)bswap32(in[2 * i]) << 32) | bswap32(in[2 * i + 1]);
    w[12] = 0x8000000000000000ull; w[13] = 0; w[14] = 0; w[15] = 96 * 8;
    sha512_oneblock(hh, w);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        out[2 * i] = bswap32((u32)(hh[i] >> 32));
        out[2 * i + 1] = bswap32((u32)hh[i]);
    }
}

// TC::verify's per-vote digest, consensus/src/messages.rs:307-311:
//   SHA-512(round.to_le_bytes() || high_qc_round.to_le_bytes())[..32]   as 8 little-endian words
HS_HD u64 bswap64(u64 x) { return ((u64)bswap32((u32)x) << 32) | bswap32((u32)(x >> 32)); }
HS_HD void tc_vote_digest(u32 out[8], u64 round, u64 high_qc_round) {
    u64 w[16], hh[8];
    w[0] = bswap64(round); w[1] = bswap64(high_qc_round); w[2] = 0x8000000000000000ull;
#pragma unroll
    for (int i = 3; i < 15; i++) w[i] = 0;
    w[15] = 16 * 8;
    sha512_oneblock(hh, w);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        out[2 * i] = bswap32((u32)(hh[i] >> 32));
        out[2 * i + 1] = bswap32((u32)hh[i]);
    }
}

// Where signature j's 32-byte message comes from: packed digests, stride 0 = one shared digest.
// (A timeout certificate's per-vote digests are built on the device too, but by a kernel of their own -
// k_tc_digests - rather than inside every verify kernel: a second inlined SHA-512 made each of them
// ~5,000 instructions longer, and the five-wavefront cached kernel, whose wavefronts run two different
// code regions at once, 4-6 % slower on the common QC path.)
struct msg_src {
    const u32 *digests;
    u32 stride_words;
};
HS_HD void load_msg(u32 msg[8], const msg_src &m, u32 j) {
    const u32 *p = m.digests + (size_t)m.stride_words * j;
#pragma unroll
    for (int t = 0; t < 8; t++) msg[t] = p[t];
}

// ---------------------------------------------------------------- scalars mod L, 32-bit limbs
// L = 2^252 + c,  c = 0x14def9dea2f79cd65812631a5cf5d3ed  (125 bits)
#define HS_L_WORDS { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u }
#define HS_C_WORDS { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu }

// s < L ?  (dalek check_scalar; subsumes ed25519::Signature::from_bytes' top-3-bit test)
HS_HD bool sc_is_canonical(const u32 s[8]) {
    const u32 Lw[8] = HS_L_WORDS;
    // borrow of s - L: s < L iff the subtraction borrows
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)s[i] - Lw[i] - borrow;
        borrow = (t >> 63) & 1;
    }
    return borrow != 0;
}

// prod[0..n+3] = c * x[0..n-1]
template <int N>
HS_HD void sc_mul_c(u32 *prod, const u32 *x) {
    const u32 Cw[4] = HS_C_WORDS;
#pragma unroll
    for (int i = 0; i < N + 4; i++) prod[i] = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u64 carry = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const u64 t = (u64)x[i] * Cw[j] + prod[i + j] + carry;
            prod[i + j] = (u32)t;
            carry = t >> 32;
        }
        prod[N + j] = (u32)carry;
    }
}
// hi = x >> 252 for an NW-word x (NH = NW - 7 words out), lo = x mod 2^252 (8 words)
template <int NW>
HS_HD void sc_split252(u32 lo[8], u32 *hi, const u32 *x) {
#pragma unroll
    for (int i = 0; i < 8; i++) lo[i] = i < NW ? x[i] : 0;
    lo[7] &= 0x0fffffffu;
#pragma unroll
    for (int i = 7; i < NW; i++) hi[i - 7] = (x[i] >> 28) | (i + 1 < NW ? x[i + 1] << 4 : 0);
}
// r = x mod L for a 512-bit x.  2^252 = -c (mod L), folded three times:
//   x = lo1 + 2^252 hi1 ; c hi1 = lo2 + 2^252 hi2 ; c hi2 = lo3 + 2^252 hi3 ; p3 = c hi3
//   x = lo1 - lo2 + lo3 - p3  (mod L), every term < 2^252.
HS_HD void sc_reduce512(u32 r[8], const u32 x[16]) {
    u32 lo1[8], hi1[9], p1[13], lo2[8], hi2[6], p2[10], lo3[8], hi3[3], p3[7];
    sc_split252<16>(lo1, hi1, x);
    sc_mul_c<9>(p1, hi1);            // < 2^385
    sc_split252<13>(lo2, hi2, p1);   // hi2 < 2^133 (5 words used, 6th = 0)
    sc_mul_c<6>(p2, hi2);            // < 2^258
    sc_split252<10>(lo3, hi3, p2);   // hi3 < 2^6
    sc_mul_c<3>(p3, hi3);            // < 2^131
    const u32 Lw[8] = HS_L_WORDS;
    // t = lo1 + lo3 + 2L - lo2 - p3  in [0, 4L)
    u32 t[9];
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        int64_t v = carry + (int64_t)lo1[i] + lo3[i] + 2 * (int64_t)Lw[i] - lo2[i] - (i < 7 ? (int64_t)p3[i] : 0);
        t[i] = (u32)v;
        carry = v >> 32;
    }
    t[8] = (u32)carry;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        // u = t - L; keep if no borrow
        u32 u[9];
        u64 borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const u64 d = (u64)t[i] - (i < 8 ? Lw[i] : 0) - borrow;
            u[i] = (u32)d;
            borrow = (d >> 63) & 1;
        }
        const bool ge = borrow == 0;
#pragma unroll
        for (int i = 0; i < 9; i++) t[i] = ge ? u[i] : t[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = t[i];
}

// r = a * b mod L (a, b < 2^256)
HS_HD void sc_mul(u32 r[8], const u32 a[8], const u32 b[8]) {
    u32 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        u64 carry = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u64 t = (u64)a[i] * b[j] + p[i + j] + carry;
            p[i + j] = (u32)t;
            carry = t >> 32;
        }
        p[8 + j] = (u32)carry;
    }
    sc_reduce512(r, p);
}

// Signed radix-16 recoding for x < 2^253: r = x + 0x88..88; digit_i = nibble_i(r) - 8
// in [-8, 7] and sum digit_i 16^i = x.
HS_HD void sc_recode16(u32 r[8], const u32 x[8]) {
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)x[i] + 0x88888888u + carry;
        r[i] = (u32)t;
        carry = t >> 32;
    }
}

// Signed radix-256 recoding for x < 2^253: r = x + 0x80..80; digit_i = byte_i(r) - 128
// in [-128, 127] and sum digit_i 256^i = x.
HS_HD void sc_recode256(u32 r[8], const u32 x[8]) {
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)x[i] + 0x80808080u + carry;
        r[i] = (u32)t;
        carry = t >> 32;
    }
}

// Signed radix-65536 recoding for x < 2^253: r = x + 0x8000..8000; digit_i = halfword_i(r) - 32768
// in [-32768, 32767] and sum digit_i 65536^i = x.
HS_HD void sc_recode65536(u32 r[8], const u32 x[8]) {
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 t = (u64)x[i] + 0x80008000u + carry;
        r[i] = (u32)t;
        carry = t >> 32;
    }
}

}  // namespace hs
