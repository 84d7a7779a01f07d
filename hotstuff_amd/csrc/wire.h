// wire.h - the data formats either side of the verification path, host+device.
//
// (1) Wire decode.  A QC / TC travels as a bincode frame (consensus/src/consensus.rs:136-161) in which
//     votes: Vec<(PublicKey, Signature)> is n elements of 116 bytes:
//         u64 LE  44                       length of the key's base64 STRING (crypto/src/lib.rs:94-101:
//         44 ASCII characters              PublicKey serialises as serialize_str(base64::encode(key)))
//         32 + 32 raw bytes                Signature { part1, part2 } (derive(Serialize) on two [u8; 32])
//     and a TC's votes: Vec<(PublicKey, Signature, Round)> is the same plus a u64 LE round: 124 bytes.
//     wire_decode_vote turns one element into the packed (pk, R || s) the kernels read, so the shim can hand
//     the received frame to the GPU as it is instead of base64-decoding n keys on the Core task.
//     Contract: the CANONICAL encoding, i.e. what the reference's own serialiser always emits - length field
//     44, standard alphabet, one '=' of padding, zero trailing bits.  Anything else marks the element bad (the
//     kernels then reject it whatever the signature says); the shim re-parses such a frame with the reference's
//     own deserialiser, which decides what it means (base64 0.13 tolerates some non-canonical forms).
//
// (2) Signing.  Signature::new (crypto/src/lib.rs:185-191) = dalek Keypair::from_bytes(secret).sign(digest):
//     RFC 8032 5.1.6 with the key pair's stored public key.  sign_one restates it for one (secret, digest).
//     Not on any node's hot path (one signature per block / vote / timeout); it exists so that 10^5 - 10^6
//     signature synthetic sets for the load generator take milliseconds.
#pragma once
#include "verify_cached.h"

namespace hs {

#define HS_WIRE_VOTE_BYTES 116
#define HS_WIRE_TC_VOTE_BYTES 124

// one base64 character -> 6 bits, or 0xff
HS_HD u32 b64_value(u32 c) {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return 0xffu;
}
// frame: the element's bytes as little-endian words (29 for a QC vote; a TC vote's round follows at word 29).
// pk[8], sig[16]: the packed key and signature words.  Returns false for a non-canonical element.
HS_HD bool wire_decode_vote(u32 pk[8], u32 sig[16], const u32 *frame) {
    bool ok = frame[0] == 44u && frame[1] == 0u;
    // 44 characters at bytes 8..51 = words 2..12
    u32 acc = 0, nbits = 0, outw = 0, outn = 0, bad = 0;
    int k = 0;
#pragma unroll
    for (int i = 0; i < 44; i++) {
        const u32 c = (frame[2 + (i >> 2)] >> (8 * (i & 3))) & 0xffu;
        if (i == 43) { ok = ok && c == '='; break; }
        const u32 v = b64_value(c);
        bad |= v & 0x80u;
        acc = (acc << 6) | (v & 63u); nbits += 6;
        if (nbits >= 8) {
            nbits -= 8;
            const u32 byte = (acc >> nbits) & 0xffu;
            if (k < 32) { outw |= byte << (8 * outn); if (++outn == 4) { pk[k >> 2] = outw; outw = 0; outn = 0; } }
            k++;
        }
    }
    // 43 characters = 258 bits = 32 bytes + 2 trailing bits, which the canonical form leaves zero
    ok = ok && !bad && (acc & 3u) == 0;
#pragma unroll
    for (int j = 0; j < 16; j++) sig[j] = frame[13 + j];
    if (!ok) {       // poison: s = 2^256 - 1 >= L is rejected by every kernel before any curve arithmetic
#pragma unroll
        for (int j = 8; j < 16; j++) sig[j] = 0xffffffffu;
    }
    return ok;
}

// ---- signing ------------------------------------------------------------------------------------
// SHA-512 of a 32- or 64-byte message given as little-endian words; digest as 16 little-endian words
template <int NWORDS>
HS_HD void sha512_words(u32 out[16], const u32 *in) {
    u64 w[16], hh[8];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < NWORDS / 2; i++) w[i] = ((u64)bswap32(in[2 * i]) << 32) | bswap32(in[2 * i + 1]);
    w[NWORDS / 2] = 0x8000000000000000ull;
    w[15] = (u64)NWORDS * 32;
    sha512_oneblock(hh, w);
#pragma unroll
    for (int i = 0; i < 8; i++) { out[2 * i] = bswap32((u32)(hh[i] >> 32)); out[2 * i + 1] = bswap32((u32)hh[i]); }
}
// r = a + b mod L for a, b < L
HS_HD void sc_add(u32 r[8], const u32 a[8], const u32 b[8]) {
    const u32 Lw[8] = HS_L_WORDS;
    u32 t[9], u[9];
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { const u64 v = (u64)a[i] + b[i] + carry; t[i] = (u32)v; carry = v >> 32; }
    t[8] = (u32)carry;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { const u64 d = (u64)t[i] - (i < 8 ? Lw[i] : 0) - borrow; u[i] = (u32)d; borrow = (d >> 63) & 1; }
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = borrow ? t[i] : u[i];
}
// compressed encoding of a point
HS_HD void ge_compress(u32 out[8], const ge_p3 &p) {
    fe zi, x, y;
    fe_invert(zi, p.Z);
    fe_mul(x, p.X, zi);
    fe_mul(y, p.Y, zi);
    u32 xw[8];
    fe_to_words(xw, x);
    fe_to_words(out, y);
    out[7] |= (xw[0] & 1u) << 31;
}
// secret: the 64 key-pair bytes (seed || public key) as words; msg: the 32-byte digest; btab: the base point's
// window table (verify_cached.h layout).  sig = R || S.
HS_HD void sign_one(u32 sig[16], const u32 secret[16], const u32 msg[8], const u32 *btab) {
    u32 h[16], a[8], in[24], rh[16], r[8], rw[8], kh[16], k[8], ka[8], s[8];
    sha512_words<8>(h, secret);                            // h = SHA-512(seed)
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = h[i];
    a[0] &= 0xfffffff8u; a[7] &= 0x7fffffffu; a[7] |= 0x40000000u;      // clamp (RFC 8032 5.1.5)
#pragma unroll
    for (int i = 0; i < 8; i++) { in[i] = h[8 + i]; in[8 + i] = msg[i]; }
    sha512_words<16>(rh, in);                              // r = SHA-512(prefix || M) mod L
    sc_reduce512(r, rh);
    sc_recode256(rw, r);
    ge_p3 R;
    cached_chain(R, btab, rw);                             // R = [r]B: 32 table additions, no doublings
    ge_compress(sig, R);
#pragma unroll
    for (int i = 0; i < 8; i++) { in[i] = sig[i]; in[8 + i] = secret[8 + i]; in[16 + i] = msg[i]; }
    sha512_96(kh, in);                                     // k = SHA-512(R || A || M) mod L
    sc_reduce512(k, kh);
    sc_mul(ka, k, a);                                      // S = r + k a mod L
    sc_add(s, ka, r);
#pragma unroll
    for (int i = 0; i < 8; i++) sig[8 + i] = s[i];
}
// public key of a 32-byte seed
HS_HD void public_key_of(u32 pk[8], const u32 seed[8], const u32 *btab) {
    u32 h[16], a[8], aw[8], ar[8], z[16];
    sha512_words<8>(h, seed);
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = h[i];
    a[0] &= 0xfffffff8u; a[7] &= 0x7fffffffu; a[7] |= 0x40000000u;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = i < 8 ? a[i] : 0;
    sc_reduce512(ar, z);                                   // B has order L: [a]B = [a mod L]B, and the recoding needs < 2^253
    sc_recode256(aw, ar);
    ge_p3 A;
    cached_chain(A, btab, aw);
    ge_compress(pk, A);
}

}  // namespace hs
