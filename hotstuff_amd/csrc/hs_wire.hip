// hs_wire.hip - what sits either side of the verification path:
//   rows 3   k_wire_decode: bincode vote elements as received (116 bytes: u64 44 | 44 base64 characters | 64
//            signature bytes; a timeout certificate's carry a u64 round more) -> the packed (pk, R || s) the
//            verify kernels read, on the device and on the same stream, so QC::verify / TC::verify can be fed the
//            received frame (hs_ed25519_verify_wire_*).  Non-canonical elements are poisoned (s >= L) and counted.
//   row 4    k_sign / k_public_keys: RFC 8032 signing as Signature::new does it (crypto/src/lib.rs:185-191), one lane
//            per signature, R = [r]B from the base point's window table (no doublings) - for load generation.
//   the native load generator used to measure the host-pointer API end to end.
// Arithmetic in wire.h (host+device, validated on the CPU by tests/emu against OpenSSL / Python base64).
#include <deque>
#include <hip/hip_runtime.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include "wire.h"
#include "engine.h"

using namespace hs;
#define fail hs::hs_fail

// ------------------------------------------------------------------------------------------ device
// one lane per vote element; rounds != null: timeout-certificate elements (the u64 round follows the signature)
__global__ void __launch_bounds__(64)
k_wire_decode(const u32 *__restrict__ frames, u32 stride_words, u32 n, u32 *__restrict__ pks, u32 *__restrict__ sigs,
              unsigned long long *__restrict__ rounds, u32 *__restrict__ nbad) {
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const u32 *f = frames + (size_t)stride_words * i;
    u32 fr[31], pk[8], sig[16];
#pragma unroll
    for (int j = 0; j < 29; j++) fr[j] = f[j];
#pragma unroll
    for (int j = 0; j < 8; j++) pk[j] = 0;
    const bool ok = wire_decode_vote(pk, sig, fr);
    uint4 *po = reinterpret_cast<uint4 *>(pks + (size_t)8 * i), *so = reinterpret_cast<uint4 *>(sigs + (size_t)16 * i);
    po[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]); po[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
#pragma unroll
    for (int j = 0; j < 4; j++) so[j] = make_uint4(sig[4 * j], sig[4 * j + 1], sig[4 * j + 2], sig[4 * j + 3]);
    if (rounds) rounds[i] = ((unsigned long long)f[30] << 32) | f[29];
    if (!ok && nbad) atomicAdd(nbad, 1u);
}
__global__ void __launch_bounds__(64)
k_sign(const u32 *__restrict__ secrets, msg_src msgs, u32 n, const u32 *__restrict__ btab, u32 *__restrict__ sigs) {
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 sk[16], msg[8], sig[16];
#pragma unroll
    for (int j = 0; j < 16; j++) sk[j] = secrets[(size_t)16 * i + j];
    load_msg(msg, msgs, i);
    sign_one(sig, sk, msg, btab);
    uint4 *so = reinterpret_cast<uint4 *>(sigs + (size_t)16 * i);
#pragma unroll
    for (int j = 0; j < 4; j++) so[j] = make_uint4(sig[4 * j], sig[4 * j + 1], sig[4 * j + 2], sig[4 * j + 3]);
}
__global__ void __launch_bounds__(64)
k_public_keys(const u32 *__restrict__ seeds, u32 n, const u32 *__restrict__ btab, u32 *__restrict__ pks) {
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    u32 sd[8], pk[8];
#pragma unroll
    for (int j = 0; j < 8; j++) sd[j] = seeds[(size_t)8 * i + j];
    public_key_of(pk, sd, btab);
#pragma unroll
    for (int j = 0; j < 8; j++) pks[(size_t)8 * i + j] = pk[j];
}

// ------------------------------------------------------------------------------------------ host
void hs::wire_release(hs_ed25519_ctx *c) {
    if (c->d_wire) (void)hipFree(c->d_wire);
    c->d_wire = nullptr; c->wire_cap = 0;
}
static int wire_reserve(hs_ed25519_ctx *c, size_t n) {
    if (n <= c->wire_cap) return 0;
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    HS_HIP(hipStreamSynchronize(c->stream), "stream sync");
    wire_release(c);
    const size_t cap = round_up(n < 1024 ? 1024 : n, 64);
    HS_HIP(hipMalloc((void **)&c->d_wire, cap * 104 + 64), "hipMalloc wire buffers");
    c->wire_cap = cap;
    return 0;
}

// decode n vote elements (frame_bytes = 116: QC votes; 124: TC votes, d_rounds required) into packed arrays
extern "C" int hs_ed25519_wire_decode_device(hs_ed25519_ctx *ctx, const uint8_t *d_frames, size_t frame_bytes, size_t n,
                                             uint8_t *d_pks, uint8_t *d_sigs, uint64_t *d_rounds, uint32_t *d_nbad, void *stream) {
    if (!engine_ready()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx || !d_frames || !d_pks || !d_sigs) return fail(HS_ED25519_E_ARG, "wire_decode: null pointer");
    if (frame_bytes != HS_WIRE_VOTE_BYTES && frame_bytes != HS_WIRE_TC_VOTE_BYTES) return fail(HS_ED25519_E_ARG, "wire_decode: frame_bytes must be 116 or 124");
    if (frame_bytes == HS_WIRE_TC_VOTE_BYTES && !d_rounds) return fail(HS_ED25519_E_ARG, "wire_decode: timeout-certificate elements need d_rounds");
    if (misaligned(d_frames, 4) || misaligned(d_pks, 16) || misaligned(d_sigs, 16) || misaligned(d_rounds, 8) || misaligned(d_nbad, 4))
        return fail(HS_ED25519_E_ARG, "wire_decode: misaligned pointer");
    if (n == 0) return 0;
    if (n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "wire_decode: bad n");
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const u32 *p_f = (const u32 *)d_frames; u32 sw = (u32)(frame_bytes / 4), nn = (u32)n;
    u32 *p_pk = (u32 *)d_pks, *p_sg = (u32 *)d_sigs; unsigned long long *p_r = frame_bytes == HS_WIRE_TC_VOTE_BYTES ? (unsigned long long *)d_rounds : nullptr;
    u32 *p_nb = d_nbad;
    void *a[] = { &p_f, &sw, &nn, &p_pk, &p_sg, &p_r, &p_nb };
    HS_HIP(hipLaunchKernel((const void *)k_wire_decode, dim3((nn + 63) / 64), dim3(64), a, 0, stream ? (hipStream_t)stream : ctx->stream),
           "launch k_wire_decode");
    return 0;
}

// QC::verify / TC::verify fed with the vote elements as received (host buffers): decode on the device, then the
// verification kernels on the same stream.  tc: strict, per-vote digests from (round, element rounds).
static int verify_wire_host(const uint8_t *digest, bool tc, uint64_t round, const uint8_t *frames, size_t n, uint32_t mode, uint8_t *bitmap) {
    if (mode > HS_ED25519_BATCH_RLC) return fail(HS_ED25519_E_ARG, "verify_wire: bad mode");
    if (n == 0) return engine_ready() ? 1 : fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!frames || (!tc && !digest)) return fail(HS_ED25519_E_ARG, "verify_wire: null pointer");
    const size_t fb = tc ? HS_WIRE_TC_VOTE_BYTES : HS_WIRE_VOTE_BYTES;
    hs_ed25519_ctx *c = nullptr;
    int rc = pool_get(&c, 0, n);
    if (rc) return rc;
    struct putter { hs_ed25519_ctx *c; ~putter() { hs::pool_put(c); } } guard{c};
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    const size_t hint = c->narrow ? 1024 : max_batch_hint();
    rc = ctx_reserve(c, n > hint ? n : hint, true);
    if (!rc) rc = wire_reserve(c, n);
    if (rc) return rc;
    // staging: [digest 32 B][frames n * fb] (capacity 128 B per signature); always copied to HBM: the decode
    // kernel's 4-byte loads of 116-byte elements would be poor PCIe reads
    if (!tc) memcpy(c->h_in, digest, 32);
    memcpy(c->h_in + 32, frames, n * fb);
    const size_t in_bytes = 32 + n * fb;
    HS_HIP(hipMemsetAsync(c->d_out, 0, bitmap_bytes(n), c->stream), "clear bitmap");
    HS_HIP(hipMemcpyAsync(c->d_in, c->h_in, in_bytes, hipMemcpyHostToDevice, c->stream), "H2D");
    const size_t cap = c->wire_cap;
    uint8_t *d_pk = c->d_wire, *d_sg = c->d_wire + 32 * cap; uint64_t *d_rd = (uint64_t *)(c->d_wire + 96 * cap);
    rc = hs_ed25519_wire_decode_device(c, c->d_in + 32, fb, n, d_pk, d_sg, tc ? d_rd : nullptr, nullptr, c->stream);
    if (!rc) {
        if (tc) rc = hs_ed25519_verify_device_tc(c, round, d_rd, d_pk, d_sg, n, c->d_out, nullptr, c->stream);
        else rc = hs_ed25519_verify_device(c, c->d_in, 0, d_pk, d_sg, n, mode, c->d_out, nullptr, c->stream);
    }
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    HS_HIP(hipMemcpyAsync(c->h_out, c->d_out, bitmap_bytes(n), hipMemcpyDeviceToHost, c->stream), "D2H");
    HS_HIP(hipStreamSynchronize(c->stream), "stream sync");
    if (bitmap) memcpy(bitmap, c->h_out, (n + 7) / 8);
    return bitmap_all_ones(c->h_out, n) ? 1 : 0;
}
extern "C" int hs_ed25519_verify_wire_shared(const uint8_t digest[32], const uint8_t *frames, size_t n, uint32_t mode, uint8_t *bitmap) {
    return verify_wire_host(digest, false, 0, frames, n, mode, bitmap);
}
extern "C" int hs_ed25519_verify_wire_tc(uint64_t round, const uint8_t *frames, size_t n, uint8_t *bitmap) {
    return verify_wire_host(nullptr, true, round, frames, n, HS_ED25519_STRICT, bitmap);
}

// ---- signing ----
extern "C" int hs_ed25519_sign_device(hs_ed25519_ctx *ctx, const uint8_t *d_secrets, const uint8_t *d_digests, size_t digest_stride,
                                      size_t n, uint8_t *d_sigs, void *stream) {
    if (!engine_ready()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (!ctx || !d_secrets || !d_sigs) return fail(HS_ED25519_E_ARG, "sign_device: null pointer");
    if (misaligned(d_secrets, 16) || misaligned(d_sigs, 16)) return fail(HS_ED25519_E_ARG, "sign_device: misaligned pointer");
    if (n == 0) return 0;
    if (n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "sign_device: bad n");
    msg_src m;
    int rc = make_digest_src(m, d_digests, digest_stride);
    if (rc) return rc;
    const u32 *btab = dev_base_table(ctx->slot);
    if (!btab) return HS_ED25519_E_DEVICE;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const u32 *p_sk = (const u32 *)d_secrets; u32 nn = (u32)n; u32 *p_sg = (u32 *)d_sigs;
    void *a[] = { &p_sk, &m, &nn, &btab, &p_sg };
    HS_HIP(hipLaunchKernel((const void *)k_sign, dim3((nn + 63) / 64), dim3(64), a, 0, stream ? (hipStream_t)stream : ctx->stream), "launch k_sign");
    return 0;
}
// host buffers: secrets n x 64 (seed || public key, crypto::SecretKey), digests n x 32 (stride 32) or one (stride 0)
extern "C" int hs_ed25519_sign_batch(const uint8_t *secrets, const uint8_t *digests, size_t digest_stride, size_t n, uint8_t *sigs_out) {
    if (!engine_ready()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (n == 0) return 0;
    if (!secrets || !digests || !sigs_out || (digest_stride != 0 && digest_stride != 32)) return fail(HS_ED25519_E_ARG, "sign_batch: bad arguments");
    hs_ed25519_ctx *c = nullptr;
    int rc = pool_get(&c, 0, (size_t)-1);
    if (rc) return rc;
    struct putter { hs_ed25519_ctx *c; ~putter() { hs::pool_put(c); } } guard{c};
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    uint8_t *d = nullptr;
    const size_t dg_bytes = digest_stride ? 32 * n : 32, off_dg = 64 * n, off_sig = round_up(off_dg + dg_bytes, 16);
    HS_HIP(hipMalloc((void **)&d, off_sig + 64 * n), "hipMalloc sign buffers");
    hipError_t e = hipMemcpyAsync(d, secrets, 64 * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + off_dg, digests, dg_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        rc = hs_ed25519_sign_device(c, d, d + off_dg, digest_stride, n, d + off_sig, c->stream);
        if (!rc) e = hipMemcpyAsync(sigs_out, d + off_sig, 64 * n, hipMemcpyDeviceToHost, c->stream);
    }
    (void)hipMemsetAsync(d, 0, 64 * n, c->stream);          // the secrets do not outlive the call in device memory
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(HS_ED25519_E_DEVICE, "sign_batch", e != hipSuccess ? e : e2);
    return 0;
}
// device buffers: n 32-byte seeds -> n 32-byte public keys, on the context's stream (or `stream`)
int hs::public_keys_device(hs_ed25519_ctx *ctx, const uint8_t *d_seeds, size_t n, uint8_t *d_pks, void *stream) {
    if (!ctx || !d_seeds || !d_pks || n == 0 || n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "public_keys_device: bad arguments");
    const u32 *btab = dev_base_table(ctx->slot);
    if (!btab) return HS_ED25519_E_DEVICE;
    HS_HIP(hipSetDevice(ctx->device), "hipSetDevice");
    const u32 *p_sd = (const u32 *)d_seeds; u32 nn = (u32)n; u32 *p_pk = (u32 *)d_pks;
    void *a[] = { &p_sd, &nn, &btab, &p_pk };
    HS_HIP(hipLaunchKernel((const void *)k_public_keys, dim3((nn + 63) / 64), dim3(64), a, 0, stream ? (hipStream_t)stream : ctx->stream),
           "launch k_public_keys");
    return 0;
}
extern "C" int hs_ed25519_public_keys(const uint8_t *seeds, size_t n, uint8_t *pks_out) {
    if (!engine_ready()) return fail(HS_ED25519_E_NOT_INIT, "hs_ed25519_init has not succeeded");
    if (n == 0) return 0;
    if (!seeds || !pks_out || n > 0x3fffffffu) return fail(HS_ED25519_E_ARG, "public_keys: bad arguments");
    hs_ed25519_ctx *c = nullptr;
    int rc = pool_get(&c, 0, (size_t)-1);
    if (rc) return rc;
    struct putter { hs_ed25519_ctx *c; ~putter() { hs::pool_put(c); } } guard{c};
    HS_HIP(hipSetDevice(c->device), "hipSetDevice");
    uint8_t *d = nullptr;
    HS_HIP(hipMalloc((void **)&d, 64 * n), "hipMalloc key buffers");
    hipError_t e = hipMemcpyAsync(d, seeds, 32 * n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) rc = public_keys_device(c, d, n, d + 32 * n, c->stream);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(pks_out, d + 32 * n, 32 * n, hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess || e2 != hipSuccess) return fail(HS_ED25519_E_DEVICE, "public_keys", e != hipSuccess ? e : e2);
    return 0;
}
