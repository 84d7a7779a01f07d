// hotstuff_crypto.hpp - C++ host-side mirror of the reference's interface for the verification
// path, over the C ABI in hs_ed25519.h (header-only, C++17).
//
// The reference is Rust and this image has no Rust toolchain, so the host side above the C ABI is
// restated in C++ (the reference is compiled code) with the reference's names, argument meaning
// and error behaviour; INTEGRATION.md holds the Rust shim itself.  What is mirrored:
//
//   crypto/src/lib.rs:18        CryptoError (opaque: callers only learn "Err")
//   crypto/src/lib.rs:21-22     Digest([u8; 32])
//   crypto/src/lib.rs:65-66     PublicKey([u8; 32])
//   crypto/src/lib.rs:178-219   Signature { part1, part2 }, flatten, verify, verify_batch
//   consensus/src/config.rs:39-72          Committee: stake, quorum_threshold (2 * total / 3 + 1)
//   consensus/src/error.rs                 the ConsensusError variants this path can return
//   consensus/src/messages.rs:164-208      QC { hash, round, votes }, digest, verify
//   consensus/src/messages.rs:283-315      TC { round, votes }, verify
//   consensus/src/messages.rs:15-90        Block { qc, tc, author, round, payload, signature }, digest, verify
//   consensus/src/messages.rs:112-156      Vote { hash, round, author, signature }, digest, verify
//   consensus/src/messages.rs:211-275      Timeout { high_qc, round, author, signature }, digest, verify
// and the two callers either side of the path that SURVEY.md 8f ranks first (protocol-neutral changes
// on the consensus side; INTEGRATION.md section 6 has the Rust patch):
//   VoteIngest        consensus/src/core.rs:203-225 + aggregator.rs:74-94: accumulate votes, verify them in one call
//   VerifiedQCCache   consensus/src/messages.rs:261-263 / core.rs:227-238: a timeout storm verifies each distinct
//                     high_qc once
// Small calls: below hs_ed25519_min_gpu_batch() a call is cheaper on one CPU core than the
// GPU round trip.  The Rust shim keeps those on dalek; this mirror has no CPU implementation of its own (the product
// never links the test oracle), so the policy is a hook: SmallBatch::cpu_verify, when set, receives such calls.
//
// Rust's Result<(), E> becomes a small value type with is_ok() / is_err() so that the parity
// tests read like the reference's own (crypto_tests.rs, messages_tests.rs).  There is no CPU
// fallback: an engine error (no GPU, HIP failure) is an Err like any other, never an Ok.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include "hs_ed25519.h"
#include "../hotstuff_amd/csrc/sha512_sc.h"      // host+device SHA-512 (one padded block)

namespace hotstuff {

using Stake = uint32_t;   // consensus/src/config.rs:8
using Round = uint64_t;   // consensus/src/messages.rs (pub type Round = u64, core.rs)

struct Digest {
    std::array<uint8_t, 32> b{};
    bool operator==(const Digest &o) const { return b == o.b; }
    bool operator!=(const Digest &o) const { return !(b == o.b); }
};
struct PublicKey {
    std::array<uint8_t, 32> b{};
    bool operator==(const PublicKey &o) const { return b == o.b; }
    bool operator<(const PublicKey &o) const { return b < o.b; }        // BTreeMap / HashSet key
};

// Result<(), CryptoError>: the reference's error is opaque (signature::Error), so is this one.
struct CryptoResult {
    bool ok;
    bool is_ok() const { return ok; }
    bool is_err() const { return !ok; }
};

namespace detail {
// 1 -> Ok, 0 (rejected) and < 0 (engine error) -> Err; the engine never reports success on failure.
inline CryptoResult to_result(int rc) { return CryptoResult{rc == 1}; }
// Devices: HS_ED25519_DEVICES="0,1,2,..." opens several GPUs behind this one process (the reference node is one
// process); unset = the GPU named by HS_ED25519_DEVICE (default 0).  Idempotent and thread-safe in the library.
inline bool ensure_init() {
    static const int rc = [] {
        // the library must have been built from the header this mirror was compiled against (same ABI major): what the Rust
        // shim's engine_start does.  A mismatch fails every call, loudly, instead of calling through
        // argument lists that may have changed.
        if ((hs_ed25519_abi_version() >> 16) != (HS_ED25519_ABI_VERSION >> 16)) return HS_ED25519_E_ARG;
        std::vector<int> devs;
        if (const char *e = std::getenv("HS_ED25519_DEVICES")) {
            int v = 0; bool have = false;
            for (const char *p = e;; p++) {
                if (*p >= '0' && *p <= '9') { v = 10 * v + (*p - '0'); have = true; }
                else { if (have) devs.push_back(v); v = 0; have = false; if (!*p) break; }
            }
        }
        if (devs.empty()) { const char *e = std::getenv("HS_ED25519_DEVICE"); devs.push_back(e ? std::atoi(e) : 0); }
        return hs_ed25519_init_devices(devs.data(), devs.size(), 8192);
    }();
    return rc == 0;
}
inline bool several_devices() { return hs_ed25519_device_count() > 1; }
}  // namespace detail

// SURVEY.md 8a row A7.  threshold(): the measured crossover the library reports (bench.py
// `small_batch_crossover`); cpu_verify: the CPU path of the embedding program (the Rust shim: dalek), same
// contract as hs_ed25519_verify_multi (digest_stride 32) / _shared (0); null = everything goes to the GPU.
struct SmallBatch {
    using CpuVerify = int (*)(const uint8_t *digests, size_t digest_stride, const uint8_t *pks, const uint8_t *sigs,
                              size_t n, uint32_t mode, uint8_t *bitmap);
    static CpuVerify &cpu_verify() { static CpuVerify f = nullptr; return f; }
    static size_t threshold(bool keys_registered) { return hs_ed25519_min_gpu_batch(keys_registered ? 1 : 0); }
    static bool on_cpu(size_t n, bool keys_registered) { return cpu_verify() && n > 0 && n < threshold(keys_registered); }
};

namespace detail {
// An engine fault (rc < 0: no device, a HIP error, a library that was shut down) is not a verdict.  With a CPU verifier supplied
// by the embedder (SmallBatch::cpu_verify - in the Rust shim: the reference's own dalek bodies) the
// call is answered THERE, in every entry point alike (round 6; until then only verify_batch's shim text fell back): a device
// fault then neither accepts anything nor rejects a valid certificate - a TC refused during a view change costs liveness
// (consensus/src/core.rs:434-439 drops what fails to verify).  Without a CPU verifier a fault stays an error: never Ok.
// engine_faults() counts the calls answered on the CPU for that reason, so a host can see that it runs without its GPU.
inline std::atomic<unsigned long long> &engine_faults() { static std::atomic<unsigned long long> n{0}; return n; }
inline int settle(int rc, const uint8_t *digests, size_t digest_stride, const uint8_t *pks, const uint8_t *sigs, size_t n,
                  uint32_t mode, uint8_t *bitmap) {
    if (rc >= 0 || !SmallBatch::cpu_verify()) return rc;
    engine_faults()++;
    return SmallBatch::cpu_verify()(digests, digest_stride, pks, sigs, n, mode == HS_ED25519_BATCH_RLC ? (uint32_t)HS_ED25519_BATCH : mode, bitmap);
}
}  // namespace detail

struct Signature {
    std::array<uint8_t, 32> part1{}, part2{};             // R, s.  Default = 64 zero bytes (lib.rs Default derive)

    static Signature from_bytes(const uint8_t sig[64]) {
        Signature s;
        std::memcpy(s.part1.data(), sig, 32);
        std::memcpy(s.part2.data(), sig + 32, 32);
        return s;
    }
    std::array<uint8_t, 64> flatten() const {              // lib.rs:193-198
        std::array<uint8_t, 64> f;
        std::memcpy(f.data(), part1.data(), 32);
        std::memcpy(f.data() + 32, part2.data(), 32);
        return f;
    }
    // lib.rs:200-204: ed25519-dalek verify_strict
    CryptoResult verify(const Digest &digest, const PublicKey &public_key) const {
        const auto sig = flatten();
        if (SmallBatch::on_cpu(1, false))                  // one signature: the CPU path when the embedder provides one
            return detail::to_result(SmallBatch::cpu_verify()(digest.b.data(), 32, public_key.b.data(), sig.data(), 1,
                                                              HS_ED25519_STRICT, nullptr));
        const int rc = detail::ensure_init() ? hs_ed25519_verify_multi(digest.b.data(), public_key.b.data(), sig.data(), 1, HS_ED25519_STRICT, nullptr)
                                             : HS_ED25519_E_NOT_INIT;
        return detail::to_result(detail::settle(rc, digest.b.data(), 32, public_key.b.data(), sig.data(), 1, HS_ED25519_STRICT, nullptr));
    }
    // lib.rs:206-219: one digest shared by every (key, signature) pair; an empty batch is Ok.
    // Vec<(PublicKey, Signature)> has no guaranteed layout, so the votes are copied into packed buffers.
    static CryptoResult verify_batch(const Digest &digest, const std::vector<std::pair<PublicKey, Signature>> &votes) {
        std::vector<uint8_t> pks(32 * votes.size()), sigs(64 * votes.size());
        for (size_t i = 0; i < votes.size(); i++) {
            std::memcpy(&pks[32 * i], votes[i].first.b.data(), 32);
            std::memcpy(&sigs[64 * i], votes[i].second.part1.data(), 32);
            std::memcpy(&sigs[64 * i + 32], votes[i].second.part2.data(), 32);
        }
        if (SmallBatch::on_cpu(votes.size(), false))       // e.g. the 3 votes of a 4-node committee
            return detail::to_result(SmallBatch::cpu_verify()(digest.b.data(), 0, pks.data(), sigs.data(), votes.size(),
                                                              HS_ED25519_BATCH, nullptr));
        int rc = HS_ED25519_E_NOT_INIT;
        if (detail::ensure_init()) {
            // dalek's verify_batch as it is computed (one random linear combination), per-signature localisation
            // inside the same call; below hs_ed25519_rlc_min_batch() the engine runs the per-signature kernels alone
            // with several device slots open the call is cut into one shard per GPU (hs_shard.hip)
            const auto fn = detail::several_devices() ? hs_ed25519_verify_shared_sharded : hs_ed25519_verify_shared;
            rc = fn(digest.b.data(), pks.data(), sigs.data(), votes.size(), HS_ED25519_BATCH_RLC, nullptr);
        }
        return detail::to_result(detail::settle(rc, digest.b.data(), 0, pks.data(), sigs.data(), votes.size(), HS_ED25519_BATCH, nullptr));
    }
    // Several certificates at once - a block's payload certificates, the QCs of a catch-up range: every entry is
    // one verify_batch call (same law, same Ok-on-empty), all in flight together through the ticket API, small
    // certificates sharing launches (HS_ED25519_SUBMIT_MORE).  Result i = verify_batch(certs[i]).
    static std::vector<CryptoResult> verify_batches(const std::vector<std::pair<Digest, std::vector<std::pair<PublicKey, Signature>>>> &certs) {
        std::vector<CryptoResult> out(certs.size(), CryptoResult{false});
        if (certs.empty()) return out;
        const bool engine = detail::ensure_init();
        std::vector<std::pair<size_t, hs_ed25519_ticket *>> inflight;
        std::vector<uint8_t> pks, sigs;
        auto pack = [&](size_t c) {
            const auto &votes = certs[c].second;
            pks.resize(32 * votes.size()); sigs.resize(64 * votes.size());
            for (size_t i = 0; i < votes.size(); i++) {
                std::memcpy(&pks[32 * i], votes[i].first.b.data(), 32);
                const auto f = votes[i].second.flatten();
                std::memcpy(&sigs[64 * i], f.data(), 64);
            }
        };
        // an engine fault on certificate c: the CPU verifier answers for it, when there is one (detail::settle)
        auto on_cpu_after_fault = [&](size_t c, int rc) {
            std::vector<uint8_t> keep_p, keep_s;
            keep_p.swap(pks); keep_s.swap(sigs);            // (the caller may be mid-way through packing another certificate)
            pack(c);
            const CryptoResult r = detail::to_result(detail::settle(rc, certs[c].first.b.data(), 0, pks.data(), sigs.data(), certs[c].second.size(), HS_ED25519_BATCH, nullptr));
            pks.swap(keep_p); sigs.swap(keep_s);
            return r;
        };
        auto collect_oldest = [&]() {
            const size_t c = inflight.front().first;
            const int rc = hs_ed25519_ticket_wait(inflight.front().second, nullptr);
            inflight.erase(inflight.begin());
            out[c] = rc < 0 ? on_cpu_after_fault(c, rc) : detail::to_result(rc);
        };
        for (size_t c = 0; c < certs.size(); c++) {
            const auto &votes = certs[c].second;
            pack(c);
            if (!engine) { out[c] = votes.empty() ? CryptoResult{true} : on_cpu_after_fault(c, HS_ED25519_E_NOT_INIT); continue; }
            if (SmallBatch::on_cpu(votes.size(), false)) {
                out[c] = detail::to_result(SmallBatch::cpu_verify()(certs[c].first.b.data(), 0, pks.data(), sigs.data(), votes.size(),
                                                                  HS_ED25519_BATCH, nullptr));
                continue;
            }
            const uint32_t flags = c + 1 < certs.size() ? HS_ED25519_SUBMIT_MORE : 0;      // submit copies: the buffers are reused at once
            for (;;) {
                hs_ed25519_ticket *t = nullptr;
                const int rc = hs_ed25519_submit(&t, nullptr, certs[c].first.b.data(), 0, pks.data(), sigs.data(), votes.size(),
                                                 HS_ED25519_BATCH_RLC, flags);
                if (rc == 0) { inflight.push_back({c, t}); break; }
                if (rc != HS_ED25519_E_BUSY || inflight.empty()) { out[c] = on_cpu_after_fault(c, rc); break; }   // engine error: the CPU verifier, or Err
                collect_oldest();
            }
        }
        while (!inflight.empty()) collect_oldest();
        return out;
    }
    // One digest per pair, strict semantics: the batched form of a loop of verify() calls (what VoteIngest and
    // TC::verify need).  Returns the per-pair verdicts; empty on an engine error.
    static std::vector<bool> verify_many(const std::vector<Digest> &digests,
                                         const std::vector<std::pair<PublicKey, Signature>> &votes) {
        const size_t n = votes.size();
        std::vector<bool> out;
        if (n == 0 || digests.size() != n) return out;
        std::vector<uint8_t> dg(32 * n), pks(32 * n), sigs(64 * n), bm((n + 7) / 8, 0);
        for (size_t i = 0; i < n; i++) {
            std::memcpy(&dg[32 * i], digests[i].b.data(), 32);
            std::memcpy(&pks[32 * i], votes[i].first.b.data(), 32);
            const auto f = votes[i].second.flatten();
            std::memcpy(&sigs[64 * i], f.data(), 64);
        }
        int rc;
        if (SmallBatch::on_cpu(n, false)) rc = SmallBatch::cpu_verify()(dg.data(), 32, pks.data(), sigs.data(), n, HS_ED25519_STRICT, bm.data());
        else {
            rc = detail::ensure_init() ? hs_ed25519_verify_multi(dg.data(), pks.data(), sigs.data(), n, HS_ED25519_STRICT, bm.data()) : HS_ED25519_E_NOT_INIT;
            if (rc < 0) { std::fill(bm.begin(), bm.end(), 0); rc = detail::settle(rc, dg.data(), 32, pks.data(), sigs.data(), n, HS_ED25519_STRICT, bm.data()); }
        }
        if (rc < 0) return out;
        out.resize(n);
        for (size_t i = 0; i < n; i++) out[i] = (bm[i >> 3] >> (i & 7)) & 1;
        return out;
    }
};

// ---- consensus side ------------------------------------------------------------------------
struct ConsensusError {
    enum Kind { AuthorityReuse, UnknownAuthority, QCRequiresQuorum, TCRequiresQuorum, InvalidSignature } kind;
    PublicKey name{};                                      // AuthorityReuse(name) / UnknownAuthority(name)
};
struct ConsensusResult {                                   // Result<(), ConsensusError>
    bool ok;
    ConsensusError err{};
    bool is_ok() const { return ok; }
    bool is_err() const { return !ok; }
    static ConsensusResult Ok() { return ConsensusResult{true, {}}; }
    static ConsensusResult Err(ConsensusError::Kind k, const PublicKey &who = PublicKey{}) {
        return ConsensusResult{false, ConsensusError{k, who}};
    }
};

struct Committee {                                         // config.rs:39-43 reduced to name -> stake
    std::map<PublicKey, Stake> authorities;
    Stake stake(const PublicKey &name) const {             // config.rs:63-65: 0 for a stranger
        const auto it = authorities.find(name);
        return it == authorities.end() ? 0 : it->second;
    }
    Stake quorum_threshold() const {                       // config.rs:67-72
        uint64_t total = 0;
        for (const auto &a : authorities) total += a.second;
        return (Stake)(2 * total / 3 + 1);
    }

    // The committee is static for an epoch: register its keys once (per-key tables in HBM,
    // hs_ed25519_keyset_create) and every later QC / TC verification skips the doublings and the
    // decompression of A.  Call after `authorities` is final; returns false on an engine error.
    bool cache_keys() {
        drop_key_cache();
        if (authorities.empty() || !detail::ensure_init()) return false;
        std::vector<uint8_t> pks;
        uint32_t i = 0;
        fast_.reserve(2 * authorities.size());
        for (const auto &a : authorities) {
            pks.insert(pks.end(), a.first.b.begin(), a.first.b.end());
            fast_[a.first] = Member{a.second, i};
            index_[a.first] = i++;
        }
        total_stake_ = 0;
        for (const auto &a : authorities) total_stake_ += a.second;
        if (hs_ed25519_keyset_create(&keyset_, pks.data(), authorities.size()) != 0) { keyset_ = nullptr; index_.clear(); fast_.clear(); return false; }
        return true;
    }
    void drop_key_cache() { if (keyset_) hs_ed25519_keyset_destroy(keyset_); keyset_ = nullptr; index_.clear(); fast_.clear(); }
    // With the keys registered the committee is final for the epoch, and the structural loop of a certificate (messages.rs:182-194,
    // 292-304) can run on ONE hash lookup per vote - stake, position in the key set and, through a seen-bitmap over positions, the
    // reuse check - instead of a std::set insertion and two std::map searches over 32-byte keys: at N = 10,000 the loop was 2.8 of
    // the 2.9 ms a QC::verify took through this twin (the engine: 0.1 ms).  The reference's own loop uses a HashSet.
    struct Member { Stake stake; uint32_t index; };
    const Member *member(const PublicKey &name) const { const auto it = fast_.find(name); return it == fast_.end() ? nullptr : &it->second; }
    Stake registered_quorum_threshold() const { return (Stake)(2 * total_stake_ / 3 + 1); }
    const hs_ed25519_keyset *keyset() const { return keyset_; }
    // position of `name` in the key set, or an index outside it (the engine rejects that signature)
    uint32_t key_index(const PublicKey &name) const {
        const auto it = index_.find(name);
        return it == index_.end() ? 0xffffffffu : it->second;
    }
    Committee() = default;
    Committee(const Committee &) = delete;                 // owns a device resource
    Committee &operator=(const Committee &) = delete;
    ~Committee() { drop_key_cache(); }

 private:
    struct KeyHash { size_t operator()(const PublicKey &k) const { uint64_t h; std::memcpy(&h, k.b.data() + 8, 8); return (size_t)(h * 0x9e3779b97f4a7c15ull); } };   // keys are curve points: their bytes are as good as random
    struct KeyEq { bool operator()(const PublicKey &a, const PublicKey &b) const { return a.b == b.b; } };
    hs_ed25519_keyset *keyset_ = nullptr;
    std::map<PublicKey, uint32_t> index_;
    std::unordered_map<PublicKey, Member, KeyHash, KeyEq> fast_;
    uint64_t total_stake_ = 0;
};

namespace detail {
// the structural loop both certificates start with (messages.rs:182-194, 292-304): runs on the
// host, in the reference's order, before any signature is looked at
template <typename Votes, typename NameOf>
inline ConsensusResult check_quorum(const Votes &votes, NameOf name_of, const Committee &committee,
                                    ConsensusError::Kind too_light) {
    Stake weight = 0;
    std::set<PublicKey> used;
    for (const auto &v : votes) {
        const PublicKey &name = name_of(v);
        if (used.count(name)) return ConsensusResult::Err(ConsensusError::AuthorityReuse, name);
        const Stake voting_rights = committee.stake(name);
        if (voting_rights == 0) return ConsensusResult::Err(ConsensusError::UnknownAuthority, name);
        used.insert(name);
        weight += voting_rights;
    }
    if (weight < committee.quorum_threshold()) return ConsensusResult::Err(too_light);
    return ConsensusResult::Ok();
}
// SHA-512 of an arbitrary-length message, truncated to 32 bytes (Block::digest hashes author, round, every
// payload digest and the parent: more than one block)
// the same loop for a committee whose keys are registered (Committee::cache_keys): one hash lookup per vote; idx[i] = the vote's
// position in the key set.  Same errors in the same order as check_quorum: a name seen before is AuthorityReuse, a stranger (or a
// member without stake) UnknownAuthority, too little weight the certificate's own error.
template <typename Votes, typename NameOf>
inline ConsensusResult check_quorum_registered(const Votes &votes, NameOf name_of, const Committee &committee,
                                               ConsensusError::Kind too_light, std::vector<uint32_t> &idx) {
    Stake weight = 0;
    std::vector<uint8_t> seen(committee.authorities.size(), 0);
    idx.resize(votes.size());
    size_t i = 0;
    for (const auto &v : votes) {
        const PublicKey &name = name_of(v);
        const Committee::Member *m = committee.member(name);
        if (m && seen[m->index]) return ConsensusResult::Err(ConsensusError::AuthorityReuse, name);
        if (!m || m->stake == 0) return ConsensusResult::Err(ConsensusError::UnknownAuthority, name);
        seen[m->index] = 1;
        weight += m->stake;
        idx[i++] = m->index;
    }
    if (weight < committee.registered_quorum_threshold()) return ConsensusResult::Err(too_light);
    return ConsensusResult::Ok();
}

inline Digest sha512_32_long(const std::vector<uint8_t> &msg) {
    const hs::u64 H0[8] = HS_SHA512_H0_INIT;
    hs::u64 st[8];
    for (int i = 0; i < 8; i++) st[i] = H0[i];
    std::vector<uint8_t> m(msg);
    const uint64_t bits = 8 * (uint64_t)msg.size();
    m.push_back(0x80);
    while (m.size() % 128 != 112) m.push_back(0);
    for (int i = 0; i < 8; i++) m.push_back(0);                                  // high 64 bits of the length
    for (int i = 7; i >= 0; i--) m.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < m.size(); off += 128) {
        hs::u64 w[16], s2[8];
        for (int i = 0; i < 16; i++) { w[i] = 0; for (int j = 0; j < 8; j++) w[i] = (w[i] << 8) | m[off + 8 * i + j]; }
        for (int i = 0; i < 8; i++) s2[i] = st[i];
        hs::sha512_rounds16<false>(s2, w, 0);
        for (int r = 16; r < 80; r += 16) hs::sha512_rounds16<true>(s2, w, r);
        for (int i = 0; i < 8; i++) st[i] += s2[i];
    }
    Digest d;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) d.b[8 * i + j] = (uint8_t)(st[i] >> (56 - 8 * j));
    return d;
}
inline Digest sha512_32(const uint8_t *msg, size_t len) {  // len <= 111: one padded block
    hs::u64 w[16] = {0}, hh[8];
    uint8_t block[128] = {0};
    std::memcpy(block, msg, len);
    block[len] = 0x80;
    const uint64_t bits = 8 * (uint64_t)len;
    for (int i = 0; i < 8; i++) block[127 - i] = (uint8_t)(bits >> (8 * i));
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 8; j++) w[i] = (w[i] << 8) | block[8 * i + j];
    hs::sha512_oneblock(hh, w);
    Digest d;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) d.b[8 * i + j] = (uint8_t)(hh[i] >> (56 - 8 * j));
    return d;
}
}  // namespace detail

struct QC {                                                // messages.rs:164-169
    Digest hash;
    Round round = 0;
    std::vector<std::pair<PublicKey, Signature>> votes;

    Digest digest() const {                                // messages.rs:201-208: SHA-512(hash || round_le)[..32]
        uint8_t m[40];
        std::memcpy(m, hash.b.data(), 32);
        for (int i = 0; i < 8; i++) m[32 + i] = (uint8_t)(round >> (8 * i));
        return detail::sha512_32(m, 40);
    }
    ConsensusResult verify(const Committee &committee) const {        // messages.rs:180-198
        std::vector<uint32_t> idx;
        const ConsensusResult s = committee.keyset()
            ? detail::check_quorum_registered(votes, [](const auto &v) -> const PublicKey & { return v.first; }, committee, ConsensusError::QCRequiresQuorum, idx)
            : detail::check_quorum(votes, [](const auto &v) -> const PublicKey & { return v.first; }, committee, ConsensusError::QCRequiresQuorum);
        if (s.is_err()) return s;
        if (committee.keyset() && !votes.empty()) {        // every author passed the stake lookup, so every key is in the set
            const Digest d = digest();
            std::vector<uint8_t> sigs(64 * votes.size());
            for (size_t i = 0; i < votes.size(); i++) {
                const auto f = votes[i].second.flatten();
                std::memcpy(&sigs[64 * i], f.data(), 64);
            }
            int rc = hs_ed25519_verify_shared_cached(committee.keyset(), d.b.data(), idx.data(), sigs.data(), votes.size(),
                                                     HS_ED25519_BATCH_RLC, nullptr);
            if (rc < 0) {                                  // engine fault: the votes carry their keys - the CPU verifier answers (detail::settle)
                std::vector<uint8_t> pks(32 * votes.size());
                for (size_t i = 0; i < votes.size(); i++) std::memcpy(&pks[32 * i], votes[i].first.b.data(), 32);
                rc = detail::settle(rc, d.b.data(), 0, pks.data(), sigs.data(), votes.size(), HS_ED25519_BATCH, nullptr);
            }
            return rc == 1 ? ConsensusResult::Ok() : ConsensusResult::Err(ConsensusError::InvalidSignature);
        }
        return Signature::verify_batch(digest(), votes).is_ok() ? ConsensusResult::Ok()
                                                                 : ConsensusResult::Err(ConsensusError::InvalidSignature);
    }
};

struct TC {                                                // messages.rs:283-287
    Round round = 0;
    std::vector<std::tuple<PublicKey, Signature, Round>> votes;

    // messages.rs:290-315.  The reference hashes SHA-512(round || high_qc_round)[..32] per vote on
    // the host and calls Signature::verify n times; here the rounds go to the engine, which builds
    // the digests on the device and verifies every vote in one strict call.
    ConsensusResult verify(const Committee &committee) const {
        std::vector<uint32_t> idx;
        const ConsensusResult s = committee.keyset()
            ? detail::check_quorum_registered(votes, [](const auto &v) -> const PublicKey & { return std::get<0>(v); }, committee, ConsensusError::TCRequiresQuorum, idx)
            : detail::check_quorum(votes, [](const auto &v) -> const PublicKey & { return std::get<0>(v); }, committee, ConsensusError::TCRequiresQuorum);
        if (s.is_err()) return s;
        if (votes.empty()) return ConsensusResult::Ok();
        const bool engine = detail::ensure_init();
        std::vector<uint8_t> pks(32 * votes.size()), sigs(64 * votes.size());
        std::vector<uint64_t> hq(votes.size());
        for (size_t i = 0; i < votes.size(); i++) {
            std::memcpy(&pks[32 * i], std::get<0>(votes[i]).b.data(), 32);
            const auto f = std::get<1>(votes[i]).flatten();
            std::memcpy(&sigs[64 * i], f.data(), 64);
            hq[i] = std::get<2>(votes[i]);
        }
        int rc = HS_ED25519_E_NOT_INIT;
        if (engine && committee.keyset()) {
            rc = hs_ed25519_verify_tc_cached(committee.keyset(), round, hq.data(), idx.data(), sigs.data(), votes.size(), nullptr);
        } else if (engine) {
            rc = hs_ed25519_verify_tc(round, hq.data(), pks.data(), sigs.data(), votes.size(), nullptr);
        }
        if (rc < 0 && SmallBatch::cpu_verify()) {
            // engine fault: the reference's own loop (messages.rs:307-313) - the per-vote digests hashed on the host, strict
            // verification by the CPU verifier - so that a device fault during a view change does not reject a valid TC
            std::vector<uint8_t> dg(32 * votes.size());
            for (size_t i = 0; i < votes.size(); i++) {
                uint8_t m[16];
                for (int b = 0; b < 8; b++) { m[b] = (uint8_t)(round >> (8 * b)); m[8 + b] = (uint8_t)(hq[i] >> (8 * b)); }
                const Digest d = detail::sha512_32(m, 16);
                std::memcpy(&dg[32 * i], d.b.data(), 32);
            }
            rc = detail::settle(rc, dg.data(), 32, pks.data(), sigs.data(), votes.size(), HS_ED25519_STRICT, nullptr);
        }
        return rc == 1 ? ConsensusResult::Ok() : ConsensusResult::Err(ConsensusError::InvalidSignature);
    }
};

// ---- single-signature messages ------------------------------------------
struct Vote {                                              // messages.rs:112-118
    Digest hash;
    Round round = 0;
    PublicKey author;
    Signature signature;
    Digest digest() const {                                // messages.rs:149-156: SHA-512(hash || round_le)[..32]
        uint8_t m[40];
        std::memcpy(m, hash.b.data(), 32);
        for (int i = 0; i < 8; i++) m[32 + i] = (uint8_t)(round >> (8 * i));
        return detail::sha512_32(m, 40);
    }
    ConsensusResult verify(const Committee &committee) const {       // messages.rs:136-146
        if (committee.stake(author) == 0) return ConsensusResult::Err(ConsensusError::UnknownAuthority, author);
        return signature.verify(digest(), author).is_ok() ? ConsensusResult::Ok()
                                                          : ConsensusResult::Err(ConsensusError::InvalidSignature);
    }
};

// core.rs:203-225 + aggregator.rs:74-94, batched.  The reference verifies each incoming vote singly on the Core
// task, then QCMaker::append marks the author used and adds its stake.  Here the structural checks run at
// arrival, signatures wait, and ONE call verifies everything pending as soon as the pending + verified stake
// could form a quorum (or max_pending is reached).  As in the reference an author is `used` only once a signature
// from it has VERIFIED: a forged vote naming authority X is dropped by the flush and X's genuine vote still
// counts.  AuthorityReuse is returned where the reference returns it: for a vote whose author already has an
// accepted vote.  Invalid votes are dropped individually (the per-signature bitmap says which) - the set the
// reference drops one by one.  The QC is made once.
struct VoteIngest {
    using Verifier = std::vector<bool> (*)(const std::vector<Digest> &, const std::vector<std::pair<PublicKey, Signature>> &);
    const Committee &committee;
    Verifier verifier = &Signature::verify_many;           // injectable (tests: the oracle)
    // What a hostile peer can cost (the reference verifies every vote on arrival and holds no queue):
    //   memory - at most per_author_cap candidates wait per author; the one that would exceed it triggers a verification
    //            of everything pending first.  The queue never exceeds per_author_cap x committee size, and max_pending
    //            (default: committee size + 64) verifies before that.
    //   calls  - a verification that could complete the quorum runs at once (the honest case, and one or two bad votes);
    //            from the second such attempt in a row that failed, the next waits for 2, 4, 8 ... max_coalesce pending
    //            votes or for max_wait (tick(), driven by the owner's timer): a forgery stream is verified in growing
    //            groups instead of one GPU call per vote.
    size_t max_pending = 0, per_author_cap = 2, max_coalesce = 64;
    double max_wait_s = 0.002;
    double (*clock)() = &VoteIngest::steady_seconds;
    std::set<PublicKey> used;
    std::vector<Vote> pending;
    std::map<PublicKey, size_t> pend_count;                // author -> candidates waiting
    Stake pend_weight = 0;                                 // stake that could verify: one candidate per author at most
    double oldest = -1;                                    // arrival of the oldest waiting candidate
    unsigned misses = 0;                                   // attempts in a row that could have completed the quorum and did not
    std::vector<std::pair<PublicKey, Signature>> valid;
    std::vector<PublicKey> dropped;
    Stake valid_weight = 0;
    bool done = false, started = false;
    size_t verify_calls = 0;
    Digest hash;
    Round round = 0;

    static double steady_seconds() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    explicit VoteIngest(const Committee &c, size_t max_pending_ = 0)
        : committee(c), max_pending(max_pending_ ? max_pending_ : c.authorities.size() + 64), quorum(c.quorum_threshold()) {}

    struct Outcome { ConsensusResult status; std::unique_ptr<QC> qc; };   // qc set the first time a quorum verified

    // the committee's threshold, taken once: Committee::quorum_threshold() sums the stakes as the reference's does (config.rs:67-72),
    // and asking it per vote - as QCMaker::append does (aggregator.rs:85) - made the 6,667 votes of N = 10,000 cost 66 M map steps,
    // 300 of the 324 ms this path took when it was first timed (round 6, bench.py `vote_ingest_qc_ms`)
    const Stake quorum;
    bool could_complete() const { return valid_weight + pend_weight >= quorum; }
    size_t group_size() const {
        if (misses < 2) return 1;
        const size_t g = misses - 1 >= 20 ? max_coalesce : (size_t)1 << (misses - 1);
        return g < max_coalesce ? g : max_coalesce;
    }
    Outcome add_vote(const Vote &v) {
        if (!started) { hash = v.hash; round = v.round; started = true; }
        if (committee.stake(v.author) == 0) return {ConsensusResult::Err(ConsensusError::UnknownAuthority, v.author), nullptr};
        if (used.count(v.author)) return {ConsensusResult::Err(ConsensusError::AuthorityReuse, v.author), nullptr};
        if (done) return {ConsensusResult::Ok(), nullptr};
        auto it = pend_count.find(v.author);
        if (it != pend_count.end() && it->second > 0)      // only an author who HAS a candidate waiting can repeat one (round 6: the scan
            for (const Vote &p : pending)                  // ran for every vote - 22 M comparisons for the 6,667 votes of N = 10,000)
                if (p.author == v.author && p.signature.flatten() == v.signature.flatten()) return {ConsensusResult::Ok(), nullptr};   // byte-identical to a candidate already waiting
        if (it != pend_count.end() && it->second >= per_author_cap) {
            Outcome o = flush();                           // settle this author's waiting candidates before queueing another
            if (!o.status.is_ok()) return o;
            if (o.qc || done) return o;                    // the flush completed the certificate: hand it over with Ok, never
                                                           // beside an error a caller may stop at
            if (used.count(v.author)) return {ConsensusResult::Err(ConsensusError::AuthorityReuse, v.author), nullptr};
        }
        pending.push_back(v);
        if (pend_count[v.author]++ == 0) pend_weight += committee.stake(v.author);
        if (oldest < 0) oldest = clock();
        if (pending.size() >= max_pending) return flush();
        if (could_complete() && (pending.size() >= group_size() || clock() - oldest >= max_wait_s)) return flush();
        return {ConsensusResult::Ok(), nullptr};
    }
    // the owner's timer: verifies what is waiting once the oldest candidate has waited max_wait_s
    Outcome tick() {
        if (!pending.empty() && !done && could_complete() && clock() - oldest >= max_wait_s) return flush();
        return {ConsensusResult::Ok(), nullptr};
    }
    Outcome flush() {
        const bool could = could_complete();
        if (!pending.empty()) {
            std::vector<Digest> dg;
            std::vector<std::pair<PublicKey, Signature>> votes;
            Vote of_instance; of_instance.hash = hash; of_instance.round = round;
            const Digest same = of_instance.digest();          // one (hash, round) per instance: one digest, not one SHA-512 per vote
            for (const Vote &p : pending) { dg.push_back((p.hash == hash && p.round == round) ? same : p.digest()); votes.push_back({p.author, p.signature}); }
            const std::vector<bool> ok = verifier(dg, votes);
            verify_calls++;
            const bool engine_error = ok.size() != pending.size();
            for (size_t i = 0; !engine_error && i < pending.size(); i++) {
                if (ok[i] && !used.count(pending[i].author)) {
                    used.insert(pending[i].author);
                    valid.push_back(votes[i]);
                    valid_weight += committee.stake(pending[i].author);
                } else if (!ok[i]) {
                    dropped.push_back(pending[i].author);  // the reference warns and drops (core.rs:434-439)
                }
            }
            pending.clear(); pend_count.clear(); pend_weight = 0; oldest = -1;
            if (engine_error) return {ConsensusResult::Err(ConsensusError::InvalidSignature), nullptr};   // engine error: never Ok
        }
        if (!done && valid_weight >= quorum) {
            done = true;                                   // QCMaker: "ensures QC is only made once"
            std::unique_ptr<QC> qc(new QC());
            qc->hash = hash; qc->round = round; qc->votes = valid;
            return {ConsensusResult::Ok(), std::move(qc)};
        }
        if (could && !done) misses++;                      // this attempt could have completed the certificate and did not
        return {ConsensusResult::Ok(), nullptr};
    }
};

// messages.rs:261-263 / core.rs:227-238: every received timeout re-verifies its embedded high_qc - up to N
// identical batch verifications of 2f+1 signatures per view change.  Fingerprints of QCs that verified, LRU.
struct VerifiedQCCache {
    size_t capacity;
    size_t hits = 0, misses = 0;
    explicit VerifiedQCCache(size_t cap = 1024) : capacity(cap) {}
    static Digest fingerprint(const QC &qc) {
        std::vector<uint8_t> m(qc.hash.b.begin(), qc.hash.b.end());
        for (int i = 0; i < 8; i++) m.push_back((uint8_t)(qc.round >> (8 * i)));
        const uint64_t n = qc.votes.size();
        for (int i = 0; i < 8; i++) m.push_back((uint8_t)(n >> (8 * i)));
        for (const auto &v : qc.votes) {
            m.insert(m.end(), v.first.b.begin(), v.first.b.end());
            const auto f = v.second.flatten();
            m.insert(m.end(), f.begin(), f.end());
        }
        return detail::sha512_32_long(m);
    }
    // QC::verify with memoisation; failures are not cached
    ConsensusResult verify(const QC &qc, const Committee &committee) {
        const Digest fp = fingerprint(qc);
        const auto it = index_.find(fp.b);
        if (it != index_.end()) { order_.splice(order_.begin(), order_, it->second); hits++; return ConsensusResult::Ok(); }
        misses++;
        const ConsensusResult r = qc.verify(committee);
        if (r.is_err()) return r;
        order_.push_front(fp.b);
        index_[fp.b] = order_.begin();
        if (order_.size() > capacity) { index_.erase(order_.back()); order_.pop_back(); }
        return r;
    }
 private:
    std::list<std::array<uint8_t, 32>> order_;
    std::map<std::array<uint8_t, 32>, std::list<std::array<uint8_t, 32>>::iterator> index_;
};

struct Timeout {                                           // messages.rs:211-216
    QC high_qc;
    Round round = 0;
    PublicKey author;
    Signature signature;
    Digest digest() const {                                // messages.rs:268-275: SHA-512(round_le || high_qc.round_le)[..32]
        uint8_t m[16];
        for (int i = 0; i < 8; i++) { m[i] = (uint8_t)(round >> (8 * i)); m[8 + i] = (uint8_t)(high_qc.round >> (8 * i)); }
        return detail::sha512_32(m, 16);
    }
    ConsensusResult verify(const Committee &committee, VerifiedQCCache *qc_cache = nullptr) const {   // messages.rs:250-265
        if (committee.stake(author) == 0) return ConsensusResult::Err(ConsensusError::UnknownAuthority, author);
        if (signature.verify(digest(), author).is_err()) return ConsensusResult::Err(ConsensusError::InvalidSignature);
        // `high_qc != QC::genesis()`: QC's PartialEq compares hash and round only (messages.rs:217-221)
        if (high_qc.round != 0 || high_qc.hash != Digest{}) return qc_cache ? qc_cache->verify(high_qc, committee) : high_qc.verify(committee);
        return ConsensusResult::Ok();
    }
};

struct Block {                                             // messages.rs:15-24
    QC qc;
    std::unique_ptr<TC> tc;                                // Option<TC>
    PublicKey author;
    Round round = 0;
    std::vector<Digest> payload;
    Signature signature;
    Digest digest() const {                                // messages.rs:79-90: SHA-512(author || round_le || payload... || qc.hash)[..32]
        std::vector<uint8_t> m(author.b.begin(), author.b.end());
        for (int i = 0; i < 8; i++) m.push_back((uint8_t)(round >> (8 * i)));
        for (const Digest &x : payload) m.insert(m.end(), x.b.begin(), x.b.end());
        m.insert(m.end(), qc.hash.b.begin(), qc.hash.b.end());
        return detail::sha512_32_long(m);
    }
    ConsensusResult verify(const Committee &committee) const {       // messages.rs:55-76
        if (committee.stake(author) == 0) return ConsensusResult::Err(ConsensusError::UnknownAuthority, author);
        if (signature.verify(digest(), author).is_err()) return ConsensusResult::Err(ConsensusError::InvalidSignature);
        if (qc.round != 0 || qc.hash != Digest{}) {        // != QC::genesis()
            const ConsensusResult r = qc.verify(committee);
            if (r.is_err()) return r;
        }
        if (tc) return tc->verify(committee);
        return ConsensusResult::Ok();
    }
};

}  // namespace hotstuff
