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
//
// Rust's Result<(), E> becomes a small value type with is_ok() / is_err() so that the parity
// tests read like the reference's own (crypto_tests.rs, messages_tests.rs).  There is no CPU
// fallback: an engine error (no GPU, HIP failure) is an Err like any other, never an Ok.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <tuple>
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
inline bool ensure_init() {
    static const int rc = hs_ed25519_init(0, 8192);      // idempotent and thread-safe in the library
    return rc == 0;
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
        if (!detail::ensure_init()) return CryptoResult{false};
        const auto sig = flatten();
        return detail::to_result(hs_ed25519_verify_multi(digest.b.data(), public_key.b.data(), sig.data(), 1,
                                                         HS_ED25519_STRICT, nullptr));
    }
    // lib.rs:206-219: one digest shared by every (key, signature) pair; an empty batch is Ok.
    // Vec<(PublicKey, Signature)> has no guaranteed layout, so the votes are copied into packed buffers.
    static CryptoResult verify_batch(const Digest &digest, const std::vector<std::pair<PublicKey, Signature>> &votes) {
        if (!detail::ensure_init()) return CryptoResult{false};
        std::vector<uint8_t> pks(32 * votes.size()), sigs(64 * votes.size());
        for (size_t i = 0; i < votes.size(); i++) {
            std::memcpy(&pks[32 * i], votes[i].first.b.data(), 32);
            std::memcpy(&sigs[64 * i], votes[i].second.part1.data(), 32);
            std::memcpy(&sigs[64 * i + 32], votes[i].second.part2.data(), 32);
        }
        return detail::to_result(hs_ed25519_verify_shared(digest.b.data(), pks.data(), sigs.data(), votes.size(),
                                                          HS_ED25519_BATCH, nullptr));
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
        for (const auto &a : authorities) { pks.insert(pks.end(), a.first.b.begin(), a.first.b.end()); index_[a.first] = i++; }
        if (hs_ed25519_keyset_create(&keyset_, pks.data(), authorities.size()) != 0) { keyset_ = nullptr; index_.clear(); return false; }
        return true;
    }
    void drop_key_cache() { if (keyset_) hs_ed25519_keyset_destroy(keyset_); keyset_ = nullptr; index_.clear(); }
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
    hs_ed25519_keyset *keyset_ = nullptr;
    std::map<PublicKey, uint32_t> index_;
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
        const ConsensusResult s = detail::check_quorum(votes, [](const auto &v) -> const PublicKey & { return v.first; },
                                                       committee, ConsensusError::QCRequiresQuorum);
        if (s.is_err()) return s;
        if (committee.keyset() && !votes.empty()) {        // every author passed the stake lookup, so every key is in the set
            const Digest d = digest();
            std::vector<uint32_t> idx(votes.size());
            std::vector<uint8_t> sigs(64 * votes.size());
            for (size_t i = 0; i < votes.size(); i++) {
                idx[i] = committee.key_index(votes[i].first);
                const auto f = votes[i].second.flatten();
                std::memcpy(&sigs[64 * i], f.data(), 64);
            }
            const int rc = hs_ed25519_verify_shared_cached(committee.keyset(), d.b.data(), idx.data(), sigs.data(), votes.size(),
                                                           HS_ED25519_BATCH, nullptr);
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
        const ConsensusResult s = detail::check_quorum(votes, [](const auto &v) -> const PublicKey & { return std::get<0>(v); },
                                                       committee, ConsensusError::TCRequiresQuorum);
        if (s.is_err()) return s;
        if (votes.empty()) return ConsensusResult::Ok();
        if (!detail::ensure_init()) return ConsensusResult::Err(ConsensusError::InvalidSignature);
        std::vector<uint8_t> pks(32 * votes.size()), sigs(64 * votes.size());
        std::vector<uint64_t> hq(votes.size());
        for (size_t i = 0; i < votes.size(); i++) {
            std::memcpy(&pks[32 * i], std::get<0>(votes[i]).b.data(), 32);
            const auto f = std::get<1>(votes[i]).flatten();
            std::memcpy(&sigs[64 * i], f.data(), 64);
            hq[i] = std::get<2>(votes[i]);
        }
        int rc;
        if (committee.keyset()) {
            std::vector<uint32_t> idx(votes.size());
            for (size_t i = 0; i < votes.size(); i++) idx[i] = committee.key_index(std::get<0>(votes[i]));
            rc = hs_ed25519_verify_tc_cached(committee.keyset(), round, hq.data(), idx.data(), sigs.data(), votes.size(), nullptr);
        } else {
            rc = hs_ed25519_verify_tc(round, hq.data(), pks.data(), sigs.data(), votes.size(), nullptr);
        }
        return rc == 1 ? ConsensusResult::Ok() : ConsensusResult::Err(ConsensusError::InvalidSignature);
    }
};

}  // namespace hotstuff
