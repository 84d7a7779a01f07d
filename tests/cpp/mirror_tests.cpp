// mirror_tests.cpp - the reference's own test cases for this path (crypto/src/tests/crypto_tests.rs:
// verify_valid_signature, verify_invalid_signature, verify_valid_batch, verify_invalid_batch;
// consensus/src/tests/messages_tests.rs: verify_valid_qc, verify_qc_authority_reuse,
// verify_qc_unknown_authority, verify_qc_insufficient_stake) restated against the C++ mirror
// include/hotstuff_crypto.hpp, which sits on the C ABI - plus the TC::verify shape.
// Keys and signatures come from a fixture file written by tests/test_gpu_cpp_mirror.py (OpenSSL):
// signing is not on the path.  usage: mirror_tests <fixture.bin>     exit code = failed cases
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/hotstuff_crypto.hpp"

using namespace hotstuff;

struct Fixture {
    PublicKey pk[4], unknown;
    Digest digest, bad_digest, qc_hash, qc_digest;
    Signature sig[4], qc_sig[4], tc_sig[4];
    Round qc_round, tc_round, tc_hq[4];
};
static bool rd(FILE *f, void *p, size_t n) { return fread(p, 1, n, f) == n; }
static bool load(const char *path, Fixture &x) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    bool ok = true;
    uint8_t s[64];
    for (auto &k : x.pk) ok = ok && rd(f, k.b.data(), 32);
    ok = ok && rd(f, x.digest.b.data(), 32) && rd(f, x.bad_digest.b.data(), 32);
    for (auto &g : x.sig) { ok = ok && rd(f, s, 64); g = Signature::from_bytes(s); }
    ok = ok && rd(f, x.qc_hash.b.data(), 32) && rd(f, &x.qc_round, 8) && rd(f, x.qc_digest.b.data(), 32);
    for (auto &g : x.qc_sig) { ok = ok && rd(f, s, 64); g = Signature::from_bytes(s); }
    ok = ok && rd(f, x.unknown.b.data(), 32) && rd(f, &x.tc_round, 8) && rd(f, x.tc_hq, 32);
    for (auto &g : x.tc_sig) { ok = ok && rd(f, s, 64); g = Signature::from_bytes(s); }
    fclose(f);
    return ok;
}

static int failures = 0;
#define CHECK(name, cond) do { const bool c_ = (cond); printf("%s %s\n", c_ ? "ok  " : "FAIL", name); if (!c_) failures++; } while (0)

int main(int argc, char **argv) {
    Fixture x;
    if (argc < 2 || !load(argv[1], x)) { fprintf(stderr, "cannot read the fixture\n"); return 100; }
    Committee committee;                                   // common.rs: four authorities, stake 1 each
    for (const auto &k : x.pk) committee.authorities[k] = 1;
    CHECK("committee quorum_threshold is 3 of 4", committee.quorum_threshold() == 3);

    // ---- crypto_tests.rs
    CHECK("verify_valid_signature", x.sig[3].verify(x.digest, x.pk[3]).is_ok());
    CHECK("verify_invalid_signature", x.sig[3].verify(x.bad_digest, x.pk[3]).is_err());
    {
        std::vector<std::pair<PublicKey, Signature>> votes;
        for (int i = 3; i >= 1; i--) votes.push_back({x.pk[i], x.sig[i]});
        CHECK("verify_valid_batch", Signature::verify_batch(x.digest, votes).is_ok());
    }
    {
        std::vector<std::pair<PublicKey, Signature>> votes;
        for (int i = 3; i >= 2; i--) votes.push_back({x.pk[i], x.sig[i]});
        votes.push_back({x.pk[1], Signature{}});           // Signature::default(): 64 zero bytes
        CHECK("verify_invalid_batch", Signature::verify_batch(x.digest, votes).is_err());
    }
    CHECK("verify_batch of nothing is Ok", Signature::verify_batch(x.digest, {}).is_ok());
    CHECK("wrong key is Err", x.sig[3].verify(x.digest, x.pk[2]).is_err());

    // ---- messages_tests.rs
    auto qc = [&]() {                                      // common.rs qc(): votes of three authorities
        QC q; q.hash = x.qc_hash; q.round = x.qc_round;
        for (int i = 0; i < 3; i++) q.votes.push_back({x.pk[i], x.qc_sig[i]});
        return q;
    };
    CHECK("QC::digest is SHA-512(hash || round_le)[..32]", qc().digest() == x.qc_digest);
    CHECK("verify_valid_qc", qc().verify(committee).is_ok());
    {
        QC q = qc();
        q.votes.pop_back();
        const auto vote = q.votes[0];
        q.votes.push_back(vote);
        const ConsensusResult r = q.verify(committee);
        CHECK("verify_qc_authority_reuse", r.is_err() && r.err.kind == ConsensusError::AuthorityReuse && r.err.name == vote.first);
    }
    {
        QC q = qc();
        const Signature sig = q.votes.back().second;
        q.votes.pop_back();
        q.votes.push_back({x.unknown, sig});
        const ConsensusResult r = q.verify(committee);
        CHECK("verify_qc_unknown_authority", r.is_err() && r.err.kind == ConsensusError::UnknownAuthority && r.err.name == x.unknown);
    }
    {
        QC q = qc();
        q.votes.pop_back();
        const ConsensusResult r = q.verify(committee);
        CHECK("verify_qc_insufficient_stake", r.is_err() && r.err.kind == ConsensusError::QCRequiresQuorum);
    }
    {
        QC q = qc();
        q.votes[1].second = x.qc_sig[0];                   // a valid signature under the wrong author
        const ConsensusResult r = q.verify(committee);
        CHECK("QC with a misattributed signature is InvalidSignature", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }
    {
        QC q = qc();
        q.round += 1;                                      // signatures no longer cover the digest
        const ConsensusResult r = q.verify(committee);
        CHECK("QC for another round is InvalidSignature", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }

    // ---- TC::verify (messages.rs:290-315)
    auto tc = [&]() {
        TC t; t.round = x.tc_round;
        for (int i = 0; i < 3; i++) t.votes.push_back({x.pk[i], x.tc_sig[i], x.tc_hq[i]});
        return t;
    };
    CHECK("verify_valid_tc", tc().verify(committee).is_ok());
    {
        TC t = tc();
        std::get<2>(t.votes[1]) += 1;                      // high_qc_round is part of what was signed
        const ConsensusResult r = t.verify(committee);
        CHECK("TC with a tampered high_qc_round is InvalidSignature", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }
    {
        TC t = tc();
        t.votes.pop_back();
        const ConsensusResult r = t.verify(committee);
        CHECK("verify_tc_insufficient_stake", r.is_err() && r.err.kind == ConsensusError::TCRequiresQuorum);
    }
    // ---- the same certificates with the committee's keys registered once (the key cache)
    CHECK("committee.cache_keys()", committee.cache_keys() && committee.keyset() != nullptr);
    CHECK("verify_valid_qc (key cache)", qc().verify(committee).is_ok());
    {
        QC q = qc();
        q.votes[1].second = x.qc_sig[0];
        const ConsensusResult r = q.verify(committee);
        CHECK("misattributed signature is InvalidSignature (key cache)", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }
    {
        QC q = qc();
        const Signature sig = q.votes.back().second;
        q.votes.pop_back();
        q.votes.push_back({x.unknown, sig});
        const ConsensusResult r = q.verify(committee);
        CHECK("verify_qc_unknown_authority (key cache)", r.is_err() && r.err.kind == ConsensusError::UnknownAuthority);
    }
    CHECK("verify_valid_tc (key cache)", tc().verify(committee).is_ok());
    {
        TC t = tc();
        std::get<2>(t.votes[0]) += 1;
        const ConsensusResult r = t.verify(committee);
        CHECK("tampered high_qc_round is InvalidSignature (key cache)", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }
    committee.drop_key_cache();
    hs_ed25519_shutdown();
    printf("%d failed\n", failures);
    return failures;
}
