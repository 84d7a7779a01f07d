// mirror_tests.cpp - the reference's own test cases for this path (crypto/src/tests/crypto_tests.rs:
// verify_valid_signature, verify_invalid_signature, verify_valid_batch, verify_invalid_batch;
// consensus/src/tests/messages_tests.rs: verify_valid_qc, verify_qc_authority_reuse,
// verify_qc_unknown_authority, verify_qc_insufficient_stake) restated against the C++ mirror
// include/hotstuff_crypto.hpp, which sits on the C ABI - plus the TC::verify shape.
// Keys and signatures come from a fixture file written by tests/test_gpu_cpp_mirror.py (OpenSSL):
// signing is not on the path.  usage: mirror_tests <fixture.bin>     exit code = failed cases
#include <cstdio>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdlib>
#include <vector>
#include "../../include/hotstuff_crypto.hpp"

using namespace hotstuff;

struct Fixture {
    PublicKey pk[4], unknown;
    Digest digest, bad_digest, qc_hash, qc_digest;
    Signature sig[4], qc_sig[4], tc_sig[4];
    Round qc_round, tc_round, tc_hq[4];
    // second part: single-signature messages and the callers either side of the path
    Digest vote_hash, block_payload[2], block_digest;
    Round vote_round, timeout_round, block_round;
    Signature vote_sig[4], timeout_sig[4], block_sig;
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
    ok = ok && rd(f, x.vote_hash.b.data(), 32) && rd(f, &x.vote_round, 8);
    for (auto &g : x.vote_sig) { ok = ok && rd(f, s, 64); g = Signature::from_bytes(s); }
    ok = ok && rd(f, &x.timeout_round, 8);
    for (auto &g : x.timeout_sig) { ok = ok && rd(f, s, 64); g = Signature::from_bytes(s); }
    ok = ok && rd(f, &x.block_round, 8) && rd(f, x.block_payload[0].b.data(), 32) && rd(f, x.block_payload[1].b.data(), 32)
            && rd(f, x.block_digest.b.data(), 32) && rd(f, s, 64);
    x.block_sig = Signature::from_bytes(s);
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
    {
        // forty certificates in flight at once (more than the pool holds): valid ones, one with a default signature,
        // one signed over another digest, an empty one - each entry must equal its own verify_batch
        std::vector<std::pair<PublicKey, Signature>> good, bad;
        for (int i = 3; i >= 1; i--) good.push_back({x.pk[i], x.sig[i]});
        bad = good; bad[2].second = Signature{};
        std::vector<std::pair<Digest, std::vector<std::pair<PublicKey, Signature>>>> certs;
        std::vector<bool> want;
        for (int c = 0; c < 40; c++) {
            if (c % 7 == 3) { certs.push_back({x.digest, bad}); want.push_back(false); }
            else if (c % 7 == 5) { certs.push_back({x.bad_digest, good}); want.push_back(false); }
            else if (c == 20) { certs.push_back({x.digest, {}}); want.push_back(true); }
            else { certs.push_back({x.digest, good}); want.push_back(true); }
        }
        const auto got = Signature::verify_batches(certs);
        bool same = got.size() == want.size();
        for (size_t c = 0; same && c < want.size(); c++) same = got[c].is_ok() == want[c] && got[c].is_ok() == Signature::verify_batch(certs[c].first, certs[c].second).is_ok();
        CHECK("verify_batches: forty certificates in flight, each = its own verify_batch", same);
    }
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
    {   // the structural loop of a registered committee (one hash lookup per vote, round 6) returns the reference's errors in its order
        QC q = qc();
        q.votes.pop_back();
        const auto again = q.votes[0];
        q.votes.push_back(again);
        const ConsensusResult r = q.verify(committee);
        CHECK("verify_qc_authority_reuse (key cache)", r.is_err() && r.err.kind == ConsensusError::AuthorityReuse && r.err.name == again.first);
        QC few = qc();
        few.votes.pop_back();
        const ConsensusResult r2 = few.verify(committee);
        CHECK("verify_qc_insufficient_stake (key cache)", r2.is_err() && r2.err.kind == ConsensusError::QCRequiresQuorum);
        QC both = qc();                                    // a stranger first, a repeated member after it: the stranger is what is reported
        both.votes[0].first = x.unknown;
        both.votes[2] = both.votes[1];
        const ConsensusResult r3 = both.verify(committee);
        CHECK("a stranger before a repeated member is UnknownAuthority (key cache)", r3.is_err() && r3.err.kind == ConsensusError::UnknownAuthority && r3.err.name == x.unknown);
    }
    CHECK("verify_valid_tc (key cache)", tc().verify(committee).is_ok());
    {
        TC t = tc();
        std::get<2>(t.votes[0]) += 1;
        const ConsensusResult r = t.verify(committee);
        CHECK("tampered high_qc_round is InvalidSignature (key cache)", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }
    committee.drop_key_cache();

    // ---- single-signature messages (messages.rs:55-76, 136-146, 250-265)
    auto vote = [&](int i) { Vote v; v.hash = x.vote_hash; v.round = x.vote_round; v.author = x.pk[i]; v.signature = x.vote_sig[i]; return v; };
    CHECK("Vote::verify valid", vote(2).verify(committee).is_ok());
    { Vote v = vote(2); v.round += 1; const ConsensusResult r = v.verify(committee);
      CHECK("Vote for another round is InvalidSignature", r.is_err() && r.err.kind == ConsensusError::InvalidSignature); }
    { Vote v = vote(2); v.author = x.unknown; const ConsensusResult r = v.verify(committee);
      CHECK("Vote of a stranger is UnknownAuthority", r.is_err() && r.err.kind == ConsensusError::UnknownAuthority); }
    {
        Block b; b.qc = qc(); b.author = x.pk[0]; b.round = x.block_round; b.payload = {x.block_payload[0], x.block_payload[1]};
        b.signature = x.block_sig;
        CHECK("Block::digest is SHA-512(author || round_le || payload || qc.hash)[..32]", b.digest() == x.block_digest);
        CHECK("Block::verify valid (signature + embedded QC)", b.verify(committee).is_ok());
        b.tc.reset(new TC(tc()));
        CHECK("Block::verify valid with an embedded TC", b.verify(committee).is_ok());
        std::get<2>(b.tc->votes[0]) += 1;
        CHECK("Block with a bad embedded TC is InvalidSignature", b.verify(committee).is_err());
        b.tc.reset();
        b.payload.pop_back();
        const ConsensusResult r = b.verify(committee);
        CHECK("Block with a tampered payload is InvalidSignature", r.is_err() && r.err.kind == ConsensusError::InvalidSignature);
    }
    // ---- VoteIngest (core.rs:203-225 + aggregator.rs:74-94, batched)
    {
        VoteIngest ing(committee);
        Vote forged[3];
        for (int i = 0; i < 3; i++) { forged[i] = vote(i); forged[i].signature.part1[0] ^= 1; }
        bool none = true;
        for (int i = 0; i < 3; i++) { auto o = ing.add_vote(forged[i]); none = none && o.status.is_ok() && !o.qc; }
        CHECK("VoteIngest: three forgeries could form a quorum -> one call, all three dropped", none && ing.verify_calls == 1
              && ing.dropped.size() == 3 && ing.used.empty());
        auto o0 = ing.add_vote(vote(0));                   // the genuine votes of the same authorities are still welcome
        auto o1 = ing.add_vote(vote(1));
        auto dup = ing.add_vote(vote(1));                  // byte-identical duplicate of a waiting candidate: ignored
        CHECK("VoteIngest: a dropped forgery does not lock its authority out", o0.status.is_ok() && o1.status.is_ok() && dup.status.is_ok()
              && !o0.qc && !o1.qc && !dup.qc && ing.pending.size() == 2);
        auto o2 = ing.add_vote(vote(2));
        CHECK("VoteIngest: QC made once, from exactly the verified votes", o2.qc && o2.qc->votes.size() == 3 && ing.verify_calls == 2
              && o2.qc->verify(committee).is_ok());
        const auto again = ing.add_vote(forged[0]);
        CHECK("VoteIngest: AuthorityReuse only after a signature from the author verified",
              again.status.is_err() && again.status.err.kind == ConsensusError::AuthorityReuse);
        const auto late = ing.add_vote(vote(3));
        CHECK("VoteIngest: late votes after the QC need no verification", late.status.is_ok() && !late.qc && ing.verify_calls == 2);
    }
    // ---- VoteIngest under a hostile peer: bounded queue, coalesced verifications
    {
        VoteIngest ing(committee);
        size_t peak = 0;
        bool quiet = true;
        for (int t = 0; t < 20; t++) {                     // twenty different garbage votes naming authority 0
            Vote g = vote(0);
            g.signature.part1[0] ^= 1; g.signature.part1[1] ^= (uint8_t)(t + 1);
            auto o = ing.add_vote(g);
            quiet = quiet && o.status.is_ok() && !o.qc;
            if (ing.pending.size() > peak) peak = ing.pending.size();
        }
        CHECK("VoteIngest: a flood naming one authority stays bounded (queue <= per_author_cap, one call per cap votes)",
              quiet && peak <= 2 && ing.verify_calls <= 10 && ing.used.empty() && ing.max_pending == committee.authorities.size() + 64);
        auto a0 = ing.add_vote(vote(0)), a1 = ing.add_vote(vote(1)), a2 = ing.add_vote(vote(2));
        CHECK("VoteIngest: the flooded authority's genuine vote still counts", a0.status.is_ok() && a1.status.is_ok() && a2.qc
              && a2.qc->votes.size() == 3 && a2.qc->verify(committee).is_ok());
    }
    {
        static double now = 0;
        VoteIngest ing(committee);
        ing.clock = +[]() { return now; };
        ing.add_vote(vote(0)); ing.add_vote(vote(1));      // quorum 3 of 4: two verified-to-be, one missing
        size_t calls0 = ing.verify_calls;
        Vote f2 = vote(2); f2.signature.part1[0] ^= 1;
        Vote f3 = vote(3); f3.signature.part1[0] ^= 1;
        ing.add_vote(f2);                                  // could complete -> verified at once: miss 1
        ing.add_vote(f3);                                  // again: miss 2 -> from now on groups of two
        const size_t calls1 = ing.verify_calls;
        Vote f2b = vote(2); f2b.signature.part1[0] ^= 2;
        auto w = ing.add_vote(f2b);                        // waits: one pending, group size two
        const bool waited = ing.verify_calls == calls1 && ing.pending.size() == 1 && !w.qc;
        auto t0 = ing.tick();                              // not due yet
        const bool no_call_yet = ing.verify_calls == calls1;
        now += 0.01;
        auto last = ing.add_vote(vote(3));                 // the honest last vote: past the deadline it is verified with the group
        CHECK("VoteIngest: a forgery stream is verified in groups, the honest last vote completes the QC",
              calls1 - calls0 == 2 && waited && !t0.qc && no_call_yet && ing.verify_calls == calls1 + 1 && last.qc && last.qc->votes.size() == 3);
    }
    {
        // the per-author cap's flush completes the certificate for the very author whose third vote triggered it:
        // the QC must come back with Ok (it was returned beside AuthorityReuse, in Python it was lost altogether)
        static double now2 = 0;
        VoteIngest ing(committee);
        ing.clock = +[]() { return now2; };
        ing.max_wait_s = 1.0;
        auto forged = [&](int i, uint8_t tag) { Vote f = vote(i); f.signature.part1[0] ^= 1; f.signature.part1[1] ^= tag; return f; };
        ing.add_vote(vote(0)); ing.add_vote(vote(1));
        ing.add_vote(forged(2, 1));                        // miss 1
        ing.add_vote(forged(3, 1));                        // miss 2: groups of two
        ing.add_vote(forged(2, 2)); ing.add_vote(forged(3, 2));   // miss 3: groups of four
        const bool staged = ing.misses == 3 && ing.pending.empty() && ing.valid.size() == 2;
        auto w0 = ing.add_vote(forged(2, 3));
        auto w1 = ing.add_vote(vote(2));                   // the genuine vote waits beside the forgery
        const bool waiting = !w0.qc && !w1.qc && ing.pending.size() == 2 && !ing.done;
        auto o = ing.add_vote(forged(2, 4));               // cap reached -> flush -> quorum
        CHECK("VoteIngest: a cap flush that completes the quorum returns the QC with Ok", staged && waiting && o.status.is_ok() && o.qc
              && o.qc->votes.size() == 3 && ing.done && o.qc->verify(committee).is_ok());
        const auto again = ing.add_vote(vote(2));
        CHECK("VoteIngest: ... and the author counts as reused only afterwards", again.status.is_err()
              && again.status.err.kind == ConsensusError::AuthorityReuse);
    }
    // ---- Timeout::verify and the verified-QC cache (messages.rs:250-265, core.rs:227-238)
    {
        VerifiedQCCache cache;
        bool all_ok = true;
        for (int i = 0; i < 4; i++) {
            Timeout t; t.high_qc = qc(); t.round = x.timeout_round; t.author = x.pk[i]; t.signature = x.timeout_sig[i];
            all_ok = all_ok && t.verify(committee, &cache).is_ok();
        }
        CHECK("timeout storm: four timeouts, the shared high_qc verified once", all_ok && cache.misses == 1 && cache.hits == 3);
        Timeout bad; bad.high_qc = qc(); bad.high_qc.votes[1].second = x.qc_sig[0]; bad.round = x.timeout_round; bad.author = x.pk[0];
        bad.signature = x.timeout_sig[0];
        const ConsensusResult r1 = bad.verify(committee, &cache), r2 = bad.verify(committee, &cache);
        CHECK("a forged high_qc is rejected every time (failures are not cached)", r1.is_err() && r2.is_err() && cache.misses == 3);
        Timeout g; g.round = x.timeout_round; g.author = x.pk[0]; g.signature = x.timeout_sig[0];
        g.high_qc.votes.push_back({x.pk[0], Signature{}});   // hash 0 / round 0 = genesis by QC's PartialEq, votes ignored
        const ConsensusResult rg = g.verify(committee, &cache);
        CHECK("Timeout: high_qc equal to genesis by (hash, round) is not verified", rg.is_err() && rg.err.kind == ConsensusError::InvalidSignature && cache.misses == 3);
    }
    // ---- the small-batch policy hook
    {
        static int cpu_calls = 0;
        SmallBatch::cpu_verify() = [](const uint8_t *, size_t, const uint8_t *, const uint8_t *, size_t n, uint32_t, uint8_t *bm) -> int {
            cpu_calls++; if (bm) std::memset(bm, 0xff, (n + 7) / 8); return 1; };
        const size_t th = SmallBatch::threshold(false);
        CHECK("hs_ed25519_min_gpu_batch() is a small positive number", th >= 2 && th <= 64 && SmallBatch::threshold(true) <= th);
        const bool one = x.sig[3].verify(x.bad_digest, x.pk[3]).is_ok();          // the stub says Ok: proves the call went to the hook
        std::vector<std::pair<PublicKey, Signature>> three;
        for (int i = 1; i <= 3; i++) three.push_back({x.pk[i], x.sig[i]});
        const bool b3 = Signature::verify_batch(x.bad_digest, three).is_ok();
        CHECK("calls below the threshold go to the embedder's CPU path when one is set", one && b3 && cpu_calls == 2);
        SmallBatch::cpu_verify() = nullptr;
        CHECK("without a CPU path every call goes to the GPU", x.sig[3].verify(x.bad_digest, x.pk[3]).is_err());
    }
    // ---- the multi-GPU entry points of the C ABI, driven here on the one device that is open
    {
        uint8_t pks[3 * 32], sigs[3 * 64], bm[1] = {0};
        for (int i = 0; i < 3; i++) {
            std::memcpy(pks + 32 * i, x.pk[i].b.data(), 32);
            const auto f = x.qc_sig[i].flatten();
            std::memcpy(sigs + 64 * i, f.data(), 64);
        }
        size_t lo = 1, hi = 0;
        const bool plan = hs_ed25519_shard_plan(3, 1, 0, &lo, &hi) == 0 && lo == 0 && hi == 3;
        const int ok = hs_ed25519_verify_shared_sharded(x.qc_digest.b.data(), pks, sigs, 3, HS_ED25519_BATCH, bm);
        sigs[64 + 40] ^= 1;
        uint8_t bm2[1] = {0xff};
        const int bad = hs_ed25519_verify_shared_sharded(x.qc_digest.b.data(), pks, sigs, 3, HS_ED25519_STRICT, bm2);
        CHECK("hs_ed25519_verify_shared_sharded on one device slot", hs_ed25519_device_count() == 1 && plan && ok == 1 && bm[0] == 7 && bad == 0 && bm2[0] == 5);
    }
    hs_ed25519_shutdown();
    // ---- an engine fault is not a verdict (round 6): with the library shut down every C-ABI call returns
    // HS_ED25519_E_NOT_INIT.  Without a CPU verifier each entry point is Err - never Ok; with one (here the test oracle's per-
    // signature laws, loaded at run time: what the Rust shim's dalek bodies are) EVERY entry point answers there - a valid TC is
    // accepted, an invalid one rejected - where until round 6 only verify_batch's shim text fell back.
    {
        std::vector<std::pair<PublicKey, Signature>> votes;
        for (int i = 3; i >= 1; i--) votes.push_back({x.pk[i], x.sig[i]});
        TC tc; tc.round = x.tc_round;
        for (int i = 0; i < 4; i++) tc.votes.push_back(std::make_tuple(x.pk[i], x.tc_sig[i], x.tc_hq[i]));
        TC tc_bad = tc; std::get<2>(tc_bad.votes[1]) += 1;       // a vote that claims another high_qc_round than it signed
        std::vector<Digest> dgs(3, x.digest);
        SmallBatch::cpu_verify() = nullptr;
        const bool dead = hs_ed25519_verify_shared(x.digest.b.data(), x.pk[0].b.data(), x.sig[0].flatten().data(), 1, HS_ED25519_BATCH, nullptr) < 0;
        CHECK("engine shut down, no CPU verifier: every entry point is Err, none Ok",
              dead && x.sig[3].verify(x.digest, x.pk[3]).is_err() && Signature::verify_batch(x.digest, votes).is_err() &&
              Signature::verify_many(dgs, votes).empty() && qc().verify(committee).is_err() && tc.verify(committee).is_err());
        typedef void (*many_fn)(const uint8_t *, size_t, size_t, const uint8_t *, const uint8_t *, size_t, int, uint8_t *, int);
        static many_fn oracle_many = nullptr;
        const char *so = std::getenv("HS_ORACLE_SO");
        void *h = dlopen(so ? so : "oracle/libhs_oracle.so", RTLD_NOW);
        oracle_many = h ? (many_fn)dlsym(h, "hs_oracle_verify_many") : nullptr;
        CHECK("the test oracle (the CPU verifier of the fault-injection case) loads", oracle_many != nullptr);
        if (oracle_many) {
            static int cpu_calls = 0;
            SmallBatch::cpu_verify() = [](const uint8_t *dg, size_t stride, const uint8_t *pks, const uint8_t *sigs, size_t n, uint32_t mode, uint8_t *bm) -> int {
                cpu_calls++;
                std::vector<uint8_t> out(n ? n : 1);
                oracle_many(dg, 32, stride, pks, sigs, n, mode == HS_ED25519_STRICT ? 1 : 0, out.data(), 1);
                int all = 1;
                for (size_t i = 0; i < n; i++) { if (bm && out[i]) bm[i >> 3] |= (uint8_t)(1u << (i & 7)); if (!out[i]) all = 0; }
                return all; };
            const unsigned long long f0 = detail::engine_faults().load();
            // (three or four votes are below hs_ed25519_min_gpu_batch and would go to the CPU path by POLICY: a threshold of zero sends
            //  every call to the engine first, so that what is tested is the fault path.  The library reads the variable per call.)
            setenv("HS_ED25519_MIN_GPU_BATCH", "0", 1);
            CHECK("with a threshold of zero every call reaches the engine first", SmallBatch::threshold(false) == 0 && !SmallBatch::on_cpu(1, false));
            CHECK("engine fault + CPU verifier: a valid signature / batch / QC / TC is accepted through the fallback",
                  x.sig[3].verify(x.digest, x.pk[3]).is_ok() && Signature::verify_batch(x.digest, votes).is_ok() &&
                  Signature::verify_many(dgs, votes) == std::vector<bool>(3, true) && qc().verify(committee).is_ok() && tc.verify(committee).is_ok());
            QC q_bad = qc(); q_bad.votes[1].second = x.qc_sig[0];
            std::vector<std::pair<PublicKey, Signature>> bad_votes = votes; bad_votes[2].second = Signature{};
            CHECK("engine fault + CPU verifier: an invalid one is rejected",
                  x.sig[3].verify(x.bad_digest, x.pk[3]).is_err() && Signature::verify_batch(x.digest, bad_votes).is_err() &&
                  Signature::verify_many(dgs, bad_votes) == std::vector<bool>({true, true, false}) && q_bad.verify(committee).is_err() && tc_bad.verify(committee).is_err());
            std::vector<std::pair<Digest, std::vector<std::pair<PublicKey, Signature>>>> certs = {{x.digest, votes}, {x.digest, bad_votes}, {x.digest, {}}, {x.bad_digest, votes}};
            const auto got = Signature::verify_batches(certs);
            CHECK("engine fault + CPU verifier: verify_batches answers every certificate on the CPU",
                  got.size() == 4 && got[0].is_ok() && got[1].is_err() && got[2].is_ok() && got[3].is_err());
            CHECK("every one of those calls was counted as an engine fault", detail::engine_faults().load() >= f0 + 12 && cpu_calls >= 12);
            SmallBatch::cpu_verify() = nullptr;
        }
    }
    printf("%d failed\n", failures);
    if (const char *q = std::getenv("HS_TEST_QUICK_EXIT")) if (q[0] == '1') { fflush(stdout); fflush(stderr); _exit(failures); }   // see tests/cpp/stress_host.cpp
    return failures;
}
