"""The C++ host-side mirror of the reference's interface (include/hotstuff_crypto.hpp:
Digest / PublicKey / Signature{verify, verify_batch}, Committee, QC::verify, TC::verify over the
C ABI) run through the reference's own test cases (tests/cpp/mirror_tests.cpp restates
crypto_tests.rs and messages_tests.rs).  The reference is compiled code, so its host side is
mirrored in C++ as well as in Python; signing is off the path and done here with OpenSSL."""
import hashlib
import os
import struct
import subprocess

import pytest
import torch

import synth
from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "mirror_tests.cpp")


def _build(tmp_path):
    exe = str(tmp_path / "mirror_tests")
    libdir = os.path.join(ROOT, "hotstuff_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unknown-pragmas", SRC, "-o", exe,
                    "-L" + libdir, "-lhs_ed25519", "-Wl,-rpath," + libdir], check=True, capture_output=True)
    return exe


def _fixture(tmp_path):
    seeds = [synth.key_seed(7000 + i) for i in range(4)]
    digest = hashlib.sha512(b"Hello, world!").digest()[:32]
    bad_digest = hashlib.sha512(b"Bad message!").digest()[:32]
    pks, sigs = zip(*(synth.sign(s, digest) for s in seeds))
    qc_hash, qc_round = hashlib.sha512(b"block").digest()[:32], 7
    qc_digest = hashlib.sha512(qc_hash + struct.pack("<Q", qc_round)).digest()[:32]
    qc_sigs = [synth.sign(s, qc_digest)[1] for s in seeds]
    unknown = synth.sign(synth.key_seed(7999), digest)[0]
    tc_round, tc_hq = 9, [3, 5, 5, 8]
    tc_sigs = [synth.sign(s, hashlib.sha512(struct.pack("<QQ", tc_round, hq)).digest()[:32])[1] for s, hq in zip(seeds, tc_hq)]
    blob = (b"".join(pks) + digest + bad_digest + b"".join(sigs) + qc_hash + struct.pack("<Q", qc_round) + qc_digest
            + b"".join(qc_sigs) + unknown + struct.pack("<Q", tc_round) + struct.pack("<4Q", *tc_hq) + b"".join(tc_sigs))
    # single-signature messages: votes over SHA-512(hash || round), timeouts over SHA-512(round || high_qc.round),
    # a block by authority 0 over SHA-512(author || round || payload... || qc.hash) - messages.rs:79-90, 149-156, 268-275
    vote_hash, vote_round = hashlib.sha512(b"a block").digest()[:32], 11
    vote_digest = hashlib.sha512(vote_hash + struct.pack("<Q", vote_round)).digest()[:32]
    vote_sigs = [synth.sign(s, vote_digest)[1] for s in seeds]
    timeout_round = 12
    timeout_digest = hashlib.sha512(struct.pack("<QQ", timeout_round, qc_round)).digest()[:32]
    timeout_sigs = [synth.sign(s, timeout_digest)[1] for s in seeds]
    block_round = 13
    payload = [hashlib.sha512(b"batch %d" % i).digest()[:32] for i in range(2)]
    block_digest = hashlib.sha512(pks[0] + struct.pack("<Q", block_round) + b"".join(payload) + qc_hash).digest()[:32]
    block_sig = synth.sign(seeds[0], block_digest)[1]
    blob += (vote_hash + struct.pack("<Q", vote_round) + b"".join(vote_sigs) + struct.pack("<Q", timeout_round)
             + b"".join(timeout_sigs) + struct.pack("<Q", block_round) + b"".join(payload) + block_digest + block_sig)
    path = str(tmp_path / "fixture.bin")
    with open(path, "wb") as f:
        f.write(blob)
    return path


def _run(tmp_path):
    # HS_ORACLE_SO: the CPU verifier of the fault-injection case (test infrastructure, loaded by the test program at run time)
    env = dict(os.environ, HS_ORACLE_SO=os.path.join(ROOT, "oracle", "libhs_oracle.so"))
    r = subprocess.run([_build(tmp_path), _fixture(tmp_path)], capture_output=True, text=True, timeout=120, env=env)
    lines = [l for l in r.stdout.splitlines() if l[:4] in ("ok  ", "FAIL")]
    return r.returncode, {l[5:]: l.startswith("ok") for l in lines}


@pytest.mark.gpu
def test_reference_test_cases_through_the_cpp_mirror(tmp_path):
    assert torch.cuda.is_available()
    rc, cases = _run(tmp_path)
    assert len(cases) == 59, cases
    assert rc == 0 and all(cases.values()), [k for k, v in cases.items() if not v]


def test_cpp_mirror_builds_and_never_says_ok_without_a_gpu(tmp_path):
    """Compiles against include/ and links libhs_ed25519.so without a GPU.  Run here, the
    structural checks (which precede the crypto, messages.rs:182-194) and QC::digest behave as in
    the reference; with no device every signature check is Err - the mirror has no CPU fallback."""
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    rc, cases = _run(tmp_path)
    for name in ("committee quorum_threshold is 3 of 4", "QC::digest is SHA-512(hash || round_le)[..32]",
                 "verify_qc_authority_reuse", "verify_qc_unknown_authority", "verify_qc_insufficient_stake",
                 "verify_tc_insufficient_stake", "verify_invalid_signature", "verify_invalid_batch",
                 "Block::digest is SHA-512(author || round_le || payload || qc.hash)[..32]",
                 "Vote of a stranger is UnknownAuthority", "hs_ed25519_min_gpu_batch() is a small positive number",
                 "calls below the threshold go to the embedder's CPU path when one is set",
                 "without a CPU path every call goes to the GPU"):
        assert cases[name], name
    for name in ("verify_qc_authority_reuse (key cache)", "verify_qc_insufficient_stake (key cache)",
                 "a stranger before a repeated member is UnknownAuthority (key cache)"):
        assert cases[name], name          # (no GPU: the key set cannot be registered, the plain loop answers - the same errors)
    for name in ("verify_valid_signature", "verify_valid_batch", "verify_valid_qc", "verify_valid_tc",
                 "committee.cache_keys()", "verify_valid_qc (key cache)", "verify_valid_tc (key cache)",
                 "Vote::verify valid", "Block::verify valid (signature + embedded QC)",
                 "VoteIngest: QC made once, from exactly the verified votes",
                 "timeout storm: four timeouts, the shared high_qc verified once",
                 "hs_ed25519_verify_shared_sharded on one device slot"):
        assert not cases[name], name          # would need the GPU: must not be Ok
    assert rc != 0
    # an engine fault is not a verdict (round 6): no GPU here, so EVERY call is an engine fault - without a CPU verifier all Err,
    # with one (the test oracle) valid certificates are accepted and invalid ones rejected, in every entry point
    for name in ("engine shut down, no CPU verifier: every entry point is Err, none Ok",
                 "the test oracle (the CPU verifier of the fault-injection case) loads",
                 "with a threshold of zero every call reaches the engine first",
                 "engine fault + CPU verifier: a valid signature / batch / QC / TC is accepted through the fallback",
                 "engine fault + CPU verifier: an invalid one is rejected",
                 "engine fault + CPU verifier: verify_batches answers every certificate on the CPU",
                 "every one of those calls was counted as an engine fault"):
        assert cases[name], name


if __name__ == "__main__":      # python tests/test_gpu_cpp_mirror.py fixture <path>: the fixture file alone (the sanitizer runs of the mirror)
    import pathlib
    import shutil
    import sys
    import tempfile
    if len(sys.argv) == 3 and sys.argv[1] == "fixture":
        with tempfile.TemporaryDirectory() as td:
            shutil.copy(_fixture(pathlib.Path(td)), sys.argv[2])
