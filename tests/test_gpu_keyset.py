"""GPU parity of the committee key cache (hs_ed25519_keyset_* / *_cached, verify_cached.h)
against the oracle and the edge vectors, through the C-ABI."""
import os

import pytest
import torch

import synth
from conftest import bits, oracle_verify_many
from gpu_helpers import DeviceRunner
from hotstuff_amd import _lib
from hotstuff_amd.crypto import Digest, PublicKey, Signature
from hotstuff_amd.messages import QC, TC, Committee, InvalidSignature

pytestmark = pytest.mark.gpu
H = bytes.fromhex
CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")


@pytest.fixture(scope="module", autouse=True)
def engine():
    assert torch.cuda.is_available()
    _lib.init(0, 8192)
    yield


def cached_bits(ks, dg, keys, sg, n, mode, shared):
    packed, _ = ks.indices(keys)
    ok, bm = ks.verify_packed(dg, packed, sg, n, mode, shared, want_bitmap=True)
    b = bits(bm, n)
    assert ok == all(b)
    return b


def test_edge_vectors_with_every_key_registered(edge_vectors):
    n = len(edge_vectors)
    keys = [H(v["pk"]) for v in edge_vectors]
    ks = _lib.KeySet(keys)
    assert ks.hbm_bytes() == n * (32 * 129 * 128 + 36 + 128)      # affine 128-byte table entries + key bytes + flags + the Niels record of the RLC path
    dg = b"".join(H(v["msg"]) for v in edge_vectors)
    sg = b"".join(H(v["sig"]) for v in edge_vectors)
    # KeySet.index keeps the first position of a repeated key; the tables of equal keys are equal
    got = cached_bits(ks, dg, keys, sg, n, _lib.STRICT, False)
    for v, g in zip(edge_vectors, got):
        assert bool(g) == v["strict"], ("strict", v["name"])
    got = cached_bits(ks, dg, keys, sg, n, _lib.BATCH, False)
    for v, g in zip(edge_vectors, got):
        assert bool(g) == v["cofactorless"], ("batch", v["name"])
    ks.close()


@pytest.mark.parametrize("n", [1, 3, 16, 17, 64, 65, 500])
@pytest.mark.parametrize("shared", [True, False])
def test_sizes_vs_oracle_with_corruption(oracle, n, shared):
    dg, pk, sg = synth.make_batch(n, shared=shared, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=40)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    stride = 0 if shared else 32
    for mode in (_lib.BATCH, _lib.STRICT):
        got = cached_bits(ks, dg, keys, sg, n, mode, shared)
        assert got == oracle_verify_many(oracle, dg, stride, pk, sg, n, mode)
        assert [i for i in range(n) if not got[i]] == bad
    ks.close()


def test_cached_equals_cold_on_config2_and_unknown_index(oracle):
    n = 1024
    dg, pk, sg = synth.make_batch(n, shared=False, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    dg2, pk2, sg2, bad = synth.corrupt(dg, pk, sg, n, False)
    # the committee's keys are the registered ones; a vote whose key bytes were corrupted in
    # flight is simply a different (unknown) key -> index UNKNOWN -> rejected
    keys2 = [pk2[32 * i:32 * i + 32] for i in range(n)]
    packed, idx = ks.indices(keys2)
    assert any(i == _lib.KeySet.UNKNOWN for i in idx)
    ok, bm = ks.verify_packed(dg2, packed, sg2, n, _lib.STRICT, False, want_bitmap=True)
    got = bits(bm, n)
    assert not ok and [i for i in range(n) if not got[i]] == bad
    cold_ok, cold_bm = _lib.verify_packed(dg2, pk2, sg2, n, _lib.STRICT, shared=False, want_bitmap=True)
    assert bits(cold_bm, n) == got
    # keys presented in a different order than registered
    perm = list(reversed(range(n)))
    rp = [keys[i] for i in perm]
    rs = b"".join(sg[64 * i:64 * i + 64] for i in perm)
    rd = b"".join(dg[32 * i:32 * i + 32] for i in perm)
    assert cached_bits(ks, rd, rp, rs, n, _lib.STRICT, False) == b"\x01" * n
    ks.close()


def test_qc_and_tc_through_the_committee_cache(openssl_vectors):
    fx = openssl_vectors["reference_fixture"]
    committee = Committee({PublicKey(H(k["pk"])): 1 for k in fx["keys"]}, 100, cache_keys=True)
    votes = [(PublicKey(H(v["pk"])), Signature.from_bytes(H(v["sig"]))) for v in fx["qc_votes"]]
    assert QC(Digest(), 1, votes).verify(committee) is None            # messages_tests.rs:8-10
    assert committee.keyset is not None and committee.keyset.n == 4
    broken = [votes[0], votes[1], (votes[2][0], votes[1][1])]
    with pytest.raises(InvalidSignature):
        QC(Digest(), 1, broken).verify(committee)
    names = list(committee.authorities)
    seeds = {H(k["pk"]): H(k["seed"]) for k in fx["keys"]}
    tc = TC(9, [])
    tvotes = []
    for i, name in enumerate(names[:3]):
        d = tc.vote_digest(i + 1)
        _, sg = synth.sign(seeds[name.b], d.b)
        tvotes.append((name, Signature.from_bytes(sg), i + 1))
    assert TC(9, tvotes).verify(committee) is None
    tvotes[0] = (tvotes[0][0], tvotes[0][1], 7)
    with pytest.raises(InvalidSignature):
        TC(9, tvotes).verify(committee)


def test_qc_10000_validators_cached_properties():
    """BASELINE config 3 with the committee cached: N = 10,000 keys (5.3 GB of tables in HBM),
    a QC of 6,667 votes by a subset; oracle-free properties."""
    N, n = 10000, 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    filler = [synth.key_seed(10_000_000 + i) for i in range(N - n)]     # never-voting members: any 32 bytes
    ks = _lib.KeySet(keys + filler)
    assert ks.n == N
    assert cached_bits(ks, dg, keys, sg, n, _lib.BATCH, True) == b"\x01" * n
    for victim in (0, 3333, n - 1):
        s2 = bytearray(sg); s2[64 * victim + 9] ^= 0x20
        got = cached_bits(ks, dg, keys, bytes(s2), n, _lib.BATCH, True)
        assert [i for i in range(n) if not got[i]] == [victim]
    assert cached_bits(ks, bytes(32), keys, sg, n, _lib.BATCH, True) == b"\x00" * n
    ks.close()


def test_device_resident_cached_api(oracle):
    n = 777
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    s2 = bytearray(sg); s2[64 * 5 + 40] ^= 1; s2[64 * 700] ^= 2
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_sg = t(dg), t(bytes(s2))
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    d_bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    d_nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    torch.cuda.synchronize()        # the fills above ran on torch's stream; the engine uses its own
    ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, d_bm.data_ptr(),
                      d_nr.data_ptr(), ctx.stream)
    torch.cuda.synchronize()
    got = bits(bytes(d_bm.cpu().tolist()), n)
    assert [i for i in range(n) if not got[i]] == [5, 700] and int(d_nr.item()) == 2
    ctx.close(); ks.close()


def test_tc_digests_built_on_device_cold_and_cached(oracle):
    """hs_ed25519_verify_tc[_cached]: (round, high_qc_round_i) in, verdicts equal to hashing on
    the host (messages.rs:307-311) + strict verification by the oracle."""
    import hashlib
    import struct
    n, rnd = 400, 77
    hq = [(i * 7919) % 60 for i in range(n)]
    digests = [hashlib.sha512(struct.pack("<QQ", rnd, h)).digest()[:32] for h in hq]
    pks, sigs = bytearray(), bytearray()
    for i in range(n):
        p, s = synth.sign(synth.key_seed(i), digests[i])
        pks += p; sigs += s
    bad = [3, 64, 399]
    sigs[64 * 3 + 5] ^= 1                    # corrupted signature
    hq[64] += 1                               # vote claims another high_qc_round than it signed
    sigs[64 * 399 + 40] ^= 0x80
    dg = b"".join(hashlib.sha512(struct.pack("<QQ", rnd, h)).digest()[:32] for h in hq)
    expect = oracle_verify_many(oracle, dg, 32, bytes(pks), bytes(sigs), n, 1)
    assert [i for i in range(n) if not expect[i]] == bad
    ok, bm = _lib.verify_tc_packed(rnd, hq, bytes(pks), bytes(sigs))
    assert not ok and bits(bm, n) == expect
    ks = _lib.KeySet([bytes(pks[32 * i:32 * i + 32]) for i in range(n)])
    ok, bm = _lib.verify_tc_packed(rnd, hq, bytes(pks), bytes(sigs), keyset=ks)
    assert not ok and bits(bm, n) == expect
    # the two-kernel cold path (n > 32,768 by size), forced per context (key-cache calls of that context: the default kernel)
    r = DeviceRunner(n, kernel=_lib.KERNEL_LANE)
    assert r.tc_bits(rnd, hq, bytes(pks), bytes(sigs)) == expect
    keys = [bytes(pks[32 * i:32 * i + 32]) for i in range(n)]
    assert r.tc_bits(rnd, hq, None, bytes(sigs), keyset=ks, keys=keys) == expect
    r.close()
    ks.close()


@pytest.mark.parametrize("kernel", [_lib.KERNEL_LANE, _lib.KERNEL_FUSED, "cached-lane2r", "cached-c"])
def test_torsion_fuzz_every_path(torsion_fuzz, kernel):
    """160 randomized torsion-mix signatures labelled by the big-int oracle, through both cold
    kernels and the two key-cache forms (forced per context: two lanes + R wavefront, chains + batched finish
    by compression), both modes."""
    vec = torsion_fuzz
    n = len(vec)
    dg = b"".join(H(v["msg"]) for v in vec)
    keys = [H(v["pk"]) for v in vec]
    sg = b"".join(H(v["sig"]) for v in vec)
    if isinstance(kernel, str):
        ks = _lib.KeySet(keys)
        r = DeviceRunner(n, kernel={"cached-lane2r": _lib.KERNEL_FUSED, "cached-c": _lib.KERNEL_CACHED_C}[kernel])
        strict = r.bits(dg, None, sg, n, _lib.STRICT, False, keyset=ks, keys=keys)
        batch = r.bits(dg, None, sg, n, _lib.BATCH, False, keyset=ks, keys=keys)
        r.close()
        ks.close()
    else:
        r = DeviceRunner(n, kernel=kernel)
        strict = r.bits(dg, b"".join(keys), sg, n, _lib.STRICT, False)
        batch = r.bits(dg, b"".join(keys), sg, n, _lib.BATCH, False)
        r.close()
    assert [bool(b) for b in strict] == [v["strict"] for v in vec]
    assert [bool(b) for b in batch] == [v["cofactorless"] for v in vec]


@pytest.mark.parametrize("kernel", [_lib.KERNEL_FUSED, _lib.KERNEL_CACHED_C])
def test_cached_kernel_variants_agree(oracle, edge_vectors, kernel):
    """The default cached kernel up to 98,304 signatures is the three-wavefront one (two lanes per signature + concurrent
    R decompression), above that two-lane chains + a batched finish by compression (k_cached_chain2 + k_cached_finish).
    Force each on small inputs: edge vectors in both modes and a
    corrupted QC vs the oracle (700 signatures: 64 finish lanes walking 11 each)."""
    r = DeviceRunner(1024, kernel=kernel)
    n = len(edge_vectors)
    keys = [H(v["pk"]) for v in edge_vectors]
    ks = _lib.KeySet(keys)
    dg = b"".join(H(v["msg"]) for v in edge_vectors)
    sg = b"".join(H(v["sig"]) for v in edge_vectors)
    assert [bool(b) for b in r.bits(dg, None, sg, n, _lib.STRICT, False, keyset=ks, keys=keys)] == [v["strict"] for v in edge_vectors]
    assert [bool(b) for b in r.bits(dg, None, sg, n, _lib.BATCH, False, keyset=ks, keys=keys)] == [v["cofactorless"] for v in edge_vectors]
    ks.close()
    n = 700
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=30)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    for mode in (_lib.BATCH, _lib.STRICT):
        assert r.bits(dg, None, sg, n, mode, True, keyset=ks, keys=keys) == oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
    ks.close()
    r.close()


def test_cached_large_qc_takes_the_two_lane_form():
    """n > 16,384 with the committee cached (k_verify_cached_lane2): 20,001 votes (N = 30,000 validators, 19.8 GB of
    tables) by distinct keys; properties on the whole bitmap."""
    n = 20001
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys + [synth.key_seed(30_000_000 + i) for i in range(30000 - n)])
    victims = [0, 9999, 16384, 20000]
    s2 = bytearray(sg)
    for v in victims:
        s2[64 * v + 3] ^= 0x40
    got = cached_bits(ks, dg, keys, bytes(s2), n, _lib.BATCH, True)
    assert [i for i in range(n) if not got[i]] == victims
    assert cached_bits(ks, dg, keys, sg, n, _lib.BATCH, True) == b"\x01" * n
    ks.close()


def test_heavy_random_corruption_8192_cold_and_cached(oracle):
    """8,192 signatures of which about half get one random bit flipped somewhere in (digest, key,
    R, s) - far denser than the 3 % parity mix - both modes, cold and cached, bit-exact vs the
    oracle.  Duplicated pairs included."""
    import random
    n = 8192
    dg, pk, sg = synth.make_batch(4096, shared=False, cache_dir=CACHE)
    dg, pk, sg = bytearray(dg * 2), bytearray(pk * 2), bytearray(sg * 2)      # every pair appears twice
    rnd = random.Random(2024)
    for i in range(n):
        if rnd.random() < 0.5:
            which = rnd.randrange(4)
            bit = 1 << rnd.randrange(8)
            if which == 0:
                dg[32 * i + rnd.randrange(32)] ^= bit
            elif which == 1:
                pk[32 * i + rnd.randrange(32)] ^= bit
            else:
                sg[64 * i + (rnd.randrange(32) if which == 2 else 32 + rnd.randrange(32))] ^= bit
    dg, pk, sg = bytes(dg), bytes(pk), bytes(sg)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(sorted(set(keys)))
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = oracle_verify_many(oracle, dg, 32, pk, sg, n, mode)
        _, bm = _lib.verify_packed(dg, pk, sg, n, mode, shared=False, want_bitmap=True)
        assert bits(bm, n) == expect
        assert cached_bits(ks, dg, keys, sg, n, mode, False) == expect
        assert 3000 < sum(expect) < 5200
    ks.close()


def test_destroy_waits_for_tickets_in_flight_on_the_key_set(oracle):
    """hs_ed25519_keyset_destroy does not synchronise the device (it would break a graph capture on another thread); it waits
    for the host API's uncollected work on the set instead.  A thread destroys the set while tickets on it are out: it returns
    only after the last one is collected, and their verdicts are the oracle's - the tables were not freed under the kernels."""
    import threading
    n = 20000
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    dgc, pkc, sgc, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=30)
    expect = oracle_verify_many(oracle, dgc, 0, pkc, sgc, n, 0)
    keys = [pkc[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    packed, idx = ks.indices(keys)
    tickets = [_lib.Ticket(dgc, packed, sgc, n, _lib.BATCH, shared=True, keyset=ks) for _ in range(3)]
    small = _lib.Ticket(dgc, packed[:4 * 500], sgc[:64 * 500], 500, _lib.BATCH, shared=True, keyset=ks)      # a group member holds it too
    done = threading.Event()

    def destroy():
        ks.close()
        done.set()

    t = threading.Thread(target=destroy)
    t.start()
    assert not done.wait(0.3), "destroy returned with four tickets out on the key set"
    for tk in tickets[:2]:
        ok, bm = tk.wait()
        assert bits(bm, n) == expect
    assert not done.wait(0.1), "destroy returned with two tickets out on the key set"
    ok, bm = small.wait()
    assert bits(bm, 500) == expect[:500]
    ok, bm = tickets[2].wait()
    assert bits(bm, n) == expect
    assert done.wait(5), "destroy did not return after the last ticket was collected"
    t.join()


def test_destroy_is_not_held_by_a_collected_burst_left_open(oracle):
    """every ticket of a burst flagged HS_ED25519_SUBMIT_MORE ("launches when any ticket is waited on") has been
    collected, the submitting thread submits nothing else - the thread's own reference to its open group used to pin the key set
    (and the pool context) until that thread's next submit or exit: hs_ed25519_keyset_destroy spun ten seconds and leaked the
    tables.  The last collected member now gives both back."""
    import time
    n = 300
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    packed, idx = ks.indices(keys)
    tks = [_lib.Ticket(dg, packed, sg, n, _lib.BATCH, shared=True, keyset=ks, more=True) for _ in range(3)]
    for tk in tks:
        ok, bm = tk.wait()
        assert ok and all(bits(bm, n))
    t0 = time.perf_counter()
    ks.close()
    dt = time.perf_counter() - t0
    assert dt < 1.0, "keyset_destroy took %.2f s after every ticket of the open burst had been collected" % dt
    # the thread's stale reference to that group is harmless: its next burst starts a group of its own
    ks2 = _lib.KeySet(keys)
    packed2, _ = ks2.indices(keys)
    tk = _lib.Ticket(dg, packed2, sg, n, _lib.BATCH, shared=True, keyset=ks2, more=True)
    ok, bm = tk.wait()
    assert ok and all(bits(bm, n))
    ks2.close()
