"""GPU: QC::verify / TC::verify fed with the vote elements as received (bincode, base64 keys; SURVEY.md 8f
row 3) and batch signing for load generation (row 4), through the C ABI, against Python's base64 / struct,
the oracle and OpenSSL."""
import base64
import hashlib
import os
import struct

import pytest
import torch

import synth
from conftest import bits, oracle_verify_many
from hotstuff_amd import _lib
from test_wire_and_sign_emu import element, python_canonical, wire_cases

pytestmark = pytest.mark.gpu
CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")


@pytest.fixture(scope="module", autouse=True)
def engine():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.init(0, 8192)
    yield


@pytest.mark.parametrize("n", [1, 3, 64, 1000, 6667])
def test_qc_votes_from_the_received_frame(oracle, n):
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=20)
    frames = b"".join(element(base64.b64encode(pk[32 * i:32 * i + 32]).decode(), sg[64 * i:64 * i + 64]) for i in range(n))
    for mode in (_lib.BATCH, _lib.STRICT):
        expect = oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
        ok, bm = _lib.verify_wire(frames, n, digest=dg, mode=mode)
        assert ok == all(expect) and bits(bm, n) == expect
    if n >= 64:     # a non-canonical element is rejected whatever its signature says
        f2 = bytearray(frames)
        f2[116 * 5 + 8 + 9] = ord("*")
        f2[116 * 17] = 43
        expect = bytearray(oracle_verify_many(oracle, dg, 0, pk, sg, n, 0))
        expect[5] = expect[17] = 0
        ok, bm = _lib.verify_wire(bytes(f2), n, digest=dg, mode=_lib.BATCH)
        assert not ok and bits(bm, n) == bytes(expect)


def test_tc_votes_from_the_received_frame(oracle):
    n, rnd = 400, 21
    hq = [(i * 7) % 19 for i in range(n)]
    pks, sigs, frames = bytearray(), bytearray(), bytearray()
    for i in range(n):
        d = hashlib.sha512(struct.pack("<QQ", rnd, hq[i])).digest()[:32]
        p, s = synth.sign(synth.key_seed(31000 + i), d)
        if i in (4, 250):
            s = s[:33] + bytes([s[33] ^ 2]) + s[34:]
        pks += p; sigs += s
        frames += element(base64.b64encode(p).decode(), s) + struct.pack("<Q", hq[i] + (1 if i == 99 else 0))   # vote 99 claims another round
    dg = b"".join(hashlib.sha512(struct.pack("<QQ", rnd, h + (1 if i == 99 else 0))).digest()[:32] for i, h in enumerate(hq))
    expect = oracle_verify_many(oracle, dg, 32, bytes(pks), bytes(sigs), n, 1)
    assert [i for i in range(n) if not expect[i]] == [4, 99, 250]
    ok, bm = _lib.verify_wire(bytes(frames), n, tc_round=rnd)
    assert not ok and bits(bm, n) == expect


def test_malformed_elements_on_the_device():
    pk, sg, cases = wire_cases()
    dev = torch.device("cuda", 0)
    n = len(cases)
    frames = torch.frombuffer(bytearray(b"".join(el for _, el, _ in cases)), dtype=torch.uint8).to(dev)
    d_pk = torch.zeros(32 * n, dtype=torch.uint8, device=dev)
    d_sg = torch.zeros(64 * n, dtype=torch.uint8, device=dev)
    nbad = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(64)
    torch.cuda.synchronize(dev)
    _lib.check(_lib.lib().hs_ed25519_wire_decode_device(ctx._h, frames.data_ptr(), 116, n, d_pk.data_ptr(), d_sg.data_ptr(), None,
                                                        nbad.data_ptr(), None), "wire_decode")
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
    o_pk, o_sg = bytes(d_pk.cpu().tolist()), bytes(d_sg.cpu().tolist())
    for i, (name, el, canonical) in enumerate(cases):
        if canonical:
            assert o_pk[32 * i:32 * i + 32] == python_canonical(el) and o_sg[64 * i:64 * i + 64] == el[52:], name
        else:
            assert o_sg[64 * i + 32:64 * i + 64] == b"\xff" * 32, name
    assert int(nbad.item()) == sum(1 for c in cases if not c[2])
    ctx.close()


def test_batch_signing_is_bit_exact_and_verifies(oracle):
    n = 5000
    seeds = b"".join(synth.key_seed(40000 + i) for i in range(n))
    pks = _lib.public_keys(seeds, n)
    for i in (0, 1, 777, n - 1):
        assert pks[32 * i:32 * i + 32] == synth.sign(synth.key_seed(40000 + i), b"x" * 32)[0]
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pks[32 * i:32 * i + 32] for i in range(n))
    dgs = b"".join(synth.msg_digest(i) for i in range(n))
    sigs = _lib.sign_batch(secrets, dgs, n, shared=False)
    for i in (0, 5, 4999):
        assert sigs[64 * i:64 * i + 64] == synth.sign(synth.key_seed(40000 + i), synth.msg_digest(i))[1]    # OpenSSL
    assert all(oracle_verify_many(oracle, dgs, 32, pks, sigs, n, 1))
    ok, _ = _lib.verify_packed(dgs, pks, sigs, n, _lib.STRICT, shared=False)
    assert ok
    shared = synth.shared_digest(3)
    sigs2 = _lib.sign_batch(secrets, shared, n, shared=True)
    ok, _ = _lib.verify_packed(shared, pks, sigs2, n, _lib.BATCH_RLC, shared=True)
    assert ok
