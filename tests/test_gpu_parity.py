"""GPU parity tests proper: the HIP path, called through the C-ABI (ctypes ->
libhs_ed25519.so), against the golden fixtures, the edge-vector set and the CPU oracle on
the same seeded inputs; size-independent properties at BASELINE.json's full sizes.
Bit-exact: this is integer work, the accept bitmap must be identical.
Never reads /root/reference (it does not exist on the GPU box)."""
import ctypes
import os

import pytest
import torch

import synth
from conftest import bits, oracle_verify_many
from gpu_helpers import DeviceRunner
from hotstuff_amd import _lib
from hotstuff_amd.crypto import CryptoError, Digest, PublicKey, Signature
from hotstuff_amd.messages import QC, TC, Committee, InvalidSignature

pytestmark = pytest.mark.gpu
H = bytes.fromhex
CACHE = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")


@pytest.fixture(scope="module", autouse=True)
def engine():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    _lib.init(0, 8192)
    yield


def gpu_bits(dg, pk, sg, n, mode, shared):
    ok, bm = _lib.verify_packed(dg, pk, sg, n, mode, shared=shared, want_bitmap=True)
    b = bits(bm, n)
    assert ok == all(b)
    return b


def test_openssl_vectors(openssl_vectors):
    # (the RFC 8032 vectors have 0-2 byte messages; the boundary takes 32-byte digests only,
    #  crypto/src/lib.rs:21 - they pin the oracle in test_oracle_golden.py instead)
    vec = openssl_vectors["openssl_sign"]
    n = len(vec)
    dg = b"".join(H(v["msg"]) for v in vec)
    pk = b"".join(H(v["pk"]) for v in vec)
    sg = b"".join(H(v["sig"]) for v in vec)
    for mode in (_lib.BATCH, _lib.STRICT):
        assert gpu_bits(dg, pk, sg, n, mode, False) == b"\x01" * n
    # every recorded single-bit tampering is rejected (OpenSSL's own verdict is in the fixture)
    t_dg, t_pk, t_sg, expect = bytearray(), bytearray(), bytearray(), []
    for v in vec:
        for t in v["tamper"]:
            f = {"sig": bytearray(H(v["sig"])), "pk": bytearray(H(v["pk"])), "msg": bytearray(H(v["msg"]))}
            f[t["field"]][t["byte"]] ^= 1 << t["bit"]
            t_dg += f["msg"]; t_pk += f["pk"]; t_sg += f["sig"]
            expect.append(1 if t["openssl_accepts"] else 0)
    m = len(expect)
    for mode in (_lib.BATCH, _lib.STRICT):
        assert gpu_bits(bytes(t_dg), bytes(t_pk), bytes(t_sg), m, mode, False) == bytes(expect)


def test_reference_own_test_cases(openssl_vectors):
    """crypto_tests.rs:50-115 and messages_tests.rs:8-10 through the interface mirror."""
    fx = openssl_vectors["reference_fixture"]
    order = fx["pop_order"]
    pk = [PublicKey(H(fx["keys"][i]["pk"])) for i in order]
    sig = [Signature.from_bytes(H(s)) for s in fx["hello_sigs"]]
    hello, bad = Digest(H(fx["hello_digest"])), Digest(H(fx["bad_digest"]))
    assert sig[0].verify(hello, pk[0]) is None                       # verify_valid_signature
    with pytest.raises(CryptoError):
        sig[0].verify(bad, pk[0])                                    # verify_invalid_signature
    assert Signature.verify_batch(hello, list(zip(pk[:3], sig[:3]))) is None   # verify_valid_batch
    with pytest.raises(CryptoError):                                 # verify_invalid_batch
        Signature.verify_batch(hello, list(zip(pk[:3], sig[:2] + [Signature.default()])))
    assert Signature.verify_batch(hello, []) is None                 # empty iterator -> Ok
    committee = Committee({PublicKey(H(k["pk"])): 1 for k in fx["keys"]}, 100)
    votes = [(PublicKey(H(v["pk"])), Signature.from_bytes(H(v["sig"]))) for v in fx["qc_votes"]]
    assert QC(Digest(), 1, votes).verify(committee) is None          # verify_valid_qc
    broken = [votes[0], votes[1], (votes[2][0], votes[1][1])]
    with pytest.raises(InvalidSignature):
        QC(Digest(), 1, broken).verify(committee)


def test_tc_verify_shape(oracle):
    """TC::verify (messages.rs:290-315): per-vote digests, strict semantics, one GPU call."""
    n = 4
    names = [PublicKey(synth.sign(synth.key_seed(i), b"")[0]) for i in range(n)]
    committee = Committee({k: 1 for k in names})
    tc = TC(9, [])
    votes = []
    for i in range(3):
        d = tc.vote_digest(i + 1)
        _, sg = synth.sign(synth.key_seed(i), d.b)
        votes.append((names[i], Signature.from_bytes(sg), i + 1))
    assert TC(9, votes).verify(committee) is None
    votes[1] = (votes[1][0], votes[1][1], 5)      # signed round differs from the claimed one
    with pytest.raises(InvalidSignature):
        TC(9, votes).verify(committee)


def test_edge_vectors_both_modes(edge_vectors):
    n = len(edge_vectors)
    dg = b"".join(H(v["msg"]) for v in edge_vectors)
    pk = b"".join(H(v["pk"]) for v in edge_vectors)
    sg = b"".join(H(v["sig"]) for v in edge_vectors)
    got = gpu_bits(dg, pk, sg, n, _lib.STRICT, False)
    for v, g in zip(edge_vectors, got):
        assert bool(g) == v["strict"], ("strict", v["name"])
    got = gpu_bits(dg, pk, sg, n, _lib.BATCH, False)
    for v, g in zip(edge_vectors, got):
        assert bool(g) == v["cofactorless"], ("batch", v["name"])


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 94, 95, 128, 1000])
@pytest.mark.parametrize("shared", [True, False])
def test_sizes_vs_oracle_with_corruption(oracle, n, shared):
    dg, pk, sg = synth.make_batch(n, shared=shared, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=40)
    stride = 0 if shared else 32
    for mode in (_lib.BATCH, _lib.STRICT):
        got = gpu_bits(dg, pk, sg, n, mode, shared)
        assert got == oracle_verify_many(oracle, dg, stride, pk, sg, n, mode)
        assert [i for i in range(n) if not got[i]] == bad


def test_config2_1024_tuples_bit_exact(oracle):
    """BASELINE config 2: 1,024 distinct (digest, pk, sig) tuples, ~3 % corrupted (SURVEY 8d mix)."""
    n = 1024
    dg, pk, sg = synth.make_batch(n, shared=False, cache_dir=CACHE)
    assert gpu_bits(dg, pk, sg, n, _lib.STRICT, False) == b"\x01" * n
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, False)
    for mode in (_lib.BATCH, _lib.STRICT):
        got = gpu_bits(dg, pk, sg, n, mode, False)
        assert got == oracle_verify_many(oracle, dg, 32, pk, sg, n, mode)
        assert [i for i in range(n) if not got[i]] == bad and len(bad) > 10


def test_config3_qc_10000_validators_properties():
    """BASELINE config 3: N = 10,000 -> n = 6,667 signatures over one digest.  Full-size,
    oracle-free properties: all-valid accepts; a single flipped bit clears exactly that
    signature's bit; the verdict is idempotent and invariant under permutation."""
    n = 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    ok, bm = _lib.verify_packed(dg, pk, sg, n, _lib.BATCH, shared=True, want_bitmap=True)
    assert ok and bits(bm, n) == b"\x01" * n
    ok2, bm2 = _lib.verify_packed(dg, pk, sg, n, _lib.BATCH, shared=True, want_bitmap=True)
    assert ok2 and bm2 == bm                                         # idempotent
    for victim in (0, 4242, n - 1):
        s2 = bytearray(sg); s2[64 * victim + 17] ^= 0x04
        ok, bm = _lib.verify_packed(dg, pk, bytes(s2), n, _lib.BATCH, shared=True, want_bitmap=True)
        b = bits(bm, n)
        assert not ok and [i for i in range(n) if not b[i]] == [victim]
    # permutation: reverse the votes, corrupt one, the cleared bit moves with it
    rp = b"".join(pk[32 * i:32 * i + 32] for i in reversed(range(n)))
    rs = bytearray(b"".join(sg[64 * i:64 * i + 64] for i in reversed(range(n))))
    rs[64 * 10 + 40] ^= 1
    ok, bm = _lib.verify_packed(dg, rp, bytes(rs), n, _lib.BATCH, shared=True, want_bitmap=True)
    b = bits(bm, n)
    assert not ok and [i for i in range(n) if not b[i]] == [10]
    # wrong shared digest: every signature rejected
    ok, bm = _lib.verify_packed(bytes(32), pk, sg, n, _lib.BATCH, shared=True, want_bitmap=True)
    assert not ok and bits(bm, n) == b"\x00" * n


def test_device_resident_api_matches_host_api(oracle):
    """hs_ed25519_verify_device on torch-owned HBM buffers and torch's current stream."""
    n = 1000
    dg, pk, sg = synth.make_batch(n, shared=False, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, False, rate_byte=30)
    dev = torch.device("cuda:0")
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    d_bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    d_nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    torch.cuda.synchronize()        # the fills above ran on torch's stream; the engine uses its own
    ctx.verify(d_dg.data_ptr(), 32, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.STRICT, d_bm.data_ptr(),
               d_nr.data_ptr(), ctx.stream)
    torch.cuda.synchronize()
    got = bits(bytes(d_bm.cpu().tolist()), n)
    assert got == oracle_verify_many(oracle, dg, 32, pk, sg, n, 1)
    assert int(d_nr.item()) == len(bad)
    assert bytes(d_bm.cpu().tolist())[(n + 7) // 8:] == bytes(len(d_bm) - (n + 7) // 8)   # tail bits zero
    ctx.close()


def test_argument_errors_are_errors_not_accepts():
    ctx = _lib.DeviceContext(64)
    with pytest.raises(_lib.EngineError):
        ctx.verify(8, 32, 16, 16, 4, 0, 16)            # misaligned device pointers
    with pytest.raises(_lib.EngineError):
        ctx.verify(0, 32, 0, 0, 4, 0, 0)               # null pointers
    with pytest.raises(_lib.EngineError):
        _lib.check(_lib.lib().hs_ed25519_verify_multi(None, None, None, 3, 0, None), "verify")
    ctx.close()


def test_concurrent_host_calls_are_thread_safe(oracle):
    """the reference's tests call verify from several threads of one process."""
    import threading
    n = 200
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    _, pk_bad, sg_bad, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=40)
    results = {}

    def work(k):
        p, s = (pk_bad, sg_bad) if k % 2 else (pk, sg)
        results[k] = [_lib.verify_packed(dg, p, s, n, _lib.BATCH, shared=True, want_bitmap=True) for _ in range(3)]

    th = [threading.Thread(target=work, args=(k,)) for k in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k, rs in results.items():
        for ok, bm in rs:
            b = bits(bm, n)
            assert [i for i in range(n) if not b[i]] == (bad if k % 2 else [])
            assert ok == (k % 2 == 0)


def test_single_gpu_sharded_path_over_rccl():
    """hotstuff_amd.sharding with the real GPU verifier and the nccl (= RCCL) backend at
    world_size 1 (gpurun boxes have one GPU; N > 1 is the driver's scaling run)."""
    import torch.distributed as dist
    from hotstuff_amd.sharding import GpuShardVerifier, verify_sharded
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        n = 1500
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
        s2 = bytearray(sg); s2[64 * 77] ^= 1
        dev = torch.device("cuda:0")
        t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        v = GpuShardVerifier(t(dg), 0, t(pk), t(bytes(s2)), _lib.BATCH, n)
        ok, bm = verify_sharded(v, n)
        b = bits(bytes(bm.cpu().tolist()), n)
        assert not ok and [i for i in range(n) if not b[i]] == [77]
        # the same shard through the committee key cache
        ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n)])
        v = GpuShardVerifier(t(dg), 0, None, t(bytes(s2)), _lib.BATCH, n, keyset=ks,
                             key_idx=torch.arange(n, dtype=torch.int32, device=dev))
        ok, bm = verify_sharded(v, n)
        b = bits(bytes(bm.cpu().tolist()), n)
        assert not ok and [i for i in range(n) if not b[i]] == [77]
        ks.close()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("kernel", [_lib.KERNEL_LANE, _lib.KERNEL_FUSED])
def test_every_kernel_variant_has_the_same_accept_set(oracle, edge_vectors, kernel):
    """The two cold-path forms (fused four-lanes-per-signature kernel; k_prepare + one lane per
    signature) are selected by batch size in production; force each one (per context) over the
    edge vectors and a corrupted batch so the default test run covers both at small sizes too."""
    r = DeviceRunner(1024, kernel=kernel)
    n = len(edge_vectors)
    dg = b"".join(H(v["msg"]) for v in edge_vectors)
    pk = b"".join(H(v["pk"]) for v in edge_vectors)
    sg = b"".join(H(v["sig"]) for v in edge_vectors)
    assert [bool(b) for b in r.bits(dg, pk, sg, n, _lib.STRICT, False)] == [v["strict"] for v in edge_vectors]
    assert [bool(b) for b in r.bits(dg, pk, sg, n, _lib.BATCH, False)] == [v["cofactorless"] for v in edge_vectors]
    n = 333
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=30)
    for mode in (_lib.BATCH, _lib.STRICT):
        assert r.bits(dg, pk, sg, n, mode, True) == oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
    r.close()


def test_large_batch_takes_the_two_kernel_path(oracle):
    """n > 32,768 dispatches k_prepare + k_verify (one lane per signature).  40,000 signatures
    (6,667 distinct, tiled): oracle on a window, properties on the whole."""
    base = 6667
    dg, pk, sg = synth.make_batch(base, shared=True, cache_dir=CACHE)
    n = 40000
    pk_t = (pk * 6)[: 32 * n]
    sg_t = bytearray((sg * 6)[: 64 * n])
    victims = [1, 6667, 32768, 39999]
    for v in victims:
        sg_t[64 * v + 33] ^= 0x08
    sg_t = bytes(sg_t)
    ok, bm = _lib.verify_packed(dg, pk_t, sg_t, n, _lib.BATCH, shared=True, want_bitmap=True)
    b = bits(bm, n)
    assert not ok and [i for i in range(n) if not b[i]] == victims
    lo = 32500
    assert b[lo:lo + 800] == oracle_verify_many(oracle, dg, 0, pk_t[32 * lo:], sg_t[64 * lo:], 800, 0)


def test_streamed_rounds_through_the_host_api():
    """BASELINE config 5 (streamed rounds): QC-sized batches (6,667 signatures, host buffers)
    issued back to back from several threads; the context pool overlaps pinned H2D, kernels and
    D2H on separate HIP streams.  Every call's verdict is checked; sustained rate must clear
    the config's 1 M signatures/s by a wide margin."""
    import threading
    import time
    n = 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    bad_sg = bytearray(sg); bad_sg[64 * 100 + 1] ^= 1; bad_sg = bytes(bad_sg)
    lib = _lib.lib()
    bufs = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg, bad_sg)]
    a = [ctypes.addressof(b) for b in bufs]
    duration, threads = 1.5, 6
    counts, errors, lat = [0] * threads, [], [[] for _ in range(threads)]

    def work(k):
        t_end = time.perf_counter() + duration
        i = 0
        while time.perf_counter() < t_end:
            use_bad = (i % 7 == 3)
            t0 = time.perf_counter()
            rc = lib.hs_ed25519_verify_shared(a[0], a[1], a[3] if use_bad else a[2], n, _lib.BATCH, None)
            lat[k].append(time.perf_counter() - t0)
            if rc != (0 if use_bad else 1):
                errors.append((k, i, rc))
            counts[k] += 1
            i += 1

    for _ in range(3):
        assert lib.hs_ed25519_verify_shared(a[0], a[1], a[2], n, _lib.BATCH, None) == 1
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    assert not errors, errors[:5]
    rate = sum(counts) * n / wall
    allat = sorted(x for l in lat for x in l)
    print("streamed rounds: %.1f M signatures/s sustained from host memory, %d calls, p50 %.3f ms, p99 %.3f ms"
          % (rate / 1e6, sum(counts), allat[len(allat) // 2] * 1e3, allat[int(len(allat) * 0.99)] * 1e3))
    assert rate > 1e6


def test_lifecycle_shutdown_reinit_and_keyset_churn():
    """hs_ed25519_shutdown releases the pooled contexts and the base-point table; a later init
    brings everything back.  Creating and destroying key sets repeatedly must not leak HBM."""
    n = 64
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir=CACHE)
    assert _lib.verify_packed(dg, pk, sg, n, _lib.BATCH, shared=True)[0]
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys); ks.close()                 # warm: base table + allocator pools
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        ks = _lib.KeySet(keys)
        packed, _ = ks.indices(keys)
        assert ks.verify_packed(dg, packed, sg, n, _lib.BATCH, True)[0]
        ks.close()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) < 64 << 20      # no 42 MB-per-round leak
    _lib.lib().hs_ed25519_shutdown()
    bufs = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
    rc = _lib.lib().hs_ed25519_verify_shared(*[ctypes.addressof(b) for b in bufs], n, 0, None)
    assert rc == -1                                                   # HS_ED25519_E_NOT_INIT, never "accepted"
    _lib.init(0, 8192)
    assert _lib.verify_packed(dg, pk, sg, n, _lib.BATCH, shared=True)[0]
    ks = _lib.KeySet(keys)
    packed, _ = ks.indices(keys)
    assert ks.verify_packed(dg, packed, sg, n, _lib.STRICT, True)[0]
    ks.close()


@pytest.mark.parametrize("kernel", [_lib.KERNEL_FUSED, _lib.KERNEL_FUSED_PAIRS, _lib.KERNEL_LANE, _lib.KERNEL_AUTO])
def test_several_batches_in_one_launch(oracle, kernel):
    """hs_ed25519_verify_device_batches: 19 independent batches of different sizes, digest forms and
    modes (> 16, so two grids), each with its own bitmap and reject counter, vs the oracle - through the
    one-signature-per-quad kernel, the pairs form (two signatures per quad, 32 per wavefront), the one-lane pair
    over the batch list (k_prepare_batches + k_verify_batches, workspace slices) and the default."""
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    sizes = [1, 16, 17, 63, 64, 65, 100, 333, 1024, 1000, 5, 128, 129, 700, 2, 1024, 31, 257, 64]
    ctx = _lib.DeviceContext(1024)
    ctx.set_kernel(kernel)
    keep, entries, expect = [], [], []
    nr = torch.zeros(len(sizes), dtype=torch.int32, device=dev)
    for k, n in enumerate(sizes):
        shared, mode = bool(k % 2), (_lib.STRICT if k % 3 else _lib.BATCH)
        dg, pk, sg = synth.make_batch(n, shared=shared, first=1000 * k, cache_dir=CACHE)
        dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=40)
        expect.append(oracle_verify_many(oracle, dg, 0 if shared else 32, pk, sg, n, mode))
        d = [t(dg), t(pk), t(sg), torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)]
        keep.append(d)
        entries.append((d[0].data_ptr(), 0 if shared else 32, d[1].data_ptr(), d[2].data_ptr(), n, mode, d[3].data_ptr(),
                        nr[k:].data_ptr()))
    torch.cuda.synchronize(dev)
    ctx.verify_batches(_lib.batch_array(entries))
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
    for k, n in enumerate(sizes):
        got = bits(bytes(keep[k][3].cpu().tolist()), n)
        assert got == expect[k], "batch %d (n = %d)" % (k, n)
        assert int(nr[k].item()) == sum(1 for b in got if not b)
    # argument errors: a batch above the fused kernel's range, an RLC mode
    big = _lib.batch_array([(keep[0][0].data_ptr(), 0, keep[0][1].data_ptr(), keep[0][2].data_ptr(), 40000, 0, keep[0][3].data_ptr(), 0)])
    with pytest.raises(_lib.EngineError):
        ctx.verify_batches(big)
    ctx.close()


@pytest.mark.parametrize("kernel", [_lib.KERNEL_FUSED, _lib.KERNEL_FUSED_PAIRS, _lib.KERNEL_LANE])
def test_edge_vectors_through_batches_per_launch(edge_vectors, torsion_fuzz, kernel):
    """Every labelled edge vector and torsion-fuzz vector (tests/golden) through hs_ed25519_verify_device_batches in
    both modes: five batches per call (the vectors in order, reversed, and rotated so that every vector lands in both
    halves of a pairs-form quad), verdicts = the vectors' labels."""
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    vs = list(edge_vectors) + list(torsion_fuzz)
    ctx = _lib.DeviceContext(1024)
    ctx.set_kernel(kernel)
    orders = [vs, vs[::-1], vs[16:] + vs[:16], vs[7:] + vs[:7], vs[:33]]
    for mode, label in ((_lib.STRICT, "strict"), (_lib.BATCH, "cofactorless")):
        keep, entries = [], []
        for o in orders:
            n = len(o)
            d = [t(b"".join(bytes.fromhex(v["msg"]) for v in o)), t(b"".join(bytes.fromhex(v["pk"]) for v in o)),
                 t(b"".join(bytes.fromhex(v["sig"]) for v in o)), torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)]
            keep.append(d)
            entries.append((d[0].data_ptr(), 32, d[1].data_ptr(), d[2].data_ptr(), n, mode, d[3].data_ptr(), 0))
        torch.cuda.synchronize(dev)
        ctx.verify_batches(_lib.batch_array(entries))
        torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
        for o, d in zip(orders, keep):
            got = bits(bytes(d[3].cpu().tolist()), len(o))
            assert [bool(b) for b in got] == [v[label] for v in o]
    ctx.close()


@pytest.mark.parametrize("kernel", [_lib.KERNEL_AUTO, _lib.KERNEL_CACHED_C])
def test_several_key_cache_batches_in_one_launch(oracle, kernel):
    """hs_ed25519_verify_device_batches_cached: 18 independent batches (two grids) against one key set - different
    sizes, digest forms and modes, permuted / repeated / unknown key indices, corrupted signatures - each with its
    own bitmap and reject counter; verdicts vs the oracle (an unknown index is a rejected signature).  Through the
    default kernel (two lanes per signature + R wavefront) and two-lane chains + the
    batched finish by compression (k_cached_chain2_batches / k_cached_finish_batches, by selector)."""
    import random
    dev = torch.device("cuda", 0)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    nk = 300
    _, kpk, _ = synth.make_batch(nk, shared=True, first=0, cache_dir=CACHE)
    keys = [kpk[32 * i:32 * i + 32] for i in range(nk)]
    ks = _lib.KeySet(keys)
    sizes = [1, 2, 31, 32, 33, 63, 64, 65, 127, 200, 300, 1024, 1000, 5, 129, 700, 257, 64]
    ctx = _lib.DeviceContext(1024)
    ctx.set_kernel(kernel)
    rng = random.Random(11)
    keep, entries, expect = [], [], []
    nr = torch.zeros(len(sizes), dtype=torch.int32, device=dev)
    for k, n in enumerate(sizes):
        shared, mode = bool(k % 2), (_lib.STRICT if k % 3 else _lib.BATCH)
        who = [rng.randrange(nk) for _ in range(n)]
        dg = os.urandom(32) if shared else os.urandom(32 * n)
        sg = b"".join(synth.sign(synth.key_seed(who[i]), dg if shared else dg[32 * i:32 * i + 32])[1] for i in range(n))
        pk = b"".join(keys[w] for w in who)
        dg, _, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=40)      # key-byte flips cannot be expressed here: keys stay as built
        want = bytearray(oracle_verify_many(oracle, dg, 0 if shared else 32, pk, sg, n, mode))
        idx = list(who)
        if n > 4:
            idx[3] = _lib.KeySet.UNKNOWN
            want[3] = 0
        d = [t(dg), torch.tensor([x if x < 2 ** 31 else x - 2 ** 32 for x in idx], dtype=torch.int32, device=dev),
             t(sg), torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)]
        keep.append(d)
        expect.append(bytes(want))
        entries.append((d[0].data_ptr(), 0 if shared else 32, d[1].data_ptr(), d[2].data_ptr(), n, mode, d[3].data_ptr(),
                        nr[k:].data_ptr()))
    torch.cuda.synchronize(dev)
    ctx.verify_batches_cached(ks, _lib.batch_array(entries))
    torch.cuda.ExternalStream(ctx.stream, device=dev).synchronize()
    for k, n in enumerate(sizes):
        got = bits(bytes(keep[k][3].cpu().tolist()), n)
        assert got == expect[k], "batch %d (n = %d)" % (k, n)
        assert int(nr[k].item()) == sum(1 for b in got if not b)
    assert any(all(e) for e in expect) and sum(1 for e in expect if not all(e)) >= 10
    bad_mode = _lib.batch_array([(keep[0][0].data_ptr(), 0, keep[0][1].data_ptr(), keep[0][2].data_ptr(), 1, _lib.BATCH_RLC,
                                  keep[0][3].data_ptr(), 0)])
    with pytest.raises(_lib.EngineError):
        ctx.verify_batches_cached(ks, bad_mode)
    ctx.close()
    ks.close()


def test_mixed_concurrent_callers(oracle):
    """24 host threads hammer the host-pointer API at once with every kind of call - small strict batches (narrow
    pool), QC-sized batch / RLC calls (wide pool), key-cache calls, wire frames, timeout certificates - each with a
    few corrupted signatures.  Every verdict and bitmap must equal the oracle's; nothing may deadlock (pools of 16 + 4
    contexts serve 24 callers)."""
    import base64
    import hashlib
    import struct
    import threading
    from test_wire_and_sign_emu import element
    cases = []
    for k, (n, shared, mode) in enumerate([(1, False, _lib.STRICT), (3, True, _lib.BATCH), (64, False, _lib.STRICT), (700, True, _lib.BATCH_RLC),
                                           (1024, False, _lib.STRICT), (6667, True, _lib.BATCH), (6667, True, _lib.BATCH_RLC), (200, False, _lib.BATCH)]):
        dg, pk, sg = synth.make_batch(n, shared=shared, first=50000 + 7000 * k, cache_dir=CACHE)
        if n > 2:
            dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=24)
        expect = oracle_verify_many(oracle, dg, 0 if shared else 32, pk, sg, n, 1 if mode == _lib.STRICT else 0)
        cases.append(("plain", (dg, pk, sg, n, mode, shared), expect))
    n = 500
    dg, pk, sg = synth.make_batch(n, shared=True, first=90000, cache_dir=CACHE)
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys)
    dgc, pkc, sgc, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=16)
    keysc = [pkc[32 * i:32 * i + 32] for i in range(n)]
    packed, idx = ks.indices(keysc)
    expect = bytearray(oracle_verify_many(oracle, dgc, 0, pkc, sgc, n, 0))
    for i in range(n):
        if idx[i] == _lib.KeySet.UNKNOWN:
            expect[i] = 0
    cases.append(("cached", (dgc, packed, sgc, n), bytes(expect)))
    frames = b"".join(element(base64.b64encode(pkc[32 * i:32 * i + 32]).decode(), sgc[64 * i:64 * i + 64]) for i in range(n))
    cases.append(("wire", (dgc, frames, n), oracle_verify_many(oracle, dgc, 0, pkc, sgc, n, 0)))
    rnd, hq = 31, [i % 5 for i in range(n)]
    tpk, tsg = bytearray(), bytearray()
    for i in range(n):
        p, s = synth.sign(synth.key_seed(95000 + i), hashlib.sha512(struct.pack("<QQ", rnd, hq[i])).digest()[:32])
        tpk += p; tsg += s if i % 97 else s[:40] + bytes([s[40] ^ 8]) + s[41:]
    tdg = b"".join(hashlib.sha512(struct.pack("<QQ", rnd, h)).digest()[:32] for h in hq)
    cases.append(("tc", (rnd, hq, bytes(tpk), bytes(tsg)), oracle_verify_many(oracle, tdg, 32, bytes(tpk), bytes(tsg), n, 1)))
    errors = []

    def worker(t):
        try:
            for r in range(12):
                kind, a, expect = cases[(t + r) % len(cases)]
                if kind == "plain":
                    dg_, pk_, sg_, n_, mode_, shared_ = a
                    ok, bm = _lib.verify_packed(dg_, pk_, sg_, n_, mode_, shared=shared_, want_bitmap=True)
                elif kind == "cached":
                    ok, bm = ks.verify_packed(a[0], a[1], a[2], a[3], _lib.BATCH, shared=True, want_bitmap=True)
                    n_ = a[3]
                elif kind == "wire":
                    ok, bm = _lib.verify_wire(a[1], a[2], digest=a[0], mode=_lib.BATCH)
                    n_ = a[2]
                else:
                    ok, bm = _lib.verify_tc_packed(a[0], a[1], a[2], a[3])
                    n_ = len(a[1])
                if bits(bm, n_) != expect or ok != all(expect):
                    errors.append((t, r, kind))
        except Exception as e:       # noqa: BLE001 - the test reports it
            errors.append((t, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    for x in threads:
        x.start()
    for x in threads:
        x.join(timeout=120)
    assert not any(x.is_alive() for x in threads), "a caller is stuck"
    assert not errors, errors[:5]
    ks.close()


@pytest.mark.parametrize("split", ["0", "1"])
def test_both_eight_lane_forms_on_any_box(oracle, edge_vectors, torsion_fuzz, split):
    """Between 4,097 and 8,192 signatures the library picks whichever eight-lane kernel - one wavefront, or main + helper - it
    measured faster on this GPU at init (hs_ed25519_autotune), so one box exercises only one of them at QC size.  Pin each
    through the environment in a process of its own: all labelled vectors in both modes, and a ~3 % corrupted 6,667-signature
    QC bit-exact against the oracle.  (The per-context selectors 6 / 7 reach both forms too: tests/test_gpu_differential.py.)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, HS_ED25519_OCT_SPLIT=split)
    o = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oct_form_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert o.returncode == 0, o.stderr[-2000:]
    got = json.loads(o.stdout.strip().splitlines()[-1])
    assert got["kernel"] == ("k_verify_fused8h" if split == "1" else "k_verify_fused8")
    vs = list(edge_vectors) + list(torsion_fuzz)
    assert [bool(b) for b in got["strict"]] == [v["strict"] for v in vs]
    assert [bool(b) for b in got["batch"]] == [v["cofactorless"] for v in vs]
    m = 6667
    qdg, qpk, qsg = synth.make_batch(m, shared=True, cache_dir=CACHE)
    qdg, qpk, qsg, bad = synth.corrupt(qdg, qpk, qsg, m, True, rate_byte=8)
    for mode in (_lib.BATCH, _lib.STRICT):
        assert bytes.fromhex(got["qc"][str(mode)]) == oracle_verify_many(oracle, qdg, 0, qpk, qsg, m, mode)
