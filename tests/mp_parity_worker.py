"""One node PROCESS of tests/test_gpu_multiprocess.py: its own engine on device 0; a mixed, ~3 % corrupted workload
(small strict batches by ticket bursts, a QC-sized batch, the same through a registered key set, caller-pinned buffers),
every bitmap compared with the C oracle (test infrastructure).  Prints "ready", waits for a line on stdin, runs
`rounds` rounds, prints one JSON object: {"checks": k, "mismatches": [...]}."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    index, rounds = int(sys.argv[1]), int(sys.argv[2])
    import synth
    from conftest import bits, oracle_verify_many
    from hotstuff_amd import _lib
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "libhs_oracle.so"))
    cache = os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth")
    _lib.init(0, 8192)
    work = []
    # (name, n, shared, mode): small strict (TC shape), QC-sized batch semantics, a ragged one
    for name, n, shared, mode in (("small", 1024, False, _lib.STRICT), ("qc", 6667, True, _lib.BATCH), ("ragged", 333, False, _lib.STRICT),
                                  ("tiny", 5, True, _lib.BATCH)):
        dg, pk, sg = synth.make_batch(n, shared=shared, first=index * 100_000, cache_dir=cache)
        dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=8)
        expect = oracle_verify_many(orc, dg, 0 if shared else 32, pk, sg, n, mode)
        work.append((name, n, shared, mode, dg, pk, sg, expect))
    qc = work[1]
    keys = [qc[5][32 * i:32 * i + 32] for i in range(qc[1])]
    ks = _lib.KeySet(keys)
    packed, _ = ks.indices(keys)
    pin = [_lib.PinnedBuffer(x) for x in (work[0][4], work[0][5], work[0][6])]
    print("ready", flush=True)
    sys.stdin.readline()
    checks, mism = 0, []
    for r in range(rounds):
        for name, n, shared, mode, dg, pk, sg, expect in work:
            ok, bm = _lib.verify_packed(dg, pk, sg, n, mode, shared=shared, want_bitmap=True)
            checks += 1
            if bits(bm, n) != expect or ok != all(expect):
                mism.append("%s round %d" % (name, r))
        # a burst of four small batches sharing one launch
        name, n, shared, mode, dg, pk, sg, expect = work[0]
        tickets = [_lib.Ticket(dg, pk, sg, n, mode=mode, shared=shared, more=(j < 3)) for j in range(4)]
        for t in tickets:
            ok, bm = t.wait()
            checks += 1
            if bits(bm, n) != expect:
                mism.append("burst round %d" % r)
        # the QC through the key cache
        name, n, shared, mode, dg, pk, sg, expect = qc
        ok, bm = ks.verify_packed(dg, packed, sg, n, mode, shared=True, want_bitmap=True)
        checks += 1
        if bits(bm, n) != expect:
            mism.append("key cache round %d" % r)
        # caller-owned pinned buffers (read in place)
        name, n, shared, mode, dg, pk, sg, expect = work[0]
        bmp = (ctypes.c_ubyte * ((n + 7) // 8))()
        rc = _lib.lib().hs_ed25519_verify_multi(pin[0].addr, pin[1].addr, pin[2].addr, n, mode, ctypes.cast(bmp, ctypes.c_void_p))
        checks += 1
        if rc not in (0, 1) or bits(bytes(bmp), n) != expect:
            mism.append("pinned round %d (rc %d)" % (r, rc))
    ks.close()
    print(json.dumps({"checks": checks, "mismatches": mism[:10]}), flush=True)


if __name__ == "__main__":
    main()
