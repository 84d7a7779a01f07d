"""SOLO launches of the kernels whose memory traffic the profiles quote - one launch at a time, the stream drained
in between, so a dispatch's chip-wide TCC counters hold that dispatch's traffic and nothing else (the
round-2 figures were averages over up to 16 overlapping dispatches).  Run under
    rocprofv3 --pmc FETCH_SIZE -d <dir> -o p -- python tools/pmc_solo.py       (and again with WRITE_SIZE)
then  python tools/rocpd_pmc.py <dir>/p_results.db --by-grid k_  .
Synthetic signatures are made on the GPU (hs_ed25519_sign_batch)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import synth
from hotstuff_amd import _lib

REPS = 3
_lib.init(0, 8192)
dev = torch.device("cuda", 0)
t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)


def signed(n, shared, first):
    seeds = b"".join(synth.key_seed(first + i) for i in range(n))
    pk = _lib.public_keys(seeds, n)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pk[32 * i:32 * i + 32] for i in range(n))
    dg = synth.shared_digest(3) if shared else b"".join(synth.msg_digest(first + i) for i in range(n))
    return dg, pk, _lib.sign_batch(secrets, dg, n, shared=shared)


which = sys.argv[1:] or ["batches", "cached", "lane", "rlc"]
if "bench" in which:
    # bench.py's counter passes (round 5): SOLO launches of every kernel whose roofline the bench line quotes, in ONE process, so that
    # three rocprofv3 passes (FETCH_SIZE, WRITE_SIZE, the SQ set) cover them all.  Dispatches are told apart by name@workgroups.
    #   headline pair                  k_prepare_batches / k_verify_batches, sixteen 1,024-tuple batches per launch
    #   committee_cached list kernel   k_verify_cached_lane2r_batches, four 1,024-signature batches per launch, 16,384 registered keys
    #   QC of N = 10,000 (6,667 sigs)  both eight-lane cold forms (selectors 6 / 7) and the key-cache kernel, 10,000 registered keys
    #   QC of N = 100,000 (66,667)     the lone key-cache form and the two-launch form (selector 5), 100,000 registered keys
    n, K = 1024, 16
    dg, pk, sg = signed(n * K, False, 50_000_000)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(K)]
    nr = torch.zeros(K, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    lst = _lib.batch_array([(d_dg[32 * n * j:].data_ptr(), 32, d_pk[32 * n * j:].data_ptr(), d_sg[64 * n * j:].data_ptr(), n, _lib.STRICT,
                             bm[j].data_ptr(), nr[j:].data_ptr()) for j in range(K)])
    for r in range(REPS + 2):
        ctx.verify_batches(lst)
        st.synchronize()
    assert int(nr.sum().item()) == 0
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n * K)])
    d_ki = torch.arange(n * K, dtype=torch.int32, device=dev)
    lst = _lib.batch_array([(d_dg[32 * n * j:].data_ptr(), 32, d_ki[n * j:].data_ptr(), d_sg[64 * n * j:].data_ptr(), n, _lib.STRICT,
                             bm[j].data_ptr(), nr[j:].data_ptr()) for j in range(4)])
    torch.cuda.synchronize(dev)
    for r in range(REPS + 2):
        ctx.verify_batches_cached(ks, lst)
        st.synchronize()
    assert int(nr.sum().item()) == 0
    ks.close(); ctx.close()
    for nq, Nv, cold in ((6667, 10000, True), (66667, 100000, False)):
        dg, pk, sg = signed(nq, True, 60_000_000)
        d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
        qbm = torch.zeros(8 * ((nq + 63) // 64), dtype=torch.uint8, device=dev)
        qnr = torch.zeros(1, dtype=torch.int32, device=dev)
        ctx = _lib.DeviceContext(nq)
        st = torch.cuda.ExternalStream(ctx.stream, device=dev)
        torch.cuda.synchronize(dev)
        if cold:
            for sel in (_lib.KERNEL_OCT, _lib.KERNEL_OCT_SPLIT):
                ctx.set_kernel(sel)
                for r in range(REPS + 2):
                    ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), nq, _lib.BATCH, qbm.data_ptr(), qnr.data_ptr(), 0)
                    st.synchronize()
            ctx.set_kernel(_lib.KERNEL_AUTO)
        ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(nq)] + [synth.key_seed(300_000_000 + i) for i in range(Nv - nq)])
        d_ki = torch.arange(nq, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        for sel in ((_lib.KERNEL_AUTO,) if cold else (_lib.KERNEL_AUTO, _lib.KERNEL_CACHED_C)):
            ctx.set_kernel(sel)
            for r in range(REPS + 2):
                ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), nq, _lib.BATCH, qbm.data_ptr(), qnr.data_ptr(), 0)
                st.synchronize()
        assert int(qnr.item()) == 0
        ks.close(); ctx.close()
    sys.exit(0)
if "headline" in which:
    # bench.py's live traffic pass: only the headline pair (sixteen 1,024-tuple batches per launch), solo
    n, K = 1024, 16
    dg, pk, sg = signed(n * K, False, 50_000_000)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(K)]
    nr = torch.zeros(K, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    lst = _lib.batch_array([(d_dg[32 * n * j:].data_ptr(), 32, d_pk[32 * n * j:].data_ptr(), d_sg[64 * n * j:].data_ptr(), n, _lib.STRICT,
                             bm[j].data_ptr(), nr[j:].data_ptr()) for j in range(K)])
    for r in range(REPS + 2):
        ctx.verify_batches(lst)
        st.synchronize()
    assert int(nr.sum().item()) == 0
    ctx.close()
if "batches" in which:
    # the headline pair and the two quad forms: K independent 1,024-tuple batches per launch
    n, K = 1024, 16
    dg, pk, sg = signed(n * K, False, 50_000_000)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(K)]
    nr = torch.zeros(K, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    for k in (16, 4, 2):
        lst = _lib.batch_array([(d_dg[32 * n * j:].data_ptr(), 32, d_pk[32 * n * j:].data_ptr(), d_sg[64 * n * j:].data_ptr(), n, _lib.STRICT,
                                 bm[j].data_ptr(), nr[j:].data_ptr()) for j in range(k)])
        for r in range(REPS + 1):
            ctx.verify_batches(lst)
            st.synchronize()
        assert int(nr.sum().item()) == 0
        print("batches per launch %d: %s" % (k, _lib.lib().hs_ed25519_batches_kernel_for(ctx._h, k).decode()), flush=True)
    # the key-cache list form: four batches per launch
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n * 4)])
    d_ki = torch.arange(n * 4, dtype=torch.int32, device=dev)
    lst = _lib.batch_array([(d_dg[32 * n * j:].data_ptr(), 32, d_ki[n * j:].data_ptr(), d_sg[64 * n * j:].data_ptr(), n, _lib.STRICT,
                             bm[j].data_ptr(), nr[j:].data_ptr()) for j in range(4)])
    torch.cuda.synchronize(dev)
    for r in range(REPS + 1):
        ctx.verify_batches_cached(ks, lst)
        st.synchronize()
    assert int(nr.sum().item()) == 0
    ks.close(); ctx.close()

if "cached" in which or "lane" in which or "rlc" in which:
    nl, Nv = 66667, 100000
    dg, pk, sg = signed(nl, True, 60_000_000)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = torch.zeros(8 * ((nl + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(nl)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    if "cached" in which:
        ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(nl)] + [synth.key_seed(300_000_000 + i) for i in range(Nv - nl)])
        # a QC is a scattered 2f+1 subset: here the first nl entries, which is what bench.py's large_committee uses
        d_ki = torch.arange(nl, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        for r in range(REPS + 1):
            ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), nl, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), 0)
            st.synchronize()
        assert int(nr.item()) == 0
        print("cached 66,667: %s" % _lib.lib().hs_ed25519_cached_kernel_for(nl).decode(), flush=True)
        ks.close()
    if "lane" in which:
        for r in range(REPS + 1):
            ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), nl, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), 0)
            st.synchronize()
        assert int(nr.item()) == 0
        print("cold 66,667 per signature: %s" % _lib.lib().hs_ed25519_dominant_kernel_for(nl).decode(), flush=True)
    if "rlc" in which:
        ctx.set_rlc_min(1)
        for r in range(REPS + 1):
            ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), nl, _lib.BATCH_RLC, bm.data_ptr(), nr.data_ptr(), 0)
            st.synchronize()
        assert int(nr.item()) == 0 and ctx.rlc_last()[0] == 1
        print("cold 66,667 RLC stages", flush=True)
    ctx.close()
