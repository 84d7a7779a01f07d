"""Probe used for the round-2 profiles: `python tools/rlc_probe.py qc` (device-resident QC at N = 10,000, cold and
cached) or `python tools/rlc_probe.py rlc <n> [cached]` (six HS_ED25519_BATCH_RLC calls, then six plain BATCH calls, on
one context with the RLC threshold lowered to 1; prints the last call's wall time) - run under rocprofv3 to get the
per-stage kernel times in profiles/r2_rlc_*_kernel_stats.csv."""
import ctypes, os, sys, time, struct
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import torch, synth
from hotstuff_amd import _lib
_lib.init(0, 8192)
dev = torch.device('cuda', 0)
t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
which = sys.argv[1]
if which == 'qc':
    n = 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir='/tmp/hs_synth')
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(st):
        for i in range(5):
            nr.zero_(); st.synchronize()
            ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
            print('cold bad', int(nr.item()), int((bm.view(torch.int64)[:n//64] != -1).sum()))
    keys = [pk[32*i:32*i+32] for i in range(n)] + [synth.key_seed(20_000_000 + i) for i in range(10000 - n)]
    ks = _lib.KeySet(keys)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    with torch.cuda.stream(st):
        for i in range(5):
            nr.zero_(); st.synchronize()
            ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
            print('cached bad', int(nr.item()), int((bm.view(torch.int64)[:n//64] != -1).sum()))
elif which == 'rlc':
    n = int(sys.argv[2]); cached = len(sys.argv) > 3
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir='/tmp/hs_synth')
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n); ctx.set_rlc_min(1)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    ks = d_ki = None
    if cached:
        ks = _lib.KeySet([pk[32*i:32*i+32] for i in range(n)])
        d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize(dev)
    for mode in (_lib.BATCH_RLC, _lib.BATCH):
        for i in range(6):
            st.synchronize(); t0 = time.perf_counter()
            if cached:
                ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, mode, bm.data_ptr(), nr.data_ptr(), 0)
            else:
                ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, mode, bm.data_ptr(), nr.data_ptr(), 0)
            st.synchronize(); dt = time.perf_counter() - t0
        print('mode', mode, 'n', n, 'cached', cached, 'ms %.3f' % (dt*1e3), 'rlc_last', ctx.rlc_last(), 'bad', int(nr.item()), flush=True)
