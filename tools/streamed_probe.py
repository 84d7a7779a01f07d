import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tools"))
import torch, synth, bench
from hotstuff_amd import _lib
_lib.init(0, 1024)
def show(tag):
    r = bench.streamed_rounds(_lib, synth, seconds=0.4)
    print(tag, {k: (round(v['signatures_per_s']/1e6,1), v['submit_to_verdict_p50_ms']) for k, v in r.items() if isinstance(v, dict)}, flush=True)
which = sys.argv[1]
show("fresh")
if which == "e2e":
    n = 1024
    batches = [synth.make_batch(n, shared=False, first=j * n, cache_dir="/tmp/hs_synth") for j in range(16)]
    bench.end_to_end(_lib, batches, n, 32, _lib.STRICT, seconds=0.3)
    show("after end_to_end")
elif which == "qc":
    dev = torch.device("cuda", 0)
    bench.qc_latency(_lib, torch, synth, dev, calls=50)
    show("after qc_latency")
    bench.rlc_ab(_lib, torch, synth, dev, calls=10)
    show("after rlc_ab")
if which == "streams":
    ctxs = [_lib.DeviceContext(1024) for _ in range(int(sys.argv[2]))]
    show("with %s idle contexts alive" % sys.argv[2])
    for c in ctxs: c.close()
    show("after closing them")
if which == "hostqc":
    import ctypes
    n = 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
    for i in range(30):
        assert _lib.lib().hs_ed25519_verify_shared(ctypes.addressof(b[0]), ctypes.addressof(b[1]), ctypes.addressof(b[2]), n, _lib.BATCH, None) == 1
    show("after 30 host-API QC calls (one wide pool context)")
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(n)] + [synth.key_seed(20_000_000 + i) for i in range(3333)])
    show("with a 10,000-key set alive")
    ks.close()
    show("after closing it")
if which == "cause":
    import ctypes
    def host_calls(n, reps=30):
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
        b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
        for i in range(reps):
            assert _lib.lib().hs_ed25519_verify_shared(ctypes.addressof(b[0]), ctypes.addressof(b[1]), ctypes.addressof(b[2]), n, _lib.BATCH, None) == 1
    host_calls(1024)
    show("after host calls of 1,024 (narrow pool)")
    host_calls(6667)
    show("after host calls of 6,667 (wide pool, zero-copy)")
    _lib.shutdown(); _lib.init(0, 1024)
    show("after shutdown + init (pools destroyed)")
    host_calls(6667, reps=1)
    show("after ONE host call of 6,667")
if which == "bisect":
    import ctypes
    def host_calls(n, reps, api="host"):
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
        if api == "host":
            b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
            for i in range(reps):
                assert _lib.lib().hs_ed25519_verify_shared(ctypes.addressof(b[0]), ctypes.addressof(b[1]), ctypes.addressof(b[2]), n, _lib.BATCH, None) == 1
        else:       # device API: resident inputs, one context, synchronised after every call
            dev = torch.device("cuda", 0)
            t = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (dg, pk, sg)]
            out = torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev)
            ctx = _lib.DeviceContext(n)
            for i in range(reps):
                ctx.verify(t[0].data_ptr(), 0, t[1].data_ptr(), t[2].data_ptr(), n, _lib.BATCH, out.data_ptr(), 0, 0)
                torch.cuda.synchronize()
            ctx.close()
    def fresh():
        _lib.shutdown(); _lib.init(0, 1024)
    for reps in (3, 10, 30, 100):
        fresh(); host_calls(6667, reps); show("host API x%d" % reps)
    fresh(); host_calls(6667, 30); time.sleep(1.0); show("host API x30, then 1 s idle")
    fresh(); host_calls(6667, 30, api="device"); show("device API x30 (context closed after)")
    fresh(); host_calls(9000, 30); show("host API x30 of 9,000 (four-lane kernel, copy engines)")
    fresh(); show("fresh again")
if which == "alive":
    # is it a LIVE extra context (stream + buffers) that slows the ring built after it, or something the host-pointer path does?
    import ctypes
    n = 6667
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    dev = torch.device("cuda", 0)
    t = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (dg, pk, sg)]
    out = torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev)
    ctx = _lib.DeviceContext(n)
    for i in range(3):
        ctx.verify(t[0].data_ptr(), 0, t[1].data_ptr(), t[2].data_ptr(), n, _lib.BATCH, out.data_ptr(), 0, 0)
        torch.cuda.synchronize()
    show("a device-API context of 6,667 ALIVE (3 calls made)")
    ctx.close()
    show("after closing it")
    pb = [_lib.PinnedBuffer(x) for x in (dg, pk, sg)]
    for i in range(3):
        assert _lib.lib().hs_ed25519_verify_shared(pb[0].addr, pb[1].addr, pb[2].addr, n, _lib.BATCH, None) == 1
    show("after 3 host-pointer calls from CALLER-PINNED buffers (wide pool context alive, no staging copy)")
    _lib.shutdown(); _lib.init(0, 1024)
    show("after shutdown + init")
    b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
    for i in range(3):
        assert _lib.lib().hs_ed25519_verify_shared(ctypes.addressof(b[0]), ctypes.addressof(b[1]), ctypes.addressof(b[2]), n, _lib.BATCH, None) == 1
    show("after 3 host-pointer calls from ordinary buffers (second wide context of the process)")
if which == "regions":
    # does the key-cache kernel's rate depend on WHICH memory a key set landed in?  Two sets alive at once, then a third after a small
    # allocation took a piece of the first one's freed region.  Device-resident, four contexts in flight, 6,667-signature calls.
    n = 6667
    dev = torch.device("cuda", 0)
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    keys = [pk[32 * i:32 * i + 32] for i in range(n)] + [synth.key_seed(20_000_000 + i) for i in range(3333)]
    t = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (dg, sg)]
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)

    def rate(ks, tag, nctx=4, reps=60):
        ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
        bms = [torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev) for _ in range(nctx)]
        sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
        torch.cuda.synchronize(dev)
        def call(j):
            ctxs[j].verify_cached(ks, t[0].data_ptr(), 0, d_ki.data_ptr(), t[1].data_ptr(), n, _lib.BATCH, bms[j].data_ptr(), 0, 0)
        for j in range(nctx):
            call(j)
        for s in sts:
            s.synchronize()
        lat = []
        for i in range(30):
            t0 = time.perf_counter(); call(0); sts[0].synchronize(); lat.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for i in range(reps):
            for j in range(nctx):
                call(j)
        for s in sts:
            s.synchronize()
        wall = time.perf_counter() - t0
        for c in ctxs:
            c.close()
        lat.sort()
        print("%-70s p50 %.4f ms, %.1f M sig/s with four in flight" % (tag, lat[15] * 1e3, reps * nctx * n / wall / 1e6), flush=True)

    ks_a = _lib.KeySet(keys); rate(ks_a, "set A (first 5.3 GB allocation of its kind)")
    ks_b = _lib.KeySet(keys); rate(ks_b, "set B, A still alive"); rate(ks_a, "set A again")
    ks_a.close()
    small = _lib.DeviceContext(6667)
    ks_c = _lib.KeySet(keys); rate(ks_c, "set C: A freed, a small context allocated, then C"); rate(ks_b, "set B again")
    ks_c.close(); small.close()
    ks_d = _lib.KeySet(keys); rate(ks_d, "set D: C and the small context freed, then D")
    ks_d.close(); ks_b.close()
    show("rings after all that")
if which == "coupling":
    # NEXT PROBE (not run yet: the round's GPU budget was spent): what about an idle context slows other queues - its stream, or
    # its buffers?  Key-cache kernel, device-resident, four contexts in flight (as in `regions`), one key set, with in turn:
    # nothing extra / an idle torch stream / 24 MB of device buffers / pinned host memory / an idle engine context / that context
    # after one call.  Run under GPU_MAX_HW_QUEUES=8, 16 (default), 24 to see whether the hardware-queue count matters.
    n = 6667
    dev = torch.device("cuda", 0)
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    keys = [pk[32 * i:32 * i + 32] for i in range(n)] + [synth.key_seed(20_000_000 + i) for i in range(3333)]
    t = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (dg, sg)]
    tp = torch.frombuffer(bytearray(pk), dtype=torch.uint8).to(dev)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    ks = _lib.KeySet(keys)

    def rate(tag, nctx=4, reps=60):
        ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
        bms = [torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev) for _ in range(nctx)]
        sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
        torch.cuda.synchronize(dev)
        def call(j):
            ctxs[j].verify_cached(ks, t[0].data_ptr(), 0, d_ki.data_ptr(), t[1].data_ptr(), n, _lib.BATCH, bms[j].data_ptr(), 0, 0)
        for j in range(nctx):
            call(j)
        for s in sts:
            s.synchronize()
        lat = []
        for i in range(60):                       # the metric's own number: p50 of a LONE call
            t0 = time.perf_counter(); call(0); sts[0].synchronize(); lat.append(time.perf_counter() - t0)
        lat.sort()
        t0 = time.perf_counter()
        for i in range(reps):
            for j in range(nctx):
                call(j)
        for s in sts:
            s.synchronize()
        wall = time.perf_counter() - t0
        for c in ctxs:
            c.close()
        print("%-64s lone p50 %.4f ms (p10 %.4f)  %.1f M sig/s with four in flight" % (tag, lat[30] * 1e3, lat[6] * 1e3, reps * nctx * n / wall / 1e6), flush=True)

    for rnd in range(2):
        rate("nothing extra")
        s_idle = torch.cuda.Stream(device=dev); rate("+ an idle torch stream"); del s_idle
        buf = torch.empty(24 << 20, dtype=torch.uint8, device=dev); rate("+ 24 MB of device buffers, no stream"); del buf
        pin = torch.empty(1 << 20, dtype=torch.uint8).pin_memory(); rate("+ 1 MB of pinned host memory"); del pin
        c = _lib.DeviceContext(n); rate("+ an idle engine context of 6,667 signatures (never used)")
        bm = torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev)
        c.verify(t[0].data_ptr(), 0, tp.data_ptr(), t[1].data_ptr(), n, _lib.BATCH, bm.data_ptr(), 0, 0); torch.cuda.synchronize(dev)
        rate("+ that context after one cold call"); c.close()
        c = _lib.DeviceContext(n, partition=(0, 8)); rate("+ an idle CU-MASKED engine context (never used)"); c.close()
        c = _lib.DeviceContext(64); rate("+ an idle engine context of 64 signatures (stream, tiny buffers)"); c.close()
        rate("nothing extra again")
    ks.close()

if which == "coupling2":
    # Is the "idle-context coupling" streams sharing hardware queues?  HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES
    # (16 here) hardware queues; a stream is bound when it is first used.  N extra contexts, each USED ONCE (so that its stream
    # holds a queue) and then idle, beside four contexts that keep key-cache QCs in flight: the rate against N.
    n = 6667
    dev = torch.device("cuda", 0)
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    keys = [pk[32 * i:32 * i + 32] for i in range(n)] + [synth.key_seed(20_000_000 + i) for i in range(3333)]
    t = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (dg, sg)]
    tp = torch.frombuffer(bytearray(pk), dtype=torch.uint8).to(dev)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    ks = _lib.KeySet(keys)
    sdg, spk, ssg = synth.make_batch(64, shared=True, cache_dir="/tmp/hs_synth")
    st_ = [torch.frombuffer(bytearray(x), dtype=torch.uint8).to(dev) for x in (sdg, spk, ssg)]
    sbm = torch.zeros(8, dtype=torch.uint8, device=dev)

    def rate(tag, nctx=4, reps=60):
        ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
        bms = [torch.zeros((n + 63) // 64 * 8, dtype=torch.uint8, device=dev) for _ in range(nctx)]
        sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
        torch.cuda.synchronize(dev)
        def call(j):
            ctxs[j].verify_cached(ks, t[0].data_ptr(), 0, d_ki.data_ptr(), t[1].data_ptr(), n, _lib.BATCH, bms[j].data_ptr(), 0, 0)
        for j in range(nctx):
            call(j)
        for s in sts:
            s.synchronize()
        lat = []
        for i in range(60):
            t0 = time.perf_counter(); call(0); sts[0].synchronize(); lat.append(time.perf_counter() - t0)
        lat.sort()
        t0 = time.perf_counter()
        for i in range(reps):
            for j in range(nctx):
                call(j)
        for s in sts:
            s.synchronize()
        wall = time.perf_counter() - t0
        for c in ctxs:
            c.close()
        print("%-64s lone p50 %.4f ms  %.1f M sig/s with four in flight" % (tag, lat[30] * 1e3, reps * nctx * n / wall / 1e6), flush=True)

    for N in (0, 4, 8, 10, 11, 12, 13, 14, 16, 20, 28, 0):
        extra = [_lib.DeviceContext(64) for _ in range(N)]
        for c in extra:      # used once: the stream now holds its hardware queue
            c.verify(st_[0].data_ptr(), 0, st_[1].data_ptr(), st_[2].data_ptr(), 64, _lib.BATCH, sbm.data_ptr(), 0, 0)
        torch.cuda.synchronize(dev)
        rate("%d extra contexts, each used once, idle (GPU_MAX_HW_QUEUES=%s)" % (N, os.environ.get("GPU_MAX_HW_QUEUES")))
        for c in extra:
            c.close()
    ks.close()
