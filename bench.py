#!/usr/bin/env python3
"""bench.py - Ed25519 verifies/sec of the MI355X engine on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

Workload = BASELINE.json configs[1]: batches of 1,024 distinct (digest, pk, sig) tuples, one digest per
tuple, strict semantics (the TC::verify / Signature::verify predicate, consensus/src/messages.rs:307-313),
inputs resident in HBM.  A "step" is one pass of the hot path over a synthetic input set of
`--batches-per-step` (default 1,024) such batches = 1,048,576 verifications, EVERY TUPLE OF THE STEP DISTINCT
(`--distinct-batches`; rounds 1 - 4 re-verified sixteen batches - the rate is the same, profiles/r5_n_distinct_batches.txt):
64 hs_ed25519_verify_device_batches calls of sixteen batches each (k_prepare_batches + k_verify_batches) issued
round-robin over 16 stream contexts, each context restricted to its own slice of the CUs - the way a node under load
issues independent certificates.  With the driver's `--steps 20 --warmup 5` the timed region is 1,280 launch pairs /
~0.2 s, steady state whatever `steps` is.  The timed region is bracketed by a barrier and a device synchronize on both
sides; the maximum over ranks is taken.

`value` is the HBM-resident rate the contract asks for.  `value_host_to_host` (same JSON line, = config.end_to_end_verifies_per_s_best)
is SURVEY.md 8d's metric proper - host buffers in, host verdict out through the host-pointer C ABI (tickets from 1 - 16 native
host threads; pinned staging, PCIe both ways included).  `config` opens with the keys that ARE the metric, in the order the
driver's record keeps them (CONFIG_FIRST / finalize_line): QC and TC p50 cold / key cache, their CPU twins, the host-to-host rates.

With N > 1 every rank verifies its own independent batches (weak scaling; the units are independent,
SURVEY.md 8e, so the headline has no data-path collective).  The workload with a real exchange step -
BASELINE configs[3], a 100,000-validator QC cut into one shard per GPU - is run after the headline two
ways: `sharded_committee` (one process per GPU, torch.distributed RCCL all-gather of the bitmaps) and
`sharded_c_abi` (ONE process - rank 0 - driving every GPU through hs_ed25519_verify_shared_sharded,
which is what the reference's single-process node would call; host and RCCL exchange).

Prints ONE JSON line on rank 0 (contract in the task statement) with:
  roofline      algorithmic bytes (128 B per verification) of the dominant kernel
                divided by its average duration, measured live with HIP events attached to sampled
                dispatches on the launch streams inside the timed region, against the 8 TB/s HBM
                peak.  This path is integer-VALU bound, so the fraction is tiny by construction;
                `int_roofline` prices the same kernel against the measured v_mad_u64_u32 issue rate.
  cpu_baseline  the CPU restatement of the reference's dalek path (oracle/, kind "port"), timed on
                the host cores this process may use, on a bounded sample of the same workload.
  extras        `end_to_end`, `committee_cached`, `qc_verify_latency` (p50 at N = 10,000, with the
                CPU batch verification timed beside it), `tc_verify_latency` (TC::verify at N = 10,000: per-vote digests built
                on the device, strict; the reference's loop on the CPU beside it), `rlc_ab` (per-signature vs batch-RLC/MSM path
                at 6,667 and 66,667 signatures), `streamed_rounds` (configs[4]),
                `small_batch_crossover` (GPU vs one CPU core at n = 1 ... 32).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

METRIC = "Ed25519 verifies/sec (whole node) + QC-verify p50 latency at N validators"     # BASELINE.json's metric, verbatim: `value` is
# its first half; the second half is config.qc_verify_p50_ms_* at config.qc_validators (the driver's record keeps `config`)
ALGO_BYTES_MULTI = 128      # 32 B digest + 32 B key + 64 B signature per verification
ALGO_BYTES_SHARED = 96
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
# Informative integer roofline: v_mad_u64_u32 lane-operations per strict
# verification, counted statically from the kernels' field-operation counts
#   (field mul = 100 mads, square = 55; quad kernels count all four lanes of the quad),
# against the chip-wide issue rate measured by tools/ubench_valu.hip (profiles/r1_ubench_valu.txt).
# Round 3: the cold kernels verify the half-size form (halfscalar.h): [v s]B + [u](-A) + [v](-R) == O with |u|, v ~ 2^128 -
# NWIN windows of four doublings (33 for 77 % of wavefronts, 34 for most of the rest; 34 is priced), two table additions per
# window, two per-signature tables, sixteen additions of [w]B.  Round 2 priced 252 doublings + 64 + 16 additions: 232,690 (lane).
NWIN = 34
NWIN3 = 45
MAD_PER_VERIFY = {
    # fused quad, per wavefront of 16 signatures: (NWIN - 1) x 4 doublings x (55 + 100) + (2 NWIN + 16) additions x 200
    # + two tables 2 x 7 x 300 + decompress (255 sq + 21 mul), all x 64 lanes / 16 signatures
    "k_verify_fused": ((NWIN - 1) * 4 * 155 + (2 * NWIN + 16) * 200 + 14 * 300 + 255 * 55 + 21 * 100) * 64 // 16,
    # eight lanes per signature (single batches up to 8,192 signatures): every quad of the octet runs ONE table's chain -
    # (NWIN - 1) x 4 doublings, NWIN + 8 additions, 7 table additions, the join - and all eight lanes decompress
    "k_verify_fused8h": ((NWIN - 1) * 4 * 155 + (NWIN + 8) * 200 + 7 * 300 + 300 + 255 * 55 + 21 * 100) * 64 // 8,      # the same arithmetic on the main wavefront; the helper adds none of it (hash + reduction are not multiply-adds of the field)
    "k_verify_fused8": ((NWIN - 1) * 4 * 155 + (NWIN + 8) * 200 + 7 * 300 + 300 + 255 * 55 + 21 * 100) * 64 // 8,
    # the same with 3-bit windows (NWIN3 windows of three doublings, four-entry tables): the kernels that put several
    # batches into one launch, two wavefronts per SIMD
    "k_verify_fused3": ((NWIN3 - 1) * 3 * 155 + (2 * NWIN3 + 16) * 200 + 6 * 300 + 255 * 55 + 21 * 100) * 64 // 16,
    # pairs form: a wavefront carries 32 signatures; one decompression pass of all 64 lanes serves both window loops
    "k_verify_fused_pairs": (2 * ((NWIN3 - 1) * 3 * 155 + (2 * NWIN3 + 16) * 200 + 6 * 300) + 255 * 55 + 21 * 100) * 64 // 32,
    # lane: doublings 4 sq + 3 mul (+ 1 with T, every fourth); per window 15 mul in two additions; tables 2 x 65 mul;
    # [w]B 16 x 7 mul; k_prepare adds 2 x (255 sq + 21 mul)
    "k_verify": (NWIN - 1) * (4 * 520 + 100) + NWIN * 1500 + 13000 + 11200,
    "k_verify_batches": (NWIN - 1) * (4 * 520 + 100) + NWIN * 1500 + 13000 + 11200 + 2 * (255 * 55 + 21 * 100),
}
MAD_PEAK_PER_S = 2.96e13    # 48.2 lane-ops/clk/CU x 256 CUs x 2.4e9 (measured, 8 waves/SIMD)
# Total VALU issue (VERDICT r3 "Next round" #4: the fraction that actually binds).  SQ counters of the headline pair
# (profiles/r4_p_pmc_sq_headline.json, round 4's kernels: SQ_INSTS_VALU of k_prepare_batches 1,262.9 + k_verify_batches 4,050.4 per
# verification - round 3: 5,349.7 -, 52.8 % of them v_mad_u64_u32: 179,390 lane-operations = 2,803 wave-instructions)
# against what 1,024 SIMDs can issue of THAT mix at the microbenchmarked best rates (profiles/r1_ubench_valu.txt, 8 wavefronts
# per SIMD: 5.32 cycles per wave64 v_mad_u64_u32, 2.5 per plain 32-bit instruction) at 2.4 GHz.
VALU_WAVE_INSTS_PER_VERIFY = {"k_verify_batches": 5313.3}
MAD_SHARE_OF_VALU = 0.528
CYCLES_PER_WAVE_INST_MIX = MAD_SHARE_OF_VALU * 5.32 + (1 - MAD_SHARE_OF_VALU) * 2.5
SIMDS, SCLK_HZ = 1024, 2.4e9


def host_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def load_oracle():
    so = os.path.join(ROOT, "oracle", "libhs_oracle.so")
    native = "/tmp/hs_oracle_native.so"
    try:    # a -march=native build for a fair host number; falls back to the shipped build
        if not os.path.exists(native):
            subprocess.run(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-o", native,
                            os.path.join(ROOT, "oracle", "ed25519_oracle.c"), "-lpthread"],
                           check=True, capture_output=True, timeout=120)
        so = native
    except Exception:
        pass
    return ctypes.CDLL(so)


def cpu_baseline(dg, pk, sg, n, stride, mode, budget_s=14.0):
    """Time the oracle (the CPU port of the reference's per-signature verify loop) on one core and on
    every core this process may use.  Each thread gets >= 1,000 verifications per pass so that thread
    start-up (the oracle spawns its workers per call) is noise."""
    orc = load_oracle()
    cores = host_cores()
    S = ctypes.c_size_t
    tile = max(1, (1000 * cores + n - 1) // n)
    big_n = n * tile
    big_dg = dg * tile if stride else dg
    big_pk, big_sg = pk * tile, sg * tile

    def run(threads, nn, a_dg, a_pk, a_sg, budget):
        out = ctypes.create_string_buffer(nn)
        t0 = time.perf_counter()
        reps = 0
        while True:
            orc.hs_oracle_verify_many(a_dg, S(32), S(stride), a_pk, a_sg, S(nn), mode, out, threads)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget:
                break
        assert all(out.raw), "oracle rejected a valid synthetic signature"
        return reps * nn / dt, reps

    one, r1 = run(1, n, dg, pk, sg, budget_s * 0.3)
    allc, rn = run(cores, big_n, big_dg, big_pk, big_sg, budget_s * 0.7)
    try:        # the system OpenSSL's single verification on one core, a bracket independent of oracle/
        import synth
        openssl = round(synth.openssl_verify_rate(dg, pk, sg, n, stride, 0.5), 1)
    except Exception as e:      # noqa: BLE001 - a bracket, not a result
        openssl = "unavailable: %s" % e
    try:        # a second bracket: libsodium's single verification (ref10 lineage, as dalek's serial backend)
        sodium = round(synth.sodium_verify_rate(dg, pk, sg, n, stride, 0.5), 1)
    except Exception as e:      # noqa: BLE001
        sodium = "unavailable: %s" % e
    return {
        "value": round(allc, 1), "unit": "verifies/s", "cores": cores, "kind": "port",
        "one_core": round(one, 1), "parallel_efficiency": round(allc / (cores * one), 3),
        "openssl_evp_digestverify_one_core": openssl,
        "libsodium_verify_detached_one_core": sodium,
        "os_cpu_count": os.cpu_count(),
        "sample": "oracle port of verify_strict: %d tuples x%d = %d per pass, %d passes on %d threads; one core: %d passes" % (n, tile, big_n, rn, cores, r1),
        "sample_detail": "oracle/ed25519_oracle.c (C restatement of ed25519-dalek 1.0.1, 51-bit limbs, -O3 -march=native)",
        "threads_are": "sched_getaffinity capped by the cgroup quota",
    }


def cpu_qc_latency(dg, pk, sg, n, calls=5):
    """The CPU twin of `qc_verify_latency`: dalek's verify_batch (RLC + Pippenger, oracle restatement) on
    one core, and split into one sub-batch per core (hs_oracle_verify_batch_mt)."""
    orc = load_oracle()
    S = ctypes.c_size_t
    cores = host_cores()
    one, many = [], []
    for i in range(calls):
        t0 = time.perf_counter()
        ok = orc.hs_oracle_verify_batch(dg, S(32), S(0), pk, sg, S(n), ctypes.c_uint64(i))
        one.append(time.perf_counter() - t0)
        assert ok == 1
    for i in range(calls * 4):
        t0 = time.perf_counter()
        ok = orc.hs_oracle_verify_batch_mt(dg, S(32), S(0), pk, sg, S(n), ctypes.c_uint64(i), cores)
        many.append(time.perf_counter() - t0)
        assert ok == 1
    med = lambda v: sorted(v)[len(v) // 2]
    return {"kind": "port", "algorithm": "dalek verify_batch restated: 128-bit RLC + Pippenger (w = 8)",
            "p50_ms_one_core": round(med(one) * 1e3, 3), "p50_ms_all_cores": round(med(many) * 1e3, 3), "cores": cores}


def cpu_tc_latency(rnd, hq, pk, sg, n, calls=3):
    """The CPU twin of `tc_verify_latency`: the reference's loop itself (consensus/src/messages.rs:307-313) - per vote one SHA-512 of
    (round || high_qc_round), then Signature::verify = verify_strict - on one core (oracle port), and the same loop cut over every
    core this process may use."""
    import struct
    orc = load_oracle()
    S = ctypes.c_size_t
    cores = host_cores()
    out = ctypes.create_string_buffer(n)
    hq_b = struct.pack("<%dQ" % n, *hq)
    one, many = [], []
    for th, acc, reps in ((1, one, calls), (cores, many, calls * 3)):
        for _ in range(reps):
            t0 = time.perf_counter()
            orc.hs_oracle_verify_tc(ctypes.c_uint64(rnd), hq_b, pk, sg, S(n), out, th)
            acc.append(time.perf_counter() - t0)
            assert all(out.raw), "oracle rejected a valid TC vote"
    med = lambda v: sorted(v)[len(v) // 2]
    return {"kind": "port", "algorithm": "the reference's loop: n x (SHA-512(round || high_qc_round), verify_strict)",
            "p50_ms_one_core": round(med(one) * 1e3, 3), "p50_ms_all_cores": round(med(many) * 1e3, 3), "cores": cores}


def tc_latency(_lib, torch, dev, validators=10000, calls=200):
    """TC::verify (consensus/src/messages.rs:290-315) at N validators: n = floor(2N/3) + 1 votes, every vote signed over ITS OWN
    digest SHA-512(round || high_qc_round_i)[..32], strict semantics.  The per-vote digests are built on the device
    (hs_ed25519_verify_tc[_cached]: the host passes the round and n high_qc_rounds).  p50 over `calls` calls, device-resident and
    through the host-pointer C ABI, cold and with the committee registered; the CPU twin beside it."""
    import hashlib
    import struct
    import synth
    n = 2 * validators // 3 + 1
    rnd = 1_000_003
    hq = [rnd - 1 - (i * 7919) % 50 for i in range(n)]
    dgs = b"".join(hashlib.sha512(struct.pack("<QQ", rnd, h)).digest()[:32] for h in hq)
    seeds = b"".join(synth.key_seed(700_000_000 + i) for i in range(n))
    pk = _lib.public_keys(seeds, n)
    sg = _lib.sign_batch(b"".join(seeds[32 * i:32 * i + 32] + pk[32 * i:32 * i + 32] for i in range(n)), dgs, n, shared=False)
    hq_b = struct.pack("<%dQ" % n, *hq)
    lib = _lib.lib()
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    ks = _lib.KeySet(keys + [synth.key_seed(710_000_000 + i) for i in range(validators - n)])
    idx_b = struct.pack("<%dI" % n, *range(n))
    b_hq, b_pk, b_sg, b_ix = (ctypes.create_string_buffer(x, len(x)) for x in (hq_b, pk, sg, idx_b))
    a_hq, a_pk, a_sg, a_ix = (ctypes.addressof(x) for x in (b_hq, b_pk, b_sg, b_ix))
    med = lambda v: sorted(v)[len(v) // 2]
    p99 = lambda v: sorted(v)[int(len(v) * 0.99)]
    res = {"n_validators": validators, "signatures": n, "calls": calls,
           "what": "TC::verify: per-vote digests SHA-512(round || high_qc_round)[..32] built on the device, strict"}
    for name, call in (("cold", lambda: lib.hs_ed25519_verify_tc(rnd, a_hq, a_pk, a_sg, n, None)),
                       ("key_cache", lambda: lib.hs_ed25519_verify_tc_cached(ks.handle, rnd, a_hq, a_ix, a_sg, n, None))):
        lat = []
        for i in range(calls + 20):
            t0 = time.perf_counter()
            rc = call()
            dt = time.perf_counter() - t0
            assert rc == 1, "TC rejected (%s, host API): %d %s" % (name, rc, _lib.last_error())
            if i >= 20:
                lat.append(dt)
        res["p50_ms_host_api_" + name], res["p99_ms_host_api_" + name] = round(med(lat) * 1e3, 4), round(p99(lat) * 1e3, 4)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_hq, d_pk, d_sg = t(hq_b), t(pk), t(sg)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)
    torch.cuda.synchronize(dev)
    for name, kw in (("cold", {}), ("key_cache", {"keyset": ks, "d_key_idx": d_ki.data_ptr()})):
        lat = []
        ctx.set_profiling(calls + 20)
        with torch.cuda.stream(st):
            for i in range(calls + 20):
                nr.zero_()
                st.synchronize()
                t0 = time.perf_counter()
                ctx.verify_tc(rnd, d_hq.data_ptr(), d_pk.data_ptr(), d_sg.data_ptr(), n, bm.data_ptr(), nr.data_ptr(), ctx.stream, **kw)
                bad = int(nr.item())
                dt = time.perf_counter() - t0
                assert bad == 0, "device-resident TC (%s): %d rejects" % (name, bad)
                if i >= 20:
                    lat.append(dt)
        kt = ctx.kernel_times()
        res["p50_ms_" + name], res["p99_ms_" + name] = round(med(lat) * 1e3, 4), round(p99(lat) * 1e3, 4)
        res["kernel_ms_" + name] = round((kt[0] + kt[1]) / max(1, kt[2]), 4)      # the verification kernel(s); k_tc_digests runs ahead of them
    assert all((int(w) & ((1 << 64) - 1)) == (1 << 64) - 1 for w in bm.view(torch.int64)[: n // 64].tolist()), "TC bitmap"
    ctx.close()
    ks.close()
    try:
        res["cpu"] = cpu_tc_latency(rnd, hq, pk, sg, n)
    except Exception as e:      # noqa: BLE001
        res["cpu"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def qc_latency(_lib, torch, synth, dev, validators=10000, calls=200):
    """BASELINE.json's second metric: p50 wall time of one QC::verify-equivalent call at N
    validators (n = floor(2N/3) + 1 signatures over one digest, batch semantics).  Measured
    (a) through the host-pointer C-ABI (pinned staging + PCIe both ways included) and
    (b) device-resident (inputs in HBM, verdict word read back)."""
    n = 2 * validators // 3 + 1
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    lib = _lib.lib()
    # what distinguishes this box from the next one, recorded BESIDE the latencies (VERDICT r3 "Next round" #1): clocks, power
    # cap, partition modes, queue budget, and the measured clock of a lone wavefront - a QC's latency is one wavefront's chain
    try:
        import gpu_env
        env = gpu_env.snapshot()
    except Exception as e:      # noqa: BLE001
        env = {"error": "%s: %s" % (type(e).__name__, e)}
    b_dg, b_pk, b_sg = (ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg))
    host = []
    for i in range(calls + 20):
        t0 = time.perf_counter()
        rc = lib.hs_ed25519_verify_shared(ctypes.addressof(b_dg), ctypes.addressof(b_pk), ctypes.addressof(b_sg),
                                          n, _lib.BATCH, None)
        dt = time.perf_counter() - t0
        assert rc == 1, "QC batch rejected"
        if i >= 20:
            host.append(dt)
    # (a') the same from caller-owned pinned buffers (hs_ed25519_host_alloc): the kernels read them in place, no staging copy
    pinned = None
    try:
        pb = [_lib.PinnedBuffer(x) for x in (dg, pk, sg)]
        pinned = []
        for i in range(calls + 20):
            t0 = time.perf_counter()
            rc = lib.hs_ed25519_verify_shared(pb[0].addr, pb[1].addr, pb[2].addr, n, _lib.BATCH, None)
            dt = time.perf_counter() - t0
            assert rc == 1, "QC batch rejected (caller-pinned buffers)"
            if i >= 20:
                pinned.append(dt)
        for b in pb:
            b.close()
    except _lib.EngineError:
        pinned = None
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    bm = torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev)
    nr = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx = _lib.DeviceContext(n)
    st = torch.cuda.ExternalStream(ctx.stream, device=dev)     # the context's own stream
    torch.cuda.synchronize(dev)                                # uploads / fills above ran on torch's stream
    devt = []
    ctx.set_profiling(calls + 20)                              # the kernel's own duration (dispatch timestamps) beside the call's wall time
    with torch.cuda.stream(st):
        for i in range(calls + 20):
            nr.zero_()
            st.synchronize()
            t0 = time.perf_counter()
            ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(),
                       nr.data_ptr(), ctx.stream)
            bad = int(nr.item())        # D2H of the verdict word on the same stream = end of the call
            dt = time.perf_counter() - t0
            assert bad == 0, "device-resident QC (cold): %d rejects at call %d" % (bad, i)
            if i >= 20:
                devt.append(dt)
    full = bm.view(torch.int64)[: n // 64].tolist()
    assert all((int(w) & ((1 << 64) - 1)) == (1 << 64) - 1 for w in full), "QC bitmap"
    kt = ctx.kernel_times()
    cold_kernel_ms = (kt[0] + kt[1]) / max(1, kt[2])
    # (b') BOTH eight-lane cold forms on THIS box, in this run: the library chose one of them by timing both
    # at init (hs_ed25519_autotune); here each runs in four alternating blocks of 30 calls through the per-context selectors,
    # kernel time from the dispatch timestamps, call time from the host clock - so the line shows which form is faster here,
    # by how much, and whether the library's pick agrees
    forms = {}
    try:
        acc = {"k_verify_fused8": ([], [0.0, 0]), "k_verify_fused8h": ([], [0.0, 0])}
        with torch.cuda.stream(st):
            for block in range(4):
                for name, sel in (("k_verify_fused8", _lib.KERNEL_OCT), ("k_verify_fused8h", _lib.KERNEL_OCT_SPLIT)):
                    ctx.set_kernel(sel)
                    ctx.set_profiling(30)
                    for i in range(32):
                        nr.zero_()
                        st.synchronize()
                        t0 = time.perf_counter()
                        ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                        bad = int(nr.item())
                        dt = time.perf_counter() - t0
                        assert bad == 0, "device-resident QC (%s): %d rejects" % (name, bad)
                        if i >= 2:
                            acc[name][0].append(dt)
                    k = ctx.kernel_times()
                    acc[name][1][0] += k[0] + k[1]
                    acc[name][1][1] += k[2]
        ctx.set_kernel(_lib.KERNEL_AUTO)
        for name, (lat, (ms, cnt)) in acc.items():
            lat.sort()
            forms[name] = {"kernel_ms": round(ms / max(1, cnt), 4), "call_p50_ms": round(lat[len(lat) // 2] * 1e3, 4)}
        faster = min(forms, key=lambda k: forms[k]["kernel_ms"])
        slower = max(forms, key=lambda k: forms[k]["kernel_ms"])
        forms["faster_here"] = faster
        forms["faster_by"] = round(forms[slower]["kernel_ms"] / forms[faster]["kernel_ms"] - 1.0, 4)
    except Exception as e:      # noqa: BLE001
        forms = {"error": "%s: %s" % (type(e).__name__, e)}
        ctx.set_kernel(_lib.KERNEL_AUTO)
    try:
        cfg = _lib.config_dump()
        d0 = cfg["devices"][0]
        forms["library"] = {"chosen": _lib.lib().hs_ed25519_dominant_kernel_for(n).decode(),
                            "chosen_by": ("environment HS_ED25519_OCT_SPLIT" if cfg["oct_split_forced"] >= 0 else
                                          "hs_ed25519_autotune at init" if d0["oct_pick"] != "not measured" else "default (not measured)"),
                            "autotune_ms": d0["oct_ms"]}
    except Exception as e:      # noqa: BLE001
        forms["library"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # (c) the same QC with the whole committee registered as a key set (static per epoch)
    t0 = time.perf_counter()
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    keys += [synth.key_seed(20_000_000 + i) for i in range(validators - n)]    # non-voting members
    ks = _lib.KeySet(keys)
    build_ms = (time.perf_counter() - t0) * 1e3
    import struct
    idx = struct.pack("<%dI" % n, *range(n))
    b_ix = ctypes.create_string_buffer(idx, len(idx))
    hostc = []
    for i in range(calls + 20):
        t0 = time.perf_counter()
        rc = lib.hs_ed25519_verify_shared_cached(ks.handle, ctypes.addressof(b_dg), ctypes.addressof(b_ix),
                                                 ctypes.addressof(b_sg), n, _lib.BATCH, None)
        dt = time.perf_counter() - t0
        assert rc == 1, "cached QC batch rejected"
        if i >= 20:
            hostc.append(dt)
    d_ki = torch.arange(n, dtype=torch.int32, device=dev)
    devc = []
    ctx.set_profiling(calls + 20)
    with torch.cuda.stream(st):
        for i in range(calls + 20):
            nr.zero_()
            st.synchronize()
            t0 = time.perf_counter()
            ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(),
                              nr.data_ptr(), ctx.stream)
            bad = int(nr.item())
            dt = time.perf_counter() - t0
            assert bad == 0, "device-resident QC (key cache): %d rejects at call %d" % (bad, i)
            if i >= 20:
                devc.append(dt)
    kt = ctx.kernel_times()
    cached_kernel_ms = (kt[0] + kt[1]) / max(1, kt[2])
    # (d) what the back-to-back p50 does not say: the same two calls ALTERNATING - every QC kernel then starts on
    # another large kernel's instructions, as it does in a node that verifies one certificate per round between other work
    alternating = None
    try:
        lc, lk = [], []
        ctx.set_profiling(0)
        with torch.cuda.stream(st):
            for i in range(70):
                for which, acc in (("cold", lc), ("key_cache", lk)):
                    nr.zero_()
                    st.synchronize()
                    t0 = time.perf_counter()
                    if which == "cold":
                        ctx.verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                    else:
                        ctx.verify_cached(ks, d_dg.data_ptr(), 0, d_ki.data_ptr(), d_sg.data_ptr(), n, _lib.BATCH, bm.data_ptr(), nr.data_ptr(), ctx.stream)
                    bad = int(nr.item())
                    dt = time.perf_counter() - t0
                    assert bad == 0, "alternating QC (%s): %d rejects" % (which, bad)
                    if i >= 10:
                        acc.append(dt)
        alternating = {"p50_ms_cold": round(sorted(lc)[len(lc) // 2] * 1e3, 4), "p50_ms_key_cache": round(sorted(lk)[len(lk) // 2] * 1e3, 4),
                       "what": "device-resident, one cold and one key-cache QC call in turn, 60 of each"}
    except Exception as e:      # noqa: BLE001
        alternating = {"error": "%s: %s" % (type(e).__name__, e)}
    ks_bytes = ks.hbm_bytes()
    ks.close()
    ctx.close()
    med = lambda v: sorted(v)[len(v) // 2]
    p99 = lambda v: sorted(v)[int(len(v) * 0.99)]
    try:
        cpu = cpu_qc_latency(dg, pk, sg, n)
    except Exception as e:
        cpu = {"error": "%s: %s" % (type(e).__name__, e)}
    p10 = lambda v: sorted(v)[len(v) // 10]
    return {"n_validators": validators, "signatures": n, "calls": calls, "cpu": cpu, "env": env,
            "p10_p50_p99_ms_device_resident": [round(f(devt) * 1e3, 4) for f in (p10, med, p99)],
            "kernel_ms_device_resident": round(cold_kernel_ms, 4), "kernel": _lib.lib().hs_ed25519_dominant_kernel_for(n).decode(),
            "cold_forms": forms, "alternating_calls": alternating,
            "p50_ms_host_api": round(med(host) * 1e3, 4), "p99_ms_host_api": round(p99(host) * 1e3, 4),
            "p50_ms_host_api_caller_pinned": round(med(pinned) * 1e3, 4) if pinned else None,
            "p50_ms_device_resident": round(med(devt) * 1e3, 4),
            "verifies_per_s_at_p50_host_api": round(n / med(host), 1),
            "committee_cached": {"keys": validators, "keyset_hbm_bytes": ks_bytes, "keyset_build_ms": round(build_ms, 1),
                                 "kernel": _lib.lib().hs_ed25519_cached_kernel_for(n).decode(),
                                 "p50_ms_host_api": round(med(hostc) * 1e3, 4),
                                 "verifies_per_s_at_p50_host_api": round(n / med(hostc), 1),
                                 "p50_ms_device_resident": round(med(devc) * 1e3, 4),
                                 "p10_p50_p99_ms_device_resident": [round(f(devc) * 1e3, 4) for f in (p10, med, p99)],
                                 "kernel_ms_device_resident": round(cached_kernel_ms, 4)}}


def streamed_rounds(_lib, synth, validators=10000, slots=4, seconds=0.6):
    """BASELINE configs[4]: QC rounds streamed from host memory by ONE host thread (the
    reference's single Core task) through the round ring - pinned slots, H2D / kernel / D2H of
    different slots overlapped on separate HIP streams; a round travels as one hipGraph launch,
    as the same three operations issued to the stream one by one, or with no copies at all (the
    kernels read the pinned slot over PCIe).  PCIe-inclusive."""
    import numpy as np
    n = 2 * validators // 3 + 1
    dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
    keys = [pk[32 * i:32 * i + 32] for i in range(n)]
    out = {"n_validators": validators, "signatures_per_round": n, "slots": slots, "host_threads": 1,
           "target_signatures_per_s": 1e6}
    for cached in (False, True):
        ks = _lib.KeySet(keys) if cached else None
        kb = ks.indices(keys)[0] if cached else pk
        src = [np.frombuffer(x, dtype=np.uint8) for x in (dg, kb, sg)]
        for kind in ("graph", "stream-ops", "zero-copy"):
            ring = _lib.RoundRing(n, slots=slots, keyset=ks, kind=kind)
            inflight, lat, t_sub, done, r = set(), [], {}, 0, 0
            warm = 2 * slots
            t0 = t_end = None
            while t_end is None or time.perf_counter() < t_end or inflight:
                if r == warm:
                    t0 = time.perf_counter(); t_end = t0 + seconds; done = 0; lat = []
                slot = r % slots
                if slot in inflight:
                    ok, _ = ring.wait(slot, want_bitmap=False)
                    lat.append(time.perf_counter() - t_sub[slot])
                    assert ok, "streamed round rejected"
                    inflight.discard(slot)
                    done += 1
                if t_end is None or time.perf_counter() < t_end:
                    d, k, s_ = ring.begin(slot, n, True)
                    d[:] = src[0]; k[:] = src[1]; s_[:] = src[2]
                    t_sub[slot] = time.perf_counter()
                    ring.submit(slot, _lib.BATCH)
                    inflight.add(slot)
                r += 1
            wall = time.perf_counter() - t0
            captures, _ = ring.stats()
            ring.close()
            lat.sort()
            out[("key_cache_" if cached else "cold_") + kind.replace("-", "_")] = {
                "signatures_per_s": round(done * n / wall, 1), "rounds": done,
                "submit_to_verdict_p50_ms": round(lat[len(lat) // 2] * 1e3, 4), "graph_captures": captures}
        if ks:
            ks.close()
    return out


def sharded_committee(_lib, torch, dist, synth, sharding, dev, rank, world, validators=100000, iters=60):
    """BASELINE configs[3]: ONE very large committee's QC (n = floor(2N/3)+1 signatures over one
    digest) cut into one contiguous shard per rank; every rank verifies its shard on its GPU and an
    RCCL all-gather of the per-shard accept bitmaps gives every rank the full bitmap and the
    verdict (hotstuff_amd/sharding.py - the product path, verdict read on the host included).
    Called by every rank in lockstep.  Times whole calls: p50 on rank 0, and signatures/s from
    the slowest rank's total."""
    n = 2 * validators // 3 + 1
    lo, hi, S = sharding.shard_bounds(n, world, rank)
    m = hi - lo
    # this rank's shard of the synthetic QC, signed on the GPU (hs_ed25519_sign_batch): milliseconds for any shard
    seeds = b"".join(synth.key_seed(500_000_000 + lo + i) for i in range(m))
    pk = _lib.public_keys(seeds, m)
    dg = synth.shared_digest(11)
    sg = _lib.sign_batch(b"".join(seeds[32 * i:32 * i + 32] + pk[32 * i:32 * i + 32] for i in range(m)), dg, m, shared=True)
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
    out = {"n_validators": validators, "signatures": n, "ranks": world, "signatures_per_shard": S,
           "exchange": "%s all-gather of %d-byte accept bitmaps, one per QC"
                       % ("RCCL" if dist.get_backend() == "nccl" else "gloo (ranks share a device: rehearsal)", S // 8), "calls": iters}
    for cached in (False, True):
        ks = d_ki = None
        if cached:
            ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(m)])      # this rank's slice of the committee
            d_ki = torch.arange(m, dtype=torch.int32, device=dev)
        v = sharding.GpuShardVerifier(d_dg, 0, None if cached else d_pk, d_sg, _lib.BATCH, max_batch=S,
                                      keyset=ks, key_idx=d_ki, first=lo)
        for _ in range(5):
            ok, _bm = sharding.verify_sharded(v, n)
            assert ok, "sharded QC rejected"
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
        times = []
        for _ in range(iters):
            t0 = time.perf_counter()
            ok, _bm = sharding.verify_sharded(v, n)
            times.append(time.perf_counter() - t0)
            assert ok, "sharded QC rejected"
        tot = torch.tensor([sum(times)], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tot, op=dist.ReduceOp.MAX)
        times.sort()
        out["key_cache" if cached else "cold"] = {
            "p50_ms": round(times[len(times) // 2] * 1e3, 4), "p99_ms": round(times[int(len(times) * 0.99)] * 1e3, 4),
            "verifies_per_s": round(iters * n / float(tot.item()), 1)}
        v.ctx.close()
        if ks:
            ks.close()
    return out


def end_to_end(_lib, batches, n, stride, mode, threads=16, seconds=0.8,
               inflight=((16, 2, 2), (1, 32, 2), (16, 4, 4), (1, 64, 4), (16, 8, 8), (16, 16, 16), (16, 32, 16), (4, 64, 16), (4, 128, 16),
                         (1, 128, 8), (1, 256, 16), (1, 512, 16)),
               pinned_shapes=((1, 256, 16), (1, 512, 16), (1, 64, 4), (4, 128, 16))):
    """SURVEY.md 8d's metric proper: host buffers in, host verdict out, through the host-pointer C ABI.
    `threads` NATIVE host threads (tools/libhs_loadgen.so, a client of the public ABI - interpreter threads would measure
    the interpreter lock) each verify their own 1,024-tuple batch in a loop; every call stages into pinned memory,
    launches on a pool context's CU-partitioned stream and waits for its verdict.  PCIe-inclusive.  Rows: the synchronous
    calls (hs_ed25519_verify_multi) from 1, threads/2 and threads threads, then the asynchronous form
    (hs_ed25519_submit / hs_ed25519_ticket_wait) as (threads, tickets in flight per thread, batches per launch).
    `caller_pinned` rows: the same calls on buffers the caller allocated with hs_ed25519_host_alloc - the engine then reads
    them in place instead of copying them into its staging (what the shim's packing of Vec<(PublicKey, Signature)> straight
    into such memory gives): the reference's caller is ONE task, and one thread's bursts of sixteen are bound by that copy."""
    dg = b"".join(b[0] for b in batches)
    pk = b"".join(b[1] for b in batches)
    sg = b"".join(b[2] for b in batches)
    bufs = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
    a = [ctypes.addressof(x) for x in bufs]
    out = {"unit": "verifies/s", "signatures_per_call": n,
           "path": "host buffers -> pinned staging -> kernel reads over PCIe (zero-copy) -> pinned bitmap -> host verdict",
           "api": "rows with in_flight_per_thread 1: hs_ed25519_verify_multi (synchronous); above: hs_ed25519_submit bursts "
                  "(HS_ED25519_SUBMIT_MORE on all but the last of `batches_per_launch`) + hs_ed25519_ticket_wait"}
    best = None
    # (host threads, tickets in flight per thread): the synchronous calls at 1, half and all the threads, then
    # the asynchronous form (hs_ed25519_submit / hs_ed25519_ticket_wait) with the same threads
    shapes = [(th, 1, 1) for th in sorted({1, threads // 2, threads}) if th >= 1]
    shapes += [(th, k, bpl) for th, k, bpl in inflight]

    def run(ptrs, th, k, bpl):
        calls, p50, el = _lib.loadgen(None, ptrs[0], stride, ptrs[1], ptrs[2], n, len(batches), mode, th, k, bpl,
                                      seconds if th == threads else seconds / 2)
        return {"host_threads": th, "in_flight_per_thread": k, "batches_per_launch": bpl, "calls": calls, "seconds": round(el, 3),
                "call_p50_ms": round(p50, 4), "value": round(calls * n / el, 1)}
    # untimed: the pools (the narrow pool's sixteen CU-masked streams are created at the first small call), their staging, the
    # workspaces of every kernel form - so that the first row, ONE synchronous caller, times calls and not the pool's creation
    for th, k, bpl in ((threads, 4, 4), (threads, 16, 16), (1, 1, 1)):
        _lib.loadgen(None, a[0], stride, a[1], a[2], n, len(batches), mode, th, k, bpl, 0.15)
    for th, k, bpl in shapes:
        row = run(a, th, k, bpl)
        out.setdefault("by_threads", []).append(row)
        if best is None or row["value"] > best["value"]:
            best = row
    out.update(best)
    try:
        pin = [_lib.PinnedBuffer(x) for x in (dg, pk, sg)]
        pa = [x.addr for x in pin]
        out["caller_pinned"] = {"buffers": "hs_ed25519_host_alloc (pinned, device-mapped): read in place, no staging copy",
                                "rows": [run(pa, th, k, bpl) for th, k, bpl in pinned_shapes]}
        for x in pin:
            x.close()
    except Exception as e:
        out["caller_pinned"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def multi_process(n=1024, procs=(1, 2, 4), seconds=1.0, timeout_s=120, shape=(4, 8, 4), extra_env=None, tenancy_modes=None):
    """several node processes share one GPU (the reference's `fab local` starts four node processes on one
    host, benchmark/benchmark/local.py:94-104).  k independent PROCESSES (tools/mp_worker.py, each its own HIP context,
    pools and streams on device 0) run the end-to-end host-buffer load at once - ticket bursts of four 1,024-signature
    batches from 4 native threads each - started together; aggregate verifies/s and the median call latency, without and
    with the tenancy hint (hs_ed25519_set_tenancy: each process 1/k of the stream budget and its own slice of the CUs)."""
    out = {"signatures_per_call": n, "seconds": seconds,
           "worker": "tools/mp_worker.py: %d native threads x %d tickets, bursts of %d" % tuple(shape), "rows": []}
    env0 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                              "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    for k in procs:
        for tenancy in (tenancy_modes or ((False,) if k == 1 else (False, True))):
            ps = []
            try:
                for i in range(k):
                    env = dict(env0)
                    env.update(extra_env or {})
                    if tenancy:
                        env["HS_ED25519_TENANTS"], env["HS_ED25519_TENANT_INDEX"] = str(k), str(i)
                    ps.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "mp_worker.py"), str(n), str(seconds), str(i)] + [str(x) for x in shape],
                                               stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env))
                t_end = time.time() + timeout_s
                for p in ps:            # every worker has initialised, registered nothing else, and waits for the start signal
                    line = p.stdout.readline()
                    if not line.startswith("ready"):
                        raise RuntimeError("worker did not come up: %r" % line[:200])
                for p in ps:
                    p.stdin.write("go\n"); p.stdin.flush()
                res = []
                for p in ps:
                    o, _ = p.communicate(timeout=max(1.0, t_end - time.time()))
                    res.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
                if any("error" in r for r in res):
                    raise RuntimeError(str([r["error"] for r in res if "error" in r][:2]))
                out["rows"].append({"processes": k, "tenancy_hint": tenancy, "value": round(sum(r["verifies_per_s"] for r in res), 1),
                                    "per_process": [round(r["verifies_per_s"], 1) for r in res],
                                    "call_p50_ms": round(sorted(r["p50_ms"] for r in res)[len(res) // 2], 4),
                                    "verdict_errors": sum(r.get("verdict_errors", 0) for r in res)})
            except Exception as e:
                out["rows"].append({"processes": k, "tenancy_hint": tenancy, "error": "%s: %s" % (type(e).__name__, e)})
            finally:
                for p in ps:
                    if p.poll() is None:
                        p.kill()
    one = [r["value"] for r in out["rows"] if r.get("processes") == 1 and "value" in r]
    if one:
        for r in out["rows"]:
            if "value" in r:
                r["vs_one_process"] = round(r["value"] / one[0], 3)
    return out


def small_batch_crossover(_lib, synth):
    """where the GPU round trip starts to beat one CPU core.  p50 of the host-pointer
    call at n = 1 ... 64 (strict, one digest per signature: the Vote / Timeout / TC shape), cold and with the
    keys registered, against the oracle's per-signature strict verification on one core."""
    lib = _lib.lib()
    orc = load_oracle()
    S = ctypes.c_size_t
    sizes = [1, 2, 3, 4, 8, 16, 32, 64]
    nmax = sizes[-1]
    dg, pk, sg = synth.make_batch(nmax, shared=False, cache_dir="/tmp/hs_synth")
    ks = _lib.KeySet([pk[32 * i:32 * i + 32] for i in range(nmax)])
    import struct
    idx = struct.pack("<%dI" % nmax, *range(nmax))
    b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg, idx)]
    a = [ctypes.addressof(x) for x in b]
    med = lambda v: sorted(v)[len(v) // 2]
    rows, cross = [], {"cold": None, "key_cache": None}
    for n in sizes:
        cold, cached, cpu = [], [], []
        for i in range(60):
            t0 = time.perf_counter(); rc = lib.hs_ed25519_verify_multi(a[0], a[1], a[2], n, _lib.STRICT, None); cold.append(time.perf_counter() - t0)
            assert rc == 1
            t0 = time.perf_counter(); rc = lib.hs_ed25519_verify_multi_cached(ks.handle, a[0], a[3], a[2], n, _lib.STRICT, None); cached.append(time.perf_counter() - t0)
            assert rc == 1
        out = ctypes.create_string_buffer(n)
        for i in range(20):
            t0 = time.perf_counter(); orc.hs_oracle_verify_many(dg, S(32), S(32), pk, sg, S(n), 1, out, 1); cpu.append(time.perf_counter() - t0)
        row = {"n": n, "gpu_cold_ms": round(med(cold[10:]) * 1e3, 4), "gpu_key_cache_ms": round(med(cached[10:]) * 1e3, 4),
               "cpu_one_core_ms": round(med(cpu) * 1e3, 4)}
        rows.append(row)
        if cross["cold"] is None and row["gpu_cold_ms"] <= row["cpu_one_core_ms"]:
            cross["cold"] = n
        if cross["key_cache"] is None and row["gpu_key_cache_ms"] <= row["cpu_one_core_ms"]:
            cross["key_cache"] = n
    ks.close()
    return {"rows": rows, "measured_crossover_n": cross,
            "library_hint": {"cold": lib.hs_ed25519_min_gpu_batch(0), "key_cache": lib.hs_ed25519_min_gpu_batch(1)},
            "cpu": "oracle per-signature strict verification, one core (cpu_baseline brackets it with OpenSSL's and libsodium's single verification on the same cores; dalek cannot be built here)"}


def rlc_ab(_lib, torch, synth, dev, sizes=(6667, 66667), calls=40):
    """A/B on the cold QC path: per-signature BATCH kernels vs HS_ED25519_BATCH_RLC (one
    random linear combination = Pippenger MSM, per-signature fallback skipped when it accepts).  Device
    resident; p50 of one call, and signatures/s with 4 contexts in flight."""
    out = {}
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    for n in sizes:
        dg, pk, sg = synth.make_batch(n, shared=True, cache_dir="/tmp/hs_synth")
        d_dg, d_pk, d_sg = t(dg), t(pk), t(sg)
        nctx = 4
        ctxs = [_lib.DeviceContext(n) for _ in range(nctx)]
        for c in ctxs:
            c.set_rlc_min(1)
        bms = [torch.zeros(8 * ((n + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(nctx)]
        nrs = torch.zeros(nctx, dtype=torch.int32, device=dev)
        sts = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
        torch.cuda.synchronize(dev)
        row = {}
        for name, mode in (("per_signature", _lib.BATCH), ("rlc_msm", _lib.BATCH_RLC)):
            def call(j):
                ctxs[j].verify(d_dg.data_ptr(), 0, d_pk.data_ptr(), d_sg.data_ptr(), n, mode, bms[j].data_ptr(),
                               nrs[j:].data_ptr(), 0)
            lat = []
            for i in range(calls + 5):
                sts[0].synchronize()
                t0 = time.perf_counter()
                call(0)
                sts[0].synchronize()
                if i >= 5:
                    lat.append(time.perf_counter() - t0)
            assert bool((bms[0].view(torch.int64)[: n // 64] == -1).all()), "rejected a valid QC"
            if mode == _lib.BATCH_RLC:
                assert ctxs[0].rlc_last()[0] == 1, "the RLC stage did not accept a valid QC"
            for c in ctxs:                     # four contexts kept busy: the caller says so (hs_ed25519_ctx_set_concurrent)
                c.set_concurrent(True)
            for j in range(nctx):
                call(j)
            torch.cuda.synchronize(dev)
            for s_ in sts:
                s_.synchronize()
            t0 = time.perf_counter()
            reps = 12
            for i in range(reps):
                for j in range(nctx):
                    call(j)
            for s_ in sts:
                s_.synchronize()
            wall = time.perf_counter() - t0
            for c in ctxs:
                c.set_concurrent(False)
            lat.sort()
            row[name] = {"p50_ms": round(lat[len(lat) // 2] * 1e3, 4), "signatures_per_s_4_in_flight": round(reps * nctx * n / wall, 1)}
        row["winner_latency"] = min(("per_signature", "rlc_msm"), key=lambda k: row[k]["p50_ms"])
        row["winner_throughput"] = max(("per_signature", "rlc_msm"), key=lambda k: row[k]["signatures_per_s_4_in_flight"])
        out[str(n)] = row
        for c in ctxs:
            c.close()
    out["library_threshold"] = _lib.lib().hs_ed25519_rlc_min_batch()
    return out


def sharded_c_abi(ngpus, ndev, validators=100000, timeout_s=150):
    """BASELINE configs[3] from ONE process through the C ABI (hs_ed25519_init_devices +
    hs_ed25519_verify_shared_sharded; host and in-process RCCL exchange; per-signature and RLC): tools/sharded_c_abi.py,
    run in a SUBPROCESS with a timeout - this path has never run on multi-GPU hardware, and a fault or a hang there
    must not cost the bench line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                              "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    try:
        env["HS_SHARDED_DUMP_AFTER_S"] = str(max(10, timeout_s - 20))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sharded_c_abi.py"), str(ngpus), str(validators), "30", str(ndev)],
                           capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired as e:
        # the tool prints a dated stage marker per step on stderr, and the stack of every thread 20 s before the limit
        err = e.stderr.decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        return {"error": "tools/sharded_c_abi.py did not finish within %d s" % timeout_s, "stderr_tail": err[-1500:]}
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "tools/sharded_c_abi.py exited %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
    return json.loads(lines[-1])


PMC_SETS = (("FETCH_SIZE",), ("WRITE_SIZE",),
            ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVES", "GRBM_GUI_ACTIVE"),
            # round 6: where a wavefront's cycles go - parked (s_waitcnt / barrier), issuing, and what it issues.
            # MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES (disjoint), all in quad-cycles
            ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA",
             "SQ_INST_CYCLES_VMEM_RD", "SQ_IFETCH"))
CALIBRATION = os.path.join("profiles", "pmc_calibration.json")


def pmc_passes(timeout_s=150):
    """Counters of every kernel whose roofline this line quotes, measured IN THIS RUN: four rocprofv3 --pmc passes (FETCH_SIZE
    and WRITE_SIZE do not fit one pass; two SQ sets of at most eight; no trace domains beside the counters) over SOLO launches
    (tools/pmc_solo.py bench: one launch at a time, the stream drained in between, so the chip-wide counters of a dispatch hold
    that dispatch only), summed over a counter's instances, the first two dispatches of each kernel@grid dropped.
    FETCH_SIZE / WRITE_SIZE are corrected with the calibration in profiles/pmc_calibration.json (FETCH_SIZE reports half of the
    128-byte lines fetched; WRITE_SIZE is exact for whole lines; tools/pmc_calibrate.py).  Returns
    {"kernel@workgroups": {counter: per-dispatch average, "read_bytes", "write_bytes", "bytes"}, "calibration": {...}} or None
    when rocprofv3 is not there."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    cal = json.load(open(os.path.join(ROOT, CALIBRATION)))["factors"]
    f_read, f_write = cal["fetch_line_gather_16B"], cal["write_wide_16B"]
    env = dict(os.environ, TMPDIR="/tmp")
    res, errors = {}, []
    for counters in PMC_SETS:
        d = tempfile.mkdtemp(prefix="hs_pmc_", dir="/tmp")
        try:
            r = subprocess.run(["rocprofv3", "--pmc", *counters, "-d", d, "-o", "p", "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_solo.py"), "bench"],
                               capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            dbs = glob.glob(os.path.join(d, "**", "p_results.db"), recursive=True)
            if not dbs:
                errors.append("%s: no results (rc %d): %s" % (counters[0], r.returncode, (r.stderr or "")[-200:]))
                continue
            db = sqlite3.connect(dbs[0])
            grid = {did: gx // max(1, wx) for did, gx, wx in
                    db.execute("select dispatch_id, grid_size_x, workgroup_size_x from rocpd_kernel_dispatch")}
            rows = db.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events "
                              "group by name, dispatch_id, counter_name order by dispatch_id").fetchall()
            per = {}
            for name, did, counter, val in rows:
                per.setdefault(("%s@%d" % (name.split("(")[0], grid.get(did, 0)), counter), []).append(val)
            for (key, counter), vals in per.items():
                vals = vals[2:] if len(vals) > 2 else vals
                res.setdefault(key, {})[counter] = sum(vals) / len(vals)
        except Exception as e:      # noqa: BLE001 - a pass that fails costs its counters, not the line
            errors.append("%s: %s: %s" % (counters[0], type(e).__name__, e))
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for key, c in res.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd, wr = c["FETCH_SIZE"] * 1024.0 * f_read, c["WRITE_SIZE"] * 1024.0 * f_write      # rocprofv3 reports KB
            c.update({"read_bytes": int(rd), "write_bytes": int(wr), "bytes": int(rd + wr)})
    if not res:
        return {"error": "; ".join(errors) or "no counters"}
    res["calibration"] = {"file": CALIBRATION, "fetch_factor": f_read, "write_factor": f_write}
    if errors:
        res["errors"] = errors
    return res


def pmc_pick(pmc, kernel, workgroups=None):
    """the counters of `kernel` (at a launch of `workgroups` workgroups, when several sizes were launched) out of pmc_passes()"""
    if not pmc or "error" in pmc:
        return None
    keys = [k for k in pmc if "@" in k and k.split("@")[0] == kernel]
    if workgroups is not None:
        keys = [k for k in keys if int(k.split("@")[1]) == workgroups]
    return pmc[keys[0]] if len(keys) == 1 else None


def cached_rooflines(cached, pmc, n, stride):
    """HBM rooflines of the key-cache kernels from THIS run's counter passes (round 4 read a round-3 file and
    counted table lines as algorithmic bytes).  `frac` prices SURVEY 8d's algorithmic bytes - what the reference's call consumes
    per verification: 96 B with one shared digest, 128 B with one per tuple - against the 8 TB/s peak; the 48 table lines of
    128 B a verification gathers BY DESIGN (32 of the key's table, 16 of the base point's) are `design_traffic`, with its rate
    and its ratio to the algorithmic bytes; `traffic` is what the counters saw."""
    def roof(kernel_ms, sigs, per_verify, counters, kernels):
        algo, design = per_verify * sigs, (32 + 16) * 128 * sigs
        r = {"bound": "hbm", "kernel": kernels, "achieved": round(algo / (kernel_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(algo / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_launch": algo,
             "algorithmic_bytes_per_verification": per_verify, "avg_kernel_ms": round(kernel_ms, 4),
             "design_traffic": design, "design_traffic_over_algorithmic": round(design / algo, 1),
             "design_traffic_GBps": round(design / (kernel_ms * 1e-3) / 1e9, 1),
             "design_traffic_note": "48 table lines x 128 B per verification (32 from the key set, 16 from the base point's 16-bit table): "
                                    "the gather the key cache trades for the doublings - a rate worth knowing, not algorithmic bytes",
             "traffic": None, "traffic_source": "unmeasured (no counter pass in this run)"}
        got = [c for c in counters if c and "bytes" in c]
        if got and len(got) == len(counters):
            tr = sum(c["bytes"] for c in got)
            r.update({"traffic": tr, "traffic_over_algorithmic": round(tr / algo, 1), "traffic_over_design": round(tr / design, 3),
                      "traffic_source": "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over solo launches (tools/pmc_solo.py bench), calibrated"})
        return r
    KC = cached["batches_per_launch"]
    cached["roofline"] = roof(cached["avg_kernel_ms"], n * KC, ALGO_BYTES_MULTI if stride else ALGO_BYTES_SHARED,
                              [pmc_pick(pmc, "k_verify_cached_lane2r_batches")], "k_verify_cached_lane2r_batches")
    lc = cached.get("large_committee")
    if isinstance(lc, dict) and "avg_kernel_ms" in lc:
        nl = lc["signatures"]
        wg = (nl + 63) // 64
        if lc["kernel"] == "k_verify_cached_lane2r":
            ctrs, names = [pmc_pick(pmc, "k_verify_cached_lane2r", wg)], "k_verify_cached_lane2r"
        else:
            ctrs, names = [pmc_pick(pmc, "k_cached_chain2", wg), pmc_pick(pmc, "k_cached_finish")], "k_cached_chain2 + k_cached_finish"
        lc["roofline"] = roof(lc["avg_kernel_ms"], nl, ALGO_BYTES_SHARED, ctrs, names)
        two = [pmc_pick(pmc, "k_cached_chain2", wg), pmc_pick(pmc, "k_cached_finish")]
        if all(c and "bytes" in c for c in two):
            lc["two_launch_form_traffic"] = {"k_cached_chain2": two[0]["bytes"], "k_cached_finish": two[1]["bytes"],
                                             "over_algorithmic": round((two[0]["bytes"] + two[1]["bytes"]) / (ALGO_BYTES_SHARED * nl), 1),
                                             "valu_wave_insts_per_verify": round((two[0].get("SQ_INSTS_VALU", 0) + two[1].get("SQ_INSTS_VALU", 0)) / nl, 1)}


def summarise_extras(cfg, detail, pmc):
    """The headline scalars of the extras, flat, into `config` - the part of the line the driver's record keeps: the second half of
    BASELINE.json's metric (QC-verify p50 at N validators, cold and through the key cache, device-resident and host API), both cold
    forms' kernel time on this box with the SQ counters of each, the one-synchronous-caller rates, tenants, streamed rounds."""
    q = detail.get("qc_verify_latency", {})
    if "error" in q:
        cfg["qc_error"] = q["error"][:120]
    elif q:
        c = q["committee_cached"]
        cfg.update({"qc_validators": q["n_validators"], "qc_signatures": q["signatures"],
                    "qc_verify_p50_ms_cold": q["p50_ms_device_resident"], "qc_verify_p50_ms_key_cache": c["p50_ms_device_resident"],
                    "qc_verify_p99_ms_cold": q["p10_p50_p99_ms_device_resident"][2], "qc_verify_p99_ms_key_cache": c["p10_p50_p99_ms_device_resident"][2],
                    "qc_verify_p50_ms_cold_host_api": q["p50_ms_host_api"], "qc_verify_p50_ms_key_cache_host_api": c["p50_ms_host_api"],
                    "qc_kernel_ms_cold": q["kernel_ms_device_resident"], "qc_kernel_ms_key_cache": c["kernel_ms_device_resident"],
                    "qc_kernel_cold": q["kernel"], "qc_kernel_key_cache": c.get("kernel"),
                    "one_sync_caller_verifies_per_s_qc_cold": q["verifies_per_s_at_p50_host_api"],
                    "one_sync_caller_verifies_per_s_qc_cached": c.get("verifies_per_s_at_p50_host_api")})
        cpu = q.get("cpu") or {}
        if "p50_ms_one_core" in cpu:      # dalek's verify_batch as computed (RLC + Pippenger, oracle port), the CPU twin of the QC rows
            cfg["cpu_qc_verify_batch_p50_ms_one_core"], cfg["cpu_qc_verify_batch_p50_ms_all_cores"], cfg["cpu_cores"] = cpu["p50_ms_one_core"], cpu["p50_ms_all_cores"], cpu["cores"]
        elif "error" in cpu:
            cfg["cpu_qc_error"] = cpu["error"][:120]
        alt = q.get("alternating_calls") or {}
        if "p50_ms_cold" in alt:      # the cold-instruction regime: each QC kernel right after the other one
            cfg["qc_verify_p50_ms_cold_alternating"], cfg["qc_verify_p50_ms_key_cache_alternating"] = alt["p50_ms_cold"], alt["p50_ms_key_cache"]
        f = q.get("cold_forms", {})
        for name in ("k_verify_fused8", "k_verify_fused8h"):
            if name in f:
                cfg["qc_kernel_ms_%s" % name[2:]] = f[name]["kernel_ms"]
                cfg["qc_call_p50_ms_%s" % name[2:]] = f[name]["call_p50_ms"]
        if "faster_here" in f:
            cfg["qc_cold_form_faster_here"], cfg["qc_cold_form_faster_by"] = f["faster_here"], f["faster_by"]
        lib_ = f.get("library", {})
        if "chosen" in lib_:
            cfg["qc_cold_form_chosen"], cfg["qc_cold_form_chosen_by"] = lib_["chosen"], lib_["chosen_by"]
            cfg["autotune_ms_verify_fused8"] = lib_["autotune_ms"].get("k_verify_fused8")
            cfg["autotune_ms_verify_fused8h"] = lib_["autotune_ms"].get("k_verify_fused8h")
        # one solo rocprofv3 SQ pass of the QC kernels (6,667 signatures): what would explain a kernel-time difference between boxes
        n = q["signatures"]
        for name, wg in (("k_verify_fused8", (n + 63) // 64 * 8), ("k_verify_fused8h", (n + 63) // 64 * 8), ("k_verify_cached_lane2r", (n + 63) // 64)):
            c_ = pmc_pick(pmc, name, wg)
            if c_:
                for ctr, short in (("SQ_INSTS_VALU", "insts_valu"), ("SQ_WAVE_CYCLES", "wave_cycles"), ("SQ_BUSY_CYCLES", "busy_cycles"),
                                   ("SQ_WAIT_INST_ANY", "wait_inst_any"), ("SQ_WAVES", "waves"), ("SQ_WAIT_ANY", "wait_any"),
                                   ("SQ_ACTIVE_INST_ANY", "active_any"), ("SQ_ACTIVE_INST_VALU", "active_valu"), ("SQ_ACTIVE_INST_VMEM", "active_vmem"),
                                   ("SQ_ACTIVE_INST_LDS", "active_lds"), ("SQ_ACTIVE_INST_SCA", "active_sca"), ("SQ_INST_CYCLES_VMEM_RD", "cycles_vmem_rd"),
                                   ("SQ_IFETCH", "ifetch")):
                    if ctr in c_:
                        cfg["sq_%s_%s" % (name[2:], short)] = round(c_[ctr])
                # do the three disjoint states account for the wavefronts' cycles?  (1.0 = they do; the driver's box decides what a
                # slower key-cache QC is made of: parked on memory / barriers, stalled at issue, or issuing)
                if all(k_ in c_ for k_ in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")) and c_["SQ_WAVE_CYCLES"] > 0:
                    cfg["sq_%s_states_sum_frac" % name[2:]] = round((c_["SQ_WAIT_ANY"] + c_["SQ_WAIT_INST_ANY"] + c_["SQ_ACTIVE_INST_ANY"]) / c_["SQ_WAVE_CYCLES"], 3)
    tc = detail.get("tc_verify_latency", {})
    if "error" in tc:
        cfg["tc_error"] = tc["error"][:120]
    elif tc:
        cfg.update({"tc_validators": tc["n_validators"], "tc_signatures": tc["signatures"],
                    "tc_verify_p50_ms_cold": tc["p50_ms_cold"], "tc_verify_p50_ms_key_cache": tc["p50_ms_key_cache"],
                    "tc_verify_p50_ms_cold_host_api": tc["p50_ms_host_api_cold"], "tc_verify_p50_ms_key_cache_host_api": tc["p50_ms_host_api_key_cache"],
                    "tc_kernel_ms_cold": tc["kernel_ms_cold"], "tc_kernel_ms_key_cache": tc["kernel_ms_key_cache"]})
        cpu = tc.get("cpu") or {}
        if "p50_ms_one_core" in cpu:
            cfg["cpu_tc_verify_p50_ms_one_core"], cfg["cpu_tc_verify_p50_ms_all_cores"] = cpu["p50_ms_one_core"], cpu["p50_ms_all_cores"]
            cfg.setdefault("cpu_cores", cpu["cores"])
        elif "error" in cpu:
            cfg["cpu_tc_error"] = cpu["error"][:120]
    ce = detail.get("callers_either_side", {})
    if "error" in ce:
        cfg["callers_error"] = ce["error"][:120]
    elif ce:
        # SURVEY 8f rows 1, 2 through the C++ twin (include/hotstuff_crypto.hpp), N = 10,000, committee registered: the leader's
        # 6,667 votes of a round arriving one by one (reference: one strict verification each on the Core task,
        # consensus/src/core.rs:203-225), and 200 timeouts that embed the same high_qc (reference: one verify_batch each,
        # messages.rs:261-263).  The reference's side is DERIVED from this run's own CPU rows (oracle ports), and says so.
        cfg.update({"vote_ingest_qc_ms": round(ce["vote_ingest_ms"], 3), "vote_ingest_engine_calls": ce["vote_ingest_engine_calls"],
                    "timeout_storm_200_ms": round(ce["timeout_storm_ms"], 3), "timeout_storm_qc_verifies": ce["qc_cache_misses"]})
        if "cpu_qc_verify_batch_p50_ms_one_core" in cfg:
            cfg["cpu_timeout_storm_200_ms_derived"] = round(200 * cfg["cpu_qc_verify_batch_p50_ms_one_core"], 1)
    mp = detail.get("multi_process", {})
    for r_ in mp.get("rows", []) if isinstance(mp, dict) else []:
        if r_.get("processes") == 4 and r_.get("tenancy_hint") and "value" in r_:
            cfg["four_tenants_bursts_of_4_verifies_per_s"] = r_["value"]
            cfg["four_tenants_bursts_of_4_max_over_min"] = round(max(r_["per_process"]) / max(1.0, min(r_["per_process"])), 3)
    for r_ in mp.get("rows_bursts_of_16", []) if isinstance(mp, dict) else []:
        if r_.get("processes") == 4 and r_.get("tenancy_hint") and "value" in r_:
            cfg["four_tenants_verifies_per_s"] = r_["value"]
            cfg["four_tenants_max_over_min"] = round(max(r_["per_process"]) / max(1.0, min(r_["per_process"])), 3)
    sr = detail.get("streamed_rounds", {})
    for key, short in (("cold_zero_copy", "streamed_cold_signatures_per_s"), ("key_cache_zero_copy", "streamed_key_cache_signatures_per_s")):
        if isinstance(sr.get(key), dict):
            cfg[short] = sr[key]["signatures_per_s"]
    ab = detail.get("rlc_ab", {})
    for nn in ("6667", "66667"):
        if isinstance(ab.get(nn), dict) and "rlc_msm" in ab[nn]:
            cfg["rlc_p50_ms_%s" % nn], cfg["per_signature_p50_ms_%s" % nn] = ab[nn]["rlc_msm"]["p50_ms"], ab[nn]["per_signature"]["p50_ms"]
    sb = detail.get("small_batch_crossover", {})
    if "measured_crossover_n" in sb:
        cfg["small_batch_crossover_n_cold"], cfg["small_batch_crossover_n_key_cache"] = sb["measured_crossover_n"]["cold"], sb["measured_crossover_n"]["key_cache"]


# What the driver's record keeps of the line: the first 24 keys of `config`, 40 characters of a key, 120 of a string (BENCH_r05's
# `parsed` lost both QC p50s to that).  So the keys that ARE the metric come first, in this order, and nothing in the objects the
# record keeps is longer than it may be (tests/test_bench_line.py holds finalize_line to that).
CONFIG_FIRST = (
    "workload", "qc_validators", "qc_signatures",
    "qc_verify_p50_ms_cold", "qc_verify_p50_ms_key_cache", "qc_verify_p50_ms_cold_host_api", "qc_verify_p50_ms_key_cache_host_api",
    "cpu_qc_verify_batch_p50_ms_one_core", "cpu_qc_verify_batch_p50_ms_all_cores",
    "tc_verify_p50_ms_cold", "tc_verify_p50_ms_key_cache", "cpu_tc_verify_p50_ms_one_core",
    "end_to_end_verifies_per_s_best", "one_sync_caller_1024_verifies_per_s", "one_sync_caller_verifies_per_s_qc_cold",
    "four_tenants_verifies_per_s", "int_roofline_mad_frac", "int_roofline_issue_frac", "entry_point", "cpu_cores",
    "cpu_tc_verify_p50_ms_all_cores", "e2e_one_thread_tickets_verifies_per_s", "qc_verify_p50_ms_cold_alternating", "tc_verify_p50_ms_cold_host_api")
KEY_MAX, STR_MAX = 40, 120
TOP_FIRST = ("metric", "value", "value_host_to_host", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def finalize_line(out, strict=False):
    """Order the line for the record: `value_host_to_host` beside `value`, the metric's keys first in `config`.  A key or a string
    the driver would cut is a programming error: `strict` (the unit test) raises on it; the bench run itself cuts it here, where
    the cut is chosen and counted (`line_cuts`), rather than lose a finished measurement over a name."""
    cfg = out.get("config", {})
    ordered = {k: cfg[k] for k in CONFIG_FIRST if k in cfg}
    ordered.update((k, v) for k, v in cfg.items() if k not in ordered)
    out["config"] = ordered
    if "end_to_end_verifies_per_s_best" in ordered:
        out["value_host_to_host"] = ordered["end_to_end_verifies_per_s_best"]      # SURVEY 8d's metric proper: host buffers in, host verdict out
    res = {k: out[k] for k in TOP_FIRST if k in out}
    res.update((k, v) for k, v in out.items() if k not in res)
    cuts = 0
    for name in ("config", "roofline", "cpu_baseline"):
        if not isinstance(res.get(name), dict):
            continue
        fixed = {}
        for k, v in res[name].items():
            if len(k) > KEY_MAX:
                if strict:
                    raise ValueError("bench line: %s key %r is longer than the %d characters the driver's record keeps" % (name, k, KEY_MAX))
                k, cuts = k[:KEY_MAX], cuts + 1
            if isinstance(v, str) and len(v) > STR_MAX:
                if strict:
                    raise ValueError("bench line: %s[%r] is %d characters, the driver's record keeps %d: %r" % (name, k, len(v), STR_MAX, v))
                v, cuts = v[:STR_MAX], cuts + 1
            fixed[k] = v
        res[name] = fixed
    if cuts:
        res["line_cuts"] = cuts
    return res


def launcher_command(ngpus, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when it was started WITHOUT a launcher
    (WORLD_SIZE unset): one rank per GPU under torch.distributed.run, rendezvous on 127.0.0.1 (the container's
    host name may not resolve).  The driver's own launch line is the same command, so both invocations run the
    same ranks."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


class line_watchdog:
    """Multi-GPU extras run AFTER the headline has been measured.  A collective that hangs there (this path has
    never seen a second device before the driver's SCALE run) blocks the main thread inside the runtime, and the
    process-group watchdog then aborts the process - with the headline unprinted.  Armed with the finished
    headline line, this prints it (with the extra marked as timed out) and ends the process when `seconds` pass
    without disarm()."""

    def __init__(self, fd):
        import threading
        self.fd, self.ev, self.th, self.threading = fd, None, None, threading

    def arm(self, make_line, seconds):
        self.disarm()
        ev = self.ev = self.threading.Event()

        def run():
            if not ev.wait(seconds):
                try:
                    os.write(self.fd, (make_line() + "\n").encode())
                finally:
                    os._exit(0)
        self.th = self.threading.Thread(target=run, daemon=True)
        self.th.start()

    def disarm(self):
        if self.ev is not None:
            self.ev.set()
            self.ev = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="signatures per batch (one verify call)")
    ap.add_argument("--batches-per-step", type=int, default=1024,
                    help="independent 1,024-tuple batches per step per GPU")
    ap.add_argument("--batches-per-launch", type=int, default=int(os.environ.get("HS_BENCH_BPL", "16")),
                    help="independent batches handed to one hs_ed25519_verify_device_batches call (1 = hs_ed25519_verify_device)")
    ap.add_argument("--kernel", choices=("auto", "fused", "pairs", "lane"), default=os.environ.get("HS_BENCH_KERNEL", "auto"),
                    help="hs_ed25519_verify_device_batches kernel: the library's choice by batch count, one signature per quad, "
                         "the pairs form (two signatures per quad), or the one-lane pair over the batch list (throughput at "
                         "several times the latency; wants --batches-per-launch 16 --partitions 0)")
    ap.add_argument("--shared", action="store_true", help="QC::verify shape (one digest, batch semantics)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("HS_BENCH_STREAMS", "16")))
    ap.add_argument("--distinct-batches", type=int, default=0,
                    help="distinct synthetic batches per GPU.  Default (0): as many as a step has batches - every tuple of a step is "
                         "distinct (1,048,576 per GPU), so no launch meets inputs or table lines a recent launch left in the caches.  "
                         "16 = rounds 1 - 4's set (one batch per stream context, re-verified); the rate is the same either way "
                         "(profiles/r5_n_distinct_batches.txt)")
    ap.add_argument("--cached-batches-per-launch", type=int, default=int(os.environ.get("HS_BENCH_CBPL", "4")),
                    help="batches per hs_ed25519_verify_device_batches_cached launch in the committee_cached extra")
    ap.add_argument("--cached-variants", type=int, default=16,
                    help="committee_cached extra: distinct message sets per batch of signers (1 = the same certificates re-verified)")
    ap.add_argument("--inputs", choices=("hbm", "pinned"), default="hbm",
                    help="measurement aid: 'pinned' leaves the headline's inputs in pinned host memory (hs_ed25519_host_alloc) and the kernels "
                         "read them over PCIe, as the host API's zero-copy calls do - never the figure of record (`value` is HBM-resident)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-qc", action="store_true", help="skip the QC-verify p50 latency extra")
    ap.add_argument("--no-cached", action="store_true", help="skip the committee-key-cache extra")
    ap.add_argument("--large-committee", type=int, default=100000, help="validators in the cached large-QC extra")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded-committee extra (N > 1: RCCL all-gather)")
    ap.add_argument("--no-events", action="store_true", help="do not record per-kernel HIP events (diagnostic)")
    ap.add_argument("--event-stride", type=int, default=16, help="record HIP events on every k-th call per context")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline + cpu_baseline only")
    ap.add_argument("--no-pmc", action="store_true", help="skip the three rocprofv3 counter passes (traffic, SQ counters)")
    ap.add_argument("--detail-file", default=os.environ.get("HS_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json")),
                    help="where the verbose extras go (the line carries their headline scalars in `config` and this path)")
    ap.add_argument("--partitions", type=int, default=None,
                    help="CU-masked streams: split the chip into this many slices (0 = plain streams; "
                         "default 8 (32 CUs each, two stream contexts per slice) for batches up to 4,096 signatures, 0 above)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become one (same ranks the driver's torch.distributed.run line starts)
        cmd = launcher_command(args.gpus, sys.argv[1:])
        print("bench: launching %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
        os.execv(cmd[0], cmd)
    if args.partitions is None:
        env = os.environ.get("HS_BENCH_PARTITIONS")
        # 8 slices of 32 CUs (4 on every XCD: profiles/r6_u_cu_mask_probe.md), two stream contexts per slice: measured best for the quad kernels
        # (profiles/r2_partition_sweep.txt: 16 x 16 CUs 43.9 M/s, 8 x 32 CUs with two streams each 51.1 M/s); the one-lane
        # pair over sixteen batches per launch wants wider slices: 4 x 64 CUs 88.6 M/s, 2 x 128 85.3, none 86.6, 8 x 32 lower
        lane_form = args.kernel == "lane" or (args.kernel == "auto" and args.batches_per_launch >= 8 and args.batches_per_launch * args.batch >= 8192)
        args.partitions = int(env) if env is not None else ((4 if lane_form else 8) if args.batch <= 4096 else 0)

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    # `multi_process` (four node PROCESSES on the GPU, BASELINE configs[0]'s topology) runs FIRST, before this process touches
    # the GPU: `fab local` has no fifth GPU process, and a bench that already holds ~20 hardware queues of its own beside the
    # tenants pushed the driver's runlist into oversubscription - the tenants were then time-sliced (round 3's 39-46 M
    # verifies/s and the 2.8x spread between identical tenants; tools/queue_probe.py, profiles/r4_queue_probe.md)
    # measurement aid: HS_BENCH_ONLY="qc_verify_latency,tc_verify_latency" runs the headline and only the named extras
    only = set(x for x in os.environ.get("HS_BENCH_ONLY", "").split(",") if x)
    if only:
        args.no_cached = args.no_cpu = True
        args.no_pmc = "pmc" not in only
    mp_early = None
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not (args.no_extras or args.no_qc) and (not only or "multi_process" in only):
        try:
            mp_early = multi_process(args.batch)
            mp_early["ran"] = "before this process initialised the GPU (no fifth GPU process beside the tenants)"
            # the same four tenants submitting bursts of SIXTEEN (4 threads x 32 tickets each): what a node that batches its vote
            # ingest (hotstuff_amd/ingest.py, consensus/src/core.rs:203-225) hands over at once
            mp_early["rows_bursts_of_16"] = multi_process(args.batch, procs=(4,), shape=(4, 32, 16), tenancy_modes=(True,))["rows"]
        except Exception as e:        # extras never cost the headline line
            mp_early = {"error": "%s: %s" % (type(e).__name__, e)}
    # stdout carries exactly one line: rank 0's JSON.  Everything else any rank or library writes to
    # fd 1 (RCCL prints an info banner there through C stdio) is routed to stderr until the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import synth
    from hotstuff_amd import _lib, sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench: --gpus %d but WORLD_SIZE=%d: run `python bench.py --gpus N` (it launches its own ranks) or "
              "torch.distributed.run --nproc-per-node N" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    print("bench: rank %d/%d up, %d GPU(s) visible" % (rank, world, ndev), file=sys.stderr, flush=True)
    if ndev == 0:
        print("bench: rank %d: no GPU visible - the engine has no CPU path" % rank, file=sys.stderr, flush=True)
        if world > 1:
            time.sleep(1.0)        # let the peers print their own line before the launcher tears them down with this rank
        sys.exit(3)
    # fewer GPUs than ranks (a 1-GPU rehearsal of `--gpus N`): ranks share devices, and since RCCL refuses two ranks on
    # one device the control plane (barrier, max over ranks, the bitmaps' all-gather) runs over gloo; the line says so
    virtual = ndev < world
    device_index = local % ndev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    exchange = world > 1 or os.environ.get("HS_BENCH_FORCE_EXCHANGE") == "1"
    backend = "gloo" if virtual else "nccl"
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        import datetime
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    _lib.init(device_index, args.batch)

    n = args.batch
    shared = args.shared
    mode = _lib.BATCH if shared else _lib.STRICT
    stride = 0 if shared else 32
    B = max(1, args.batches_per_step)
    ns = max(1, min(args.streams, B))
    # D distinct synthetic batches per GPU (default: a whole step's worth).  Rank 0's first ns - one per stream context, also what
    # the extras use - are OpenSSL-signed fixtures (cached on disk, independent of the engine under test); every other batch is signed on the GPU
    # (hs_ed25519_sign_batch, bit-identical to OpenSSL by tests/test_gpu_wire_and_sign.py) - milliseconds instead of the
    # ~10 s of host signing per rank, which would otherwise stagger the ranks' arrival at the first barrier
    def gpu_signed_batch(first):
        seeds = b"".join(synth.key_seed(first + i) for i in range(n))
        pks = _lib.public_keys(seeds, n)
        dgs = synth.shared_digest(1) if shared else b"".join(synth.msg_digest(first + i) for i in range(n))
        sgs = _lib.sign_batch(b"".join(seeds[32 * i:32 * i + 32] + pks[32 * i:32 * i + 32] for i in range(n)), dgs, n, shared=shared)
        return dgs, pks, sgs
    D = max(ns, args.distinct_batches if args.distinct_batches > 0 else min(B, 4096))       # distinct batches per GPU; the first ns are the stream contexts' own
    if rank == 0:
        batches = [synth.make_batch(n, shared=shared, first=j * n, cache_dir="/tmp/hs_synth") for j in range(ns)]
    else:
        batches = [gpu_signed_batch((rank * ns + j) * n) for j in range(ns)]
    more = [gpu_signed_batch((world * ns + rank * (D - ns) + j) * n) for j in range(D - ns)]      # `batches` stays the contexts' own (the extras below use it)
    dg, pk, sg = batches[0]
    t = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    d_in = [tuple(t(x) for x in bt) for bt in batches + more]
    ctxs = None
    if args.partitions:
        try:
            ctxs = [_lib.DeviceContext(n, partition=(j % args.partitions, args.partitions)) for j in range(ns)]
            streams = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
        except _lib.EngineError as e:       # CU masking unavailable: plain streams
            print("bench: CU-masked streams unavailable (%s); using plain streams" % e, file=sys.stderr)
            ctxs, args.partitions = None, 0
    if ctxs is None:
        streams = [torch.cuda.Stream(device=dev) for _ in range(ns)]
        ctxs = [_lib.DeviceContext(n) for _ in range(ns)]
    for c in ctxs:
        c.set_kernel({"auto": _lib.KERNEL_AUTO, "fused": _lib.KERNEL_FUSED, "pairs": _lib.KERNEL_FUSED_PAIRS, "lane": _lib.KERNEL_LANE}[args.kernel])
    bm_bytes = 8 * ((n + 63) // 64)
    K = max(1, min(16, args.batches_per_launch))
    B = (B + K - 1) // K * K               # whole launches
    bitmaps = [torch.zeros(bm_bytes, dtype=torch.uint8, device=dev) for _ in range(ns * K)]
    nrej = torch.zeros(ns * K, dtype=torch.int32, device=dev)

    # raw addresses once, outside the timed loop (the step itself is one C-ABI call)
    p_in = [tuple(x.data_ptr() for x in d) for d in d_in]
    if args.inputs == "pinned":
        pinned_in = [tuple(_lib.PinnedBuffer(x) for x in bt) for bt in batches + more]
        p_in = [tuple(x.addr for x in d) for d in pinned_in]
    p_dg, p_pk, p_sg = p_in[0]
    p_bm = [b.data_ptr() for b in bitmaps]
    p_nr = [nrej[j:].data_ptr() for j in range(ns * K)]
    p_st = [s.cuda_stream for s in streams]
    # context j verifies K independent batches per launch: its own tuples and its neighbours' (K distinct batches,
    # each with its own bitmap and reject counter)
    # (with --distinct-batches D > streams: context j's v-th launch takes batches v*K + j ... v*K + j + K - 1 of the D, so a batch
    # comes round again only after D / K launches of that context)
    V = max(1, D // K) if D > ns else 1
    lists = [[_lib.batch_array([(p_in[(v * K + j + k) % D][0], stride, p_in[(v * K + j + k) % D][1], p_in[(v * K + j + k) % D][2], n, mode,
                                 p_bm[j * K + k], p_nr[j * K + k]) for k in range(K)]) for v in range(V)] for j in range(ns)]
    if K == 1:
        fn = _lib.lib().hs_ed25519_verify_device
        calls = [(ctxs[j]._h, p_in[j][0], stride, p_in[j][1], p_in[j][2], n, mode, p_bm[j], p_nr[j], p_st[j]) for j in range(ns)]
    else:
        fn = _lib.lib().hs_ed25519_verify_device_batches
        calls = [[(ctxs[j]._h, ctypes.cast(lists[j][v], ctypes.c_void_p), K, p_st[j]) for v in range(V)] for j in range(ns)]
    launches_per_step = B // K
    seq = [0]

    def step(i):
        """one pass over the step's input set: B independent 1,024-tuple batches, K per launch, round-robin over the contexts"""
        for b in range(launches_per_step):
            c = calls[b % ns]
            if K > 1:
                c = c[(seq[0] // ns) % V]
                seq[0] += 1
            if fn(*c) != 0:
                raise RuntimeError("verify launch failed: " + _lib.last_error())

    def fence():
        torch.cuda.synchronize(dev)
        if exchange:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    # HIP events attached to the kernel dispatches themselves (hipExtLaunchKernel start/stop), on the
    # launch streams, for every `--event-stride`-th call of each context (default: every call)
    per_ctx = (args.steps * launches_per_step + ns - 1) // ns
    for c in ctxs:
        c.set_profiling(0 if args.no_events else max(1, per_ctx // args.event_stride + 1), every=args.event_stride)
    nrej.zero_()
    for bmp in bitmaps:          # so that the all-ones assertion below proves the TIMED launches wrote them, not the warm-up
        bmp.zero_()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    enqueue_s = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    rank_info = None
    if exchange:
        elapsed_local = elapsed
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # every rank's own elapsed time and its GPU's init-time kernel pick, gathered: the record then shows a lagging GPU, and a
        # completed all-gather over the control plane is the proof of how many ranks it really spans
        try:
            pick = {"k_verify_fused8": 0.0, "k_verify_fused8h": 1.0}.get(_lib.config_dump()["devices"][0].get("oct_pick"), -1.0)
        except Exception:      # noqa: BLE001
            pick = -2.0
        mine = torch.tensor([elapsed_local, pick, float(device_index)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_info = [[float(v) for v in x.tolist()] for x in allr]

    # correctness of what was timed: every synthetic signature is valid
    assert int(nrej.sum().item()) == 0, "engine rejected valid signatures during the timed region"
    ran = min(ns, launches_per_step)                     # contexts that were actually launched on (a short step does not reach all sixteen)
    for bmp in bitmaps[: ran * K]:
        assert bool((bmp.view(torch.int64)[: n // 64] == -1).all()), "bitmap of a timed batch is not all ones"

    prep_ms = ver_ms = 0.0
    launches = 0
    for c in ctxs:
        a, b, k = c.kernel_times()
        prep_ms += a
        ver_ms += b
        launches += k
    avg_verify_ms = ver_ms / max(1, launches)
    avg_prepare_ms = prep_ms / max(1, launches)

    # ---- extra (not the headline): the same workload with the signers' keys registered as a
    # committee key set (hs_ed25519_keyset_create) - the production path for QC / TC verification,
    # where every key is a committee member for the whole epoch.
    def run_cached():
        t0 = time.perf_counter()
        # the committee = the signers of all ns batches; batch j's keys are entries [j n, (j+1) n)
        ks = _lib.KeySet([bt[1][32 * i:32 * i + 32] for bt in batches for i in range(n)])
        build_ms = (time.perf_counter() - t0) * 1e3
        d_ki = torch.arange(n * ns, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        KC = max(1, min(16, args.cached_batches_per_launch))
        # context j verifies KC independent batches per launch (its own and its neighbours'), each with its own bitmap
        c_bm = [torch.zeros(bm_bytes, dtype=torch.uint8, device=dev) for _ in range(ns * KC)]
        c_nr = torch.zeros(ns * KC, dtype=torch.int32, device=dev)
        # The committee's keys are fixed, its certificates are not: variant v of batch j = the same n signers over OTHER messages,
        # re-signed on the GPU - other hashes, other windows, other lines of the 8.7 GB of key tables in every launch (a repeated
        # batch would find its 25 MB of lines per launch in the last-level cache: VC = 1 is that regime, for comparison)
        VC = max(1, args.cached_variants)
        first_of = lambda j: ((0 if rank == 0 else rank * ns) + j) * n          # key_seed index of batch j's first signer (as `batches` was built)
        var = [[(p_in[j][0], p_in[j][2])] for j in range(ns)]                   # [batch][variant] -> (digests ptr, sigs ptr); variant 0 = the fixture
        keep = []
        for j in range(ns):
            seeds = b"".join(synth.key_seed(first_of(j) + i) for i in range(n))
            secrets = b"".join(seeds[32 * i:32 * i + 32] + batches[j][1][32 * i:32 * i + 32] for i in range(n))
            for v in range(1, VC):
                dgs = synth.shared_digest(1000 + v) if shared else b"".join(synth.msg_digest(900_000_000 + (v * ns + j) * n + i) for i in range(n))
                sgs = _lib.sign_batch(secrets, dgs, n, shared=shared)
                t_d, t_s = t(dgs), t(sgs)
                keep.append((t_d, t_s))
                var[j].append((t_d.data_ptr(), t_s.data_ptr()))
        clists = [[_lib.batch_array([(var[(j + k) % ns][v][0], stride, d_ki[((j + k) % ns) * n:].data_ptr(), var[(j + k) % ns][v][1], n, mode,
                                      c_bm[j * KC + k].data_ptr(), c_nr[j * KC + k:].data_ptr()) for k in range(KC)]) for v in range(VC)] for j in range(ns)]
        cfn = _lib.lib().hs_ed25519_verify_device_batches_cached
        ccalls = [[(ctxs[j]._h, ks.handle, ctypes.cast(clists[j][v], ctypes.c_void_p), KC, p_st[j]) for v in range(VC)] for j in range(ns)]
        cseq = [0]
        claunches = (B + KC - 1) // KC
        for c in ctxs:
            c.set_profiling(max(1, (args.steps * claunches + ns - 1) // ns // args.event_stride + 1), every=args.event_stride)

        def cstep(i):
            for b in range(claunches):
                if cfn(*ccalls[b % ns][(cseq[0] // ns) % VC]) != 0:
                    raise RuntimeError("hs_ed25519_verify_device_batches_cached failed: " + _lib.last_error())
                cseq[0] += 1

        for i in range(args.warmup):
            cstep(i)
        c_nr.zero_()
        for bmp in c_bm:
            bmp.zero_()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            cstep(i)
        fence()
        c_elapsed = time.perf_counter() - t0
        assert int(c_nr.sum().item()) == 0, "cached path rejected valid signatures"
        for bmp in c_bm:
            assert bool((bmp.view(torch.int64)[: n // 64] == -1).all()), "bitmap of a timed key-cache batch is not all ones"
        kt = [c.kernel_times() for c in ctxs]                   # (chain kernel, finish kernel, launches) per context
        k_ms = sum(a + b for a, b, _ in kt) / max(1, sum(k for _, _, k in kt))
        k_chain_ms = sum(a for a, _, _ in kt) / max(1, sum(k for _, _, k in kt))
        cached = {"value": round(n * claunches * KC * args.steps / c_elapsed, 1), "unit": "verifies/s", "batches_per_launch": KC,
                  "ms_per_step": round(c_elapsed / args.steps * 1e3, 4), "kernel": "k_verify_cached_lane2r_batches",
                  "avg_kernel_ms": round(k_ms, 4), "avg_chain_kernel_ms": round(k_chain_ms, 4), "keys": n * ns, "keyset_hbm_bytes": ks.hbm_bytes(),
                  "keyset_build_ms": round(build_ms, 2), "message_variants_per_batch": VC,
                  "table_bytes_read_per_verify": (32 + 16) * 128,      # 32 lines of the key's table + 16 of the base point's 16-bit table
                  "note": "keys pre-registered (static committee); not comparable with `value`, which is the cold path"}
        ks.close()
        # large committee, one launch (BASELINE configs[3] on ONE GPU): N = 100,000 validators (66 GB of
        # tables - sized for 288 GB of HBM), QC of 66,667 votes
        # by distinct keys -> the two-lanes-per-signature cached kernel; its table gather is the one
        # place on this path where HBM bandwidth is a meaningful roofline
        try:
            Nv = args.large_committee
            nl = 2 * Nv // 3 + 1
            ldg, lpk, lsg = synth.make_batch(nl, shared=True, cache_dir="/tmp/hs_synth")
            lks = _lib.KeySet([lpk[32 * i:32 * i + 32] for i in range(nl)]
                              + [synth.key_seed(300_000_000 + i) for i in range(Nv - nl)])
            l_dg, l_sg = t(ldg), t(lsg)
            l_ki = torch.arange(nl, dtype=torch.int32, device=dev)
            l_bm = torch.zeros(8 * ((nl + 63) // 64), dtype=torch.uint8, device=dev)
            l_nr = torch.zeros(1, dtype=torch.int32, device=dev)
            lctx = _lib.DeviceContext(nl)
            lst = torch.cuda.ExternalStream(lctx.stream, device=dev)
            torch.cuda.synchronize(dev)
            reps = 30
            lctx.set_profiling(reps)
            for i in range(reps + 3):
                if i == 3:
                    lst.synchronize(); l_nr.zero_(); torch.cuda.synchronize(dev); t0 = time.perf_counter()
                lctx.verify_cached(lks, l_dg.data_ptr(), 0, l_ki.data_ptr(), l_sg.data_ptr(), nl, _lib.BATCH,
                                   l_bm.data_ptr(), l_nr.data_ptr(), lctx.stream)
            lst.synchronize()
            l_el = time.perf_counter() - t0
            assert int(l_nr.item()) == 0
            lk_ms = (lctx.kernel_times()[0] + lctx.kernel_times()[1]) / max(1, lctx.kernel_times()[2])
            # the same QC with four calls in flight on contexts marked concurrent: the two-launch form (chains + batched finish)
            four = None
            try:
                cx = [_lib.DeviceContext(nl) for _ in range(4)]
                cbm = [torch.zeros(8 * ((nl + 63) // 64), dtype=torch.uint8, device=dev) for _ in range(4)]
                cst = [torch.cuda.ExternalStream(c.stream, device=dev) for c in cx]
                for c in cx:
                    c.set_concurrent(True)
                torch.cuda.synchronize(dev)
                for i in range(3 + 12):
                    if i == 3:
                        for s_ in cst:
                            s_.synchronize()
                        t0 = time.perf_counter()
                    for j, c in enumerate(cx):
                        c.verify_cached(lks, l_dg.data_ptr(), 0, l_ki.data_ptr(), l_sg.data_ptr(), nl, _lib.BATCH, cbm[j].data_ptr(), 0, c.stream)
                for s_ in cst:
                    s_.synchronize()
                four = round(12 * 4 * nl / (time.perf_counter() - t0), 1)
                assert all(bool((b_.view(torch.int64)[: nl // 64] == -1).all()) for b_ in cbm), "a concurrent key-cache QC was not all ones"
                for c in cx:
                    c.close()
            except Exception as e:      # noqa: BLE001
                four = {"error": "%s: %s" % (type(e).__name__, e)}
            cached["large_committee"] = {
                "n_validators": Nv, "signatures": nl, "keyset_hbm_bytes": lks.hbm_bytes(),
                "kernel": _lib.lib().hs_ed25519_cached_kernel_for(nl).decode(), "avg_kernel_ms": round(lk_ms, 4),
                "value": round(nl * reps / l_el, 1), "unit": "verifies/s",
                "four_in_flight_two_launch_form_signatures_per_s": four}
            lctx.close(); lks.close()
        except Exception as e:        # never let the extra break the headline
            cached["large_committee"] = {"error": str(e)}
        return cached

    if args.no_extras:
        args.no_cached = args.no_qc = args.no_sharded = True
    cached = None
    if world == 1 and not args.no_cached:
        try:        # an extra must never cost the headline line
            cached = run_cached()
        except Exception as e:
            cached = {"error": "%s: %s" % (type(e).__name__, e)}

    # the headline's 16 stream contexts are done: release their streams before the extras create their own
    # (HIP multiplexes streams onto 16 hardware queues; leftover streams would share queues with the extras')
    kernel_name = (_lib.lib().hs_ed25519_batches_kernel_for(ctxs[0]._h, K).decode() if (K > 1 and n <= 32768)
                   else _lib.lib().hs_ed25519_dominant_kernel_for(n).decode())
    for c in ctxs:
        c.close()
    # Counters of the kernels this line quotes, measured in THIS run when the default workload is timed: three rocprofv3 passes
    # over solo launches in a subprocess (pmc_passes).  No committed file of an earlier round stands in when it fails: the
    # fields are then null and say why.
    pmc = None
    if rank == 0 and world == 1 and not args.no_extras and not args.no_pmc and kernel_name == "k_verify_batches" and n * K == 16384:
        try:
            pmc = pmc_passes()
        except Exception as e:
            pmc = {"error": "%s: %s" % (type(e).__name__, e)}
    traffic = traffic_src = None
    head_pmc = pmc_pick(pmc, kernel_name)
    prep_pmc = pmc_pick(pmc, "k_prepare_batches")
    if head_pmc and "bytes" in head_pmc:
        traffic = head_pmc["bytes"]
        traffic_src = "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, solo launches, x %.3f / x %.2f (pmc_calibration.json)" % (
            pmc["calibration"]["fetch_factor"], pmc["calibration"]["write_factor"])
    elif pmc and "error" in pmc:
        traffic_src = "unmeasured: " + pmc["error"][:90]
    e2e = None
    if world == 1 and not args.no_extras and (not only or "end_to_end" in only):
        try:
            e2e = end_to_end(_lib, batches, n, stride, mode)
        except Exception as e:
            e2e = {"error": "%s: %s" % (type(e).__name__, e)}

    out = detail = None
    if rank == 0:
        total = world * n * B * args.steps
        value = total / elapsed
        per_verify = ALGO_BYTES_SHARED if shared else ALGO_BYTES_MULTI
        algo = per_verify * n * K       # one launch of the dominant kernel covers K batches
        achieved = algo / (avg_verify_ms * 1e-3) / 1e9 if avg_verify_ms > 0 else 0.0
        detail = {}                      # everything verbose goes to the side file; the line keeps scalars the driver's record holds
        out = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "verifies/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 limbs (radix 2^25.5), u64 accumulate",
            "data": "synthetic",
            "config": {
                # (<= 120 characters; says which rate `value` is)
                "workload": ("QC shape: %d-sig batches, batch law; value = HBM-resident rate; value_host_to_host = host buffers in, verdict out" % n
                             if shared else
                             "configs[1]: %d-tuple batches, strict; value = HBM-resident rate; value_host_to_host = host buffers in, verdict out" % n),
                "workload_detail": "%d distinct batches per GPU, %d per launch, %d CU-masked streams" % (D if K > 1 else ns, K, ns),
                "signatures_per_batch": n, "batches_per_step_per_gpu": B, "verifications_per_step_per_gpu": B * n,
                "distinct_tuples_per_gpu": (D if K > 1 else ns) * n, "streams": ns, "cu_partitions": args.partitions,
                "timed_region_s": round(elapsed, 4), "batches_per_launch": K, "launches_in_timed_region_per_gpu": launches_per_step * args.steps,
                "entry_point": ("hs_ed25519_verify_device_batches" if K > 1 else "hs_ed25519_verify_device") + (" (inputs in PINNED HOST memory: not of record)" if args.inputs == "pinned" else ""),
                "exchange": "none on this workload (independent batches per rank); sharded_committee / sharded_c_abi have it" if exchange else "none (1 GPU)",
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name,
                "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 8),
                # `frac` is per launch of the dominant kernel (the contract's definition) while ~16 launches overlap on the
                # chip; chip_frac = whole-GPU algorithmic bytes per second (value x 128 B) over the same peak
                "chip_frac": round(value / world * per_verify / 1e9 / HBM_PEAK_GBS, 8),
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_over_algorithmic": round(traffic / algo, 2) if traffic else None,
                "traffic_k_prepare_batches": prep_pmc.get("bytes") if prep_pmc else None,
                "algorithmic_bytes_per_launch": algo, "algorithmic_bytes_per_verification": per_verify,
                "avg_kernel_ms": round(avg_verify_ms, 4), "avg_k_prepare_ms": round(avg_prepare_ms, 4) if avg_prepare_ms > 0 else None,
                "launches_timed": launches, "event_stride": args.event_stride,
                "timing": "dispatch timestamps (hipExtLaunchKernel events on the launch stream) of sampled launches in the timed region",
                "note": "integer-VALU bound (SURVEY 8d): HBM fraction tiny by construction; config.int_roofline_* are what binds",
            },
            "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
        }
        cfg = out["config"]
        kname = {"k_verify_fused_batches": "k_verify_fused3", "k_verify_fused_pairs_batches": "k_verify_fused_pairs"}.get(kernel_name, kernel_name)
        if kname in MAD_PER_VERIFY:
            mad_s = value / world * MAD_PER_VERIFY[kname]
            detail["int_roofline"] = {"bound": "valu-int-mul", "achieved": round(mad_s, 1), "peak": MAD_PEAK_PER_S,
                                      "unit": "v_mad_u64_u32 lane-ops/s (per GPU)", "frac": round(mad_s / MAD_PEAK_PER_S, 4),
                                      "mad_per_verify": MAD_PER_VERIFY[kname]}
            cfg["int_roofline_mad_frac"] = round(mad_s / MAD_PEAK_PER_S, 4)
            # total VALU wave-instructions per verification: measured in this run (SQ_INSTS_VALU of the solo pair) when the
            # counter pass ran, else the committed round-4 figure for the same kernels
            wi = wi_src = None
            if head_pmc and prep_pmc and "SQ_INSTS_VALU" in head_pmc and "SQ_INSTS_VALU" in prep_pmc:
                wi, wi_src = (head_pmc["SQ_INSTS_VALU"] + prep_pmc["SQ_INSTS_VALU"]) / (n * K), "SQ_INSTS_VALU of the solo launch pair, this run"
            elif kname in VALU_WAVE_INSTS_PER_VERIFY:
                wi, wi_src = VALU_WAVE_INSTS_PER_VERIFY[kname], "profiles/r4_p_pmc_sq_headline.json (this run's counter pass did not run)"
            if wi:
                wi_s = value / world * wi
                cap = SIMDS * SCLK_HZ / CYCLES_PER_WAVE_INST_MIX
                cfg["int_roofline_issue_frac"] = round(wi_s / cap, 4)
                cfg["valu_wave_insts_per_verify"] = round(wi, 1)
                detail["int_roofline"].update({
                    "issue_frac": round(wi_s / cap, 4), "valu_wave_insts_per_verify": round(wi, 1), "source": wi_src,
                    "valu_wave_insts_per_s": round(wi_s, 1), "simd_issue_capacity_per_s": round(cap, 1),
                    "issue_note": "all VALU wave-instructions of the launch pair (%.1f %% are v_mad_u64_u32) x value, over %d SIMDs x %.1f GHz / "
                                  "%.2f cycles per wave-instruction of that mix (profiles/r1_ubench_valu.txt): the bound that binds this kernel" % (
                                      100 * MAD_SHARE_OF_VALU, SIMDS, SCLK_HZ / 1e9, CYCLES_PER_WAVE_INST_MIX)})
        if e2e is not None:
            detail["end_to_end"] = e2e
            for r_ in e2e.get("by_threads", []):
                shape = (r_["host_threads"], r_["in_flight_per_thread"], r_["batches_per_launch"])
                if shape == (16, 8, 8):
                    cfg["e2e_16_threads_x_8_tickets_verifies_s"] = r_["value"]
                if shape == (1, 1, 1):
                    cfg["one_sync_caller_1024_verifies_per_s"] = r_["value"]
                    cfg["one_sync_caller_1024_call_p50_ms"] = r_["call_p50_ms"]
                if shape[0] == 1 and shape[1] > 1 and r_["value"] > cfg.get("e2e_one_thread_tickets_verifies_per_s", 0):
                    cfg["e2e_one_thread_tickets_verifies_per_s"] = r_["value"]      # ONE host thread, ordinary (unpinned) buffers, tickets in bursts
                    cfg["e2e_one_thread_tickets_shape"] = "%d tickets, bursts of %d" % (shape[1], shape[2])
            if "value" in e2e:
                cfg["end_to_end_verifies_per_s_best"] = e2e["value"]
                cfg["end_to_end_best_shape"] = "%d threads x %d tickets, bursts of %d" % (e2e["host_threads"], e2e["in_flight_per_thread"], e2e["batches_per_launch"])
            if "error" in e2e:
                cfg["end_to_end_error"] = e2e["error"][:120]
        if world == 1 and not args.no_cached:
            if isinstance(cached, dict) and "value" in cached:
                cached_rooflines(cached, pmc, n, stride)
                cfg["committee_cached_verifies_per_s"] = cached["value"]
                lc = cached.get("large_committee", {})
                if "roofline" in lc:
                    cfg["cached_qc_66667_kernel_ms"] = lc["avg_kernel_ms"]
                    cfg["cached_qc_66667_hbm_frac"] = lc["roofline"]["frac"]
                    cfg["cached_qc_66667_traffic_over_algorithmic"] = lc["roofline"].get("traffic_over_algorithmic")
                    cfg["cached_qc_66667_table_gather_GBps"] = lc["roofline"].get("design_traffic_GBps")
                if isinstance(lc.get("four_in_flight_two_launch_form_signatures_per_s"), float):
                    cfg["cached_qc_66667_4_in_flight_sigs_per_s"] = lc["four_in_flight_two_launch_form_signatures_per_s"]
            detail["committee_cached"] = cached
        if world == 1 and not args.no_qc:
            # streamed_rounds first: a ring (and its 5 GB key set) built while the process's first extra context of QC size is
            # alive - the wide pool context qc_latency creates - runs ~25 % slower until that context goes (tools/streamed_probe.py,
            # profiles/r3_streamed_probe.txt; the round-2 tree behaves identically): a coupling in the runtime, not the ring's steady state
            for key, fn in (("streamed_rounds", lambda: streamed_rounds(_lib, synth)),
                            ("qc_verify_latency", lambda: qc_latency(_lib, torch, synth, dev)),
                            ("tc_verify_latency", lambda: tc_latency(_lib, torch, dev)),
                            ("callers_either_side", lambda: _lib.callers_bench(10000, 200)),
                            ("rlc_ab", lambda: rlc_ab(_lib, torch, synth, dev)),
                            ("small_batch_crossover", lambda: small_batch_crossover(_lib, synth)),
                            ("multi_process", lambda: mp_early if mp_early is not None else multi_process(n))):
                if only and key not in only:
                    continue
                try:
                    detail[key] = fn()
                except Exception as e:        # extras never cost the headline line
                    detail[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            summarise_extras(cfg, detail, pmc)
        if pmc is not None:
            detail["pmc"] = pmc
        try:
            detail["library_config"] = _lib.config_dump()
        except Exception as e:      # noqa: BLE001
            detail["library_config"] = {"error": str(e)}
        if not args.no_cpu and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(dg, pk, sg, n, stride, mode)
                out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 2)
                # the CPU twins of the two latency rows (the reference's verify_batch and TC loop, oracle ports), beside the rate
                for src, dst in (("cpu_qc_verify_batch_p50_ms_one_core", "qc_verify_batch_p50_ms_one_core"),
                                 ("cpu_qc_verify_batch_p50_ms_all_cores", "qc_verify_batch_p50_ms_all_cores"),
                                 ("cpu_tc_verify_p50_ms_one_core", "tc_verify_p50_ms_one_core"), ("cpu_tc_verify_p50_ms_all_cores", "tc_verify_p50_ms_all_cores")):
                    if src in cfg:
                        out["cpu_baseline"][dst] = cfg[src]
                if "vote_ingest_qc_ms" in cfg and out["cpu_baseline"].get("one_core"):      # the reference's vote path: q single strict verifications
                    cfg["cpu_vote_ingest_ms_derived"] = round(1e3 * cfg.get("qc_signatures", 6667) / out["cpu_baseline"]["one_core"], 1)
            except Exception as e:
                out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # ---- multi-GPU extras, AFTER the headline is complete.  Each runs under a deadline on rank 0: if a collective or
    # a device hangs there, the finished headline is printed with the extra marked as timed out (line_watchdog).
    dog = line_watchdog(real_stdout)
    if rank == 0:
        out["config"].update({"ranks_world": world, "ranks_gpus_visible": ndev, "ranks_control_plane": backend if exchange else "none",
                              "ranks_virtual": virtual,
                              "ranks_launched_by": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct"})
        if rank_info:
            per_rank = [n * B * args.steps / max(r_[0], 1e-9) for r_ in rank_info]
            out["config"].update({
                ("rccl_ranks_seen" if backend == "nccl" else "gloo_ranks_seen"): len(rank_info),      # ranks whose all_gather contribution arrived
                "rank_value_min": round(min(per_rank), 1), "rank_value_max": round(max(per_rank), 1),
                "rank_value_max_over_min": round(max(per_rank) / max(min(per_rank), 1.0), 4),
                "rank_autotune_picks": ",".join({0: "fused8", 1: "fused8h"}.get(int(r_[1]), "unmeasured") for r_ in rank_info)[:120],
                "rank_devices": ",".join(str(int(r_[2])) for r_ in rank_info)[:120]})

    def guarded(key, seconds, fn, every_rank):
        """run extra `key` (on every rank, or on rank 0 while the others wait at the next barrier) under rank 0's deadline"""
        if rank == 0:
            def late():
                o = dict(out)
                o["config"] = dict(out["config"])
                o["config"][(key + "_error")[:KEY_MAX]] = "did not finish within %d s; line printed by the watchdog, ranks abandoned" % seconds
                return json.dumps(finalize_line(o))
            dog.arm(late, seconds)
        res = None
        try:
            if every_rank or rank == 0:
                res = fn()
        except Exception as e:
            res = {"error": "%s: %s" % (type(e).__name__, e)}
            # a rank that fails alone leaves its peers inside the collective until rank 0's deadline: say why, at once
            print("bench: rank %d: extra %s failed: %s" % (rank, key, res["error"]), file=sys.stderr, flush=True)
        if rank == 0:
            dog.disarm()
            detail[key] = res
            out[key] = res                # the N > 1 line has no other extras (they are gated on one rank): these two stay in it, whole
            cfgm = out["config"]
            if isinstance(res, dict) and "error" in res:
                cfgm[(key + "_error")[:KEY_MAX]] = str(res["error"])[:120]
            elif isinstance(res, dict):
                for form in ("cold", "key_cache"):          # sharded_committee: the QC of the large committee cut over the ranks
                    if isinstance(res.get(form), dict) and "p50_ms" in res[form]:
                        cfgm["%s_p50_ms_%s" % (key, form)] = res[form]["p50_ms"]
                        cfgm["%s_verifies_s_%s" % (key, form)] = res[form]["verifies_per_s"]
                if "signatures" in res:
                    cfgm[key + "_signatures"] = res["signatures"]
                for k2, v2 in res.items():                   # sharded_c_abi: flat scalar rows of the one-process form
                    if isinstance(v2, (int, float, str)) and k2 not in ("signatures",) and len(cfgm) < 120:
                        cfgm[("%s_%s" % (key, k2))[:40]] = v2[:120] if isinstance(v2, str) else v2

    # the workload that has an exchange step: one very large committee sharded over the GPUs, one process per GPU,
    # all-gather of the per-shard accept bitmaps (RCCL over xGMI; gloo when ranks share a device)
    if exchange and not args.no_sharded:
        guarded("sharded_committee", 150,
                lambda: sharded_committee(_lib, torch, dist, synth, sharding, dev, rank, world, validators=args.large_committee), True)
    # the same sharded workload from ONE process through the C ABI (rank 0's subprocess drives every GPU; the other
    # ranks have released their contexts and wait at the barrier below)
    if not args.no_sharded and (exchange or os.environ.get("HS_BENCH_SHARDED_C_ABI") == "1"):
        if exchange:
            dist.barrier()
        guarded("sharded_c_abi", 170, lambda: sharded_c_abi(world, ndev, validators=args.large_committee), False)
    if rank == 0:
        # the verbose extras go to a side file (gpurun_out/ travels back from a gpurun box); the line says where
        try:
            os.makedirs(os.path.dirname(args.detail_file), exist_ok=True)
            with open(args.detail_file, "w") as f:
                json.dump({"line": out, "detail": detail}, f, indent=1)
            out["detail_file"] = os.path.relpath(args.detail_file, ROOT)
            out["detail_keys"] = sorted(detail)
        except Exception as e:      # noqa: BLE001
            out["detail_file"] = "not written: %s" % e
    line = json.dumps(finalize_line(out)) if rank == 0 else None
    if exchange:
        if rank == 0:
            dog.arm(lambda: line, 60)
        dist.barrier()
        dist.destroy_process_group()
        dog.disarm()
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)     # C stdio buffers (still pointing at stderr)
    except Exception:
        pass
    if line is not None:
        os.write(real_stdout, (line + "\n").encode())


if __name__ == "__main__":
    main()
