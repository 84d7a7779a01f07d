"""BASELINE configs[3] the way the reference's ONE-process node would run it: this process opens `gpus` GPUs
(hs_ed25519_init_devices) and calls hs_ed25519_verify_shared_sharded - one contiguous shard per GPU launched from
one thread, bitmap slices exchanged through the host or an in-process RCCL all-gather.  Prints one JSON object.

    python tools/sharded_c_abi.py <gpus> [validators] [iters] [devices]

`devices` < `gpus` (a rehearsal on fewer GPUs than slots): slot s runs on device s mod devices - the RCCL exchange
then refuses (two slots on one device) and says so, everything else runs.

Run by bench.py (rank 0) in a subprocess with a timeout, so that a fault on hardware this path has never seen
cannot cost the bench line."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


T0 = time.perf_counter()


def note(stage):
    """One dated line per stage on stderr: when the caller's timeout fires, its tail says where the time went."""
    sys.stderr.write("[sharded_c_abi %7.2f s] %s\n" % (time.perf_counter() - T0, stage))
    sys.stderr.flush()


def main():
    dump_after = float(os.environ.get("HS_SHARDED_DUMP_AFTER_S", "0"))
    if dump_after > 0:
        import faulthandler
        faulthandler.dump_traceback_later(dump_after, exit=False)      # every thread's Python stack, once, if still running then
    ngpus = int(sys.argv[1])
    validators = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    import synth
    from hotstuff_amd import _lib
    ndev = int(sys.argv[4]) if len(sys.argv) > 4 else ngpus
    n = 2 * validators // 3 + 1
    note("init_devices %r" % [s % ndev for s in range(ngpus)])
    _lib.init_devices([s % ndev for s in range(ngpus)], 8192)
    note("signing %d signatures on the GPU" % n)
    lib = _lib.lib()
    # the synthetic QC is signed on the GPU (hs_ed25519_public_keys / hs_ed25519_sign_batch): milliseconds instead of seconds
    seeds = b"".join(synth.key_seed(9_000_000 + i) for i in range(n))
    pk = _lib.public_keys(seeds, n)
    secrets = b"".join(seeds[32 * i:32 * i + 32] + pk[32 * i:32 * i + 32] for i in range(n))
    dg = synth.shared_digest(9)
    sg = _lib.sign_batch(secrets, dg, n, shared=True)
    b = [ctypes.create_string_buffer(x, len(x)) for x in (dg, pk, sg)]
    a = [ctypes.addressof(x) for x in b]
    out = {"n_validators": validators, "signatures": n, "gpus": ngpus, "devices": ndev,
           "process": "one process drives every GPU through the C ABI",
           "signatures_per_shard": _lib.shard_plan(n, ngpus, 0)[1]}
    for name, how in (("host_exchange", _lib.EXCHANGE_HOST), ("rccl_all_gather", _lib.EXCHANGE_RCCL)):
        try:
            note("sharded QC, exchange = %s" % name)
            lib.hs_ed25519_set_shard_exchange(how)
            for mode_name, mode in (("per_signature", _lib.BATCH), ("rlc_msm", _lib.BATCH_RLC)):
                times = []
                for i in range(iters + 3):
                    t0 = time.perf_counter()
                    rc = lib.hs_ed25519_verify_shared_sharded(a[0], a[1], a[2], n, mode, None)
                    dt = time.perf_counter() - t0
                    if rc != 1:
                        raise RuntimeError("sharded QC rejected / failed (%d): %s" % (rc, _lib.last_error()))
                    if i >= 3:
                        times.append(dt)
                times.sort()
                out.setdefault(name, {})[mode_name] = {"p50_ms": round(times[len(times) // 2] * 1e3, 4),
                                                       "verifies_per_s": round(n / times[len(times) // 2], 1)}
            if how == _lib.EXCHANGE_RCCL:
                out[name]["rccl_ranks"] = lib.hs_ed25519_shard_rccl_ranks()      # communicators ncclCommInitAll gave this process
        except Exception as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
    # what the first real multi-GPU record has to say about itself, flat (bench.py lifts top-level scalars into `config`): how many
    # ranks the in-process RCCL communicators span, how many exchanges fell back to pinned host memory (and why), each slot's pick
    try:
        out["rccl_ranks"] = lib.hs_ed25519_shard_rccl_ranks()
        out["rccl_fallbacks"] = _lib.shard_rccl_fallbacks()
        if out["rccl_fallbacks"]:
            out["rccl_fallback_reason"] = _lib.last_error()[:120]
        out["autotune_picks"] = ",".join(d.get("oct_pick", "?").replace("k_verify_", "") for d in _lib.config_dump()["devices"])
    except Exception as e:
        out["rccl_ranks_error"] = "%s: %s" % (type(e).__name__, e)
    # the committee registered as a key set on every GPU: whole table per GPU, or each GPU the keys it owns
    try:
        lib.hs_ed25519_set_shard_exchange(_lib.EXCHANGE_HOST)
        import random
        import struct
        # ranks sharing devices share their memory too: a smaller committee keeps the replicated tables within one GPU
        nv = validators if ndev >= ngpus else min(validators, 12000)
        nq = 2 * nv // 3 + 1
        rng = random.Random(7)
        voters = sorted(rng.sample(range(nv), nq))          # a QC = some 2f+1 subset of the committee, in committee order
        committee = [synth.key_seed(400_000_000 + i) for i in range(nv)]      # non-voting members: any 32 bytes that are keys
        for j, v in enumerate(voters):
            committee[v] = pk[32 * j:32 * j + 32]
        vsg = sg[:64 * nq]
        vix = struct.pack("<%dI" % nq, *voters)
        for pname, placement in (("replicate", _lib.KEYS_REPLICATE), ("by_owner", _lib.KEYS_BY_OWNER)):
            try:
                note("key sets on every GPU: %s, %d keys" % (pname, nv))
                t0 = time.perf_counter()
                kss = _lib.ShardedKeySet(committee, placement)
                build_ms = (time.perf_counter() - t0) * 1e3
                times = []
                for i in range(iters + 3):
                    t0 = time.perf_counter()
                    ok, _ = kss.verify_packed(dg, vix, vsg, nq, _lib.BATCH, True)
                    dt = time.perf_counter() - t0
                    if not ok:
                        raise RuntimeError("sharded key-cache QC rejected")
                    if i >= 3:
                        times.append(dt)
                times.sort()
                out.setdefault("key_cache", {"n_validators": nv, "signatures": nq})[pname] = {
                    "p50_ms": round(times[len(times) // 2] * 1e3, 4), "verifies_per_s": round(nq / times[len(times) // 2], 1),
                    "table_bytes_per_gpu": kss.bytes_per_device(0), "build_ms": round(build_ms, 1)}
                kss.close()
            except Exception as e:
                out.setdefault("key_cache", {})[pname] = {"error": "%s: %s" % (type(e).__name__, e)}
    except Exception as e:
        out["key_cache"] = {"error": "%s: %s" % (type(e).__name__, e)}
    # one corrupted signature must be localised across the shards
    try:
        note("localisation check")
        lib.hs_ed25519_set_shard_exchange(_lib.EXCHANGE_HOST)
        victim = n - 5
        s2 = bytearray(sg); s2[64 * victim + 40] ^= 1
        ok, bm = _lib.verify_packed_sharded(dg, pk, bytes(s2), n, _lib.BATCH_RLC, True, want_bitmap=True)
        bad = [i for i in range(n) if not (bm[i >> 3] >> (i & 7)) & 1]
        out["localisation_check"] = "ok" if (not ok and bad == [victim]) else "FAILED: %r" % bad[:5]
    except Exception as e:
        out["localisation_check"] = "%s: %s" % (type(e).__name__, e)
    # small certificates are not cut: host calls by key bytes take the device slots in turn.  1,024-signature batches from
    # host buffers through ticket bursts of two, 16 native threads with 2 x gpus tickets each - the one node process
    # feeding every GPU (host memcpy and PCIe included)
    try:
        bn = 1024
        nb = min(16, n // bn)
        if nb == 0:
            raise ValueError("committee too small for 1,024-signature batches")
        dgs = b"".join(synth.msg_digest(i) for i in range(nb * bn))
        pk2 = pk[:32 * nb * bn]
        sg2 = _lib.sign_batch(secrets[:64 * nb * bn], dgs, nb * bn, shared=False)
        bb = [ctypes.create_string_buffer(x, len(x)) for x in (dgs, pk2, sg2)]
        aa = [ctypes.addressof(x) for x in bb]
        rows = []
        for th, k in ((16, 2), (16, 2 * ngpus), (16, 4 * ngpus)):
            note("small batches over every GPU: %d threads x %d tickets" % (th, k))
            cv, pv, ev = _lib.loadgen(None, aa[0], 32, aa[1], aa[2], bn, nb, _lib.STRICT, th, k, 2, 1.0)
            rows.append({"host_threads": th, "in_flight_per_thread": k, "batches_per_launch": 2, "call_p50_ms": round(pv, 4),
                         "verifies_per_s": round(cv * bn / ev, 1)})
            if ngpus == 1:
                break
        out["small_batches_all_gpus"] = {"signatures_per_call": bn, "api": "hs_ed25519_submit bursts of two + hs_ed25519_ticket_wait", "rows": rows}
    except Exception as e:
        out["small_batches_all_gpus"] = {"error": "%s: %s" % (type(e).__name__, e)}
    note("shutdown")
    _lib.shutdown()
    note("done")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
