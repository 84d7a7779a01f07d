"""One process with the eight-lane cold form pinned by the environment (HS_ED25519_OCT_SPLIT=0: k_verify_fused8, =1: k_verify_fused8h
at every size up to 8,192 - the library's own choice between 4,097 and 8,192 signatures is the form it measured faster on the GPU
at init, so a single box exercises only one of them there): every edge and torsion-fuzz vector in both modes and a ~3 % corrupted 6,667-signature QC, device-resident
through the C ABI.  Prints one JSON object {"kernel": ..., "strict": [...], "batch": [...], "qc": {mode: hex bitmap}}."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import synth
    from gpu_helpers import DeviceRunner
    from hotstuff_amd import _lib
    _lib.init(0, 8192)
    golden = os.path.join(ROOT, "tests", "golden")
    vs = json.load(open(os.path.join(golden, "edge_vectors.json")))["vectors"] + json.load(open(os.path.join(golden, "torsion_fuzz.json")))["vectors"]
    H = bytes.fromhex
    n = len(vs)
    dg, pk, sg = b"".join(H(v["msg"]) for v in vs), b"".join(H(v["pk"]) for v in vs), b"".join(H(v["sig"]) for v in vs)
    out = {"kernel": _lib.lib().hs_ed25519_dominant_kernel_for(6667).decode()}
    r = DeviceRunner(8192)
    out["strict"] = list(r.bits(dg, pk, sg, n, _lib.STRICT, False))
    out["batch"] = list(r.bits(dg, pk, sg, n, _lib.BATCH, False))
    m = 6667
    qdg, qpk, qsg = synth.make_batch(m, shared=True, cache_dir=os.environ.get("HS_SYNTH_CACHE", "/tmp/hs_synth"))
    qdg, qpk, qsg, bad = synth.corrupt(qdg, qpk, qsg, m, True, rate_byte=8)
    out["qc"] = {str(mode): bytes(r.bits(qdg, qpk, qsg, m, mode, True)).hex() for mode in (_lib.BATCH, _lib.STRICT)}
    r.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
