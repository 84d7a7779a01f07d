#!/bin/bash
mkdir -p gpurun_out
export HS_SYNTH_CACHE=/tmp/hs_synth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY'
import time, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import synth
from hotstuff_amd import _lib
_lib.init(0, 8192)
for nk in (10000, 16384, 100000, 10000):
    pk = _lib.public_keys(b"".join(synth.key_seed(7_000_000 + i) for i in range(nk)), nk)
    ks0 = [pk[32 * i:32 * i + 32] for i in range(nk)]
    t0 = time.perf_counter(); ks = _lib.KeySet(ks0); dt = time.perf_counter() - t0
    print("keyset of %d keys: %.1f ms, %d bytes" % (nk, dt * 1e3, ks.hbm_bytes()), flush=True)
    ks.close()
PY
