"""`python bench.py --gpus 2` on the GPU box: the launcher starts two ranks, the headline runs on both, and the extras with
an exchange step (`sharded_committee`: one process per GPU + all-gather of the accept bitmaps; `sharded_c_abi`: one process
driving every device slot through the C ABI) come back with numbers, not errors.  On a one-GPU box the two ranks share the
device (gloo control plane; the line says so) - the rehearsal that found, in round 3, that every rank but 0 indexed its
shard's tensors with global indices."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_gpus_2_runs_by_itself_and_its_exchange_extras_work():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batches-per-step", "64", "--large-committee", "3000",
                        "--detail-file", os.path.join(ROOT, "gpurun_out", "test_bench_gpus2_detail.json")],
                       capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    try:        # keep the ranks' stderr where a gpurun visit can pick it up
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "test_bench_gpus2.err"), "w") as f:
            f.write(r.stderr)
    except OSError:
        pass
    tb = "\n".join(l for l in r.stderr.splitlines() if l.startswith(("[rank", "bench:")) or "Error" in l)
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, tb[-4000:])
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["ranks_world"] == 2
    assert j["metric"].startswith("Ed25519 verifies/sec (whole node) + QC-verify p50 latency")      # BASELINE.json's metric, whole
    assert j["config"]["sharded_committee_p50_ms_cold"] > 0 and j["config"]["sharded_committee_p50_ms_key_cache"] > 0     # flat, where the driver's record keeps them
    sc = j["sharded_committee"]
    assert "error" not in sc and sc["ranks"] == 2 and sc["cold"]["p50_ms"] > 0 and sc["key_cache"]["p50_ms"] > 0, sc
    ca = j["sharded_c_abi"]
    assert "error" not in ca and ca["gpus"] == 2 and ca["localisation_check"] == "ok", ca
    assert "error" not in ca["host_exchange"] and "error" not in ca["key_cache"]["replicate"] and "error" not in ca["key_cache"]["by_owner"], ca
    if torch.cuda.device_count() >= 2:
        assert ca["rccl_all_gather"].get("rccl_ranks") == 2, ca
