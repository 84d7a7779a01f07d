"""Several node PROCESSES sharing one GPU ("must tolerate multi-process sharing of a device"; the
reference's `fab local` starts four node processes on one host, /root/reference/benchmark/benchmark/local.py:94-104).
Four processes, each with its own engine instance on device 0, verify mixed corrupted workloads - small strict batches
(synchronous and in ticket bursts), a QC-sized batch cold and through a registered key set, caller-pinned buffers -
at the same time; every bitmap must equal the C oracle's, with and without the tenancy hint."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(procs, rounds, tenancy):
    ps = []
    try:
        for i in range(procs):
            env = dict(os.environ)
            env.pop("HS_ED25519_TENANTS", None); env.pop("HS_ED25519_TENANT_INDEX", None)
            if tenancy:
                env["HS_ED25519_TENANTS"], env["HS_ED25519_TENANT_INDEX"] = str(procs), str(i)
            ps.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_parity_worker.py"), str(i), str(rounds)],
                                       stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        for p in ps:
            line = p.stdout.readline()
            assert line.startswith("ready"), "worker did not come up: %r %s" % (line, p.stderr.read()[-2000:] if p.poll() is not None else "")
        for p in ps:
            p.stdin.write("go\n"); p.stdin.flush()
        out = []
        for p in ps:
            o, e = p.communicate(timeout=600)
            assert p.returncode == 0, e[-2000:]
            out.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
        return out
    finally:
        for p in ps:
            if p.poll() is None:
                p.kill()


@pytest.mark.parametrize("tenancy", [False, True])
def test_four_processes_on_one_gpu_agree_with_the_oracle(tenancy):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    res = _run(4, 3, tenancy)
    assert len(res) == 4
    for r in res:
        assert r["checks"] == 3 * (4 + 4 + 1 + 1) and r["mismatches"] == [], r
