"""`python bench.py --gpus N` must run by itself (VERDICT r2 "Next round" #1: it used to assert out unless it had been
started by torch.distributed.run, so the first multi-GPU SCALE run would have recorded nothing).  CPU-only checks: the
launcher command is the driver's own launch line, `--gpus 2` really starts two ranks that come up and reach the device
check (here: no GPU -> they say so and exit 3; the engine has no CPU path), and the watchdog that protects the finished
headline line from a hanging multi-GPU extra prints it."""
import json
import os
import subprocess
import sys
import time

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_launcher_command_is_the_drivers_launch_line():
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29555"]
    assert cmd[10] == os.path.join(ROOT, "bench.py") and cmd[11:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    port = int(bench.launcher_command(2, [])[9])
    assert 1024 < port < 65536


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the launcher on a machine without a GPU")
def test_gpus_2_without_a_launcher_starts_two_ranks_that_reach_the_device_check():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--no-extras"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert "bench: launching 2 ranks" in r.stderr
    # torchrun sends SIGTERM to the surviving rank as soon as the first one exits with code 3 - sometimes before that rank
    # has printed its own line (2 failures in 5 runs): at least one rank must have reached the device check
    import re
    assert re.search(r"bench: rank [01]/2 up", r.stderr), r.stderr[-2000:]
    assert "no GPU visible" in r.stderr and r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]       # no line is better than an invented one


def test_a_mismatched_world_size_is_an_error_message_not_an_assert():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_watchdog_prints_the_finished_line_when_an_extra_hangs(tmp_path):
    prog = tmp_path / "wd.py"
    prog.write_text(
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "dog = bench.line_watchdog(1)\n"
        "dog.arm(lambda: '{\"value\": 1, \"extra\": {\"error\": \"timed out\"}}', 0.3)\n"
        "time.sleep(30)\n" % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=60)
    assert time.time() - t0 < 20 and r.returncode == 0
    assert json.loads(r.stdout.strip())["extra"]["error"] == "timed out"
    prog.write_text(
        "import sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "dog = bench.line_watchdog(1)\n"
        "dog.arm(lambda: 'late', 0.2)\n"
        "dog.disarm()\n"
        "time.sleep(0.6)\n"
        "print('on time')\n" % ROOT)
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, timeout=60)
    assert r.stdout.strip() == "on time"
