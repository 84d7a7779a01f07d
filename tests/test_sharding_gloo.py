"""world_size-2 gloo run of the multi-GPU path (hotstuff_amd/sharding.py) on CPU: the
per-rank verifier is the ORACLE here (this is a test); on the GPU box the same code runs
GpuShardVerifier over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from hotstuff_amd.sharding import shard_bounds


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 63, 64, 65, 1000, 6667, 66667):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            sizes = set()
            for r in range(world):
                lo, hi, S = shard_bounds(n, world, r)
                assert lo == min(n, r * S) and lo <= hi <= n and S % 64 == 0
                seen += hi - lo
                sizes.add(S)
            assert seen == n and len(sizes) == 1
    # BASELINE config 4: N = 100 000 -> 66 667 signatures over 8 GPUs
    assert shard_bounds(66667, 8, 0)[:2] == (0, 8384) and shard_bounds(66667, 8, 7)[:2] == (58688, 66667)


def _worker(rank, world, port, n, bad_idx, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ctypes
    import synth
    from hotstuff_amd.sharding import verify_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = ctypes.CDLL(os.path.join(ROOT, "oracle", "libhs_oracle.so"))
    dg, pk, sg = synth.make_batch(n, shared=True)
    sg = bytearray(sg)
    for b in bad_idx:
        sg[64 * b + 3] ^= 0x10
    sg = bytes(sg)

    def verify_shard(lo, hi, S):
        out = ctypes.create_string_buffer(max(1, hi - lo))
        orc.hs_oracle_verify_many(dg, ctypes.c_size_t(32), ctypes.c_size_t(0), pk[32 * lo:], sg[64 * lo:],
                                  ctypes.c_size_t(hi - lo), 0, out, 1)
        bm = torch.zeros(S // 8, dtype=torch.uint8)
        for j in range(hi - lo):
            if out.raw[j]:
                bm[j >> 3] |= 1 << (j & 7)
        return bm

    ok, bitmap = verify_sharded(verify_shard, n)
    q.put((rank, ok, bytes(bitmap.tolist())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bad", [(150, []), (150, [5, 149]), (64, [63]), (3, [])])
def test_two_rank_gloo_allgather(oracle, n, bad):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bad, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = bytearray((n + 7) // 8)
    for i in range(n):
        if i not in bad:
            expect[i >> 3] |= 1 << (i & 7)
    for rank, ok, bm in res:
        assert ok == (not bad)
        assert bm == bytes(expect), rank        # every rank holds the full bitmap
