"""Seeded random walk over the host-side entry points: random sizes (1 ... 3,000, weighted towards the wavefront /
ballot / group boundaries), digest forms, modes, corruption rates and kinds (bit flips in R, s, the key, the digest;
unknown key indices on the key-cache forms), through a randomly chosen entry point each time - synchronous, ticket
(alone, or as part of a burst that is collected in a random order), key cache, device pointers with a forced kernel or
a forced RLC stage.  Every accept bitmap must equal the oracle's per-signature verdicts (STRICT / BATCH laws; BATCH_RLC
localises with the BATCH law, and a batch it accepts outright contains no signature the BATCH law rejects)."""
import hashlib
import os
import random
import struct

import pytest
import torch

import synth
from conftest import bits, oracle_verify_many
from gpu_helpers import DeviceRunner
from hotstuff_amd import _lib

pytestmark = pytest.mark.gpu
CACHE = "/tmp/hs_synth"
NKEYS = 512
SIZES = [1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 1023, 1024, 1025, 2047, 2048, 3000]


@pytest.fixture(scope="module")
def world():
    assert torch.cuda.is_available()
    _lib.init(0, 8192)
    seeds = [synth.key_seed(i) for i in range(NKEYS)]
    pk = _lib.public_keys(b"".join(seeds), NKEYS)
    keys = [pk[32 * i:32 * i + 32] for i in range(NKEYS)]
    secrets = [seeds[i] + keys[i] for i in range(NKEYS)]
    ks = _lib.KeySet(keys)
    runners = {"auto": DeviceRunner(4096), "lane": DeviceRunner(4096, kernel=_lib.KERNEL_LANE), "fused": DeviceRunner(4096, kernel=_lib.KERNEL_FUSED),
               "rlc": DeviceRunner(4096, rlc_min=1)}
    yield keys, secrets, ks, runners
    for r in runners.values():
        r.ctx.close()
    ks.close()


def _make(rng, keys, secrets, n, shared):
    who = [rng.randrange(NKEYS) for _ in range(n)]
    tag = rng.getrandbits(32)
    dg = hashlib.sha512(b"fuzz" + struct.pack("<I", tag)).digest()[:32] if shared else \
        b"".join(hashlib.sha512(b"fuzz" + struct.pack("<II", tag, i)).digest()[:32] for i in range(n))
    sg = _lib.sign_batch(b"".join(secrets[w] for w in who), dg, n, shared)          # signed on the GPU (bit-identical to OpenSSL's, tested elsewhere)
    return who, bytearray(dg), bytearray(b"".join(keys[w] for w in who)), bytearray(sg)


def _corrupt(rng, dg, pk, sg, n, shared, rate, allow_key):
    for i in range(n):
        if rng.random() >= rate:
            continue
        kind = rng.choice(["R", "s", "key", "dg"] if allow_key else ["R", "s", "dg"])
        bit = 1 << rng.randrange(8)
        if kind == "R":
            sg[64 * i + rng.randrange(32)] ^= bit
        elif kind == "s":
            sg[64 * i + 32 + rng.randrange(32)] ^= bit
        elif kind == "key":
            pk[32 * i + rng.randrange(32)] ^= bit
        elif not shared:
            dg[32 * i + rng.randrange(32)] ^= bit
        else:
            sg[64 * i + rng.randrange(64)] ^= bit


# seed 31: from step 126 the walk holds every wide context as uncollected tickets and keeps calling synchronously (batch signing
# at every step, a large synchronous verification at step 131) - each such call waits out the pool's one-second patience
@pytest.mark.parametrize("seed,steps", [(20260921, 220), (31, 134)])
def test_random_walk_over_the_entry_points(oracle, world, seed, steps):
    keys, secrets, ks, runners = world
    rng = random.Random(int(os.environ.get("HS_FUZZ_SEED", seed)))      # HS_FUZZ_SEED / HS_FUZZ_STEPS: longer one-off walks
    pending = []                                    # tickets of an open burst: (ticket, want, n)
    counts = {}

    def collect_all():
        rng.shuffle(pending)
        while pending:
            t, want, n = pending.pop()
            ok, bm = t.wait()
            assert bits(bm, n) == want and ok == all(want)

    for it in range(int(os.environ.get("HS_FUZZ_STEPS", steps))):
        n = rng.choice(SIZES) if rng.random() < 0.7 else rng.randrange(1, 3001)
        shared = rng.random() < 0.5
        mode = rng.choice([_lib.STRICT, _lib.BATCH, _lib.BATCH_RLC])
        rate = rng.choice([0.0, 0.0, 0.01, 0.05, 0.5])
        api = rng.choice(["sync", "sync_cached", "ticket", "ticket", "ticket_cached", "dev_auto", "dev_lane", "dev_fused", "dev_rlc", "dev_cached"])
        cached = api.endswith("cached")
        who, dg, pk, sg = _make(rng, keys, secrets, n, shared)
        _corrupt(rng, dg, pk, sg, n, shared, rate, allow_key=not cached)
        dg, pk, sg = bytes(dg), bytes(pk), bytes(sg)
        law = _lib.STRICT if mode == _lib.STRICT else _lib.BATCH
        want = bytearray(oracle_verify_many(oracle, dg, 0 if shared else 32, pk, sg, n, law))
        idx = list(who)
        if cached and n > 2 and rng.random() < 0.3:            # an index outside the key set rejects that signature
            j = rng.randrange(n)
            idx[j] = _lib.KeySet.UNKNOWN if rng.random() < 0.5 else NKEYS + rng.randrange(1000)
            want[j] = 0
        want = bytes(want)
        counts[api] = counts.get(api, 0) + 1
        if api == "sync":
            ok, bm = _lib.verify_packed(dg, pk, sg, n, mode, shared, want_bitmap=True)
            assert bits(bm, n) == want and ok == all(want), (it, api, n, mode)
        elif api == "sync_cached":
            ok, bm = ks.verify_packed(dg, struct.pack("<%dI" % n, *idx), sg, n, mode, shared, want_bitmap=True)
            assert bits(bm, n) == want and ok == all(want), (it, api, n, mode)
        elif api in ("ticket", "ticket_cached"):
            more = rng.random() < 0.6
            while True:
                try:
                    t = _lib.Ticket(dg, idx if cached else pk, sg, n, mode=mode, shared=shared, keyset=ks if cached else None, more=more)
                    break
                except _lib.PoolBusy:
                    assert pending
                    collect_all()
            pending.append((t, want, n))
            if not more and rng.random() < 0.5:
                collect_all()
        else:
            r = runners[api[4:]] if api != "dev_cached" else runners["auto"]
            if api == "dev_cached":
                known = [keys[w] if w < NKEYS else b"\xff" * 32 for w in idx]       # an all-ones string is not in the set
                got = r.bits(dg, pk, sg, n, law, shared, keyset=ks, keys=known)
            else:
                got = r.bits(dg, pk, sg, n, mode if api in ("dev_auto", "dev_rlc") else law, shared)
            assert got == want, (it, api, n, mode)
    collect_all()
    if seed == 20260921 and "HS_FUZZ_SEED" not in os.environ and "HS_FUZZ_STEPS" not in os.environ:
        assert len(counts) == 9 and min(counts.values()) >= 8, counts
