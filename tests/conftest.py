import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a clean clone has no built artefacts: build the engine (hipcc cross-compiles without a GPU)
    # and the CPU checker before collecting, exactly what __graft_entry__.build() does
    lib = os.path.join(ROOT, "hotstuff_amd", "libhs_ed25519.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build_engine()


def _make(path):
    subprocess.run(["make", "-C", path], check=True, capture_output=True)


S = ctypes.c_size_t


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (test infrastructure)."""
    so = os.path.join(ROOT, "oracle", "libhs_oracle.so")
    if not os.path.exists(so):
        _make(os.path.join(ROOT, "oracle"))
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def emu():
    """The product's device headers compiled for the CPU (tests/emu)."""
    so = os.path.join(ROOT, "tests", "emu", "libhs_emu.so")
    _make(os.path.join(ROOT, "tests", "emu"))      # a failed rebuild fails the tests: never run against a stale emulation
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def openssl_vectors():
    return json.load(open(os.path.join(GOLDEN, "openssl_vectors.json")))


@pytest.fixture(scope="session")
def edge_vectors():
    return json.load(open(os.path.join(GOLDEN, "edge_vectors.json")))["vectors"]


@pytest.fixture(scope="session")
def torsion_fuzz():
    return json.load(open(os.path.join(GOLDEN, "torsion_fuzz.json")))["vectors"]


def oracle_verify_many(orc, digests, stride, pks, sigs, n, mode, threads=None):
    out = ctypes.create_string_buffer(max(1, n))
    if threads is None:         # big batches: use the host's cores (the oracle shards by index)
        threads = 1 if n < 2000 else max(1, min(32, len(os.sched_getaffinity(0))))
    orc.hs_oracle_verify_many(digests, S(32), S(stride), pks, sigs, S(n), mode, out, threads)
    return out.raw[:n]


def oracle_verify_batch(orc, digests, stride, pks, sigs, n, entropy=7):
    """dalek verify_batch through the oracle's RLC + Straus/Pippenger restatement (1 / 0)."""
    return orc.hs_oracle_verify_batch(digests, S(32), S(stride), pks, sigs, S(n), ctypes.c_uint64(entropy))


def bits(bitmap, n):
    return bytes((bitmap[i >> 3] >> (i & 7)) & 1 for i in range(n))
