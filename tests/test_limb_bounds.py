"""Limb-bound audit of the device field arithmetic (fe26.h).

tests/emu is rebuilt with -DHS_BOUND_CHECK: every fe_add / fe_sub / fe_carry / fe_mul / fe_sq
then asserts its documented preconditions (subtrahend within the 2p bias, 19*g and 38*f fit 32
bits, no 64-bit column wraps) and aborts the process on a violation.  All vector sets - edge,
torsion fuzz, extreme limb patterns (y = 2^255 - 1, p - 1, all-ones ...) as keys, R's and field
inputs - are pushed through every stage (one lane, quad, key-set build, cached forms).  A lazy
reduction bound that is too optimistic anywhere on these paths kills this test."""
import ctypes
import os
import subprocess

import pytest

from conftest import ROOT, S

H = bytes.fromhex


@pytest.fixture(scope="module")
def emu_check():
    d = os.path.join(ROOT, "tests", "emu")
    subprocess.run(["make", "-C", d, "libhs_emu_check.so"], check=True, capture_output=True)
    lib = ctypes.CDLL(os.path.join(d, "libhs_emu_check.so"))
    lib.emu_keyset_words.restype = ctypes.c_size_t
    return lib


def _extreme_encodings():
    p = 2 ** 255 - 19
    vals = [0, 1, 2, p - 1, p - 2, p, p + 1, 2 ** 255 - 1, 2 ** 255 - 20, 2 ** 254, 2 ** 254 - 1,
            int("55" * 32, 16) & (2 ** 255 - 1), int("aa" * 32, 16) & (2 ** 255 - 1),
            (2 ** 255 - 1) ^ (2 ** 26 - 1), (2 ** 255 - 1) ^ ((2 ** 25 - 1) << 26), 2 ** 26 - 1, (2 ** 51) - 1]
    out = []
    for v in vals:
        for sign in (0, 1):
            out.append(int(v | (sign << 255)).to_bytes(32, "little"))
    return out


def test_every_path_respects_the_limb_bounds(emu_check, edge_vectors, torsion_fuzz):
    import numpy as np
    emu = emu_check
    B = ctypes.create_string_buffer
    vec = edge_vectors + torsion_fuzz
    msgs = [H(v["msg"]) for v in vec]
    pks = [H(v["pk"]) for v in vec]
    sigs = [H(v["sig"]) for v in vec]
    # extreme encodings as A and as R (with arbitrary s < 2^252 so the curve code runs in full)
    for e in _extreme_encodings():
        msgs.append(bytes(32)); pks.append(e); sigs.append(e[::-1][:32][::-1] + (7).to_bytes(32, "little"))
        msgs.append(bytes(range(32))); pks.append(pks[0]); sigs.append(e + (2 ** 251 + 12345).to_bytes(32, "little"))
    n = len(msgs)
    m, p, s = b"".join(msgs), b"".join(pks), b"".join(sigs)
    out = B(n)
    for fn in ("emu_verify", "emu_verify_quad", "emu_verify_quad3", "emu_verify_oct"):
        for mode in (0, 1):
            getattr(emu, fn)(m, S(32), p, s, S(n), mode, out)          # aborts on a bound violation
    # field-level extremes through the loose-operand entry points
    o = B(32)
    ext = _extreme_encodings()
    for a in ext[::2]:
        for b in ext[::3]:
            emu.emu_fe_mul_loose(a, b, b, a, o)
            emu.emu_fe_sq_loose(a, b, o)
            emu.emu_fe_mul(a, b, o)
        emu.emu_fe_invert(a, o)
    # key-set build and the three cached forms on a subset (the build is the slow part on a CPU)
    sub = list(range(0, len(vec), 6)) + list(range(len(vec), n, 9))
    k = len(sub)
    keys = b"".join(pks[i] for i in sub)
    kw = emu.emu_keyset_words(S(1))
    btab = np.zeros(kw, dtype=np.uint32)
    emu.emu_keyset_build_base(btab.ctypes.data_as(ctypes.c_void_p))
    tab = np.zeros(kw * k, dtype=np.uint32)
    flag = np.zeros(k, dtype=np.uint32)
    emu.emu_keyset_build(keys, S(k), tab.ctypes.data_as(ctypes.c_void_p), flag.ctypes.data_as(ctypes.c_void_p))
    ki = np.arange(k, dtype=np.uint32)
    sm, ss = b"".join(msgs[i] for i in sub), b"".join(sigs[i] for i in sub)
    out = B(k)
    for form in (0, 1, 2, 3, 4):
        emu.emu_cached_set_lane_form(form)
        for mode in (0, 1):
            emu.emu_verify_cached(tab.ctypes.data_as(ctypes.c_void_p), flag.ctypes.data_as(ctypes.c_void_p), keys, S(k),
                                  btab.ctypes.data_as(ctypes.c_void_p), sm, S(32), ki.ctypes.data_as(ctypes.c_void_p),
                                  ss, S(k), mode, out)
    emu.emu_cached_set_lane_form(0)


def test_the_assertions_are_live(emu_check):
    """A deliberate violation (19*g overflowing 32 bits) must abort the checked build."""
    import sys
    so = os.path.join(ROOT, "tests", "emu", "libhs_emu_check.so")
    code = "import ctypes; l = ctypes.CDLL(%r); b = ctypes.create_string_buffer(32); l.emu_violate_bounds(b)" % so
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "HS_BOUND_CHECK failed" in r.stderr, (r.returncode, r.stderr[-300:])
