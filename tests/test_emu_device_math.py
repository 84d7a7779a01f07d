"""The product's device headers (hotstuff_amd/csrc/*.h) compiled for the CPU and checked
against the big-int ground truth, hashlib and the C oracle - the same arithmetic the
gfx950 kernels run, validated without a GPU."""
import ctypes
import hashlib
import os
import random

import pytest

import ed25519_ref as E
from conftest import S, oracle_verify_many

P, L = E.P, E.L


def rb(v):
    return int(v).to_bytes(32, "little")


def B(n):
    return ctypes.create_string_buffer(n)


SPECIAL = [0, 1, 2, P - 1, P - 2, P, P + 1, 2 ** 255 - 1, 2 ** 255 - 19, 2 ** 254, 19, (1 << 255) - 20,
           2 ** 26 - 1, 2 ** 51 - 1, (2 ** 255 - 1) ^ 0x5555]


def test_field_ops(emu):
    rnd = random.Random(11)
    o = B(32)
    for i in range(400):
        a = rnd.choice(SPECIAL) if i < 60 else rnd.getrandbits(255)
        b = rnd.choice(SPECIAL) if i < 30 else rnd.getrandbits(255)
        c = rnd.getrandbits(255)
        d = rnd.choice(SPECIAL) if i % 3 == 0 else rnd.getrandbits(255)
        emu.emu_fe_mul(rb(a), rb(b), o); assert int.from_bytes(o.raw, "little") == a * b % P
        emu.emu_fe_sq(rb(a), o); assert int.from_bytes(o.raw, "little") == a * a % P
        emu.emu_fe_roundtrip(rb(a), o); assert int.from_bytes(o.raw, "little") == a % 2 ** 255 % P
        # loose operands exercise the documented 2x / 3x limb bounds
        emu.emu_fe_mul_loose(rb(a), rb(c), rb(b), rb(d), o)
        assert int.from_bytes(o.raw, "little") == ((a % 2**255) + c) * ((b % 2**255) - (d % 2**255)) % P
        emu.emu_fe_sq_loose(rb(a), rb(d), o)
        assert int.from_bytes(o.raw, "little") == ((a % 2**255) - (d % 2**255)) ** 2 % P
        if i < 60:
            emu.emu_fe_invert(rb(a), o); assert int.from_bytes(o.raw, "little") == pow(a % 2**255, P - 2, P)


def test_sha512_and_scalars(emu):
    rnd = random.Random(12)
    o64, o = B(64), B(32)
    for i in range(100):
        m = os.urandom(96)
        emu.emu_sha512_96(m, o64)
        assert o64.raw == hashlib.sha512(m).digest()
    edge = [b"\xff" * 64, b"\0" * 64, (L * L - 1).to_bytes(64, "little"), L.to_bytes(64, "little"),
            (L - 1).to_bytes(64, "little"), (2 ** 252).to_bytes(64, "little"), (L << 259).to_bytes(64, "little"),
            (2 ** 511).to_bytes(64, "little"), (5 * L + 3).to_bytes(64, "little"), (2 * L - 1).to_bytes(64, "little")]
    for h in edge + [os.urandom(64) for _ in range(200)]:
        emu.emu_sc_reduce512(h, o)
        assert int.from_bytes(o.raw, "little") == int.from_bytes(h, "little") % L
    for _ in range(100):
        a, b = rnd.getrandbits(256), rnd.getrandbits(256)
        emu.emu_sc_mul(rb(a), rb(b), o)
        assert int.from_bytes(o.raw, "little") == a * b % L
    for s in [0, 1, L - 1, L, L + 1, 2 ** 252, 2 ** 253, 2 ** 256 - 1, L - 2 ** 32, L + 2 ** 64]:
        assert emu.emu_sc_is_canonical(rb(s)) == int(s < L)
    dg = (ctypes.c_int8 * 64)()
    for i in range(60):
        s = [0, 1, L - 1, 2 ** 253 - 1, 2 ** 252, 0x8888 << 200][i] if i < 6 else rnd.getrandbits(253)
        emu.emu_sc_digits(rb(s), dg)
        v = 0
        for d in dg:
            assert -8 <= d <= 7
            v = v * 16 + d
        assert v == s


def test_tc_vote_digest_on_device_matches_messages_rs(emu):
    """consensus/src/messages.rs:307-311: SHA-512(round_le || high_qc_round_le)[..32]."""
    import struct
    o = B(32)
    for rnd, hq in [(0, 0), (1, 0), (9, 3), (2 ** 64 - 1, 2 ** 63), (123456789012345, 987654321)]:
        emu.emu_tc_vote_digest(ctypes.c_uint64(rnd), ctypes.c_uint64(hq), o)
        assert o.raw == hashlib.sha512(struct.pack("<QQ", rnd, hq)).digest()[:32]


def test_radix256_recoding(emu):
    rnd = random.Random(13)
    dg = (ctypes.c_int16 * 32)()
    for i in range(60):
        s = [0, 1, L - 1, 2 ** 253 - 1, 2 ** 252, 0x80 << 200, 0x7f << 120][i] if i < 7 else rnd.getrandbits(253)
        emu.emu_sc_digits8(rb(s), dg)
        v = 0
        for d in dg:
            assert -128 <= d <= 127
            v = v * 256 + d
        assert v == s


def test_decompress_matches_dalek_rules(emu, edge_vectors):
    x, y, sm = B(32), B(32), ctypes.c_int()
    encs = [os.urandom(32) for _ in range(80)]
    encs += [bytes.fromhex(v["pk"]) for v in edge_vectors] + [bytes.fromhex(v["sig"])[:32] for v in edge_vectors]
    # every encoding of a small-order point, canonical or not (y, y + p where that fits 255 bits; either sign bit),
    # and their neighbours: the kernels decide small order by comparing the canonical y with the five values a
    # small-order point can have (ge_is_small_order_affine) - it must agree with [8]P == O everywhere
    small_ys = sorted({t[1] * pow(t[2], -1, E.P) % E.P for t in E.eight_torsion()})
    assert len(small_ys) == 5
    for ys in small_ys:
        for yy in (ys, ys + E.P, ys + 1, ys - 1, ys ^ (1 << 200)):
            if 0 <= yy < (1 << 255):
                for sign in (0, 1):
                    encs.append((yy | (sign << 255)).to_bytes(32, "little"))
    seen_small = 0
    for e in encs:
        ok = emu.emu_decompress(e, x, y, ctypes.byref(sm))
        p = E.decompress(e)
        assert bool(ok) == (p is not None)
        if ok:
            assert int.from_bytes(x.raw, "little") == p[0] and int.from_bytes(y.raw, "little") == p[1]
            want = E.pt_is_small_order(p)
            assert sm.value == (3 if want else 0), (e.hex(), sm.value, want)
            seen_small += want
    assert seen_small >= 14      # 8 canonical + the non-canonical spellings of y = 0, 1 (and -0)


@pytest.mark.parametrize("fn", ["emu_verify", "emu_verify_quad", "emu_verify_quad3", "emu_verify_oct", "emu_verify_oct_split"])
def test_verify_stages_on_edge_vectors(emu, edge_vectors, fn):
    """fn = the one-lane-per-signature stage 2 or the quad (4 lanes per signature) stage 2."""
    verify = getattr(emu, fn)
    n = len(edge_vectors)
    msgs = b"".join(bytes.fromhex(v["msg"]) for v in edge_vectors)
    pks = b"".join(bytes.fromhex(v["pk"]) for v in edge_vectors)
    sigs = b"".join(bytes.fromhex(v["sig"]) for v in edge_vectors)
    out = B(n)
    verify(msgs, S(32), pks, sigs, S(n), 1, out)
    assert [bool(b) for b in out.raw] == [v["strict"] for v in edge_vectors]
    verify(msgs, S(32), pks, sigs, S(n), 0, out)
    assert [bool(b) for b in out.raw] == [v["cofactorless"] for v in edge_vectors]


@pytest.mark.parametrize("fn", ["emu_verify", "emu_verify_quad", "emu_verify_quad3", "emu_verify_oct", "emu_verify_oct_split"])
def test_verify_stages_vs_c_oracle_random(emu, oracle, fn):
    import synth
    verify = getattr(emu, fn)
    n = 96
    for shared in (False, True):
        dg, pk, sg = synth.make_batch(n, shared=shared)
        dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, shared, rate_byte=64)
        stride = 0 if shared else 32
        for mode in (0, 1):
            out = B(n)
            verify(dg, S(stride), pk, sg, S(n), mode, out)
            assert out.raw == oracle_verify_many(oracle, dg, stride, pk, sg, n, mode)
            assert [i for i in range(n) if not out.raw[i]] == bad


_BASE_TABLES = {}


def _emu_base_tables(emu):
    """(8-bit table, 16-bit table) of the base point, built once per session (the 16-bit one is 67 MB, ~2 s)."""
    import numpy as np
    if "t" not in _BASE_TABLES:
        emu.emu_keyset_words.restype = ctypes.c_size_t
        emu.emu_base16_words.restype = ctypes.c_size_t
        btab = np.zeros(emu.emu_keyset_words(S(1)), dtype=np.uint32)
        emu.emu_keyset_build_base(btab.ctypes.data_as(ctypes.c_void_p))
        b16 = np.zeros(emu.emu_base16_words(), dtype=np.uint32)
        emu.emu_base16_build(btab.ctypes.data_as(ctypes.c_void_p), b16.ctypes.data_as(ctypes.c_void_p))
        emu.emu_set_base16(b16.ctypes.data_as(ctypes.c_void_p))
        _BASE_TABLES["t"] = (btab, b16)
    return _BASE_TABLES["t"]


def test_base_point_16_bit_table(emu):
    """T_B16[w][j] = j 2^(16w) B as the kernels read it (built from the 8-bit table by base16_entry_projective +
    the shared-inversion normalisation): sampled entries incl. the ends of every window vs the big-int ground truth;
    and the radix-65536 recoding sums back to the scalar with digits in [-32768, 32767]."""
    _, b16 = _emu_base_tables(emu)
    rng = random.Random(16)
    out = B(96)
    picks = [(w, j) for w in range(16) for j in (0, 1, 2, 127, 128, 129, 255, 256, 257, 384, 385, 32639, 32640, 32767, 32768)]
    picks += [(rng.randrange(16), rng.randrange(32769)) for _ in range(300)]
    for w, j in picks:
        e = b16[(w * 32769 + j) * 32:(w * 32769 + j) * 32 + 32]
        emu.emu_entry_canonical(e.ctypes.data_as(ctypes.c_void_p), out)
        ymx, ypx, t2d = (int.from_bytes(out.raw[32 * c:32 * c + 32], "little") for c in range(3))
        p = E.pt_mul(j << (16 * w), E.B)
        zi = pow(p[2], -1, E.P)
        x, y = p[0] * zi % E.P, p[1] * zi % E.P
        assert (ymx, ypx, t2d) == ((y - x) % E.P, (y + x) % E.P, 2 * E.D * x * y % E.P), (w, j)
    dg = (ctypes.c_int32 * 16)()
    for _ in range(200):
        sc = rng.randrange(E.L) if rng.random() < 0.8 else rng.choice([0, 1, E.L - 1, (1 << 252) - 1, 0x8000, 0x7fff, (1 << 253) - 1])
        emu.emu_sc_digits16(sc.to_bytes(32, "little"), dg)
        assert all(-32768 <= d <= 32767 for d in dg)
        assert sum(d << (16 * (15 - i)) for i, d in enumerate(dg)) == sc


def test_two_lane_split_covers_every_window_once(emu):
    """lane2_split_* (the default key-cache kernel's work split): over the two lanes' 24 steps each of the key's 32
    8-bit windows and each of the base point's 16 16-bit windows appears exactly once, with the digit of k resp. s that
    belongs to it - so that sum_w T_A[w][k_w] + sum_w T_B16[w][s_w] = [k](-A) + [s]B."""
    rng = random.Random(24)
    tr = (ctypes.c_int32 * 72)()
    for _ in range(100):
        k = rng.randrange(E.L) if rng.random() < 0.8 else rng.choice([0, 1, E.L - 1, (1 << 252) - 1, 128, 0x8000, (1 << 253) - 1])
        sc = rng.randrange(E.L) if rng.random() < 0.8 else rng.choice([0, 1, E.L - 1, 0x7fff, 0x8000, (1 << 252) + 5])
        key_digits, base_digits = {}, {}
        for odd in (0, 1):
            emu.emu_lane2_split_trace(k.to_bytes(32, "little"), sc.to_bytes(32, "little"), odd, tr)
            for t in range(24):
                tab, w, d = tr[3 * t], tr[3 * t + 1], tr[3 * t + 2]
                dst = base_digits if tab else key_digits
                assert w not in dst, (odd, t, tab, w)
                dst[w] = d
        assert sorted(key_digits) == list(range(32)) and sorted(base_digits) == list(range(16))
        assert all(-128 <= d <= 128 for d in key_digits.values()) and all(-32768 <= d <= 32768 for d in base_digits.values())
        assert sum(d << (8 * w) for w, d in key_digits.items()) == k
        assert sum(d << (16 * w) for w, d in base_digits.items()) == sc


def _emu_cached(emu, keys, key_idx, msgs, stride, sigs, n, mode, lane_form=0):
    import numpy as np
    emu.emu_cached_set_lane_form(lane_form)
    emu.emu_keyset_words.restype = ctypes.c_size_t
    kw = emu.emu_keyset_words(S(1))
    nk = len(keys) // 32
    btab, _ = _emu_base_tables(emu)
    tab = np.zeros(kw * nk, dtype=np.uint32)
    flag = np.zeros(nk, dtype=np.uint32)
    emu.emu_keyset_build(keys, S(nk), tab.ctypes.data_as(ctypes.c_void_p), flag.ctypes.data_as(ctypes.c_void_p))
    ki = np.array(key_idx, dtype=np.uint32)
    out = B(n)
    emu.emu_verify_cached(tab.ctypes.data_as(ctypes.c_void_p), flag.ctypes.data_as(ctypes.c_void_p), keys, S(nk),
                          btab.ctypes.data_as(ctypes.c_void_p), msgs, S(stride), ki.ctypes.data_as(ctypes.c_void_p),
                          sigs, S(n), mode, out)
    return out.raw


@pytest.mark.parametrize("lane_form", [0, 1, 2, 3, 4, 5])
def test_cached_path_on_edge_vectors(emu, edge_vectors, lane_form):
    """Committee key cache algorithm (verify_cached.h: per-key window tables, 64 table additions,
    R checked through one inversion instead of a decompression): the vector's key is
    registered, verdicts must equal the cold path's / the big-int oracle's in both modes."""
    vec = edge_vectors[::2]          # half the set keeps the CPU table build short; the GPU test runs all
    n = len(vec)
    msgs = b"".join(bytes.fromhex(v["msg"]) for v in vec)
    pks = b"".join(bytes.fromhex(v["pk"]) for v in vec)
    sigs = b"".join(bytes.fromhex(v["sig"]) for v in vec)
    got = _emu_cached(emu, pks, list(range(n)), msgs, 32, sigs, n, 1, lane_form)
    assert [bool(b) for b in got] == [v["strict"] for v in vec]
    got = _emu_cached(emu, pks, list(range(n)), msgs, 32, sigs, n, 0, lane_form)
    assert [bool(b) for b in got] == [v["cofactorless"] for v in vec]


def test_cached_path_vs_c_oracle_random(emu, oracle):
    import synth
    n = 24
    dg, pk, sg = synth.make_batch(n, shared=True)
    dg, pk, sg, bad = synth.corrupt(dg, pk, sg, n, True, rate_byte=80)
    for mode in (0, 1):
        got = _emu_cached(emu, pk, list(range(n)), dg, 0, sg, n, mode)
        assert got == oracle_verify_many(oracle, dg, 0, pk, sg, n, mode)
    # an index outside the key set rejects that signature only
    got = _emu_cached(emu, pk, [0, 1, 99] + list(range(3, n)), dg, 0, sg, n, 0)
    assert not got[2] and got[:2] == oracle_verify_many(oracle, dg, 0, pk, sg, n, 0)[:2]


def test_batched_finish_over_two_blocks_of_finish_lanes_vs_c_oracle(emu, oracle):
    """The "c" form's finish lanes take signatures L, L + nl, L + 2 nl, ... (cfin_lane): 1,100 signatures by 12 committee keys,
    per-signature digests, ~30 % with one bit flipped somewhere in (digest, R, s) or pointing outside the key set - nl = 128
    (two wavefronts of finish lanes, 9 signatures each at most, ragged tails) - both modes against the C oracle."""
    import hashlib
    import struct
    import synth
    nk, n = 12, 1100
    keys = [synth.sign(synth.key_seed(900 + k), b"x" * 32)[0] for k in range(nk)]
    dg = bytearray(); sg = bytearray(); pk = bytearray(); idx = []
    for i in range(n):
        k = i % nk
        m = hashlib.sha512(b"emu-c" + struct.pack("<I", i)).digest()[:32]
        _, s_ = synth.sign(synth.key_seed(900 + k), m)
        dg += m; sg += s_; pk += keys[k]; idx.append(k)
    for i in range(0, n, 3):
        h = hashlib.sha512(b"emu-bad" + struct.pack("<I", i)).digest()
        if h[0] % 4 == 0:
            dg[32 * i + h[1] % 32] ^= 1 << (h[2] % 8)
        elif h[0] % 4 == 1:
            sg[64 * i + h[1] % 32] ^= 1 << (h[2] % 8)
        elif h[0] % 4 == 2:
            sg[64 * i + 32 + h[1] % 32] ^= 1 << (h[2] % 8)
        else:
            idx[i] = 77                                  # not a committee member
    for mode in (0, 1):
        want = bytearray(oracle_verify_many(oracle, bytes(dg), 32, bytes(pk), bytes(sg), n, mode))
        for i in range(n):
            if idx[i] >= nk:
                want[i] = 0
        for form in (4, 5):              # sixteen / four signatures per finish lane
            got = _emu_cached(emu, b"".join(keys), idx, bytes(dg), 32, bytes(sg), n, mode, lane_form=form)
            assert got == bytes(want)
        assert 250 < sum(1 for b in got if not b) < 450


@pytest.mark.parametrize("fn", ["emu_verify", "emu_verify_quad", "emu_verify_quad3", "emu_verify_oct", "emu_verify_oct_split"])
def test_torsion_fuzz_on_device_math(emu, torsion_fuzz, fn):
    vec = torsion_fuzz
    n = len(vec)
    msgs = b"".join(bytes.fromhex(v["msg"]) for v in vec)
    pks = b"".join(bytes.fromhex(v["pk"]) for v in vec)
    sigs = b"".join(bytes.fromhex(v["sig"]) for v in vec)
    out = B(n)
    getattr(emu, fn)(msgs, S(32), pks, sigs, S(n), 1, out)
    assert [bool(b) for b in out.raw] == [v["strict"] for v in vec]
    getattr(emu, fn)(msgs, S(32), pks, sigs, S(n), 0, out)
    assert [bool(b) for b in out.raw] == [v["cofactorless"] for v in vec]


@pytest.mark.parametrize("lane_form", [0, 1, 2, 3, 4])
def test_torsion_fuzz_on_cached_path(emu, torsion_fuzz, lane_form):
    """lane_form 0 = quad core finishing by inversion, 1 = one lane per signature,
    2 = quad chain + separately decompressed R compared as points (the 5-wavefront kernel),
    3 = one-lane chains + separately decompressed R (the default three-wavefront kernel),
    4 = two-lane chains, then finish lanes compressing HS_CFIN_M signatures each with one inversion (the "c" forms)."""
    vec = torsion_fuzz[:40]
    n = len(vec)
    msgs = b"".join(bytes.fromhex(v["msg"]) for v in vec)
    pks = b"".join(bytes.fromhex(v["pk"]) for v in vec)
    sigs = b"".join(bytes.fromhex(v["sig"]) for v in vec)
    assert [bool(b) for b in _emu_cached(emu, pks, list(range(n)), msgs, 32, sigs, n, 1, lane_form)] == [v["strict"] for v in vec]
    assert [bool(b) for b in _emu_cached(emu, pks, list(range(n)), msgs, 32, sigs, n, 0, lane_form)] == [v["cofactorless"] for v in vec]


def test_every_finishing_form_fails_closed_on_a_zero_accumulator(emu):
    """An all-zero accumulator is absorbing under the addition formulas and satisfies
    X == x_R * Z and Y == y_R * Z for every R; it is what a faulted doubling produced on the GPU
    once (the v_subrev_u32_dpp trap) and every signature verified.  All five ways the
    kernels finish must reject it - and still accept a genuine point (the batched finish also its genuine neighbours)."""
    import ctypes
    emu.emu_zero_accumulator_accepts.restype = ctypes.c_uint32
    emu.emu_genuine_point_accepts.restype = ctypes.c_uint32
    assert emu.emu_genuine_point_accepts() == 0b11111
    assert emu.emu_zero_accumulator_accepts() == 0


def test_affine_table_entries_device_form_equals_batch_form(emu):
    """Key-table entries are affine Niels points; the device inverts every entry on a lane of its own
    (keyset_affine_entry), the emulation normalises a window with one shared inversion - same field elements."""
    import synth
    for i, win in ((0, 0), (1, 7), (2, 31)):
        pk, _ = synth.sign(synth.key_seed(600 + i), b"m" * 32)
        assert emu.emu_affine_forms_agree(pk, win) == 1
