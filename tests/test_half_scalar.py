"""halfscalar.h on the CPU (tests/emu): the short lattice vector (u, v) that turns dalek's  [s]B + [k](-A) == R  into
[v s]B + [u](-A) + [v](-R) == O  with half-size scalars.  Checked against Python big integers: the congruence
u = v k (mod 8L) - modulo the order of the WHOLE group, so that torsion components of A and R see the same multiples -
v odd and positive, the sizes the window loop is built for, and the special k for which no short odd-v vector exists."""
import ctypes
import random

import pytest

L = 2**252 + 27742317777372353535851937790883648493
N = 8 * L


def half(emu, k):
    u = ctypes.create_string_buffer(32)
    v = ctypes.create_string_buffer(32)
    neg = ctypes.c_int()
    took = emu.emu_half_scalar(k.to_bytes(32, "little"), u, ctypes.byref(neg), v)
    uu = int.from_bytes(u.raw, "little")
    return (-uu if neg.value else uu), int.from_bytes(v.raw, "little"), took


def test_random_scalars_give_valid_short_odd_pairs(emu):
    rng = random.Random(11)
    longest, took_all = 0, 0
    hist = {}
    for _ in range(3000):
        k = rng.randrange(L)
        u, v, took = half(emu, k)
        assert (u - v * k) % N == 0, hex(k)
        assert v % 2 == 1 and 0 < v < L and abs(u) < 2**253
        bits = max(abs(u).bit_length(), v.bit_length())
        hist[bits] = hist.get(bits, 0) + 1
        longest = max(longest, bits)
        took_all += took
    assert took_all == 3000                                   # the reduced pair passed its own check every time
    short = sum(c for b, c in hist.items() if b <= 131)
    assert short >= 0.985 * 3000, sorted(hist.items())        # ~99.6 % in the model; 33 windows of four bits
    assert longest <= 150, sorted(hist.items())


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 7, 8, 2**32, 2**64 + 1, 2**127, 2**128, 2**128 + 1, 2**200 + 12345, 2**252, L - 1, L - 2, L - 3,
                               (L - 1) // 2, (L + 1) // 2, L // 3, 2 * L // 3, 0x0fffffffffffffffffffffffffffffff, 8, 16, L - 8])
def test_special_scalars(emu, k):
    u, v, took = half(emu, k)
    assert (u - v * k) % N == 0 and v % 2 == 1 and 0 < v < L and abs(u) < 2**253
    if k < 2**100:
        assert (u, v) == (k, 1)                                # already short: the plain pair


def test_no_short_odd_vector_falls_back_to_a_long_pair(emu):
    """k = L - 1: u = v (L - 1) mod 8L means u = -v mod L and u = 4 mod 8 for odd v - no short solution.  The pair is
    long (the window loop runs its full length for that wavefront), never wrong."""
    u, v, _ = half(emu, L - 1)
    assert (u - v * (L - 1)) % N == 0 and v % 2 == 1
    assert max(abs(u).bit_length(), v.bit_length()) > 200


def test_recoded_length_is_the_number_of_nonzero_windows(emu):
    rng = random.Random(5)
    for bits in list(range(1, 40)) + [126, 127, 128, 129, 130, 131, 132, 133, 200, 252]:
        for _ in range(20):
            x = rng.randrange(2**(bits - 1), 2**bits)
            n = emu.emu_recode16_len(x.to_bytes(32, "little"))
            r = x + int("8" * 64, 16)
            digits = [((r >> (4 * i)) & 15) - 8 for i in range(64)]
            assert sum(d * 16**i for i, d in enumerate(digits)) == x
            top = max(i for i, d in enumerate(digits) if d != 0)
            assert n == top + 1 and (bits + 1) // 4 <= n <= (bits + 2) // 4 + 1


def test_radix8_recoding(emu):
    """The quad kernels' 3-bit-window form: x = sum d_i 8^i with d_i in [-4, 3], 85 digits popped from the top, and the
    count of windows that are really needed."""
    rng = random.Random(9)
    for bits in list(range(1, 30)) + [126, 127, 128, 129, 130, 131, 132, 133, 134, 200, 252, 253]:
        for _ in range(10):
            x = rng.randrange(2**(bits - 1), 2**bits)
            d = (ctypes.c_int8 * 85)()
            n = emu.emu_sc_digits3(x.to_bytes(32, "little"), d)
            digits = list(d)[::-1]                       # least significant first
            assert all(-4 <= v <= 3 for v in digits) and sum(v * 8**i for i, v in enumerate(digits)) == x
            top = max(i for i, v in enumerate(digits) if v != 0)
            assert n == top + 1


def pair_ok(emu, k, u, v):
    return emu.emu_half_scalar_ok(k.to_bytes(32, "little"), abs(u).to_bytes(32, "little"), 1 if u < 0 else 0,
                                  v.to_bytes(32, "little"))


def test_the_predicate_rejects_every_pair_that_could_change_a_verdict(emu):
    """half_scalar_ok is the safety net under the reduction ("the result is CHECKED before use"): it must hold on its own,
    whatever produced the pair.  [v]D = O <=> D = O needs gcd(v, 8L) = 1, i.e. v odd and 0 < v < L.  VERDICT r3 weak #8:
    (u, v) = (0, L) passes every congruence when k = 0 (mod 8) and makes the check [L]R == O - true for every torsion-free R."""
    rng = random.Random(3)
    for _ in range(200):
        k = rng.randrange(L)
        u, v, _ = half(emu, k)
        assert pair_ok(emu, k, u, v) == 1                                  # what the reduction produces passes
        assert pair_ok(emu, k, k, 1) == 1                                  # and so does the plain pair
        assert pair_ok(emu, k, u, v + 1) == 0 and pair_ok(emu, k, u + 1, v) == 0      # broken congruences
        assert pair_ok(emu, k, (2 * u) % N if u >= 0 else -((-2 * u) % N), 2 * v) == 0 or 2 * v >= L   # even v
        assert pair_ok(emu, k, 0, 0) == 0                                  # v = 0
    for k in [0, 8, 16, 8 * 12345, (L - 5) // 8 * 8, 2**252 - 8, rng.randrange(L) // 8 * 8]:
        assert k % 8 == 0 and k < L
        assert (0 - L * k) % N == 0                                        # (0, L) IS in the lattice ...
        assert pair_ok(emu, k, 0, L) == 0, hex(k)                          # ... and must not be usable
        uu = (k * L) % N                                                   # any u = L k (mod 8L), v = L
        if uu < 2**253:
            assert pair_ok(emu, k, uu, L) == 0
        assert pair_ok(emu, k, -((N - uu) % N), L) == 0 or (N - uu) % N >= 2**253
    # v in [L, 2^253), odd, congruent: 3L is odd and 3L k = k' ... take v = L + 2 with the matching u
    for _ in range(50):
        k = rng.randrange(L)
        v = L + 2 * rng.randrange(1, 2**100)
        u = (v * k) % N
        if u >= 2**253:
            u -= N
        if abs(u) < 2**253 and v < 2**253:
            assert pair_ok(emu, k, u, v) == 0                              # congruent, odd, but not below L
