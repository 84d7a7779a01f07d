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
