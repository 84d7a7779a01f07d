"""CPU-only: the wire-decode and signing arithmetic of hotstuff_amd/csrc/wire.h, run by tests/emu.
Wire decode vs Python's base64 / struct on canonical and malformed vote elements; signing and
public-key derivation bit-exact vs OpenSSL (RFC 8032 signatures are deterministic)."""
import base64
import ctypes
import struct

import pytest

import synth


def element(key_text, sig, length=None):
    b = key_text.encode() if isinstance(key_text, str) else bytes(key_text)
    return struct.pack("<Q", len(b) if length is None else length) + b + sig


def wire_cases():
    """(name, 116-byte element, canonical?)"""
    pk, sg = synth.sign(synth.key_seed(77), synth.msg_digest(77))
    good = base64.b64encode(pk).decode()
    assert len(good) == 44 and good[43] == "="
    cases = [("canonical", element(good, sg), True)]
    cases.append(("all-zero key", element(base64.b64encode(bytes(32)).decode(), sg), True))
    cases.append(("all-ones key", element(base64.b64encode(b"\xff" * 32).decode(), sg), True))
    cases.append(("length field 43", element(good, sg, 43), False))
    cases.append(("length field 2^32 + 44", element(good, sg, (1 << 32) + 44), False))
    for bad in "*", " ", "\n", "-", "_", "=", "\x00", "\xc3":
        t = bytearray(good.encode()); t[7] = ord(bad) if len(bad.encode("latin1")) == 1 else 0xc3
        cases.append(("invalid character %r" % bad, element(bytes(t), sg), False))
    cases.append(("missing padding", element(good[:43] + "A", sg), False))
    # trailing bits: the 43rd character carries 4 data bits + 2 bits that must be zero
    alphabet = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"
    v = alphabet.index(good[42])
    cases.append(("non-zero trailing bits", element(good[:42] + alphabet[v | 1] + "=", sg), False))
    cases.append(("url-safe alphabet", element(base64.urlsafe_b64encode(b"\xfb\xff" + pk[2:]).decode(), sg), False))
    return pk, sg, cases


def python_canonical(el):
    ln = struct.unpack("<Q", el[:8])[0]
    text = el[8:52]
    if ln != 44 or text[43:44] != b"=":
        return None
    try:
        raw = base64.b64decode(text, validate=True)
    except Exception:
        return None
    if len(raw) != 32 or base64.b64encode(raw) != text:
        return None
    return raw


def test_wire_decode_matches_python_base64(emu):
    pk, sg, cases = wire_cases()
    seen_bad = 0
    for name, el, canonical in cases:
        assert len(el) == 116, name
        want = python_canonical(el)
        assert (want is not None) == canonical, name
        o_pk, o_sig = ctypes.create_string_buffer(32), ctypes.create_string_buffer(64)
        ok = emu.emu_wire_decode(el, o_pk, o_sig)
        assert bool(ok) == canonical, name
        if canonical:
            assert o_pk.raw == want and o_sig.raw == el[52:], name
        else:
            assert o_sig.raw[32:] == b"\xff" * 32, name      # poisoned: s >= L, rejected by every kernel
            seen_bad += 1
    assert seen_bad >= 10


def test_signing_is_bit_exact_vs_openssl(emu):
    for i in range(40):
        seed = synth.key_seed(9000 + i)
        msg = synth.msg_digest(i) if i % 2 else synth.shared_digest(i)
        pk, sg = synth.sign(seed, msg)
        out = ctypes.create_string_buffer(64)
        emu.emu_sign(seed + pk, msg, out)
        assert out.raw == sg
        p = ctypes.create_string_buffer(32)
        emu.emu_public_key(seed, p)
        assert p.raw == pk
