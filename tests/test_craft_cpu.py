"""The adversarial signer of the scale differential (tests/craft), pinned on the CPU before the GPU is asked anything:
every crafted tuple's labels equal the big-integer ground truth (oracle/ed25519_ref.py), the torsion classes follow the law
they were built for - cofactorless accepts iff T_j + [k]T_i = O, strict additionally rejects small-order A' and R' - and the
one-pass labeller agrees with the oracle's two entry points."""
import hashlib

import numpy as np
import pytest

import craft
import ed25519_ref as ref


@pytest.fixture(scope="module")
def chunk():
    import subprocess, os
    subprocess.run(["make", "-C", craft.HERE], check=True, capture_output=True)
    l = craft.lib()
    rng = np.random.default_rng(20260922)
    pool = craft.key_pool(16)
    return l, craft.craft_chunk(l, 520, rng, pool, 4), pool


def test_every_class_is_drawn_and_both_verdicts_occur(chunk):
    _, c, _ = chunk
    seen = set(int(x) for x in c["cls"])
    assert seen == set(range(len(craft.CLASSES))), [craft.CLASSES[i] for i in set(range(len(craft.CLASSES))) - seen]
    honest = c["cls"] == craft.CLASSES.index("honest")
    assert (c["expect"][honest] == 3).all()                          # valid signatures: both predicates accept
    for name in ("A+T", "A+T,R+T", "torsion-A,R+T"):
        m = c["cls"] == craft.CLASSES.index(name)
        assert (c["expect"][m] & 1).any() and not (c["expect"][m] & 1).all(), name      # the verdict really depends on k
    m = c["cls"] == craft.CLASSES.index("R+T")
    assert not (c["expect"][m] & 1).any()                            # a lone torsion component on R never cancels
    for name in ("s+L", "s=L", "s=L+1", "s|2^253", "s|2^255"):
        m = c["cls"] == craft.CLASSES.index(name)
        assert not c["expect"][m].any(), name                        # s >= L is rejected before any arithmetic


def test_labels_equal_the_big_integer_reference(chunk):
    _, c, _ = chunk
    bad = []
    for i in range(len(c["cls"])):
        msg, pk, sig = bytes(c["msgs"][i]), bytes(c["pk"][i]), bytes(c["sig"][i])
        want = (1 if ref.verify_cofactorless(msg, pk, sig) else 0) | (2 if ref.verify_strict(msg, pk, sig) else 0)
        if want != int(c["expect"][i]):
            bad.append((i, craft.CLASSES[c["cls"][i]], want, int(c["expect"][i])))
    assert not bad, bad[:5]


def test_torsion_classes_follow_the_law_they_were_built_for(chunk):
    """[s]B - [k]A' = R' - T_j - [k]T_i by construction, so cofactorless accepts iff T_j + [k]T_i = O."""
    _, c, _ = chunk
    T = ref.eight_torsion()
    # craft.c's T_1 is one particular generator; identify each crafted torsion component by decoding instead of by index
    checked = 0
    for i in range(len(c["cls"])):
        name = craft.CLASSES[c["cls"][i]]
        if name not in ("A+T", "R+T", "A+T,R+T", "torsion-A", "torsion-A,R+T", "torsion-R"):
            continue
        msg, pk, sig = bytes(c["msgs"][i]), bytes(c["pk"][i]), bytes(c["sig"][i])
        A, R = ref.decompress(pk), ref.decompress(sig[:32])
        tA, tR = ref.pt_mul(ref.L, A), ref.pt_mul(ref.L, R)            # [L]P = [L]T: the torsion component times L (L = 5 mod 8, invertible)
        k = int.from_bytes(hashlib.sha512(sig[:32] + pk + msg).digest(), "little") % ref.L
        linv = pow(ref.L, -1, 8)
        tA, tR = ref.pt_mul(linv, tA), ref.pt_mul(linv, tR)            # the torsion components themselves
        law = ref.pt_is_identity(ref.pt_add(tR, ref.pt_mul(k % 8, tA)))
        assert bool(c["expect"][i] & 1) == law, (i, name)
        small = ref.pt_is_small_order(A) or ref.pt_is_small_order(R)
        assert bool(c["expect"][i] & 2) == (law and not small), (i, name)
        checked += 1
    assert checked > 100


def test_alternative_encodings_are_really_alternative(chunk):
    _, c, _ = chunk
    noncanon = 0
    for name in ("noncanonical-A", "noncanonical-R", "noncanonical-A,R"):
        for i in np.nonzero(c["cls"] == craft.CLASSES.index(name))[0]:
            for enc in ((bytes(c["pk"][i]),) if name.endswith("-A") else (bytes(c["sig"][i][:32]),) if name.endswith("-R")
                        else (bytes(c["pk"][i]), bytes(c["sig"][i][:32]))):
                pt = ref.decompress(enc)
                assert pt is not None and ref.pt_is_small_order(pt)
                if ref.compress(pt) != enc:
                    noncanon += 1
    assert noncanon > 10          # y + p and "-0" forms were produced, not only requested


def test_one_pass_labeller_equals_the_oracles_two_entry_points(chunk):
    l, c, _ = chunk
    rng = np.random.default_rng(3)
    assert craft.check_labels_against_the_oracle(l, c, rng, 520, 4) == 0


def test_labels_equal_libsodium_group_arithmetic(chunk):
    """The same labels against arithmetic this project did not write (tests/referees.py: dalek's point laws over libsodium's
    crypto_core_ed25519_add, lenient decoding included): the 10^7-tuple differential's ground truth is then pinned by the
    big-integer reference AND by an independent implementation, class by class - torsion on A / R, pure-torsion keys,
    alternative encodings, every s mutation, byte flips."""
    import referees
    if referees._sodium_group() is None:
        pytest.skip("no libsodium with the ed25519 core API in this image")
    _, c, _ = chunk
    bad, per_class = [], {}
    for i in range(len(c["cls"])):
        msg, pk, sig = bytes(c["msgs"][i]), bytes(c["pk"][i]), bytes(c["sig"][i])
        laws = referees.sodium_point_laws(msg, pk, sig)
        want = (1 if laws["cofactorless"] else 0) | (2 if laws["strict"] else 0)
        name = craft.CLASSES[c["cls"][i]]
        per_class[name] = per_class.get(name, 0) + 1
        if want != int(c["expect"][i]):
            bad.append((i, name, want, int(c["expect"][i])))
    assert not bad, bad[:5]
    assert len(per_class) == len(craft.CLASSES) and min(per_class.values()) >= 2, per_class
