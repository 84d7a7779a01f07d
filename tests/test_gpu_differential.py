"""A 10^6 slice of the scale differential (tools/differential.py; VERDICT r4 "Next round" #3): crafted tuples - torsion on A and R,
pure-torsion keys, alternative encodings, s around L - through EVERY cold and key-cache kernel form, STRICT and BATCH, against
the C oracle's labels.  The full 10^7 run is kept under profiles/ (r5_differential.json)."""
import os
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
pytestmark = pytest.mark.gpu


def test_a_million_crafted_tuples_through_every_kernel_form():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import subprocess
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "craft")], check=True, capture_output=True)
    import differential
    rep = differential.run(1 << 20, 1 << 18, seed=11)          # four chunks; the fourth over one shared digest
    assert rep["total"] == 1 << 20 and rep["shared_digest_chunks"] == 1
    assert rep["label_check"]["mismatches"] == 0, rep["label_check"]
    assert rep["mismatches"] == 0, rep["first_mismatches"][:4]
    rlc = {k: f for k, f in rep["forms"].items() if k.startswith("k_rlc_")}
    assert all(f["verifications"] == 2 << 20 for k, f in rep["forms"].items() if k not in rlc), rep["forms"]
    assert len(rep["forms"]) >= 17 and len(rlc) == 2
    for f in rlc.values():       # mixed calls localised to BATCH's labels; honest-only calls accepted by the combination itself
        assert f["mismatches"] == 0 and f["honest_calls"] >= 8 and f["calls_accepted_by_the_combination"] == f["honest_calls"], f
    # the classes the differential exists for were there, with both verdicts
    for name in ("A+T", "A+T,R+T", "torsion-A,R+T"):
        c = rep["classes"][name]
        assert c["tuples"] > 10000 and 0 < c["accept_batch"] < c["tuples"], (name, c)
    assert rep["classes"]["torsion-A"]["accept_strict"] == 0 and rep["classes"]["torsion-A"]["accept_batch"] > 0
    assert rep["classes"]["R+T"]["accept_batch"] == 0 and rep["classes"]["s+L"]["accept_batch"] == 0


def test_half_scalar_reduction_on_the_device_at_scale():
    """halfscalar.h on the GPU: 10^6 random + 10^5 adversarial scalars (k = 0 mod 2^j, next to 0 and L, long runs, floor(8L / m),
    rational multiples, near-golden) - bit-identical to the host twin, every returned pair satisfying the predicate the
    verification relies on (recomputed with independent arithmetic), random scalars never falling back to the long pair.
    The 10^7 run is profiles/r5_half_differential.json."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "tools", "half_differential")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build_tools()
    r = subprocess.run([exe, "1000000", "100000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-500:])
    rep = json.loads(r.stdout)
    assert rep["device"] is True
    for part in ("random", "adversarial"):
        assert rep[part]["device_differs_from_host_twin"] == 0 and rep[part]["predicate_failures"] == 0, rep[part]["first_bad"]
    assert rep["random"]["scalars"] == 1000000 and rep["random"]["long_fallback_pairs_k_1"] == 0
    assert rep["random"]["max_bits_of_pairs_that_are_not_the_long_fallback"] <= 160
    assert rep["adversarial"]["max_bits_abs_u"] <= 253 and rep["adversarial"]["max_bits_v"] <= 192
