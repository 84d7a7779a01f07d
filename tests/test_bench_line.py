"""bench.py's line against what the driver's record keeps of it (BENCH_r05's `parsed` lost both QC p50s - the
record holds the first 24 keys of `config`, 40 characters of a key, 120 of a string).  CPU only: finalize_line on a synthetic line."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _line():
    cfg = {"signatures_per_batch": 1024, "streams": 16, "entry_point": "hs_ed25519_verify_device_batches",
           "workload": "configs[1]: 1024-tuple batches, strict; value = HBM-resident rate; value_host_to_host = host buffers in, verdict out"}
    # the extras arrive in whatever order the bench ran them
    for k in reversed(bench.CONFIG_FIRST):
        cfg.setdefault(k, 1.0)
    cfg["sq_verify_cached_lane2r_wait_inst_any"] = 5
    return {"metric": bench.METRIC, "value": 1e8, "unit": "verifies/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 10.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": cfg,
            "roofline": {"bound": "hbm", "note": "x" * 120}, "host_enqueue_ms_per_step": 0.4, "cpu_baseline": {"value": 4e5, "sample": "y" * 100}}


def test_the_metric_keys_come_first_and_in_order():
    out = bench.finalize_line(_line(), strict=True)
    keys = list(out["config"])
    assert tuple(keys[:len(bench.CONFIG_FIRST)]) == bench.CONFIG_FIRST and len(bench.CONFIG_FIRST) == 24
    # the twenty the judge named (VERDICT r5 "Next round" item 1a) are inside the first twenty
    assert keys[0] == "workload" and keys[3:7] == ["qc_verify_p50_ms_cold", "qc_verify_p50_ms_key_cache", "qc_verify_p50_ms_cold_host_api", "qc_verify_p50_ms_key_cache_host_api"]
    assert "tc_verify_p50_ms_cold" in keys[:20] and "cpu_tc_verify_p50_ms_one_core" in keys[:20] and "end_to_end_verifies_per_s_best" in keys[:20]
    assert out["value_host_to_host"] == out["config"]["end_to_end_verifies_per_s_best"]
    assert list(out)[:3] == ["metric", "value", "value_host_to_host"]
    json.dumps(out)


def test_nothing_the_record_would_cut():
    for k in bench.CONFIG_FIRST:
        assert len(k) <= bench.KEY_MAX, k
    bad = _line()
    bad["config"]["a_key_that_is_longer_than_the_forty_characters_kept"] = 1
    with pytest.raises(ValueError):
        bench.finalize_line(bad, strict=True)
    cut = bench.finalize_line(bad)          # the run itself never loses the line over a name: it cuts, and says so
    assert cut["line_cuts"] == 1 and "a_key_that_is_longer_than_the_forty_char" in cut["config"]
    bad = _line()
    bad["cpu_baseline"]["sample"] = "z" * 121
    with pytest.raises(ValueError):
        bench.finalize_line(bad, strict=True)
    assert len(bench.finalize_line(bad)["cpu_baseline"]["sample"]) == 120


def test_a_line_without_extras_is_still_a_line():
    line = _line()
    line["config"] = {"workload": "w", "entry_point": "e", "streams": 16}
    out = bench.finalize_line(line)
    assert list(out["config"]) == ["workload", "entry_point", "streams"] and "value_host_to_host" not in out


def test_every_key_the_source_writes_into_config_fits():
    """static: every literal `cfg["..."]` / cfgm key and every literal key of a cfg.update({...}) in bench.py is <= 40 characters"""
    import re
    src = open(os.path.join(ROOT, "bench.py")).read()
    keys = set(re.findall(r'cfgm?\[\s*"([a-z0-9_]+)"\s*\]', src))
    for block in re.findall(r'cfg\.update\(\{(.*?)\}\)', src, re.S):
        keys |= set(re.findall(r'"([a-z0-9_]+)"\s*:', block))
    for fmt in re.findall(r'cfg\[\s*"(sq_%s_[a-z0-9_%s]+)"\s*%', src):      # the per-kernel counter keys: the longest kernel and counter names they take
        keys.add(fmt.replace("%s", "verify_cached_lane2r", 1).replace("%s", "cycles_vmem_rd"))
    assert len(keys) > 40
    assert not [k for k in keys if len(k) > bench.KEY_MAX], [k for k in keys if len(k) > bench.KEY_MAX]
