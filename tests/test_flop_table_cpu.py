"""CPU: the per-launch FLOP table of DESIGN.md section 5 (tools/flop_table.py, derived from layer shapes alone) against what the engines
counted on the GPU (the newest profiles/r*_all_legs_bench_line.json: launches per step and TFLOP/s x ms per kernel symbol).  This is the
re-derivation the roofline line's `flop_per_launch` can be audited with."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest(suffix):
    """the newest committed profile set (names sort by round and letter: r02g > r02b > r01k)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??[a-z]_" + suffix)))
    assert files, suffix
    return files[-1]


def _table():
    spec = importlib.util.spec_from_file_location("flop_table", os.path.join(ROOT, "tools", "flop_table.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.totals()


def test_launch_counts_and_flops_match_the_profiled_bench_line():
    line = json.load(open(_newest("all_legs_bench_line.json")))
    launches, tflops, ms = line["kernel_launches_per_step"], line["kernel_tflops"], line["kernel_ms_per_step"]
    steps = line["steps"]
    table = _table()
    checked = 0
    for (band, sym), (n, flops, _) in table.items():
        key = f"{band}/{sym}"
        if sym == "attention":
            key = "depth/attention"
        assert key in launches, key
        # (the table already counts a 32-frame depth call as two launches of 16 frames per layer: DepthEngine::batch_cap in split mode)
        assert launches[key] == n, (key, launches[key], n)
        measured = tflops[key] * 1e12 * ms[key] * 1e-3            # FLOPs per step the engine counted
        assert measured == pytest.approx(flops, rel=5e-3), (key, measured, flops)
        checked += 1
    assert checked >= 20 and steps >= 1
    # every GEMM-shaped family the bench reports is in the table
    for key in tflops:
        band, sym = key.split("/", 1)
        assert (band, sym if sym != "attention" else "attention") in table, key


def test_roofline_flop_per_launch_is_the_table_entry():
    d = json.load(open(_newest("default_bench_line.json")))
    r = d["roofline"]
    band, sym = r["family"].split("/", 1)
    n, flops, by = _table()[(band, sym)]
    assert r["launches_per_step"] == n
    assert r["flop_per_launch"] == pytest.approx(flops / n, rel=5e-3)
    assert r["algorithmic_bytes"] == pytest.approx(by / n, rel=0.05)


def test_no_family_of_a_committed_bench_line_exceeds_the_hardware():
    """the bookkeeping the roofline rests on - no kernel family of the newest committed bench lines may claim more
    than the HBM's 8 TB/s of algorithmic bytes or the 2.5 PFLOP/s dense fp16 MFMA peak (MI355X_MICROARCH.md)."""
    for suffix in ("default_bench_line.json", "all_legs_bench_line.json"):
        line = json.load(open(_newest(suffix)))
        for k, v in line.get("kernel_algorithmic_tbps", {}).items():
            assert 0 < v <= 8.0, (suffix, k, v)
        for k, v in line.get("kernel_tflops", {}).items():
            assert 0 <= v <= 2500.0, (suffix, k, v)
