"""CPU: prisma_amd/power.py degrades to "nothing measured" on a box without a GPU (bench.py then reports null), parses amd-smi's JSON shape, and
windows its samples; engine.run_concurrently waits for the contexts it started when a later enqueue fails."""
import time

import pytest

from prisma_amd import power


def test_sampler_without_a_gpu_reports_nothing(monkeypatch):
    monkeypatch.setattr(power, "_hwmon_files", lambda device=0: None)
    monkeypatch.setattr(power, "_smi_sample", lambda: (None, None))
    with power.PowerSampler(interval_s=0.01) as ps:
        time.sleep(0.05)
    w = ps.window(0.0, 1e18)
    assert w["avg_power_w"] is None and w["samples"] == 0 and w["source"] == "amd-smi"


def test_sampler_reads_hwmon_files_and_windows(tmp_path, monkeypatch):
    pw, fq = tmp_path / "power1_input", tmp_path / "freq1_input"
    pw.write_text("1350000000\n"); fq.write_text("1900000000\n")
    monkeypatch.setattr(power, "_hwmon_files", lambda device=0: (str(pw), str(fq)))
    with power.PowerSampler(interval_s=0.005) as ps:
        a = time.perf_counter(); time.sleep(0.06); b = time.perf_counter()
        pw.write_text("400000000\n")
        time.sleep(0.06)
    first, late = ps.window(a, b), ps.window(b + 0.02, 1e18)
    assert first["source"] == "hwmon" and first["samples"] >= 3 and first["avg_power_w"] == 1350.0 and first["avg_sclk_mhz"] == 1900.0
    assert late["avg_power_w"] == 400.0


def test_run_concurrently_drains_started_contexts_when_an_enqueue_fails():
    from prisma_amd import engine
    log = []

    class Ctx:
        def __init__(self, name):
            self.name = name

        def sync(self):
            log.append("sync " + self.name)

    def boom():
        raise RuntimeError("enqueue failed")
    a, b, c = Ctx("a"), Ctx("b"), Ctx("c")
    with pytest.raises(RuntimeError, match="enqueue failed"):
        engine.run_concurrently([(a, lambda: log.append("enq a")), (b, boom), (c, lambda: log.append("enq c"))])
    assert log == ["enq a", "sync a", "sync b"]
    log.clear()
    done = engine.run_concurrently([(a, lambda: log.append("enq a")), (c, lambda: log.append("enq c"))])
    assert log == ["enq a", "enq c", "sync a", "sync c"] and len(done) == 2 and done[0] <= done[1]


def test_pmc_clock_summary_divides_by_the_xcds_and_the_duration(tmp_path):
    """tools/pmc_clock.py: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so a dispatch's clock is counter / 8 / duration; symbols are ranked by time."""
    import importlib.util
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / "run" / "host" ; d.mkdir(parents=True)
    rows = ["Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp",
            "gemmA,GRBM_GUI_ACTIVE,16000000,1000,1001000",        # 1 ms, 16e6 / 8 / 1e6 ns = 2.0 GHz
            "gemmA,GRBM_GUI_ACTIVE,12000000,2000000,3000000",      # 1 ms at 1.5 GHz -> symbol: 1.75 GHz over 2 ms
            "copyB,GRBM_GUI_ACTIVE,1920000,5000000,5100000",       # 0.1 ms at 2.4 GHz
            "gemmA,SQ_WAVES,5,1000,1001000"]
    (d / "1_counter_collection.csv").write_text("\n".join(rows) + "\n")
    spec = importlib.util.spec_from_file_location("pmc_clock", os.path.join(root, "tools", "pmc_clock.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    out = tmp_path / "clk.json"
    argv, sys.argv = sys.argv, ["pmc_clock.py", str(tmp_path / "run"), str(out)]
    try:
        m.main()
    finally:
        sys.argv = argv
    k = json.load(open(out))["kernels"]
    assert list(k) == ["gemmA", "copyB"]
    assert k["gemmA"]["launches"] == 2 and abs(k["gemmA"]["effective_clock_ghz"] - 1.75) < 1e-9 and abs(k["gemmA"]["total_ms"] - 2.0) < 1e-9
    assert abs(k["copyB"]["effective_clock_ghz"] - 2.4) < 1e-9
