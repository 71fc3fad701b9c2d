"""Per-kernel-symbol effective shader clock from a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE` pass (MI355X_MICROARCH.md, DVFS recipe).

usage: python tools/pmc_clock.py <rocprof output dir> <out.json>

GRBM_GUI_ACTIVE counts busy cycles on each of the 8 XCDs and rocprofv3 reports their sum, so a dispatch's effective clock is
counter / 8 / (End_Timestamp - Start_Timestamp) (profiles/r04a_clock_mfma_peak.txt pinned the factor on a bare MFMA loop: 2.386 GHz on zeros).
Counter passes serialise the dispatches, so the figure is that of a kernel running alone (the sequential pass of bench.py)."""
import collections
import csv
import glob
import json
import sys


def main():
    d, out = sys.argv[1:3]
    cyc = collections.defaultdict(float); ns = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if dur <= 0:
                continue
            k = r["Kernel_Name"]
            cyc[k] += float(r["Counter_Value"]); ns[k] += dur; n[k] += 1
    res = {k: {"launches": n[k], "total_ms": ns[k] / 1e6, "effective_clock_ghz": cyc[k] / 8.0 / ns[k]} for k in cyc}
    res = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_ms"]))
    json.dump({"note": "effective_clock_ghz = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, summed over the symbol's dispatches of the pass", "kernels": res},
              open(out, "w"), indent=1)
    for k, v in list(res.items())[:24]:
        print(f"{k[:84]:84s} n={v['launches']:5d} {v['total_ms']:9.2f} ms  {v['effective_clock_ghz']:.3f} GHz")


if __name__ == "__main__":
    main()
