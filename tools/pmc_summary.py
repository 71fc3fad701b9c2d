"""Summarise rocprofv3 --pmc counter_collection CSVs into per-kernel HBM bytes per launch.

usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE and WRITE_SIZE need separate passes (TCC slot budget, MI355X_MICROARCH.md "rocprofv3 PMC
slots").  rocprofv3 reports both in KiB.  On gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes for
wide coalesced reads (same guide, HBM section), so the fetch figure is doubled; WRITE_SIZE is left as
reported (uncalibrated in the guide; it matches the known output byte counts of these kernels).
"""
import csv, glob, json, sys, collections


def per_kernel(d, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


def main():
    fd, wd, out = sys.argv[1:4]
    fe = per_kernel(fd, "FETCH_SIZE"); wr = per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(fe, key=lambda k: -fe[k][0] * fe[k][1]):
        f_kib, n = fe[k]; w_kib = wr.get(k, (0.0, 0))[0]
        res[k] = {"launches": n, "fetch_bytes": 2.0 * f_kib * 1024, "write_bytes": w_kib * 1024,
                  "fetch_raw_kib": f_kib, "write_raw_kib": w_kib}
    json.dump({"note": "per launch; fetch_bytes = 2 x FETCH_SIZE KiB (gfx950 correction), write_bytes = WRITE_SIZE KiB",
               "kernels": res}, open(out, "w"), indent=1)
    for k, v in list(res.items())[:24]:
        print(f"{k[:70]:70s} n={v['launches']:4d} fetch {v['fetch_bytes']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB")


if __name__ == "__main__":
    main()
