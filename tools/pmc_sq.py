"""Mean SQ counter values per launch of one kernel from rocprofv3 --pmc counter_collection CSVs.
usage: python tools/pmc_sq.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "attnq_kernel"
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
wc = tot.get("SQ_WAVE_CYCLES", 0) / max(n.get("SQ_WAVE_CYCLES", 1), 1)
for k in sorted(tot):
    v = tot[k] / n[k]
    print(f"{k:34s} {v:16.0f}  launches {n[k]:3d}" + (f"  {v / wc:6.3f} of WAVE_CYCLES" if wc else ""))
