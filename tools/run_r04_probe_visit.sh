#!/bin/bash
# Round-4 first GPU visit (through gpurun): settles the matrix-pipe ceiling and clock (tools/probe/mfma_peak.hip, both clock readings
# + rocm-smi samples), prices the epilogue access patterns (tools/probe/store_probe.hip), and records this box's baseline (bench line,
# per-tile stamps of the ping-pong GEMM) before any kernel of the round changes.
# usage: bash tools/run_r04_probe_visit.sh <tag>      -> gpurun_out/<tag>_*
set -u
T=${1:-r04a}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=$PWD/gpurun_out
mkdir -p $O
SMI=$(command -v amd-smi || command -v rocm-smi)
echo "smi tool: $SMI" > $O/${T}_clock_smi_samples.txt
sample() {    # power / clock samples while a probe runs
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    if [[ "$SMI" == *amd-smi ]]; then timeout 5 $SMI metric -g 0 --clock --power 2>&1 | grep -iE "socket_power|clk|GFX_0|clock" | head -12 | tr '\n' ' '; echo;
    else timeout 5 $SMI --showclocks --showpower 2>&1 | grep -iE "sclk|power" | tr '\n' ' '; echo; fi
    sleep 0.5
  done
}
( sample >> $O/${T}_clock_smi_samples.txt 2>&1 ) &
timeout 120 tools/probe/mfma_peak.bin 2.0 2 > $O/${T}_clock_mfma_peak.txt 2>&1
wait
timeout 60 tools/probe/mfma_peak.bin 1.0 1 >> $O/${T}_clock_mfma_peak.txt 2>&1
# the guide's clock: GRBM_GUI_ACTIVE / kernel wall time (own counter run, kernel trace only)
( cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/${T}_grbm --output-format csv -- $OLDPWD/tools/probe/mfma_peak.bin 1.0 2 > $O/${T}_grbm.log 2>&1 )
python - <<EOF > $O/${T}_clock_grbm.txt 2>&1
import csv, glob
kt = {}
for f in glob.glob("$O/${T}_grbm/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kt[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for f in glob.glob("$O/${T}_grbm/**/*counter_collection.csv", recursive=True):
    rd = csv.DictReader(open(f))
    print("# columns:", rd.fieldnames)
    for r in rd:
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        ns = kt.get(r["Dispatch_Id"], 0)
        if not ns and "End_Timestamp" in r: ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if ns > 5e7:
            v = float(r["Counter_Value"])
            print(f"dispatch {r['Dispatch_Id']} {r['Kernel_Name'][:30]}: GRBM_GUI_ACTIVE {v:.4g} / {ns*1e-6:.1f} ms = {v/ns:.3f} GHz (if the counter is summed over the 8 XCDs: {v/ns/8:.3f})")
EOF
rm -rf $O/${T}_grbm
timeout 300 tools/probe/store_probe.bin > $O/${T}_store_probe.txt 2>&1
timeout 600 python bench.py > $O/${T}_bench.log 2> $O/${T}_bench.err
tail -1 $O/${T}_bench.log > $O/${T}_default_bench_line.json
timeout 300 python tools/gemm_stamps.py > $O/${T}_gemm8_phase_cycles.log 2>&1
cat $O/${T}_clock_mfma_peak.txt $O/${T}_clock_grbm.txt
head -4 $O/${T}_clock_smi_samples.txt
cat $O/${T}_store_probe.txt
cut -c1-300 $O/${T}_default_bench_line.json
cat $O/${T}_gemm8_phase_cycles.log
