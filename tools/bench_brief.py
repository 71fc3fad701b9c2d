"""One bench.py JSON line on stdin -> the few numbers an A/B visit compares."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
t = d["this_precision"]
print("fps %.2f  step %.1f ms  depth %.2f  flow %.2f  lat_b1 %s" % (d["value"], d["ms_per_step"], t["depth_ms_per_step"], t["flow_ms_per_step"], d.get("latency_720p_batch1_ms")))
k, tf = d.get("kernel_ms_per_step", {}), d.get("kernel_tflops", {})
for n, v in sorted(k.items(), key=lambda kv: -kv[1]):
    if v > 0.8:
        print("  %-75s %7.2f ms %s" % (n, v, ("%6.0f TF/s" % tf[n]) if n in tf else ""))
