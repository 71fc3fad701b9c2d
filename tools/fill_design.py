"""Fills the @@...@@ fields of DESIGN.md's measurement sections and README.md's status paragraph from a bench line (profiles/<tag>_default_bench_line.json and
<tag>_all_legs_bench_line.json).  The section texts live in docs_src/ so that a new final visit regenerates the numbers instead of hand edits.
python tools/fill_design.py <tag>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(P, tag + "_default_bench_line.json")))
a = json.load(open(os.path.join(P, tag + "_all_legs_bench_line.json")))
r, t = d["roofline"], d["this_precision"]
WHAT = {
    "flow/gemm8_kernel<1, 0, 0, true, true>": "update block: convc1, convc2 (N = 192), GRU z \\| r, flow head conv1, mask.0 - the roofline kernel",
    "flow/gemm_kernel<128, 128, 2, 2, 1, 0, true, 2, true>": "N <= 128: GRU q, motion conv, encoder layers 2 / 3 of fnet and cnet",
    "depth/gemm8_kernel<0, 1, 0, true, true>": "ViT proj + fc2 (fp32 residual epilogue)",
    "depth/attention": "`attnq_kernel<1, 2, 0, false, 8>`",
    "flow/elementwise": "instance-norm apply 7.3 / stats 2.5, correlation lookup 9.0, pooling, upsample, state init",
    "depth/gemm8_kernel<0, 0, 0, true, true>": "ViT fc1 + GELU, DPT 1 x 1 projections / out_convs",
    "depth/gemm8_kernel<1, 0, 0, true, true>": "DPT head 3 x 3 convolutions with N = 256",
    "flow/conv3x3_c64_mx2_kernel": "encoder stage 1 (halo-tiled direct 3 x 3, 64 -> 64 at 1/2 resolution)",
    "flow/corr_volume_kernel": "all-pairs correlation, 4 pyramid levels x 31 pairs (27.7 GB written; stores 6.2 ms + matrix work and DMA 8.0 ms that do not overlap)",
    "depth/gemm8_kernel<0, 2, 0, true, false>": "ViT qkv",
    "depth/gemm_kernel<128, 128, 2, 2, 1, 0, true, 2, true>": "DPT head N <= 128 and low-resolution convolutions (output_conv1, layer3/4_rn, refinenet3/4)",
    "depth/elementwise": "bilinear upsampling of the head's split maps",
    "depth/layernorm": "52 LayerNorms (HBM-bound at 5.8 TB/s)",
    "depth/gemm_kernel<256, 32, 4, 1, 1, 5, false, 2, true>": "output_conv2 3 x 3 + ReLU + 1 x 1 + ReLU (N = 32, reads the 3.9 GB upsampled map)",
    "flow/gemm8_kernel<1, 3, 0, true, true>": "encoder stems (7 x 7 s2 as space-to-depth 3 x 3, pixel-shuffle epilogue)",
    "flow/flow_head2_kernel<true>": "flow head conv2 3 x 3 256 -> 2 (direct, HBM-bound)",
    "flow/gemm_kernel<256, 64, 4, 1, 1, 0, true, 2, true>": "convf2 (N = 64)",
}
k, tf, ln = d["kernel_ms_per_step"], d["kernel_tflops"], d["kernel_launches_per_step"]
rows = []
for n, v in sorted(k.items(), key=lambda kv: -kv[1]):
    if v < 1.5:
        continue
    rows.append("| `%s` | %g | %.1f | %s | %s |" % (n, ln[n], v, ("%.0f" % tf[n]) if n in tf else "", WHAT.get(n, "")))
rest = sum(v for v in k.values() if v < 1.5)
rows.append("| (every symbol under 1.5 ms) | | %.1f | | pre / post-processing, convf1, patch embed, mask.2, small 1 x 1s |" % rest)
F = {"TAG": tag, "FPS": "%.1f" % d["value"], "MS": "%.1f" % d["ms_per_step"], "DMS": "%.1f" % t["depth_ms_per_step"], "FMS": "%.1f" % t["flow_ms_per_step"],
     "F16": "%.1f" % d["other_precision"]["value"], "PCIE": "%.1f" % d["pcie_inclusive_fps"], "LAT": "%.2f" % d["latency_720p_batch1_ms"],
     "RTF": "%.0f" % r["achieved"], "RFRAC": "%.3f" % r["frac"], "RATC": "%.2f" % r["frac_at_clock"], "CLK": "%.2f" % r["effective_clock_ghz"],
     "TRAF": "%.0f" % (r["traffic"] / 1e6), "DFRAC": "%.3f" % r["depth_frac_alone"], "FFRAC": "%.3f" % r["flow_frac_alone"], "SFRAC": "%.3f" % r["step_frac"],
     "CPU": "%.3f" % d["cpu_baseline"]["value"], "L720": "%.0f" % a["flow_raft_720p"]["value"], "LGM": "%.0f" % a["flow_gmflow"]["value"],
     "LMASK": "%.0f" % a["mask_mmdet"]["value"], "LPIPE": "%.1f" % a["pipeline"]["value"], "KTABLE": "\n".join(rows)}
src = os.path.join(ROOT, "docs_src")
parts = {n: open(os.path.join(src, n + ".md")).read() for n in ("design_sec0", "design_sec5", "design_sec7", "experiments_round5")}
def fill(s):
    return re.sub(r"@@([A-Z0-9]+)@@", lambda m: F[m.group(1)], s)
open(os.path.join(ROOT, "README.md"), "w").write(fill(open(os.path.join(src, "README.md.skeleton")).read()))
for path, subs in (("DESIGN.md", (("SEC0", "design_sec0"), ("SEC5", "design_sec5"), ("SEC7", "design_sec7"))), ("EXPERIMENTS.md", (("ROUND5", "experiments_round5"),))):
    body = open(os.path.join(src, path + ".skeleton")).read()
    for mark, name in subs:
        body = body.replace("@@" + mark + "@@", fill(parts[name]).rstrip("\n"))
    open(os.path.join(ROOT, path), "w").write(body)
    print(path, len(body.splitlines()), "lines")
