"""Which MFMA kernels still address global memory through flat 64-bit pointers?  (python tools/isa_census.py [build dir])
Per kernel of the build's ISA listings (prisma_amd/csrc/build/*-gfx950.s): global_store / global_load / buffer_store / flat_* counts,
64-bit VALU address instructions (v_mad_u64_u32, v_lshl_add_u64, v_mad_i64_i32), MFMAs, v_readfirstlane (a high count next to buffer
stores = waterfall loops around a descriptor the compiler left in VGPRs).  tools/probe/store_probe.hip prices a flat access beside a
matrix-busy wave at ~770 cycles against 35 for a buffer-addressed one; this census found the correlation volume's stores and the
pixel-shuffle epilogue (EXPERIMENTS.md 5.9, 5.10)."""
import glob
import os
import re
import sys

build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prisma_amd", "csrc", "build")
rows = []
for f in sorted(glob.glob(os.path.join(build, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
    parts = re.split(r"\n(_Z[\w]+):\s", open(f).read())
    for i in range(1, len(parts), 2):
        body = parts[i + 1].split("s_endpgm")[0]
        n = lambda pat: len(re.findall(pat, body))      # noqa: E731
        rows.append((os.path.basename(f).split("-")[0], parts[i], n(r"\bglobal_store"), n(r"\bglobal_load"), n(r"\bbuffer_store"), n(r"\bflat_(?:load|store)"),
                     n(r"v_mad_u64_u32|v_lshl_add_u64|v_mad_i64_i32"), n(r"v_mfma"), n(r"v_readfirstlane")))
print("%-14s %-78s %7s %6s %7s %5s %7s %5s %6s" % ("file", "kernel (mangled)", "gstore", "gload", "bstore", "flat", "addr64", "mfma", "rfl"))
for r in sorted(rows, key=lambda r: -(r[2] + r[5])):
    if r[7] > 0 and r[2] + r[3] + r[5] > 0:
        print("%-14s %-78s %7d %6d %7d %5d %7d %5d %6d" % ((r[0], r[1][:78]) + r[2:]))
