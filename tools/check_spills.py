"""Reads the -Rpass-analysis=kernel-resource-usage remarks the Makefile keeps per translation unit (prisma_amd/csrc/build/*.log), prints
one line per kernel and fails when a GEMM kernel spills vector registers or loses its second wave per SIMD: a spill in the K loop of the 256 x 256 kernel costs 2-3x
(round 2: a two-sided `if` in the conv tap cursor spilled 128 VGPRs in gemm8_kernel<1, 0, 0, *, true> and tripled the DPT head's time).
python tools/check_spills.py [build_dir]"""
import glob, os, re, subprocess, sys

bdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prisma_amd", "csrc", "build")
rows, cur = [], None
for f in sorted(glob.glob(os.path.join(bdir, "*.log"))):
    for line in open(f, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"sym": m.group(1), "tu": os.path.basename(f)[:-4]}
            rows.append(cur)
            continue
        for key, tag in (("vgpr", "VGPRs:"), ("agpr", "AGPRs:"), ("scratch", "ScratchSize [bytes/lane]:"), ("occ", "Occupancy [waves/SIMD]:"),
                         ("vspill", "VGPRs Spill:"), ("lds", "LDS Size [bytes/block]:")):
            if cur is not None and tag in line:
                cur[key] = int(re.search(r"(\d+)", line.split(tag)[1]).group(1))
if not rows:
    sys.exit("no resource-usage remarks under %s (run make in prisma_amd/csrc first)" % bdir)
names = subprocess.run(["c++filt"] + [r["sym"] for r in rows], capture_output=True, text=True).stdout.split("\n")
bad = []
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("(GemmArgs)", "").replace("void ", "")
    n = n if len(n) < 70 else n[:67] + "..."
    print("%-14s %-70s vgpr %3d agpr %3d scratch %4d occupancy %d" % (r["tu"], n, r.get("vgpr", -1), r.get("agpr", -1), r.get("scratch", -1), r.get("occ", -1)))
    # the persistent ping-pong kernel (round 4) is one loop body of set-up, K loop and epilogue to the register allocator: a few spills
    # around the epilogue are the price of keeping the next tile's set-up out of the K loop's registers - allowed up to 48 VGPRs as long as
    # the ISA shows none of them between the kernel's first and last MFMA (checked below); every other GEMM kernel: none
    # (the flat-addressed builds - 4th template argument false - are the fallback for operands beyond a buffer resource's 4 GB).
    # Limits = the measured values of the committed tree (15 buffer-addressed, 25 flat: build/resource_usage.txt) plus a small margin
    lim = (20 if re.match(r"gemm8_kernel<\d+, \d+, \d+, true", n) else 32) if "gemm8_kernel" in n else 2
    if ("gemm8_kernel" in n or "gemm_kernel" in n) and r.get("vspill", 0) > lim:
        bad.append((n, "%d VGPRs spilled" % r.get("vspill")))
    # the 128 x 128 / 256 x 64 / 256 x 32 tiles are built to run TWO workgroups per CU (4 waves each, <= 256 registers per lane), the ping-pong
    # kernel two waves per SIMD: an edit that pushes VGPRs + AGPRs past 256 silently halves their latency hiding (round 3: a few lines in the
    # prologue of gemm_kernel<128, 128, ..., MX> cost its second workgroup and 50 % of its speed) - fail the build instead
    # (known exception: the residual epilogue on the 128 x 128 MX tile - small-batch ViT launches only; the batch-32 path runs gemm8_kernel)
    if ("gemm8_kernel<" in n or "gemm_kernel<" in n) and r.get("occ", 2) < 2 and not re.match(r"gemm_kernel<128, 128, 2, 2, 0, 1, \w+, 2, true>", n):
        bad.append((n, "occupancy %d (vgpr %d + agpr %d)" % (r.get("occ", -1), r.get("vgpr", -1), r.get("agpr", -1))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_kloop_isa
isa = sorted(glob.glob(os.path.join(bdir, "gemm_i*-gfx950.s")))
for f in isa:
    for sym in check_kloop_isa.check(f, quiet=True):
        bad.append((sym, "scratch access between the first and last MFMA (%s)" % os.path.basename(f)))
for f in sorted(glob.glob(os.path.join(bdir, "*-gfx950.s"))):
    for sym, nm, na in check_kloop_isa.acc_shuffles(f):
        bad.append((sym, "%d accumulator moves beside %d MFMAs in one loop (%s): the accumulators cross the back edge in VGPRs" % (na, nm, os.path.basename(f))))
if not isa:
    bad.append(("gemm_i*.hip", "no ISA files under %s: the Makefile keeps them with -save-temps=obj" % bdir))
if bad:
    sys.exit("GEMM kernels with register problems: %s" % bad)
print("%d kernels; no GEMM kernel spills inside its matrix phases, more than its limit outside them, or drops below two waves per SIMD" % len(rows))
