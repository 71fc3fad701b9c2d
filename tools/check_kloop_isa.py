"""ISA-level check of the GEMM kernels' matrix phases: for every kernel in a gfx950 .s file that contains MFMAs, counts the scratch
accesses (register spills) between its first and its last MFMA - the peeled first K tile and the K loops of the ping-pong kernel.  A
spill the register allocator places OUTSIDE that span (tile set-up, epilogue) costs a few instructions per 40k-cycle tile; one inside
costs 2-3x (round 2) - and in a persistent kernel the allocator sees set-up, K loop and epilogue as ONE loop body, so small spills
outside the span are the price of the structure (gemm8_kernel header).
usage: python tools/check_kloop_isa.py file.s [...]      prints one line per kernel; exit code 1 when a matrix phase holds a scratch access"""
import re, subprocess, sys


def kernels(path):
    name, body = None, []
    for l in open(path, errors="replace"):
        m = re.match(r"(_Z\w+):", l)
        if m:
            name, body = m.group(1), []
            continue
        if l.startswith(".Lfunc_end") and name:
            yield name, body
            name = None
        elif name:
            body.append(l)


def check(path, quiet=False):
    bad = []
    for name, body in kernels(path):
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        if not mf:
            continue
        inside = sum("scratch_" in l for l in body[mf[0]:mf[-1] + 1])
        total = sum("scratch_" in l for l in body)
        if not quiet:
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
            print(f"{dn[:90]:90s} scratch instructions: {total:3d} in the kernel, {inside} between its first and last MFMA")
        if inside:
            bad.append(name)
    return bad


if __name__ == "__main__":
    allbad = []
    for p in sys.argv[1:]:
        allbad += check(p)
    sys.exit(1 if allbad else 0)
