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


def acc_shuffles(path):
    """Round 4: every MX build of the generic GEMM tile carried its accumulators across the K loop's back edge in VGPRs and copied all 64 of
    them into the AGPRs and back each K tile (64 v_accvgpr_read + 128 v_accvgpr_write per 24 MFMAs; two tile kinds as branches of ONE loop did
    it) - invisible in the resource remarks, 15-25 % of those kernels' time.  Returns [(symbol, mfmas, moves)] for every loop (backward branch
    span) of under 1000 lines that holds MFMAs and at least twice as many accumulator moves; a tile loop that contains a whole epilogue (the
    halo kernels) is longer and reads each accumulator once, which is not this."""
    out = []
    for name, body in kernels(path):
        labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"\.LBB\d+_\d+:", l)}
        worst = None
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if not m or labels.get(m.group(1), i) >= i:
                continue
            seg = body[labels[m.group(1)]:i + 1]
            if any("s_endpgm" in x for x in seg):
                continue        # not a loop: a block the compiler laid out behind the kernel's end and that jumps back into it (round 6: the
                                # split-K store path - 64 accumulator reads in front of an s_endpgm - sat inside such a span next to the clean K loop)
            nm, na = sum("v_mfma" in x for x in seg), sum("v_accvgpr" in x for x in seg)
            if nm and len(seg) < 1000 and na >= 2 * nm and (worst is None or na > worst[2]):
                worst = (name, nm, na)
        if worst:
            out.append(worst)
    return out


if __name__ == "__main__":
    allbad = []
    for p in sys.argv[1:]:
        allbad += check(p)
        for sym, nm, na in acc_shuffles(p):
            print(f"{sym}: {na} accumulator moves beside {nm} MFMAs in one loop")
            allbad.append(sym)
    sys.exit(1 if allbad else 0)
