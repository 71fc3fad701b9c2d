#!/usr/bin/env python3
"""What a kx-shared A staging could save on the 3 x 3 / 1 x 5 convolutions at best, measured instead of modelled.

A -DPB_DIAG build of the GEMM kernels (make EXTRA=-DPB_DIAG BUILD=build_diag LIB=../libprisma_bands_diag.so, PRISMA_BANDS_LIB=<that .so>) knows
PB_GEMM_ABL=4: the A DMAs of every tap with kx != 0 fetch nothing (out-of-range offsets: zeros arrive in the LDS without an L2 / HBM access;
results are wrong; the K loop, the B staging, the fragment reads, the MFMAs and the epilogue are those of the shipped kernel).  Zeros also lower
the MFMAs' switching power, which a power-limited clock turns into speed - PB_GEMM_ABL=8 separates the two: the same taps read a 32 KB window at
the start of the operand (cache hits, random data), and PB_GEMM_ABL=16 does not issue their DMAs at all (the MFMAs multiply what an earlier K
tile left in that LDS stage).  Mode 16 is the UPPER bound of what sharing one (rows + 2)-pixel window between the kx taps of a (slice, ky) buys:
the real kernel would still pay the window's two extra rows, fragment reads at a row offset and the border masks.  The ping-pong kernel
(gemm8_kernel, 256 of 256 VGPRs) takes mode 4 only: the per-lane select of mode 8 and the branch of mode 16 spill inside its K loop.
Each arm loops one shape for ~0.8 s while prisma_amd/power.py samples the socket: ms per launch, TFLOP/s, watts, MHz, pJ per FLOP.

PRISMA_BANDS_LIB=prisma_amd/libprisma_bands_diag.so python tools/kx_share_model.py > gpurun_out/r06_kx_share_model.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from prisma_amd import engine  # noqa: E402
from prisma_amd.power import PowerSampler  # noqa: E402


def arm(ops, name, M, N, K, tile, epi, abl, min_s=0.8):
    os.environ["PB_GEMM_ABL"] = str(abl)
    call = lambda it: ops.gemm_bench(M, N, K, tile=tile, epi=epi, iters=it)      # noqa: E731
    ms = call(10)
    iters = max(20, int(min_s / max(ms * 1e-3, 1e-6)))
    with PowerSampler() as ps:
        ms = call(iters)
        b = time.perf_counter()
    w = ps.window(b - 0.85 * iters * ms * 1e-3, b)
    flops = 2.0 * M * N * K
    pw = w["avg_power_w"] or float("nan")
    print(f"{name:46s} abl {abl}  {ms:7.3f} ms  {flops / ms / 1e9:7.1f} TF/s  {pw:7.1f} W  {w['avg_sclk_mhz'] or float('nan'):7.1f} MHz  "
          f"{pw * ms * 1e-3 / flops * 1e12:5.2f} pJ/FLOP", flush=True)
    return ms


def main():
    ops = engine.Ops(0)
    R = 31 * 18360
    print("# A DMAs of the kx != 0 taps not issued (PB_GEMM_ABL=4, -DPB_DIAG build; wrong results) against the shipped kernel, fp16 operands")
    rows = [("gemm8 3x3 slice-major 569160 x 256 x 2304", R, 256, 2304, 2, 11), ("gemm8 3x3 tap-major   569160 x 256 x 2304", R, 256, 2304, 2, 10),
            ("gemm8 1x5            569160 x 256 x 1280", R, 256, 1280, 2, 12), ("128x128 3x3 slice-major 569160 x 128 x 2304", R, 128, 2304, 1, 11),
            ("128x128 1x5            569160 x 128 x 1280", R, 128, 1280, 1, 12)]
    for rep in range(2):
        for name, M, N, K, tile, epi in rows:
            a = arm(ops, name, M, N, K, tile, epi, 0)
            b = arm(ops, name, M, N, K, tile, epi, 4)
            if tile == 2:           # the ping-pong kernel has no register for modes 8 / 16 (they spill inside its K loop): mode 4 only
                print(f"{'':46s} -> {100.0 * (b / a - 1.0):+.1f} % time with the kx != 0 taps fetching nothing (zeros in the LDS)", flush=True)
                continue
            c = arm(ops, name, M, N, K, tile, epi, 8)
            d = arm(ops, name, M, N, K, tile, epi, 16)
            print(f"{'':46s} -> {100.0 * (b / a - 1.0):+.1f} % time with the kx != 0 taps fetching nothing (zeros), {100.0 * (c / a - 1.0):+.1f} % with them "
                  f"reading a cached 32 KB window (random data), {100.0 * (d / a - 1.0):+.1f} % with their DMAs not issued at all (stale random data)", flush=True)
    os.environ["PB_GEMM_ABL"] = "0"
    ops.close()


if __name__ == "__main__":
    main()
