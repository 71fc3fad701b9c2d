#!/usr/bin/env python3
"""Socket power and shader clock of the bench's dominant kernel shapes, one at a time (is "the part is power-limited, so costs
add" a measurement or an assumption?).

Each arm loops ONE kernel shape on the whole chip for ~0.7 s (pb_op_gemm_bench / pb_op_attention_bench: random fp16 operands) while
prisma_amd/power.py samples this GPU's hwmon power1_input / freq1_input every 10 ms, and prints ms per launch, TFLOP/s, watts, MHz and
picojoules per algorithmic FLOP.  The last two arms are whole bands (depth_anything on 16 frames, flow_raft on 15 pairs of the bench's clip)
and the overlapped step comes from tools/overlap_bench.py / the bench line's avg_power_w.

python tools/power_clock.py > gpurun_out/r06_power_clock.txt"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from prisma_amd import engine  # noqa: E402
from prisma_amd.power import PowerSampler  # noqa: E402


def arm(name, flops, call, min_s=0.8):
    """`call(iters)` runs `iters` back-to-back launches inside ONE op call (operands are allocated and filled once per call) and returns ms per launch"""
    ms = call(10)                                     # warm + a first estimate
    iters = max(20, int(min_s / max(ms * 1e-3, 1e-6)))
    with PowerSampler() as ps:
        a = time.perf_counter()
        ms = call(iters)
        b = time.perf_counter()
    # the op's own set-up (allocation, fill, two warm launches) sits at the start of the window: keep the last (iters x ms) seconds of it
    w = ps.window(b - 0.85 * iters * ms * 1e-3, b)
    tf = flops / (ms * 1e-3) / 1e12 if flops else 0.0
    pj = w["avg_power_w"] * ms * 1e-3 / flops * 1e12 if flops and w["avg_power_w"] else float("nan")
    print(f"{name:58s} {ms:8.3f} ms  {tf:7.1f} TF/s  {w['avg_power_w'] or float('nan'):7.1f} W  {w['avg_sclk_mhz'] or float('nan'):7.1f} MHz  "
          f"{pj:6.2f} pJ/FLOP  ({w['samples']} samples, {iters} launches)", flush=True)


def main():
    ops = engine.Ops(0)
    M = 32 * 2448
    print("# one kernel shape at a time on the whole chip; power / clock = hwmon of this GPU over the arm (first 15 % dropped)")
    print("# idle:", end=" ")
    with PowerSampler() as ps:
        a = time.perf_counter(); time.sleep(0.5); b = time.perf_counter()
    print(ps.window(a, b))
    G = lambda M, N, K, tile, epi: (lambda it: ops.gemm_bench(M, N, K, tile=tile, epi=epi, iters=it))      # noqa: E731
    arm("gemm8 dense fc1 + GELU   78336 x 4096 x 1024", 2.0 * M * 4096 * 1024, G(M, 4096, 1024, 2, 1))
    arm("gemm8 dense qkv-like     78336 x 3072 x 1024", 2.0 * M * 3072 * 1024, G(M, 3072, 1024, 2, 0))
    arm("gemm8 dense proj + resid 78336 x 1024 x 1024", 2.0 * M * 1024 * 1024, G(M, 1024, 1024, 2, 2))
    arm("gemm8 dense fc2 + resid  78336 x 1024 x 4096", 2.0 * M * 1024 * 4096, G(M, 1024, 4096, 2, 2))
    R = 31 * 18360
    arm("gemm8 conv 3x3 slice-major 569160 x 256 x 2304", 2.0 * R * 256 * 2304, G(R, 256, 2304, 2, 11))
    arm("gemm8 conv 1x5 GRU       569160 x 256 x 1280", 2.0 * R * 256 * 1280, G(R, 256, 1280, 2, 12))
    arm("generic 128x128 conv 3x3 569160 x 128 x 2304", 2.0 * R * 128 * 2304, G(R, 128, 2304, 1, 11))
    arm("generic 128x128 conv 1x5 569160 x 128 x 1280", 2.0 * R * 128 * 1280, G(R, 128, 1280, 1, 12))
    arm("attention 32 frames x 16 heads x 2443 tokens", 4.0 * 32 * 16 * 2443 * 2443 * 64, lambda it: ops.attention_bench(32, 16, 2443, variant=0, iters=it))
    ops.close()


if __name__ == "__main__":
    main()
