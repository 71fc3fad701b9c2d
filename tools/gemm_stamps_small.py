"""Does the epilogue get faster when only a few CUs are storing? (per-CU limit vs shared bandwidth)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PB_GEMM_DBG"] = "/tmp/gemm_dbg.bin"
from prisma_amd import engine
ops = engine.Ops(0)
for m in (256, 1024, 4096, 16384, 78336):
    for epi in (0, 2):
        ms = ops.gemm_bench(m, 4096 if epi == 0 else 1024, 1024, tile=2, epi=epi, iters=3)
        d = np.fromfile("/tmp/gemm_dbg.bin", dtype=np.int64).reshape(-1, 8)
        d = d[d[:, 3] != 0]
        print(f"M={m:6d} epi={epi} tiles={len(d):5d}: loop {np.median(d[:,2]-d[:,1]):7.0f}  epilogue issue {np.median(d[:,6]-d[:,2]):7.0f} drain {np.median(d[:,3]-d[:,6]):6.0f} prologue {np.median(d[:,1]-d[:,0]):6.0f}", flush=True)
