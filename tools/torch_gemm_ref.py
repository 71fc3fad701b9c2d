"""What the vendor library (hipBLASLt / rocBLAS through torch.matmul) reaches on the same shapes - a ceiling reference only."""
import time, torch
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False
def bench(m, n, k, iters=20):
    a = torch.randn(m, k, device="cuda", dtype=torch.float16); w = torch.randn(n, k, device="cuda", dtype=torch.float16) * 0.05
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * m * n * k / ms / 1e9
M = 32 * 2448
for name, m, n, k in [("qkv", M, 3072, 1024), ("fc1", M, 4096, 1024), ("fc2", M, 1024, 4096), ("proj", M, 1024, 1024), ("sq8k", 8192, 8192, 8192), ("sq16k", 16384, 16384, 16384)]:
    ms, tf = bench(m, n, k)
    print(f"{name:6s} M={m} N={n} K={k}: torch.matmul fp16 {ms:7.3f} ms {tf:7.1f} TF", flush=True)
