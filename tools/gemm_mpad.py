"""Does the library's bf16 GEMM get faster when the row count is rounded up?  Forward maps x [M, K] . W[N, K]^T, dgrad
g [M, N] . W [N, K] and wgrad g^T x for M around the step's 9980 / 8031 rows (events, median of 30, idle GPU)."""
import numpy as np, torch
dev = "cuda"
def bench(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e3
N = 2048
for K in (3072, 2048):
    W = torch.randn(N, K, device=dev).bfloat16()
    for M in (9980, 9984, 10112, 10240, 8031, 8064, 8192):
        x = torch.randn(M, K, device=dev).bfloat16(); g = torch.randn(M, N, device=dev).bfloat16()
        tf = bench(lambda: torch.nn.functional.linear(x, W))
        td = bench(lambda: g @ W)
        out = torch.empty(N, K, device=dev)
        tw = bench(lambda: torch.mm(g.t(), x, out_dtype=torch.float32, out=out))
        fl = 2.0 * M * N * K
        print(f"K={K} M={M:6d}: fwd {tf:6.1f} us ({fl/tf/1e6:5.0f} TF)  dgrad {td:6.1f} us ({fl/td/1e6:5.0f} TF)  wgrad {tw:6.1f} us ({fl/tw/1e6:5.0f} TF)", flush=True)
