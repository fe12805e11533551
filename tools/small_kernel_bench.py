"""Standalone timings of the single-workgroup loss kernels (events, back to back)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd import ops
dev = "cuda"
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rng = np.random.default_rng(0)
K, D, N = 10, 20, 10000
labels = torch.as_tensor(rng.integers(0, K, N), device=dev)
cen = torch.randn(K, D, device=dev); prev = torch.randn(K, D, device=dev); gam = torch.rand(K, K, device=dev)
cl = torch.arange(K, device=dev)
for b in (64, 128, 256, 512, 1024):
    z = torch.randn(b, D, device=dev); seeds = torch.as_tensor(rng.choice(N, b, replace=False), device=dev)
    for km, ot in ((True, True), (True, False), (False, True)):
        t = timeit(lambda: ops.cluster_losses(z, labels, seeds, cen, prev, gam, cl, km, ot))
        print(f"cluster_losses fwd b={b} km={km} ot={ot}: {t:.1f} us (incl. 2 allocs + python)")
    zg = torch.randn(b, 20, device=dev); pm = torch.randn(b, 10, device=dev, dtype=torch.float64); pv = torch.rand(b, 10, device=dev, dtype=torch.float64) + 0.1
    eps = torch.randn(b, 20, device=dev)
    print(f"latent_head fwd b={b}: {timeit(lambda: ops.latent_head(zg, pm, pv, eps, 10, 10)):.1f} us")
e = torch.empty(16, device=dev)
print(f"empty alloc+fill baseline: {timeit(lambda: torch.empty(16, device=dev).fill_(0)):.1f} us")
