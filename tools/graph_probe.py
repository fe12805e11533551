"""How does a replayed hipGraph schedule two independent branches?  Branch A: chain of 30 tiny kernels + one
long single-workgroup kernel (spd sweep) + 10 tiny; branch B: 6 big GEMMs.  Prints per-kernel start times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd import ops
dev = "cuda"
order = os.environ.get("ORDER", "A_first")
a = torch.randn(512, 64, device=dev)
X = torch.randn(10000, 2048, device=dev, dtype=torch.bfloat16); W = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
m = 236
S = torch.randn(20, m, m, device=dev, dtype=torch.float64); S = S @ S.transpose(1, 2) + m * torch.eye(m, device=dev, dtype=torch.float64)
side = torch.cuda.Stream()
def branch_a(n1=30, n2=10):
    t = a
    for _ in range(n1): t = t * 1.0001 + 0.5
    Xi, ld = ops.spd_inverse_logdet(S)
    t = t + ld.sum().float()
    for _ in range(n2): t = t * 0.999 - 0.1
    return t
def branch_b():
    h = X
    for _ in range(6): h = torch.nn.functional.linear(h, W) * 0.01
    return h
def step():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    if order == "A_first":
        with torch.cuda.stream(side): ra = branch_a()
        rb = branch_b()
    else:
        rb = branch_b()
        with torch.cuda.stream(side): ra = branch_a()
    main.wait_stream(side)
    return ra.sum() + rb.float().sum()
for _ in range(3): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
for _ in range(5): g.replay()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20): g.replay()
torch.cuda.synchronize()
print(order, f"{(time.perf_counter()-t0)/20*1e3:.3f} ms per replay")
