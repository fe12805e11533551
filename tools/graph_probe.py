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

# ---- the same two branches as TWO graphs replayed on two streams (real stream-level concurrency)
def time_it(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
torch.cuda.synchronize()
gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(gA):
    ra = branch_a()
with torch.cuda.graph(gB):
    rb = branch_b()
print(f"A alone {time_it(gA.replay):.3f} ms   B alone {time_it(gB.replay):.3f} ms")
def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gA.replay()
    gB.replay()
    main.wait_stream(side)
print(f"two graphs on two streams {time_it(both):.3f} ms")
def both_rev():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    gB.replay()
    with torch.cuda.stream(side):
        gA.replay()
    main.wait_stream(side)
print(f"two graphs, B launched first {time_it(both_rev):.3f} ms")
