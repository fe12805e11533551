"""What a hipGraph boundary costs on this stack (ROCm 7.2, torch CUDAGraph.replay): chains of ~15 us kernels issued as
(A) eager launches, (B) two graphs of 10 alternating, (C) graph, eager kernel, graph, eager kernel, (D) graphs of 10 on two
streams joined by events after each (the staged step's pattern), (E) the same with three eager kernels in front of every
graph.  Prints us per kernel and the implied cost per boundary."""
import time, torch
dev = "cuda:0"
x = torch.zeros(6_000_000, device=dev)
y = torch.zeros(6_000_000, device=dev)
def k(t):
    t.mul_(1.0001)
for _ in range(20): k(x)
torch.cuda.synchronize()
def timed(fn, n=60):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def graph_of(t, n, stream=None):
    g = torch.cuda.CUDAGraph()
    s = stream or torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n): k(t)
    return g
base = timed(lambda: [k(x) for _ in range(20)])
print(f"A eager, 20 kernels:                 {base:8.1f} us  = {base / 20:5.2f} us per kernel")
g1, g2 = graph_of(x, 10), graph_of(x, 10)
tB = timed(lambda: (g1.replay(), g2.replay()))
print(f"B graph(10), graph(10):              {tB:8.1f} us  -> per boundary {(tB - base) / 2:6.1f} us")
g20 = graph_of(x, 20)
g20b = graph_of(x, 20)
tB2 = timed(lambda: (g20.replay(), g20b.replay())) / 2
print(f"B2 graph(20) alternating two execs:  {tB2:8.1f} us  -> per boundary {(tB2 - base):6.1f} us")
g9a, g9b = graph_of(x, 9), graph_of(x, 9)
tC = timed(lambda: (g9a.replay(), k(x), g9b.replay(), k(x)))
print(f"C graph(9), eager, graph(9), eager:  {tC:8.1f} us  -> per (graph, eager) pair {(tC - base) / 2:6.1f} us")
g7a, g7b = graph_of(x, 7), graph_of(x, 7)
tE = timed(lambda: (k(x), k(x), k(x), g7a.replay(), k(x), k(x), k(x), g7b.replay()))
print(f"E 3 eager + graph(7), twice:         {tE:8.1f} us  -> per group {(tE - base) / 2:6.1f} us")
# two streams joined after each pair of graphs
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    ga, gc = graph_of(x, 10, s1), graph_of(x, 10, s1)
with torch.cuda.stream(s2):
    gb, gd = graph_of(y, 10, s2), graph_of(y, 10, s2)
def pairs(pre=0):
    for A, B in ((ga, gb), (gc, gd)):
        s2.wait_stream(s1)
        with torch.cuda.stream(s1):
            for _ in range(pre): k(x)
            A.replay()
        with torch.cuda.stream(s2):
            for _ in range(pre): k(y)
            B.replay()
        s1.wait_stream(s2)
def run_pairs(pre):
    torch.cuda.current_stream().wait_stream(s1)
    with torch.cuda.stream(s1):
        pass
    pairs(pre)
    torch.cuda.current_stream().wait_stream(s1)
tD = timed(lambda: run_pairs(0))
print(f"D two streams, 2 x (graph(10) || graph(10)), joined: {tD:8.1f} us  (one stream's kernels alone: {base:.1f})")
tD3 = timed(lambda: run_pairs(3))
print(f"D3 the same with 3 eager kernels in front of each graph: {tD3:8.1f} us  (13 kernels per stream and pair: alone {base * 26 / 20:.1f})")
