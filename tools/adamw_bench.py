"""Clip + AdamW on a flat buffer of the cfg3 model's size (16.1 M parameters): rocprof-free timing of the two launches
(HIP events around a graph that replays them 20 times), against the bytes they must move."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.ops import FlatAdamW
n = 16_136_032
p = torch.nn.Parameter(torch.randn(n, device="cuda"))
opt = FlatAdamW([p], lr=3e-4)
opt.flat_grad.normal_()
for _ in range(3): opt.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): opt.step()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"clip + AdamW: {us:.1f} us per step; bytes: norm {n*4/1e6:.0f} MB + update {n*28/1e6:.0f} MB -> {n*32/us/1e6:.2f} TB/s")
