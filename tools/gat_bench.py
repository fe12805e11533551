"""Microbenchmark of the GAT edge kernels at the cfg3 batch shape (10k spots, k=30, H=4, C=512, bf16):
ms per launch of logits+forward and of the whole backward, on a batch graph built like training builds it."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd import ops, _lib
from spadot_amd.graph import knn_graph, precompute_batches
dev = "cuda"
rng = np.random.default_rng(0)
n, k, H, C = 10000, 30, 4, 512
side = int(np.sqrt(n))
coords = np.stack(np.meshgrid(np.arange(side), np.arange(side)), -1).reshape(-1, 2) + rng.uniform(-0.3, 0.3, (n, 2))
coords = coords[rng.permutation(n)]
ei = knn_graph(coords, k)
batch = precompute_batches(ei, n, 512, dev, coords=coords, plans=True)[3]
g = batch.graph
print("nodes", g.n, "edges", g.E)
print("plan_t: blocks", g.plan_t.nb, "avg distinct columns", round(g.plan_t.avg_cols, 1), "max", g.plan_t.max_cols)
for mfma in (True, False):
  ops.GAT_MFMA[0] = mfma
  print("matrix-core path" if mfma else "per-edge kernels")
  for dt in (torch.bfloat16,):
      h = (torch.randn((g.n, H * C), device=dev) * 0.5).to(dt).requires_grad_(True)
      a_s = (torch.randn((1, H, C), device=dev) * 0.1).requires_grad_(True)
      a_d = (torch.randn((1, H, C), device=dev) * 0.1).requires_grad_(True)
      bias = torch.zeros(H * C, device=dev, requires_grad=True)
      w = torch.randn((g.n, H * C), device=dev).to(dt)
      def fwd():
          return ops.gat_edge(h, a_s, a_d, bias, g, H, C, True, True)
      REPS = int(os.environ.get("GAT_BENCH_REPS", "100"))
      for _ in range(min(30, REPS)): out = fwd(); out.backward(w)
      torch.cuda.synchronize()
      ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
      tf, tb = [], []
      for _ in range(REPS):
          ev[0].record(); out = fwd(); ev[1].record(); out.backward(w); ev[2].record()
          torch.cuda.synchronize()
          tf.append(ev[0].elapsed_time(ev[1])); tb.append(ev[1].elapsed_time(ev[2]))
      print(dt, f"forward median {np.median(tf)*1e3:.0f} us (min {min(tf)*1e3:.0f})   backward median {np.median(tb)*1e3:.0f} us (min {min(tb)*1e3:.0f})")
