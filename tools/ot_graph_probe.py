"""Do the ~4 us between dependent launches of the Sinkhorn iteration (eager, one stream) shrink when the same launches are
replayed as a hipGraph?  50 iterations of the 10k x 10k fp32 pair problem, eager against a captured graph, wall clock per
iteration between synchronisations."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.ot import OTSolver

OT_CFG = dict(epsilon=0.05, epsilon0=1.0, scaling_iter=3000, inner_iter_max=50, tolerance=1e-8, max_iter=1e7, batch_size=5,
              extra_iter=1000, lambda1=1.0, lambda2=50.0, tau=10000.0, growth_iters=1)
N, ITERS = 10000, 50
rng = np.random.default_rng(0)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    sol = OTSolver(N, N, storage="f32", device="cuda:0")
    sol.set_cost_from_latents(rng.normal(size=(N, 20)), rng.normal(size=(N, 20)) + 0.3)
    sol.solve(OT_CFG)
    torch.cuda.synchronize()

    def eager():
        sol.run_iterations(OT_CFG, OT_CFG["epsilon"], ITERS, timed=False)

    def wall(fn, reps=8):
        out = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) * 1e6 / ITERS)
        return sorted(out)

    eager(); eager()
    e = wall(eager)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        eager()
    g.replay(); torch.cuda.synchronize()
    r = wall(g.replay)
print(f"eager  : median {e[len(e) // 2]:.2f} us per iteration (min {e[0]:.2f})")
print(f"graph  : median {r[len(r) // 2]:.2f} us per iteration (min {r[0]:.2f})")
