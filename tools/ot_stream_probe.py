"""Sinkhorn iterations launched into torch's default stream (HIP's legacy null stream) against a stream of the caller's own:
50 iterations of the 10k x 10k fp32 pair problem, wall clock per iteration between synchronisations."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.ot import OTSolver

OT_CFG = dict(lambda1=0.1, lambda2=5.0, epsilon=0.05, epsilon0=1.0, tolerance=1e-8, tau=1000.0, batch_size=5, max_iter=10 ** 7,
              growth_iters=3)
N, ITERS = 10000, 50
rng = np.random.default_rng(0)
lx, ly = rng.normal(size=(N, 20)), rng.normal(size=(N, 20)) + 0.3


def run(label):
    sol = OTSolver(N, N, storage="f32", device="cuda:0")
    sol.set_cost_from_latents(lx, ly)
    sol.solve(OT_CFG)
    torch.cuda.synchronize()
    out = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            sol.run_iterations(OT_CFG, OT_CFG["epsilon"], ITERS // 5, timed=False)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e6 / ITERS)
    out.sort()
    print(f"{label:34s}: median {out[len(out) // 2]:.2f} us per iteration (min {out[0]:.2f})", flush=True)


run("default (null) stream")
extra = [torch.cuda.Stream() for _ in range(3)]          # other streams exist, as after a training leg
for s in extra:
    with torch.cuda.stream(s):
        torch.zeros(8, device="cuda:0").add_(1)
torch.cuda.synchronize()
run("default stream, 3 other streams")
with torch.cuda.stream(torch.cuda.Stream()):
    run("a stream of its own")
