"""PCIe-inclusive rate of the libot.so-compatible host ABI: the loop a reference user runs (ot_solvers.py:217-289:
per epsilon stage one update_k + one update_process call on HOST numpy buffers; every call moves its matrices over
PCIe both ways, as the C ABI's in/out pointers demand) at N x N fp64, beside the device-resident solver on the same
problem (one upload of C, one download of the plan).  Only the CALL PATTERN and the buffer traffic of the reference's
loop are reproduced here (what the time depends on); the parity of the entry points themselves is tests/test_ot_gpu.py."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from spadot_amd.utils.OT_loss import ot_func
from spadot_amd.ot import OTSolver
n = int(os.environ.get("N", 10000))
x, y = bench.synthetic_latents(n, 101), bench.synthetic_latents(n, 201)
xx, yy = (x * x).sum(1), (y * y).sum(1)
C = np.maximum(xx[:, None] + yy[None, :] - 2.0 * x @ y.T, 0.0)
C = np.ascontiguousarray(C / np.median(C))
cfg = dict(bench.OT_CFG)
lam1, lam2, eps_f, eps0, tau, bs, tol = (cfg[k] for k in ("lambda1", "lambda2", "epsilon", "epsilon0", "tau", "batch_size", "tolerance"))
I = J = n
G = np.ones(I)
torch.cuda.synchronize()
t0 = time.perf_counter()
# --- the reference's stage loop (ot_solvers.py:217-289) over the C entry points
scale = np.exp(-np.log(eps_f / eps0) / 5.0)
dx, dy = np.ones(I) / I, np.ones(J) / J
p, q = G.astype(np.float64), np.ones(J) * np.mean(G)
u, v = np.zeros(I), np.zeros(J)
a, b = np.ones(I), np.ones(J)
old_a, old_b = np.ones(I), np.ones(J)
K, K_, R = np.empty((I, J)), np.empty((I, J)), np.empty((I, J))
eps_i = eps0 * scale
stage_t = []
for e in range(6):
    ts = time.perf_counter()
    eps_i = eps_i / scale
    a1, a2 = lam1 / (lam1 + eps_i), lam2 / (lam2 + eps_i)
    thr = 1e-6 if e < 5 else tol
    if e:                                   # stage transition: scalings absorbed into the potentials
        u += eps_i * np.log(a); v += eps_i * np.log(b); a[:] = 1.0; b[:] = 1.0
    ot_func.update_K_c(K, K_, C, u, v, eps_i)
    ot_func.update_process_c(R, a, b, old_a, old_b, K, K_, C, dx, dy, p, q, u, v, 5, e, bs, eps_i, thr, tau,
                             lam1, lam2, a1, a2, 0, cfg["max_iter"])
    stage_t.append(time.perf_counter() - ts)
t_host = time.perf_counter() - t0
s = OTSolver(n, n, storage="f64", device=torch.device("cuda:0"))
t0 = time.perf_counter()
s.set_cost(C)
info = s.solve(cfg, G=G)
R2 = s.plan()
torch.cuda.synchronize()
t_dev = time.perf_counter() - t0
it = int(sum(info.stage_iters))
print(f"{n}x{n} fp64: host ABI, 6 stages {t_host:.2f} s (per stage {' '.join(f'{t:.2f}' for t in stage_t)}); "
      f"device-resident solver ({it} iterations) incl. one upload of C and one download of the plan {t_dev:.2f} s "
      f"= {it / t_dev:.0f} iters/s end to end")
