"""Wall time of OTSolver.set_cost_from_latents at the cfg3 pair shape (10k x 10k, fp32 storage): the on-the-fly path
(ot_cost.hip) against round 1's materialised fp64 matrix + radix select (SPADOT_OT_COST_LEGACY=1)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.ot import OTSolver
rng = np.random.default_rng(0)
cen = rng.normal(size=(10, 20))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
x = torch.as_tensor(cen[rng.integers(0, 10, n)] + 0.3 * rng.normal(size=(n, 20)), device="cuda")
y = torch.as_tensor(cen[rng.integers(0, 10, n)] + 0.3 * rng.normal(size=(n, 20)), device="cuda")
s = OTSolver(n, n, storage="f32")
for mode in ("0", "1", "0"):
    os.environ["SPADOT_OT_COST_LEGACY"] = mode
    s.set_cost_from_latents(x, y); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): s.set_cost_from_latents(x, y)
    torch.cuda.synchronize()
    print(("legacy (fp64 matrix + 12 radix passes)" if mode == "1" else "on the fly (sampled bracket)       "),
          f"{(time.perf_counter() - t0) / 5 * 1e3:7.3f} ms per call")
    c = s.matrix("C") if n <= 4000 else None
