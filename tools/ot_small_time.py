"""Wall time of the training-size OT refresh (T - 1 pairs of 10 x 10 centres, _train_utils.py:309-321): the one-launch
small solver (csrc/ot_small.hip) against the streaming solver pair by pair, and against the C oracle on the host."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.ot import solve_small                       # noqa: E402
from spadot_amd.utils.OT_loss import ot_solvers             # noqa: E402

CFG = dict(lambda1=0.1, lambda2=5.0, epsilon=0.05, epsilon0=1.0, tolerance=1e-8, tau=1000.0, batch_size=5,
           max_iter=10 ** 7, growth_iters=3)
rng = np.random.default_rng(0)
cen = [rng.normal(size=(10, 20)) for _ in range(5)]
pairs = [(cen[i], cen[i + 1]) for i in range(4)]
for shape in ((10, 10), (32, 32), (64, 64)):
    pp = [(rng.normal(size=(shape[0], 20)), rng.normal(size=(shape[1], 20))) for _ in range(4)]
    solve_small(CFG, pairs=pp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        r = solve_small(CFG, pairs=pp)
    t = (time.perf_counter() - t0) / 20
    outs = [torch.zeros(shape, dtype=torch.float32, device="cuda:0") for _ in pp]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        solve_small(CFG, pairs=pp, gamma_out=outs, fetch=False)
    ev1.record()
    torch.cuda.synchronize()
    print(f"{shape}: 4 pairs, fetched {t * 1e3:.3f} ms per call; enqueued only {ev0.elapsed_time(ev1) / 20:.3f} ms of device time; "
          f"iters {sum(r.infos[0].stage_iters)}", flush=True)
ot_solvers.use_small_solver = False
t0 = time.perf_counter()
for a, b in pairs:
    ot_solvers.compute_transport_map(a, b, dict(CFG))
print(f"streaming solver, 4 pairs of 10 x 10: {(time.perf_counter() - t0) * 1e3:.2f} ms")
try:
    from oracle import ot_oracle
    t0 = time.perf_counter()
    for a, b in pairs:
        ot_oracle.compute_transport_map(a, b, CFG, all_growth_iters=False)
    print(f"C oracle on the host, 4 pairs: {(time.perf_counter() - t0) * 1e3:.3f} ms")
except Exception as ex:          # noqa: BLE001
    print("oracle not available:", ex)
