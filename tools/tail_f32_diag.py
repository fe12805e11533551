"""Diagnostic (round 5): relative L2 / cosine of the aggregate-first last layer's fp32 gradients against the fp64 oracle at
the cfg2 and cfg3 layer shapes (tests/test_step_parity_gpu.py found gat3.att_dst at 1.8e-3 for cfg2, 1e-5 for cfg3)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gat_tail_gpu as tg
from spadot_amd import ops
for (n, rows, k, H, C, K, scale) in [(4500, 512, 30, 4, 512, 2048, 1.0), (8000, 512, 30, 4, 512, 2048, 1.0), (4500, 512, 30, 4, 512, 2048, 0.05),
                                     (4500, 500, 30, 4, 512, 2048, 1.0)]:
    ei, x, W, a_s, a_d, bias, gsel = tg._problem(n, rows, k, H, C, K, seed=3, hub=False)
    a_s, a_d = a_s * scale, a_d * scale
    out_o, grads_o = tg._oracle(ei, x, W, a_s, a_d, bias, gsel, H, rows)
    out_d, grads_d = tg._tail(ops, ei, x, W, a_s, a_d, bias, gsel, H, C, rows, torch.float32)
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    print(f"n={n} rows={rows} att scale {scale}: out {rel(out_d, out_o):.2e} | " + " ".join(
        f"{nm} {rel(d[:n] if nm == 'x' else d, o):.2e}" for nm, d, o in zip(("x", "W", "att_src", "att_dst", "bias"), grads_d, grads_o)), flush=True)
