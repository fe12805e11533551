"""Fixed cost of one fused Sinkhorn pass: J fixed at 10 000 (ld 10 048, 5 vectors per thread, 2 rows per group),
I varied so that every workgroup gets 2 ... 40 groups; a linear fit of time against bytes separates the per-launch
cost from the streaming rate."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.ot import OTSolver
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
J = 10000
rows = []
for I in [int(v) for v in os.environ.get("OT_I", "1024,2048,4096,6144,8192,10000,10240,20480").split(",")]:
    s = OTSolver(I, J, storage="f32", device=torch.device("cuda:0"))
    s.set_cost_from_latents(bench.synthetic_latents(I, 1), bench.synthetic_latents(J, 2))
    s.run_iterations(bench.OT_CFG, 0.05, 10, timed=False)
    kt = s.time_kernels(bench.OT_CFG, 0.05, reps=30)
    g = s.fused_geometry()
    mb = I * s.ld * 4 / 1e6
    rows.append((mb, kt["fused_pass"] * 1e3))
    print(f"I={I:6d} wgs={g['workgroups']} rows/wg={g['rows_per_workgroup']:3d}  {mb:7.1f} MB  pass {kt['fused_pass']*1e3:6.1f} us  "
          f"fin {kt['fused_col_fin']*1e3:5.1f} us  {mb/kt['fused_pass']/1e3:5.2f} TB/s", flush=True)
    s.close()
x, y = np.array(rows).T
k, c = np.polyfit(x, y, 1)
print(f"fit: {c:.1f} us per launch + bytes / {1/k/1e3*1e3/1e3:.2f} TB/s" if k > 0 else "fit failed")
