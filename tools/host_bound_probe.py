"""Is the replayed training step bound by the host (graph launches + Python) or by the device?
Times N staged steps three ways at the cfg3 pair shape: host enqueue only (no sync inside), device (events), wall
with a sync per step; and the host cost of one replay of each of the step's graphs."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from spadot_amd.model import SpaDOT
from spadot_amd.ops import FlatAdamW
dev = "cuda:0"
cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
data = make_dataset(2, 10000, 3000, seed=1993)
cfg.update(input_dim=3000, timepoints=[0, 1], device=torch.device(dev), compute_dtype=torch.bfloat16, inducing_point_nums=480)
_utils.set_seed(cfg["seed"])
dd = tu.prepare_dataloader(data, cfg)
model = SpaDOT.SpaDOT(cfg, dd).to(dev)
opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
tu._update_Kmeans(model, cfg, dd); tu._update_OT_matrix(model, cfg)
model.train()
st = tu.GraphedStepper(model, opt, cfg, dd)
ep = cfg["ot_epoch"]
for _ in range(4):
    for bi in range(4): st.step(1, 1, bi, ep, 0.5)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for i in range(N): st.step(1, 1, i % 4, ep, 0.5)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host enqueue {1e3*t_host/N:.3f} ms/step   wall incl. drain {1e3*t_all/N:.3f} ms/step")
t0 = time.perf_counter()
for i in range(N):
    st.step(1, 1, i % 4, ep, 0.5); torch.cuda.synchronize()
print(f"with a sync per step {1e3*(time.perf_counter()-t0)/N:.3f} ms/step")
key = [k for k in st.graphs if k[-1] == "staged"][0]
graphs, _ = st.graphs[key]
names = ["gat_fwd", "svgp_fwd", "tail", "svgp_bwd", "gat_bwd"]
for n, g in zip(names, graphs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{n:10s} host {1e6*th/20:7.1f} us per replay call")
