"""Replay time of each graph of the staged training step on its own (cfg3 shape, bf16): where the 2.4 ms go when
nothing overlaps, and how long the sequential `tail` section between the two parallel phases is."""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from spadot_amd.model import SpaDOT
from spadot_amd.ops import FlatAdamW
dev = "cuda:0"
T, N, G = 2, 10000, 3000
cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
data = make_dataset(T, N, G, seed=1993)
cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=torch.bfloat16, inducing_point_nums=480)
_utils.set_seed(cfg["seed"])
dd = tu.prepare_dataloader(data, cfg)
model = SpaDOT.SpaDOT(cfg, dd).to(dev)
opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
tu._update_Kmeans(model, cfg, dd); tu._update_OT_matrix(model, cfg)
model.train()
st = tu.GraphedStepper(model, opt, cfg, dd)
ep = cfg["ot_epoch"]
for _ in range(4): st.step(1, 1, 0, ep, 0.5)
torch.cuda.synchronize()
key = [k for k in st.graphs if k[-1] == "staged"][0]
graphs, _ = st.graphs[key]
names = ["gat_fwd", "svgp_fwd", "tail", "svgp_bwd", "gat_bwd"] + ([] if len(graphs) == 5 else ["gat_bwd_lo"])
def timeit(fn, reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0.0
for n, g in zip(names, graphs):
    t = timeit(g.replay); tot += t
    print(f"{n:10s} {t:8.1f} us")
t = timeit(st.opt_graph.replay); print(f"{'optimizer':10s} {t:8.1f} us")
side = model._side_stream()
def pair(a, b):
    def run():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            b.replay()
        a.replay()
        main.wait_stream(side)
    return run
print(f"gat_fwd || svgp_fwd {timeit(pair(graphs[0], graphs[1])):8.1f} us   (alone: max of the two above)")
print(f"gat_bwd || svgp_bwd {timeit(pair(graphs[4], graphs[3])):8.1f} us")
print(f"sum of the five {tot:.0f} us; whole step {timeit(lambda: st.step(1, 1, 0, ep, 0.5), 30):.0f} us")
