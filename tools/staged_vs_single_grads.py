"""Diagnostic: flat gradient of the staged (six-graph) step against the single-graph step, parameter by parameter,
over several replays of several keys at the cfg3 shape (no optimizer updates, reparameterisation noise silenced)."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from spadot_amd.model import SpaDOT
from spadot_amd.ops import FlatAdamW
dev = "cuda:0"
T, N, G = 5, 10000, 3000
cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
data = make_dataset(T, N, G, seed=1993)
cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=torch.bfloat16, owned_timepoints=[0, 1])
_utils.set_seed(cfg["seed"])
dd = tu.prepare_dataloader(data, cfg)
model = SpaDOT.SpaDOT(cfg, dd).to(dev)
opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
tu._update_Kmeans(model, cfg, dd); tu._update_OT_matrix(model, cfg)
model.train()
model.fixed_noise = (torch.zeros((512, 10), device=dev), torch.zeros((512, 10), device=dev))   # same noise (none) in both modes
cfg_s = dict(cfg); cfg_s["staged_graphs"] = True
single = tu.GraphedStepper(model, opt, cfg, dd)
staged = tu.GraphedStepper(model, opt, cfg_s, dd)
names = {id(p): n for n, p in model.named_parameters()}
for rep in range(4):
    for bi in range(3):
        la = single.fb(1, 1, bi, cfg["ot_epoch"], 0.5); torch.cuda.synchronize(); ga = opt.flat_grad.clone()
        opt.flat_grad.fill_(7.0)
        lb = staged.fb(1, 1, bi, cfg["ot_epoch"], 0.5); torch.cuda.synchronize(); gb = opt.flat_grad.clone()
        print(rep, bi, "loss diff", float((la - lb).abs().max()), "grad finite", bool(torch.isfinite(ga).all()), bool(torch.isfinite(gb).all()))
        worst = []
        for p in opt.params:
            a, b = p.grad, None
        off = 0
        for p in opt.params:
            n = p.numel(); o = p.grad.data_ptr() - opt.flat_grad.data_ptr(); o //= 4
            da, db = ga[o:o + n], gb[o:o + n]
            if not torch.isfinite(db).all() or float((da - db).abs().max()) > 1e-2 * (float(da.abs().max()) + 1e-12):
                worst.append((names[id(p)], float(da.abs().max()), float((da - db).abs().max()) if torch.isfinite(db).all() else float("nan")))
        print("    suspicious:", worst[:6])
