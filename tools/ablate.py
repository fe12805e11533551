"""Exposed time of each part of the training step: the cfg3 step replayed as a hipGraph with one part
removed (ABLATE = none | svgp | gat | decoder | opt | tail).  Diagnostic only."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from spadot_amd.model import SpaDOT
from spadot_amd.ops import FlatAdamW, latent_head, sqerr_sum

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
mode = os.environ.get("ABLATE", "none")
b = 512
if mode in ("svgp", "gat", "decoder"):
    orig = SpaDOT.SpaDOT.forward
    const = {}
    def fwd(self, x, y, edge_index, tp, batch_size, noise=None, batch_key=None):
        bb = batch_size
        yb = y[:bb, :self.input_dim]
        Ls, Lg = self.SVGP_z_dim, self.GAT_z_dim
        if mode == "svgp":
            p_m = torch.zeros((bb, Ls), dtype=torch.float64, device=y.device); p_v = torch.ones_like(p_m)
            SVGP_KL = torch.zeros((), device=y.device)
            zg = self.GATEncoder.pre_head(y, edge_index, rows=bb)
        else:
            svgp = self.svgp_dict[str(tp)]
            z_enc = self.SVGPEncoder.pre_head(yb)
            bc = svgp.batch_constants(x[:bb], key=batch_key)
            p_m, p_v, l3_sum, kl_sum, ce = svgp.elbo_terms(bc, *((lambda zz: (zz[:, :Ls], torch.exp(zz[:, Ls:])))(z_enc)))
            SVGP_KL = (-torch.abs(ce - (l3_sum - (bb / float(svgp.N_train)) * kl_sum)) / Ls).float()
            if mode == "gat":
                zg = (self.GATEncoder.GAT_fc.weight.sum() * 0 + torch.zeros((bb, 2 * Lg), device=y.device))
            else:
                zg = self.GATEncoder.pre_head(y, edge_index, rows=bb)
        final_latent, GAT_KL, al = latent_head(zg, p_m, p_v, None, Ls, Lg, self._rng_state())
        if mode == "decoder":
            recon = (final_latent ** 2).sum()
        else:
            recon = sqerr_sum(yb.float(), self.decoder(final_latent), 1.0 / self.input_dim)
        return recon, SVGP_KL, GAT_KL, al, final_latent
    model.forward = types.MethodType(fwd, model)
if mode == "opt":
    opt.step = lambda: None
stepper = tu.GraphedStepper(model, opt, cfg, dd)
sched = [(1, bi) for bi in range(len(dd["dataloaders"][1]))]
epoch = cfg["ot_epoch"] if mode != "tail" else 0
def step(i):
    t, bi = sched[i % len(sched)]
    return stepper.step(t, t, bi, epoch, 0.5)
for rep in range(3):
    for i in range(20): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for rep in range(3):
    for i in range(20): step(i)
torch.cuda.synchronize()
print(f"ABLATE={mode}: {(time.perf_counter()-t0)/60*1e3:.3f} ms/step")
# host-side cost of one graph launch (no sync inside the timed call)
torch.cuda.synchronize()
hs = []
for i in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(i); hs.append(time.perf_counter() - t0)
torch.cuda.synchronize()
import numpy as np
print(f"host time per step() call (GPU idle at entry): median {np.median(hs)*1e3:.3f} ms, min {min(hs)*1e3:.3f} ms")
