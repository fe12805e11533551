"""Where the per-epoch _update_Kmeans goes after some training (cfg3): inference per time point, k-means++ seeding, Lloyd
iterations (count, seconds), final assignment + host copies."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from spadot_amd.model import SpaDOT
from spadot_amd import kmeans as km_mod
from spadot_amd.ops import FlatAdamW
dev = "cuda:0"
T, N, G = 5, 10000, 3000
cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
data = make_dataset(T, N, G, seed=1993)
cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=torch.bfloat16)
_utils.set_seed(cfg["seed"])
dd = tu.prepare_dataloader(data, cfg)
model = SpaDOT.SpaDOT(cfg, dd).to(dev)
opt = FlatAdamW(model.parameters(), lr=cfg["lr"])
st = tu.GraphedStepper(model, opt, cfg, dd)
st.clone_output = False
model.train()
for ep in range(3):
    for t in range(T):
        for bi in range(len(dd["dataloaders"][t])):
            st.step(t, t, bi, 0, 0.1)
torch.cuda.synchronize()
model.eval()
def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
with torch.no_grad():
    for _ in range(5):
        _, ms = timed(lambda: tu._update_Kmeans(model, cfg, dd)); print(f"_update_Kmeans: {ms:.1f} ms")
    loc, Y, ix = dd["datasets"][0]
    lat, ms = timed(lambda: model.all_latent_samples(loc, Y, dd["graphs"][0], 0, as_numpy=False)); print(f"all_latent_samples: {ms:.2f} ms")
    K = km_mod.KMeansDevice(cfg["n_clusters"], random_state=cfg["seed"], n_init=10)
    X = lat.to(torch.float64); mean = X.mean(0); Xc = (X - mean).contiguous(); xsq = (Xc * Xc).sum(1)
    import numpy as np
    seeds = np.random.RandomState(K.seed).randint(np.iinfo(np.int32).max, size=K.n_init)
    C, ms = timed(lambda: K._init_centers(Xc, xsq, seeds)); print(f"k-means++ seeding (10 restarts): {ms:.2f} ms")
    _, ms = timed(lambda: K.fit(lat)); print(f"fit total: {ms:.2f} ms, n_iter {K.n_iter_}")
