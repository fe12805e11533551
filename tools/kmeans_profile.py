"""Where the per-epoch _update_Kmeans goes after some training (cfg3): inference per time point, the batched fit of all
time points (spadot_amd.kmeans.fit_many) against one KMeansDevice.fit per time point, host-side state updates."""
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
    for _ in range(4):
        _, ms = timed(lambda: tu._update_Kmeans(model, cfg, dd)); print(f"_update_Kmeans: {ms:.1f} ms", flush=True)
    lats = []
    for t in range(T):
        loc, Y, ix = dd["datasets"][t]
        lat, ms = timed(lambda: model.all_latent_samples(loc, Y, dd["graphs"][t], t, as_numpy=False)); print(f"all_latent_samples tp {t}: {ms:.2f} ms")
        lats.append(lat)
    for _ in range(3):
        fits, ms = timed(lambda: km_mod.fit_many(lats, cfg["n_clusters"], random_state=cfg["seed"], n_init=10)); print(f"fit_many (5 time points): {ms:.2f} ms, n_iter {fits[0].n_iter_}")
    K = km_mod.KMeansDevice(cfg["n_clusters"], random_state=cfg["seed"], n_init=10)
    for t in range(2):
        _, ms = timed(lambda: K.fit(lats[t])); print(f"KMeansDevice.fit tp {t}: {ms:.2f} ms, n_iter {K.n_iter_}")
    _, ms = timed(lambda: [tu._set_kmeans_state(model, t, fits[t].cluster_centers_, fits[t].labels_, dd["datasets"][t][2], cfg["device"]) for t in range(T)])
    print(f"_set_kmeans_state x 5: {ms:.2f} ms")
    from torch.profiler import profile, ProfilerActivity
    loc, Y, ix = dd["datasets"][1]
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model.all_latent_samples(loc, Y, dd["graphs"][1], 1, as_numpy=False)
        torch.cuda.synchronize()
    print("== all_latent_samples, one time point")
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
    # pieces of fit_many under the torch profiler's eyes: count launches
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        km_mod.fit_many(lats, cfg["n_clusters"], random_state=cfg["seed"], n_init=10)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
