"""Per-epoch work outside the training steps at cfg3 (5 x 10k spots x 3000 genes): full-time-point inference +
K-means fit per time point (host sklearn vs device), OT plans between consecutive time points."""
import os, sys, time, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _train_utils as tu, _utils
from spadot_amd.model import SpaDOT
dev = "cuda:0"
T, N, G = 5, 10000, 3000
cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
data = make_dataset(T, N, G, seed=1993)
cfg.update(input_dim=G, timepoints=list(range(T)), device=torch.device(dev), compute_dtype=torch.bfloat16)
_utils.set_seed(cfg["seed"])
t0 = time.perf_counter(); dd = tu.prepare_dataloader(data, cfg); torch.cuda.synchronize()
print(f"prepare_dataloader (graphs, batches, batch cache): {time.perf_counter()-t0:.2f} s")
model = SpaDOT.SpaDOT(cfg, dd).to(dev)
for backend in ("sklearn", "device", "sklearn", "device"):
    cfg["kmeans_backend"] = backend
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tu._update_Kmeans(model, cfg, dd)
    torch.cuda.synchronize()
    print(f"_update_Kmeans [{backend}]: {time.perf_counter()-t0:.3f} s for {T} time points")
t0 = time.perf_counter(); tu._update_OT_matrix(model, cfg); torch.cuda.synchronize()
print(f"_update_OT_matrix: {time.perf_counter()-t0:.3f} s for {T-1} pairs")
model.eval()
with torch.no_grad():
    loc, Y, ix = dd["datasets"][0]
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lat = model.all_latent_samples(loc, Y, dd["graphs"][0], 0, as_numpy=False); torch.cuda.synchronize()
        print(f"all_latent_samples (one time point, 10k spots): {(time.perf_counter()-t0)*1e3:.1f} ms")
    from spadot_amd.kmeans import KMeansDevice
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        km = KMeansDevice(cfg["n_clusters"], random_state=cfg["seed"], n_init=10).fit(lat); torch.cuda.synchronize()
        print(f"KMeansDevice.fit (10 restarts, 10k x 20): {(time.perf_counter()-t0)*1e3:.1f} ms, n_iter {getattr(km, 'n_iter_', None)}")
