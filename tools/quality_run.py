"""End-to-end sanity of what the training loop learns: synthetic cfg2-shaped data (2 time points x 5000 spots x 2000
genes, 10 spatial domains per time point), `spadot_amd.train`, then the ARI between K-means labels of the final
latent and the generating domains, per time point.  Prints wall time per epoch as well."""
import os, sys, time, types, tempfile, numpy as np, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spadot_amd
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _utils
from sklearn.cluster import KMeans
from sklearn.metrics import adjusted_rand_score
epochs = int(os.environ.get("EPOCHS", 30))
cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
cfg.update(maxiter=epochs, ot_epoch=max(2, epochs // 2), compute_dtype="bf16", kmeans_backend=os.environ.get("KMEANS", "device"))
data = make_dataset(2, 5000, 2000, seed=1993)
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "cfg.yaml"); yaml.safe_dump({k: v for k, v in cfg.items() if k != "compute_dtype"}, open(p, "w"))
    args = types.SimpleNamespace(data=data, output_dir=os.path.join(d, "out"), prefix="q_", config=p, save_model=False, device="cuda:0")
    t0 = time.perf_counter()
    model, loss = spadot_amd.train(args)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    z = np.load(os.path.join(d, "out", "q_latent.npz"))["X"]
print(f"train: {el:.1f} s for {epochs} epochs ({el / epochs * 1e3:.0f} ms per epoch incl. set-up)")
L = loss.T if loss.shape[0] < loss.shape[1] else loss
print(L.iloc[[0, 1, epochs // 2, epochs - 1]].round(3).to_string())
dom = np.asarray(data.obs["domain"]); tp = np.asarray(data.obs["timepoint"])
for t in (0, 1):
    sel = tp == t
    lab = KMeans(10, n_init=10, random_state=0).fit_predict(z[sel])
    print(f"time point {t}: ARI(latent K-means, generating domains) = {adjusted_rand_score(dom[sel], lab):.3f}")
