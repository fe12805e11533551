"""End-to-end `spadot_amd.train` at the ONE workload the reference publishes a wall-clock for: the ChickenHeart tutorial
(/root/reference/examples/ChickenHeart.ipynb:221-236,505,522 -- 4 time points of 747 / 1966 / 1916 / 1967 spots, 2 954 genes,
default config: 1 200 inducing points, 100 epochs, 318 s of training on the authors' machine, hardware not stated;
BASELINE.md section 1).  The data here are synthetic of that shape (no .h5ad in the repository, no network), so the figure
beside the reference's is CONTEXT, not a same-node comparison.

    python tools/e2e_chickenheart.py [f32|bf16 ...]          (default: both)

Per compute dtype: total seconds of train() (data preparation, graph capture, 100 epochs, inference), seconds per epoch
in the first two epochs (eager visit, capture) and afterwards (median), the capture overhead, final ARI between K-means
labels of the latent and the generating domains per time point."""
import os, sys, time, types, tempfile
import numpy as np, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spadot_amd
from spadot_amd.synthetic import make_dataset
from spadot_amd.utils import _utils
from sklearn.cluster import KMeans
from sklearn.metrics import adjusted_rand_score

COUNTS = [747, 1966, 1916, 1967]
G = 2954
EPOCHS = int(os.environ.get("EPOCHS", 100))


def run(dtype):
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(maxiter=EPOCHS, ot_epoch=min(cfg["ot_epoch"], max(2, EPOCHS // 2)))
    data = make_dataset(len(COUNTS), COUNTS, G, seed=1993)
    secs = []
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "cfg.yaml")
        yaml.safe_dump({k: v for k, v in cfg.items() if k != "compute_dtype"}, open(p, "w"))
        args = types.SimpleNamespace(data=data, output_dir=os.path.join(d, "out"), prefix="ch_", config=p, save_model=False,
                                     device="cuda:0", compute_dtype="bfloat16" if dtype == "bf16" else "float32", epoch_seconds=secs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model, loss = spadot_amd.train(args)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        z = np.load(os.path.join(d, "out", "ch_latent.npz"))["X"]
    steady = float(np.median(secs[2:])) if len(secs) > 2 else float("nan")
    steps = sum(-(-c // cfg["batch_size"]) for c in COUNTS)
    m = {t: int(model.svgp_dict[str(t)].inducing_index_points.shape[0]) for t in range(len(COUNTS))}
    print(f"== {dtype}: {len(COUNTS)} time points {COUNTS} x {G} genes, {cfg['inducing_point_nums']} inducing points "
          f"(m per time point {list(m.values())}), {EPOCHS} epochs, {steps} steps per epoch")
    print(f"train() total {total:.1f} s; training loop {sum(secs):.1f} s; outside the loop (graph / batch preparation, final "
          f"inference, files) {total - sum(secs):.1f} s")
    print(f"epoch 0 (eager visit) {secs[0]:.2f} s, epoch 1 (capture) {secs[1]:.2f} s, epochs >= 2: median {steady * 1e3:.1f} ms "
          f"(min {min(secs[2:]) * 1e3:.1f}, max {max(secs[2:]) * 1e3:.1f}) = {steps / steady:.0f} steps/s whole-epoch")
    print(f"capture overhead (epochs 0 + 1 over two steady epochs): {secs[0] + secs[1] - 2 * steady:.2f} s")
    print(f"reference (context, other hardware, real data): 318 s for 100 epochs = 3.18 s per epoch; here {sum(secs) / EPOCHS:.3f} s "
          f"per epoch over the whole loop")
    L = loss.T if loss.shape[0] < loss.shape[1] else loss
    print(L.iloc[[0, 1, EPOCHS // 2, EPOCHS - 1]].round(3).to_string())
    dom = np.asarray(data.obs["domain"]); tp = np.asarray(data.obs["timepoint"])
    aris = []
    for t in range(len(COUNTS)):
        sel = tp == t
        lab = KMeans(10, n_init=10, random_state=0).fit_predict(z[sel])
        aris.append(adjusted_rand_score(dom[sel], lab))
    print("ARI(latent K-means, generating domains) per time point: " + ", ".join(f"{a:.3f}" for a in aris))
    return total


if __name__ == "__main__":
    for dt in (sys.argv[1:] or ["f32", "bf16"]):
        run(dt)
