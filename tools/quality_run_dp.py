"""What P-batches-per-update data parallelism does to what the loop learns: the cfg2-shaped synthetic
problem of tools/quality_run.py (2 time points x 5000 spots x 2000 genes, 10 generating domains per time point) trained
  (a) by the single-replica trainer (one optimizer step per batch, _train_utils.py:155-236), and
  (b) by two replicas sharing this GPU (gloo carries the collectives; batch-granular shard plan: every update = the mean
      gradient of 2 consecutive units of the reference's order, learning rate as configured),
and then the ARI between K-means labels of the final latent and the generating domains, and the loss curves, side by side.
No scaling claim is made from it (two ranks on one device time-slice); it backs 'P x effective batch, unscaled lr' with a number.

    python tools/quality_run_dp.py            (EPOCHS=30 by default)
"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EPOCHS = int(os.environ.get("EPOCHS", 30))
DEV = "cuda:0"


def _cfg(torch):
    from spadot_amd.utils import _utils
    cfg = _utils.load_model_config(types.SimpleNamespace(config=None))
    cfg.update(maxiter=EPOCHS, ot_epoch=max(2, EPOCHS // 2), compute_dtype=torch.bfloat16, input_dim=2000, timepoints=[0, 1],
               device=torch.device(DEV), kmeans_backend="device")
    return cfg


def _score(model, cfg, dd, data):
    """ARI per time point between K-means labels of the final latent (posterior means) and the generating domains."""
    import torch
    from sklearn.cluster import KMeans
    from sklearn.metrics import adjusted_rand_score
    dom, tps = np.asarray(data.obs["domain"]), np.asarray(data.obs["timepoint"])
    out = []
    model.eval()
    with torch.no_grad():
        for t in (0, 1):
            loc, Y, ix = dd["datasets"][t]
            z = model.all_latent_samples(loc, Y, dd["graphs"][t], t)
            lab = KMeans(10, n_init=10, random_state=0).fit_predict(z)
            out.append(float(adjusted_rand_score(dom[np.asarray(ix)], lab)))
    return out


def single():
    import torch
    from spadot_amd.synthetic import make_dataset
    from spadot_amd.utils import _train_utils as tu, _utils
    cfg = _cfg(torch)
    data = make_dataset(2, 5000, 2000, seed=1993)
    _utils.set_seed(cfg["seed"])
    dd = tu.prepare_dataloader(data, cfg)
    t0 = time.perf_counter()
    model, loss = tu.train_SpaDOT(dd, cfg, verbose=False)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L = loss.T if loss.shape[0] < loss.shape[1] else loss
    return _score(model, cfg, dd, data), el, L


def _rank(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spadot_amd import parallel as par
        from spadot_amd.synthetic import make_dataset
        from spadot_amd.utils import _train_utils as tu, _utils
        cfg = _cfg(torch)
        cfg["shard_granularity"] = "batch"
        data = make_dataset(2, 5000, 2000, seed=1993)
        par.configure_shard(data, cfg, world, rank)
        _utils.set_seed(cfg["seed"])
        dd = tu.prepare_dataloader(data, cfg)
        t0 = time.perf_counter()
        model, losses = par.train_SpaDOT_parallel(dd, cfg)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ari = _score(model, cfg, dd, data) if rank == 0 else None
        q.put((rank, ari, el, {e: [float(v) for v in np.asarray(l)] for e, l in losses.items()} if isinstance(losses, dict) else None))
    finally:
        dist.destroy_process_group()


def main():
    import socket
    import torch.multiprocessing as mp
    ari1, el1, L1 = single()
    print(f"P = 1: ARI {ari1[0]:.3f} / {ari1[1]:.3f}   ({el1:.1f} s for {EPOCHS} epochs)")
    print(L1.iloc[[0, 1, EPOCHS // 2, EPOCHS - 1]].round(3).to_string())
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=1500) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
    ari2, el2 = res[0][1], res[0][2]
    print(f"P = 2 (two ranks on one GPU, gloo, batch-granular plan): ARI {ari2[0]:.3f} / {ari2[1]:.3f}   ({el2:.1f} s for {EPOCHS} epochs; "
          "no scaling claim: the ranks time-slice one device)")
    ls = res[0][3]
    if ls:
        names = ["elbo", "Recon", "SVGP_KL", "GAT_KL", "alignment", "KMeans", "OT"]
        for e in (0, 1, EPOCHS // 2, EPOCHS - 1):
            if e in ls:
                print(f"  rank 0, epoch {e}: " + ", ".join(f"{n} {v:.3f}" for n, v in zip(names, ls[e])))


if __name__ == "__main__":
    main()
