"""Synthetic spatio-temporal transcriptomics of the shape SURVEY.md 8(d) prescribes (the reference's
.h5ad inputs are not redistributable and there is no network): per time point a jittered sqrt(N) x
sqrt(N) grid of spots, 10 spatial domains (k-means of the coordinates), expression = domain one-hot @
W + 0.5 * noise, z-scored per gene (what sc.pp.scale leaves, _preprocess_utils.py:49)."""
import numpy as np


class SpatialData:
    """Minimal stand-in for the AnnData fields SpaDOT.train reads (train.py:18-24,
    _train_utils.py:118-140): X (dense, N x G), obs['timepoint'], obsm['spatial']."""

    def __init__(self, X, timepoint, spatial):
        self.X = X
        self.obs = {"timepoint": np.asarray(timepoint)}
        self.obsm = {"spatial": np.asarray(spatial, dtype=np.float64)}
        self.n_obs, self.n_vars = X.shape


def make_timepoint(n_spots, n_genes, seed, n_domains=10, shuffle=True, dtype=np.float32):
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n_spots)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel(), gy.ravel()], 1)[:n_spots].astype(np.float64)
    xy += rng.uniform(-0.3, 0.3, size=xy.shape)
    # spatial domains: a few Lloyd iterations on the coordinates
    cen = xy[rng.choice(n_spots, n_domains, replace=False)]
    for _ in range(5):
        lab = ((xy[:, None, :] - cen[None]) ** 2).sum(-1).argmin(1)
        for c in range(n_domains):
            if np.any(lab == c):
                cen[c] = xy[lab == c].mean(0)
    W = rng.normal(size=(n_domains, n_genes)).astype(dtype)
    Y = W[lab] + 0.5 * rng.standard_normal(size=(n_spots, n_genes), dtype=dtype)
    Y -= Y.mean(0, keepdims=True)
    Y /= Y.std(0, keepdims=True) + 1e-12
    if shuffle:   # spot order of real data sets is not spatial; see DESIGN.md "batches"
        perm = rng.permutation(n_spots)
        xy, Y, lab = xy[perm], Y[perm], lab[perm]
    return xy, np.ascontiguousarray(Y, dtype=dtype), lab


def make_dataset(n_timepoints, spots_per_tp, n_genes, seed=1993, shuffle=True):
    """spots_per_tp: one count for every time point, or a list of n_timepoints counts (ragged time points, like the
    747 / 1966 / 1916 / 1967 spots of the reference's ChickenHeart tutorial, examples/ChickenHeart.ipynb:221-230)."""
    counts = [int(spots_per_tp)] * n_timepoints if np.isscalar(spots_per_tp) else [int(c) for c in spots_per_tp]
    if len(counts) != n_timepoints:
        raise ValueError("spots_per_tp must be a count or one count per time point")
    Xs, tps, locs, doms = [], [], [], []
    for t in range(n_timepoints):
        xy, Y, lab = make_timepoint(counts[t], n_genes, seed + 17 * t, shuffle=shuffle)
        Xs.append(Y); locs.append(xy); doms.append(lab)
        tps.append(np.full(counts[t], t))
    data = SpatialData(np.concatenate(Xs), np.concatenate(tps), np.concatenate(locs))
    data.obs["domain"] = np.concatenate(doms)
    return data
