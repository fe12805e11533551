#!/usr/bin/env python3
"""Turns this package's dependency-free .npz outputs into the .h5ad files the reference's stages read and write.
Meant to run in the REFERENCE's environment (needs `anndata`, which the MI355X image does not ship):

    python npz_to_h5ad.py latent  OUT/latent.npz  [--obs obs.csv]      ->  OUT/latent.h5ad
        X = the [N, 20] latent, obsm['spatial'], obs['timepoint'] (+ the columns of obs.csv, row-aligned with the
        training data) -- what SpaDOT/train.py:40-44 writes and `SpaDOT analyze` (_analyze_utils.py) reads
    python npz_to_h5ad.py table   OUT/transition_table_0_1.npz         ->  OUT/transition_table_0_1.h5ad
        the aggregated OT matrix of _analyze_utils.py:137 (obs = domains of the earlier time point, var = the later one)

Only numpy, pandas and anndata are imported; nothing of this repository is needed."""
import argparse
import os

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=["latent", "table"])
    ap.add_argument("npz")
    ap.add_argument("--obs", default=None, help="csv with one row per spot of the training data (optional)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import anndata
    import pandas as pd
    z = np.load(a.npz, allow_pickle=False)
    out = a.out or os.path.splitext(a.npz)[0] + ".h5ad"
    if a.kind == "latent":
        rows = z["rows"]
        obs = pd.DataFrame({"timepoint": z["timepoint"]}, index=[str(i) for i in rows.tolist()])
        if a.obs:
            extra = pd.read_csv(a.obs, index_col=0).iloc[rows]
            extra.index = obs.index
            obs = pd.concat([extra.drop(columns=[c for c in ("timepoint",) if c in extra.columns]), obs], axis=1)
        ad = anndata.AnnData(np.asarray(z["X"]), obs=obs)
        ad.obsm["spatial"] = np.asarray(z["spatial"])
    else:
        ad = anndata.AnnData(np.asarray(z["X"]), obs=pd.DataFrame(index=z["obs_names"].astype(str)),
                             var=pd.DataFrame(index=z["var_names"].astype(str)))
    ad.write_h5ad(out)
    print("wrote", out, ad.shape)


if __name__ == "__main__":
    main()
