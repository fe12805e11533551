"""`train(args)`: mirror of /root/reference/SpaDOT/train.py:9-44 -- same argument object (data,
output_dir, prefix, config, save_model, device) and the same output files ({prefix}inducing_points.csv,
loss.csv, SpaDOT_model.pth, {prefix}latent.h5ad).

Input: `args.data` is a path to an .h5ad file (needs the `anndata` package, as in the reference), a
path to an .npz with arrays X, timepoint, spatial (this package's dependency-free container), or an
in-memory object exposing .X, .obs['timepoint'], .obsm['spatial'] (AnnData or
spadot_amd.synthetic.SpatialData).  The latent embedding is written as .h5ad when anndata is
importable and always as {prefix}latent.npz (X, rows, timepoint, spatial).
"""
import os

import numpy as np
import torch

from .synthetic import SpatialData
from .utils import _train_utils, _utils


def _load(data):
    if not isinstance(data, (str, os.PathLike)):
        return data, None
    path = os.path.abspath(data)
    if path.endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        return SpatialData(z["X"], z["timepoint"], z["spatial"]), path
    try:
        import anndata
    except ImportError as e:
        raise ImportError("reading .h5ad needs the `anndata` package (as the reference does); "
                          "alternatively pass an .npz with X/timepoint/spatial or an in-memory object") from e
    return anndata.read_h5ad(path), path


def train(args):
    print("Loading data...")
    adata, path = _load(args.data)
    if not getattr(args, "output_dir", None):
        args.output_dir = os.path.dirname(path) if path else os.getcwd()
    os.makedirs(args.output_dir, exist_ok=True)
    if not hasattr(args, "prefix") or args.prefix is None:
        args.prefix = ""
    model_config = _utils.load_model_config(args)
    model_config["input_dim"] = adata.n_vars
    tps = sorted(set(np.asarray(adata.obs["timepoint"]).tolist()))
    model_config["timepoints"] = tps
    model_config["device"] = torch.device(args.device)
    if model_config["device"].type != "cuda":
        raise RuntimeError("spadot_amd trains on the MI355X only (device 'cuda:N'); there is no CPU path")
    model_config["dtype"] = torch.float32          # reference: float64 (train.py:27); see model/SpaDOT.py
    # compute dtype of the GAT branch and the G-sized dense maps: config.yaml's `compute_dtype` ('float32' |
    # 'bfloat16'); an `args.compute_dtype` (string or torch dtype) overrides it
    if getattr(args, "compute_dtype", None) is not None:
        model_config["compute_dtype"] = args.compute_dtype
    model_config["compute_dtype"] = _utils.resolve_compute_dtype(model_config.get("compute_dtype"))

    _utils.set_seed(model_config["seed"])
    print("Preparing data...")
    dataloader_dict = _train_utils.prepare_dataloader(adata, model_config)
    _utils._save_inducing_points(args, dataloader_dict["inducing_points"])

    print("Training model...")
    # (args.epoch_seconds, optional list: wall time per epoch -- tools/e2e_chickenheart.py)
    model, loss_df = _train_utils.train_SpaDOT(dataloader_dict, model_config, epoch_seconds=getattr(args, "epoch_seconds", None))
    loss_df.T.to_csv(args.output_dir + os.sep + "loss.csv")
    if getattr(args, "save_model", False):
        torch.save(model.state_dict(), args.output_dir + os.sep + "SpaDOT_model.pth")
        print("Model saved to %s" % (args.output_dir))
    latent, rows = _train_utils.get_latent(model, model_config, adata, dataloader_dict)
    out = args.output_dir + os.sep + args.prefix + "latent"
    np.savez_compressed(out + ".npz", X=latent, rows=rows, timepoint=np.asarray(adata.obs["timepoint"])[rows],
                        spatial=np.asarray(adata.obsm["spatial"])[rows])
    try:
        import anndata
        lat = anndata.AnnData(latent, obs=adata.obs.iloc[rows] if hasattr(adata.obs, "iloc") else None)
        lat.obsm["spatial"] = np.asarray(adata.obsm["spatial"])[rows]
        lat.write_h5ad(out + ".h5ad")
    except ImportError:
        print("anndata not installed: latent written as %s.npz only" % out)
    return model, loss_df
