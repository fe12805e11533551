"""Small utilities of the training stage: mirror of the hot-path part of
/root/reference/SpaDOT/utils/_utils.py (set_seed :22-32, load_model_config :38-50,
_save_inducing_points :102-118).  The spatial graph (_Cal_Spatial_Net :52-100) lives in
spadot_amd.graph without the dense adjacency."""
import os
import random

import numpy as np
import torch
import yaml


def set_seed(seed=1993):
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def seed_worker(worker_id=1993):
    np.random.seed(worker_id)
    random.seed(worker_id)


_COMPUTE_DTYPES = {"float32": torch.float32, "fp32": torch.float32, "f32": torch.float32,
                   "bfloat16": torch.bfloat16, "bf16": torch.bfloat16}


def resolve_compute_dtype(value):
    """config.yaml's `compute_dtype` ('float32' | 'bfloat16', also 'fp32' / 'bf16') or a torch dtype -> torch dtype.
    None means float32.  The compute dtype covers the GAT branch and the G-sized dense maps; parameters and the
    optimizer stay fp32, the SVGP m x m algebra fp64 (model/SpaDOT.py)."""
    if value is None:
        return torch.float32
    if isinstance(value, torch.dtype):
        if value not in (torch.float32, torch.bfloat16):
            raise ValueError(f"compute_dtype {value} is not supported (float32 or bfloat16)")
        return value
    try:
        return _COMPUTE_DTYPES[str(value).lower().replace("torch.", "")]
    except KeyError:
        raise ValueError(f"compute_dtype {value!r} is not supported (float32 or bfloat16)") from None


def load_model_config(args):
    """yaml.safe_load of args.config, or of the packaged default (no merging, like the reference)."""
    path = args.config if getattr(args, "config", None) else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config.yaml")
    with open(path, "r") as f:
        return yaml.safe_load(f)


def _save_inducing_points(args, inducing_points_dict):
    """{prefix}inducing_points.csv with columns norm-pixel_x, norm-pixel_y, timepoint."""
    import pandas as pd
    frames = []
    for key, value in inducing_points_dict.items():
        df = pd.DataFrame(np.asarray(value))
        df.columns = ["norm-pixel_x", "norm-pixel_y"]
        df["timepoint"] = key
        frames.append(df)
    pd.concat(frames, ignore_index=True).to_csv(args.output_dir + os.sep + args.prefix + "inducing_points.csv", index=False)
