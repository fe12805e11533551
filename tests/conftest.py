import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def solve_cfg(g):
    """cfg dict stored in an ot_solve_*.npz fixture."""
    cfg = {k: float(v) for k, v in zip(g["cfg_keys"].tolist(), g["cfg_vals"].tolist())}
    for k in ("batch_size", "max_iter", "growth_iters"):
        cfg[k] = int(cfg[k])
    return cfg


SOLVE_CASES = ["train10x10", "ragged7x13", "edge1x5", "growth64x48", "spots300x400",
               "absorb120x150", "outlier40x56", "longbatch90x110"]


@pytest.fixture(scope="session")
def oracle_ot():
    from oracle import ot_oracle
    ot_oracle.lib()
    return ot_oracle
