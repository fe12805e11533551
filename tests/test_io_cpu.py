"""CPU: the I/O leg of SURVEY 8 f4.  The MI355X image has no anndata / h5py, so tools/npz_to_h5ad.py (which turns this
package's .npz outputs into the reference's .h5ad files, train.py:18,43-44 / _analyze_utils.py:137) is executed here against
a stand-in `anndata` module that records what the script hands to AnnData(...) and write_h5ad(...): the field mapping of the
converter has then run at least once, on files written by the product's own writers."""
import os
import runpy
import sys
import types

import numpy as np
import pytest

from conftest import ROOT

SCRIPT = os.path.join(ROOT, "tools", "npz_to_h5ad.py")


class _FakeAnnData:
    written = []

    def __init__(self, X, obs=None, var=None):
        self.X, self.obs, self.var, self.obsm = np.asarray(X), obs, var, {}
        self.shape = self.X.shape

    def write_h5ad(self, path):
        _FakeAnnData.written.append((path, self))


@pytest.fixture()
def fake_anndata(monkeypatch):
    mod = types.ModuleType("anndata")
    mod.AnnData = _FakeAnnData
    _FakeAnnData.written = []
    monkeypatch.setitem(sys.modules, "anndata", mod)
    return mod


def _run(argv, monkeypatch):
    monkeypatch.setattr(sys, "argv", [SCRIPT] + argv)
    runpy.run_path(SCRIPT, run_name="__main__")


def test_latent_npz_to_h5ad_field_mapping(tmp_path, fake_anndata, monkeypatch, capsys):
    import pandas as pd
    rng = np.random.default_rng(0)
    n = 12
    rows = rng.permutation(n)                                   # the order train() writes: owned time points, then rows
    X = rng.normal(size=(n, 20)).astype(np.float32)
    tp = np.repeat([0, 1, 2], 4)[rows]
    sp = rng.uniform(size=(n, 2))
    # exactly what spadot_amd/train.py writes next to latent.h5ad
    np.savez_compressed(tmp_path / "p_latent.npz", X=X, rows=rows, timepoint=tp, spatial=sp)
    obs = pd.DataFrame({"timepoint": np.repeat([0, 1, 2], 4), "annotation": [f"a{i}" for i in range(n)]},
                       index=[f"spot{i}" for i in range(n)])
    obs.to_csv(tmp_path / "obs.csv")
    _run(["latent", str(tmp_path / "p_latent.npz"), "--obs", str(tmp_path / "obs.csv")], monkeypatch)
    (path, ad), = _FakeAnnData.written
    assert path == str(tmp_path / "p_latent.h5ad")
    np.testing.assert_array_equal(ad.X, X)                       # X = N x 20 latent (train.py:43)
    np.testing.assert_array_equal(ad.obsm["spatial"], sp)        # obsm['spatial'] (train.py:44)
    assert list(ad.obs.index) == [str(i) for i in rows.tolist()]
    np.testing.assert_array_equal(ad.obs["timepoint"].to_numpy(), tp)
    assert list(ad.obs["annotation"]) == [f"a{i}" for i in rows.tolist()]    # extra obs columns follow the rows
    assert list(ad.obs.columns).count("timepoint") == 1
    assert "wrote" in capsys.readouterr().out
    # without --obs: time point only, default output name, --out honoured
    _run(["latent", str(tmp_path / "p_latent.npz"), "--out", str(tmp_path / "x.h5ad")], monkeypatch)
    path2, ad2 = _FakeAnnData.written[-1]
    assert path2 == str(tmp_path / "x.h5ad") and list(ad2.obs.columns) == ["timepoint"]


def test_transition_table_npz_to_h5ad_field_mapping(tmp_path, fake_anndata, monkeypatch):
    tab = np.arange(12, dtype=np.float64).reshape(3, 4)
    # the layout spadot_amd.analyze_ot.write_transition_tables leaves (tests/test_ot_gpu.py checks the writer itself)
    np.savez(tmp_path / "transition_table_0_1.npz", X=tab, obs_names=np.array(["E10_0", "E10_1", "E10_2"]),
             var_names=np.array(["E11_0", "E11_1", "E11_2", "E11_3"]))
    _run(["table", str(tmp_path / "transition_table_0_1.npz")], monkeypatch)
    (path, ad), = _FakeAnnData.written
    assert path.endswith("transition_table_0_1.h5ad")
    np.testing.assert_array_equal(ad.X, tab)
    assert list(ad.obs.index) == ["E10_0", "E10_1", "E10_2"] and list(ad.var.index) == ["E11_0", "E11_1", "E11_2", "E11_3"]


def test_converter_needs_anndata_and_says_so(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "anndata", None)            # import anndata -> ImportError, as in this image
    np.savez(tmp_path / "t.npz", X=np.zeros((1, 1)), obs_names=np.array(["a"]), var_names=np.array(["b"]))
    with pytest.raises(ImportError):
        _run(["table", str(tmp_path / "t.npz")], monkeypatch)
