"""CPU: oracle/model_oracle.py against the reference's own outputs (tests/golden/model_*.npz, made by
oracle/gen_golden_model.py).  GATConv / NeighborLoader pieces are parity-unpinned (see the oracle's
header); everything around them is pinned here."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import model_oracle as mo

F64 = torch.float64


def T(x):
    return torch.as_tensor(np.asarray(x), dtype=F64)


@pytest.mark.parametrize("tag", ["s", "l"])
def test_kernels_and_svgp_match_reference(tag):
    g = load_golden("model_svgp.npz")
    x, z, y, noise = T(g[f"{tag}_x"]), T(g[f"{tag}_z"]), T(g[f"{tag}_y"]), T(g[f"{tag}_noise"])
    for kt in ("Gaussian", "Cauchy", "Quadratic"):
        np.testing.assert_allclose(mo.rbf_kernel(x, z, kt, 0.1).numpy(), g[f"{tag}_K_{kt}"], rtol=1e-13, atol=1e-300)
    sv = mo.SVGPOracle(g[f"{tag}_z"], float(g[f"{tag}_N_train"]))
    # two jittered m x m inverses (cond ~1e5-1e6) amplify BLAS-order differences to ~1e-8 relative
    mean, B, mu_hat, A_hat = sv.approximate_posterior_params(x, x, y, noise)
    np.testing.assert_allclose(mean.numpy(), g[f"{tag}_mean"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(B.numpy(), g[f"{tag}_B"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mu_hat.numpy(), g[f"{tag}_mu_hat"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(A_hat.numpy(), g[f"{tag}_A_hat"], rtol=1e-6, atol=1e-9)
    l3, kl = sv.variational_loss(x, y, noise, mu_hat, A_hat)
    assert float(l3) == pytest.approx(float(g[f"{tag}_l3"]), rel=1e-7)
    assert float(kl) == pytest.approx(float(g[f"{tag}_kl"]), rel=1e-7)
    xt = T(g[f"{tag}_xt"])
    mean_t, B_t, _, _ = sv.approximate_posterior_params(xt, x, y, noise)
    np.testing.assert_allclose(mean_t.numpy(), g[f"{tag}_mean_t"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(B_t.numpy(), g[f"{tag}_B_t"], rtol=1e-6, atol=1e-9)


def _params(g):
    return {k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("sd/")}


def test_encoder_decoder_match_reference():
    g = load_golden("model_composite.npz")
    P = _params(g)
    b = int(g["batch_size"])
    yb = T(g["Y"])[torch.as_tensor(g["n_id"])]
    mu, var = mo.svgp_encoder(P, yb[:b], train=True)
    np.testing.assert_allclose(mu.numpy(), g["enc_mu_train"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(var.numpy(), g["enc_var_train"], rtol=1e-11)
    mu, var = mo.svgp_encoder(P, yb[:b], train=False)
    np.testing.assert_allclose(mu.numpy(), g["enc_mu_eval"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(var.numpy(), g["enc_var_eval"], rtol=1e-11)
    np.testing.assert_allclose(mo.decoder(P, T(g["final_latent"])).numpy(), g["dec_out"], rtol=1e-11, atol=1e-13)


def test_composite_forward_matches_reference():
    g = load_golden("model_composite.npz")
    P = _params(g)
    b, heads = int(g["batch_size"]), int(g["heads"])
    n_id = torch.as_tensor(g["n_id"])
    xb, yb = T(g["X"])[n_id], T(g["Y"])[n_id]
    sv = mo.SVGPOracle(g["ind0"], float(g["N_train0"]))
    (recon, skl, gkl, align, z), _ = mo.spadot_forward(P, sv, xb, yb, torch.as_tensor(g["sub_edge_index"]), b, heads,
                                                       T(g["noise_svgp"]), T(g["noise_gat"]), train=True)
    assert float(recon) == pytest.approx(float(g["recon"]), rel=1e-7)
    assert float(skl) == pytest.approx(float(g["SVGP_KL"]), rel=1e-6)
    assert float(gkl) == pytest.approx(float(g["GAT_KL"]), rel=1e-10)
    assert float(align) == pytest.approx(float(g["alignment"]), rel=1e-6)
    np.testing.assert_allclose(z.numpy(), g["final_latent"], rtol=1e-6, atol=1e-8)


def test_svgp_branch_without_a_tape_leaves_the_other_gradients_exact():
    """spadot_forward(svgp_no_grad=True) (the cfg2-size gradient check of tests/test_step_parity_gpu.py): same values, and the
    gradients of every GAT-encoder / decoder parameter equal the fully taped ones -- no path from them runs through the SVGP
    branch; the SVGP encoder's parameters get none."""
    g = load_golden("model_composite.npz")
    b, heads = int(g["batch_size"]), int(g["heads"])
    n_id = torch.as_tensor(g["n_id"])
    xb, yb = T(g["X"])[n_id], T(g["Y"])[n_id]
    sv = mo.SVGPOracle(g["ind0"], float(g["N_train0"]))
    w = (0.1, 0.5, 1e-4, 0.1, 0.1, 1.0)
    out = {}
    for flag in (False, True):
        P = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
             for k, v in _params(g).items()}
        loss, terms, z = mo.step_loss(P, sv, xb, yb, torch.as_tensor(g["sub_edge_index"]), b, heads, T(g["noise_svgp"]),
                                      T(g["noise_gat"]), w, svgp_no_grad=flag)
        loss.backward()
        out[flag] = (float(loss.detach()), z.detach().numpy(), {k: v.grad for k, v in P.items() if torch.is_tensor(v) and v.requires_grad})
    assert out[True][0] == pytest.approx(out[False][0], rel=1e-13)
    np.testing.assert_allclose(out[True][1], out[False][1], rtol=1e-13)
    n_checked = 0
    for k, gr in out[False][2].items():
        if k.startswith("SVGPEncoder."):
            assert out[True][2][k] is None and gr is not None
        else:
            np.testing.assert_allclose(out[True][2][k].numpy(), gr.numpy(), rtol=1e-12, atol=1e-300)
            n_checked += 1
    assert n_checked >= 20


def test_all_latent_samples_matches_reference():
    g = load_golden("model_composite.npz")
    P = _params(g)
    sv = mo.SVGPOracle(g["ind1"], float(g["N_train1"]))
    lat = mo.all_latent_samples(P, sv, T(g["X"]), T(g["Y"]), torch.as_tensor(g["edge_index"]), int(g["heads"]), 10)
    np.testing.assert_allclose(lat.numpy(), g["all_latent_tp1"], rtol=1e-6, atol=1e-8)


def test_induced_batch_is_the_two_hop_in_neighbourhood():
    g = load_golden("model_composite.npz")
    ei = torch.as_tensor(g["edge_index"])
    n_id, sub = mo.induced_batch(ei, g["X"].shape[0], np.arange(int(g["batch_size"])))
    np.testing.assert_array_equal(n_id.numpy(), g["n_id"])
    np.testing.assert_array_equal(sub.numpy(), g["sub_edge_index"])
    assert n_id[: int(g["batch_size"])].tolist() == list(range(int(g["batch_size"])))
    # every edge of the subgraph is an original edge, relabelled
    orig = set(map(tuple, ei.numpy().T.tolist()))
    back = n_id.numpy()[sub.numpy()]
    assert all((int(s), int(d)) in orig for s, d in back.T)


def test_glue_functions_match_reference():
    g = load_golden("model_glue.npz")
    np.testing.assert_array_equal(mo.beta_cycle_linear(100, stop=1.0), g["beta_100_stop1"])
    np.testing.assert_array_equal(mo.beta_cycle_linear(100, stop=0.5), g["beta_100_stop05"])
    np.testing.assert_array_equal(mo.beta_cycle_linear(37, stop=1.0), g["beta_37"])
    lat = T(g["latent"])
    kl = mo.kmeans_loss(lat, g["centers1"], g["batch_labels"])
    assert float(kl) == pytest.approx(float(g["kmeans_loss"]), rel=1e-12)
    ot = mo.ot_loss(lat, g["batch_labels"], g["all_labels"], g["centers1"], g["centers0"], g["gamma"])
    assert float(ot) == pytest.approx(float(g["ot_loss"]), rel=1e-12)


def test_kmeans_assignment_fixture_is_nearest_centre():
    g = load_golden("model_glue.npz")
    d = ((g["km_points"][:, None, :] - g["km_centers"][None]) ** 2).sum(-1)
    np.testing.assert_array_equal(d.argmin(1).astype(np.int32), g["km_predict"])
    np.testing.assert_array_equal(g["km_labels"], g["km_predict"])
