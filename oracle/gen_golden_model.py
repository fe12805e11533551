"""oracle/gen_golden_model.py -- generates tests/golden/model_*.npz FROM THE REFERENCE ITSELF.

Build-container only (needs /root/reference).  Imports the reference's model/svgp.py, decoder.py,
encoder.py and SpaDOT.py by path.  torch_geometric is absent from this image, so the one name the
reference imports from it (GATConv, encoder.py:4) is provided by THIS repo's restatement
(oracle/model_oracle.gat_conv wrapped in an nn.Module with PyG's parameter names): the composite
fixtures therefore pin everything AROUND GATConv (SVGP loop, KL sign trick, BatchNorm/LayerNorm,
alignment, reconstruction) and are conditional on GATConv itself, whose parity stays unpinned.
Three pure functions of utils/_train_utils.py (which cannot be imported: scanpy/anndata/PyG at
module top) are executed from their source lines at generation time only.

    python oracle/gen_golden_model.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import model_oracle as mo  # noqa: E402

REF = "/root/reference/SpaDOT"
OUT = os.path.join(HERE, "..", "tests", "golden")


class GATConvStandIn(nn.Module):
    """Parameter layout of torch_geometric.nn.GATConv (lin.weight, att_src, att_dst, bias);
    forward = oracle/model_oracle.gat_conv."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True):
        super().__init__()
        self.heads, self.concat, self.out_channels = heads, concat, out_channels
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)

    def forward(self, x, edge_index):
        return mo.gat_conv(x, edge_index, self.lin.weight, self.att_src, self.att_dst, self.bias,
                           self.heads, self.concat)


def load_reference_model():
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    ns("SpaDOT", REF)
    ns("SpaDOT.model", REF + "/model")
    tg = types.ModuleType("torch_geometric")
    tgn = types.ModuleType("torch_geometric.nn")
    tgn.GATConv = GATConvStandIn
    tg.nn = tgn
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tgn

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    svgp = load("SpaDOT.model.svgp", REF + "/model/svgp.py")
    load("SpaDOT.model.decoder", REF + "/model/decoder.py")
    load("SpaDOT.model.encoder", REF + "/model/encoder.py")
    spadot = load("SpaDOT.model.SpaDOT", REF + "/model/SpaDOT.py")
    return svgp, spadot


def lift(path, first, last, env):
    """exec lines [first, last] (1-based, inclusive) of a reference file in `env`."""
    lines = open(path).read().split("\n")[first - 1:last]
    exec("\n".join(lines), env)


def main():
    import warnings
    warnings.filterwarnings("ignore")
    os.makedirs(OUT, exist_ok=True)
    svgp_mod, spadot_mod = load_reference_model()
    rng = np.random.default_rng(1993)
    torch.manual_seed(1993)
    f64 = torch.float64

    # ------------------------------------------------------------------ (3) kernels + SVGP
    out = {}
    for tag, (b, m) in {"s": (64, 17), "l": (96, 35)}.items():
        x = torch.tensor(rng.normal(size=(b, 2)), dtype=f64)
        z = torch.tensor(rng.normal(size=(m, 2)), dtype=f64)
        y = torch.tensor(rng.normal(size=b), dtype=f64)
        noise = torch.tensor(rng.uniform(0.3, 2.0, size=b), dtype=f64)
        out.update({f"{tag}_x": x.numpy(), f"{tag}_z": z.numpy(), f"{tag}_y": y.numpy(),
                    f"{tag}_noise": noise.numpy(), f"{tag}_N_train": np.array(700.0)})
        for kt in ("Gaussian", "Cauchy", "Quadratic"):
            cfg = dict(dtype=f64, device="cpu", kernel_type=kt, kernel_scale=0.1)
            sv = svgp_mod.SVGP(cfg, z.numpy(), N_train=700)
            out[f"{tag}_K_{kt}"] = sv.kernel_matrix(x, sv.inducing_index_points).numpy()
            if kt == "Gaussian":
                mean, B, mu_hat, A_hat = sv.approximate_posterior_params(x, x, y, noise)
                l3, kl = sv.variational_loss(x, y, noise, mu_hat, A_hat)
                out.update({f"{tag}_mean": mean.numpy(), f"{tag}_B": B.numpy(), f"{tag}_mu_hat": mu_hat.numpy(),
                            f"{tag}_A_hat": A_hat.numpy(), f"{tag}_l3": l3.numpy(), f"{tag}_kl": kl.numpy()})
                # test points different from training points (the inference call shape)
                xt = torch.tensor(rng.normal(size=(11, 2)), dtype=f64)
                mean_t, B_t, _, _ = sv.approximate_posterior_params(xt, x, y, noise)
                out.update({f"{tag}_xt": xt.numpy(), f"{tag}_mean_t": mean_t.numpy(), f"{tag}_B_t": B_t.numpy()})
    np.savez_compressed(os.path.join(OUT, "model_svgp.npz"), **out)
    print("model_svgp:", {k: v.shape for k, v in out.items() if k.startswith("s_") and v.ndim}, flush=True)

    # ------------------------------------------------------------------ (4)+(5) composite model
    G, N, k, b, heads, hid = 40, 150, 6, 32, 4, 8
    tps = [0, 1]
    coords = rng.uniform(0, 10, size=(N, 2))
    coords = (coords - coords.mean(0)) / coords.std(0)
    edge_index = mo.knn_graph(coords, k)
    Y = torch.tensor(rng.normal(size=(N, G)), dtype=f64)
    X = torch.tensor(coords, dtype=f64)
    m = 13
    ind = {0: coords[rng.choice(N, m, replace=False)], 1: coords[rng.choice(N, m + 4, replace=False)]}
    cfg = dict(input_dim=G, z_dim=20, dtype=f64, device="cpu", svgp_encoder_layers=[24, 12],
               gat_encoder_hidden=hid, gat_attention_heads=heads, decoder_layers=[12, 24],
               kernel_type="Gaussian", kernel_scale=0.1, timepoints=tps)
    ref = spadot_mod.SpaDOT(cfg, {"inducing_points": ind, "N_train": {0: N, 1: N + 30}})
    # non-trivial BatchNorm running statistics / affine parameters / GAT biases
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("bias") or ".1.weight" in name or ".4.weight" in name:
                p.add_(0.1 * torch.randn_like(p))
        for name, buf in ref.named_buffers():
            if name.endswith("running_mean"):
                buf.add_(0.2 * torch.randn_like(buf))
            if name.endswith("running_var"):
                buf.mul_(torch.empty_like(buf).uniform_(0.5, 1.5))
    seeds = np.arange(b)
    n_id, sub_ei = mo.induced_batch(edge_index, N, seeds)
    xb, yb = X[n_id], Y[n_id]
    comp = {}
    sd0 = {kk: v.detach().clone() for kk, v in ref.state_dict().items()}
    for kk, v in sd0.items():
        comp["sd/" + kk] = v.numpy()
    ref.train()
    torch.manual_seed(7)
    recon, skl, gkl, align, zlat = ref.forward(x=xb, y=yb, edge_index=sub_ei, tp=0, batch_size=b)
    torch.manual_seed(7)
    n1 = torch.randn((b, 10), dtype=f64)
    n2 = torch.randn((b, 10), dtype=f64)
    comp.update(dict(X=X.numpy(), Y=Y.numpy(), edge_index=edge_index.numpy(), n_id=n_id.numpy(),
                     sub_edge_index=sub_ei.numpy(), batch_size=np.array(b), heads=np.array(heads),
                     noise_svgp=n1.numpy(), noise_gat=n2.numpy(), ind0=ind[0], ind1=ind[1],
                     N_train0=np.array(float(N)), N_train1=np.array(float(N + 30)),
                     recon=recon.detach().numpy(), SVGP_KL=skl.detach().numpy(), GAT_KL=gkl.detach().numpy(),
                     alignment=align.detach().numpy(), final_latent=zlat.detach().numpy()))
    # gradients of the reference's step loss w.r.t. every parameter (lambda1*recon - beta1*SVGP_KL + ...)
    loss = 0.1 * recon - 0.5 * skl + 1e-4 * gkl + 0.1 * align
    ref.zero_grad()
    loss.backward()
    for name, p in ref.named_parameters():
        comp["grad/" + name] = p.grad.detach().numpy().copy() if p.grad is not None else np.zeros(0)
    # running statistics after that ONE train-mode forward (momentum 0.1, unbiased batch variance)
    for kk, v in ref.state_dict().items():
        if "running" in kk or "num_batches" in kk:
            comp["sd_after/" + kk] = v.detach().numpy().copy()   # state_dict() aliases the live buffers
    # encoder / decoder alone, train + eval
    ref.load_state_dict(sd0)
    ref.train()
    mu_t, var_t = ref.SVGPEncoder(yb[:b])
    ref.load_state_dict(sd0)
    ref.eval()
    mu_e, var_e = ref.SVGPEncoder(yb[:b])
    dec = ref.decoder(zlat.detach())
    g_mu, g_var = ref.GATEncoder(yb, sub_ei)
    lat_all = ref.all_latent_samples(X.numpy(), Y.numpy(), edge_index.numpy(), 1)
    comp.update(dict(enc_mu_train=mu_t.detach().numpy(), enc_var_train=var_t.detach().numpy(),
                     enc_mu_eval=mu_e.detach().numpy(), enc_var_eval=var_e.detach().numpy(),
                     dec_out=dec.detach().numpy(), gat_mu=g_mu.detach().numpy(), gat_var=g_var.detach().numpy(),
                     all_latent_tp1=lat_all))
    np.savez_compressed(os.path.join(OUT, "model_composite.npz"), **comp)
    print("model_composite: recon", float(recon), "SVGP_KL", float(skl), "GAT_KL", float(gkl), "align", float(align),
          "n_sub", int(n_id.numel()), "E_sub", int(sub_ei.shape[1]), flush=True)

    # ------------------------------------------------------------------ (6) glue functions
    env = {"np": np, "torch": torch}
    tu = REF + "/utils/_train_utils.py"
    lift(tu, 143, 153, env)
    lift(tu, 240, 253, env)
    lift(tu, 272, 307, env)
    glue = {"beta_100_stop1": env["_beta_cycle_linear"](100, stop=1.0),
            "beta_100_stop05": env["_beta_cycle_linear"](100, stop=0.5),
            "beta_37": env["_beta_cycle_linear"](37, stop=1.0)}

    class M:
        pass

    mdl = M()
    nb, nc = 48, 10
    lat = torch.tensor(rng.normal(size=(nb, 20)), dtype=f64)
    centers = {0: rng.normal(size=(nc, 20)), 1: rng.normal(size=(nc, 20))}
    gidx = rng.choice(5000, size=300, replace=False)
    all_labels = rng.integers(0, nc, size=300)
    all_labels[all_labels == 7] = 3            # cluster 7 never occurs
    batch_pos = rng.choice(300, size=nb, replace=False)
    batch_labels = all_labels[batch_pos].copy()
    mdl.kmeans_index_dict = {1: dict(zip(gidx.tolist(), all_labels.tolist()))}
    mdl.kmeans_center_dict = centers
    mdl.kmeans_cluster_dict = {1: all_labels.tolist()}
    gamma = rng.uniform(0, 1, size=(nc, len(set(all_labels.tolist()))))
    gamma[4, :] = 0.0                          # zero row: 0/0 -> NaN -> 0 (_train_utils.py:299-300)
    mdl.gammas = {"0_1": gamma}
    mcfg = {"dtype": f64, "device": "cpu"}
    tp_ix = torch.tensor(gidx[batch_pos], dtype=torch.int)
    kl_val = env["_compute_kmeans_loss"](mdl, mcfg, 1, tp_ix, lat)
    ot_val = env["_compute_OT_loss"](mdl, mcfg, 1, tp_ix, lat, 0)
    glue.update(dict(latent=lat.numpy(), centers0=centers[0], centers1=centers[1], batch_labels=batch_labels,
                     all_labels=all_labels, gamma=gamma, kmeans_loss=kl_val.numpy(), ot_loss=ot_val.numpy()))
    # (7) k-means assignment: latent + centres -> labels (sklearn KMeans.predict)
    from sklearn.cluster import KMeans
    pts = rng.normal(size=(500, 20)) + 3.0 * rng.normal(size=(10, 20))[rng.integers(0, 10, 500)]
    km = KMeans(n_clusters=10, random_state=1993, n_init=10).fit(pts)
    glue.update(dict(km_points=pts, km_centers=km.cluster_centers_, km_labels=km.labels_.astype(np.int32),
                     km_predict=km.predict(pts).astype(np.int32)))
    np.savez_compressed(os.path.join(OUT, "model_glue.npz"), **glue)
    print("model_glue: kmeans_loss", float(kl_val), "ot_loss", float(ot_val), flush=True)


if __name__ == "__main__":
    main()
