"""oracle/model_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU fp64 restatement (torch tensors as the array library, no autograd tricks, no custom kernels)
of the model side of the hot path, written from the reference's formulas:

  rbf_kernel / SVGPOracle      /root/reference/SpaDOT/model/svgp.py
  svgp_encoder / decoder       /root/reference/SpaDOT/model/encoder.py:7-34, decoder.py:3-20
  gat_conv / gat_encoder       encoder.py:37-61 + torch_geometric GATConv semantics
  spadot_forward / all_latent  /root/reference/SpaDOT/model/SpaDOT.py:52-142
  kmeans_loss / ot_loss        /root/reference/SpaDOT/utils/_train_utils.py:240-253, :272-307
  beta_cycle_linear            _train_utils.py:143-153
  knn_graph / induced_batch    _utils.py:52-100, _train_utils.py:69-85 (+ SURVEY App. B)

Parity status:
  * SVGP, kernels, encoders' MLPs, decoder, composite forward, k-means/OT losses, beta schedule:
    PINNED by tests/golden/model_*.npz, generated from the reference itself
    (oracle/gen_golden_model.py).
  * GATConv arithmetic and the NeighborLoader batch construction live in torch_geometric, which
    is un-vendored, unpinned (not even listed in the reference's pyproject.toml) and absent from
    this image: those two pieces are "parity unpinned" -- restated from the library's documented
    semantics.  The composite-forward fixtures are valid conditional on this GATConv (it is the
    one injected into the reference when they are generated).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parameters are passed as a flat dict with the reference's state_dict key names.
"""
import math

import numpy as np
import torch

F64 = torch.float64
JITTER = 1e-2   # svgp.py:6


# ----------------------------------------------------------------------------- SVGP (svgp.py)

def rbf_kernel(x, y, kernel_type="Gaussian", scale=0.1):
    """svgp.py:110-125: Euclidean cdist, squared, then the kernel profile."""
    d2 = torch.square(torch.cdist(x, y, p=2))
    if kernel_type == "Gaussian":
        return torch.exp(-d2 / scale)
    if kernel_type == "Cauchy":
        return 1.0 / (1.0 + d2 / scale)
    if kernel_type == "Quadratic":
        return 1.0 - d2 / (d2 + scale)
    raise ValueError(kernel_type)


def _jit(M, jitter=JITTER):
    return M + jitter * torch.eye(M.shape[-1], dtype=M.dtype)


class SVGPOracle:
    """One time point's SVGP (no trainable tensors: svgp.py:24-30)."""

    def __init__(self, inducing_points, N_train, kernel_type="Gaussian", scale=0.1, jitter=JITTER):
        self.z = torch.as_tensor(inducing_points, dtype=F64)
        self.N_train = float(N_train)
        self.kernel_type, self.scale, self.jitter = kernel_type, scale, jitter

    def k(self, x, y):
        return rbf_kernel(x, y, self.kernel_type, self.scale)

    def approximate_posterior_params(self, x_test, x_train, y, noise, mean_only=False):
        """svgp.py:62-84.  Returns (mean[b_test], B[b_test], mu_hat[m], A_hat[m,m]).  mean_only (checker convenience for
        whole-time-point inference, SpaDOT.py:96-123 reads only the mean): the same formula for the mean, without the
        N_t x N_t products behind B that nobody reads there -- returns (mean, None, None, None)."""
        b = x_train.shape[0]
        K_mm = self.k(self.z, self.z)
        K_xm = self.k(x_test, self.z)
        K_nm = self.k(x_train, self.z)
        c = self.N_train / b
        sigma_l = K_mm + c * (K_nm.T @ (K_nm / noise[:, None]))
        sigma_l_inv = torch.linalg.inv(_jit(sigma_l, self.jitter))
        mean = c * (K_xm @ (sigma_l_inv @ (K_nm.T @ (y / noise))))
        if mean_only:
            return mean, None, None, None
        K_mm_inv = torch.linalg.inv(_jit(K_mm, self.jitter))
        K_xx = torch.diagonal(self.k(x_test, x_test))
        B = K_xx + torch.diagonal(-(K_xm @ (K_mm_inv @ K_xm.T)) + K_xm @ (sigma_l_inv @ K_xm.T))
        mu_hat = c * ((K_mm @ (sigma_l_inv @ K_nm.T)) @ (y / noise))
        A_hat = K_mm @ (sigma_l_inv @ K_mm)
        return mean, B, mu_hat, A_hat

    def variational_loss(self, x, y, noise, mu_hat, A_hat):
        """svgp.py:47-60 with :86-104.  Returns (L3 sum term, KL term).  The (b,m,m) tensor of
        :99-101 is formed literally here -- this is the reference's arithmetic, not a fast path."""
        b, m = x.shape[0], self.z.shape[0]
        K_mm = self.k(self.z, self.z)
        K_mm_inv = torch.linalg.inv(_jit(K_mm, self.jitter))
        K_nn = torch.diagonal(self.k(x, x))
        K_nm = self.k(x, self.z)
        mean_vector = K_nm @ (K_mm_inv @ mu_hat)
        # KL(q(u) || p(u))  (svgp.py:86-94)
        L_k = torch.linalg.cholesky(_jit(K_mm, self.jitter))
        L_s = torch.linalg.cholesky(_jit(A_hat, self.jitter))
        logdet_k = 2.0 * torch.sum(torch.log(torch.diagonal(L_k)))
        logdet_s = 2.0 * torch.sum(torch.log(torch.diagonal(L_s)))
        kl = 0.5 * (logdet_k - logdet_s - m + torch.trace(K_mm_inv @ A_hat)
                    + torch.sum(mu_hat * (K_mm_inv @ mu_hat)))
        # L3 (svgp.py:96-104)
        prec = 1.0 / noise
        k_tilde = prec * (K_nn - torch.diagonal(K_nm @ (K_mm_inv @ K_nm.T)))
        lam = K_nm.unsqueeze(2) @ K_nm.unsqueeze(1)            # (b, m, m)
        lam = K_mm_inv @ (lam @ K_mm_inv)
        tr = prec * torch.einsum("bii->b", A_hat @ lam)
        l3 = -0.5 * (k_tilde.sum() + tr.sum() + torch.log(noise).sum() + b * math.log(2.0 * math.pi)
                     + torch.sum(prec * (y - mean_vector) ** 2))
        return l3, kl


# ----------------------------------------------------------------------------- dense nets

def leaky_relu(x, slope=0.01):
    return torch.where(x >= 0, x, slope * x)


def _linear(x, P, prefix):
    return x @ P[prefix + ".weight"].T + P[prefix + ".bias"]


def _batch_norm(x, P, prefix, train, eps=1e-5):
    """nn.BatchNorm1d forward: batch statistics (biased variance) in train mode, running stats in
    eval mode.  (Running-stat updates are a side effect the oracle does not model.)"""
    if train:
        mu = x.mean(dim=0)
        var = x.var(dim=0, unbiased=False)
    else:
        mu, var = P[prefix + ".running_mean"], P[prefix + ".running_var"]
    return (x - mu) / torch.sqrt(var + eps) * P[prefix + ".weight"] + P[prefix + ".bias"]


def _layer_norm(x, P, prefix, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * P[prefix + ".weight"] + P[prefix + ".bias"]


def svgp_encoder(P, y, train, n_hidden=2):
    """encoder.py:7-34: [Linear, BatchNorm1d, LeakyReLU(0.01)] x n_hidden, Linear -> (mu, exp(logvar))."""
    h = y
    for i in range(n_hidden):
        h = _linear(h, P, f"SVGPEncoder.SVGP_encoder_net.{3 * i}")
        h = _batch_norm(h, P, f"SVGPEncoder.SVGP_encoder_net.{3 * i + 1}", train)
        h = leaky_relu(h)
    z = _linear(h, P, "SVGPEncoder.SVGP_fc")
    mu, logvar = torch.chunk(z, 2, dim=1)
    return mu, torch.exp(logvar)


def decoder(P, z, n_hidden=2):
    """decoder.py:3-20: [Linear, LayerNorm, LeakyReLU(0.01)] x n_hidden, Linear."""
    h = z
    for i in range(n_hidden):
        h = _linear(h, P, f"decoder.decoder_net.{3 * i}")
        h = _layer_norm(h, P, f"decoder.decoder_net.{3 * i + 1}")
        h = leaky_relu(h)
    return _linear(h, P, f"decoder.decoder_net.{3 * n_hidden}")


# ----------------------------------------------------------------------------- GAT (PyG semantics)

def with_self_loops(edge_index, n):
    """GATConv(add_self_loops=True): existing self loops are removed, then exactly one i->i edge per
    node is appended."""
    src, dst = edge_index[0], edge_index[1]
    keep = src != dst
    loops = torch.arange(n, dtype=edge_index.dtype)
    return torch.stack([torch.cat([src[keep], loops]), torch.cat([dst[keep], loops])])


def gat_conv(x, edge_index, W, att_src, att_dst, bias, heads, concat, negative_slope=0.2):
    """torch_geometric.nn.GATConv forward (no edge features, dropout 0).  edge_index row 0 = source j,
    row 1 = target i; node i aggregates over its incoming edges."""
    n = x.shape[0]
    C = W.shape[0] // heads
    h = (x @ W.T).view(n, heads, C)
    s_src = (h * att_src.view(1, heads, C)).sum(-1)          # (n, H)
    s_dst = (h * att_dst.view(1, heads, C)).sum(-1)
    ei = with_self_loops(edge_index, n)
    src, dst = ei[0], ei[1]
    e = s_src[src] + s_dst[dst]
    e = torch.where(e >= 0, e, negative_slope * e)            # (E, H)
    emax = torch.full((n, heads), -float("inf"), dtype=x.dtype)
    emax = emax.scatter_reduce(0, dst[:, None].expand(-1, heads), e, reduce="amax", include_self=True)
    ex = torch.exp(e - emax[dst])
    den = torch.zeros((n, heads), dtype=x.dtype).index_add_(0, dst, ex) + 1e-16
    alpha = ex / den[dst]                                     # (E, H)
    out = torch.zeros((n, heads, C), dtype=x.dtype).index_add_(0, dst, alpha[:, :, None] * h[src])
    out = out.reshape(n, heads * C) if concat else out.mean(dim=1)
    return out + bias


def gat_encoder(P, x, edge_index, heads):
    """encoder.py:37-61."""
    h = x
    for name, concat in (("gat1", True), ("gat2", True), ("gat3", False)):
        pre = f"GATEncoder.{name}"
        h = gat_conv(h, edge_index, P[pre + ".lin.weight"], P[pre + ".att_src"], P[pre + ".att_dst"],
                     P[pre + ".bias"], heads, concat)
        if name != "gat3":
            h = leaky_relu(h)
    z = _linear(h, P, "GATEncoder.GAT_fc")
    mu, logvar = torch.chunk(z, 2, dim=1)
    return mu, torch.exp(logvar)


# ----------------------------------------------------------------------------- composite (SpaDOT.py)

def gauss_cross_entropy(mu1, var1, mu2, var2):
    """SpaDOT.py:125-142."""
    return -0.5 * (1.8378770664093453 + torch.log(var2) + (var1 + mu1 ** 2 - 2 * mu1 * mu2 + mu2 ** 2) / var2)


def spadot_forward(P, svgp, x, y, edge_index, batch_size, heads, noise_svgp, noise_gat, train=True, svgp_no_grad=False):
    """SpaDOT.py:52-94 with the two reparameterisation noises supplied by the caller.
    Returns (recon, SVGP_KL, GAT_KL, alignment, final_latent) and a dict of intermediates.
    svgp_no_grad: the SVGP branch (encoder, posterior, ELBO) is evaluated without a tape -- same VALUES; a backward pass then
    yields the exact gradients of the GAT encoder's and the decoder's parameters (no path from them runs through that branch)
    and none for the SVGP encoder's.  For shapes whose taped (b, m, m) ELBO tensors do not fit the host (m ~ 600: 1.5 GB per
    latent dimension, ten of them alive until the backward pass)."""
    b = batch_size
    L = noise_svgp.shape[1]
    G = y.shape[1]
    with (torch.no_grad() if svgp_no_grad else torch.enable_grad()):
        q_mu, q_var = svgp_encoder(P, y[:b], train)
        rec, kl, pm, pv = [], [], [], []
        for l in range(L):
            m_l, v_l, mu_hat, A_hat = svgp.approximate_posterior_params(x[:b], x[:b], q_mu[:, l], q_var[:, l])
            r_l, k_l = svgp.variational_loss(x[:b], q_mu[:, l], q_var[:, l], mu_hat, A_hat)
            rec.append(r_l); kl.append(k_l); pm.append(m_l); pv.append(v_l)
        elbo = torch.stack(rec).sum() - (b / svgp.N_train) * torch.stack(kl).sum()
        p_m, p_v = torch.stack(pm, dim=1), torch.stack(pv, dim=1)
        ce = gauss_cross_entropy(p_m, p_v, q_mu, q_var).sum()
        diff = ce - elbo
        svgp_kl = (-diff if ce.item() > elbo.item() else diff) / L      # SpaDOT.py:76-77 sign trick
    z_svgp = p_m + noise_svgp * torch.sqrt(p_v)
    g_mu, g_var = gat_encoder(P, y, edge_index, heads)
    g_mu, g_var = g_mu[:b], g_var[:b]
    z_gat = g_mu + noise_gat * torch.sqrt(g_var)
    gat_kl = -0.5 * torch.sum(1 + torch.log(g_var) - g_mu ** 2 - g_var) / L
    z = torch.cat([z_svgp, z_gat], dim=1)
    recon = torch.sum((y[:b] - decoder(P, z)) ** 2) / G
    align = torch.sum((z_svgp.norm(dim=1) / L - z_gat.norm(dim=1) / L) ** 2)
    inter = dict(q_mu=q_mu, q_var=q_var, p_m=p_m, p_v=p_v, g_mu=g_mu, g_var=g_var, elbo=elbo, ce=ce)
    return (recon, svgp_kl, gat_kl, align, z), inter


def all_latent_samples(P, svgp, X, Y, edge_index, heads, L, mean_only=False):
    """SpaDOT.py:96-123 (eval mode: BatchNorm running stats; posterior means, no noise).  mean_only: see
    SVGPOracle.approximate_posterior_params (full-size checks: the N_t x N_t intermediates are not formed)."""
    q_mu, q_var = svgp_encoder(P, Y, train=False)
    pm = [svgp.approximate_posterior_params(X, X, q_mu[:, l], q_var[:, l], mean_only=mean_only)[0] for l in range(L)]
    g_mu, _ = gat_encoder(P, Y, edge_index, heads)
    return torch.cat([torch.stack(pm, dim=1), g_mu], dim=1)


# ----------------------------------------------------------------------------- regulariser glue

def beta_cycle_linear(n_iter, start=0.0, stop=1.0, n_cycle=10, ratio=1.0):
    """_train_utils.py:143-153."""
    L = np.ones(n_iter) * stop
    period = n_iter / n_cycle
    step = (stop - start) / (period * ratio)
    for c in range(n_cycle):
        v, i = start, 0
        while v <= stop and int(i + c * period) < n_iter:
            L[int(i + c * period)] = v
            v += step
            i += 1
    return L


def kmeans_loss(latent, centers, labels):
    """_train_utils.py:240-253: ||z - c[label]||_F^2 / z_dim / (#distinct labels in the batch)."""
    labels = [int(l) for l in labels]
    c = torch.as_tensor(centers, dtype=latent.dtype)[labels]
    return torch.sum(torch.norm(latent - c) ** 2 / latent.shape[1] / len(set(labels)))


def ot_loss(latent, labels, all_labels, cur_centers, prev_centers, gamma):
    """_train_utils.py:272-307: batch cluster means (stored centre when a cluster is absent from the
    batch), row-normalised gamma (NaN/inf -> 0), mean(gamma * cdist(prev_centres, batch_centres))."""
    clusters = sorted(set(int(c) for c in all_labels))
    rows = []
    labels = np.asarray(labels)
    for c in clusters:
        idx = np.nonzero(labels == c)[0]
        if idx.size == 0:
            rows.append(torch.as_tensor(cur_centers[c], dtype=latent.dtype))
        else:
            rows.append(latent[torch.as_tensor(idx)].mean(dim=0))
    cur = torch.stack(rows)
    g = np.asarray(gamma, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = g / g.sum(axis=1, keepdims=True)
    g = np.nan_to_num(g, nan=0.0, posinf=0.0, neginf=0.0)
    cost = torch.cdist(torch.as_tensor(prev_centers, dtype=latent.dtype), cur, p=2)
    return torch.mean(torch.as_tensor(g, dtype=latent.dtype) * cost)


# ----------------------------------------------------------------------------- graph + batches

def knn_graph(coords, k):
    """_utils.py:52-100 without the dense adjacency: directed edges i -> j for j in kNN(i) (self
    excluded), plus one self loop per node; returned in the row-major order dense_to_sparse gives
    (sorted by source, then target).  Row 0 = source, row 1 = target (int64)."""
    from sklearn.neighbors import NearestNeighbors
    n = coords.shape[0]
    kk = min(k + 1, n)
    _, idx = NearestNeighbors(n_neighbors=kk, algorithm="auto").fit(coords).kneighbors(coords)
    pairs = set()
    for i in range(n):
        for j in idx[i, 1:k + 1]:
            pairs.add((i, int(j)))
        pairs.add((i, i))
    e = np.array(sorted(pairs), dtype=np.int64).T
    return torch.as_tensor(e)


def induced_batch(edge_index, n_nodes, seeds, hops=2):
    """NeighborLoader(num_neighbors=[f, f], subgraph_type='induced') with f >= max in-degree
    (SURVEY App. B): nodes = seeds + their `hops`-hop in-neighbourhood (sources of edges pointing at
    the frontier), seeds first, the rest in order of discovery by ascending node id per hop; edges =
    every original edge with both ends in the node set, relabelled.  Returns (n_id, edge_index_sub)."""
    src, dst = edge_index[0].numpy(), edge_index[1].numpy()
    order = np.argsort(dst, kind="stable")
    s_sorted, d_sorted = src[order], dst[order]
    ptr = np.searchsorted(d_sorted, np.arange(n_nodes + 1))
    seen = np.zeros(n_nodes, dtype=bool)
    n_id = [int(s) for s in seeds]
    seen[n_id] = True
    frontier = list(n_id)
    for _ in range(hops):
        new = set()
        for t in frontier:
            for s in s_sorted[ptr[t]:ptr[t + 1]]:
                if not seen[s]:
                    new.add(int(s))
        new = sorted(new)
        seen[new] = True
        n_id.extend(new)
        frontier = new
    n_id = np.array(n_id, dtype=np.int64)
    relabel = -np.ones(n_nodes, dtype=np.int64)
    relabel[n_id] = np.arange(n_id.size)
    keep = seen[src] & seen[dst]
    sub = np.stack([relabel[src[keep]], relabel[dst[keep]]])
    return torch.as_tensor(n_id), torch.as_tensor(sub)


# ----------------------------------------------------------------------------- one optimizer step

def step_loss(P, svgp, x, y, edge_index, batch_size, heads, noise_svgp, noise_gat, weights, km=None, ot=None, svgp_no_grad=False):
    """Forward + composite loss of one batch (_train_utils.py:193-212).  Returns (loss, terms dict of 0-dim
    tensors, final_latent)."""
    (recon, skl, gkl, align, z), _ = spadot_forward(P, svgp, x, y, edge_index, batch_size, heads, noise_svgp,
                                                    noise_gat, train=True, svgp_no_grad=svgp_no_grad)
    l1, b1, b2, o1, o2, o3 = weights
    kml = kmeans_loss(z, km[0], km[1]) if km is not None else torch.zeros((), dtype=z.dtype)
    otl = ot_loss(z, *ot) if ot is not None else torch.zeros((), dtype=z.dtype)
    loss = l1 * recon - b1 * skl + b2 * gkl + o1 * align + o2 * kml + o3 * otl
    return loss, dict(elbo=loss, Recon=recon, SVGP_KL=skl, GAT_KL=gkl, alignment=align, KMeans=kml, OT=otl), z


def training_step(P, svgp, x, y, edge_index, batch_size, heads, noise_svgp, noise_gat, weights,
                  km=None, ot=None, lr=3e-4, max_norm=0.3, opt_state=None, detail=None):
    """_train_utils.py:187-217 on the CPU in fp64: forward, composite loss, backward (torch autograd),
    clip_grad_norm_(0.3), AdamW(lr) step.  P: dict name -> leaf tensor (requires_grad for parameters).
    weights = (lambda1, beta1, beta2, omiga1, omiga2, omiga3); km = (centers, labels) or None;
    ot = (labels, all_labels, cur_centers, prev_centers, gamma) or None.
    detail (optional dict): receives 'latent' (final_latent, detached) and 'grads' (name -> gradient BEFORE the
    clip, detached clones) -- what a parity check of the device step compares against.
    Returns (loss terms dict, optimizer) -- the optimizer can be passed back in as opt_state."""
    params = [v for k, v in P.items() if v.requires_grad]
    opt = opt_state if opt_state is not None else torch.optim.AdamW(params, lr=lr)
    loss, terms, z = step_loss(P, svgp, x, y, edge_index, batch_size, heads, noise_svgp, noise_gat, weights, km, ot)
    opt.zero_grad()
    loss.backward()
    if detail is not None:
        detail["latent"] = z.detach().clone()
        detail["grads"] = {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v))
                           for k, v in P.items() if v.requires_grad}
    torch.nn.utils.clip_grad_norm_(params, max_norm)
    opt.step()
    return {k: float(v.detach()) for k, v in terms.items()}, opt
