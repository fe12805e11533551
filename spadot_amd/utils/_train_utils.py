"""Training driver: mirror of /root/reference/SpaDOT/utils/_train_utils.py (same function names and
meaning) with the whole step resident on the MI355X.

What differs from the reference, on purpose (SURVEY 7 "hard parts", App. D):
  * no dense N_t x N_t adjacency and no torch_geometric: the kNN graph is CSR from the start and the
    (unshuffled, hence epoch-invariant) NeighborLoader batches are built once (spadot_amd.graph);
  * expression, coordinates, graphs, K-means labels/centres and OT plans live on the device; the
    step has no .item() / per-spot Python dict look-ups (losses are read back once per epoch);
  * clip_grad_norm_(0.3) + AdamW.step are two HIP kernels over one flat parameter buffer, which is
    also what the data-parallel path all-reduces (spadot_amd.parallel);
  * backward(retain_graph=True) is not replicated (App. D.11);
  * only the first growth solve of compute_transport_map is run (it is the one returned, App. D.1).
"""
import contextlib
import os
import random
from collections import OrderedDict
from collections.abc import Mapping
from time import time

import numpy as np
import torch

from ..graph import build_batch_graph, knn_graph, precompute_batches
from ..model import SpaDOT
from ..ops import FlatAdamW, cluster_losses, cluster_losses_fb, mix_losses
from .OT_loss.ot_solvers import compute_transport_map, compute_transport_maps

LOSS_NAMES = ["elbo", "Recon", "SVGP_KL", "GAT_KL", "alignment", "KMeans", "OT"]


def _obtain_tp_loc_info(adata):
    """_train_utils.py:118-140: per-time-point standardised coordinates ++ one-hot time point."""
    tps = np.asarray(adata.obs["timepoint"])
    uniq = sorted(set(tps.tolist()))
    code = np.array([uniq.index(t) for t in tps.tolist()], dtype=int)
    tp_mat = np.zeros((code.size, len(uniq)))
    tp_mat[np.arange(code.size), code] = 1
    loc = np.asarray(adata.obsm["spatial"], dtype=np.float64)
    loc_scaled = np.zeros(loc.shape, dtype=np.float64)
    for i in range(len(uniq)):
        sel = code == i
        tp_loc = loc[sel]
        sd = tp_loc.std(axis=0)
        sd[sd == 0] = 1.0                                   # StandardScaler's zero-variance rule
        loc_scaled[sel] = (tp_loc - tp_loc.mean(axis=0)) / sd
    return np.concatenate((loc_scaled, tp_mat), axis=1)


def prepare_dataloader(adata, model_config):
    """_train_utils.py:37-94.  Returns the same dict keys the reference's callers use
    ('inducing_points', 'N_train', 'dataloaders', 'datasets') plus 'graphs' (full time-point CSR for
    inference); 'adjacency_matrices' (dense N_t^2) is deliberately not built."""
    device = torch.device(model_config["device"])
    from ._utils import resolve_compute_dtype
    store = model_config["compute_dtype"] = resolve_compute_dtype(model_config.get("compute_dtype"))
    loc = _obtain_tp_loc_info(adata)
    inducing_idx = random.sample(range(loc.shape[0]), model_config["inducing_point_nums"])
    inducing_points = loc[inducing_idx, :]
    timepoints = model_config["timepoints"]
    tp_to_idx = {v: k for k, v in enumerate(timepoints)}
    idx = np.argmax(loc[:, 2:], axis=1)
    tp_index = {tp: np.nonzero(idx == tp_to_idx[tp])[0] for tp in timepoints}
    inducing_tp = np.argmax(inducing_points[:, 2:], axis=1)
    inducing_points_dict, N_train_dict = OrderedDict(), OrderedDict()
    for tp in timepoints:
        inducing_points_dict[tp] = inducing_points[np.where(inducing_tp == tp_to_idx[tp])[0], :2]
        N_train_dict[tp] = int(np.sum(np.asarray(adata.obs["timepoint"]) == tp))
    dataloaders, datasets, graphs = OrderedDict(), OrderedDict(), OrderedDict()
    owned = model_config.get("owned_timepoints", timepoints)
    X = adata.X
    spatial = np.asarray(adata.obsm["spatial"], dtype=np.float64)
    for tp in timepoints:
        if tp not in owned:
            continue
        ix = tp_index[tp]
        n = ix.size
        k_cut = min(model_config["max_neighbors"], model_config["knn_cutoff"] * round(1 / 1000 * n))
        print("Calculating spatial graph...")
        ei = knn_graph(spatial[ix], k_cut, max_neigh=model_config["max_neighbors"],
                       backend=model_config.get("knn_backend", "sklearn"), device=device)
        print("The graph contains %d edges, %d cells." % (ei.shape[1] - n, n))
        Y = torch.as_tensor(np.ascontiguousarray(np.asarray(X[ix]))).to(device=device, dtype=store)
        datasets[tp] = (torch.as_tensor(loc[ix, :2]).to(device), Y, ix)
        # block plans for the matrix-core GAT edge kernels: bf16 rows only (fp32 compute keeps the per-edge kernels)
        plans = store == torch.bfloat16 and model_config.get("gat_block_plans", True)
        # data-parallel runs at batch granularity (spadot_amd.parallel.configure_shard): only this rank's batches are
        # built; the others stay None in the list (its length is still the time point's batch count)
        only = model_config.get("owned_batches")
        dataloaders[tp] = precompute_batches(ei, n, model_config["batch_size"], device, coords=spatial[ix], plans=plans,
                                             only=None if only is None else set(only.get(tp, ())))
        from ..graph import morton_key
        graphs[tp] = build_batch_graph(ei, n, device, order_key=morton_key(spatial[ix]), plans=plans)
    _cache_batch_inputs(dataloaders, datasets, model_config)
    return {"inducing_points": inducing_points_dict, "N_train": N_train_dict, "dataloaders": dataloaders,
            "datasets": datasets, "graphs": graphs}


def _cache_batch_inputs(dataloaders, datasets, model_config):
    """The loader is not shuffled, so every batch gathers the same rows each epoch: keep them gathered in
    HBM (288 GB: cfg3 needs 6 GB) instead of re-gathering ~60 MB per step.  In a 16-bit compute dtype the
    gene axis is zero-padded to a multiple of 128 (the first GAT GEMM runs ~20 % faster on an aligned K), and so is the
    ROW count (model_config['batch_row_pad'], default 128; 0: off): the library's GEMMs on 9 980 rows run 10-17 % slower
    than on 10 112 (tools/gemm_mpad.py: layer 1 forward 143 -> 118 us, its weight gradient 190 -> 161 us, layer 2 forward
    104 -> 89 us).  The zero rows are no nodes of the graph: the GAT kernels index rows by node id, hand the padding on
    to the first layer's output and give it a zero gradient (ops.gat_edge).
    model_config['batch_cache_gb'] (default 64) bounds the cache; beyond it batches gather per step."""
    budget = float(model_config.get("batch_cache_gb", 64.0)) * 2 ** 30
    need = 0
    for tp, batches in dataloaders.items():
        Y = datasets[tp][1]
        G = Y.shape[1]
        Gp = (G + 127) // 128 * 128 if Y.element_size() == 2 else G
        need += sum(b.n_id.numel() + 128 for b in batches if b is not None) * Gp * Y.element_size()
    if need > budget:
        return False
    for tp, batches in dataloaders.items():
        loc, Y, _ = datasets[tp]
        G = Y.shape[1]
        Gp = (G + 127) // 128 * 128 if Y.element_size() == 2 else G
        rpad = int(model_config.get("batch_row_pad", 128)) if Y.element_size() == 2 else 0
        for b in batches:
            if b is None:
                continue
            b.x = loc[b.n_id]
            nrow = b.n_id.numel()
            nrow_p = (nrow + rpad - 1) // rpad * rpad if rpad > 0 else nrow
            if Gp == G and nrow_p == nrow:
                b.y = Y[b.n_id]
            else:
                b.y = torch.zeros((nrow_p, Gp), dtype=Y.dtype, device=Y.device)
                b.y[:nrow, :G] = Y[b.n_id]
            if Y.dtype != torch.float32:       # the seeds' rows once more in fp32: what the reconstruction term reads
                b.y_seed32 = Y[b.n_id[:b.batch_size]].float()
    return True


def get_latent(model, model_config, adata, dataloader_dict):
    """_train_utils.py:98-116 without AnnData: returns (latent [N, z_dim] in the order of adata rows of the
    owned time points, row indices)."""
    model.eval()
    lat, rows = [], []
    with torch.no_grad():
        for tp in dataloader_dict["datasets"]:
            loc, Y, ix = dataloader_dict["datasets"][tp]
            lat.append(model.all_latent_samples(loc, Y, dataloader_dict["graphs"][tp], tp))
            rows.append(ix)
    return np.concatenate(lat), np.concatenate(rows)


def _beta_cycle_linear(n_iter, start=0.0, stop=1, n_cycle=10, ratio=1):
    """_train_utils.py:143-153."""
    L = np.ones(n_iter) * stop
    period = n_iter / n_cycle
    step = (stop - start) / (period * ratio)
    for c in range(n_cycle):
        v, i = start, 0
        while v <= stop and (int(i + c * period) < n_iter):
            L[int(i + c * period)] = v
            v += step
            i += 1
    return L


# ------------------------------------------------------------------------------ regularisers

def _device_state(model, tp):
    """Per-time-point K-means state mirrored on the device: labels by local spot id, centres, and the
    sorted list of cluster ids that occur in the time point."""
    return model._kmeans_dev[tp]


def _compute_kmeans_loss(model, model_config, tp, seed_ids, latent):
    """_train_utils.py:240-253: ||z - c[label]||_F^2 / z_dim / (#distinct labels in the batch).
    `seed_ids` are local spot ids of the time point (device int64)."""
    st = _device_state(model, tp)
    return cluster_losses(latent, st["labels"], seed_ids, st["centers"], do_km=True, do_ot=False)[0]


def _compute_OT_loss(model, model_config, cur_tp, seed_ids, tp_p_m, prev_tp):
    """_train_utils.py:272-307 on the device: batch means per cluster (stored centre when the cluster is
    absent from the batch), row-normalised plan with NaN/inf -> 0, mean(gamma * cdist)."""
    st = _device_state(model, cur_tp)
    return cluster_losses(tp_p_m, st["labels"], seed_ids, st["centers"], _device_state(model, prev_tp)["centers"],
                          model._gamma_dev[f"{prev_tp}_{cur_tp}"], st["cluster_list"], do_km=False, do_ot=True)[1]


def _assign_or_copy(store, key, value):
    """Keep device tensors at stable addresses (captured hipGraphs read them): copy in place when the shape
    is unchanged, else replace and report it so that stale graphs can be dropped."""
    cur = store.get(key)
    if cur is not None and cur.shape == value.shape and cur.dtype == value.dtype:
        cur.copy_(value)
        return False
    store[key] = value
    return cur is not None


class _IndexToLabel(Mapping):
    """{global spot index: label} of one time point (the reference's kmeans_index_dict[tp], SpaDOT.py:49) as a read-only
    mapping whose dict is built from its two arrays on first use: nothing on the training path reads it (the step gets
    its labels from the device mirror), and building 10^4-entry dicts for every time point every epoch cost
    milliseconds of pure Python."""

    def __init__(self, idx, labels):
        self._pending, self._d = (np.asarray(idx), np.asarray(labels)), None

    def _dict(self):
        if self._d is None:
            idx, labels = self._pending
            self._d, self._pending = dict(zip(idx.tolist(), labels.tolist())), None
        return self._d

    def __getitem__(self, k):
        return self._dict()[k]

    def __iter__(self):
        return iter(self._dict())

    def __len__(self):
        return len(self._pending[0]) if self._d is None else len(self._d)

    def __repr__(self):
        return repr(self._dict())


def _set_kmeans_state(model, tp, centers, labels, global_idx, device):
    """Stores the reference's three dicts (SpaDOT.py:47-50) and their device mirror."""
    model.kmeans_center_dict[tp] = centers
    model.kmeans_cluster_dict[tp] = labels.tolist()
    model.kmeans_index_dict[tp] = _IndexToLabel(global_idx, labels)
    if not hasattr(model, "_kmeans_dev"):
        model._kmeans_dev, model._gamma_dev = {}, {}
    cl = sorted(set(labels.tolist()))
    st = model._kmeans_dev.setdefault(tp, {})
    changed = _assign_or_copy(st, "labels", torch.as_tensor(labels, dtype=torch.int64).to(device))
    changed |= _assign_or_copy(st, "centers", torch.as_tensor(centers, dtype=torch.float32).to(device))
    changed |= _assign_or_copy(st, "cluster_list", torch.as_tensor(cl, dtype=torch.int64).to(device))
    if changed:
        model._state_version = getattr(model, "_state_version", 0) + 1


def _update_Kmeans(model, model_config, dataloader_dict, timepoints=None):
    """_train_utils.py:255-269: full-time-point inference + KMeans(n_clusters, random_state=seed, n_init=10)
    per time point.  model_config['kmeans_backend']: 'device' (default: spadot_amd.kmeans.KMeansDevice, the same
    algorithm with the latents left in HBM -- at cfg3 the refit of all time points takes ~0.03 s per epoch against
    ~0.28 s, i.e. more than the epoch's 100 training steps, on the host) or 'sklearn' (the reference's host fit, kept
    as the parity option: labels for given centres are bit-identical either way, the fit itself is third-party RNG).
    timepoints (extension, data-parallel runs): refit only these (default: every time point this process holds)."""
    model.eval()
    device = torch.device(model_config["device"])
    backend = model_config.get("kmeans_backend", "device")
    tps = [tp for tp in dataloader_dict["datasets"] if timepoints is None or tp in timepoints]
    with torch.no_grad():
        if backend == "device":
            # all inferences first (no host synchronisation between them), then ONE batched fit for all time points
            # (spadot_amd.kmeans.fit_many: selection rounds and Lloyd iterations of all T x n_init restarts together)
            from ..kmeans import fit_many
            lat = []
            for tp in tps:
                loc, Y, ix = dataloader_dict["datasets"][tp]
                lat.append(model.all_latent_samples(loc, Y, dataloader_dict["graphs"][tp], tp, as_numpy=False))
            fits = fit_many(lat, model_config["n_clusters"], random_state=model_config["seed"], n_init=10) if lat else []
            for tp, km in zip(tps, fits):
                _set_kmeans_state(model, tp, km.cluster_centers_, km.labels_, dataloader_dict["datasets"][tp][2], device)
            return
        for tp in tps:
            loc, Y, ix = dataloader_dict["datasets"][tp]
            from sklearn.cluster import KMeans
            latent = model.all_latent_samples(loc, Y, dataloader_dict["graphs"][tp], tp)
            km = KMeans(n_clusters=model_config["n_clusters"], random_state=model_config["seed"], n_init=10).fit(latent)
            _set_kmeans_state(model, tp, km.cluster_centers_, km.labels_, ix, device)


def _set_gamma(model, key, gamma, device):
    model.gammas[key] = gamma
    with np.errstate(divide="ignore", invalid="ignore"):
        g = gamma / gamma.sum(axis=1, keepdims=True)
    g = np.nan_to_num(g, nan=0.0, posinf=0.0, neginf=0.0)
    if _assign_or_copy(model._gamma_dev, key, torch.as_tensor(g, dtype=torch.float32).to(device)):
        model._state_version = getattr(model, "_state_version", 0) + 1


def _update_OT_matrix(model, model_config):
    """_train_utils.py:309-321: transport plan between the K-means centres of consecutive time points.  All pairs are
    solved by ONE launch (ot_solvers.compute_transport_maps -> csrc/ot_small.hip: one wavefront per pair, cost, median,
    six epsilon stages and the row-normalised plan of _compute_OT_loss on the device); the plans come back in one copy
    for the reference's `model.gammas`.  Centre sets beyond 64 clusters take the streaming solver pair by pair."""
    model.eval()
    device = torch.device(model_config["device"])
    timepoints = model_config["timepoints"]
    keys, pairs = [], []
    for tp_i, tp in enumerate(timepoints[:-1]):
        nxt = timepoints[tp_i + 1]
        if tp not in model.kmeans_center_dict or nxt not in model.kmeans_center_dict:
            continue
        keys.append(f"{tp}_{nxt}")
        pairs.append((np.asarray(model.kmeans_center_dict[tp]), np.asarray(model.kmeans_center_dict[nxt])))
    if not pairs:
        return
    if not hasattr(model, "_gamma_dev"):
        model._kmeans_dev, model._gamma_dev = getattr(model, "_kmeans_dev", {}), {}
    outs = []
    for key, (a, b) in zip(keys, pairs):               # stable addresses: captured graphs read these tensors
        cur = model._gamma_dev.get(key)
        if cur is None or tuple(cur.shape) != (a.shape[0], b.shape[0]):
            if cur is not None:
                model._state_version = getattr(model, "_state_version", 0) + 1
            cur = model._gamma_dev[key] = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=device)
        outs.append(cur)
    plans = compute_transport_maps(pairs, model_config["ot_config"], gamma_out=outs, device=device)
    if plans is not None:
        for key, gamma in zip(keys, plans):
            model.gammas[key] = gamma
        return
    for key, (a, b) in zip(keys, pairs):
        # (plans is None: a pair exceeded the small solver's range or hit its iteration cap -- not through it again)
        gamma = compute_transport_map(a, b, model_config["ot_config"], G=None, device=device, skip_small=True)
        _set_gamma(model, key, gamma, device)


# ------------------------------------------------------------------------------ the step

def _cluster_terms(model, model_config, tp, tp_i, seed_ids, z, do_km, do_ot, weights=None):
    """(K-means loss, OT loss) of one batch through ops.cluster_losses (one launch each way); zeros for the
    terms that are not active yet (_train_utils.py:198-204).
    weights (the device vector of _loss_weights): (km, ot, dz) through ops.cluster_losses_fb instead -- the gradient
    dz = d(omiga2 km + omiga3 ot)/dz comes out of the forward launch and the caller routes it -- or None when that kernel does
    not take the shape."""
    if not (do_km or do_ot):
        zero = torch.zeros((), dtype=torch.float32, device=z.device)
        return zero, zero
    st = _device_state(model, tp)
    prev = gamma = None
    if do_ot:
        prev_tp = model_config["timepoints"][tp_i - 1]
        prev = _device_state(model, prev_tp)["centers"]
        gamma = model._gamma_dev[f"{prev_tp}_{tp}"]
    if weights is not None:
        return cluster_losses_fb(z, st["labels"], seed_ids, st["centers"], prev, gamma, st["cluster_list"], do_km, do_ot,
                                 weights[4], weights[5])
    return cluster_losses(z, st["labels"], seed_ids, st["centers"], prev, gamma, st["cluster_list"], do_km, do_ot)


def _loss_weights(model, model_config, beta1):
    """Device vector (lambda1, -beta1, beta2, omiga1, omiga2, omiga3) of _train_utils.py:205-212.  `beta1` may
    already be such a vector (GraphedStepper keeps one at a fixed address and rewrites entry 1)."""
    if isinstance(beta1, torch.Tensor) and beta1.numel() == 6:
        return beta1
    return torch.tensor([model_config["lambda1"], -float(beta1), model_config["beta2"], model_config["omiga1"],
                         model_config["omiga2"], model_config["omiga3"]], dtype=torch.float32,
                        device=next(model.parameters()).device)


def forward_backward(model, model_config, dataloader_dict, tp_i, tp, bi, epoch, beta1, optimizer=None, noise=None,
                     latent_out=None):
    """Forward of batch `bi` of time point `tp`, composite loss (_train_utils.py:193-212) and backward
    into the parameters' .grad (the flat gradient buffer; with `optimizer` a FlatAdamW, through its
    backward(), which overwrites the buffer and so needs no zero_grad()).  Returns the seven loss terms
    as a device tensor (no host sync).
    noise = (eps_svgp, eps_gat), each [b, L]: replaces the two reparameterisation draws (SpaDOT.py:78,83) so that
    the step can be compared with a host restatement fed the same numbers; latent_out (a list) receives the
    batch's final_latent.  Both are for parity checks; training passes neither."""
    batch = dataloader_dict["dataloaders"][tp][bi]
    loc, Y, _ = dataloader_dict["datasets"][tp]
    if batch.y is not None:                                    # gathered once in prepare_dataloader
        x_b, y_b = batch.x, batch.y
    else:
        x_b, y_b = loc[batch.n_id], Y[batch.n_id]
    seeds = batch.n_id[:batch.batch_size]
    recon, svgp_kl, gat_kl, align, z = model.forward(x=x_b, y=y_b, edge_index=batch.graph, tp=tp,
                                                     batch_size=batch.batch_size, batch_key=(tp, bi), noise=noise,
                                                     y_seed32=batch.y_seed32 if batch.y is not None else None)
    if latent_out is not None:
        latent_out.append(z.detach())
    do_km = epoch >= 1
    do_ot = bool(epoch >= model_config["ot_epoch"] and tp_i != 0)
    km, ot = _cluster_terms(model, model_config, tp, tp_i, seeds, z, do_km, do_ot)
    elbo, losses = mix_losses(_loss_weights(model, model_config, beta1), (recon, svgp_kl, gat_kl, align, km, ot))
    if optimizer is not None and hasattr(optimizer, "backward"):
        optimizer.backward(elbo)
    else:
        elbo.backward()
    return losses


def _ranks_share_a_device(device):
    """True when two ranks of the process group run on the same GPU (one-GPU rehearsals).  Collective: every rank
    calls it."""
    import socket
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return False
    idx = device.index if device.index is not None else torch.cuda.current_device()
    prop = torch.cuda.get_device_properties(idx)
    me = (socket.gethostname(), getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", idx), getattr(prop, "pci_device_id", 0))
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, me)
    return len(set(everyone)) < len(everyone)


class GraphedStepper:
    """Training steps as replayed hipGraphs (torch.cuda.CUDAGraph on ROCm = hipGraph).

    The unshuffled loader makes every (time point, batch) recur each epoch with identical shapes and
    index tensors, and the step has no host synchronisation, so the launches of one step (forward,
    backward, clip, AdamW, on two streams) are captured once per (time point, batch, active loss terms)
    and replayed.  First visit of a key runs eagerly (warm-up: library handles, SVGP batch constants), the second
    visit captures, later visits replay.  Anything that changes between replays lives in device memory at a fixed
    address: beta1 (entry 1 of the loss-weight vector), K-means labels/centres and OT plans (copied in place), the
    optimizer's step count.

    Arrangements (DESIGN section 7 has the table; round 5 pruned the measured-and-lost ones):
      single graph   `staged_graphs: false` (and ranks that SHARE a device): the whole step body as one graph per key;
      staged         the default: EIGHT graphs per key on two streams --
                         gat_fwd (main) || svgp_fwd (side)
                         tail: loss tail forward + its backward (main) || svgp_pre: rest of the ELBO + the
                               gradient-independent half of the SVGP backward (side)
                         gat_bwd_a: head, layer 3, layer 2's edge phase (main)
                         gat_bwd_b: layer 2's dense map, layer 1 (main) || svgp_bwd, then `late`: the gradient work
                               that only the optimizer reads, queued by the stages above (ops.DEFERRED) (side)
                     + the update as two graphs (gradient norm + SVGP encoder, then the rest: chained());
      staged + bucketed exchange   the same graphs between the two collectives of a data-parallel step (grad_sync_async);
      eager          no stepper at all (training_step / forward_backward).
    """

    STAGES = ("gat_fwd", "svgp_fwd", "tail", "svgp_bwd", "gat_bwd_a", "gat_bwd_b", "late", "svgp_pre")
    SIDE_STAGES = (1, 3, 6, 7)

    def __init__(self, model, optimizer, model_config, dataloader_dict, grad_sync=None, grad_sync_async=None):
        """grad_sync (data-parallel replicas): callable on the flat gradient buffer, e.g. an RCCL all-reduce, issued
        between the replay of the forward + backward graph(s) and the clip + AdamW graphs (nothing of the collective is
        captured).
        grad_sync_async: callable on a slice of the flat gradient buffer returning a handle with .wait() (e.g.
        `dist.all_reduce(view, async_op=True)`).  With it, staged graphs and an optimizer built with
        `last=model.GATEncoder.first_layer_parameters()`, the exchange is BUCKETED: everything but the first GAT layer's
        gradients leaves when the queue of deferred gradient work has run (the side stream: ~135 us before the main stream
        finishes the first layer's backward), the first layer's gradients when the main stream has.  Every rank issues the
        same two collectives per step in the same order, whatever path it takes."""
        self.model, self.opt, self.cfg, self.dd = model, optimizer, model_config, dataloader_dict
        self.grad_sync, self.grad_sync_async = grad_sync, grad_sync_async
        self.beta1_t = _loss_weights(model, model_config, 0.0)     # (lambda1, -beta1, beta2, omiga1..3); entry 1 rewritten per step
        self.graphs, self.seen = {}, set()
        self.opt_graph = None
        self.pool = self.pool_side = None
        self._groups = None
        # staged replay is the default, except for replicas that SHARE a device (the one-GPU rehearsal of the multi-rank
        # path): there the multi-stream replays of the processes time-slice against each other (2 steps/s against 97), so
        # those keep the single-graph form
        import torch.distributed as dist
        multi = grad_sync is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        shared = multi and _ranks_share_a_device(torch.device(model_config["device"]))
        self.staged = bool(model_config.get("staged_graphs", not shared))
        first = {id(p) for p in model.GATEncoder.first_layer_parameters()}
        self.overlap = bool(self.staged and grad_sync_async is not None and optimizer.tail_offset is not None
                            and {id(p) for p in optimizer.tail_params} == first
                            and model_config.get("overlap_grad_sync", True))
        # every launch of the step is ours or a plain library GEMM (the SPD inverse has no library factorisation at
        # any number of inducing points: ops._spd_inverse_logdet_nograd), so the step is always capturable
        self.capturable = True
        self._beta1 = None
        # step() returns a copy of the graph's loss vector by default; a caller that consumes it on the same stream before its
        # next step (the training loop's `tot += ...`) may take the graph's own buffer and save the copy launch
        self.clone_output = True
        self.version = getattr(model, "_state_version", 0)
        # parity checks only: with keep_latents set before a key is captured, latents[(tp, batch)] is the final_latent
        # tensor of that key's tail graph (rewritten by every replay)
        self.keep_latents, self.latents = False, {}
        # the optimizer keeps the bf16 images of the GAT layers' weights current (its update kernel writes them), so the
        # steps need no weight-cast launch; anything else that rewrites parameters must refresh them
        if model_config.get("optimizer_weight_images", True) and hasattr(optimizer, "maintain_image"):
            object.__setattr__(model.GATEncoder, "_image_optimizer", optimizer)
            object.__setattr__(model.decoder, "_image_optimizer", optimizer)       # (the output map's weight: decoder.py)
            if not getattr(model, "_image_refresh_hook", False):
                # (whichever optimizer is pinned WHEN a state_dict is loaded, not the one that was pinned first)
                def _refresh(mod, incompatible):
                    o = getattr(mod.GATEncoder, "_image_optimizer", None)
                    if o is not None:
                        o.refresh_images()
                model.register_load_state_dict_post_hook(_refresh)
                object.__setattr__(model, "_image_refresh_hook", True)
        self._images_version = getattr(optimizer, "images_version", 0)
        # the update in two graphs around an event (see update(), chained()); needs FlatAdamW(first=...)
        self._head_event, self._head_ready, self._chain = None, False, False
        # gradient work that only the optimizer reads -- the second GAT layer's weight gradient and attention / bias sums, the
        # last layer's weight / attention-vector chain, the decoder output map's weight gradient, column sums of parameter
        # gradients, the loss VALUES -- is queued by the backward functions of the staged form (ops.DEFERRED) and replayed as
        # the `late` graph on the side stream: the layer's flag only PERMITS the queueing
        if self.staged:
            object.__setattr__(model.GATEncoder.gat2, "defer_wgrad", True)
        self._late_event = None
        self.stamps = (torch.zeros(32, dtype=torch.int64, device=next(model.parameters()).device)
                       if os.environ.get("SPADOT_STAMPS") == "1" else None)
        if self.stamps is not None:
            from .. import ops as _ops
            _ops.STAMP_BUF[0] = self.stamps
        # the model knows its steppers (weakly): its public entries that touch the encoder between steps break the chain
        import weakref
        reg = getattr(model, "_steppers", None)
        if reg is None:
            reg = weakref.WeakSet()
            object.__setattr__(model, "_steppers", reg)
        reg.add(self)

    def _body(self, tp_i, tp, bi, epoch, with_update=True):
        losses = forward_backward(self.model, self.cfg, self.dd, tp_i, tp, bi, epoch, self.beta1_t,
                                  optimizer=self.opt)
        if with_update:
            self.opt.step()
        return losses

    def _capture(self, fn):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: a process group's watchdog thread may touch the runtime while this thread captures
        with torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local"):
            out = fn()
        return g, out

    def update(self):
        """clip + AdamW as graphs of their own (shared by all keys): eager once, then replayed.  With an optimizer that has
        a `first` group (the SVGP encoder's parameters: FlatAdamW(first=...)) the update is TWO graphs -- gradient norm +
        the first group, then everything else -- with an event between them that the next step's SVGP branch waits for
        instead of the whole update (chained())."""
        if getattr(self.opt, "images_version", 0) != self._images_version:     # the update's image table changed:
            self._images_version = self.opt.images_version                      # a captured launch carries the old one
            self.opt_graph = None if self.opt_graph in (None, False) else False
        if getattr(self.opt, "head_count", 0) <= 0:
            if self.opt_graph is None:
                self.opt.step()
                self.opt_graph = False                  # warmed up; capture on the next call
            elif self.opt_graph is False:
                self.opt_graph, _ = self._capture(self.opt.step)
                self.opt_graph.replay()
            else:
                self.opt_graph.replay()
            return
        if self._head_event is None:
            self._head_event = torch.cuda.Event()
        main = torch.cuda.current_stream()
        if self.opt_graph is None:
            self.opt.step_head()
            self._head_event.record(main)
            self.opt.step_rest()
            self.opt_graph = False
        else:
            if self.opt_graph is False:
                head, rest = self.opt.step_head, self.opt.step_rest
                if self.stamps is not None:
                    from ..ops import stamp

                    def head():
                        stamp(self.stamps, 24)
                        self.opt.step_head()
                        stamp(self.stamps, 25)

                    def rest():
                        stamp(self.stamps, 26)
                        self.opt.step_rest()
                        stamp(self.stamps, 27)
                ga, _ = self._capture(head)
                gb, _ = self._capture(rest)
                self.opt_graph = (ga, gb)
            self.opt_graph[0].replay()
            self._head_event.record(main)
            self.opt_graph[1].replay()
        self._head_ready = True

    @contextlib.contextmanager
    def chained(self):
        """Consecutive step() calls with NOTHING else enqueued in between that reads or writes the SVGP encoder (its
        parameters, BatchNorm statistics) or the side stream's buffers -- the inner loop of an epoch.  Inside, the SVGP branch of
        step k + 1 waits only for the part of step k's update that writes ITS parameters (update()'s first graph), and runs
        beside the rest of the update and the first GAT GEMM instead of behind them: that branch (~0.6-0.7 ms beside the
        GAT GEMMs: encoder, 2 L inverses of m x m) bounded the forward pair of most steps by 0.12-0.23 ms (rocprofv3
        timeline, round 3).  The first step of a chain, and every step outside one, waits for the whole stream."""
        prev = self._chain
        self._chain, self._head_ready = True, False
        try:
            yield self
        finally:
            self._chain, self._head_ready = prev, False

    def barrier(self):
        """Call after enqueuing ANYTHING between two steps of a chain that reads or writes the SVGP encoder (parameters,
        BatchNorm statistics) or side-stream tensors on the main stream -- an EMA or logging read of the weights, an eval
        forward, average_buffers: the next step's SVGP branch then waits for the whole main stream again instead of only
        for the previous update's first graph.  model.eval() / model.train(), all_latent_samples and
        parallel.average_buffers call it themselves (SpaDOT._steppers); user hooks belong outside `with chained()` or
        must call this."""
        self._head_ready = False

    def _run(self, tp_i, tp, bi, epoch, beta1, with_update):
        # (a replayed graph holds no weight-cast launch: anything torch saw writing the GAT weights since their bf16 images
        # were cast -- load_state_dict, a broadcast, flat_param.copy_ -- is caught here, on EVERY path into a step: step()
        # and the data-parallel fb(); three integer compares otherwise)
        if getattr(self.model.GATEncoder, "_image_optimizer", None) is self.opt:
            self.opt.sync_images()
        if getattr(self.model, "_state_version", 0) != self.version:      # a state tensor was re-allocated
            self.graphs.clear()
            self.version = getattr(self.model, "_state_version", 0)
        if self._beta1 != float(beta1):               # (constant within an epoch: one fill launch per epoch, not per step)
            self.beta1_t[1].fill_(-float(beta1))
            self._beta1 = float(beta1)
        if self.staged:
            return self._run_staged(tp_i, tp, bi, epoch, with_update)
        key = (tp, bi, epoch >= 1, epoch >= self.cfg["ot_epoch"] and tp_i != 0, with_update)
        if key in self.graphs:
            g, out = self.graphs[key]
            g.replay()
            return out.clone() if self.clone_output else out
        if key not in self.seen:                                            # warm-up visit: plain eager step
            self.seen.add(key)
            return self._body(tp_i, tp, bi, epoch, with_update=with_update)
        g, out = self._capture(lambda: self._body(tp_i, tp, bi, epoch, with_update=with_update))
        self.graphs[key] = (g, out)
        g.replay()
        return out.clone() if self.clone_output else out

    # ---- staged mode.  A replayed hipGraph runs its two branches mostly one after the other (tools/graph_probe.py); two
    # graphs replayed on two streams do overlap.  Graphs that may run at the same time capture into different memory pools.
    def _stages(self, tp_i, tp, bi, epoch):
        model, cfg, dd, opt = self.model, self.cfg, self.dd, self.opt
        batch = dd["dataloaders"][tp][bi]
        b = batch.batch_size
        seeds = batch.n_id[:b]
        cached = batch.y is not None
        loc, Y, _ = dd["datasets"][tp]
        do_km = epoch >= 1
        do_ot = bool(epoch >= cfg["ot_epoch"] and tp_i != 0)
        P = self._param_groups()
        st = {}
        from .. import ops as _ops
        from ..model import svgp as _svgp

        # Batch inputs: the cached gathers (prepare_dataloader) live at fixed addresses.  Without the cache the rows
        # are gathered INSIDE the stages, i.e. inside the captured graphs -- a gather made out here would hand the
        # graphs the address of a temporary that is gone by the first replay.  Each branch gathers what it reads on
        # its own stream: the GAT branch all n_sub rows, the SVGP branch (and the tail after it) the seeds' rows.
        def gat_fwd():
            model.decoder.step_begin()         # (the loss tail's transposed weight image: made here, where this stream has slack)
            y_all = batch.y if cached else Y[batch.n_id]
            st["zg"] = model.branch_gat(y_all, batch.graph, b, taps=st)          # (taps: st["d2"], where the backward is cut)

        def svgp_fwd():
            st["xs"] = batch.x[:b] if cached else loc[seeds]
            st["ys"] = batch.y[:b] if cached else Y[seeds]
            # the part of the ELBO the tail does not wait for goes to the svgp_pre stage (svgp.ELBO_LATE)
            _svgp.ELBO_LATE[0] = st.setdefault("svgp_late", [])
            try:
                st["pm"], st["pv"], st["skl"] = model.branch_svgp(st["xs"], st["ys"], tp, b, batch_key=(tp, bi),
                                                                  y_seed32=getattr(batch, "y_seed32", None) if cached else None)
            finally:
                _svgp.ELBO_LATE[0] = None
            st["svgp_holder"] = _svgp.holder_of(st["pm"])

        def svgp_pre():             # rest of the ELBO + gradient-independent part of the SVGP backward (side stream, beside the tail)
            with torch.no_grad():
                for job in st.pop("svgp_late", []):
                    job()
            _svgp.precompute_backward(st["svgp_holder"])

        def tail():
            leaves = [st[k].detach().requires_grad_(True) for k in ("zg", "pm", "pv", "skl")]
            box = {}

            def hook(z):
                # K-means / OT terms AND their gradient w.r.t. z in one launch (the seeds of this stage's backward are the
                # loss weights: backward_partial's constant-one seed through mix_losses); the decoder's backward adds it
                if not (do_km or do_ot):
                    return None
                res = _cluster_terms(model, cfg, tp, tp_i, seeds, z, do_km, do_ot, weights=self.beta1_t)
                if res is None:
                    return None
                box["km"], box["ot"] = res[0], res[1]
                return res[2]

            recon, gkl, align, z = model.tail(leaves[0], leaves[1], leaves[2], st["ys"], b,
                                              y_seed32=getattr(batch, "y_seed32", None) if cached else None, z_hook=hook,
                                              recon_weight=self.beta1_t[0])      # (the seed mix_losses hands this term: lambda1)
            km, ot = (box["km"], box["ot"]) if "km" in box else _cluster_terms(model, cfg, tp, tp_i, seeds, z, do_km, do_ot)
            elbo, losses = mix_losses(self.beta1_t, (recon, leaves[3], gkl, align, km, ot))
            st["g"] = opt.backward_partial(elbo, None, P["tail"], extra_inputs=leaves)
            if self.keep_latents:
                self.latents[(tp, bi)] = z.detach()
            return losses

        def svgp_bwd():
            g = st["g"]
            opt.backward_partial([st["pm"], st["pv"], st["skl"]], [g[1], g[2], g[3]], P["svgp"])

        def gat_bwd_a():            # head, layer 3, the second layer's edge phase; stops at the second layer's dense output
            st["gd2"] = opt.backward_partial([st["zg"]], [st["g"][0]], P["gat_above_d2"], extra_inputs=[st["d2"]])[0]

        def gat_bwd_b():            # the second layer's dense map (its weight gradient queued) and layer 1
            opt.backward_partial([st["d2"]], [st["gd2"]], P["gat_below_d2"])

        def late():                 # what the three stages above queued: side stream, behind the SVGP backward
            # (no tape: the queued closures hold saved activations that require grad, and a library product written with
            # out= into a flat-gradient view would otherwise make the whole flat buffer a non-leaf -- found by the
            # ChickenHeart-shaped bf16 run of round 5, whose 747-spot time point takes the library path)
            with torch.no_grad():
                for job in st.pop("late", []):
                    job()

        def queued(fn):             # off-chain gradient work of `fn` goes to st["late"] instead of being launched
            def run():
                _ops.DEFERRED[0] = st.setdefault("late", [])
                try:
                    return fn()
                finally:
                    _ops.DEFERRED[0] = None
            return run

        fns = (gat_fwd, svgp_fwd, queued(tail), svgp_bwd, queued(gat_bwd_a), queued(gat_bwd_b), late, svgp_pre)
        if self.stamps is not None:
            # measurement aid (SPADOT_STAMPS=1): a device timestamp at the head and the end of every stage graph
            # (slots 2 k, 2 k + 1), read back by tools/stage_stamps.py -- when each stage really starts with no profiler attached
            from ..ops import stamp

            def stamped(k, fn):
                def run():
                    stamp(self.stamps, 2 * k)
                    r = fn()
                    stamp(self.stamps, 2 * k + 1)
                    return r
                return run
            fns = tuple(stamped(k, fn) for k, fn in enumerate(fns))
        return fns

    def _param_groups(self):
        if self._groups is None:
            own = {id(p) for p in self.opt.params}
            gat = [p for p in self.model.GATEncoder.parameters() if id(p) in own]
            svgp = [p for p in self.model.SVGPEncoder.parameters() if id(p) in own]
            taken = {id(p) for p in gat + svgp}
            above = {id(p) for p in self.model.GATEncoder.above_second_dense()}
            self._groups = {"svgp": svgp, "tail": [p for p in self.opt.params if id(p) not in taken],
                            "gat_above_d2": [p for p in gat if id(p) in above],
                            "gat_below_d2": [p for p in gat if id(p) not in above]}
        return self._groups

    def _issue_staged(self, fns, two_streams=True):
        """The stages in order: `fns` are the replay methods of their graphs (GAT stages on the main stream, the SVGP stages
        and the queue beside them on the side stream) or, on the eager warm-up visit, the stage callables themselves (one
        stream).  With the bucketed exchange the all-reduce of everything but the first layer's gradients is issued behind
        the queue (side stream), the first layer's behind the main stream's last graph."""
        gat_fwd, svgp_fwd, tail, svgp_bwd, gat_bwd_a, gat_bwd_b, late, svgp_pre = fns
        main = torch.cuda.current_stream()
        if not two_streams:
            gat_fwd(); svgp_fwd(); svgp_pre()
            res = tail()
            svgp_bwd(); gat_bwd_a(); gat_bwd_b(); late()
            if self.overlap:
                self._exchange_buckets(main, main)
            return res
        side = self.model._side_stream()
        # Issue order inside a pair: the GAT graph (main stream, the longer one) FIRST.  A graph launch costs the host
        # ~2.5 us per node, and the graph launched second only starts once the first has been handed over.  The side
        # stream's wait on `main` is recorded before the GAT launch, so it covers the work in front of the pair.
        if self._chain and self._head_ready and self._head_event is not None:
            side.wait_event(self._head_event)        # (chained(): the previous step's update of the SVGP encoder)
        else:
            side.wait_stream(main)
        self._head_ready = False
        gat_fwd()
        with torch.cuda.stream(side):
            svgp_fwd()
        main.wait_stream(side)                       # (an event behind the SVGP forward: what follows on `side` is not waited for)
        res = tail()
        with torch.cuda.stream(side):                # (launched behind the tail: the host hands the critical graph over first)
            svgp_pre()
        side.wait_stream(main)
        gat_bwd_a()
        if self._late_event is None:
            self._late_event = torch.cuda.Event()
        self._late_event.record(main)
        with torch.cuda.stream(side):
            svgp_bwd()
        gat_bwd_b()
        with torch.cuda.stream(side):
            # the queue reads what the tail and gat_bwd_a produced (the event) and, of gat_bwd_b, only operands that exist
            # before that stage starts (ops._DenseCD.backward checks it: its incoming gradient needs no copy)
            side.wait_event(self._late_event)
            late()
        if self.overlap:
            self._exchange_buckets(main, side)
        main.wait_stream(side)
        return res

    def _exchange_buckets(self, main, side):
        """The step's two collectives, in this order on every rank: flat_grad[:tail_offset] (everything but the first GAT
        layer: final when the side stream's queue has run) and the tail (the first layer: final when the main stream's last
        backward graph has).  The process group's stream waits for the stream each is issued on; both are waited for on
        the main stream (the update follows there)."""
        cut = self.opt.tail_offset
        with torch.cuda.stream(side):
            w1 = self.grad_sync_async(self.opt.flat_grad[:cut])
        w2 = self.grad_sync_async(self.opt.flat_grad[cut:])
        w1.wait()
        w2.wait()

    def exchange_idle(self):
        """A replica without a batch in this step (its flat gradient is zero) joins the step's two collectives."""
        main = torch.cuda.current_stream()
        self._exchange_buckets(main, main)

    def _run_staged(self, tp_i, tp, bi, epoch, with_update):
        key = (tp, bi, epoch >= 1, epoch >= self.cfg["ot_epoch"] and tp_i != 0, "staged")
        if key in self.graphs:
            graphs, out = self.graphs[key]
            self._issue_staged([g.replay for g in graphs])
            res = out.clone() if self.clone_output else out
        elif key not in self.seen:                                          # warm-up visit: eager, same stages
            self.seen.add(key)
            res = self._issue_staged(list(self._stages(tp_i, tp, bi, epoch)), two_streams=False)
        else:
            fns = self._stages(tp_i, tp, bi, epoch)
            if self.pool is None:
                self.pool, self.pool_side = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
            torch.cuda.synchronize()
            out = None
            graphs = [None] * len(fns)
            # capture order = data order (svgp_pre fills the holder the SVGP backward's capture reads)
            for k in (0, 1, 7, 2, 3, 4, 5, 6):
                g = torch.cuda.CUDAGraph()
                pool = self.pool_side if k in self.SIDE_STAGES else self.pool      # the side stream's graphs run beside the others
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    r = fns[k]()
                if k == 2:
                    out = r
                graphs[k] = g
            self.graphs[key] = (graphs, out)
            self._issue_staged([g.replay for g in graphs])
            res = out.clone() if self.clone_output else out
        if with_update:
            self.update()
        return res

    def fb(self, tp_i, tp, bi, epoch, beta1):
        """Forward + backward of one batch into the flat gradient buffer, no parameter update (data-parallel
        replicas: exchange the gradient, then update())."""
        return self._run(tp_i, tp, bi, epoch, beta1, False)

    def step(self, tp_i, tp, bi, epoch, beta1):
        if self.grad_sync is None:                  # single replica: the update follows the backward directly
            return self._run(tp_i, tp, bi, epoch, beta1, True)
        res = self.fb(tp_i, tp, bi, epoch, beta1)
        if not self.overlap:                        # (bucketed exchange: already issued behind the backward stages)
            self.grad_sync(self.opt.flat_grad)
        self.update()
        return res


def training_step(model, optimizer, model_config, dataloader_dict, tp_i, tp, bi, epoch, beta1, grad_sync=None):
    """One optimizer step (_train_utils.py:187-217): forward_backward (which overwrites the flat
    gradient, the reference's zero_grad + backward), optional gradient all-reduce (data-parallel path), clip + AdamW."""
    losses = forward_backward(model, model_config, dataloader_dict, tp_i, tp, bi, epoch, beta1,
                              optimizer=optimizer)
    if grad_sync is not None:
        grad_sync(optimizer.flat_grad)
    optimizer.step()
    return losses


def train_SpaDOT(dataloader_dict, model_config, verbose=True, epoch_seconds=None):
    """_train_utils.py:155-236.  Returns (model, loss_df) with loss_df indexed like the reference's
    (columns = epochs, rows = loss names; train.py:38 writes its transpose).
    epoch_seconds (optional list): receives the host wall time of every epoch, K-means / OT refresh included (each epoch ends
    with a device synchronisation of its own: the loss read-back and the refit).
    The steps of an epoch run inside GraphedStepper.chained(): anything a caller adds BETWEEN two steps that touches the
    SVGP encoder (weights, BatchNorm statistics) belongs outside the chain or must call stepper.barrier()."""
    import pandas as pd
    device = torch.device(model_config["device"])
    model = SpaDOT.SpaDOT(model_config, dataloader_dict).to(device)
    optimizer = FlatAdamW(model.parameters(), lr=model_config["lr"], first=model.SVGPEncoder.parameters())
    stepper = GraphedStepper(model, optimizer, model_config, dataloader_dict) if model_config.get("use_hip_graphs", True) else None
    if stepper is not None:
        stepper.clone_output = False          # the loop below adds every step's losses to `tot` right away
    beta1s = _beta_cycle_linear(model_config["maxiter"], stop=model_config["beta1"])
    tp_indexed_list = list(enumerate(model_config["timepoints"]))
    loss_dict = OrderedDict((e, OrderedDict((n, 0.0) for n in LOSS_NAMES)) for e in range(model_config["maxiter"]))
    if verbose:
        print("Training SpaDOT model...")
        # (the per-epoch K-means fit: 'device' = spadot_amd.kmeans on the GPU, its own k-means++ stream; 'sklearn' = the
        # reference's host fit.  Labels for given centres are bit-identical either way, the fitted centres are not.)
        print("K-means backend: %s" % model_config.get("kmeans_backend", "device"))
    t_start = time()
    for epoch in range(model_config["maxiter"]):
        beta1 = float(beta1s[epoch])
        model.train()
        ep_start = time()
        random.shuffle(tp_indexed_list)
        acc = {}
        # (one chain: between two steps of an epoch only the loss accumulation is enqueued -- GraphedStepper.chained)
        with (stepper.chained() if stepper is not None else contextlib.nullcontext()):
            for tp_i, tp in tp_indexed_list:
                if tp not in dataloader_dict["dataloaders"]:
                    continue
                nb = len(dataloader_dict["dataloaders"][tp])
                tot = torch.zeros(len(LOSS_NAMES), dtype=torch.float32, device=device)
                for bi in range(nb):
                    if stepper is not None:
                        tot += stepper.step(tp_i, tp, bi, epoch, beta1).float()
                    else:
                        tot += training_step(model, optimizer, model_config, dataloader_dict, tp_i, tp, bi, epoch, beta1).float()
                acc[tp] = tot / nb
        for tp, v in acc.items():                      # one device->host read per time point per epoch
            for name, val in zip(LOSS_NAMES, v.cpu().tolist()):
                loss_dict[epoch][name] += val
        if verbose and epoch % 10 == 0:
            print(f"Epoch {epoch + 1}: Training time: {int(time() - ep_start)} seconds, "
                  + ", ".join(f"{n}: {loss_dict[epoch][n]:.8f}" for n in LOSS_NAMES))
        _update_Kmeans(model, model_config, dataloader_dict)
        if (epoch + 1) % model_config["ot_config"]["ot_epochs"] == 0:
            _update_OT_matrix(model, model_config)
        if epoch_seconds is not None:
            torch.cuda.synchronize(device)
            epoch_seconds.append(time() - ep_start)
    if verbose:
        print("Training finished...")
        print("Training time: %d seconds." % int(time() - t_start))
    return model, pd.DataFrame.from_dict(loss_dict)
