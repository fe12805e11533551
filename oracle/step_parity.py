"""oracle/step_parity.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

One training step of the reference's arithmetic (_train_utils.py:187-217 -> SpaDOT.py:52-94) on the host in fp64,
on exactly the inputs a device step used -- same weights, batch, induced graph, reparameterisation noise, K-means
state and OT plan -- and the comparison of the two.  It only READS the product's objects (state_dict, batch tensors,
the reference-named kmeans_* / gammas attributes); the arithmetic is oracle/model_oracle.py's.

Used by tests/test_step_parity_gpu.py and by bench.py next to the cpu_baseline leg (the oracle step it times there
is the same call, so the check costs nothing extra).  Only tests/, __graft_entry__.smoke() and bench.py may import it.
"""
import time

import numpy as np
import torch

from . import model_oracle as mo

LOSS_NAMES = ["elbo", "Recon", "SVGP_KL", "GAT_KL", "alignment", "KMeans", "OT"]


def make_noise(b, L=10, seed=0):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn((b, L), dtype=torch.float64, generator=gen),
            torch.randn((b, L), dtype=torch.float64, generator=gen))


def oracle_inputs(model, dd, cfg, tp, bi, tp_prev, do_km=True, do_ot=True):
    """Host fp64 copies of everything the step of batch `bi` of time point `tp` reads."""
    batch = dd["dataloaders"][tp][bi]
    loc, Y, _ = dd["datasets"][tp]
    n_id = batch.n_id
    x = loc[n_id].cpu().double()
    y = Y[n_id].float().cpu().double()          # the expression matrix as stored (bf16 storage is exact in fp64)
    g = batch.graph
    tgt = torch.repeat_interleave(torch.arange(g.n), (g.rowptr[1:] - g.rowptr[:-1]).cpu().long())
    ei = torch.stack([g.col.cpu().long(), tgt])
    P = {k: v.detach().cpu().double().clone() for k, v in model.state_dict().items()}
    for k, _ in model.named_parameters():          # (buffers -- running statistics, batch counters -- stay constants)
        P[k].requires_grad_(True)
    sv = mo.SVGPOracle(dd["inducing_points"][tp], dd["N_train"][tp], kernel_type=cfg.get("kernel_type", "Gaussian"),
                       scale=cfg.get("kernel_scale", 0.1))
    b = batch.batch_size
    seeds = n_id[:b].cpu().numpy()
    km = ot = None
    if do_km or do_ot:
        labels = np.asarray(model.kmeans_cluster_dict[tp])
        if do_km:
            km = (model.kmeans_center_dict[tp], labels[seeds])
        if do_ot:
            ot = (labels[seeds], labels, model.kmeans_center_dict[tp], model.kmeans_center_dict[tp_prev],
                  model.gammas[f"{tp_prev}_{tp}"])
    return dict(P=P, svgp=sv, x=x, y=y, ei=ei, b=b, km=km, ot=ot, n_sub=g.n, E=g.E, m=int(sv.z.shape[0]))


def oracle_step(inp, cfg, beta1, noise, n_steps=1):
    """`n_steps` optimizer steps on the same batch (the first from the given weights; its loss terms, latent and
    pre-clip gradients are returned, the later ones only add timing samples).  Returns dict(losses [7], latent,
    grads {name: ndarray}, seconds [n_steps])."""
    w = (cfg["lambda1"], float(beta1), cfg["beta2"], cfg["omiga1"], cfg["omiga2"], cfg["omiga3"])
    detail, secs, opt, first = {}, [], None, None
    for k in range(n_steps):
        t0 = time.perf_counter()
        terms, opt = mo.training_step(inp["P"], inp["svgp"], inp["x"], inp["y"], inp["ei"], inp["b"],
                                      cfg["gat_attention_heads"], noise[0], noise[1], w, km=inp["km"], ot=inp["ot"],
                                      lr=cfg["lr"], opt_state=opt, detail=detail if k == 0 else None)
        secs.append(time.perf_counter() - t0)
        if k == 0:
            first = terms
    return dict(losses=np.array([first[n] for n in LOSS_NAMES]), latent=detail["latent"].numpy(),
                grads={k: v.numpy() for k, v in detail["grads"].items()}, seconds=secs)


def oracle_step_without_svgp_tape(inp, cfg, beta1, noise):
    """oracle_step's loss terms and latent, and the gradients of the GAT encoder's and the decoder's parameters ONLY: the SVGP
    branch is evaluated without a tape (model_oracle.spadot_forward svgp_no_grad), so the ten (b, m, m) ELBO tensors are never
    alive together -- affordable at cfg2's m ~ 600.  Those gradients are exact: no path from these parameters to the loss
    runs through the SVGP branch.  Returns dict(losses [7], latent, grads {name: ndarray}, seconds)."""
    w = (cfg["lambda1"], float(beta1), cfg["beta2"], cfg["omiga1"], cfg["omiga2"], cfg["omiga3"])
    t0 = time.perf_counter()
    for v in inp["P"].values():
        v.grad = None
    loss, terms, z = mo.step_loss(inp["P"], inp["svgp"], inp["x"], inp["y"], inp["ei"], inp["b"], cfg["gat_attention_heads"],
                                  noise[0], noise[1], w, km=inp["km"], ot=inp["ot"], svgp_no_grad=True)
    loss.backward()
    grads = {k: v.grad.detach().numpy().copy() for k, v in inp["P"].items() if v.requires_grad and v.grad is not None}
    assert grads and not any(k.startswith("SVGPEncoder.") for k in grads)
    return dict(losses=np.array([float(terms[n].detach()) for n in LOSS_NAMES]), latent=z.detach().numpy(), grads=grads,
                seconds=[time.perf_counter() - t0])


def compare(dev_losses, dev_latent, dev_grads, ref):
    """Agreement of a device step with the oracle's.  dev_losses [7], dev_latent [b, z], dev_grads {name: array}.
    Per parameter: relative L2 error ||g_dev - g_ref|| / ||g_ref|| and cosine; parameters whose reference gradient
    is numerically zero (a Linear bias in front of a BatchNorm: the batch mean removes it) are reported apart and
    must be (near) zero on the device too.  Returns a dict of plain floats / short lists (JSON-ready)."""
    dl, rl = np.asarray(dev_losses, dtype=np.float64), ref["losses"]
    rel_loss = np.abs(dl - rl) / np.maximum(np.abs(rl), 1e-12)
    dz, rz = np.asarray(dev_latent, dtype=np.float64), ref["latent"]
    per = {}
    tot_dd = tot_rr = tot_dr = 0.0
    gmax = max(float(np.linalg.norm(v)) for v in ref["grads"].values())
    zero_params, zero_dev_max = [], 0.0
    for name, r in ref["grads"].items():
        d = np.asarray(dev_grads[name], dtype=np.float64).reshape(r.shape)
        nr, nd = float(np.linalg.norm(r)), float(np.linalg.norm(d))
        tot_dd += nd * nd; tot_rr += nr * nr; tot_dr += float((d * r).sum())
        if nr <= 1e-10 * gmax:
            zero_params.append(name)
            zero_dev_max = max(zero_dev_max, nd / gmax)
            continue
        per[name] = (float(np.linalg.norm(d - r)) / nr, float((d * r).sum()) / max(nr * nd, 1e-300))
    worst_l2 = max(per, key=lambda k: per[k][0])
    worst_cos = min(per, key=lambda k: per[k][1])
    return {
        "loss_names": LOSS_NAMES,
        "loss_dev": dl.tolist(), "loss_ref": rl.tolist(), "loss_rel_err": rel_loss.tolist(),
        "max_rel_loss_err": float(rel_loss.max()),
        "latent_max_abs_err": float(np.abs(dz - rz).max()),
        "latent_rel_l2_err": float(np.linalg.norm(dz - rz) / np.linalg.norm(rz)),
        "grad_cos_min": per[worst_cos][1], "grad_cos_min_param": worst_cos,
        "grad_rel_l2_max": per[worst_l2][0], "grad_rel_l2_max_param": worst_l2,
        "grad_cos_global": tot_dr / max(np.sqrt(tot_dd * tot_rr), 1e-300),
        "grad_norm_dev": float(np.sqrt(tot_dd)), "grad_norm_ref": float(np.sqrt(tot_rr)),
        "zero_grad_params": zero_params, "zero_grad_dev_rel_norm_max": zero_dev_max,
        "per_param": {k: [round(v[0], 6), round(v[1], 8)] for k, v in per.items()},
    }


def device_step(model, opt, cfg, dd, tu, tp_i, tp, bi, epoch, beta1, noise):
    """The product's eager step body on the same batch with the same noise (no parameter update): returns
    (losses [7] ndarray, latent ndarray, {state_dict name: gradient ndarray}).  `tu` = spadot_amd.utils._train_utils."""
    dev = next(model.parameters()).device
    lat = []
    was_training = model.training
    model.train()
    losses = tu.forward_backward(model, cfg, dd, tp_i, tp, bi, epoch, beta1, optimizer=opt,
                                 noise=(noise[0].to(dev), noise[1].float().to(dev)), latent_out=lat)
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().float().cpu().numpy().copy() for n, p in model.named_parameters()}
    model.train(was_training)
    return losses.detach().float().cpu().numpy().astype(np.float64), lat[0].float().cpu().numpy(), grads
