"""Single-node multi-GPU path: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) -- or gloo in the CPU tests.  Nothing like this exists in the reference (it is single
process / single device); the design follows SURVEY 8(e):

  * Sinkhorn: the T-1 consecutive-pair solves are independent units -> pairs are dealt round-robin to
    ranks, NO data-path collective (only the small 10 x 10 training plans are gathered).
  * Training: time points are dealt round-robin to ranks (data, graph, SVGP constants and K-means state
    of a time point live on its owner); the model is replicated; each global step every rank computes
    the gradient of ITS next batch into the flat gradient buffer, ONE all-reduce (sum) of that buffer
    crosses xGMI, and every replica applies the same clip + AdamW update to the MEAN over the replicas that
    had a batch in that step (the 1/n sits in the update kernel's `grad_scale`).  Ranks that have run out of
    batches in an epoch contribute zeros.  Per epoch: BatchNorm running statistics are averaged, weighted by
    the number of steps each replica ran, and the
    K-means centres (T x 10 x 20 floats) are all-gathered (the OT term of time point t needs the
    centres of t-1, which may live on another rank).
    This turns the reference's sequential one-step-per-batch schedule into synchronous steps with P
    batches per update: 1-GPU runs are the parity-checked ones (SURVEY 8e caveat).
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


class ShardPlan:
    """Who owns what.  timepoints: ordered list; pairs are (timepoints[i], timepoints[i+1])."""

    def __init__(self, timepoints, world_size, rank):
        self.timepoints = list(timepoints)
        self.world_size, self.rank = int(world_size), int(rank)
        T = len(self.timepoints)
        self.owner = {tp: i % self.world_size for i, tp in enumerate(self.timepoints)}
        self.pair_owner = {(self.timepoints[i], self.timepoints[i + 1]): i % self.world_size for i in range(T - 1)}

    def owned_timepoints(self, rank=None):
        r = self.rank if rank is None else rank
        return [tp for tp in self.timepoints if self.owner[tp] == r]

    def owned_pairs(self, rank=None):
        r = self.rank if rank is None else rank
        return [p for p, o in self.pair_owner.items() if o == r]

    def data_timepoints(self, rank=None):
        """Time points whose DATA a rank must hold: the ones it trains on (their K-means state is
        computed locally); centres of predecessors arrive through the all-gather."""
        return self.owned_timepoints(rank)


def epoch_schedule(plan, batches_per_tp, order):
    """Per-rank list of (tp_i, tp, batch) work items of one epoch, and the number of global steps
    (= the longest list).  `order`: the epoch's shuffled [(tp_i, tp), ...] (same on every rank: it
    comes from Python's `random` seeded identically, _train_utils.py:181)."""
    per_rank = [[] for _ in range(plan.world_size)]
    for tp_i, tp in order:
        r = plan.owner[tp]
        per_rank[r].extend((tp_i, tp, bi) for bi in range(batches_per_tp[tp]))
    return per_rank, max(len(x) for x in per_rank)


def allreduce_flat_grad(flat_grad):
    """The step's one exchange: sum of the flat gradient buffer over all ranks (in place)."""
    if world()[1] > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def make_grad_sync(opt):
    """(blocking, async) exchange of an ops.FlatAdamW's flat gradient.  When the optimizer keeps a tail group
    (`last=`: the first GAT layer's parameters) the buffer travels as TWO buckets, [0, tail_offset) and the tail, in
    that order on every rank and every path -- GraphedStepper issues them beside the end of the backward pass,
    everything else one after the other."""
    cut = opt.tail_offset

    def sync(flat):
        if world()[1] > 1:
            if cut:
                dist.all_reduce(flat[:cut], op=dist.ReduceOp.SUM)
                dist.all_reduce(flat[cut:], op=dist.ReduceOp.SUM)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def sync_async(view):
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True)

    return sync, sync_async


def average_buffers(module, weight=1.0):
    """BatchNorm running_mean / running_var averaged over ranks, each replica weighted by `weight` = the number of
    training steps it ran since the last call (each replica saw different batches; a rank that ran none -- more
    ranks than time points, or a short time point -- would otherwise drag the statistics back towards the last
    average).  If no rank ran a step the buffers are left alone."""
    _, P = world()
    if P == 1:
        return
    bufs = [b for n, b in module.named_buffers() if b.is_floating_point()]
    if not bufs:
        return
    w = float(weight)
    flat = torch.cat([b.reshape(-1).float() for b in bufs] + [torch.ones(1, device=bufs[0].device)]) * w
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    total = float(flat[-1])
    if total <= 0.0:
        return
    flat = flat[:-1] / total
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b).to(b.dtype))
        off += n


def gather_centres(local_centres, plan, n_clusters, z_dim, device):
    """local_centres: {tp: ndarray [n_clusters, z_dim]} for owned time points -> the same dict for ALL
    time points on every rank (one all-reduce of a T x K x D zero-padded tensor)."""
    T = len(plan.timepoints)
    buf = torch.zeros((T, n_clusters, z_dim), dtype=torch.float64, device=device)
    for i, tp in enumerate(plan.timepoints):
        if tp in local_centres:
            buf[i] = torch.as_tensor(np.asarray(local_centres[tp]), dtype=torch.float64)
    if plan.world_size > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    return {tp: out[i] for i, tp in enumerate(plan.timepoints)}


def gather_small_plans(local_plans, plan, shape, device):
    """{(tp, next): ndarray shape} from the pair owners -> every rank (training-size plans only)."""
    pairs = list(plan.pair_owner)
    buf = torch.zeros((len(pairs),) + tuple(shape), dtype=torch.float64, device=device)
    for i, p in enumerate(pairs):
        if p in local_plans:
            buf[i] = torch.as_tensor(np.asarray(local_plans[p]), dtype=torch.float64)
    if plan.world_size > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    return {p: out[i] for i, p in enumerate(pairs)}


def run_epoch(plan, batches_per_tp, order, compute_grad, zero_grad, flat_grad, apply_update, exchange=None,
              set_grad_scale=None):
    """One synchronous data-parallel epoch.
        compute_grad(tp_i, tp, bi): backward of this rank's batch into flat_grad (after zero_grad())
        exchange(did_work):         the step's gradient exchange (default: one all-reduce of flat_grad); told whether
                                    this rank had a batch, because a stepper with the bucketed exchange has already
                                    issued its collectives inside compute_grad
        apply_update():             clip + AdamW on the (now global) flat_grad
        set_grad_scale(x):          told 1 / (number of ranks that had a batch in this step) before apply_update: the
                                    exchange SUMS, the update is made with the MEAN over the contributing replicas, so
                                    the fixed clip threshold (0.3) and the step size mean what they mean on one GPU
                                    whatever the number of ranks (known from the schedule: no extra collective)
    Every rank calls the collectives the same number of times (n_steps), whatever its own item count.
    Returns (n_steps, number of steps this rank computed)."""
    per_rank, n_steps = epoch_schedule(plan, batches_per_tp, order)
    mine = per_rank[plan.rank]
    last_scale = None
    for s in range(n_steps):
        zero_grad()
        did = s < len(mine)
        if did:
            compute_grad(*mine[s])
        if exchange is not None:
            exchange(did)
        else:
            allreduce_flat_grad(flat_grad)
        if set_grad_scale is not None:
            scale = 1.0 / max(1, sum(1 for items in per_rank if s < len(items)))
            if scale != last_scale:
                set_grad_scale(scale)
                last_scale = scale
        apply_update()
    return n_steps, len(mine)


# ------------------------------------------------------------------------------ training driver

def train_SpaDOT_parallel(dataloader_dict, model_config, verbose=False):
    """Data-parallel counterpart of _train_utils.train_SpaDOT (must be called by every rank with
    dataloader_dict built for plan.data_timepoints()).  Returns (model, per-epoch loss dict of the
    LOCAL batches)."""
    import random
    from .model import SpaDOT
    from .ops import FlatAdamW
    from .utils import _train_utils as tu
    rank, P = world()
    device = torch.device(model_config["device"])
    plan = ShardPlan(model_config["timepoints"], P, rank)
    model = SpaDOT.SpaDOT(model_config, dataloader_dict).to(device)
    # identical replicas: broadcast rank 0's initial parameters and buffers
    if P > 1:
        for t in list(model.parameters()) + [b for b in model.buffers()]:
            dist.broadcast(t.data, src=0)
    opt = FlatAdamW(model.parameters(), lr=model_config["lr"], last=model.GATEncoder.first_layer_parameters())
    sync, sync_async = make_grad_sync(opt)
    beta1s = tu._beta_cycle_linear(model_config["maxiter"], stop=model_config["beta1"])
    order = list(enumerate(model_config["timepoints"]))
    batches_per_tp = {tp: 0 for tp in model_config["timepoints"]}
    for tp, bl in dataloader_dict["dataloaders"].items():
        batches_per_tp[tp] = len(bl)
    if P > 1:   # every rank needs every time point's batch count to agree on the schedule
        cnt = torch.tensor([batches_per_tp[tp] for tp in model_config["timepoints"]], device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        batches_per_tp = dict(zip(model_config["timepoints"], cnt.cpu().tolist()))
    losses = {}
    # replayed hipGraphs, as in the single-replica trainer: one forward+backward graph per (time point, batch),
    # the all-reduce of the flat gradient between replays (not captured), one clip + AdamW graph
    stepper = (tu.GraphedStepper(model, opt, model_config, dataloader_dict, grad_sync=sync,
                                 grad_sync_async=sync_async if P > 1 else None)
               if model_config.get("use_hip_graphs", True) else None)

    def exchange(did_work):
        if stepper is not None and stepper.overlap:        # bucketed: a rank with a batch has exchanged inside fb()
            if not did_work:
                stepper.exchange_idle()
        else:
            sync(opt.flat_grad)

    for epoch in range(model_config["maxiter"]):
        beta1 = float(beta1s[epoch])
        model.train()
        random.shuffle(order)
        acc = []

        def compute_grad(tp_i, tp, bi):
            if stepper is not None:
                acc.append(stepper.fb(tp_i, tp, bi, epoch, beta1))
            else:
                acc.append(tu.forward_backward(model, model_config, dataloader_dict, tp_i, tp, bi, epoch, beta1,
                                               optimizer=opt))

        _, n_mine = run_epoch(plan, batches_per_tp, order, compute_grad, opt.zero_grad, opt.flat_grad,
                              stepper.update if stepper is not None else opt.step, exchange=exchange,
                              set_grad_scale=lambda x: opt.grad_scale.fill_(x))
        losses[epoch] = torch.stack(acc).mean(0).cpu().tolist() if acc else None
        average_buffers(model, weight=n_mine)
        tu._update_Kmeans(model, model_config, dataloader_dict)
        centres = gather_centres({tp: model.kmeans_center_dict[tp] for tp in dataloader_dict["datasets"]},
                                 plan, model_config["n_clusters"], model_config["z_dim"], device)
        for tp, c in centres.items():
            if tp not in dataloader_dict["datasets"]:
                tu._set_kmeans_state(model, tp, c, np.arange(c.shape[0]), np.arange(c.shape[0]), device)
        if (epoch + 1) % model_config["ot_config"]["ot_epochs"] == 0:
            # pair solves are sharded; every rank ends up with every (tiny) plan
            local = {}
            for (a, b) in plan.owned_pairs():
                from .utils.OT_loss.ot_solvers import compute_transport_map
                local[(a, b)] = compute_transport_map(centres[a], centres[b], model_config["ot_config"], G=None)
            K = model_config["n_clusters"]
            for (a, b), g in gather_small_plans(local, plan, (K, K), device).items():
                tu._set_gamma(model, f"{a}_{b}", g, device)
    return model, losses
