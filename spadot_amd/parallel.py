"""Single-node multi-GPU path: one process per GPU, torch.distributed over RCCL (backend "nccl" on
ROCm) -- or gloo in the CPU tests.  Nothing like this exists in the reference (it is single
process / single device); the design follows SURVEY 8(e):

  * Sinkhorn: the T-1 consecutive-pair solves are independent units -> pairs are dealt round-robin to
    ranks, NO data-path collective (only the small 10 x 10 training plans are gathered).
  * Training: the model is replicated and the epoch's work units are dealt to the ranks at one of two
    granularities (model_config['shard_granularity']):
      'batch' (default)  -- the unit is one (time point, batch): batch bi of time point tp belongs to rank
          (its index in the canonical, unshuffled list of all batches) mod P, for the whole run, so a rank captures
          the hipGraphs of its own batches only and caches only their inputs.  Every rank draws batches from every
          time point, so it holds every time point's rows (cfg3: 0.3 GB in bf16, cfg5: 2 GB -- of 288 GB) and
          K-means state; the per-epoch inference + K-means refit of a time point still runs on ONE rank
          (time points round-robin) and its centres AND labels are gathered.  All P ranks have work in every global
          step whatever T is (cfg3's T = 5 keeps 8 ranks busy; cfg5's T = 10 balances), and a global step averages P
          batches that mostly belong to the SAME time point -- the ranks walk the epoch's shuffled time-point order
          together -- which is the closest synchronous analogue of the reference's time point after time point loop
          (_train_utils.py:181-217).
      'timepoint'        -- the unit is a whole time point (data, graph, SVGP constants and K-means state live on
          its owner only); a global step then averages batches of P DIFFERENT time points, ranks beyond T idle, and
          T not divisible by P leaves up to 2:1 imbalance.  Kept for data sets whose time points do not fit one GPU
          next to each other.
    Each global step every rank computes
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
import contextlib
import numpy as np
import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


class ShardPlan:
    """Who owns what.  timepoints: ordered list; pairs are (timepoints[i], timepoints[i+1]).
    granularity 'timepoint': a time point's batches all run on its owner.  granularity 'batch' (needs
    batches_per_tp = {tp: number of batches}): batch bi of time point tp runs on rank (index of (tp, bi) in the
    canonical list of all batches) mod P; `owner` then only says who refits a time point's K-means / solves a pair."""

    def __init__(self, timepoints, world_size, rank, granularity="timepoint", batches_per_tp=None):
        self.timepoints = list(timepoints)
        self.world_size, self.rank = int(world_size), int(rank)
        if granularity not in ("timepoint", "batch"):
            raise ValueError("shard granularity must be 'timepoint' or 'batch'")
        if granularity == "batch" and batches_per_tp is None:
            raise ValueError("granularity 'batch' needs the number of batches of every time point")
        self.granularity = granularity
        T = len(self.timepoints)
        self.owner = {tp: i % self.world_size for i, tp in enumerate(self.timepoints)}
        self.pair_owner = {(self.timepoints[i], self.timepoints[i + 1]): i % self.world_size for i in range(T - 1)}
        self.batches_per_tp = None if batches_per_tp is None else {tp: int(batches_per_tp[tp]) for tp in self.timepoints}
        self._base, acc = {}, 0
        for tp in self.timepoints:                # canonical index of a time point's first batch
            self._base[tp] = acc
            acc += self.batches_per_tp[tp] if self.batches_per_tp is not None else 0

    def unit_owner(self, tp, bi):
        """Rank that runs batch `bi` of time point `tp` (fixed for the whole run)."""
        if self.granularity == "timepoint":
            return self.owner[tp]
        return (self._base[tp] + int(bi)) % self.world_size

    def owned_batches(self, tp, rank=None):
        r = self.rank if rank is None else rank
        if self.batches_per_tp is None:
            raise ValueError("this plan was built without batch counts")
        return [bi for bi in range(self.batches_per_tp[tp]) if self.unit_owner(tp, bi) == r]

    def owned_timepoints(self, rank=None):
        """Time points whose per-epoch inference + K-means refit this rank runs."""
        r = self.rank if rank is None else rank
        return [tp for tp in self.timepoints if self.owner[tp] == r]

    def owned_pairs(self, rank=None):
        r = self.rank if rank is None else rank
        return [p for p, o in self.pair_owner.items() if o == r]

    def data_timepoints(self, rank=None):
        """Time points whose DATA a rank must hold: the ones it draws batches from and the ones whose K-means it
        refits.  Per-time-point plans: the owned ones (centres of predecessors arrive through the all-gather);
        per-batch plans: every time point that has a batch on this rank."""
        own = self.owned_timepoints(rank)
        if self.granularity == "timepoint":
            return own
        return [tp for tp in self.timepoints if tp in own or self.owned_batches(tp, rank)]


def plan_from_counts(timepoints, n_per_tp, batch_size, world_size, rank, granularity="batch"):
    """ShardPlan from the spot count of every time point: the loader cuts a time point into ceil(N_t / batch_size)
    consecutive seed blocks (_train_utils.py:80-85)."""
    nb = {tp: -(-int(n_per_tp[tp]) // int(batch_size)) for tp in timepoints}
    return ShardPlan(timepoints, world_size, rank, granularity=granularity, batches_per_tp=nb)


def configure_shard(adata, model_config, world_size, rank):
    """Call before prepare_dataloader on every rank: builds the plan from the data's time-point column and tells
    prepare_dataloader which time points (model_config['owned_timepoints']) and which of their batches
    (model_config['owned_batches']) this rank needs.  Returns the plan."""
    tps = np.asarray(adata.obs["timepoint"])
    counts = {tp: int(np.sum(tps == tp)) for tp in model_config["timepoints"]}
    plan = plan_from_counts(model_config["timepoints"], counts, model_config["batch_size"], world_size, rank,
                            model_config.get("shard_granularity", "batch"))
    model_config["owned_timepoints"] = plan.data_timepoints()
    if plan.granularity == "batch":
        model_config["owned_batches"] = {tp: plan.owned_batches(tp) for tp in plan.data_timepoints()}
    else:
        model_config.pop("owned_batches", None)
    return plan


def epoch_schedule(plan, batches_per_tp, order):
    """Per-rank list of (tp_i, tp, batch) work items of one epoch, and the number of global steps
    (= the longest list).  `order`: the epoch's shuffled [(tp_i, tp), ...] (same on every rank: it
    comes from Python's `random` seeded identically, _train_utils.py:181)."""
    per_rank = [[] for _ in range(plan.world_size)]
    for tp_i, tp in order:
        for bi in range(batches_per_tp[tp]):
            per_rank[plan.unit_owner(tp, bi)].append((tp_i, tp, bi))
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


def shard_range(count, rank, P):
    """[lo, hi) of rank's slice of a flat buffer of `count` elements cut into P equal slices of `per` elements (a multiple of 4:
    the update kernels work on 16-byte pieces); the last slices may be short or empty.  Returns (lo, hi, per)."""
    per = (-(-count // P) + 3) // 4 * 4
    lo = min(rank * per, count)
    return lo, min(lo + per, count), per


def sharded_update(flat_grad, flat_param, local_sumsq, update_range, group=None):
    """One optimizer step of P replicas with the flat buffers SHARDED over the ranks (opt-in,
    model_config['sharded_update']): instead of all-reducing the whole gradient (every rank receives and then updates all
    of it) the gradient is REDUCE-SCATTERED -- rank r receives only the sum of slice r -- each rank clips and updates its own
    slice, and the new parameters are ALL-GATHERED.  The same bytes cross the wire as in a ring all-reduce, but every rank
    runs the update (and keeps the two moment buffers current) for 1/P of the parameters; on the point-to-point xGMI
    topology the two halves are direct exchanges.
        local_sumsq(lo, hi)         -> 1-element tensor: squared norm of flat_grad[lo:hi] (after the reduce-scatter; zero for
                                       an empty slice, same dtype on every rank)
        update_range(lo, hi, sumsq) -> clip (by the GLOBAL squared norm `sumsq`, a 1-element tensor) + AdamW on [lo, hi)
    The clip norm is the one scalar all-reduce.  Backends without reduce_scatter_tensor / all_gather_into_tensor (gloo: the
    CPU tests and the one-GPU rehearsal) and buffers that do not divide into equal slices take all-reduce + list all-gather:
    the same values everywhere, only more traffic.  Every rank ends with identical parameters (the gathered ones)."""
    rank, P = world()
    n = flat_grad.numel()
    lo, hi, per = shard_range(n, rank, P)
    if P == 1:
        update_range(0, n, local_sumsq(0, n))
        return
    direct = dist.get_backend(group) == "nccl" and per * P == n
    if direct:
        dist.reduce_scatter_tensor(flat_grad[lo:hi], flat_grad, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    sq = local_sumsq(lo, hi)                       # (an empty slice -- more ranks than pieces -- contributes zero)
    dist.all_reduce(sq, op=dist.ReduceOp.SUM, group=group)
    if hi > lo:
        update_range(lo, hi, sq)
    if direct:
        dist.all_gather_into_tensor(flat_param, flat_param[lo:hi].clone(), group=group)
    else:
        mine = torch.zeros(per, dtype=flat_param.dtype, device=flat_param.device)
        mine[:hi - lo] = flat_param[lo:hi]
        parts = [torch.empty_like(mine) for _ in range(P)]
        dist.all_gather(parts, mine, group=group)
        for r, part in enumerate(parts):
            a, b, _ = shard_range(n, r, P)
            if b > a and r != rank:
                flat_param[a:b] = part[:b - a]


def average_buffers(module, weight=1.0):
    """BatchNorm running_mean / running_var averaged over ranks, each replica weighted by `weight` = the number of
    training steps it ran since the last call (each replica saw different batches; a rank that ran none -- more
    ranks than time points, or a short time point -- would otherwise drag the statistics back towards the last
    average).  If no rank ran a step the buffers are left alone."""
    _, P = world()
    if P == 1:
        return
    if hasattr(module, "_break_step_chains"):         # writes BatchNorm statistics between steps: end any stepper chain
        module._break_step_chains()
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


def gather_labels(local_labels, plan, n_per_tp, device):
    """local_labels: {tp: int array [N_tp]} for the time points whose K-means this rank refitted -> the same dict for
    ALL time points on every rank (one all-reduce of a T x max N zero-padded int32 tensor).  Per-batch plans need it:
    every rank trains on batches of every time point, and the K-means / OT loss terms read the seeds' labels."""
    T = len(plan.timepoints)
    nmax = max(int(n_per_tp[tp]) for tp in plan.timepoints)
    buf = torch.zeros((T, nmax), dtype=torch.int32, device=device)
    for i, tp in enumerate(plan.timepoints):
        if tp in local_labels:
            lab = torch.as_tensor(np.asarray(local_labels[tp]), dtype=torch.int32)
            buf[i, :lab.numel()] = lab.to(device)
    if plan.world_size > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    out = buf.cpu().numpy()
    return {tp: out[i, :int(n_per_tp[tp])].copy() for i, tp in enumerate(plan.timepoints)}


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


def reduce_epoch_losses(sums, counts, plan):
    """The reference's per-epoch loss record (_train_utils.py:219-224: for every time point the MEAN of its batches' loss
    vectors, the record = the SUM of those means over the time points) from per-rank partial sums.
    sums [T, n_terms], counts [T] (float tensors on the collectives' device): this rank's sum of loss vectors and number of
    batches per time point (rows of time points it drew nothing from stay zero).  One all-reduce of a T x (n_terms + 1)
    tensor; every rank gets the same record, and it is what a single process computes from all batches -- whatever P is."""
    buf = torch.cat([sums.double(), counts.double().reshape(-1, 1)], dim=1)
    if plan.world_size > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    tot, cnt = buf[:, :-1], buf[:, -1:]
    means = torch.where(cnt > 0, tot / cnt.clamp(min=1.0), torch.zeros_like(tot))
    return means.sum(dim=0)


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
    dataloader_dict built for plan.data_timepoints()).  Returns (model, loss_df) like train_SpaDOT: loss_df's column of an
    epoch is the reference's loss.csv row (sum over time points of the per-time-point batch means, _train_utils.py:219-224),
    reduced over the ranks -- the same on every rank and the same quantity a single process reports."""
    import random
    from collections import OrderedDict
    from .model import SpaDOT
    from .ops import FlatAdamW
    from .utils import _train_utils as tu
    rank, P = world()
    device = torch.device(model_config["device"])
    gran = model_config.get("shard_granularity")
    if gran is None:
        # not stated: a loader prepared through configure_shard carries 'owned_batches' (batch units); one prepared the
        # older way -- owned_timepoints only, whole time points -- keeps meaning the time-point plan
        gran = "batch" if "owned_batches" in model_config or "owned_batches" in dataloader_dict else "timepoint"
    plan = plan_from_counts(model_config["timepoints"], dataloader_dict["N_train"], model_config["batch_size"], P, rank, gran)
    for tp in model_config["timepoints"]:          # the loader this rank was given must cover its share of the plan
        mine = plan.owned_batches(tp)
        have = dataloader_dict["dataloaders"].get(tp, [])
        if mine and (len(have) != plan.batches_per_tp[tp] or any(have[bi] is None for bi in mine)):
            raise ValueError(f"rank {rank}: the dataloader lacks batches of time point {tp} that the shard plan gives it "
                             f"(granularity '{gran}': call parallel.configure_shard before prepare_dataloader, or set "
                             "shard_granularity='timepoint' for a loader that holds whole time points)")
    model = SpaDOT.SpaDOT(model_config, dataloader_dict).to(device)
    # identical replicas: broadcast rank 0's initial parameters and buffers
    if P > 1:
        for t in list(model.parameters()) + [b for b in model.buffers()]:
            dist.broadcast(t.data, src=0)
    opt = FlatAdamW(model.parameters(), lr=model_config["lr"], last=model.GATEncoder.first_layer_parameters(),
                    first=model.SVGPEncoder.parameters())
    sync, sync_async = make_grad_sync(opt)
    beta1s = tu._beta_cycle_linear(model_config["maxiter"], stop=model_config["beta1"])
    order = list(enumerate(model_config["timepoints"]))
    batches_per_tp = plan.batches_per_tp           # (from the spot counts: the same on every rank, no collective)
    refit = list(plan.owned_timepoints())
    missing = [tp for tp in refit if tp not in dataloader_dict["datasets"]]
    if missing:         # (its centres and labels would be gathered as all-zero from this rank, silently)
        raise ValueError(f"rank {rank}: time points {missing} are this rank's to refit but their rows are not in "
                         "dataloader_dict['datasets']")
    loss_dict = OrderedDict()
    tp_row = {tp: i for i, tp in enumerate(model_config["timepoints"])}
    n_terms = len(tu.LOSS_NAMES)
    # replayed hipGraphs, as in the single-replica trainer: one forward+backward graph per (time point, batch),
    # the all-reduce of the flat gradient between replays (not captured), one clip + AdamW graph
    sharded = bool(model_config.get("sharded_update", False)) and P > 1
    stepper = (tu.GraphedStepper(model, opt, model_config, dataloader_dict, grad_sync=sync,
                                 grad_sync_async=sync_async if (P > 1 and not sharded) else None)
               if model_config.get("use_hip_graphs", True) else None)

    def exchange(did_work):
        if sharded:                                        # the exchange IS the update (reduce-scatter .. all-gather)
            return
        if stepper is not None and stepper.overlap:        # bucketed: a rank with a batch has exchanged inside fb()
            if not did_work:
                stepper.exchange_idle()
        else:
            sync(opt.flat_grad)

    def apply_update():
        if sharded:            # (eager launches between the replayed steps: the step's chain to the update's first part is not used)
            opt.step_sharded()
        elif stepper is not None:
            stepper.update()
        else:
            opt.step()

    for epoch in range(model_config["maxiter"]):
        beta1 = float(beta1s[epoch])
        model.train()
        random.shuffle(order)
        sums = torch.zeros((len(tp_row), n_terms), dtype=torch.float32, device=device)
        counts = torch.zeros(len(tp_row), dtype=torch.float32, device=device)

        def compute_grad(tp_i, tp, bi):
            if stepper is not None:
                l = stepper.fb(tp_i, tp, bi, epoch, beta1)
            else:
                l = tu.forward_backward(model, model_config, dataloader_dict, tp_i, tp, bi, epoch, beta1, optimizer=opt)
            sums[tp_row[tp]] += l.float()           # (in-stream: the stepper's own loss buffer may be rewritten by the next replay)
            counts[tp_row[tp]] += 1.0

        with (stepper.chained() if stepper is not None else contextlib.nullcontext()):
            _, n_mine = run_epoch(plan, batches_per_tp, order, compute_grad, opt.zero_grad, opt.flat_grad,
                                  apply_update, exchange=exchange, set_grad_scale=lambda x: opt.grad_scale.fill_(x))
        rec = reduce_epoch_losses(sums, counts, plan).cpu().tolist()
        loss_dict[epoch] = OrderedDict(zip(tu.LOSS_NAMES, rec))
        average_buffers(model, weight=n_mine)
        # inference + K-means refit of a time point on ONE rank; centres (and, where other ranks train on that time
        # point's batches, labels) to everyone
        tu._update_Kmeans(model, model_config, dataloader_dict, timepoints=refit)
        centres = gather_centres({tp: model.kmeans_center_dict[tp] for tp in refit},
                                 plan, model_config["n_clusters"], model_config["z_dim"], device)
        labels = None
        if plan.granularity == "batch" and P > 1:
            labels = gather_labels({tp: model.kmeans_cluster_dict[tp] for tp in refit}, plan,
                                   dataloader_dict["N_train"], device)
        for tp, c in centres.items():
            if tp in refit:
                continue
            if tp in dataloader_dict["datasets"] and labels is not None:
                tu._set_kmeans_state(model, tp, c, labels[tp], dataloader_dict["datasets"][tp][2], device)
            else:
                tu._set_kmeans_state(model, tp, c, np.arange(c.shape[0]), np.arange(c.shape[0]), device)
        if (epoch + 1) % model_config["ot_config"]["ot_epochs"] == 0:
            # pair solves are sharded; every rank ends up with every (tiny) plan
            local = {}
            for (a, b) in plan.owned_pairs():
                from .utils.OT_loss.ot_solvers import compute_transport_map
                local[(a, b)] = compute_transport_map(centres[a], centres[b], model_config["ot_config"], G=None,
                                                      device=device)
            K = model_config["n_clusters"]
            for (a, b), g in gather_small_plans(local, plan, (K, K), device).items():
                tu._set_gamma(model, f"{a}_{b}", g, device)
    import pandas as pd
    return model, pd.DataFrame.from_dict(loss_dict)
