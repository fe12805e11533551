"""CPU, world_size 2 over gloo: the multi-GPU path's sharding and collectives (spadot_amd.parallel).
The HIP model cannot run here, so a small torch model stands in for `compute_grad` and the oracle
stands in for the pair solver; what is under test is the schedule, the flat-gradient all-reduce, the
buffer averaging, and the centre / plan gathers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spadot_amd import parallel as par


def test_shard_plan_round_robin():
    plan = par.ShardPlan([0, 1, 2, 3, 4], 2, 0)
    assert plan.owned_timepoints(0) == [0, 2, 4] and plan.owned_timepoints(1) == [1, 3]
    assert plan.owned_pairs(0) == [(0, 1), (2, 3)] and plan.owned_pairs(1) == [(1, 2), (3, 4)]
    # more ranks than time points: the surplus ranks own nothing and only take part in collectives
    plan8 = par.ShardPlan([0, 1, 2, 3, 4], 8, 6)
    assert plan8.owned_timepoints() == [] and plan8.owned_pairs() == []
    assert sorted(sum((plan8.owned_timepoints(r) for r in range(8)), [])) == [0, 1, 2, 3, 4]
    per_rank, n = par.epoch_schedule(plan, {0: 3, 1: 2, 2: 1, 3: 4, 4: 2}, [(2, 2), (0, 0), (1, 1), (4, 4), (3, 3)])
    assert per_rank[0] == [(2, 2, 0), (0, 0, 0), (0, 0, 1), (0, 0, 2), (4, 4, 0), (4, 4, 1)]
    assert per_rank[1] == [(1, 1, 0), (1, 1, 1), (3, 3, 0), (3, 3, 1), (3, 3, 2), (3, 3, 3)]
    assert n == 6


def test_batch_granular_plan_keeps_every_rank_busy():
    """cfg3 on 8 GPUs (T = 5 < P), cfg5 (T = 10, P = 8) and cfg4 (T = 8, P = 4): (time point, batch) units dealt to
    ranks for the whole run."""
    nb3 = {t: 20 for t in range(5)}                               # cfg3: 10 000 spots / 512 seeds per time point
    plans = [par.ShardPlan(list(range(5)), 8, r, granularity="batch", batches_per_tp=nb3) for r in range(8)]
    order = [(3, 3), (0, 0), (4, 4), (1, 1), (2, 2)]               # an epoch's shuffled time-point order
    per_rank, n = par.epoch_schedule(plans[0], nb3, order)
    assert sorted(len(x) for x in per_rank) == [12, 12, 12, 12, 13, 13, 13, 13] and n == 13
    assert sorted(sum(per_rank, [])) == sorted((t, t, b) for t in range(5) for b in range(20))   # every unit exactly once
    for r in range(8):
        assert plans[r].data_timepoints() == [0, 1, 2, 3, 4]      # every rank draws from every time point
        assert [u for u in per_rank[r]] == [(ti, t, b) for ti, t in order for b in plans[r].owned_batches(t)]
        # ownership is fixed for the run: another epoch order gives the same units per rank, in another order
        other, _ = par.epoch_schedule(plans[r], nb3, order[::-1])
        assert sorted(other[r]) == sorted(per_rank[r])
    # all ranks have a batch in every one of the first 12 global steps, and they walk the time points together:
    # the batches of one global step belong to at most two (adjacent in the epoch's order) time points
    for s_ in range(12):
        tps = {per_rank[r][s_][1] for r in range(8)}
        assert len(tps) <= 2
    # K-means refits and pair solves stay dealt by time point / pair
    assert [plans[0].owner[t] for t in range(5)] == [0, 1, 2, 3, 4]
    assert plans[3].owned_pairs() == [(3, 4)] and plans[5].owned_pairs() == [] and plans[5].owned_timepoints() == []
    # cfg5: T = 10 time points of 20 000 spots on 8 ranks -- 400 units, 50 each (time-point sharding: 80 vs 40)
    nb5 = {t: 40 for t in range(10)}
    p5 = par.ShardPlan(list(range(10)), 8, 0, granularity="batch", batches_per_tp=nb5)
    per5, n5 = par.epoch_schedule(p5, nb5, [(t, t) for t in range(10)])
    assert [len(x) for x in per5] == [50] * 8 and n5 == 50
    ptp = par.ShardPlan(list(range(10)), 8, 0)
    pertp, ntp = par.epoch_schedule(ptp, nb5, [(t, t) for t in range(10)])
    assert sorted(len(x) for x in pertp) == [40] * 6 + [80, 80] and ntp == 80
    # cfg4: T = 8 over 4 ranks with ragged batch counts
    nb4 = {t: c for t, c in enumerate([3, 5, 4, 7, 6, 2, 4, 5])}
    for gran in ("timepoint", "batch"):
        p4 = [par.ShardPlan(list(range(8)), 4, r, granularity=gran, batches_per_tp=nb4) for r in range(4)]
        assert [p4[r].owned_timepoints() for r in range(4)] == [[0, 4], [1, 5], [2, 6], [3, 7]]
        assert [p4[r].owned_pairs() for r in range(4)] == [[(0, 1), (4, 5)], [(1, 2), (5, 6)], [(2, 3), (6, 7)], [(3, 4)]]
        per4, n4 = par.epoch_schedule(p4[0], nb4, [(t, t) for t in range(8)])
        assert sum(len(x) for x in per4) == 36
        if gran == "timepoint":
            assert [len(x) for x in per4] == [9, 7, 8, 12] and n4 == 12
            assert p4[1].data_timepoints() == [1, 5]
        else:
            assert [len(x) for x in per4] == [9, 9, 9, 9] and n4 == 9
            assert p4[1].data_timepoints() == list(range(8))
    with pytest.raises(ValueError):
        par.ShardPlan([0, 1], 2, 0, granularity="batch")
    cfgd = {"timepoints": [0, 1, 2], "batch_size": 4, "shard_granularity": "batch"}

    class _D:
        obs = {"timepoint": np.array([0] * 9 + [1] * 4 + [2] * 6)}
    pl = par.configure_shard(_D, cfgd, 2, 1)
    assert pl.batches_per_tp == {0: 3, 1: 1, 2: 2}
    assert cfgd["owned_batches"] == {0: [1], 1: [0], 2: [1]} and cfgd["owned_timepoints"] == [0, 1, 2]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_model(seed):
    torch.manual_seed(seed)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 1)).double()
    flat = torch.zeros(sum(p.numel() for p in m.parameters()), dtype=torch.float64)
    grad = torch.zeros_like(flat)
    off = 0
    for p in m.parameters():
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1)); p.data = flat[off:off + n].view_as(p.data)
        p.grad = grad[off:off + n].view_as(p.data)
        off += n
    return m, flat, grad


def _data(tp, bi):
    g = torch.Generator().manual_seed(100 * tp + bi)
    return torch.randn((8, 6), generator=g, dtype=torch.float64), torch.randn((8, 1), generator=g, dtype=torch.float64)


BATCHES = {0: 3, 1: 1, 2: 2}
ORDER = [(1, 1), (0, 0), (2, 2)]
BATCHES5 = {0: 3, 1: 1, 2: 2, 3: 4, 4: 2}
ORDER5 = [(3, 3), (1, 1), (0, 0), (4, 4), (2, 2)]


def _worker(rank, world, port, q, tps=(0, 1, 2), batches=None, order=None, granularity="timepoint"):
    batches = BATCHES if batches is None else batches
    order = ORDER if order is None else order
    tps = list(tps)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        plan = par.ShardPlan(tps, world, rank, granularity=granularity, batches_per_tp=batches)
        m, flat, grad = _make_model(0 if rank == 0 else 5)     # different init: broadcast must fix it
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=0)
        opt = torch.optim.SGD(m.parameters(), lr=0.1)

        def compute_grad(tp_i, tp, bi):
            x, y = _data(tp, bi)
            ((m(x) - y) ** 2).mean().backward()

        scale = [1.0]              # stands in for FlatAdamW.grad_scale (a device scalar the update kernel reads)

        def apply_update():
            grad.mul_(scale[0])
            opt.step()

        steps, n_mine = par.run_epoch(plan, batches, order, compute_grad, grad.zero_, grad, apply_update,
                                      set_grad_scale=lambda x: scale.__setitem__(0, x))
        own_bufs = [b.clone().numpy() for b in m.buffers() if b.is_floating_point()]
        par.average_buffers(m, weight=n_mine)
        centres = par.gather_centres({tp: np.full((4, 3), float(tp + 1)) for tp in plan.owned_timepoints()}, plan, 4, 3, "cpu")
        n_per_tp = {tp: 5 + 2 * tp for tp in tps}
        labels = par.gather_labels({tp: (np.arange(n_per_tp[tp]) + tp) % 4 for tp in plan.owned_timepoints()}, plan,
                                   n_per_tp, "cpu")
        # pair-sharded Sinkhorn: each rank solves its own pairs with no collective, plans gathered after
        from oracle import ot_oracle
        cfg = dict(lambda1=0.1, lambda2=5.0, epsilon=0.05, epsilon0=1.0, tolerance=1e-8, tau=1000.0,
                   batch_size=5, max_iter=10 ** 7, growth_iters=3)
        rng = np.random.default_rng(0)
        cen = {tp: rng.normal(size=(4, 3)) for tp in tps}
        local = {p: ot_oracle.compute_transport_map(cen[p[0]], cen[p[1]], cfg, all_growth_iters=False)
                 for p in plan.owned_pairs()}
        plans = par.gather_small_plans(local, plan, (4, 4), "cpu")
        q.put((rank, steps, flat.clone().numpy(), [b.clone().numpy() for b in m.buffers() if b.is_floating_point()],
               {k: v.copy() for k, v in centres.items()}, {k: v.copy() for k, v in plans.items()},
               sorted(local), own_bufs, n_mine, {k: v.copy() for k, v in labels.items()}))
    finally:
        dist.destroy_process_group()


def test_data_parallel_epoch_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, f0, b0, c0, p0, l0, ob0, n0, lab0), (r1, s1, f1, b1, c1, p1, l1, ob1, n1, lab1) = res
    for tp in (0, 1, 2):
        np.testing.assert_array_equal(lab0[tp], (np.arange(5 + 2 * tp) + tp) % 4)
        np.testing.assert_array_equal(lab1[tp], lab0[tp])
    # rank 0 owns time points 0 and 2 (3 + 2 batches), rank 1 owns 1 (1 batch): 5 global steps on both
    assert s0 == s1 == 5
    np.testing.assert_array_equal(f0, f1)                       # replicas stay identical
    assert (n0, n1) == (5, 1)
    for x, y, o0, o1 in zip(b0, b1, ob0, ob1):
        np.testing.assert_array_equal(x, y)                      # averaged BatchNorm statistics ...
        np.testing.assert_allclose(x, (5.0 * o0 + 1.0 * o1) / 6.0, rtol=1e-6)   # ... weighted by the steps each rank ran
    # single-process emulation of the same schedule: per step the MEAN of the gradients of the ranks that had a batch
    m, flat, grad = _make_model(0)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    plan0, plan1 = par.ShardPlan([0, 1, 2], 2, 0), par.ShardPlan([0, 1, 2], 2, 1)
    per_rank, n = par.epoch_schedule(plan0, BATCHES, ORDER)
    import copy
    bn_states = []
    for s in range(n):
        grad.zero_()
        contributors = 0
        for r in range(2):
            if s < len(per_rank[r]):
                tp_i, tp, bi = per_rank[r][s]
                x, y = _data(tp, bi)
                ((m(x) - y) ** 2).mean().backward()
                contributors += 1
        grad.div_(contributors)
        opt.step()
    # (BatchNorm running stats differ between the emulation and the replicas by construction; the
    # parameters only depend on batch statistics in train mode, so they must match exactly)
    np.testing.assert_allclose(f0, flat.numpy(), rtol=1e-12, atol=1e-14)
    for tp in (0, 1, 2):
        np.testing.assert_array_equal(c0[tp], np.full((4, 3), float(tp + 1)))
        np.testing.assert_array_equal(c1[tp], c0[tp])
    assert l0 == [(0, 1)] and l1 == [(1, 2)]                    # pairs sharded, disjoint
    for k in p0:
        np.testing.assert_array_equal(p0[k], p1[k])
        assert p0[k].shape == (4, 4) and p0[k].sum() > 0


def test_batch_granular_epoch_four_ranks_over_gloo():
    """World size 4, T = 5 (more ranks than a time-point sharding could use evenly; cfg3's situation on 8 GPUs in
    small): units dealt batch by batch, every rank has work in every global step but the last, replicas end identical and
    equal to a single-process emulation of the same schedule (per step the MEAN gradient of the contributing ranks)."""
    world, tps = 4, [0, 1, 2, 3, 4]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, tps, BATCHES5, ORDER5, "batch")) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    plan0 = par.ShardPlan(tps, world, 0, granularity="batch", batches_per_tp=BATCHES5)
    per_rank, n = par.epoch_schedule(plan0, BATCHES5, ORDER5)
    assert sorted(len(x) for x in per_rank) == [3, 3, 3, 3] and n == 3          # 12 units over 4 ranks
    assert all(r[1] == 3 for r in res) and [r[8] for r in res] == [3, 3, 3, 3]
    for r in res[1:]:
        np.testing.assert_array_equal(r[2], res[0][2])                           # replicas identical
        for x, y in zip(r[3], res[0][3]):
            np.testing.assert_array_equal(x, y)                                  # averaged BatchNorm statistics
        for tp in tps:
            np.testing.assert_array_equal(r[4][tp], res[0][4][tp])
            np.testing.assert_array_equal(r[9][tp], (np.arange(5 + 2 * tp) + tp) % 4)
        assert sorted(r[5]) == sorted(res[0][5])
        for k in r[5]:
            np.testing.assert_array_equal(r[5][k], res[0][5][k])
    assert sorted(sum((r[6] for r in res), [])) == [(0, 1), (1, 2), (2, 3), (3, 4)]   # pair solves: one owner each
    m, flat, grad = _make_model(0)
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    for s_ in range(n):
        grad.zero_()
        contributors = 0
        for r in range(world):
            if s_ < len(per_rank[r]):
                tp_i, tp, bi = per_rank[r][s_]
                x, y = _data(tp, bi)
                ((m(x) - y) ** 2).mean().backward()
                contributors += 1
        grad.div_(contributors)
        opt.step()
    np.testing.assert_allclose(res[0][2], flat.numpy(), rtol=1e-12, atol=1e-14)


def _loss_worker(rank, world, port, q, gran):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tps = [0, 1, 2, 3, 4]
        plan = par.ShardPlan(tps, world, rank, granularity=gran, batches_per_tp=BATCHES5)
        per_rank, _ = par.epoch_schedule(plan, BATCHES5, ORDER5)
        sums, counts = torch.zeros((5, 7)), torch.zeros(5)
        for tp_i, tp, bi in per_rank[rank]:
            sums[tp] += _batch_loss(tp, bi)
            counts[tp] += 1.0
        q.put((rank, par.reduce_epoch_losses(sums, counts, plan).numpy()))
    finally:
        dist.destroy_process_group()


def _batch_loss(tp, bi):
    """A made-up loss vector per (time point, batch): what a step would have returned."""
    g = torch.Generator().manual_seed(100 * tp + bi)
    return torch.rand(7, generator=g) * (tp + 1)


@pytest.mark.parametrize("world,gran", [(2, "batch"), (4, "batch"), (2, "timepoint")])
def test_epoch_loss_record_is_the_reference_quantity_whatever_the_rank_count(world, gran):
    """parallel.reduce_epoch_losses: the per-epoch record every rank ends with = the reference's loss.csv row
    (_train_utils.py:219-224: for each time point the mean of its batches' loss vectors, summed over the time points) --
    identical on all ranks and equal to what ONE process computes from all batches (P = 1), for both shard granularities."""
    want = torch.zeros(7, dtype=torch.float64)
    for tp, nb in BATCHES5.items():
        want += torch.stack([_batch_loss(tp, bi) for bi in range(nb)]).double().mean(0)
    one = par.reduce_epoch_losses(torch.stack([sum(_batch_loss(tp, bi) for bi in range(nb)) for tp, nb in sorted(BATCHES5.items())]),
                                  torch.tensor([float(nb) for _, nb in sorted(BATCHES5.items())]),
                                  par.ShardPlan([0, 1, 2, 3, 4], 1, 0, granularity="batch", batches_per_tp=BATCHES5))
    np.testing.assert_allclose(one.numpy(), want.numpy(), rtol=1e-6)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, world, port, q, gran)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, rec in res:
        assert rec.shape == (7,)
        np.testing.assert_array_equal(rec, res[0][1])                          # the same record on every rank
        np.testing.assert_allclose(rec, one.numpy(), rtol=1e-6)               # ... and the P = 1 record


# ---------------------------------------------------------------- sharded update (reduce-scatter .. all-gather)

def _cpu_clip_adamw(p, g, m, v, lo, hi, sumsq, t, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2, max_norm=0.3):
    """clip_grad_norm_(max_norm) + AdamW (_train_utils.py:214-217) on the slice [lo, hi) of flat CPU buffers, clipped by the
    GLOBAL squared norm `sumsq` -- the arithmetic the device kernel applies per element."""
    coef = min(1.0, max_norm / (float(sumsq) ** 0.5 + 1e-6))
    gs = g[lo:hi] * coef
    m[lo:hi] = b1 * m[lo:hi] + (1 - b1) * gs
    v[lo:hi] = b2 * v[lo:hi] + (1 - b2) * gs * gs
    p[lo:hi] *= 1 - lr * wd
    p[lo:hi] -= lr * (m[lo:hi] / (1 - b1 ** t)) / ((v[lo:hi] / (1 - b2 ** t)).sqrt() + eps)


def _local_grad(rank, step, n):
    return torch.randn(n, generator=torch.Generator().manual_seed(1000 * step + rank), dtype=torch.float64)


def _sharded_worker(rank, world, port, q, n, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = torch.linspace(-1.0, 1.0, n, dtype=torch.float64)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for t in range(1, steps + 1):
            g = _local_grad(rank, t, n)
            par.sharded_update(g, p, lambda lo, hi: (g[lo:hi] ** 2).sum().reshape(1),
                               lambda lo, hi, sq: _cpu_clip_adamw(p, g, m, v, lo, hi, sq, t))
        lo, hi, _ = par.shard_range(n, rank, world)
        q.put((rank, p.numpy().copy(), m[lo:hi].numpy().copy(), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 4096), (4, 4100), (4, 12)])
def test_sharded_update_leaves_identical_replicas_and_the_unsharded_parameters(world, n):
    """parallel.sharded_update: every rank reduce-scatters its gradient, clips by the all-reduced squared
    norm, updates ITS slice and all-gathers the parameters.  All ranks end with the same parameters, and they are what ONE
    process computes from the summed gradients with the unsharded update -- also when the buffer does not divide evenly
    (4100 over 4 ranks) and when there are more ranks than 4-element pieces (12 over 4: slices of 4, 4, 4, 0)."""
    steps = 3
    p = torch.linspace(-1.0, 1.0, n, dtype=torch.float64)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, steps + 1):
        g = sum(_local_grad(r, t, n) for r in range(world))
        _cpu_clip_adamw(p, g, m, v, 0, n, (g ** 2).sum(), t)
    assert [par.shard_range(12, r, 4)[:2] for r in range(4)] == [(0, 4), (4, 8), (8, 12), (12, 12)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q, n, steps)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for _, pp, mm, (lo, hi) in res:
        np.testing.assert_array_equal(pp, res[0][1])                                    # identical replicas
        np.testing.assert_allclose(pp, p.numpy(), rtol=1e-12, atol=1e-14)               # = the unsharded update
        np.testing.assert_allclose(mm, m[lo:hi].numpy(), rtol=1e-12, atol=1e-14)        # each rank keeps ITS slice's moments
