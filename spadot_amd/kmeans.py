"""K-means on the device: the per-epoch `_update_Kmeans` of the reference fits
sklearn.cluster.KMeans(n_clusters, random_state=seed, n_init=10) on the host for every time point
(/root/reference/SpaDOT/utils/_train_utils.py:255-269) -- at cfg3 that costs more wall time per epoch than
the 100 training steps.  This module runs the same algorithm (k-means++ seeding with sklearn's candidate
rule, Lloyd iterations for all n_init restarts at once (two HIP launches per iteration: spadot_lloyd_step), tol = 1e-4 * mean feature variance, best inertia
wins) on the MI355X; random draws come from a host numpy RandomState seeded like sklearn's, everything
that touches the data stays in HBM.

Parity: sklearn's fit is third-party and not pinned bit for bit (its own chunked arithmetic decides ties and
the exact iteration count); what IS exact is the assignment rule -- labels are produced by the
spadot_kmeans_assign kernel (nearest centre, first minimum wins), the same rule sklearn's predict applies.
Selected by model_config['kmeans_backend']: 'device' (default) | 'sklearn' (the reference's host fit, the parity option).
"""
import numpy as np
import torch

from .ops import kmeans_assign, lloyd_steps, lloyd_steps_groups


class KMeansDevice:
    def __init__(self, n_clusters, random_state=1993, n_init=10, max_iter=300, tol=1e-4, check_every=8):
        self.k, self.seed, self.n_init, self.max_iter, self.tol, self.check_every = \
            int(n_clusters), int(random_state), int(n_init), int(max_iter), float(tol), int(check_every)

    # ---- k-means++ (sklearn _kmeans_plusplus: 2 + log(k) candidates per centre, best potential wins),
    # all restarts at once: the random draws of every restart are made up front on the host (one RandomState per
    # restart, same call order as a restart-by-restart loop), the k - 1 selection rounds run batched in HBM
    def _init_centers(self, X, xsq, seeds):
        n, d = X.shape
        k, R = self.k, len(seeds)
        trials = 2 + int(np.log(k))
        first = np.empty(R, dtype=np.int64)
        U = np.empty((R, max(k - 1, 1), trials), dtype=np.float64)
        for r, s in enumerate(seeds):
            rs = np.random.RandomState(int(s))
            first[r] = int(rs.choice(n))
            for c in range(1, k):
                U[r, c - 1] = rs.uniform(size=trials)
        first = torch.as_tensor(first, device=X.device)
        U = torch.as_tensor(U, dtype=X.dtype, device=X.device)
        centers = torch.empty((R, k, d), dtype=X.dtype, device=X.device)
        centers[:, 0] = X[first]
        c0 = centers[:, 0]                                                        # [R, d]
        closest = (xsq[None, :] - 2.0 * (c0 @ X.T) + (c0 * c0).sum(1)[:, None]).clamp_(min=0)     # [R, n]
        pot = closest.sum(1)                                                      # [R]
        ar = torch.arange(R, device=X.device)
        for c in range(1, k):
            rv = U[:, c - 1] * pot[:, None]                                       # [R, trials]
            cand = torch.searchsorted(torch.cumsum(closest, 1), rv).clamp_(max=n - 1)
            Xc = X[cand]                                                          # [R, trials, d]
            dist = (xsq[None, None, :] - 2.0 * torch.matmul(Xc, X.T) + (Xc * Xc).sum(2)[:, :, None]).clamp_(min=0)
            dist = torch.minimum(dist, closest[:, None, :])                       # [R, trials, n]
            pots = dist.sum(2)                                                    # [R, trials]
            best = torch.argmin(pots, dim=1)                                      # [R]
            centers[:, c] = Xc[ar, best]
            closest = dist[ar, best]
            pot = pots[ar, best]
        return centers

    def fit(self, X):
        """X: [n, d] device tensor.  Sets cluster_centers_ (numpy [k, d]), labels_ (numpy int32 [n]),
        inertia_ (float); returns self."""
        assert X.is_cuda, "KMeansDevice runs on the MI355X"
        X = X.to(torch.float64)
        n, d = X.shape
        mean = X.mean(0)
        Xc = X - mean                                     # sklearn centres the data for accuracy
        xsq = (Xc * Xc).sum(1)
        tol = float(self.tol) * Xc.var(0, unbiased=False).mean()
        rs = np.random.RandomState(self.seed)
        seeds = rs.randint(np.iinfo(np.int32).max, size=self.n_init)
        C = self._init_centers(Xc, xsq, seeds)                                    # [R, k, d]
        R, k = C.shape[0], self.k
        C = C.contiguous()
        Xc = Xc.contiguous()
        done = torch.zeros(R, dtype=torch.int32, device=X.device)
        inertia = torch.zeros(R, dtype=torch.float64, device=X.device)
        part = torch.empty(R * ((n + 255) // 256) * (k * (d + 1) + 1), dtype=torch.float64, device=X.device)
        it = 0
        while it < self.max_iter:
            steps = min(self.check_every, self.max_iter - it)
            lloyd_steps(Xc, C, tol, done, inertia, part, steps)          # two launches per iteration, all restarts
            it += steps
            if bool(done.all()):                          # one host sync per `check_every` Lloyd iterations
                break
        # inertia of the final centres (a frozen restart's centres did not move: its last value is exact already)
        final = C.clone()
        lloyd_steps(Xc, final, -1.0, torch.ones_like(done), inertia, part, 1)
        best = int(torch.argmin(inertia))
        centers = C[best] + mean
        labels = kmeans_assign(X, centers)                # the exact nearest-centre rule (HIP kernel)
        self.cluster_centers_ = centers.cpu().numpy()
        self.labels_ = labels.cpu().numpy()
        self.inertia_ = float(inertia[best])
        self.n_iter_ = it
        return self


class KMeansResult:
    """What sklearn's fitted estimator exposes and _update_Kmeans reads: cluster_centers_, labels_, inertia_, n_iter_."""

    def __init__(self, centers, labels, inertia, n_iter):
        self.cluster_centers_, self.labels_, self.inertia_, self.n_iter_ = centers, labels, inertia, n_iter


class _ManyPlan:
    """Static buffers, constants and (from the second call on) captured hipGraphs of fit_many for one problem shape
    (device, sizes of the data sets, dimension, k, restarts, seed).  The per-epoch refit has the same shape every epoch
    and was HOST-bound (~420 launches, 11 ms of launch overhead for ~4 ms of device work), so its three phases --
    preparation + k-means++ selection rounds, a group of Lloyd iterations, final inertia + assignment -- are captured once
    and replayed; the only host synchronisations left are the convergence checks between Lloyd groups and the final copy."""

    def __init__(self, dev, ns, d, k, R, seed, max_iter, tol, check_every, in_dtype):
        self.dev, self.ns, self.d, self.k, self.R = dev, list(ns), d, k, R
        self.max_iter, self.check_every = int(max_iter), int(check_every)
        T, n_max = len(ns), max(ns)
        self.T, self.n_max = T, n_max
        f64 = torch.float64
        self.Xin = torch.zeros((sum(ns), d), dtype=in_dtype, device=dev)        # the caller's latents, copied in per call
        self.offs = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        # constants (created here, outside any capture)
        self.valid = torch.zeros((T, n_max), dtype=f64, device=dev)
        for t in range(T):
            self.valid[t, :ns[t]] = 1.0
        self.nf = torch.tensor(ns, dtype=f64, device=dev)
        self.nlast = torch.tensor([n - 1 for n in ns], device=dev).view(T, 1, 1)
        self.tol = float(tol)
        # random draws: per data set exactly those of KMeansDevice.fit -- the restart seeds come from
        # RandomState(random_state); each restart draws its first centre (depends on n), then `trials` uniforms per further
        # centre (they do not)
        rs = np.random.RandomState(int(seed))
        seeds = rs.randint(np.iinfo(np.int32).max, size=R)
        self.trials = trials = 2 + int(np.log(k))
        first = np.empty((T, R), dtype=np.int64)
        U = np.empty((T, R, max(k - 1, 1), trials), dtype=np.float64)
        for t in range(T):
            for r, sd in enumerate(seeds):
                g = np.random.RandomState(int(sd))
                first[t, r] = int(g.choice(ns[t]))
                for c in range(1, k):
                    U[t, r, c - 1] = g.uniform(size=trials)
        self.first = torch.as_tensor(first, device=dev)
        self.U = torch.as_tensor(U, device=dev)
        self.tt = torch.arange(T, device=dev)
        self.ti = self.tt[:, None].expand(T, R)
        self.ri = torch.arange(R, device=dev)[None, :].expand(T, R)
        self.xoff = torch.tensor(self.offs[:-1], dtype=torch.int32, device=dev)
        self.npts = torch.tensor(ns, dtype=torch.int32, device=dev)
        # state the phases share (fixed addresses)
        self.Xall = torch.zeros((sum(ns), d), dtype=f64, device=dev)            # centred data, all sets back to back
        self.means = torch.zeros((T, d), dtype=f64, device=dev)
        self.tolv = torch.zeros(T, dtype=f64, device=dev)
        self.C = torch.zeros((T * R, k, d), dtype=f64, device=dev)
        self.done = torch.zeros(T * R, dtype=torch.int32, device=dev)
        self.ones = torch.ones(T * R, dtype=torch.int32, device=dev)
        self.inertia = torch.zeros(T * R, dtype=f64, device=dev)
        self.part = torch.empty(T * R * ((n_max + 255) // 256) * (k * (d + 1) + 1), dtype=f64, device=dev)
        self.cen = torch.zeros((T, k, d), dtype=f64, device=dev)
        self.best_inertia = torch.zeros(T, dtype=f64, device=dev)
        self.labels = torch.zeros(sum(ns), dtype=torch.int32, device=dev)
        self.calls, self.graphs = 0, None

    # ---- the three phases, functions of the static buffers only (no host synchronisation, no host-side tensor creation)
    def _prepare_and_seed(self):
        T, R, k, d, ns, n_max = self.T, self.R, self.k, self.d, self.ns, self.n_max
        X64 = self.Xin.to(torch.float64)
        Xp = torch.zeros((T, n_max, d), dtype=torch.float64, device=self.dev)
        for t in range(T):
            x = X64[self.offs[t]:self.offs[t + 1]]
            m = x.mean(0)                                   # sklearn centres the data for accuracy
            self.means[t] = m
            xc = x - m
            self.Xall[self.offs[t]:self.offs[t + 1]] = xc
            Xp[t, :ns[t]] = xc
        xsq = (Xp * Xp).sum(2)                                                    # [T, n_max]
        torch.mul(xsq.sum(1) / (self.nf * d), self.tol, out=self.tolv)            # tol * mean feature variance
        tt, valid = self.tt, self.valid
        centers = self.C.view(T, R, k, d)
        c0 = Xp[tt[:, None], self.first]                                          # [T, R, d]
        centers[:, :, 0] = c0
        closest = (xsq[:, None, :] - 2.0 * torch.matmul(c0, Xp.transpose(1, 2)) + (c0 * c0).sum(2)[:, :, None]).clamp_(min=0)
        closest = closest * valid[:, None, :]                                     # [T, R, n_max]; padding weighs nothing
        pot = closest.sum(2)
        for c in range(1, k):
            rv = self.U[:, :, c - 1] * pot[:, :, None]                            # [T, R, trials]
            cand = torch.minimum(torch.searchsorted(torch.cumsum(closest, 2), rv), self.nlast)
            Xcand = Xp[tt[:, None, None], cand]                                   # [T, R, trials, d]
            dist = (xsq[:, None, None, :] - 2.0 * torch.matmul(Xcand, Xp[:, None].transpose(2, 3))
                    + (Xcand * Xcand).sum(3)[..., None]).clamp_(min=0)
            dist = torch.minimum(dist * valid[:, None, None, :], closest[:, :, None, :])       # [T, R, trials, n_max]
            pots = dist.sum(3)
            best = torch.argmin(pots, dim=2)                                      # [T, R]
            centers[:, :, c] = Xcand[self.ti, self.ri, best]
            closest = dist[self.ti, self.ri, best]
            pot = pots[self.ti, self.ri, best]
        self.done.zero_()

    def _lloyd_group(self):
        lloyd_steps_groups(self.Xall, self.C, self.xoff, self.npts, self.n_max, self.T, self.R, self.tolv, self.done,
                           self.inertia, self.part, self.check_every, skip_done=True)

    def _finish(self):
        T, R, k, d = self.T, self.R, self.k, self.d
        final = self.C.clone()                  # inertia of the FINAL centres: one assignment pass with every restart frozen
        lloyd_steps_groups(self.Xall, final, self.xoff, self.npts, self.n_max, T, R, self.tolv, self.ones, self.inertia,
                           self.part, 1)
        best = torch.argmin(self.inertia.view(T, R), dim=1)                       # [T]
        self.best_inertia.copy_(self.inertia.view(T, R)[self.tt, best])
        self.cen.copy_(self.C.view(T, R, k, d)[self.tt, best] + self.means[:, None, :])
        X64 = self.Xin.to(torch.float64)
        for t in range(T):                      # the exact nearest-centre rule on the ORIGINAL coordinates (HIP kernel)
            self.labels[self.offs[t]:self.offs[t + 1]] = kmeans_assign(X64[self.offs[t]:self.offs[t + 1]], self.cen[t].contiguous())

    def _capture(self):
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        graphs = []
        for fn in (self._prepare_and_seed, self._lloyd_group, self._finish):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                fn()
            graphs.append(g)
        self.graphs = graphs

    def run(self, Xs, use_graphs=True):
        off = 0
        for x in Xs:
            self.Xin[off:off + x.shape[0]].copy_(x)
            off += x.shape[0]
        self.calls += 1
        if use_graphs and self.graphs is None and self.calls >= 2 and not torch.cuda.is_current_stream_capturing():
            self._capture()                      # (first call: eager -- library handles, allocator warm-up)
        phases = [g.replay for g in self.graphs] if (use_graphs and self.graphs is not None) else \
            [self._prepare_and_seed, self._lloyd_group, self._finish]
        phases[0]()
        it = 0
        while it < self.max_iter:
            phases[1]()
            it += self.check_every
            if bool(self.done.all()):            # one host sync per group of Lloyd iterations, for all data sets
                break
        phases[2]()
        cen = self.cen.cpu().numpy()             # (synchronises)
        inert = self.best_inertia.cpu().numpy()
        lab = self.labels.cpu().numpy()
        return [KMeansResult(cen[t].copy(), lab[self.offs[t]:self.offs[t + 1]].copy(), float(inert[t]), it) for t in range(self.T)]


_PLANS = {}


def fit_many(Xs, n_clusters, random_state=1993, n_init=10, max_iter=300, tol=1e-4, check_every=8, use_graphs=True):
    """KMeansDevice(...).fit(X) for SEVERAL data sets (the latents of all time points, refitted every epoch:
    _train_utils.py:255-269) with the work of all of them batched: the k-means++ selection rounds run on [T, R, n_max]
    tensors (shorter sets are padded with rows that can never be drawn and weigh nothing), the Lloyd iterations are one
    launch pair per iteration for all T * R restarts (spadot_lloyd_step_groups; converged restarts drop out), and from the
    second call with the same sizes on the three phases are replayed hipGraphs (_ManyPlan).  Same algorithm, same random
    draws per data set as fit(); the arithmetic of a batched matrix product may round differently from the unbatched one,
    so centres agree to rounding, not bit for bit (labels then follow from the exact assignment kernel).  The iteration
    count is checked every `check_every` Lloyd iterations, so max_iter is honoured up to that granularity.
    Returns a list of KMeansResult."""
    if len(Xs) == 0:
        return []
    dev = Xs[0].device
    assert all(x.is_cuda and x.dim() == 2 and x.shape[1] == Xs[0].shape[1] and x.dtype == Xs[0].dtype for x in Xs), \
        "fit_many runs on the MI355X, on data sets of one dimension and dtype"
    ns = tuple(int(x.shape[0]) for x in Xs)
    key = (str(dev), ns, int(Xs[0].shape[1]), int(n_clusters), int(n_init), int(random_state), int(max_iter), float(tol),
           int(check_every), Xs[0].dtype)
    plan = _PLANS.get(key)
    if plan is None:
        if len(_PLANS) >= 8:                     # a few shapes at most are alive in a run; do not hoard buffers
            _PLANS.clear()
        plan = _PLANS[key] = _ManyPlan(dev, ns, int(Xs[0].shape[1]), int(n_clusters), int(n_init), int(random_state), max_iter,
                                       tol, check_every, Xs[0].dtype)
    return plan.run(Xs, use_graphs=use_graphs)
