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

from .ops import kmeans_assign, lloyd_steps


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
