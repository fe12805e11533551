"""K-means on the device: the per-epoch `_update_Kmeans` of the reference fits
sklearn.cluster.KMeans(n_clusters, random_state=seed, n_init=10) on the host for every time point
(/root/reference/SpaDOT/utils/_train_utils.py:255-269) -- at cfg3 that costs more wall time per epoch than
the 100 training steps.  This module runs the same algorithm (k-means++ seeding with sklearn's candidate
rule, Lloyd iterations for all n_init restarts at once, tol = 1e-4 * mean feature variance, best inertia
wins) on the MI355X; random draws come from a host numpy RandomState seeded like sklearn's, everything
that touches the data stays in HBM.

Parity: sklearn's fit is third-party and not pinned bit for bit (its own chunked arithmetic decides ties and
the exact iteration count); what IS exact is the assignment rule -- labels are produced by the
spadot_kmeans_assign kernel (nearest centre, first minimum wins), the same rule sklearn's predict applies.
It is therefore opt-in: model_config['kmeans_backend'] = 'device' (default 'sklearn' = reference behaviour).
"""
import numpy as np
import torch

from .ops import kmeans_assign


class KMeansDevice:
    def __init__(self, n_clusters, random_state=1993, n_init=10, max_iter=300, tol=1e-4, check_every=8):
        self.k, self.seed, self.n_init, self.max_iter, self.tol, self.check_every = \
            int(n_clusters), int(random_state), int(n_init), int(max_iter), float(tol), int(check_every)

    # ---- k-means++ (sklearn _kmeans_plusplus: 2 + log(k) candidates per centre, best potential wins)
    def _init_centers(self, X, xsq, rs):
        n, d = X.shape
        k = self.k
        trials = 2 + int(np.log(k))
        centers = torch.empty((k, d), dtype=X.dtype, device=X.device)
        first = int(rs.choice(n))
        centers[0] = X[first]
        closest = (xsq - 2.0 * (X @ centers[0]) + centers[0].dot(centers[0])).clamp_(min=0)
        pot = closest.sum()
        for c in range(1, k):
            rv = torch.as_tensor(rs.uniform(size=trials), dtype=X.dtype, device=X.device) * pot
            cand = torch.searchsorted(torch.cumsum(closest, 0), rv).clamp_(max=n - 1)
            Xc = X[cand]                                                          # [trials, d]
            dist = (xsq[None, :] - 2.0 * (Xc @ X.T) + (Xc * Xc).sum(1)[:, None]).clamp_(min=0)
            dist = torch.minimum(dist, closest[None, :])
            pots = dist.sum(1)
            best = torch.argmin(pots)
            centers[c] = Xc[best]
            closest = dist[best]
            pot = pots[best]
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
        C = torch.stack([self._init_centers(Xc, xsq, np.random.RandomState(int(s))) for s in seeds])   # [R, k, d]
        R, k = C.shape[0], self.k
        done = torch.zeros(R, dtype=torch.bool, device=X.device)
        it = 0
        while it < self.max_iter:
            for _ in range(self.check_every):
                d2 = xsq[None, :, None] - 2.0 * torch.einsum("nd,rkd->rnk", Xc, C) + (C * C).sum(2)[:, None, :]
                lab = d2.argmin(2)                                                  # [R, n]
                # segment sums as a one-hot GEMM: no atomics, so two fits of the same data are bitwise identical
                onehot = torch.nn.functional.one_hot(lab, k).to(torch.float64)    # [R, n, k]
                sums = torch.einsum("rnk,nd->rkd", onehot, Xc)
                cnt = onehot.sum(1)
                newC = torch.where(cnt[:, :, None] > 0, sums / cnt.clamp(min=1)[:, :, None], C)   # empty cluster: keep
                shift = ((newC - C) ** 2).sum((1, 2))
                C = torch.where(done[:, None, None], C, newC)
                done = done | (shift <= tol)
                it += 1
                if it >= self.max_iter:
                    break
            if bool(done.all()):                          # one host sync per `check_every` Lloyd iterations
                break
        d2 = xsq[None, :, None] - 2.0 * torch.einsum("nd,rkd->rnk", Xc, C) + (C * C).sum(2)[:, None, :]
        inertia = d2.min(2).values.clamp_(min=0).sum(1)
        best = int(torch.argmin(inertia))
        centers = C[best] + mean
        labels = kmeans_assign(X, centers)                # the exact nearest-centre rule (HIP kernel)
        self.cluster_centers_ = centers.cpu().numpy()
        self.labels_ = labels.cpu().numpy()
        self.inertia_ = float(inertia[best])
        self.n_iter_ = it
        return self
