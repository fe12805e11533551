"""Sparse variational GP branch: mirror of /root/reference/SpaDOT/model/svgp.py (same class names and
per-latent-dimension methods) plus the batched path the training step actually uses.

What changed relative to the reference's arithmetic (same values, far less work -- SURVEY 7 "hard parts"):
  * K_mm, (K_mm + jI)^-1 and log|K_mm + jI| are constants of the run (inducing points and kernel scale
    are not trainable, svgp.py:24-30): computed once per time point, not 20x per step;
  * K_nm and everything built from it and K_mm alone are constants of a batch: cached per batch;
  * all L latent dimensions are solved together (batched m x m algebra) instead of a Python loop;
  * diag(A B^T) products never form the b x b (or N_t x N_t) matrix (svgp.py:67,80,98);
  * the (b, m, m) tensor of svgp.py:99-101 is replaced by the identity
        tr(A_hat K^-1 k k^T K^-1) = (k^T K^-1 K_mm) Sigma^-1 (K_mm K^-1 k);
  * K(x, x)'s diagonal is k(0) = 1 for all three kernel profiles.
The m x m algebra stays in fp64 whatever the model's compute dtype: Sigma_l has a condition number of
1e6-1e7 (jitter 1e-2), which fp32 cannot invert.  Kernel matrices, row-wise dot products and the
scalar ELBO reductions are hand-written HIP kernels (spadot_amd.ops); inverses/Cholesky are the
library's batched routines.
"""
import torch
import torch.nn as nn

from .._lib import model_lib
from ..ops import stamp_if, _SPLIT, _check, _p, _stream, elbo_reduce, kernel_matrix, rowdot, spd_inverse_logdet

SWEEP_DIRECT_M = _SPLIT[0]      # up to here one sweep launch takes the matrices as they are (ops._spd_inverse_logdet_nograd)

F64 = torch.float64


def _contract_rows(K, Y):
    """sum_b K[b, :]^T Y[b, :] -> [L, m] for K [b, m], Y [b, L] (einsum "bm,bl->lm").  For a whole time point (b = N_t ~ 10^4) the
    library runs this 236 x 20 output as ONE workgroup walking all b rows (0.29 ms for 0.09 GFLOP, a sixth of the per-epoch
    inference of a time point); in slabs of rows it is a batched product + a fixed-order sum over the slabs (two launches)."""
    b, m = K.shape
    L = Y.shape[1]
    slab = 256
    if b < 8 * slab:
        return torch.einsum("bm,bl->lm", K, Y)
    nfull = b // slab
    part = torch.bmm(Y[:nfull * slab].reshape(nfull, slab, L).transpose(1, 2), K[:nfull * slab].reshape(nfull, slab, m))   # [nfull, L, m]
    t = part.sum(0)
    if nfull * slab < b:
        t = t + Y[nfull * slab:].T @ K[nfull * slab:]
    return t


class Kernel(nn.Module):
    """svgp.py:107-125."""

    def __init__(self, kernel_type="Gaussian", scale=0.1, dtype=F64, device="cpu"):
        super().__init__()
        self.kernel_type = kernel_type
        self.scale = torch.tensor([scale], dtype=F64, device=device)
        self._scale = float(scale)

    def forward(self, x, y):
        return kernel_matrix(x.to(F64), y.to(F64), self.kernel_type, self._scale)


class BatchConstants:
    """Everything about (time point, batch coordinates) that does not depend on the encoder output."""
    __slots__ = ("K_nm", "ktilde", "Q", "P", "X2", "c", "b")


class RunConstants:
    """Per time point: everything built from the inducing points alone (not trainable: svgp.py:24-30)."""
    __slots__ = ("K_mm", "K_inv", "logdet_K", "logdet_K_f", "eye", "KjI", "K2j", "M", "m", "mlogj")


class _SVGPCore(torch.autograd.Function):
    """The encoder-dependent part of svgp.py:47-104 for all L latent dimensions plus SpaDOT.py:72-77, forward and
    HAND-WRITTEN backward: (mu, var) -> (p_m, p_v, SVGP_KL).  Autograd through the same algebra costs ~200
    launches a step, this ~30.

    With S_l = (K + jI + c K_mn diag(w_l) K_nm)^-1, w = 1/var, t_l = K_mn (mu_l w_l), r_l = S_l t_l,
    M = K K_j^-1 K and P = K_nm K_j^-1 K (batch constant):
        p_m = c K_nm r          mv = K_nm K_j^-1 mu_hat = c P r
        p_v = k~ + diag(K_nm S K_mn)        tr = diag(P S P^T)
        KL_l = 1/2 (log|K_j| - log|A_hat_l + jI| - m + <S_l, M> + c^2 r_l^T M r_l)
    (tr(K_j^-1 A_hat) = <S, M> and mu_hat^T K_j^-1 mu_hat = c^2 r^T M r: A_hat and mu_hat are never formed),
    log|A_hat + jI| by Sylvester's identity from log|Sigma| and log|Sigma + K^2/j| (one sweep, 2L matrices);
    l3, the cross entropy and SVGP_KL = -|ce - (l3 - (b/N) KL)| / L in the same kernel (k_svgp_post_fwd).
    Backward, with D_l = K_mn diag(G_pv) K_nm + P^T diag(G_tr) P + g/2 M, dr = c K_mn G_pm + c P^T G_mv + g c^2 M r,
    dt = S dr:   dSigma_l = -S D S - dt r^T + g/2 (S - S2),  S2 = (Sigma + K^2/j)^-1,
        dw = c diag(K_nm dSigma K_mn) + mu (K_nm dt),   dmu = w (K_nm dt),   dvar = -dw w^2  (+ the direct terms)."""

    @staticmethod
    def start(z, bc, rc, partials=None):
        """First half of forward, from z = (mu | logvar) [b, 2L] fp32: Sigma_l for every latent dim, the sweep launch
        (the long pole of the branch, ~0.2 ms on 2L compute units) and t.  Separate so that the caller can issue
        it early.  partials = (pz, their number, SVGP_fc's bias): z is NOT filled yet -- the encoder left SVGP_fc as partial
        products (encoder.SVGPEncoder.pre_head(defer_fc=True)) and the first kernel here sums them, storing z on the way."""
        with torch.no_grad():
            assert partials is None or (z.dtype == torch.float32 and z.is_contiguous())
            z = z.contiguous().float()
            b, L = z.shape[0], z.shape[1] // 2
            m, c = rc.m, bc.c
            Kn = bc.K_nm
            mu, var, w, muw = (torch.empty((b, L), dtype=F64, device=z.device) for _ in range(4))
            lib = model_lib()
            if m <= SWEEP_DIRECT_M:
                # one stored G_l = c K_mn diag(w_l) K_nm; the sweep adds (K + jI) -- and K^2/j for the second set of
                # L matrices -- while it loads (no copy / add launches in front of the inverse: every short launch of
                # this branch waits for a free slot beside the GAT branch's GEMMs)
                A = torch.empty((L, b, m), dtype=F64, device=z.device)       # diag(w_l) K_nm, written by the pre kernel
                if partials is not None:
                    pz, npz, bfc = partials
                    _check(lib.spadot_svgp_pre2_partials(_p(pz), int(npz), _p(bfc), _p(Kn), b, L, m, _p(z), _p(mu), _p(var), _p(w), _p(muw),
                                                         _p(A), _stream()), "spadot_svgp_pre2_partials")
                else:
                    _check(lib.spadot_svgp_pre2(_p(z), _p(Kn), b, L, m, _p(mu), _p(var), _p(w), _p(muw), _p(A), _stream()),
                           "spadot_svgp_pre2")
                stamp_if(23)                                                 # (SPADOT_STAMPS=1 only: pre2 done)
                G = torch.empty((L, m, m), dtype=F64, device=z.device)
                torch.baddbmm(G, A.transpose(1, 2), Kn.unsqueeze(0).expand(L, b, m), beta=0.0, alpha=c, out=G)
                stamp_if(28)                                                 # (SPADOT_STAMPS=1 only: G done)
                t = muw.T @ Kn                                               # [L, m]
                stamp_if(17)                                                 # (SPADOT_STAMPS=1 only: Sigma built, in front of the inverse)
                X = torch.empty((2 * L, m, m), dtype=F64, device=z.device)
                ld = torch.empty(2 * L, dtype=F64, device=z.device)
                _check(lib.spadot_spd_inverse_logdet2(_p(G), L, 2 * L, m, _p(rc.KjI), _p(rc.K2j), _p(X), _p(ld), _stream()),
                       "spadot_spd_inverse_logdet2")
                return mu, var, w, X, ld, t
            if partials is not None:
                pz, npz, bfc = partials
                _check(lib.spadot_enc_sum_z(_p(pz), int(npz), _p(bfc), b, 2 * L, _p(z), _stream()), "spadot_enc_sum_z")
            _check(lib.spadot_svgp_pre(_p(z), b, L, _p(mu), _p(var), _p(w), _p(muw), _stream()), "spadot_svgp_pre")
            A = Kn.unsqueeze(0) * w.T.unsqueeze(2)                           # [L, b, m] = diag(w_l) K_nm
            buf = torch.empty((2 * L, m, m), dtype=F64, device=mu.device)
            torch.baddbmm(rc.KjI.expand(L, m, m), A.transpose(1, 2), Kn.unsqueeze(0).expand(L, b, m), alpha=c, out=buf[:L])
            torch.add(buf[:L], rc.K2j, out=buf[L:])
            t = muw.T @ Kn                                                   # [L, m]
            X, ld = spd_inverse_logdet(buf)                                  # last launch of this half: see SpaDOT.forward
        return mu, var, w, X, ld, t

    @staticmethod
    def forward(ctx, z, bc, rc, started, b_over_N):
        mu, var, w, X, ld, t = started if started is not None else _SVGPCore.start(z, bc, rc)
        b, L = mu.shape
        m, c = rc.m, bc.c
        X2 = bc.X2
        lib = model_lib()
        S = X[:L]
        dev = mu.device
        # r_l = S_l t_l, Mr_l = M r_l (M symmetric), raw = X2 r^T, sm_l = <S_l, M>: two launches for all of them
        r = torch.empty((L, m), dtype=F64, device=dev)
        Mr = torch.empty((L, m), dtype=F64, device=dev)
        raw = torch.empty((2 * b, L), dtype=F64, device=dev)
        sm = torch.empty(L, dtype=F64, device=dev)
        nparts = (m + 3) // 4 * 4
        smpart = torch.empty(L * nparts, dtype=F64, device=dev)
        _check(lib.spadot_svgp_mid(_p(S), _p(t.contiguous()), _p(rc.M), _p(X2), L, m, 2 * b, _p(r), _p(Mr), _p(raw), _p(sm),
                                   _p(smpart), L * nparts, _stream()), "spadot_svgp_mid")
        p_m, mv, p_v, tr = (torch.empty((b, L), dtype=F64, device=dev) for _ in range(4))
        out4 = torch.empty(4, dtype=F64, device=dev)
        skl32 = torch.empty(1, dtype=torch.float32, device=dev)
        kl_const = rc.logdet_K_f - rc.mlogj - m
        holder = {"M": rc.M, "Kn": bc.K_nm, "S2": X[L:], "b": b}
        if ELBO_LATE[0] is not None:
            # The loss tail waits for p_m and p_v only: K_nm S_l (half of X2 S_l), its row dot and a small kernel.  The other
            # half (P S_l), mv / tr and the ELBO scalars (l3, ce, kl, SVGP_KL: read by backward() and, as VALUES, by the
            # logging vector, which is queued too) are appended to the open queue and run off the critical chain.
            KS = torch.matmul(bc.K_nm, S)                                    # [L, b, m]
            rd_a = rowdot(KS, bc.K_nm)                                       # [L, b]
            _check(lib.spadot_svgp_post_pm_pv(_p(raw), _p(rd_a), _p(bc.ktilde), b, L, c, _p(p_m), _p(p_v), _stream()),
                   "spadot_svgp_post_pm_pv")
            holder["KS"] = KS

            def rest(KS=KS, rd_a=rd_a, holder=holder):
                PS = torch.matmul(bc.P, S)                                   # [L, b, m]
                rd = torch.cat([rd_a, rowdot(PS, bc.P)], dim=1)              # [L, 2b]
                scratch = torch.empty((2, b, L), dtype=F64, device=dev)      # (p_m, p_v again: the tail may be reading the real ones)
                _check(lib.spadot_svgp_post_forward(_p(raw), _p(rd), _p(r), _p(Mr), _p(ld), _p(sm), _p(mu), _p(var), _p(bc.ktilde),
                                                    b, L, m, c, kl_const, b_over_N, _p(scratch[0]), _p(mv), _p(scratch[1]), _p(tr),
                                                    _p(out4), _p(skl32), _stream()), "spadot_svgp_post_forward")
                holder["PS"] = PS
            ELBO_LATE[0].append(rest)
        else:
            X2S = torch.matmul(X2, S)                                        # [L, 2b, m]
            rd = rowdot(X2S, X2)                                             # [L, 2b]
            _check(lib.spadot_svgp_post_forward(_p(raw), _p(rd), _p(r), _p(Mr), _p(ld), _p(sm), _p(mu), _p(var), _p(bc.ktilde),
                                                b, L, m, c, kl_const, b_over_N, _p(p_m), _p(mv), _p(p_v), _p(tr),
                                                _p(out4), _p(skl32), _stream()), "spadot_svgp_post_forward")
            holder["X2S"] = X2S
        ctx.save_for_backward(mu, var, w, X, r, Mr, p_m, p_v, mv, tr, out4)
        ctx.bc, ctx.rc, ctx.bN = bc, rc, b_over_N
        # what backward() needs that does NOT depend on the incoming gradients (q2 = diag(K_nm S2 K_mn), K_nm S_l, T = X2 S_l
        # K_mn, m0): precompute_backward() may fill this holder between forward and backward -- GraphedStepper does, on the
        # side stream while the loss tail runs on the main stream and the side stream would idle; backward() computes
        # whatever the holder lacks
        ctx.holder = holder                       # (reachable from the outputs: holder_of(p_m))
        ctx.mark_non_differentiable(out4)
        ctx.set_materialize_grads(False)          # no zero-filled gradient tensors for outputs the loss does not use
        return p_m, p_v, skl32[0], out4

    @staticmethod
    def backward(ctx, G_pm, G_pv, g_skl, _unused):
        mu, var, w, X, r, Mr, p_m, p_v, mv, tr, out4 = ctx.saved_tensors
        bc, rc = ctx.bc, ctx.rc
        pre = ctx.holder
        b, L = mu.shape
        m, c = rc.m, bc.c
        Kn, X2 = bc.K_nm, bc.X2
        S, S2 = X[:L], X[L:]
        lib = model_lib()
        dev = mu.device
        g_mu, g_var = (torch.empty((b, L), dtype=F64, device=dev) for _ in range(2))
        dz = torch.empty((b, 2 * L), dtype=torch.float32, device=dev)
        G1 = torch.empty((2 * b, L), dtype=F64, device=dev)
        G2T = torch.empty((L, 2 * b), dtype=F64, device=dev)
        g_kl = torch.empty(1, dtype=F64, device=dev)
        gMr = torch.empty((L, m), dtype=F64, device=dev)
        gM = torch.empty((m, m), dtype=F64, device=dev)
        keep = [None if g is None else g.contiguous() for g in (G_pm, G_pv)]
        gs = None if g_skl is None else g_skl.contiguous().float()
        opt = lambda t_: None if t_ is None else _p(t_)
        _check(lib.spadot_svgp_post_backward(opt(gs), _p(out4), opt(keep[0]), opt(keep[1]), _p(mu), _p(var), _p(mv), _p(tr),
                                             _p(p_m), _p(p_v), _p(bc.ktilde), _p(Mr), _p(rc.M), b, L, m, c, ctx.bN, _p(g_mu),
                                             _p(g_var), _p(G1), _p(G2T), _p(g_kl), _p(gMr), _p(gM), _stream()),
               "spadot_svgp_post_backward")
        dr = gMr.addmm_(G1.T, X2, alpha=c)                               # [L, m] (in place: no copy of gMr in front of the GEMM)
        Kdt = None
        if MID_BWD[0]:
            # dt_l = S_l dr_l and K_nm dt^T are the forward's r = S t / raw = X2 r with other operands: the same two
            # wave-per-row launches (spadot_svgp_mid) instead of a batched library GEMM with ONE output column
            # (MT64x128x16: 41 us in the step) and a [b, m] x [m, L] product (27 us)
            dt = torch.empty((L, m), dtype=F64, device=dev)
            Kdt = torch.empty((b, L), dtype=F64, device=dev)
            nparts = (m + 3) // 4 * 4
            junk = torch.empty(L * m + L + L * nparts, dtype=F64, device=dev)
            _check(lib.spadot_svgp_mid(_p(S), _p(dr), _p(rc.M), _p(Kn), L, m, b, _p(dt), _p(junk), _p(Kdt), _p(junk[L * m:]),
                                       _p(junk[L * m + L:]), L * nparts, _stream()), "spadot_svgp_mid")
        else:
            dt = torch.bmm(S, dr.unsqueeze(2)).squeeze(2)                # [L, m]
        if "Ta" in pre:
            q1 = torch.empty((L, b), dtype=F64, device=dev)
            _check(lib.spadot_svgp_q1t(_p(pre["Ta"]), _p(pre["Tb"]), _p(G2T), _p(pre["m0"]), _p(g_kl), L, b, b, _p(q1),
                                       _stream()), "spadot_svgp_q1t")
        else:
            A2 = X2.unsqueeze(0) * G2T.unsqueeze(2)                      # [L, 2b, m]
            D = torch.baddbmm(gM.expand(L, m, m), A2.transpose(1, 2), X2.unsqueeze(0).expand(L, 2 * b, m))
            KS = _holder_KS(pre)                                         # [L, b, m] = K_nm S_l, contiguous
            q1 = rowdot(torch.bmm(KS, D).reshape(1, L * b, m), KS.reshape(L * b, m)).reshape(L, b)
        q2 = pre["q2"] if "q2" in pre else rowdot(torch.matmul(Kn, S2), Kn)     # diag(K_nm S2 K_mn)  [L, b]
        if Kdt is None:
            Kdt = Kn @ dt.T                                              # [b, L]
        _check(lib.spadot_svgp_grad_tail(_p(q1), _p(q2), _p(Kdt), _p(p_v), _p(bc.ktilde), _p(p_m), _p(mu), _p(w), _p(g_kl),
                                         _p(g_mu), _p(g_var), b, L, c, None, None, _p(dz), _stream()), "spadot_svgp_grad_tail")
        return dz, None, None, None, None


# test hooks (tests/test_model_gpu.py compares the restructured backward with the plain one); both forms are the default path
Q1T = [True]            # q1 through T = X2 S_l K_mn formed ahead of the backward (precompute_backward); [False]: the D_l route
MID_BWD = [True]        # dt = S dr and K_nm dt^T on the forward's wave-per-row kernels; [False]: library products
# a list while a caller wants the part of forward() the loss tail does not wait for queued instead of run (GraphedStepper's
# svgp_pre stage runs the queue right behind the SVGP forward graph, beside the tail); None: forward() runs everything
ELBO_LATE = [None]


def holder_of(out):
    """The precompute holder of the _SVGPCore.forward that produced `out` (its p_m, p_v or SVGP_KL): the backward node of a
    custom Function IS its ctx, so the holder travels with the outputs -- no module-level "most recent forward" that would
    pin one model's fp64 [L, m, m] tensors and couple concurrent models."""
    node = getattr(out, "grad_fn", None)
    h = getattr(node, "holder", None)
    if h is None:
        raise ValueError("not an output of _SVGPCore.forward (or detached): no precompute holder")
    return h


def _holder_KS(h):
    if "KS" not in h:
        h["KS"] = h["X2S"][:, :h["b"]].contiguous()
    return h["KS"]


def precompute_backward(holder):
    """The gradient-independent products of _SVGPCore.backward, computed ahead of the backward pass into `holder` (holder_of
    an output of that forward): q2 = diag(K_nm S2 K_mn) (a [b, m] x [L, m, m] product + a row dot: 94 + 32 us inside
    the backward pair of a cfg3 step), the contiguous K_nm S_l, and -- Q1T -- T = X2 S_l K_mn and m0 (below).  A forward
    that queued the rest of its ELBO (ELBO_LATE) must have had that queue run before.  No-op when already done."""
    h = holder
    if h is None or "q2" in h:
        return h
    Kn, S2, b = h["Kn"], h["S2"], h["b"]
    with torch.no_grad():
        h["q2"] = rowdot(torch.matmul(Kn, S2), Kn)
        _holder_KS(h)
        if Q1T[0]:
            _form_T(h)
    return h


def _form_T(h):
    """q1 = diag(K_nm S D S K_mn) with D = X2^T diag(G2) X2 + g/2 M needs the gradients only as WEIGHTS of squares:
        q1[l, i] = sum_n G2[l, n] T_l[n, i]^2 + g/2 m0[l, i],   T_l = X2 S_l K_mn [2b, b],   m0_l = diag(K_nm S_l M S_l K_mn).
    T (as its halves Ta = K_nm S_l K_mn, Tb = P S_l K_mn) and m0 are formed here: 2.5 + 0.6 GFLOP, beside the loss tail; the
    backward pass then needs ONE reduction launch (spadot_svgp_q1t) in place of a scaled copy of X2, the D product, K_nm S_l D_l
    and a row dot (~140 us in the step)."""
    Kn, b = h["Kn"], h["b"]
    KS = _holder_KS(h)
    with torch.no_grad():
        PS = h["PS"] if "PS" in h else h["X2S"][:, b:]
        h["Ta"] = torch.matmul(KS, Kn.t())                                   # [L, b, b]
        h["Tb"] = torch.matmul(PS, Kn.t())                                   # [L, b, b]
        Lb = KS.shape[0] * KS.shape[1]                                       # (rowdot's second operand is ONE [rows, m] matrix)
        h["m0"] = rowdot(torch.matmul(KS, h["M"]).reshape(1, Lb, -1), KS.reshape(Lb, -1)).reshape(KS.shape[0], KS.shape[1])
    return h


class SVGP(nn.Module):
    def __init__(self, model_config, inducing_points, N_train, jitter=1e-2):
        super().__init__()
        self.N_train = N_train
        self.jitter = jitter
        dev = model_config["device"]
        self.inducing_index_points = torch.as_tensor(inducing_points, dtype=F64).to(dev)
        self.kernel = Kernel(kernel_type=model_config["kernel_type"], scale=model_config["kernel_scale"], device=dev)
        self._consts = None
        self._batch_cache = {}

    # ---- run constants -------------------------------------------------------------------
    def _run_constants(self):
        if self._consts is None:
            z = self.inducing_index_points
            m = z.shape[0]
            K_mm = self.kernel(z, z)
            K_j = K_mm + self.jitter * torch.eye(m, dtype=F64, device=z.device)
            K_inv = torch.linalg.inv(K_j)
            logdet = 2.0 * torch.sum(torch.log(torch.diagonal(torch.linalg.cholesky(K_j))))
            self._consts = (K_mm, K_inv, logdet, torch.eye(m, dtype=F64, device=z.device))
        return self._consts

    def batch_constants(self, x, key=None):
        """K_nm, k~ = K_nn - diag(K_nm K^-1 K_mn), Q = K_nm K^-1, P = Q K_mm for coordinates x [b, 2]."""
        if key is not None and key in self._batch_cache:
            return self._batch_cache[key]
        K_mm, K_inv, _, _ = self._run_constants()
        bc = BatchConstants()
        x = x.to(F64)
        bc.b = x.shape[0]
        bc.c = float(self.N_train) / bc.b
        bc.K_nm = self.kernel(x, self.inducing_index_points)
        bc.Q = bc.K_nm @ K_inv
        bc.P = bc.Q @ K_mm
        bc.ktilde = 1.0 - rowdot(bc.Q.unsqueeze(0), bc.K_nm)[0]          # K(x,x)_ii = k(0) = 1
        bc.X2 = torch.cat([bc.K_nm, bc.P], dim=0).contiguous()            # [2b, m]: both row sets one GEMM serves
        if key is not None:
            self._batch_cache[key] = bc
        return bc

    # ---- batched path --------------------------------------------------------------------
    def _sigma_inv(self, bc, W, want_logdet_A=False):
        """Sigma_l'^-1 = (K_mm + c K_mn diag(w_l) K_nm + jI)^-1 for all latent dims  [L, m, m].
        With want_logdet_A it also returns log|A_hat_l + jI| (svgp.py:88,90) WITHOUT a second, dependent
        factorisation: by Sylvester's identity
            |K S K + jI| = j^m |S| |S^-1 + K K / j|,      S = Sigma_l'^-1,  K = K_mm,
        so  log|A_hat + jI| = m log j - log|Sigma_l'| + log|Sigma_l' + K_mm^2 / j|,
        and the two log-determinants are of INDEPENDENT matrices: one sweep launch over 2L matrices gives
        both (the chain Sigma^-1 -> A_hat -> second factorisation was the critical path of the step)."""
        K_mm, _, _, eye = self._run_constants()
        A = bc.K_nm.unsqueeze(0) * W.T.unsqueeze(2)                        # [L, b, m] = diag(w_l) K_nm
        sigma = K_mm.unsqueeze(0) + bc.c * torch.matmul(A.transpose(1, 2), bc.K_nm) + self.jitter * eye
        if not want_logdet_A:
            return spd_inverse_logdet(sigma, need_logdet=False)[0]
        L, m = sigma.shape[0], sigma.shape[1]
        X, ld = spd_inverse_logdet(torch.cat([sigma, sigma + self._k2_over_j()], dim=0))
        logdet_A = m * float(torch.log(torch.tensor(self.jitter, dtype=F64))) - ld[:L] + ld[L:]
        return X[:L], logdet_A

    def _k2_over_j(self):
        if getattr(self, "_k2j", None) is None:
            K_mm = self._run_constants()[0]
            self._k2j = ((K_mm @ K_mm) / self.jitter).unsqueeze(0)
        return self._k2j

    def posterior(self, bc_train, mu, var, bc_test=None, want_logdet_A=False, want_var=True):
        """Posterior mean and variance at the test points for all latent dims at once.
        mu, var: [b, L] (encoder output at the training points).  Returns (p_m, p_v, extras); want_var=False skips the
        variance (p_v = None): the per-epoch inference only reads the means (SpaDOT.py:121), and K_nm S_l for b = N_t
        rows is the larger of its two batched products."""
        mu, var = mu.to(F64), var.to(F64)
        bt = bc_train if bc_test is None else bc_test
        W = 1.0 / var
        logdet_A = None
        if want_logdet_A:
            S_inv, logdet_A = self._sigma_inv(bc_train, W, True)
        else:
            S_inv = self._sigma_inv(bc_train, W)
        t = _contract_rows(bc_train.K_nm, mu * W)                          # K_mn (y / noise)  [L, m]
        St = torch.einsum("lmn,ln->lm", S_inv, t)
        p_m = bc_train.c * (bt.K_nm @ St.T)                                # [b_test, L]
        if not want_var:
            return p_m, None, (S_inv, St, logdet_A)
        KS = torch.einsum("bm,lmn->lbn", bt.K_nm, S_inv)                  # [L, b_test, m]
        p_v = bt.ktilde.unsqueeze(1) + rowdot(KS, bt.K_nm).T
        return p_m, p_v, (S_inv, St, logdet_A)

    def elbo_terms(self, bc, mu, var):
        """(p_m, p_v, l3_sum, kl_sum, ce_sum) of one training batch: svgp.py:47-104 over all latent
        dimensions + the Gaussian cross entropy of SpaDOT.py:74-75 (values only; the training step uses
        elbo_start/elbo_finish, whose SVGP_KL carries the gradient)."""
        z = torch.cat([mu, torch.log(var)], dim=1)
        p_m, p_v, _, out4 = self._finish(bc, self.elbo_start(bc, z))
        return p_m, p_v, out4[0], out4[2], out4[1]

    def elbo_start(self, bc, z, partials=None):
        """z = SVGP_fc output (mu | logvar) [b, 2L].  Builds Sigma_l and launches the batched inverse; elbo_finish()
        does the rest.  Two calls so that the composite model can issue the GAT kernels in between."""
        return z, _SVGPCore.start(z.detach(), bc, self._rc(), partials)

    def _finish(self, bc, started):
        z, pre = started
        return _SVGPCore.apply(z, bc, self._rc(), pre, bc.b / float(self.N_train))

    def elbo_finish(self, bc, started):
        """(p_m, p_v, SVGP_KL): posterior at the batch points and -|ce - (l3 - b/N KL)| / L (SpaDOT.py:72-77: the
        sign trick without its host round trip)."""
        p_m, p_v, skl, _ = self._finish(bc, started)
        return p_m, p_v, skl

    def _rc(self):
        if getattr(self, "_rc_obj", None) is None:
            K_mm, K_inv, logdet_K, eye = self._run_constants()
            rc = RunConstants()
            rc.K_mm, rc.K_inv, rc.logdet_K, rc.eye = K_mm, K_inv, logdet_K, eye
            rc.m = K_mm.shape[0]
            rc.KjI = (K_mm + self.jitter * eye).contiguous()
            rc.K2j = self._k2_over_j()[0].contiguous()
            rc.M = (K_mm @ K_inv @ K_mm)
            rc.M = (0.5 * (rc.M + rc.M.T)).contiguous()
            rc.mlogj = rc.m * float(torch.log(torch.tensor(self.jitter, dtype=F64)))
            rc.logdet_K_f = float(logdet_K)
            self._rc_obj = rc
        return self._rc_obj

    # ---- the reference's per-latent-dimension API (svgp.py:43-104) -------------------------
    def kernel_matrix(self, x, y, diag_only=False):
        if diag_only:
            return torch.ones(x.shape[0], dtype=F64, device=x.device)
        return self.kernel(x, y)

    def approximate_posterior_params(self, index_points_test, index_points_train, y, noise):
        bc_tr = self.batch_constants(index_points_train)
        same = index_points_test is index_points_train
        bc_te = bc_tr if same else self.batch_constants(index_points_test)
        K_mm = self._run_constants()[0]
        p_m, p_v, (S_inv, St, _) = self.posterior(bc_tr, y.reshape(-1, 1), noise.reshape(-1, 1), None if same else bc_te)
        mu_hat = bc_tr.c * (St @ K_mm)[0]
        A_hat = K_mm @ S_inv[0] @ K_mm
        return p_m[:, 0], p_v[:, 0], mu_hat, A_hat

    def variational_loss(self, x, y, noise, mu_hat, A_hat):
        bc = self.batch_constants(x)
        y, noise = y.to(F64).reshape(-1, 1), noise.to(F64).reshape(-1, 1)
        K_mm, K_inv, logdet_K, eye = self._run_constants()
        m = K_mm.shape[0]
        mv = (bc.Q @ mu_hat).reshape(-1, 1)
        tr = rowdot((bc.Q @ A_hat).unsqueeze(0), bc.Q).T
        logdet_S = spd_inverse_logdet((A_hat + self.jitter * eye).unsqueeze(0))[1][0]
        kl = 0.5 * (logdet_K - logdet_S - m + torch.trace(K_inv @ A_hat) + torch.sum(mu_hat * (K_inv @ mu_hat)))
        zero = torch.zeros_like(y)
        l3, _ = elbo_reduce(y, noise, mv, tr, zero, zero, bc.ktilde)
        return l3, kl
