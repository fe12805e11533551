"""Device-resident unbalanced-OT solver: thin Python handle over include/spadot_ot.h part B.

Replaces the per-stage numpy/ctypes loop of the reference
(/root/reference/SpaDOT/utils/OT_loss/ot_solvers.py:164-449) with one C call that keeps the
cost matrix, the kernel matrix and the scalings in HBM for all six epsilon stages.
torch is used only to own device memory and the stream.
"""
import ctypes

import numpy as np
import torch

from ._lib import OTConfig, OTInfo, OTSmallInfo, OTSmallProblem, ot_lib

F64, F32 = 0, 1
_TORCH_DT = {F64: torch.float64, F32: torch.float32}

# keys of ot_config the solver reads (config.yaml:39-57 / ot_solvers.py:164-179)
CONFIG_KEYS = ("lambda1", "lambda2", "epsilon", "epsilon0", "tolerance", "tau", "batch_size", "max_iter")


def make_config(cfg):
    c = OTConfig()
    c.lambda1 = float(cfg["lambda1"]); c.lambda2 = float(cfg["lambda2"])
    c.epsilon = float(cfg["epsilon"]); c.epsilon0 = float(cfg["epsilon0"])
    c.tolerance = float(cfg["tolerance"]); c.tau = float(cfg["tau"])
    c.batch_size = int(cfg["batch_size"]); c.max_iter = int(cfg["max_iter"])
    return c


class OTSolver:
    """One I x J problem resident on `device`.  storage: 'f64' (reference arithmetic) or 'f32'
    (fp32 cost/kernel matrices in HBM, fp64 scalings and accumulations)."""

    def __init__(self, I, J, storage="f64", device="cuda:0", stream=None):
        self.lib = ot_lib()
        self.I, self.J = int(I), int(J)
        self.storage = {"f64": F64, "f32": F32}[storage]
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("OTSolver needs a HIP device (torch device type 'cuda'); there is no CPU path")
        torch.cuda.set_device(self.device)
        self._stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        h = ctypes.c_void_p()
        rc = self.lib.spadot_ot_create(ctypes.byref(h), self.I, self.J, self.storage,
                                       ctypes.c_void_p(self._stream.cuda_stream))
        if rc != 0 or not h:
            raise RuntimeError(f"spadot_ot_create({I}, {J}) failed with {rc}")
        self.h = h
        self.ld = self.lib.spadot_ot_ld(self.h)
        self.info = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.spadot_ot_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- cost ----
    def set_cost(self, C):
        """C: (I, J) numpy array or torch tensor (any device); row-major."""
        if isinstance(C, np.ndarray):
            C = np.ascontiguousarray(C, dtype=np.float64)
            assert C.shape == (self.I, self.J)
            rc = self.lib.spadot_ot_set_cost_host(self.h, C.ctypes.data_as(ctypes.c_void_p))
        else:
            C = C.to(self.device)
            if C.dtype not in (torch.float64, torch.float32):
                C = C.to(torch.float64)
            C = C.contiguous()
            assert tuple(C.shape) == (self.I, self.J)
            dt = F64 if C.dtype == torch.float64 else F32
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
            rc = self.lib.spadot_ot_set_cost_dev(self.h, ctypes.c_void_p(C.data_ptr()), dt, self.J)
            self._keep = C  # alive until the conversion kernel has run
        if rc != 0:
            raise RuntimeError(f"set_cost failed with {rc}")

    def set_cost_from_latents(self, x, y, divide_by_median=True):
        """C = sqeuclidean(x, y) [/ median] computed on the device (ot_solvers.py:101-103)."""
        x = torch.as_tensor(x).to(self.device, torch.float64).contiguous()
        y = torch.as_tensor(y).to(self.device, torch.float64).contiguous()
        assert x.shape[0] == self.I and y.shape[0] == self.J and x.shape[1] == y.shape[1]
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        rc = self.lib.spadot_ot_set_cost_from_latents_dev(
            self.h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), int(x.shape[1]),
            1 if divide_by_median else 0)
        if rc != 0:
            raise RuntimeError(f"set_cost_from_latents failed with {rc}")

    # ---- solve ----
    def solve(self, cfg, G=None):
        """One whole solve.  Returns the OTInfo; raises like ot_solvers.py:446-447 on a NaN gap."""
        c = make_config(cfg)
        info = OTInfo()
        gp = None
        if G is not None:
            g = np.ascontiguousarray(np.asarray(G, dtype=np.float64))
            assert g.shape == (self.I,)
            gp = g.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.spadot_ot_solve(self.h, gp, ctypes.byref(c), ctypes.byref(info))
        if rc < 0:
            raise RuntimeError(f"spadot_ot_solve failed with {rc}")
        self.info = info
        if rc == 1:
            raise RuntimeError("Overflow encountered in duality gap computation, please report this incident")
        return info

    def plan(self, out="numpy", dtype=None):
        """Transport plan R/J.  out='numpy' -> fp64 ndarray (what the reference returns);
        out='torch' -> device tensor in `dtype` (default: storage dtype), no host copy."""
        if out == "numpy":
            P = np.empty((self.I, self.J), dtype=np.float64)
            rc = self.lib.spadot_ot_plan_host(self.h, P.ctypes.data_as(ctypes.c_void_p))
            if rc != 0:
                raise RuntimeError(f"plan_host failed with {rc}")
            return P
        dt = self.storage if dtype is None else {torch.float64: F64, torch.float32: F32}[dtype]
        P = torch.empty((self.I, self.J), dtype=_TORCH_DT[dt], device=self.device)
        rc = self.lib.spadot_ot_plan_dev(self.h, ctypes.c_void_p(P.data_ptr()), dt, self.J)
        if rc != 0:
            raise RuntimeError(f"plan_dev failed with {rc}")
        torch.cuda.current_stream(self.device).wait_stream(self._stream)
        return P

    def transition_table(self, row_labels, col_labels, n_row_groups=None, n_col_groups=None):
        """Cluster-by-cluster block sums of the plan (what wot's transition_table reads off a spot-level
        transport map, _analyze_utils.py:131-137) computed on the device without materialising the plan.
        Returns a torch fp64 tensor [n_row_groups, n_col_groups]."""
        rl = torch.as_tensor(row_labels, dtype=torch.int64, device=self.device)
        cl = torch.as_tensor(col_labels, dtype=torch.int32, device=self.device).contiguous()
        assert rl.numel() == self.I and cl.numel() == self.J
        Ka = int(n_row_groups) if n_row_groups is not None else int(rl.max()) + 1
        Kb = int(n_col_groups) if n_col_groups is not None else int(cl.max()) + 1
        Q = torch.empty((self.I, Kb), dtype=torch.float64, device=self.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        rc = self.lib.spadot_ot_plan_group_sums_dev(self.h, ctypes.c_void_p(cl.data_ptr()), Kb, ctypes.c_void_p(Q.data_ptr()))
        if rc != 0:
            raise RuntimeError(f"plan_group_sums failed with {rc}")
        torch.cuda.current_stream(self.device).wait_stream(self._stream)
        onehot = torch.zeros((Ka, self.I), dtype=torch.float64, device=self.device)
        onehot[rl, torch.arange(self.I, device=self.device)] = 1.0
        return onehot @ Q

    def plan_rowsums(self):
        r = np.empty(self.I, dtype=np.float64)
        rc = self.lib.spadot_ot_plan_rowsums_host(self.h, r.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(f"plan_rowsums failed with {rc}")
        return r

    def vector(self, name):
        """Copy of a state vector ('a','b','u','v','old_a','old_b') as numpy fp64."""
        idx = {"a": 0, "b": 1, "u": 2, "v": 3, "old_a": 4, "old_b": 5}[name]
        n = self.I if name in ("a", "u", "old_a") else self.J
        out = np.empty(n, dtype=np.float64)
        rc = self.lib.spadot_ot_vector_host(self.h, idx, out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(f"vector_host failed with {rc}")
        return out

    def matrix(self, name):
        """Copy of the cost ('C') or kernel ('K') matrix as an (I, J) numpy fp64 array."""
        out = np.empty((self.I, self.J), dtype=np.float64)
        rc = self.lib.spadot_ot_matrix_host(self.h, {"C": 0, "K": 1}[name], out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError(f"matrix_host failed with {rc}")
        return out

    def run_iterations(self, cfg, eps_stage, iters, timed=True):
        """Benchmark hook: `iters` scaling iterations, no checks.  Returns HIP-event ms (or None)."""
        c = make_config(cfg)
        ms = ctypes.c_float(0.0)
        rc = self.lib.spadot_ot_run_iterations(self.h, ctypes.byref(c), float(eps_stage), int(iters),
                                               ctypes.byref(ms) if timed else None)
        if rc == 2:
            raise RuntimeError("a scaling exceeded tau during run_iterations: not a steady-state timing run")
        if rc != 0:
            raise RuntimeError(f"run_iterations failed with {rc}")
        return ms.value if timed else None

    def tau_flag(self, reset=True):
        """Whether a scaling exceeded tau in the untimed run_iterations calls since the flag was last cleared (one 4-byte
        read-back + stream sync)."""
        rc = self.lib.spadot_ot_run_tau_flag(self.h, 1 if reset else 0)
        if rc < 0:
            raise RuntimeError(f"run_tau_flag failed with {rc}")
        return bool(rc)

    def run_checked(self, cfg, eps_stage, nbatches, last_stage=True):
        """The solver's real inner loop (with convergence measure + sync) `nbatches` times.
        Returns (iterations run, milliseconds)."""
        c = make_config(cfg)
        it, ms = ctypes.c_int(0), ctypes.c_float(0.0)
        rc = self.lib.spadot_ot_run_checked(self.h, ctypes.byref(c), float(eps_stage), int(last_stage), int(nbatches),
                                            ctypes.byref(it), ctypes.byref(ms))
        if rc != 0:
            raise RuntimeError(f"run_checked failed with {rc}")
        return it.value, ms.value

    def time_kernels(self, cfg, eps_stage, reps=20):
        """Average HIP-event milliseconds per launch of each kernel of one scaling iteration."""
        c = make_config(cfg)
        ms = (ctypes.c_float * 6)()
        rc = self.lib.spadot_ot_time_kernels(self.h, ctypes.byref(c), float(eps_stage), int(reps), ms)
        if rc != 0:
            raise RuntimeError(f"time_kernels failed with {rc}")
        return {"row_pass": ms[0], "col_pass": ms[1], "col_fin": ms[2], "absorb_idle": ms[3],
                "fused_pass": ms[4], "fused_col_fin": ms[5]}

    def fused_geometry(self):
        g = (ctypes.c_int * 4)()
        self.lib.spadot_ot_fused_geometry(self.h, g)
        return {"vpt": g[0], "rows_per_group": g[1], "workgroups": g[2], "rows_per_workgroup": g[3]}


# ------------------------------------------------------------------------------ small problems (include/spadot_ot.h part C)

SMALL_MAX = 64          # spadot_ot_small_max(): largest I, J of the one-wavefront solver
SMALL_MAX_D = 32


def small_problem_ok(I, J, d=None):
    """True when an I x J problem (latent dimension d) can take the single-launch small solver."""
    return 1 <= int(I) <= SMALL_MAX and 1 <= int(J) <= SMALL_MAX and (d is None or 1 <= int(d) <= SMALL_MAX_D)


class SmallSolveResult:
    """plans: list of numpy fp64 (I, J) arrays (R / J, ot_solvers.py:449); infos: list of OTSmallInfo."""

    def __init__(self, plans, infos):
        self.plans, self.infos = plans, infos


def solve_small(cfg, pairs=None, costs=None, growth=None, divide_by_median=True, gamma_out=None, device="cuda:0",
                fetch=True):
    """Whole solves of a batch of small problems in ONE launch (csrc/ot_small.hip): what _update_OT_matrix needs for
    its T - 1 pairs of 10 x 10 K-means centres (_train_utils.py:309-321).

    pairs: list of (x [I, d], y [J, d]) latents (numpy or torch, any device) -> cost = sqeuclidean [/ median]
           (ot_solvers.py:101-103), or
    costs: list of (I, J) cost matrices (numpy or torch) used as they are;
    growth: optional list of length-I vectors (None entries = ones);
    gamma_out: optional list of contiguous fp32 device tensors (I, J) (None entries allowed) that receive the
           row-normalised plan with NaN / inf -> 0 (_train_utils.py:299-300), written in place by the kernel;
    fetch=False enqueues the launch and returns None without any host synchronisation (the results are then only the
           gamma_out tensors)."""
    lib = ot_lib()
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("solve_small needs a HIP device (torch device type 'cuda'); there is no CPU path")
    items = pairs if pairs is not None else costs
    n = len(items)
    if n == 0:
        return SmallSolveResult([], [])
    c = make_config(cfg)

    # everything that is still on the host goes up in ONE copy
    host_parts, slots = [], []

    def stage(arr, k, field):
        if isinstance(arr, torch.Tensor) and arr.is_cuda:
            t = arr.to(device=device, dtype=torch.float64).contiguous()
            slots.append((k, field, t, None))
        else:
            a = np.ascontiguousarray(arr.detach().cpu().numpy() if isinstance(arr, torch.Tensor) else arr, dtype=np.float64)
            slots.append((k, field, None, (sum(p.size for p in host_parts), a.size)))
            host_parts.append(a.reshape(-1))

    shapes, d = [], 0
    for k, it in enumerate(items):
        if pairs is not None:
            x, y = it
            I, J, dk = int(x.shape[0]), int(y.shape[0]), int(x.shape[1])
            if int(y.shape[1]) != dk or (d and dk != d):
                raise ValueError("all latents of a batch must share their dimension")
            d = dk
            stage(x, k, "x_dev"); stage(y, k, "y_dev")
        else:
            I, J = int(it.shape[0]), int(it.shape[1])
            stage(it, k, "C_dev")
        if not small_problem_ok(I, J, d or None):
            raise ValueError(f"problem {k} ({I} x {J}, d = {d}) is outside the small solver's range")
        shapes.append((I, J))
        if growth is not None and growth[k] is not None:
            g = growth[k]
            if int(np.prod(g.shape)) != I:
                raise ValueError("growth vector of the wrong length")
            stage(g, k, "G_dev")
    up = torch.as_tensor(np.concatenate(host_parts)).to(device) if host_parts else None
    keep = [up]
    total = sum(I * J for I, J in shapes)
    plan_buf = torch.empty(total, dtype=torch.float64, device=device) if fetch else None
    info_buf = torch.zeros(n * ctypes.sizeof(OTSmallInfo), dtype=torch.uint8, device=device) if fetch else None
    probs = (OTSmallProblem * n)()
    off = 0
    for k, (I, J) in enumerate(shapes):
        probs[k].I, probs[k].J = I, J
        if plan_buf is not None:
            probs[k].plan_dev = plan_buf.data_ptr() + 8 * off
        off += I * J
        if gamma_out is not None and gamma_out[k] is not None:
            g = gamma_out[k]
            if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == (I, J)):
                raise ValueError("gamma_out entries must be contiguous fp32 device tensors of the plan's shape")
            probs[k].gamma_rownorm_dev = g.data_ptr()
    for k, field, t, span in slots:
        if t is not None:
            keep.append(t)
            setattr(probs[k], field, t.data_ptr())
        else:
            setattr(probs[k], field, up.data_ptr() + 8 * span[0])
    stream = torch.cuda.current_stream(device)
    rc = lib.spadot_ot_small_solve(n, probs, d, 1 if divide_by_median else 0, ctypes.byref(c),
                                   ctypes.c_void_p(info_buf.data_ptr()) if info_buf is not None else None,
                                   ctypes.c_void_p(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(f"spadot_ot_small_solve failed with {rc}")
    for t in keep:                       # inputs stay alive until the kernel has read them
        if t is not None:
            t.record_stream(stream)
    if not fetch:
        return None
    flat = plan_buf.cpu().numpy()        # (synchronises)
    raw = info_buf.cpu().numpy().tobytes()
    plans, infos, off = [], [], 0
    for k, (I, J) in enumerate(shapes):
        plans.append(flat[off:off + I * J].reshape(I, J).copy())
        off += I * J
        infos.append(OTSmallInfo.from_buffer_copy(raw, k * ctypes.sizeof(OTSmallInfo)))
    return SmallSolveResult(plans, infos)
