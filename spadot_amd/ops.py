"""Autograd bindings of the hand-written HIP kernels in libspadot_model.so (include/spadot_model.h).

torch owns device memory, streams and the autograd tape; every numeric op below is a HIP kernel
reached through the C-ABI with raw device pointers.  There is no CPU implementation: calling any of
these with CPU tensors, or without the built library, raises.
"""
import ctypes

import torch

from ._lib import model_lib

DT_F32, DT_BF16, DT_F64 = 0, 1, 2
_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float64: DT_F64}
KERNEL_KINDS = {"Gaussian": 0, "Cauchy": 1, "Quadratic": 2}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("spadot_amd ops run on the MI355X only (got a CPU tensor); there is no CPU path")


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with {rc}")


# ----------------------------------------------------------------------------- graph container

class BatchGraph:
    """CSR (by target) + transposed CSR (by source) of one induced batch graph, on the device.
    Built once per batch by spadot_amd.graph.build_batch_graph; self loops are already in the GATConv
    convention (exactly one per node)."""

    def __init__(self, n, rowptr, col, rowptr_t, col_t, eid_t, n_tgt=None):
        self.n = int(n)                                   # source nodes (rows of h)
        self.n_tgt = self.n if n_tgt is None else int(n_tgt)   # target nodes = the first n_tgt (rows of the output)
        self.rowptr, self.col = rowptr, col
        self.rowptr_t, self.col_t, self.eid_t = rowptr_t, col_t, eid_t
        self.E = int(col.numel())
        assert rowptr.numel() == self.n_tgt + 1 and rowptr_t.numel() == self.n + 1
        self.seed_graph = None      # optional: the same graph restricted to the seeds as targets (last GAT layer)
        self.layer_graphs = None    # optional: (g2, g3) of graph.build_batch_graph(tiers=...)
        self.plan_t = None          # optional: graph.BlockPlan by target (matrix-core forward / target-side backward)
        self.plan_s = None          # optional: graph.BlockPlan by source (matrix-core source-side backward)

    def to(self, device):
        g = BatchGraph(self.n, *(t.to(device) for t in (self.rowptr, self.col, self.rowptr_t, self.col_t, self.eid_t)),
                       n_tgt=self.n_tgt)
        g.seed_graph = self.seed_graph.to(device) if self.seed_graph is not None else None
        g.layer_graphs = tuple(t.to(device) for t in self.layer_graphs) if self.layer_graphs is not None else None
        g.plan_t = self.plan_t.to(device) if self.plan_t is not None else None
        g.plan_s = self.plan_s.to(device) if self.plan_s is not None else None
        return g


# ----------------------------------------------------------------------------- GAT edge phase

_gat_scratch = {}


def _gat_att_scratch(device, floats):
    key = str(device)
    cur = _gat_scratch.get(key)
    if cur is None or cur.numel() < floats:
        cur = torch.empty(floats, dtype=torch.float32, device=device)
        _gat_scratch[key] = cur
    return cur


GAT_MFMA = [True]      # test hook -- [False]: the per-edge kernels everywhere (tests compare the two paths)


def _mfma_plans(h, graph, H, C, concat):
    """The graph's block plans if the matrix-core edge kernels take this layer: bf16 rows, C = 512, H in {1, 2, 4, 8},
    head concat, plans attached (graph.attach_plans) and every block's column list within the kernels' LDS budget."""
    if not GAT_MFMA[0] or not concat or h.dtype != torch.bfloat16:
        return None
    pt, ps = getattr(graph, "plan_t", None), getattr(graph, "plan_s", None)
    if pt is None or ps is None:
        return None
    lib = model_lib()
    if not (lib.spadot_gat_mfma_supported(DT_BF16, H, C, pt.max_cols) and lib.spadot_gat_mfma_supported(DT_BF16, H, C, ps.max_cols)):
        return None
    return pt, ps


class _GATEdgeMFMA(torch.autograd.Function):
    """gat_edge on the matrix cores (csrc/gat_mfma.hip): logits, softmax per (target, head) written into the plan's dense
    weight image, then per block of 32 targets one dense product weights . h_rows with every distinct source row
    fetched once.  Backward: the mirror product for d(alpha) (k_gat_edot), softmax backward, and weights^T . g_pre on
    the transposed plan."""

    @staticmethod
    def forward(ctx, h, att_src, att_dst, bias, graph, H, C, act, plans, defer=False, act_cell=None):
        _need_cuda(h, att_src, att_dst, bias)
        lib = model_lib()
        pt, ps = plans
        h = h.contiguous()
        a_s = att_src.reshape(H, C).contiguous().float()
        a_d = att_dst.reshape(H, C).contiguous().float()
        bias_f = bias.contiguous().float()
        n, nt = graph.n, graph.n_tgt
        # h may carry rows past the last node (row padding, see _cache_batch_inputs): every kernel indexes rows by node id
        assert h.shape[0] >= n and h.shape[1] == H * C, (h.shape, n, H, C)
        dev = h.device
        s_src = torch.empty((n, H), dtype=torch.float32, device=dev)
        s_dst = torch.empty((n, H), dtype=torch.float32, device=dev)
        _check(lib.spadot_gat_logits(_p(h), DT_BF16, _p(a_s), _p(a_d), n, H, C, _p(s_src), _p(s_dst), _stream()), "spadot_gat_logits")
        alpha = torch.empty((graph.E, H), dtype=torch.float32, device=dev)
        img = pt.weight_image(H)
        # a layer that will be differentiated: the same weights also into the by-SOURCE plan's image, which the backward
        # product alpha^T g_pre reads (gradient-independent: written here, beside the SVGP branch, not on the backward chain)
        ctx.owner = att_src.data_ptr()         # (two layers that share a graph keep their own backward image and partials)
        img_s = ps.weight_image(H, owner=ctx.owner) if any(ctx.needs_input_grad[:4]) else None
        _check(lib.spadot_gat_alpha(_p(s_src), _p(s_dst), _p(graph.rowptr), _p(graph.col), _p(pt.cellq), nt, H, _p(alpha), _p(img),
                                    _p(ps.cellq) if img_s is not None else None, _p(img_s), _stream()), "spadot_gat_alpha")
        ctx.img_s = img_s
        # a padded input whose nodes are all targets (the first layer) hands its padding on: the next dense map then also
        # sees a row count that is a multiple of 128; the kernel writes the pad rows as zeros
        out_rows = h.shape[0] if (nt == n and h.shape[0] - n <= 4096) else nt
        out = torch.empty((out_rows, H * C), dtype=h.dtype, device=dev)
        _check(lib.spadot_gat_aggregate(_p(h), DT_BF16, _p(img), _p(pt.rows), _p(pt.sptr), _p(pt.cols), pt.nb, pt.max_cols, H, C, 0,
                                        _p(bias_f), None, int(act), None, None, _p(out), None, None, 0, nt, out_rows,
                                        None, None, None, None, _stream()), "spadot_gat_aggregate")
        ctx.save_for_backward(h, s_src, s_dst, out, alpha, a_s, a_d)
        ctx.graph, ctx.H, ctx.C, ctx.act, ctx.plans = graph, H, C, act, plans
        ctx.bias_dtype, ctx.att_shape, ctx.att_dtype = bias.dtype, att_src.shape, att_src.dtype
        ctx.flat3, ctx.defer = _adjacent_flat_grads((att_src, att_dst, bias), H * C), bool(defer)
        # act_cell (a dict shared with the op that consumes `out`): armed here, it tells that consumer that it may hand the
        # gradient back ALREADY multiplied by this layer's LeakyReLU'(out) -- it holds `out` as its input -- and say so by setting
        # cell["premasked"] in its backward (ops._DenseCD, ops._GATTail); backward() below then skips its own masking
        ctx.cell = act_cell if (act and act_cell is not None) else None
        if ctx.cell is not None:
            ctx.cell["armed"], ctx.cell["rows"], ctx.cell["slope"] = True, out_rows, ACT_SLOPE
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = model_lib()
        h, s_src, s_dst, out, alpha, a_s, a_d = ctx.saved_tensors
        graph, H, C = ctx.graph, ctx.H, ctx.C
        pt, ps = ctx.plans
        n, nt = graph.n, graph.n_tgt
        dev = h.device
        g_out = g_out.contiguous().to(h.dtype)
        # a gradient that arrives already multiplied by LeakyReLU'(out) (the consumer of `out` did it as it produced the gradient):
        # no read of `out`, no masked copy -- the source-side product reads g_out itself
        premasked = ctx.cell is not None and bool(ctx.cell.pop("premasked", False))
        act_here = 0 if premasked else int(ctx.act)
        g_pre = g_out if premasked else torch.empty((nt, H * C), dtype=h.dtype, device=dev)
        dz = torch.empty((graph.E, H), dtype=torch.float32, device=dev)
        ds_dst = torch.empty((n, H), dtype=torch.float32, device=dev)      # (rows >= nt zeroed by the softmax kernel)
        # attention-vector and bias gradients: per-block partial sums left by the two matrix-core kernels (the bias columns by
        # k_gat_edot, which holds the block's g_pre rows in LDS; the att_src / att_dst columns by k_gat_agg<1>, which reads
        # the block's 32 rows of h for it), then ONE fixed-order column sum over the blocks -- instead of k_gat_datt_part's
        # pass of its own over h and g_pre (25 us per layer).  The partial matrix belongs to the batch graph: rows a plan
        # does not have stay zero.
        W3 = 3 * H * C
        R = max(pt.nb, ps.nb)
        bufs = graph.__dict__.setdefault("_bwd_bufs", {})
        part = bufs.get(("attpart", H, C, ctx.owner))
        if part is None or part.device != dev or part.shape != (R, W3):
            part = bufs[("attpart", H, C, ctx.owner)] = torch.zeros((R, W3), dtype=torch.float32, device=dev)
        _check(lib.spadot_gat_edge_dot(_p(g_out), _p(out), _p(h), DT_BF16, _p(pt.rows), _p(pt.sptr), _p(pt.cols), _p(pt.cell),
                                       pt.nb, pt.max_cols, H, C, act_here, None if premasked else _p(g_pre), _p(dz), _p(part), W3,
                                       2 * H * C, _stream()), "spadot_gat_edge_dot")
        img = ctx.img_s            # (alpha in the by-source plan's layout: written by the forward pass)
        _check(lib.spadot_gat_softmax_backward(_p(alpha), _p(s_src), _p(s_dst), _p(graph.rowptr), _p(graph.col), None, nt, n,
                                               H, _p(dz), _p(ds_dst), None, _stream()), "spadot_gat_softmax_backward")
        # ds_src[j] = sum of dz over the outgoing edges of j: taken by the source-side product's workgroups for their own rows
        dh = torch.empty_like(h)                  # (h's pad rows, if any, get a zero gradient: written by the kernel)
        pad_ok = h.shape[0] - n <= 4096
        if not pad_ok:
            dh[n:].zero_()
        _check(lib.spadot_gat_aggregate(_p(g_pre), DT_BF16, _p(img), _p(ps.rows), _p(ps.sptr), _p(ps.cols), ps.nb, ps.max_cols, H, C, 1,
                                        _p(a_s), _p(a_d), 0, None, _p(ds_dst), _p(dh), _p(h), _p(part), W3, n,
                                        h.shape[0] if pad_ok else n, _p(dz), _p(graph.rowptr_t), _p(graph.eid_t), None, _stream()),
               "spadot_gat_aggregate")
        if ctx.flat3 is not None and _DIRECT_GRAD[0]:
            # the three gradients lie back to back in the optimizer's flat gradient buffer: the column sum over the blocks
            # writes them THERE (no copy launch behind it), and -- nothing in the backward pass reads them -- not HERE, on the
            # chain in front of this layer's dense map (in the profiled step the first layer's column sum waited 79 us for a
            # compute unit beside the side stream's weight-gradient GEMM, with the step's last GEMM queued behind it): queued
            # for the side stream where the layer's inputs are final when that queue runs (`defer`, ops.DEFERRED), else run
            # at the END of this backward stage (ops.POST_CHAIN: behind the last GEMM, same stream)
            fa, fd, fb_ = ctx.flat3
            nrow = part.shape[0]
            job = lambda: _check(lib.spadot_colsum(_p(part), nrow, W3, _p(fa), _stream()), "spadot_colsum")
            if ctx.defer and _deferring():
                DEFERRED[0].append(job)
            elif POST_CHAIN[0] is not None:
                POST_CHAIN[0].append(job)
            else:
                job()
            return dh, fa, fd, fb_, None, None, None, None, None, None, None
        datt = torch.empty((3, H * C), dtype=torch.float32, device=dev)
        _check(lib.spadot_colsum(_p(part), part.shape[0], W3, _p(datt), _stream()), "spadot_colsum")
        return (dh, datt[0].view(ctx.att_shape).to(ctx.att_dtype), datt[1].view(ctx.att_shape).to(ctx.att_dtype),
                datt[2].to(ctx.bias_dtype), None, None, None, None, None, None, None)


ACT_SLOPE = 0.01        # F.leaky_relu's default slope (encoder.py:56-57): the activation fused into gat_edge(act=True)


def _adjacent_flat_grads(params, width):
    """The .grad views of `params` (each `width` fp32 numbers) if they lie back to back in memory -- consecutive segments of
    a FlatAdamW gradient buffer -- so that ONE kernel can write all of them through the first one's address; else None."""
    gs = [getattr(p, "grad", None) for p in params]
    if any(g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != width or g.shape != p.shape
           for g, p in zip(gs, params)):
        return None
    base = gs[0].data_ptr()
    if any(g.data_ptr() != base + 4 * width * k for k, g in enumerate(gs)):
        return None
    return tuple(gs)


class _GATEdge(torch.autograd.Function):
    """One GATConv layer after its dense map: logits s = h . att (k_gat_logits), scatter-softmax over incoming
    edges + weighted scatter-add + bias + activation + head concat/mean (k_gat_fwd); backward in three
    launches + a reduction for the attention vectors."""

    @staticmethod
    def forward(ctx, h, att_src, att_dst, bias, graph, H, C, concat, act):
        _need_cuda(h, att_src, att_dst, bias)
        lib = model_lib()
        h = h.contiguous()
        a_s = att_src.reshape(H, C).contiguous().float()
        a_d = att_dst.reshape(H, C).contiguous().float()
        bias_f = bias.contiguous().float()
        n, nt = graph.n, graph.n_tgt        # targets are the first nt nodes (nt < n: seeds-only last layer)
        assert h.shape[0] >= n and h.shape[1] == H * C, (h.shape, n, H, C)      # (rows past n: padding, never read)
        s_src = torch.empty((n, H), dtype=torch.float32, device=h.device)
        s_dst = torch.empty((n, H), dtype=torch.float32, device=h.device)
        _check(lib.spadot_gat_logits(_p(h), _DT[h.dtype], _p(a_s), _p(a_d), n, H, C, _p(s_src), _p(s_dst), _stream()),
               "spadot_gat_logits")
        out = torch.empty((nt, H * C if concat else C), dtype=h.dtype, device=h.device)
        alpha = torch.empty((graph.E, H), dtype=torch.float32, device=h.device)
        _check(lib.spadot_gat_forward(_p(h), _DT[h.dtype], _p(s_src), _p(s_dst), _p(graph.rowptr), _p(graph.col),
                                      _p(bias_f), nt, H, C, int(concat), int(act), _p(out), _p(alpha), _stream()),
               "spadot_gat_forward")
        ctx.save_for_backward(h, s_src, s_dst, out, alpha, a_s, a_d)
        ctx.graph, ctx.H, ctx.C, ctx.concat, ctx.act = graph, H, C, concat, act
        ctx.bias_dtype, ctx.att_shape, ctx.att_dtype = bias.dtype, att_src.shape, att_src.dtype
        return out

    @staticmethod
    def backward(ctx, g_out):
        lib = model_lib()
        h, s_src, s_dst, out, alpha, a_s, a_d = ctx.saved_tensors
        graph, H, C = ctx.graph, ctx.H, ctx.C
        n, nt = graph.n, graph.n_tgt
        g_out = g_out.contiguous().to(h.dtype)
        g_pre = torch.empty((nt, H * C), dtype=h.dtype, device=h.device)
        dz = torch.empty((graph.E, H), dtype=torch.float32, device=h.device)
        # rows >= nt are not targets: their destination-logit gradient is zero
        if nt == n:
            ds_dst = torch.empty((n, H), dtype=torch.float32, device=h.device)
        else:
            # kept with the graph: the kernel rewrites rows < nt every step, rows >= nt stay zero -- no fill launch per step
            key = ("ds_dst", H)
            bufs = graph.__dict__.setdefault("_bwd_bufs", {})
            ds_dst = bufs.get(key)
            if ds_dst is None or ds_dst.device != h.device:
                ds_dst = bufs[key] = torch.zeros((n, H), dtype=torch.float32, device=h.device)
        _check(lib.spadot_gat_backward_target(_p(g_out), _p(out), _p(h), _DT[h.dtype], _p(s_src), _p(s_dst), _p(alpha),
                                              _p(graph.rowptr), _p(graph.col), nt, H, C, int(ctx.concat), int(ctx.act),
                                              _p(g_pre), _p(dz), _p(ds_dst), _stream()), "spadot_gat_backward_target")
        dh = torch.empty_like(h)
        if h.shape[0] > n:
            dh[n:].zero_()                    # pad rows of the input: zero gradient
        ds_src = torch.empty((n, H), dtype=torch.float32, device=h.device)
        _check(lib.spadot_gat_backward_source(_p(g_pre), _DT[h.dtype], _p(alpha), _p(dz), _p(graph.rowptr_t),
                                              _p(graph.col_t), _p(graph.eid_t), n, H, C, _p(dh), _p(ds_src),
                                              _p(ds_dst), _p(a_s), _p(a_d), _stream()), "spadot_gat_backward_source")
        # attention-vector gradients and the bias gradient (column sums of g_pre) in one slab pass over the nodes
        datt = torch.empty((3, H * C), dtype=torch.float32, device=h.device)
        floats = 3 * H * C * max(1, min((n + 15) // 16, 1024))
        scratch = _gat_att_scratch(h.device, floats)
        _check(lib.spadot_gat_att_grad(_p(h), _DT[h.dtype], _p(ds_src), _p(ds_dst), n, H, C, _p(scratch), floats,
                                       _p(datt), ctypes.c_void_p(datt.data_ptr() + 4 * H * C), _p(g_pre), nt, _stream()),
               "spadot_gat_att_grad")
        dbias = datt[2] if ctx.concat else datt[2].view(H, C).sum(dim=0)
        return (dh, datt[0].view(ctx.att_shape).to(ctx.att_dtype), datt[1].view(ctx.att_shape).to(ctx.att_dtype),
                dbias.to(ctx.bias_dtype), None, None, None, None, None)


def gat_edge(h, att_src, att_dst, bias, graph, heads, channels, concat=True, act=False, defer=False, act_cell=None):
    """Everything of one GATConv layer after the dense map h = x W^T: attention logits, edge softmax,
    aggregation, bias, optional leaky_relu(0.01), head concat/mean.  att_src / att_dst: [1, H, C] parameters.
    defer: the layer's attention-vector / bias gradient sum may be queued in ops.DEFERRED (the caller runs that queue only
    when this layer's backward kernels have finished: GraphedStepper, the second layer).
    act_cell: a dict to share with the op that consumes the result (dense_cd / gat_tail `in_cell`): see _GATEdgeMFMA.forward."""
    plans = _mfma_plans(h, graph, heads, channels, concat) if h.is_cuda else None
    if plans is not None:
        return _GATEdgeMFMA.apply(h, att_src, att_dst, bias, graph, heads, channels, act, plans, defer, act_cell)
    return _GATEdge.apply(h, att_src, att_dst, bias, graph, heads, channels, concat, act)


# ----------------------------------------------------------------------------- last GAT layer for the seeds, aggregate-first

_BMM_OUT_DTYPE = [None]     # whether torch.bmm takes out_dtype= (probed on the first call)


def _bmm_f32(a, b, out=None):
    """fp32 [B, M, N] = a [B, M, K] . b [B, K, N] with fp32 accumulation AND fp32 result from bf16 operands (the weight
    gradient goes straight into the optimizer's flat gradient buffer, no bf16 round trip)."""
    if a.dtype == torch.float32:
        return torch.bmm(a, b, out=out) if out is not None else torch.bmm(a, b)
    if _BMM_OUT_DTYPE[0] is None:
        try:
            torch.bmm(a[:1, :1], b[:1, :, :1], out_dtype=torch.float32)
            _BMM_OUT_DTYPE[0] = True
        except (TypeError, RuntimeError):
            _BMM_OUT_DTYPE[0] = False
    if _BMM_OUT_DTYPE[0]:
        if out is not None:
            return torch.bmm(a, b, out_dtype=torch.float32, out=out)
        return torch.bmm(a, b, out_dtype=torch.float32)
    r = torch.bmm(a.float(), b.float())
    if out is not None:
        out.copy_(r)
        return out
    return r


def gat_tail_ok(x, W, graph, H, C, concat):
    """Whether the aggregate-first form (csrc/gat_tail.hip) takes this layer: head mean, far fewer targets than source rows,
    K = W.shape[1] <= 2048 input channels."""
    if not (not concat and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 2 and x.is_contiguous()):
        return False
    K = W.shape[1]
    return bool(graph.n_tgt * 4 <= graph.n and x.shape[0] >= graph.n and x.shape[1] >= K and x.shape[1] % 8 == 0 and C % 8 == 0
                and W.shape[0] == H * C and W.is_contiguous() and model_lib().spadot_gat_tail_supported(_DT[x.dtype], H, K))


class _GATTail(torch.autograd.Function):
    """GATConv(concat = False) for targets = the first n_tgt nodes, aggregate-first (include/spadot_model.h,
    spadot_gat_tail_*): logits from w = W_h^T att (no dense map over the source rows), A_h = sum_j alpha x_j for the
    targets, O_h = A_h W_h^T on n_tgt rows (one batched library GEMM), head mean + bias.  Backward: the mirror image --
    dA_h = g / H W_h and dW_h = (g / H)^T A_h as batched GEMMs on n_tgt rows, per-edge d alpha = <dA, x_j>, softmax backward,
    dx by the transposed CSR, d w = dS^T x as per-block partials + one fixed-order column sum, and the chain through
    w = W_h^T att folded into dW (rank-2 update per head) and datt = W_h d w."""

    @staticmethod
    def forward(ctx, x, W, wimg, att_src, att_dst, bias, graph, H, C, in_cell=None):
        _need_cuda(x, W, att_src, att_dst, bias)
        lib = model_lib()
        dt = _DT[x.dtype]
        n, nt, K, Kp = graph.n, graph.n_tgt, W.shape[1], x.shape[1]
        dev = x.device
        a_s = att_src.reshape(H * C).contiguous().float()
        a_d = att_dst.reshape(H * C).contiguous().float()
        Wd = W.detach()
        S = 32
        part = torch.empty(S * 2 * H * K, dtype=torch.float32, device=dev)
        wv = torch.empty((2 * H, K), dtype=torch.float32, device=dev)
        wsplit = torch.empty((2, 2 * H, K), dtype=torch.bfloat16, device=dev) if (x.dtype == torch.bfloat16 and K % 32 == 0) else None
        whi, wlo = (wsplit[0], wsplit[1]) if wsplit is not None else (None, None)
        _check(lib.spadot_gat_tail_wvec(_p(Wd), K, _p(a_s), _p(a_d), H, C, K, _p(part), S, _p(wv), _p(whi), _p(wlo), _stream()),
               "spadot_gat_tail_wvec")
        s = torch.empty((n, 2 * H), dtype=torch.float32, device=dev)
        _check(lib.spadot_gat_tail_logits(_p(x), dt, Kp, _p(wv), _p(whi), _p(wlo), n, H, K, _p(s), _stream()), "spadot_gat_tail_logits")
        A = torch.empty((H, nt, K), dtype=x.dtype, device=dev)
        alpha = torch.empty((graph.E, H), dtype=torch.float32, device=dev)
        _check(lib.spadot_gat_tail_aggregate(_p(x), dt, Kp, _p(s), _p(graph.rowptr), _p(graph.col), nt, H, K, _p(A), _p(alpha), _stream()),
               "spadot_gat_tail_aggregate")
        # O_h = A_h W_h^T on the n_tgt aggregated rows: [H, nt, K] x [H, K, C]  (W_h = rows h C .. of the weight (image))
        Wg = (wimg if x.dtype != torch.float32 else Wd).view(H, C, -1)[:, :, :K]
        O = torch.bmm(A, Wg.transpose(1, 2))
        out = torch.empty((nt, C), dtype=x.dtype, device=dev)
        _check(lib.spadot_gat_tail_headmean(_p(O), dt, _p(bias.detach().contiguous().float()), nt, H, C, _p(out), _stream()),
               "spadot_gat_tail_headmean")
        ctx.save_for_backward(x, Wd, Wg, a_s, a_d, wv, s, A, alpha)
        ctx.graph, ctx.H, ctx.C = graph, H, C
        ctx.bias_dtype, ctx.att_shape, ctx.att_dtype = bias.dtype, att_src.shape, att_src.dtype
        g = W.grad
        okv = lambda g_, p_: g_ if (g_ is not None and g_.dtype == torch.float32 and g_.is_contiguous() and g_.shape == p_.shape) else None
        ctx.wgrad = okv(g, W)
        ctx.agrad = (okv(att_src.grad, att_src), okv(att_dst.grad, att_dst))     # flat-gradient views (FlatAdamW)
        # in_cell: x is the activated output of a gat_edge(act=True, act_cell=in_cell) over exactly these rows: dx may leave
        # already multiplied by that activation's derivative (see _GATEdgeMFMA.forward)
        ctx.in_cell = in_cell if (in_cell is not None and in_cell.get("armed") and in_cell.get("rows") == x.shape[0]) else None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = model_lib()
        x, Wd, Wg, a_s, a_d, wv, s, A, alpha = ctx.saved_tensors
        graph, H, C = ctx.graph, ctx.H, ctx.C
        dt = _DT[x.dtype]
        n, nt, K, Kp = graph.n, graph.n_tgt, Wd.shape[1], x.shape[1]
        dev = x.device
        g = g.contiguous()
        dbias = torch.empty(C, dtype=torch.float32, device=dev)
        if g.dtype == x.dtype:             # bias gradient (column sums of g) and d O_h = g / H for every head: one pass over g
            gs = torch.empty_like(g)
            _check(lib.spadot_gat_tail_scale_colsum(_p(g), dt, nt, C, 1.0 / H, _p(gs), _p(dbias), _stream()), "spadot_gat_tail_scale_colsum")
        else:
            _check(lib.spadot_gat_tail_colsum_rows(_p(g), _DT[g.dtype], nt, C, _p(dbias), _stream()), "spadot_gat_tail_colsum_rows")
            gs = (g * (1.0 / H)).to(x.dtype)
        dA = torch.bmm(gs.unsqueeze(0).expand(H, nt, C), Wg)              # [H, nt, K]
        direct = ctx.wgrad is not None and _DIRECT_GRAD[0]
        dW = ctx.wgrad if direct else torch.empty((H * C, K), dtype=torch.float32, device=dev)
        # what only the optimizer reads -- dW_h = (g/H)^T A_h, d w = dS^T x, the chain through w = W_h^T att -- may be queued
        # (ops.DEFERRED) and written into the flat gradient later; the chain to dx runs now
        defer = direct and _deferring() and ctx.agrad[0] is not None and ctx.agrad[1] is not None
        if not defer:
            _bmm_f32(gs.t().unsqueeze(0).expand(H, C, nt), A, out=dW.view(H, C, K))
        dz = torch.empty((graph.E, H), dtype=torch.float32, device=dev)
        ds_dst = torch.empty((nt, H), dtype=torch.float32, device=dev)
        _check(lib.spadot_gat_tail_edge_backward(_p(x), dt, Kp, _p(dA), _p(s), _p(alpha), _p(graph.rowptr), _p(graph.col), nt, H, K,
                                                 _p(dz), _p(ds_dst), _stream()), "spadot_gat_tail_edge_backward")
        dx = torch.empty_like(x) if Kp == K else torch.zeros_like(x)      # (pad columns of the input: zero gradient)
        ds_src = torch.empty((n, H), dtype=torch.float32, device=dev)
        _check(lib.spadot_gat_tail_source_backward(_p(dA), dt, _p(alpha), _p(dz), _p(ds_dst), _p(wv), _p(graph.rowptr_t), _p(graph.col_t),
                                                   _p(graph.eid_t), n, nt, x.shape[0], H, K, _p(dx), Kp, _p(ds_src),
                                                   _p(x) if ctx.in_cell is not None else None, Kp,
                                                   float(ctx.in_cell["slope"]) if ctx.in_cell is not None else 1.0, _stream()),
               "spadot_gat_tail_source_backward")
        if ctx.in_cell is not None:
            ctx.in_cell["premasked"] = True
        R = int(lib.spadot_gat_tail_dwvec_rows(n))

        def weight_side(datt_src_ptr, datt_dst_ptr, with_dw):
            if with_dw:
                _bmm_f32(gs.t().unsqueeze(0).expand(H, C, nt), A, out=dW.view(H, C, K))
            part = torch.empty((R, 2 * H * K), dtype=torch.float32, device=dev)
            _check(lib.spadot_gat_tail_dwvec(_p(x), dt, Kp, _p(ds_src), _p(ds_dst), n, nt, H, K, _p(part), _stream()), "spadot_gat_tail_dwvec")
            dwv = torch.empty(2 * H * K, dtype=torch.float32, device=dev)
            _check(lib.spadot_colsum(_p(part), R, 2 * H * K, _p(dwv), _stream()), "spadot_colsum")
            _check(lib.spadot_gat_tail_wvec_backward(_p(Wd), K, _p(a_s), _p(a_d), _p(dwv), H, C, K, _p(dW), K, 1, datt_src_ptr, datt_dst_ptr,
                                                     _stream()), "spadot_gat_tail_wvec_backward")

        if defer:
            ga_s, ga_d = ctx.agrad
            DEFERRED[0].append(lambda: weight_side(_p(ga_s), _p(ga_d), True))
            return dx, dW, None, ga_s, ga_d, dbias.to(ctx.bias_dtype), None, None, None, None
        datt = torch.empty((2, H * C), dtype=torch.float32, device=dev)
        weight_side(_p(datt), ctypes.c_void_p(datt.data_ptr() + 4 * H * C), False)
        return (dx, dW, None, datt[0].view(ctx.att_shape).to(ctx.att_dtype), datt[1].view(ctx.att_shape).to(ctx.att_dtype),
                dbias.to(ctx.bias_dtype), None, None, None, None)


def gat_tail(x, W, wimg, att_src, att_dst, bias, graph, heads, channels, in_cell=None):
    """The whole GATConv(concat = False) layer -- dense map included -- for the first graph.n_tgt nodes as targets."""
    return _GATTail.apply(x, W, wimg, att_src, att_dst, bias, graph, heads, channels, in_cell)


# ----------------------------------------------------------------------------- dense maps in the compute dtype

_DIRECT_GRAD = [False]     # True only inside FlatAdamW.backward (a plain .backward() would ADD the returned view to itself)

# Off-chain gradient work (round 4).  A weight gradient -- dW of a dense map, the attention-vector chain of the last GAT layer --
# feeds nothing in the backward pass: only the optimizer reads it.  While DEFERRED[0] is a list, the backward functions below
# append that work as closures instead of launching it (and return the flat-gradient VIEW it will be written to); the caller
# (GraphedStepper, `defer_wgrad`) runs the closures later, on the side stream, beside the rest of the GAT backward -- whose
# dependency chain they would otherwise lengthen by their own duration.
DEFERRED = [None]
# Gradient work that feeds nothing in the backward pass and whose inputs are final only when its stage ends: run at the END of
# the stage that queued it, on the same stream (FlatAdamW.backward / backward_partial drain it behind autograd.grad).
POST_CHAIN = [None]
_UNIT_SEEDS = set()        # addresses of the constant-one tensors FlatAdamW.backward_partial seeds a scalar loss with
_UNIT_ONES = {}            # (device type, index) -> THE constant one of that device: never freed, so its address in
                           # _UNIT_SEEDS can never come to belong to another tensor


def unit_seed(device):
    """The process-wide constant-one fp32 scalar of `device`: the seed of every scalar-loss backward of FlatAdamW.  The loss
    kernels recognise it by address (no launch for d loss / d term); it lives as long as the process."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    one = _UNIT_ONES.get(key)
    if one is None:
        one = _UNIT_ONES[key] = torch.ones((), dtype=torch.float32, device=device)
        _UNIT_SEEDS.add(one.data_ptr())
    return one


def _deferring():
    return DEFERRED[0] is not None and _DIRECT_GRAD[0]


def cast_rows(pairs):
    """[(src fp32 [R, K] contiguous, dst bf16 [R, Kp >= K] contiguous), ...] -> dst[:, :K] = src, ONE launch for up to
    four matrices (csrc: k_cast_rows_multi) instead of one library cast launch each."""
    # (the kernel reads 16-byte and writes 8-byte vectors: a view at an odd storage offset takes the library copy below)
    ok = all(s_.dtype == torch.float32 and d_.dtype == torch.bfloat16 and s_.is_contiguous() and d_.is_contiguous()
             and s_.dim() == 2 and s_.shape[1] % 4 == 0 and d_.shape[1] % 4 == 0 and s_.is_cuda
             and s_.data_ptr() % 16 == 0 and d_.data_ptr() % 8 == 0 for s_, d_ in pairs)
    if not ok:
        for s_, d_ in pairs:
            d_[:, :s_.shape[1]].copy_(s_)
        return
    lib = model_lib()
    for k0 in range(0, len(pairs), 4):
        grp = pairs[k0:k0 + 4]
        n = len(grp)
        src = (ctypes.c_void_p * n)(*(s_.data_ptr() for s_, _ in grp))
        dst = (ctypes.c_void_p * n)(*(d_.data_ptr() for _, d_ in grp))
        rows = (ctypes.c_int * n)(*(s_.shape[0] for s_, _ in grp))
        K = (ctypes.c_int * n)(*(s_.shape[1] for s_, _ in grp))
        Kp = (ctypes.c_int * n)(*(d_.shape[1] for _, d_ in grp))
        _check(lib.spadot_cast_rows_multi(src, dst, rows, K, Kp, n, _stream()), "spadot_cast_rows_multi")


# csrc/gemm_bf16.hip under the GAT layers' dense maps.  Measured on one box (tools/gemm_bench.py, cfg3): alone it beats the
# library at the layer shapes (9980 x 2048 x 3072: 110 vs 119 us, x 2048: 80 vs 90 us); inside the step the FIRST layer's forward
# map loses in every form measured (its 256 workgroups hold every compute unit's registers and LDS for the whole GEMM while the
# latency-bound SVGP branch on the other stream, which bounds the forward pair, waits behind them; the library's ~800
# short-lived workgroups let that branch's small kernels in) and the third layer's is stream-K in the library (1.18 PFLOP/s);
# the SECOND layer's forward map (+1.1 % of the step, round 3) and the input gradients take it.  The numbers of the forms that
# lost are in DESIGN section 4 and profiles/r02-r04; the switches that selected them are gone (round 5).
GEMM_DGRAD_MIN_WGS = 240           # the input-gradient GEMM takes the own kernel from this many 320 x 256 tiles on
GEMM_MAX_WGS = 256                 # own GEMM only up to this many tiles (one round of the chip's compute units)
GEMM_OWN = [True]                  # test hook -- [False]: gemm_tn() and the input gradient go to the library
# compute units the OTHER stream holds while a forward map runs (the SVGP branch's inverse: one 512-thread workgroup per
# matrix, 2 L = 40 of them for ~0.25 ms): gemm_tn() then cuts the row panels beyond one round of the free units into slices
# of the contraction (spadot_gemm_tn_bf16_split).
GEMM_BUSY_CUS = 40
# (padded contraction width, minimum rows) of the forward maps that go through the own kernel: the second GAT layer at the
# benchmarked batch shape (~10^4 rows x 2048 -> 2048: 256 tiles, one round)
GEMM_FWD_SHAPES = ((2048, 9000),)


def _own_forward(M, Kp):
    return any(t[0] == Kp and M >= t[1] for t in GEMM_FWD_SHAPES)


def gemm_tn(x, w):
    """x [M, K] . w[N, K]^T -> [M, N], bf16 operands, fp32 accumulation: csrc/gemm_bf16.hip (one 320 x 256 tile per compute
    unit) where its shape conditions hold and the output fills at least half the chip's compute units with whole tiles,
    otherwise the library."""
    M, K = x.shape
    N = w.shape[0]
    if (GEMM_OWN[0] and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.is_contiguous()
            and w.is_contiguous() and N % 256 == 0 and K % 64 == 0 and w.shape[1] == K and M >= 2560 and N >= 1024
            and ((M + 319) // 320) * (N // 256) <= GEMM_MAX_WGS):
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
        lib = model_lib()
        # a grid of exactly one round that will share the chip (GEMM_BUSY_CUS compute units held by the other stream's
        # inverse): the row panels that would form a second round are cut into quarter tiles along the contraction
        mt, nt_ = (M + 319) // 320, N // 256
        tail = mt - (256 - GEMM_BUSY_CUS) // nt_ if GEMM_BUSY_CUS > 0 else 0
        if 0 < tail < mt and mt * nt_ > 256 - GEMM_BUSY_CUS and K >= 4 * 64:
            need = int(lib.spadot_gemm_bf16_split_workspace(M, N, tail, 4))
            if need > 0:
                ws = torch.empty(need, dtype=torch.float32, device=x.device)
                rc = lib.spadot_gemm_tn_bf16_split(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, tail, 4,
                                                   ws.data_ptr(), _stream())
                if rc == 0:
                    return out
                if rc != -22:
                    _check(rc, "spadot_gemm_tn_bf16_split")
        rc = lib.spadot_gemm_tn_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, _stream())
        if rc == 0:
            return out
        if rc != -22:
            _check(rc, "spadot_gemm_tn_bf16")
    return torch.nn.functional.linear(x, w)


WGRAD_MIN_WGS = 200          # same-box A/B: 2048 x 3000 (192 workgroups) is 0.4 % better on the library


_ZERO_ROWS = {}


def _zero_row(device):
    """512 bytes of zeros on `device` (what the weight-gradient kernel reads for rows past M): one per device, created on
    the first eager call; a first call under graph capture gets a row of its own that the capturing graph keeps."""
    key = (device.type, device.index)
    z = _ZERO_ROWS.get(key)
    if z is None:
        z = torch.zeros(256, dtype=torch.bfloat16, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _ZERO_ROWS[key] = z
    return z




def wgrad_plan(N, K, Kp):
    """(tile_k, slices, workgroups) of csrc/gemm_wgrad_bf16.hip for a [N x K] weight gradient, or None (library): the tile
    width whose grid fills more of the chip's 256 compute units in one round; the narrow tile only where it gains a fifth
    (it reuses each staged byte less)."""
    best = None
    for tk in (256, 192):
        kt = (K + tk - 1) // tk
        if Kp < kt * tk:
            continue
        tiles = (N // 256) * kt
        if tiles == 0 or tiles > 256:
            wgs, slices = tiles, 1
        else:
            slices = min(8, max(1, 256 // tiles))
            wgs = tiles * slices
        fill = wgs / (256.0 * ((wgs + 255) // 256)) if wgs else 0.0
        score = fill * (1.0 if tk == 256 else 0.85)
        if best is None or score > best[0]:
            best = (score, tk, slices, wgs)
    if best is None or best[3] < WGRAD_MIN_WGS:
        return None
    return best[1], best[2], best[3]


def wgrad_bf16(g, x, K, out=None):
    """g^T x[:, :K] -> fp32 [N, K] for g [M, N], x [M, Kp >= K] (bf16): csrc/gemm_wgrad_bf16.hip where its conditions hold
    (N % 256 == 0, the X rows readable up to the next multiple of the tile width), else the library."""
    M, N = g.shape
    Kp = x.shape[1]
    if (g.is_cuda and g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and g.is_contiguous()
            and x.is_contiguous() and N % 256 == 0 and Kp % 8 == 0 and M >= 1024 and N >= 1024 and K >= 1024):
        if out is None:
            out = torch.empty((N, K), dtype=torch.float32, device=g.device)
        plan = wgrad_plan(N, K, Kp) if (out.is_contiguous() and out.dtype == torch.float32 and out.shape == (N, K)) else None
        if plan is not None:
            tile_k, slices, _ = plan
            lib = model_lib()
            need = int(lib.spadot_gemm_wgrad_bf16_workspace_tiled(M, N, K, slices, tile_k))
            # the split-contraction partials belong to THIS call: under capture they come out of the capturing graph's
            # pool, so graphs replayed on different streams never share them and nothing a graph points at is ever freed
            ws = torch.empty(max(need, 4), dtype=torch.float32, device=g.device) if need >= 0 else None
            rc = -22 if ws is None else lib.spadot_gemm_wgrad_bf16_tiled(g.data_ptr(), N, x.data_ptr(), Kp, out.data_ptr(), K, M, N, K,
                                                                        slices, tile_k, ws.data_ptr(), _zero_row(g.device).data_ptr(),
                                                                        _stream())
            if rc == 0:
                return out
            if rc != -22:
                _check(rc, "spadot_gemm_wgrad_bf16_tiled")
    if out is not None:
        return torch.mm(g.t(), x[:, :K], out_dtype=torch.float32, out=out)
    return torch.mm(g.t(), x[:, :K], out_dtype=torch.float32)


class _DenseCD(torch.autograd.Function):
    """h = x W^T with x already in the compute dtype (bf16) and possibly zero-padded along K (so that the
    G-sized GEMM gets a K that is a multiple of 128), W the fp32 parameter [N, K].  Backward writes the weight
    gradient in fp32 straight out of the GEMM (no bf16 round trip)."""

    @staticmethod
    def forward(ctx, x, W, wbuf, fresh, defer=False, in_cell=None):
        N, K = W.shape
        Kp = x.shape[1]
        ctx.defer = bool(defer)
        # in_cell: see _GATTail.forward (x is the activated output of the layer below, over exactly these rows and columns)
        ctx.in_cell = in_cell if (in_cell is not None and in_cell.get("armed") and in_cell.get("rows") == x.shape[0]
                                  and x.dtype == torch.bfloat16 and x.is_contiguous()) else None
        assert Kp >= K and wbuf.shape == (N, Kp) and wbuf.dtype == x.dtype
        if not fresh:                              # (fresh: the caller has just cast W into wbuf, e.g. ops.cast_rows)
            wbuf[:, :K].copy_(W)                   # cast into the persistent padded image (pad columns stay zero)
        ctx.save_for_backward(x, wbuf)
        ctx.K = K
        # FlatAdamW keeps W.grad as a view of its flat gradient buffer: the weight-gradient GEMM writes there
        # directly (FlatAdamW.backward then has nothing to copy for this parameter)
        g = W.grad
        ctx.wgrad = g if (g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == W.shape) else None
        return gemm_tn(x, wbuf) if _own_forward(x.shape[0], Kp) else torch.nn.functional.linear(x, wbuf)

    @staticmethod
    def backward(ctx, g):
        x, wbuf = ctx.saved_tensors
        # a queued weight gradient reads g LATER, on another stream, ordered only behind what was enqueued before this stage
        # (GraphedStepper._late_event): it may be queued only if g needs no copy or cast launched here
        g_final = g.is_contiguous() and g.dtype == x.dtype
        g = g.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            # dx = g W: the same kernel as the forward map on the transposed weight image (contraction index contiguous)
            dx = None
            M_, Kp_ = g.shape[0], wbuf.shape[1]
            if (GEMM_OWN[0] and g.is_cuda and g.dtype == torch.bfloat16 and M_ >= 2560 and Kp_ % 256 == 0
                    and wbuf.shape[0] % 64 == 0 and wbuf.is_contiguous() and ((M_ + 319) // 320) * (Kp_ // 256) >= GEMM_DGRAD_MIN_WGS):
                dx = torch.empty((M_, Kp_), dtype=torch.bfloat16, device=g.device)
                if ctx.in_cell is not None and x.shape == (M_, Kp_):
                    # the LeakyReLU' of the layer below in this GEMM's epilogue (x IS that layer's activated output)
                    rc = model_lib().spadot_gemm_nn_bf16_masked(g.data_ptr(), g.shape[1], wbuf.data_ptr(), Kp_, dx.data_ptr(), Kp_, M_,
                                                                Kp_, g.shape[1], x.data_ptr(), Kp_, float(ctx.in_cell["slope"]), _stream())
                    if rc == 0:
                        ctx.in_cell["premasked"] = True
                else:
                    rc = model_lib().spadot_gemm_nn_bf16(g.data_ptr(), g.shape[1], wbuf.data_ptr(), Kp_, dx.data_ptr(), Kp_, M_, Kp_,
                                                         g.shape[1], _stream())
                if rc == -22:
                    dx = None
                elif rc != 0:
                    _check(rc, "spadot_gemm_nn_bf16")
            if dx is None:
                dx = g @ wbuf
        # (x[:, :K] is a strided view: the GEMM takes its row stride, the result is a dense [N, K])
        dW = None
        if ctx.needs_input_grad[1]:
            if ctx.defer and ctx.wgrad is not None and _deferring() and g_final:
                # nothing downstream reads dW: queued (the closure keeps g and x alive), written into the flat gradient later
                DEFERRED[0].append(lambda g=g, x=x, K=ctx.K, out=ctx.wgrad: wgrad_bf16(g, x, K, out))
                dW = ctx.wgrad
            else:
                dW = wgrad_bf16(g, x, ctx.K, ctx.wgrad if (ctx.wgrad is not None and _DIRECT_GRAD[0]) else None)
        return dx, dW, None, None, None, None


class _FirstMapSeeds(torch.autograd.Function):
    """h = x32 W^T for the SVGP encoder's G-sized first map on the b seeds (encoder.py:7-34), forward in fp32 from the cached
    fp32 seed rows as before; the WEIGHT GRADIENT g^T x [N x G] is taken on the matrix cores (csrc/gemm_wgrad_bf16.hip:
    N / 256 x G / 256 tiles, each walking all b rows) from the bf16 image of the same rows that the GAT branch reads.
    The library ran this fp32 product on 12-47 workgroups: 41 us alone, 130-160 us at the end of the SVGP backward beside
    the first GAT layer's weight-gradient GEMM -- the optimizer waited for it.  Only in the bf16 compute dtype (the
    operands are rounded to bf16 like every other large product of that mode; fp32 compute keeps the library GEMM)."""

    @staticmethod
    def forward(ctx, x32, W, xbf):
        K = W.shape[1]
        ctx.save_for_backward(x32, xbf)
        ctx.K = K
        g = W.grad
        ctx.wgrad = g if (g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == W.shape) else None
        return torch.nn.functional.linear(x32[:, :K], W)

    @staticmethod
    def backward(ctx, g):
        x32, xbf = ctx.saved_tensors
        K = ctx.K
        M, N = g.shape
        lib = model_lib()
        gb = g.contiguous().to(torch.bfloat16)
        out = ctx.wgrad if (ctx.wgrad is not None and _DIRECT_GRAD[0]) else torch.empty((N, K), dtype=torch.float32, device=g.device)
        # ONE slice: N / 256 x G / 256 workgroups walk all b rows and write dW themselves.  (Eight slices -- until round 5 -- were
        # 96 workgroups of one 64-row chunk each plus a 25 MB sum of partials: 80 + 39 us in the step for 0.8 GFLOP; one slice
        # shortens the SVGP backward stage by 70 us, profiles/r05/ab_first_map_slices.txt.)
        S_ = 1
        need = int(lib.spadot_gemm_wgrad_bf16_workspace(M, N, K, S_))
        rc = -22
        if need >= 0:
            ws = torch.empty(max(need, 4), dtype=torch.float32, device=g.device)
            rc = lib.spadot_gemm_wgrad_bf16(gb.data_ptr(), N, xbf.data_ptr(), xbf.shape[1], out.data_ptr(), K, M, N, K, S_,
                                            ws.data_ptr(), _zero_row(g.device).data_ptr(), _stream())
        if rc == -22:                                   # (a shape the kernel refuses: the library's fp32 product)
            return None, torch.mm(g.t(), x32[:, :K]), None
        _check(rc, "spadot_gemm_wgrad_bf16")
        return None, out, None


def first_map_seeds_ok(x32, W, xbf):
    """Whether _FirstMapSeeds takes (x32 [b, >= G] fp32, W [N, G] fp32, xbf [b, Gp] bf16 image of the same rows)."""
    if xbf is None or not (x32.is_cuda and x32.dtype == torch.float32 and xbf.dtype == torch.bfloat16 and xbf.is_contiguous()):
        return False
    N, K = W.shape
    return (N % 256 == 0 and K % 4 == 0 and xbf.shape[0] == x32.shape[0] and xbf.shape[1] >= (K + 255) // 256 * 256
            and xbf.shape[1] % 8 == 0 and xbf.data_ptr() % 16 == 0 and x32.shape[0] >= 64 and not x32.requires_grad)


def first_map_seeds(x32, W, xbf):
    return _FirstMapSeeds.apply(x32, W, xbf)


def _small_weight_grad(g, x):
    """g^T x for the MLP stages' weight gradients (g [M, N], x [M, K], fp32).  For a mid-sized output over a long
    batch axis (256 x 64 from M = 512: the decoder's and the SVGP encoder's middle stage) the library picks ONE
    64 x 256 tile, i.e. one CU walks all of M: 33 us.  Eight slices of M as a batched GEMM + a fixed-order sum:
    12 us (tools/small_gemm_probe.py); everything else goes to the library as it is."""
    M, N = g.shape
    K = x.shape[1]
    if M >= 256 and M % 8 == 0 and min(N, K) >= 64 and N * K <= 65536:
        return torch.bmm(g.view(8, M // 8, N).transpose(1, 2), x.view(8, M // 8, K)).sum(0)
    return g.t() @ x


def sgemm_small_ok(*ts):
    """The small fp32 products of the MLP stages take csrc's k_sgemm_small: contiguous fp32 device matrices, small work."""
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.is_contiguous() for t in ts)


def sgemm_small(mode, A, B, bias=None):
    """mode 0: A [M, K] . B [K, N]; mode 1: A [M, K] . B[N, K]^T (+ bias [N]); mode 2: A[K, M]^T . B [K, N]  -> fp32 [M, N]
    (include/spadot_model.h: spadot_sgemm_small -- small tiles that find room beside the GAT branch's GEMMs)."""
    _need_cuda(A, B, bias)
    if mode == 0:
        (M, K), N = A.shape, B.shape[1]
    elif mode == 1:
        (M, K), N = A.shape, B.shape[0]
    else:
        (K, M), N = A.shape, B.shape[1]
    C = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _check(model_lib().spadot_sgemm_small(mode, _p(A), A.shape[1], _p(B), B.shape[1], _p(C), N, M, N, K,
                                          _p(bias) if bias is not None else None, 1, 0, 0, 0, _stream()), "spadot_sgemm_small")
    return C


def wgrad_small(g, x, slices=8):
    """g^T x [N, K] for g [M, N], x [M, K] (fp32) as `slices` row slices in ONE launch of k_sgemm_small (mode 2) and a
    fixed-order sum of the partial results: 8 x (N / 32) x (K / 32) small workgroups instead of the library's handful of
    256 x 64 macro tiles (which waited 150-180 us for a whole compute unit beside the GAT branch's GEMMs)."""
    _need_cuda(g, x)
    M, N = g.shape
    K = x.shape[1]
    S = slices if (M % slices == 0 and M // slices >= 32) else 1
    Ms = M // S
    part = torch.empty((S, N, K), dtype=torch.float32, device=g.device)
    _check(model_lib().spadot_sgemm_small(2, _p(g), N, _p(x), K, _p(part), K, N, K, Ms, None, S, Ms * N, Ms * K, N * K, _stream()),
           "spadot_sgemm_small")
    return part.sum(0) if S > 1 else part[0]


_SMALL_WORK = 1 << 27        # multiply-adds up to which a product counts as small (0.27 GFLOP)


class _HiddenMap(torch.autograd.Function):
    """h W^T for a small hidden -> hidden map of the SVGP encoder (b x 256 -> 64 at the default sizes; fp32, no bias: the next
    BatchNorm kernel folds it in).  The forward is the library's product as before; BOTH gradients go through k_sgemm_small
    (dx = g W as 128 workgroups of 32 x 32 outputs, dW = g^T x as 8 row slices x 16 such workgroups + a fixed-order sum):
    for these shapes (512 x 64 . 64 x 256, and 64 x 512 . 512 x 256) the library picks 256 x 64 macro tiles whose eight
    workgroups each need most of a compute unit's LDS and registers -- on the side stream beside the first GAT layer's
    weight-gradient GEMM they waited for that GEMM to end (150-185 us for 17 MFLOP, and the optimizer waited for them:
    rocprofv3 timeline, round 3).  (One workgroup per output tile walking the whole contraction lost too: 72-97 us.)"""

    @staticmethod
    def forward(ctx, x, W):
        ctx.save_for_backward(x, W)
        return torch.nn.functional.linear(x, W)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous()
        dx = dW = None
        if ctx.needs_input_grad[0]:
            small = sgemm_small_ok(g, W) and g.shape[0] * W.shape[0] * W.shape[1] <= _SMALL_WORK
            dx = sgemm_small(0, g, W) if small else g @ W
        if ctx.needs_input_grad[1]:
            small = sgemm_small_ok(g, x) and g.shape[0] * W.shape[0] * W.shape[1] <= _SMALL_WORK
            dW = wgrad_small(g, x) if small else g.t() @ x
        return dx, dW


def hidden_map(x, W):
    return _HiddenMap.apply(x.contiguous(), W)


class _LinearBias(torch.autograd.Function):
    """y = x W^T + b (fp32, or the compute dtype given by `cd`) with the bias gradient as a fixed-order column
    sum (k_colsum_parts) instead of the library's semaphore-based reduction."""

    @staticmethod
    def forward(ctx, x, W, b, cd):
        ctx.cd = cd
        if cd is None or cd == torch.float32:
            ctx.save_for_backward(x, W)
            return torch.addmm(b, x, W.t())
        # compute-dtype operands, fp32 accumulate AND fp32 result (no rounding of the output to the compute dtype,
        # no cast launch after the GEMM); the images are kept for the backward pass.  Small dependent launches cost
        # ~5 us each inside a replayed graph, so this stage is counted in launches: 4 forward, 4 backward.
        if cd == torch.bfloat16 and x.dtype == torch.float32 and x.shape[1] % 4 == 0:
            xc = torch.empty(x.shape, dtype=cd, device=x.device)
            Wc = torch.empty(W.shape, dtype=cd, device=W.device)
            cast_rows([(x, xc), (W.detach().contiguous(), Wc)])        # both casts in one launch
        else:
            xc, Wc = x.to(cd), W.to(cd)
        ctx.save_for_backward(xc, Wc)
        return torch.mm(xc, Wc.t(), out_dtype=torch.float32).add_(b)

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        g = g.contiguous().float()
        db = torch.empty(g.shape[1], dtype=torch.float32, device=g.device)
        _check(model_lib().spadot_colsum(_p(g), g.shape[0], g.shape[1], _p(db), _stream()), "spadot_colsum")
        cd = ctx.cd
        if cd is None or cd == torch.float32:
            dx = g @ W if ctx.needs_input_grad[0] else None
            dW = _small_weight_grad(g, x)
        else:
            gc = g.to(cd)
            dx = torch.mm(gc, W, out_dtype=torch.float32) if ctx.needs_input_grad[0] else None
            dW = torch.mm(gc.t(), x, out_dtype=torch.float32)
        return dx, dW, db, None


def linear_bias(x, W, b, compute_dtype=None):
    """F.linear(x, W, b) for 2-D x with a reproducible bias gradient (see _LinearBias)."""
    _need_cuda(x, W, b)
    return _LinearBias.apply(x.contiguous(), W, b, compute_dtype)


def weight_image(W, width, dtype, holder, tag="_wpad"):
    """The persistent compute-dtype image [N, width >= K] of the fp32 weight W [N, K] kept on `holder` (a module);
    one image per input width: training (padded) and inference (unpadded) keep theirs, addresses stay valid."""
    N, K = W.shape
    tag = f"{tag}_{width}_{str(dtype).split('.')[-1]}"
    buf = getattr(holder, tag, None)
    if buf is None or buf.device != W.device:
        buf = torch.zeros((N, width), dtype=dtype, device=W.device)
        object.__setattr__(holder, tag, buf)       # plain attribute: not a parameter, not a buffer (state_dict unchanged)
    return buf


def dense_cd(x, W, holder, tag="_wpad", fresh=False, defer=False, in_cell=None):
    """x [n, Kp >= K] in the compute dtype, W fp32 [N, K]; `holder` (a module) keeps the padded compute-dtype
    image of W between calls.  fresh=True: the image already holds the current W (cast by the caller).  defer=True: this
    map's weight gradient may be queued (ops.DEFERRED) instead of launched inside the backward pass."""
    return _DenseCD.apply(x, W, weight_image(W, x.shape[1], x.dtype, holder, tag), fresh, defer, in_cell)


# ----------------------------------------------------------------------------- small-MLP stages

_ZEROS = {}


def _const_zeros(like):
    """A read-only all-zero tensor shaped like `like`, kept per (shape, dtype, device): gradients that are identically zero
    are handed to autograd from here (consumers copy or add them, nobody writes them)."""
    key = (tuple(like.shape), like.dtype, str(like.device))
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros_like(like)
    return z


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lin_bias, gamma, beta, bn, slope):
        _need_cuda(x, gamma, beta)
        x = x.contiguous()
        b, F_ = x.shape
        y = torch.empty((b, F_), dtype=torch.float32, device=x.device)
        sm = torch.empty(F_, dtype=torch.float32, device=x.device)
        si = torch.empty(F_, dtype=torch.float32, device=x.device)
        mom = 0.1 if bn.momentum is None else float(bn.momentum)
        _check(model_lib().spadot_bn_act_forward(_p(x), _DT[x.dtype], None if lin_bias is None else _p(lin_bias), _p(gamma),
                                                 _p(beta), _p(bn.running_mean), _p(bn.running_var),
                                                 _p(bn.num_batches_tracked), b, F_, mom, float(bn.eps), float(slope), _p(y),
                                                 _p(sm), _p(si), _stream()), "spadot_bn_act_forward")
        ctx.save_for_backward(x, lin_bias, gamma, y, sm, si)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, lin_bias, gamma, y, sm, si = ctx.saved_tensors
        b, F_ = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty(F_, dtype=torch.float32, device=x.device)
        db = torch.empty(F_, dtype=torch.float32, device=x.device)
        _check(model_lib().spadot_bn_act_backward(_p(dy.contiguous().float()), _p(y), _p(x), _DT[x.dtype],
                                                  None if lin_bias is None else _p(lin_bias), _p(gamma), _p(sm), _p(si), b, F_,
                                                  ctx.slope, _p(dx), _p(dg), _p(db), _stream()), "spadot_bn_act_backward")
        # the batch mean removes a per-feature shift: the preceding Linear's bias has a zero gradient
        dlb = None if lin_bias is None else _const_zeros(lin_bias)       # (a kept all-zero tensor: no fill launch per step)
        return dx, dlb, dg, db, None, None


def bn_act(x, lin_bias, bn, slope=0.01):
    """leaky_relu(BatchNorm1d(x + lin_bias)) in training mode as one launch each way (bn: the nn.BatchNorm1d
    module whose parameters and running statistics are used and updated; x fp32 or bf16, result fp32)."""
    assert bn.training and bn.track_running_stats and bn.affine
    return _BNAct.apply(x, lin_bias, bn.weight, bn.bias, bn, slope)


class _LNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, slope):
        _need_cuda(x, gamma, beta)
        x = x.contiguous().float()
        b, F_ = x.shape
        y = torch.empty_like(x)
        sm = torch.empty(b, dtype=torch.float32, device=x.device)
        si = torch.empty(b, dtype=torch.float32, device=x.device)
        _check(model_lib().spadot_ln_act_forward(_p(x), _p(gamma), _p(beta), b, F_, float(eps), float(slope), _p(y), _p(sm),
                                                 _p(si), _stream()), "spadot_ln_act_forward")
        ctx.save_for_backward(x, gamma, y, sm, si)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, y, sm, si = ctx.saved_tensors
        b, F_ = x.shape
        dx = torch.empty_like(x)
        dg = torch.empty(F_, dtype=torch.float32, device=x.device)
        db = torch.empty(F_, dtype=torch.float32, device=x.device)
        _check(model_lib().spadot_ln_act_backward(_p(dy.contiguous().float()), _p(y), _p(x), _p(gamma), _p(sm), _p(si), b, F_,
                                                  ctx.slope, _p(dx), _p(dg), _p(db), _stream()), "spadot_ln_act_backward")
        return dx, dg, db, None, None


def ln_act(x, ln, slope=0.01):
    """leaky_relu(LayerNorm(x)) over the last dimension of a 2-D fp32 tensor, one launch forward, two backward."""
    assert ln.elementwise_affine and len(ln.normalized_shape) == 1
    return _LNAct.apply(x, ln.weight, ln.bias, ln.eps, slope)


# ----------------------------------------------------------------------------- SVGP encoder behind its first map (csrc/enc_fused.hip)

def encoder_mid_ok(h1, W2, Wfc):
    """Whether the three-launch form takes the encoder's stages behind the first map: fp32 rows of the training batch."""
    return bool(h1.is_cuda and h1.dtype == torch.float32 and h1.dim() == 2 and h1.is_contiguous() and W2.dtype == torch.float32
                and W2.is_contiguous() and Wfc.is_contiguous() and W2.shape[1] == h1.shape[1] and Wfc.shape[1] == W2.shape[0]
                and W2.data_ptr() % 16 == 0
                and model_lib().spadot_enc_fused_supported(h1.shape[0], h1.shape[1], W2.shape[0], Wfc.shape[0]))


class _EncoderMid(torch.autograd.Function):
    """[BatchNorm1d + LeakyReLU] -> hidden map -> [BatchNorm1d + LeakyReLU] -> SVGP_fc of encoder.py:7-34 (training mode) as TWO
    launches whose cross-workgroup data are partial products in global memory (include/spadot_model.h: spadot_enc_*), instead of
    bn_act, hidden_map, bn_act, linear_bias (four launches, each waiting for a compute-unit slot beside the GAT branch's GEMMs).
    Returns (z, pz): z [b, Q] is FILLED only when `fill` (one more small launch); otherwise the consumer of the partials pz fills it
    (spadot_svgp_pre2_partials: the SVGP stage's first kernel reads the partials and stores z on the way).
    Backward: the same kernels and products as the four separate ops, in their order."""

    @staticmethod
    def forward(ctx, h1, b1, g1, be1, bn1, slope1, W2, b2, g2, be2, bn2, slope2, Wfc, bfc, fill):
        _need_cuda(h1, W2, Wfc)
        lib = model_lib()
        b, F1 = h1.shape
        F2, Q = W2.shape[0], Wfc.shape[0]
        dev = h1.device
        np1, np2 = F1 // 16, F2 // 4
        ws = torch.empty(int(lib.spadot_enc_fused_workspace(b, F1, F2, Q)), dtype=torch.float32, device=dev)
        part, pz = ws[:np1 * b * F2], ws[np1 * b * F2:]
        y1 = torch.empty((b, F1), dtype=torch.float32, device=dev)
        h2 = torch.empty((b, F2), dtype=torch.float32, device=dev)
        y2 = torch.empty((b, F2), dtype=torch.float32, device=dev)
        st = torch.empty(2 * F1 + 2 * F2, dtype=torch.float32, device=dev)
        sm1, si1, sm2, si2 = st[:F1], st[F1:2 * F1], st[2 * F1:2 * F1 + F2], st[2 * F1 + F2:]
        mom = lambda bn: 0.1 if bn.momentum is None else float(bn.momentum)
        _check(lib.spadot_enc_bn_map(_p(h1), _p(b1), _p(g1), _p(be1), _p(bn1.running_mean), _p(bn1.running_var),
                                     _p(bn1.num_batches_tracked), b, F1, mom(bn1), float(bn1.eps), float(slope1), _p(y1), _p(sm1), _p(si1),
                                     _p(W2), F2, _p(part), _stream()), "spadot_enc_bn_map")
        _check(lib.spadot_enc_bn_fc(_p(part), np1, _p(b2), _p(g2), _p(be2), _p(bn2.running_mean), _p(bn2.running_var),
                                    _p(bn2.num_batches_tracked), b, F2, mom(bn2), float(bn2.eps), float(slope2), _p(h2), _p(y2), _p(sm2),
                                    _p(si2), _p(Wfc), Q, _p(pz), _stream()), "spadot_enc_bn_fc")
        z = torch.empty((b, Q), dtype=torch.float32, device=dev)
        if fill:
            _check(lib.spadot_enc_sum_z(_p(pz), np2, _p(bfc), b, Q, _p(z), _stream()), "spadot_enc_sum_z")
        ctx.save_for_backward(h1, b1, g1, y1, sm1, si1, W2, b2, g2, h2, y2, sm2, si2, Wfc)
        ctx.slopes = (float(slope1), float(slope2))
        ctx.mark_non_differentiable(pz)
        return z, pz

    @staticmethod
    def backward(ctx, dz, _unused):
        h1, b1, g1, y1, sm1, si1, W2, b2, g2, h2, y2, sm2, si2, Wfc = ctx.saved_tensors
        lib = model_lib()
        b, F1 = h1.shape
        F2, Q = W2.shape[0], Wfc.shape[0]
        dev = h1.device
        dz = dz.contiguous().float()
        # SVGP_fc (ops._LinearBias.backward)
        dbfc = torch.empty(Q, dtype=torch.float32, device=dev)
        _check(lib.spadot_colsum(_p(dz), b, Q, _p(dbfc), _stream()), "spadot_colsum")
        dy2 = dz @ Wfc
        dWfc = _small_weight_grad(dz, y2)
        # second BatchNorm + LeakyReLU (ops._BNAct.backward)
        dh2 = torch.empty_like(h2)
        dg2 = torch.empty(F2, dtype=torch.float32, device=dev)
        dbe2 = torch.empty(F2, dtype=torch.float32, device=dev)
        _check(lib.spadot_bn_act_backward(_p(dy2), _p(y2), _p(h2), DT_F32, None if b2 is None else _p(b2), _p(g2), _p(sm2), _p(si2), b, F2,
                                          ctx.slopes[1], _p(dh2), _p(dg2), _p(dbe2), _stream()), "spadot_bn_act_backward")
        # hidden map (ops._HiddenMap.backward)
        small = sgemm_small_ok(dh2, W2) and b * F2 * F1 <= _SMALL_WORK
        dy1 = sgemm_small(0, dh2, W2) if small else dh2 @ W2
        small = sgemm_small_ok(dh2, y1) and b * F2 * F1 <= _SMALL_WORK
        dW2 = wgrad_small(dh2, y1) if small else dh2.t() @ y1
        # first BatchNorm + LeakyReLU
        dh1 = torch.empty_like(h1)
        dg1 = torch.empty(F1, dtype=torch.float32, device=dev)
        dbe1 = torch.empty(F1, dtype=torch.float32, device=dev)
        _check(lib.spadot_bn_act_backward(_p(dy1.contiguous()), _p(y1), _p(h1), DT_F32, None if b1 is None else _p(b1), _p(g1), _p(sm1),
                                          _p(si1), b, F1, ctx.slopes[0], _p(dh1), _p(dg1), _p(dbe1), _stream()), "spadot_bn_act_backward")
        # (the batch mean removes a per-feature shift: the biases of the two maps in front of a BatchNorm have zero gradients)
        return (dh1, None if b1 is None else _const_zeros(b1), dg1, dbe1, None, None, dW2, None if b2 is None else _const_zeros(b2),
                dg2, dbe2, None, None, dWfc, dbfc, None)


def encoder_mid(h1, lin1_bias, bn1, slope1, W2, lin2_bias, bn2, slope2, Wfc, bfc, fill=True):
    """(z, pz, number of partials): see _EncoderMid."""
    assert bn1.training and bn2.training and bn1.track_running_stats and bn2.track_running_stats and bn1.affine and bn2.affine
    z, pz = _EncoderMid.apply(h1, lin1_bias, bn1.weight, bn1.bias, bn1, slope1, W2, lin2_bias, bn2.weight, bn2.bias, bn2, slope2,
                               Wfc, bfc, fill)
    return z, pz, W2.shape[0] // 4


# ----------------------------------------------------------------------------- fused hidden stages (csrc/mlp_chain.hip)

MLP_CHAIN = [True]     # test hook -- [False]: one launch per piece (tests compare the two paths)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*(t_.data_ptr() for t_ in tensors))


def _contiguous_grad_block(params):
    """One fp32 tensor over the .grad views of `params` when those are consecutive pieces of one storage, else None."""
    gs_ = [p.grad for p in params]
    if any(g is None or g.dtype != torch.float32 or not g.is_contiguous() for g in gs_):
        return None
    st0, off = gs_[0].untyped_storage(), gs_[0].storage_offset()
    pos = off
    for g in gs_:
        if g.untyped_storage().data_ptr() != st0.data_ptr() or g.storage_offset() != pos:
            return None
        pos += g.numel()
    return torch.empty(0, dtype=torch.float32, device=gs_[0].device).set_(st0, off, (pos - off,))


CHAIN_BF16_OUT = [False]
CHAIN_DX_ADD = [None]
_LAST_CHAIN_BF16 = []


class _MLPChain(torch.autograd.Function):
    """[Linear, LayerNorm, LeakyReLU] x n on a 2-D fp32 input: one launch forward, one + a column sum backward."""

    @staticmethod
    def forward(ctx, x, eps, slope, *params):
        n = len(params) // 4
        Ws, bs, gs, betas = params[0::4], params[1::4], params[2::4], params[3::4]
        b = x.shape[0]
        dims = [x.shape[1]] + [W.shape[0] for W in Ws]
        dev = x.device
        a = [torch.empty((b, d), dtype=torch.float32, device=dev) for d in dims[1:]]
        y = [torch.empty((b, d), dtype=torch.float32, device=dev) for d in dims[1:]]
        stats = torch.empty((2 * n, b), dtype=torch.float32, device=dev)
        mean, inv = [stats[2 * l] for l in range(n)], [stats[2 * l + 1] for l in range(n)]
        cd = (ctypes.c_int * (n + 1))(*dims)
        ce = (ctypes.c_double * n)(*[float(e) for e in eps])
        cs = (ctypes.c_double * n)(*[float(v) for v in slope])
        # (bf16 copy of the result for a matrix-core consumer: decoder's output map reads it instead of casting)
        ybf = torch.empty((b, dims[-1]), dtype=torch.bfloat16, device=dev) if CHAIN_BF16_OUT[0] else None
        _check(model_lib().spadot_mlp_chain_forward_bf16(_p(x), b, n, cd, _ptr_array(Ws), _ptr_array(bs), _ptr_array(gs),
                                                         _ptr_array(betas), ce, cs, _ptr_array(a), _ptr_array(y), _ptr_array(mean),
                                                         _ptr_array(inv), None if ybf is None else _p(ybf), _stream()),
               "spadot_mlp_chain_forward_bf16")
        if ybf is not None:
            _LAST_CHAIN_BF16.append(ybf)
        ctx.dx_add = CHAIN_DX_ADD[0]             # (set by mlp_chain(dx_add=...) around this call)
        ctx.save_for_backward(x, stats, *Ws, *gs, *a, *y)
        ctx.n, ctx.dims, ctx.slope = n, dims, [float(v) for v in slope]
        # FlatAdamW keeps every .grad as a view of its flat buffer; when this chain's parameters sit there back to back in
        # (W, bias, gamma, beta) order -- the module order -- the backward pass writes its result row straight into it
        ctx.flat_out = _contiguous_grad_block(params)
        return y[-1]

    @staticmethod
    def backward(ctx, dy):
        n, dims = ctx.n, ctx.dims
        sv = ctx.saved_tensors
        x, stats = sv[0], sv[1]
        Ws, gs, a, y = sv[2:2 + n], sv[2 + n:2 + 2 * n], sv[2 + 2 * n:2 + 3 * n], sv[2 + 3 * n:2 + 4 * n]
        b = x.shape[0]
        dev = x.device
        cd = (ctypes.c_int * (n + 1))(*dims)
        rows, width = ctypes.c_int(0), ctypes.c_int(0)
        lib = model_lib()
        _check(lib.spadot_mlp_chain_workspace(b, n, cd, ctypes.byref(rows), ctypes.byref(width)), "spadot_mlp_chain_workspace")
        ws = torch.empty((rows.value, width.value), dtype=torch.float32, device=dev)
        direct = ctx.flat_out is not None and _DIRECT_GRAD[0] and ctx.flat_out.numel() == width.value
        grads = ctx.flat_out if direct else torch.empty(width.value, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        mean, inv = [stats[2 * l] for l in range(n)], [stats[2 * l + 1] for l in range(n)]
        cs = (ctypes.c_double * n)(*ctx.slope)
        late = direct and _deferring()          # the parameter gradients feed only the optimizer: their column sum is queued
        add = ctx.dx_add
        assert add is None or (dx is not None and add.shape == x.shape and add.dtype == torch.float32 and add.is_contiguous())
        _check(lib.spadot_mlp_chain_backward_add(_p(dy.contiguous().float()), _p(x), b, n, cd, _ptr_array(Ws), _ptr_array(gs), cs,
                                                 _ptr_array(a), _ptr_array(y), _ptr_array(mean), _ptr_array(inv),
                                                 None if dx is None else _p(dx), None if add is None else _p(add), _p(ws),
                                                 None if late else _p(grads), _stream()),
               "spadot_mlp_chain_backward_add")
        if late:
            DEFERRED[0].append(lambda ws=ws, grads=grads: _check(
                lib.spadot_colsum(_p(ws), ws.shape[0], ws.shape[1], _p(grads), _stream()), "spadot_colsum"))
        out = []
        off = 0
        for l in range(n):
            din, dout = dims[l], dims[l + 1]
            out += [grads[off:off + dout * din].view(dout, din), grads[off + dout * din:off + dout * din + dout],
                    grads[off + dout * din + dout:off + dout * din + 2 * dout],
                    grads[off + dout * din + 2 * dout:off + dout * din + 3 * dout]]
            off += dout * din + 3 * dout
        return (dx, None, None, *out)


def mlp_chain_ok(x, stages):
    """stages: [(Linear, LayerNorm, negative_slope), ...].  True when csrc/mlp_chain.hip covers them."""
    if not (MLP_CHAIN[0] and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= len(stages) <= 4):
        return False
    dims = [x.shape[1]] + [lin.out_features for lin, _, _ in stages]
    if any(lin.in_features != d for (lin, _, _), d in zip(stages, dims[:-1])):
        return False
    ts = [t_ for lin, ln, _ in stages for t_ in (lin.weight, lin.bias, ln.weight, ln.bias)]
    if any(t_ is None or t_.dtype != torch.float32 or not t_.is_contiguous() or t_.data_ptr() % 16 for t_ in ts):
        return False
    return bool(model_lib().spadot_mlp_chain_supported(len(stages), (ctypes.c_int * len(dims))(*dims)))


def mlp_chain(x, stages, bf16_out=False, dx_add=None):
    """leaky_relu(LayerNorm(Linear(.))) applied stage after stage (decoder.py:3-20's hidden part).  bf16_out: also return a
    bf16 copy of the result (written by the same launch; detached: a GEMM operand, not a differentiable output).
    dx_add [like x, fp32]: a gradient that reaches x by another path and is NOT returned by that path's own backward (see
    cluster_losses_fb); the chain's backward launch adds it to the input gradient it returns."""
    params = [t_ for lin, ln, _ in stages for t_ in (lin.weight, lin.bias, ln.weight, ln.bias)]
    CHAIN_BF16_OUT[0], CHAIN_DX_ADD[0] = bool(bf16_out), dx_add
    try:
        out = _MLPChain.apply(x.contiguous(), [ln.eps for _, ln, _ in stages], [sl for _, _, sl in stages], *params)
    finally:
        CHAIN_BF16_OUT[0], CHAIN_DX_ADD[0] = False, None
    return (out, _LAST_CHAIN_BF16.pop()) if bf16_out else out


class _GradBias(torch.autograd.Function):
    """Identity whose backward adds a constant `extra` to the gradient (the fallback of mlp_chain(dx_add=...) for consumers
    that cannot fold the addition into a launch of their own)."""

    @staticmethod
    def forward(ctx, x, extra):
        ctx.extra = extra
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (ctx.extra if g is None else g + ctx.extra), None


def grad_bias(x, extra):
    return _GradBias.apply(x, extra)




class _HeadFC(torch.autograd.Function):
    """(mu | logvar) = h W^T + b on the bf16 rows of the last GAT layer with a two-launch backward: dh comes out in bf16 (what
    that layer's backward reads), dW and db as per-8-row partials + one column sum -- in place of a bias column sum, two
    library GEMMs and a bf16 cast."""

    @staticmethod
    def forward(ctx, h, W, bias):
        b, K = h.shape
        N = W.shape[0]
        # (a one-launch forward from the bf16 rows measured 18-35 us at the end of the forward pair against 5 + 7 us for the
        # cast + library GEMM, round 2: removed in round 5)
        out = torch.addmm(bias, h.float(), W.t())
        ctx.save_for_backward(h, W)
        ctx.flat_out = _contiguous_grad_block((W, bias))
        return out

    @staticmethod
    def backward(ctx, g):
        h, W = ctx.saved_tensors
        b, K = h.shape
        N = W.shape[0]
        dev = h.device
        dh = torch.empty_like(h)
        ws = torch.empty(((b + 7) // 8, N * K + N), dtype=torch.float32, device=dev)
        direct = ctx.flat_out is not None and _DIRECT_GRAD[0] and ctx.flat_out.numel() == N * K + N
        grads = ctx.flat_out if direct else torch.empty(N * K + N, dtype=torch.float32, device=dev)
        late = direct and _deferring()
        _check(model_lib().spadot_headfc_backward(_p(g.contiguous().float()), _p(h), _p(W), b, K, N, _p(dh), _p(ws),
                                                  None if late else _p(grads), _stream()), "spadot_headfc_backward")
        if late:
            DEFERRED[0].append(lambda ws=ws, grads=grads: _check(
                model_lib().spadot_colsum(_p(ws), ws.shape[0], ws.shape[1], _p(grads), _stream()), "spadot_colsum"))
        return dh, grads[:N * K].view(N, K), grads[N * K:]


def head_fc_ok(h, W, bias):
    return bool(h.is_cuda and h.dtype == torch.bfloat16 and h.dim() == 2 and h.is_contiguous()
                and W.dtype == torch.float32 and W.is_contiguous() and bias is not None and W.shape[0] <= 32
                and h.shape[1] % 8 == 0 and W.shape[1] == h.shape[1] and W.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0
                and W.numel() * 4 <= 65536)


def head_fc(h, W, bias):
    return _HeadFC.apply(h, W, bias)


# ----------------------------------------------------------------------------- loss tail (single-workgroup kernels)

_counters = {}


def _zero_counter(device):
    """One device word per device that the last-workgroup-finishes kernels count on (they leave it at 0)."""
    key = str(device)
    if key not in _counters:
        _counters[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _counters[key]


class _LatentHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zg, p_m, p_v, eps, Ls, Lg, rng_state=None):
        _need_cuda(zg, p_m, p_v)
        zg = zg.contiguous().float()
        p_m, p_v = p_m.contiguous().double(), p_v.contiguous().double()
        b = zg.shape[0]
        if eps is None:             # drawn inside the kernel from rng_state = (seed, launch count) on the device
            assert rng_state is not None and rng_state.dtype == torch.int64 and rng_state.numel() == 2
            eps = torch.empty((b, Ls + Lg), dtype=torch.float32, device=zg.device)
        else:
            eps, rng_state = eps.contiguous().float(), None
        assert zg.shape == (b, 2 * Lg) and p_m.shape == (b, Ls) and p_v.shape == (b, Ls) and eps.shape == (b, Ls + Lg)
        latent = torch.empty((b, Ls + Lg), dtype=torch.float32, device=zg.device)
        scal = torch.empty(2, dtype=torch.float32, device=zg.device)
        partials = torch.empty(2 * ((b + 7) // 8), dtype=torch.float64, device=zg.device)
        _check(model_lib().spadot_latent_head_forward(_p(zg), _p(p_m), _p(p_v), _p(eps), b, Ls, Lg, _p(latent), _p(scal),
                                                      _p(partials), _p(_zero_counter(zg.device)),
                                                      None if rng_state is None else _p(rng_state), _stream()),
               "spadot_latent_head_forward")
        ctx.save_for_backward(zg, p_v, eps, latent)
        ctx.dims = (b, Ls, Lg)
        return latent, scal[0], scal[1]

    @staticmethod
    def backward(ctx, g_latent, g_kl, g_al):
        zg, p_v, eps, latent = ctx.saved_tensors
        b, Ls, Lg = ctx.dims
        d_zg = torch.empty_like(zg)
        d_pm = torch.empty((b, Ls), dtype=torch.float64, device=zg.device)
        d_pv = torch.empty_like(d_pm)
        keep = [None if t is None else t.contiguous().float() for t in (g_latent, g_kl, g_al)]
        _check(model_lib().spadot_latent_head_backward(_p(zg), _p(p_v), _p(eps), _p(latent),
                                                       *(None if t is None else _p(t) for t in keep), b, Ls, Lg,
                                                       _p(d_zg), _p(d_pm), _p(d_pv), _stream()), "spadot_latent_head_backward")
        return d_zg, d_pm, d_pv, None, None, None, None


def latent_head(zg, p_m, p_v, eps, Ls, Lg, rng_state=None):
    """Reparameterised samples of both branches + GAT KL + alignment (SpaDOT.py:78-93) in one launch.
    zg [b, 2 Lg] = GAT_fc output (mu | logvar); p_m, p_v [b, Ls] SVGP posterior; eps [b, Ls+Lg] ~ N(0, 1).
    eps None: the kernel draws the noise itself from `rng_state` (int64 [2] on the device: seed, launch count).
    Returns (final_latent [b, Ls+Lg] fp32, GAT_KL, alignment)."""
    return _LatentHead.apply(zg, p_m, p_v, eps, Ls, Lg, rng_state)


class _ClusterLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, labels_all, seed_ids, centres, prev, gamma, cluster_list, do_km, do_ot):
        _need_cuda(z, labels_all, seed_ids, centres)
        z = z.contiguous().float()
        b, D = z.shape
        K = centres.shape[0]
        Kp = Kl = 0
        if do_ot:
            Kp, Kl = prev.shape[0], cluster_list.shape[0]
            assert gamma.shape == (Kp, Kl) and prev.shape[1] == D, (gamma.shape, prev.shape, Kl)
            assert gamma.is_contiguous() and prev.is_contiguous() and gamma.dtype == prev.dtype == torch.float32
        assert centres.shape == (K, D) and centres.dtype == torch.float32 and centres.is_contiguous()
        assert labels_all.dtype == seed_ids.dtype == torch.int64 and seed_ids.numel() == b
        out = torch.empty(2, dtype=torch.float32, device=z.device)
        work = torch.empty(K * D + K + 1 + b, dtype=torch.float32, device=z.device)
        ptr = lambda t: None if t is None else _p(t)
        _check(model_lib().spadot_cluster_losses_forward(_p(z), _p(labels_all), _p(seed_ids.contiguous()), _p(centres), ptr(prev),
                                                         ptr(gamma), ptr(cluster_list), b, D, K, Kp, Kl, int(do_km), int(do_ot),
                                                         _p(out), _p(work), _stream()), "spadot_cluster_losses_forward")
        ctx.save_for_backward(z, centres, work)
        ctx.ot = (prev, gamma, cluster_list)
        ctx.dims = (b, D, K, Kp, Kl, int(do_km), int(do_ot))
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_km, g_ot):
        z, centres, work = ctx.saved_tensors
        prev, gamma, cluster_list = ctx.ot
        b, D, K, Kp, Kl, do_km, do_ot = ctx.dims
        dz = torch.empty_like(z)
        keep = [None if t is None else t.contiguous().float() for t in (g_km, g_ot)]
        ptr = lambda t: None if t is None else _p(t)
        _check(model_lib().spadot_cluster_losses_backward(_p(z), _p(centres), ptr(prev), ptr(gamma), ptr(cluster_list), _p(work),
                                                          ptr(keep[0]), ptr(keep[1]), b, D, K, Kp, Kl, do_km, do_ot, _p(dz),
                                                          _stream()), "spadot_cluster_losses_backward")
        return dz, None, None, None, None, None, None, None, None


class _ClusterLossesFB(torch.autograd.Function):
    """cluster_losses whose gradient w.r.t. z is formed by the FORWARD launch for gradient seeds known then (w_km, w_ot: device
    scalars, e.g. views of the loss-weight vector) and handed to the caller (`dz`), who routes it to z's other consumer
    (mlp_chain(dx_add=dz)).  backward() returns NO gradient for z -- and refuses seeds other than the promised ones."""

    @staticmethod
    def forward(ctx, z, labels_all, seed_ids, centres, prev, gamma, cluster_list, do_km, do_ot, w_km, w_ot, box):
        b, D = z.shape
        K = centres.shape[0]
        Kp, Kl = (prev.shape[0], cluster_list.shape[0]) if do_ot else (0, 0)
        out = torch.empty(2, dtype=torch.float32, device=z.device)
        work = torch.empty(K * D + K + 1 + b, dtype=torch.float32, device=z.device)
        dz = torch.empty_like(z)
        ptr = lambda t: None if t is None else _p(t)
        rc = model_lib().spadot_cluster_losses_fb(_p(z), _p(labels_all), _p(seed_ids), _p(centres), ptr(prev), ptr(gamma),
                                                  ptr(cluster_list), b, D, K, Kp, Kl, int(do_km), int(do_ot), _p(w_km), _p(w_ot),
                                                  _p(out), _p(work), _p(dz), _stream())
        if rc == -95:
            box["dz"] = None
            return None, None
        _check(rc, "spadot_cluster_losses_fb")
        box["dz"] = dz
        ctx.seeds = (w_km.data_ptr() if do_km else None, w_ot.data_ptr() if do_ot else None)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_km, g_ot):
        for g, want in zip((g_km, g_ot), ctx.seeds):
            if want is not None and (g is None or g.data_ptr() != want):
                raise RuntimeError("cluster_losses_fb: backward seeded with something other than the weights its gradient was "
                                   "formed with (use ops.cluster_losses outside GraphedStepper's staged tail)")
        return (None,) * 12


def cluster_losses_fb(z, labels_all, seed_ids, centres, prev_centres, gamma, cluster_list, do_km, do_ot, w_km, w_ot):
    """(km, ot, dz) in ONE launch: the two losses (autograd outputs without a gradient path to z) and dz = d(w_km km + w_ot ot)
    / dz, or None when the shape is outside the fused kernel's range.  The caller must (a) route dz to z's gradient itself and
    (b) seed km / ot's backward with exactly w_km / w_ot (a backward seeded otherwise raises)."""
    z = z.contiguous()
    if not (z.is_cuda and z.dtype == torch.float32 and centres.dtype == torch.float32 and centres.is_contiguous()
            and labels_all.dtype == seed_ids.dtype == torch.int64 and seed_ids.numel() == z.shape[0]
            and w_km.dtype == w_ot.dtype == torch.float32):
        return None
    if do_ot and not (gamma.is_contiguous() and prev_centres.is_contiguous() and gamma.dtype == prev_centres.dtype == torch.float32
                      and gamma.shape == (prev_centres.shape[0], cluster_list.shape[0]) and prev_centres.shape[1] == z.shape[1]):
        return None
    box = {}
    km, ot = _ClusterLossesFB.apply(z, labels_all, seed_ids.contiguous(), centres, prev_centres, gamma, cluster_list, do_km, do_ot,
                                    w_km, w_ot, box)
    if box.get("dz") is None:
        return None
    return km, ot, box["dz"]


def cluster_losses(z, labels_all, seed_ids, centres, prev_centres=None, gamma=None, cluster_list=None, do_km=True, do_ot=False):
    """(K-means loss, OT loss) of one batch (_train_utils.py:240-253, 272-307) in one launch; see
    include/spadot_model.h.  Gradients flow to z only (centres, plan and labels are epoch constants)."""
    return _ClusterLosses.apply(z, labels_all, seed_ids, centres, prev_centres, gamma, cluster_list, do_km, do_ot)


class _MixLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w6, *terms):
        assert len(terms) == 6 and w6.numel() == 6 and w6.dtype == torch.float32
        terms = [t.reshape(()).float() if t.dtype != torch.float32 else t for t in terms]
        _need_cuda(w6, *terms)
        out8 = torch.empty(8, dtype=torch.float32, device=w6.device)

        def run(terms=terms, w6=w6, out8=out8):
            arr = (ctypes.c_void_p * 6)(*(t.data_ptr() for t in terms))
            _check(model_lib().spadot_mix_losses_forward(arr, _p(w6), _p(out8), _stream()), "spadot_mix_losses_forward")

        # the loss VALUES feed nothing in the backward pass (its seed is a constant): with a queue open they are computed
        # later, off the step's dependency chain (ops.DEFERRED); out8 is filled then
        if DEFERRED[0] is not None:
            DEFERRED[0].append(run)
        else:
            run()
        ctx.save_for_backward(w6)
        log7 = out8[:7]
        ctx.mark_non_differentiable(log7)
        ctx.set_materialize_grads(False)          # (no zero-filled gradient for the logging vector: one launch less per step)
        return out8[7], log7

    @staticmethod
    def backward(ctx, g, _):
        (w6,) = ctx.saved_tensors
        if g is None:
            return (None,) * 7
        if g.data_ptr() in _UNIT_SEEDS:            # d elbo / d term_k = w6[k] for the constant-one seed: no launch
            return (None, *(w6[k] if ctx.needs_input_grad[k + 1] else None for k in range(6)))
        g6 = torch.empty(6, dtype=torch.float32, device=w6.device)
        _check(model_lib().spadot_mix_losses_backward(_p(g.contiguous().float()), _p(w6), _p(g6), _stream()),
               "spadot_mix_losses_backward")
        return (None, *(g6[k] if ctx.needs_input_grad[k + 1] else None for k in range(6)))


def mix_losses(w6, terms):
    """elbo = sum_k w6[k] * terms[k] (_train_utils.py:205-212) and the logging vector (elbo, terms...),
    one launch each way.  w6: device fp32 [6]; terms: six 0-dim device tensors."""
    return _MixLosses.apply(w6, *terms)


# ----------------------------------------------------------------------------- SVGP pieces

def kernel_matrix(x, z, kernel_type="Gaussian", scale=0.1):
    """K(x, z) of svgp.py:110-125 as a HIP kernel; x [n,d], z [m,d] (fp32 or fp64), no gradient
    (coordinates and inducing points are not trainable: svgp.py:24-30)."""
    _need_cuda(x, z)
    x, z = x.contiguous(), z.contiguous().to(x.dtype)
    K = torch.empty((x.shape[0], z.shape[0]), dtype=x.dtype, device=x.device)
    _check(model_lib().spadot_kernel_matrix(_p(x), _p(z), x.shape[0], z.shape[0], x.shape[1], float(scale),
                                            KERNEL_KINDS[kernel_type], _DT[x.dtype], _p(K), _stream()),
           "spadot_kernel_matrix")
    return K


class _RowDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B):
        _need_cuda(A, B)
        A, B = A.contiguous(), B.contiguous()
        L, n, m = A.shape
        assert B.shape == (n, m) and B.dtype == A.dtype, (A.shape, B.shape)     # (ONE [n, m] matrix under every l)
        out = torch.empty((L, n), dtype=A.dtype, device=A.device)
        _check(model_lib().spadot_rowdot_forward(_p(A), _p(B), L, n, m, _DT[A.dtype], _p(out), _stream()),
               "spadot_rowdot_forward")
        ctx.save_for_backward(B)
        ctx.shape = (L, n, m)
        return out

    @staticmethod
    def backward(ctx, g):
        (B,) = ctx.saved_tensors
        L, n, m = ctx.shape
        g = g.contiguous()
        gA = torch.empty((L, n, m), dtype=g.dtype, device=g.device)
        _check(model_lib().spadot_rowdot_backward(_p(g), _p(B), L, n, m, _DT[g.dtype], _p(gA), _stream()),
               "spadot_rowdot_backward")
        return gA, None


def rowdot(A, B):
    """out[l, i] = sum_k A[l, i, k] * B[i, k]  (B constant)."""
    return _RowDot.apply(A, B)


SWEEP_MAX_M = 310      # largest matrix the register/LDS-resident sweep kernel takes (include/spadot_model.h)

# sizes up to this go to one sweep launch, larger ones are split in two (measured, 10 matrices, graph replay: one launch
# 0.204 ms at m = 236 but 0.54 / 0.56 ms at 300 / 310, where two blocks take 0.26 / 0.27 ms; m = 600: 0.68 ms as 4 blocks
# against 1.26 ms as 2; tools/spd_split.py).  279 = the largest m whose 9 x 9 tiles stay in registers.  Alone the two-block
# form wins from ~256 on, but its ~25 short launches queue behind the GAT branch's GEMM workgroups inside the step: with m =
# 261 / 262 (two of the benchmark's five time points) one launch gave 559.3 / 561.8 steps/s against 557.3 / 556.8 (same box).
_SPLIT = [279]        # (a list: tests lower it to run the two-block elimination on small matrices)


def _sweep_kernel(A):
    L, m, _ = A.shape
    X = torch.empty_like(A)
    logdet = torch.empty(L, dtype=torch.float64, device=A.device)
    _check(model_lib().spadot_spd_inverse_logdet(_p(A), L, m, _p(X), _p(logdet), _stream()), "spadot_spd_inverse_logdet")
    return X, logdet


def _spd_inverse_logdet_nograd(A, need_logdet=True):
    """A [L, m, m] SPD fp64 -> (A^-1, log|A|).  m <= SWEEP_MAX_M: ONE launch of the sweep kernel gives both.
    Larger m: the same elimination in two blocks, recursively -- invert the leading block, form the Schur complement
    with batched GEMMs, invert it, back-substitute (Gauss-Jordan in block order: what the kernel would do pivot by
    pivot).  No library factorisation anywhere, so the step stays capturable in a hipGraph at every m."""
    L, m, _ = A.shape
    if m <= _SPLIT[0]:
        return _sweep_kernel(A.contiguous())
    m1 = (m + 1) // 2
    A11 = A[:, :m1, :m1]
    A12 = A[:, :m1, m1:]
    S11, ld1 = _spd_inverse_logdet_nograd(A11)
    B = S11 @ A12                                               # A11^-1 A12            [L, m1, m2]
    # one refinement step: entries of B are O(1) while S11's are O(cond), so the product with the explicit
    # inverse alone carries an eps * cond * |S11||A12| error that the Schur complement's inverse would amplify
    B = B + S11 @ (A12 - A11 @ B)
    C = A[:, m1:, m1:] - A12.transpose(1, 2) @ B                # Schur complement      [L, m2, m2]
    C = 0.5 * (C + C.transpose(1, 2))
    Ci, ld2 = _spd_inverse_logdet_nograd(C)
    X12 = -(B @ Ci)
    X = torch.empty_like(A)
    X[:, :m1, :m1] = S11 - X12 @ B.transpose(1, 2)
    X[:, :m1, m1:] = X12
    X[:, m1:, :m1] = X12.transpose(1, 2)
    X[:, m1:, m1:] = Ci
    return X, ld1 + ld2


class _SPDInverse(torch.autograd.Function):
    """(A^-1, log|A|) of a batch of SPD matrices [L, m, m] (fp64).  Forward: the register-resident symmetric
    sweep kernel (one launch for m <= 310, recursive two-block elimination above).  Backward (A symmetric):
    dA = -X G_X X + g_logdet X with X = A^-1 -- two batched GEMMs."""

    @staticmethod
    def forward(ctx, A, need_logdet):
        _need_cuda(A)
        assert A.dtype == torch.float64 and A.dim() == 3 and A.shape[1] == A.shape[2]
        X, logdet = _spd_inverse_logdet_nograd(A.contiguous(), need_logdet)
        ctx.save_for_backward(X)
        return X, logdet

    @staticmethod
    def backward(ctx, gX, gl):
        (X,) = ctx.saved_tensors
        gA = None
        if gX is not None:
            gA = -(X @ gX @ X)
        if gl is not None:
            t = gl.view(-1, 1, 1) * X
            gA = t if gA is None else gA + t
        return gA, None


def spd_inverse_logdet(A, need_logdet=True):
    """A [L, m, m] SPD fp64 -> (A^-1 [L, m, m], log|A| [L]).  need_logdet=False lets matrices beyond the
    kernel's size skip the library Cholesky (the returned logdet is then zeros)."""
    return _SPDInverse.apply(A, need_logdet)


class _ELBO(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, var, mv, tr, pm, pv, ktilde):
        ts = [t.contiguous() for t in (mu, var, mv, tr, pm, pv, ktilde)]
        _need_cuda(*ts)
        b, L = ts[0].shape
        out = torch.empty(2, dtype=ts[0].dtype, device=ts[0].device)
        _check(model_lib().spadot_elbo_forward(*(_p(t) for t in ts), b, L, _DT[ts[0].dtype], _p(out), _stream()),
               "spadot_elbo_forward")
        ctx.save_for_backward(*ts)
        return out

    @staticmethod
    def backward(ctx, g2):
        ts = ctx.saved_tensors
        b, L = ts[0].shape
        g2 = g2.contiguous()
        outs = [torch.empty_like(ts[0]) for _ in range(6)]
        _check(model_lib().spadot_elbo_backward(_p(g2), *(_p(t) for t in ts), b, L, _DT[ts[0].dtype],
                                                *(_p(o) for o in outs), _stream()), "spadot_elbo_backward")
        return (*outs, None)


def elbo_reduce(mu, var, mv, tr, pm, pv, ktilde):
    """(l3_sum, ce_sum) of svgp.py:97-104 and SpaDOT.py:125-142 over a [b, L] batch."""
    out = _ELBO.apply(mu, var, mv, tr, pm, pv, ktilde)
    return out[0], out[1]


_scratch = {}


def _get_scratch(device):
    key = str(device)
    if key not in _scratch:
        _scratch[key] = torch.empty(4096, dtype=torch.float64, device=device)
    return _scratch[key]


class _SqErr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, yhat, inv_scale):
        _need_cuda(y, yhat)
        y, yhat = y.contiguous(), yhat.contiguous()
        assert y.shape == yhat.shape and y.dtype == yhat.dtype
        out = torch.empty(1, dtype=y.dtype, device=y.device)
        _check(model_lib().spadot_sqerr_forward(_p(y), _p(yhat), y.numel(), float(inv_scale), _DT[y.dtype],
                                                _p(_get_scratch(y.device)), _p(out), _stream()), "spadot_sqerr_forward")
        ctx.save_for_backward(y, yhat)
        ctx.inv_scale = float(inv_scale)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        y, yhat = ctx.saved_tensors
        g1 = g.reshape(1).contiguous().to(y.dtype)
        gy = torch.empty_like(yhat)
        _check(model_lib().spadot_sqerr_backward(_p(g1), _p(y), _p(yhat), y.numel(), ctx.inv_scale, _DT[y.dtype],
                                                 _p(gy), _stream()), "spadot_sqerr_backward")
        return None, gy, None


def sqerr_sum(y, yhat, inv_scale):
    """inv_scale * sum (y - yhat)^2 (SpaDOT.py:89 with inv_scale = 1/input_dim); gradient to yhat only."""
    return _SqErr.apply(y, yhat, inv_scale)




class _LinearSqErr(torch.autograd.Function):
    """inv_scale * sum (y - (h W^T + bias))^2 with the output map in bf16 on the matrix cores (fp32 accumulate, fp32 result):
    cast, GEMM, [bias + squared error], sum forward; [d/d(o) in bf16 + bias gradient], two GEMMs backward -- 4 + 3
    launches where linear_bias + sqerr_sum take 5 + 5."""

    @staticmethod
    def forward(ctx, h, W, bias, y, inv_scale, hc=None, Wc=None):
        """hc: a bf16 copy of h that already exists (ops.mlp_chain(bf16_out=True)); Wc: a current bf16 image of W
        (FlatAdamW.maintain_image).  Whatever is missing is cast here (one launch)."""
        b, G = y.shape
        pairs = []
        if hc is None:
            hc = torch.empty(h.shape, dtype=torch.bfloat16, device=h.device)
            pairs.append((h, hc))
        if Wc is None:
            Wc = torch.empty(W.shape, dtype=torch.bfloat16, device=W.device)
            pairs.append((W.detach().contiguous(), Wc))
        if pairs:
            cast_rows(pairs)
        o = torch.mm(hc, Wc.t(), out_dtype=torch.float32)
        out = torch.empty(1, dtype=torch.float32, device=h.device)

        # (a queued launch gets its own partial-sum scratch: it must not share the per-device one with what runs meanwhile)
        scratch = _get_scratch(h.device) if DEFERRED[0] is None else torch.empty_like(_get_scratch(h.device))

        def value(o=o, bias=bias, y=y, out=out, scratch=scratch):
            _check(model_lib().spadot_bias_sqerr_forward(_p(o), _p(bias), _p(y), b, G, float(inv_scale), _p(scratch), _p(out), _stream()),
                   "spadot_bias_sqerr_forward")

        if DEFERRED[0] is not None:            # the VALUE of the reconstruction term feeds only the logging vector: queued
            DEFERRED[0].append(value)
        else:
            value()
        ctx.save_for_backward(hc, Wc, o, bias, y)
        ctx.inv_scale = float(inv_scale)
        ok = lambda g_, p_: g_ if (g_ is not None and g_.dtype == torch.float32 and g_.is_contiguous() and g_.shape == p_.shape) else None
        ctx.wgrad, ctx.bgrad = ok(W.grad, W), ok(bias.grad, bias)      # views of the flat gradient buffer (FlatAdamW): written in place
        return out[0]

    @staticmethod
    def backward(ctx, g):
        hc, Wc, o, bias, y = ctx.saved_tensors
        b, G = y.shape
        g1 = g.reshape(1).contiguous().float()
        gc = torch.empty((b, G), dtype=torch.bfloat16, device=y.device)
        direct = _DIRECT_GRAD[0]
        db = ctx.bgrad if (direct and ctx.bgrad is not None) else torch.empty(G, dtype=torch.float32, device=y.device)
        _check(model_lib().spadot_bias_sqerr_backward(_p(g1), _p(o), _p(bias), _p(y), b, G, ctx.inv_scale, _p(gc), _p(db),
                                                      _stream()), "spadot_bias_sqerr_backward")
        dh = torch.mm(gc, Wc, out_dtype=torch.float32) if ctx.needs_input_grad[0] else None
        if direct and ctx.wgrad is not None and _deferring():
            DEFERRED[0].append(lambda gc=gc, hc=hc, out=ctx.wgrad: torch.mm(gc.t(), hc, out_dtype=torch.float32, out=out))
            dW = ctx.wgrad
        elif direct and ctx.wgrad is not None:
            dW = torch.mm(gc.t(), hc, out_dtype=torch.float32, out=ctx.wgrad)
        else:
            dW = torch.mm(gc.t(), hc, out_dtype=torch.float32)
        return dh, dW, db, None, None, None, None


class _ReconFB(torch.autograd.Function):
    """_LinearSqErr with forward AND backward in ONE launch (csrc/recon_fb.hip) for a gradient seed known when the forward runs:
    gw = d loss / d recon, the loss weight lambda1 as a device scalar (GraphedStepper's loss-weight vector, entry 0 -- what
    _MixLosses.backward hands over for the constant-one seed).  The output map o = h W^T is never written; g = d recon / d o in
    bf16 (for the weight gradient g^T h, a library GEMM only the optimizer waits for) and dh = g W leave the forward launch.
    backward() returns them and REFUSES a seed other than the promised one (like ops.cluster_losses_fb)."""

    @staticmethod
    def forward(ctx, h, W, bias, y, inv_scale, hc, Wc, WcT, gw):
        lib = model_lib()
        b, K = h.shape
        G = W.shape[0]
        dev = h.device
        nfl = int(lib.spadot_recon_fb_workspace(b, K, G))
        GB, RB = (G + 127) // 128, (b + 63) // 64
        ws = torch.empty(nfl, dtype=torch.float32, device=dev)
        lossp = torch.empty(GB * RB, dtype=torch.float64, device=dev)
        gc = torch.empty((b, G), dtype=torch.bfloat16, device=dev)
        dh = torch.empty((b, K), dtype=torch.float32, device=dev)
        _check(lib.spadot_recon_fb(_p(hc), _p(Wc), _p(WcT), WcT.shape[1], _p(bias), _p(y), b, K, G, float(inv_scale), _p(gw), _p(gc), _p(ws),
                                   _p(lossp), _p(dh), _stream()), "spadot_recon_fb")
        out = torch.empty(1, dtype=torch.float32, device=dev)

        def value(lossp=lossp, out=out):
            _check(lib.spadot_sum_parts(_p(lossp), GB * RB, float(inv_scale), _p(out), _stream()), "spadot_sum_parts")

        if DEFERRED[0] is not None:            # the VALUE of the term feeds only the logging vector: queued
            DEFERRED[0].append(value)
        else:
            value()
        ctx.save_for_backward(hc, gc, dh)
        ctx.dbp = ws[GB * b * K:].view(RB, G)
        ctx.seed = gw.data_ptr()
        ok = lambda g_, p_: g_ if (g_ is not None and g_.dtype == torch.float32 and g_.is_contiguous() and g_.shape == p_.shape) else None
        ctx.wgrad, ctx.bgrad = ok(W.grad, W), ok(bias.grad, bias)      # views of the flat gradient buffer (FlatAdamW): written in place
        return out[0]

    @staticmethod
    def backward(ctx, g):
        if g is None or g.data_ptr() != ctx.seed:
            raise RuntimeError("recon_sqerr_fb: backward seeded with something other than the weight its gradient was formed for "
                               "(the forward launch already holds d loss / d h for THAT seed)")
        hc, gc, dh = ctx.saved_tensors
        lib = model_lib()
        dbp = ctx.dbp
        direct = _DIRECT_GRAD[0]
        db = ctx.bgrad if (direct and ctx.bgrad is not None) else torch.empty(dbp.shape[1], dtype=torch.float32, device=dh.device)
        dW = ctx.wgrad if (direct and ctx.wgrad is not None) else torch.empty((gc.shape[1], hc.shape[1]), dtype=torch.float32, device=dh.device)

        def rest(dbp=dbp, db=db, dW=dW, gc=gc, hc=hc):
            _check(lib.spadot_colsum(_p(dbp), dbp.shape[0], dbp.shape[1], _p(db), _stream()), "spadot_colsum")
            torch.mm(gc.t(), hc, out_dtype=torch.float32, out=dW)

        if direct and ctx.wgrad is not None and ctx.bgrad is not None and _deferring():
            DEFERRED[0].append(rest)           # nothing in the backward pass reads them: off the chain
        else:
            with torch.no_grad():
                rest()
        return dh, dW, db, None, None, None, None, None, None


def recon_fb_ok(h, W, bias, y, hc, Wc, WcT, gw):
    ok = lambda t_, like: (t_ is not None and t_.dtype == torch.bfloat16 and t_.is_contiguous() and t_.shape == like.shape)
    okT = (WcT is not None and WcT.dtype == torch.bfloat16 and WcT.is_contiguous() and WcT.dim() == 2 and WcT.shape[0] == W.shape[1]
           and WcT.shape[1] >= (W.shape[0] + 127) // 128 * 128 and WcT.shape[1] % 8 == 0 and WcT.data_ptr() % 16 == 0)
    return bool(okT and gw is not None and gw.is_cuda and gw.dtype == torch.float32 and h.is_cuda and h.dtype == torch.float32 and h.dim() == 2
                and y.dtype == torch.float32 and y.is_contiguous() and y.shape == (h.shape[0], W.shape[0]) and bias is not None
                and bias.dtype == torch.float32 and bias.is_contiguous() and bias.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0
                and ok(hc, h) and ok(Wc, W) and model_lib().spadot_recon_fb_supported(h.shape[0], h.shape[1], W.shape[0]))


def recon_sqerr_fb(h, W, bias, y, inv_scale, h_bf16, W_image, W_image_T, grad_weight):
    """recon_sqerr whose gradient is formed by the forward launch for the seed `grad_weight` (a device scalar): see _ReconFB.
    W_image_T: the transposed bf16 image of W, [K, G rounded up to 128] (zero pad columns)."""
    return _ReconFB.apply(h, W, bias, y, inv_scale, h_bf16, W_image, W_image_T, grad_weight)


def recon_sqerr_ok(h, W, bias, y):
    return bool(h.is_cuda and h.dtype == torch.float32 and y.dtype == torch.float32 and h.dim() == 2
                and y.dim() == 2 and h.shape[1] % 4 == 0 and W.shape[1] % 4 == 0 and y.shape == (h.shape[0], W.shape[0])
                and h.shape[0] <= 4096
                and bias is not None)


def recon_sqerr(h, W, bias, y, inv_scale, h_bf16=None, W_image=None):
    """inv_scale * sum (y - linear(h, W, bias))^2, the output map computed in bf16 (see _LinearSqErr)."""
    ok = lambda t_, like: (t_ is not None and t_.dtype == torch.bfloat16 and t_.is_contiguous() and t_.shape == like.shape)
    return _LinearSqErr.apply(h.contiguous(), W, bias, y.contiguous(), inv_scale, h_bf16 if ok(h_bf16, h) else None,
                              W_image if ok(W_image, W) else None)


def kmeans_assign(x, centers):
    """int32 labels = nearest centre (fp64 distance accumulation, first minimum wins)."""
    _need_cuda(x, centers)
    x = x.contiguous()
    centers = centers.contiguous().to(x.dtype)
    labels = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    _check(model_lib().spadot_kmeans_assign(_p(x), _p(centers), x.shape[0], centers.shape[0], x.shape[1],
                                            _DT[x.dtype], _p(labels), _stream()), "spadot_kmeans_assign")
    return labels


# ----------------------------------------------------------------------------- optimiser

def lloyd_steps(X, C, tol, done, inertia, part, steps):
    """`steps` Lloyd iterations for all restarts (include/spadot_model.h: spadot_lloyd_step); X [n, D], C [R, K, D]
    fp64 device tensors, C / done / inertia updated in place."""
    n, D = X.shape
    R, K, _ = C.shape
    lib = model_lib()
    for _ in range(steps):
        _check(lib.spadot_lloyd_step(_p(X), _p(C), n, D, K, R, float(tol), _p(part), _p(done), _p(inertia), None, 1,
                                     _stream()), "spadot_lloyd_step")


def lloyd_steps_groups(X, C, xoff, npts, n_max, groups, rpg, tol, done, inertia, part, steps, update=True, skip_done=False):
    """`steps` Lloyd iterations for several data sets at once (spadot_lloyd_step_groups): X [sum n, D] fp64, C [groups * rpg, K,
    D], xoff / npts int32 [groups], tol fp64 [groups] device tensors; C / done / inertia updated in place."""
    D = X.shape[1]
    K = C.shape[1]
    lib = model_lib()
    for _ in range(steps):
        _check(lib.spadot_lloyd_step_groups(_p(X), _p(C), _p(xoff), _p(npts), int(n_max), int(groups), int(rpg), D, K, _p(tol),
                                            _p(part), _p(done), _p(inertia), 1 if update else 0, 1 if skip_done else 0, _stream()),
               "spadot_lloyd_step_groups")


def knn(coords, kk):
    """Indices [n, kk] (int32, device) of the kk nearest points of every point, itself included, ordered by
    (distance, index); brute force in fp64 on the device (include/spadot_model.h: spadot_knn)."""
    _need_cuda(coords)
    x = coords.contiguous().double()
    n, d = x.shape
    out = torch.empty((n, kk), dtype=torch.int32, device=x.device)
    _check(model_lib().spadot_knn(_p(x), n, d, int(kk), _p(out), _stream()), "spadot_knn")
    return out


STAMP_BUF = [None]      # measurement aid: the int64 stamp buffer of a GraphedStepper built with SPADOT_STAMPS=1 (else None)


def stamp_if(slot):
    """A device timestamp into slot `slot` of the active stamp buffer, if there is one (tools/stage_stamps.py: where the time
    inside a captured stage goes, with no profiler attached)."""
    if STAMP_BUF[0] is not None:
        stamp(STAMP_BUF[0], slot)


def stamp(buf, slot):
    """buf[slot] (int64 device tensor) = the device timestamp counter (10 ns units) at this point of the current stream
    (include/spadot_model.h: spadot_stamp); capturable."""
    _check(model_lib().spadot_stamp(_p(buf), int(slot), _stream()), "spadot_stamp")


class FlatAdamW:
    """clip_grad_norm_(max_norm) + AdamW.step (_train_utils.py:214-217) as two HIP kernels over ONE
    flat fp32 parameter buffer.  Parameters of `module` are re-pointed into the flat buffer (so the
    gradient all-reduce of the data-parallel path is a single collective over `flat_grad`)."""

    def __init__(self, params, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=0.3, last=None, first=None):
        """last: parameters to place at the END of the flat buffers (the update is element-wise, so the order is free).
        The data-parallel path puts the parameters whose gradients arrive last in the backward pass there -- the
        first GAT layer's -- so that `flat_grad[:tail_offset]` can be all-reduced while they are still being
        computed (GraphedStepper, bucketed exchange).
        first: parameters to place at the START: step_head() updates them alone, step_rest() the others -- the SVGP
        encoder's go there, so that the next step's SVGP branch (the long pole of the forward pass) can start while the
        bulk of the update is still streaming (GraphedStepper.chained)."""
        params = [p for p in params if p.requires_grad]
        tail_ids = {id(p) for p in (last or [])}
        head_ids = {id(p) for p in (first or [])} - tail_ids
        params = ([p for p in params if id(p) in head_ids] + [p for p in params if id(p) not in tail_ids and id(p) not in head_ids]
                  + [p for p in params if id(p) in tail_ids])
        n_tail = sum(1 for p in params if id(p) in tail_ids)
        n_head = sum(1 for p in params if id(p) in head_ids)
        assert params and all(p.is_cuda and p.dtype == torch.float32 for p in params), \
            "FlatAdamW needs fp32 parameters on the MI355X"
        dev = params[0].device
        sizes = [p.numel() for p in params]
        # 16-byte aligned segments so every view supports vector access
        offs, tot = [], 0
        for s in sizes:
            offs.append(tot)
            tot += (s + 3) // 4 * 4
        self.flat_param = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scratch = torch.empty(4096, dtype=torch.float64, device=dev)
        for p, o, s in zip(params, offs, sizes):
            self.flat_param[o:o + s].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + s].view_as(p.data)
            p.grad = self.flat_grad[o:o + s].view_as(p.data)
        self.params, self.count = params, tot
        self.tail_offset = offs[len(params) - n_tail] if n_tail else None     # start of the `last` group (16-byte aligned)
        self.tail_params = params[len(params) - n_tail:] if n_tail else []
        self.head_count = (offs[n_head] if n_head < len(params) else tot) if n_head else 0     # elements of the `first` group
        self.lr, self.betas, self.eps, self.weight_decay, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.t = 0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)   # device-side step count (graph replays)
        # device scalar: what one entry of flat_grad has to be multiplied by to be THE gradient (1 on one replica;
        # data-parallel steps sum over replicas and set 1 / number of replicas that had a batch: parallel.run_epoch)
        self.grad_scale = torch.ones(1, dtype=torch.float32, device=dev)
        self._counter = torch.zeros(1, dtype=torch.int32, device=dev)   # last-workgroup-finishes counter of k_sumsq_last
        # bf16 images of weight matrices that step() keeps current (maintain_image); images_version changes whenever the
        # set does, so that a captured optimizer graph (whose launch carries the table by value) can be dropped
        self._images, self._image_table, self.images_version = [], None, 0
        self._image_stamps = {}          # image address -> (flat buffer version, parameter version) when it was last cast

    # ---- bf16 weight images kept current by the update itself ------------------------------------------------------
    def owns(self, param):
        """True while `param`'s storage is a slice of THIS optimizer's flat parameter buffer (a later FlatAdamW over the same
        module re-points the parameters into its own buffer: this one then owns nothing and must not be trusted to keep
        anything about them current)."""
        lo = self.flat_param.data_ptr()
        return param.is_cuda and lo <= param.data_ptr() < lo + 4 * self.count

    def _stamp(self, param):
        # torch's version counters: in-place writes through the parameter (load_state_dict, dist.broadcast(p), p.copy_) bump
        # the parameter's, writes through the flat buffer (flat_param.copy_, a broadcast of the flat buffer) the buffer's;
        # this optimizer's own update kernels go through neither.  (A write through `p.data` is invisible to both: call
        # refresh_images() after one.)
        return (self.flat_param._version, param._version)

    def maintain_image(self, param, image):
        """From now on every step() also writes bf16(param) into `image` [rows, Kp >= K] (a persistent tensor, e.g.
        ops.weight_image): the dense maps that read the image need no cast launch per step.  The image is brought up to
        date here.  Returns False (nothing registered) when the pair does not fit the kernel's conditions or the parameter
        is not (or no longer) this optimizer's."""
        if not self.owns(param):
            return False
        if any(im.data_ptr() == image.data_ptr() and p is param for p, im in self._images):
            return True
        idx = next((i for i, p in enumerate(self.params) if p is param), None)
        rows, K = (int(param.shape[0]), int(param.shape[1])) if param.dim() == 2 else (0, 0)
        ok = (idx is not None and param.dim() == 2 and image.dtype == torch.bfloat16 and image.is_contiguous()
              and image.dim() == 2 and image.shape[0] == rows and image.shape[1] >= K and K % 4 == 0 and image.shape[1] % 4 == 0
              and rows * K < 2 ** 31 and image.data_ptr() % 8 == 0 and len(self._images) < 8 and self.count % 4 == 0
              and not any(im.data_ptr() == image.data_ptr() for _, im in self._images))
        if not ok:
            return False
        self._images.append((param, image))
        self._image_table = None
        self.images_version += 1
        self.refresh_images([image])
        return True

    def maintains(self, image):
        return any(im.data_ptr() == image.data_ptr() for _, im in self._images)

    def refresh_images(self, only=None):
        """Re-cast the registered images from the fp32 parameters (after anything but step() changed them: a loaded
        state_dict, a broadcast)."""
        pairs = [(p.detach(), im) for p, im in self._images if only is None or any(im is o for o in only)]
        if pairs:
            with torch.no_grad():
                cast_rows(pairs)
        for p, im in self._images:
            if only is None or any(im is o for o in only):
                self._image_stamps[im.data_ptr()] = self._stamp(p)

    def sync_images(self):
        """Self-validation of the images: re-cast every registered image whose parameter was written by
        anything torch can see since the image was last brought up to date.  A handful of integer compares when nothing
        happened; GraphedStepper calls it in front of every step (a replayed graph holds no cast launch), the GAT encoder
        in its eager forward.  Returns the number of images refreshed."""
        stale = [im for p, im in self._images if self._image_stamps.get(im.data_ptr()) != self._stamp(p)]
        if stale:
            self.refresh_images(stale)
        return len(stale)

    def _images_struct(self):
        if self._image_table is None:
            from ._lib import WeightImages
            tab = WeightImages()
            tab.n = len(self._images)
            for k, (p, im) in enumerate(self._images):
                off = (p.data_ptr() - self.flat_param.data_ptr()) // 4
                tab.w[k].offset, tab.w[k].rows, tab.w[k].K, tab.w[k].Kp = off, p.shape[0], p.shape[1], im.shape[1]
                tab.w[k].image = im.data_ptr()
            self._image_table = tab
        return self._image_table

    def zero_grad(self):
        self.flat_grad.zero_()

    def backward(self, loss):
        """Gradients of `loss` written (not accumulated) into the flat gradient buffer: one multi-tensor
        copy instead of one AccumulateGrad add per parameter (~45 launches a step).  Parameters the loss
        does not reach get zeros, like a backward() after zero_grad()."""
        _DIRECT_GRAD[0] = True
        post = POST_CHAIN[0] = []
        try:
            grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        finally:
            _DIRECT_GRAD[0] = False
            POST_CHAIN[0] = None
        with torch.no_grad():
            for job in post:
                job()
        # (a gradient that already IS the flat view -- written in place by ops.dense_cd -- needs no copy)
        pairs = [(p.grad, g) for p, g in zip(self.params, grads) if g is not None and g.data_ptr() != p.grad.data_ptr()]
        dst = [d for d, _ in pairs]
        src = [s for _, s in pairs]
        self._zero_unreached(self.params, grads)
        torch._foreach_copy_(dst, src)

    def backward_partial(self, outputs, grad_outputs, params, extra_inputs=()):
        """One stage of a backward pass that is issued in pieces (GraphedStepper's staged mode): gradients of
        `outputs` (weighted by grad_outputs; None for a scalar loss) w.r.t. `params` go to their views of the flat
        buffer (written, not accumulated; zeros where unreachable), those w.r.t. `extra_inputs` are returned."""
        ins = list(params) + list(extra_inputs)
        if grad_outputs is None and torch.is_tensor(outputs) and outputs.dim() == 0 and outputs.dtype == torch.float32:
            # the seed of a scalar loss from a constant instead of a ones_like fill launch per step
            grad_outputs = unit_seed(outputs.device)
        _DIRECT_GRAD[0] = True
        post = POST_CHAIN[0] = []
        try:
            grads = torch.autograd.grad(outputs, ins, grad_outputs=grad_outputs, allow_unused=True)
        finally:
            _DIRECT_GRAD[0] = False
            POST_CHAIN[0] = None
        with torch.no_grad():
            for job in post:        # what the stage's backward functions left for its end (ops.POST_CHAIN)
                job()
        pg = grads[:len(params)]
        pairs = [(p.grad, g) for p, g in zip(params, pg) if g is not None and g.data_ptr() != p.grad.data_ptr()]
        self._zero_unreached(params, pg)
        if pairs:
            try:
                torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])
            except RuntimeError as ex:          # say WHICH slot refused the write (a .grad that is not a flat-buffer view any more)
                bad = [(tuple(d.shape), d.requires_grad, d.is_leaf, d.data_ptr() - self.flat_grad.data_ptr()) for d, _ in pairs
                       if d.requires_grad or not (0 <= d.data_ptr() - self.flat_grad.data_ptr() < 4 * self.count)]
                raise RuntimeError(f"{ex}; offending gradient slots (shape, requires_grad, is_leaf, byte offset in flat_grad): {bad}") from ex
        return grads[len(params):]

    def _zero_unreached(self, params, grads):
        """Slots of parameters the loss does not reach read zero afterwards (like backward() after zero_grad())."""
        for p, g in zip(params, grads):
            if g is None:
                p.grad.zero_()

    def grad_norm_sq(self):
        """Device scalar: squared global gradient norm (deterministic reduction)."""
        _check(model_lib().spadot_grad_sumsq(_p(self.flat_grad), self.count, _p(self.scratch), _p(self.sumsq),
                                             _stream()), "spadot_grad_sumsq")
        return self.sumsq

    def step(self):
        """clip + AdamW in two launches; the step count lives on the device, so the same launches can be captured in
        a hipGraph and replayed."""
        self.t += 1
        if self._images:
            _check(model_lib().spadot_clip_adamw_images_dev(_p(self.flat_param), _p(self.flat_grad), _p(self.exp_avg),
                                                            _p(self.exp_avg_sq), self.count, self.lr, self.betas[0], self.betas[1],
                                                            self.eps, self.weight_decay, self.max_norm, _p(self.scratch),
                                                            _p(self.sumsq), _p(self.step_dev), _p(self.grad_scale),
                                                            ctypes.byref(self._images_struct()), _stream()),
                   "spadot_clip_adamw_images_dev")
            return
        _check(model_lib().spadot_clip_adamw_dev(_p(self.flat_param), _p(self.flat_grad), _p(self.exp_avg),
                                                 _p(self.exp_avg_sq), self.count, self.lr, self.betas[0], self.betas[1],
                                                 self.eps, self.weight_decay, self.max_norm, _p(self.scratch), _p(self.sumsq),
                                                 _p(self.step_dev), _p(self._counter), _p(self.grad_scale), _stream()),
               "spadot_clip_adamw_dev")

    def _update_range(self, lo, hi):
        tab = ctypes.byref(self._images_struct()) if self._images else None
        _check(model_lib().spadot_adamw_range_dev(_p(self.flat_param), _p(self.flat_grad), _p(self.exp_avg), _p(self.exp_avg_sq),
                                                  lo, hi - lo, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                                  self.max_norm, _p(self.sumsq), _p(self.step_dev), _p(self.grad_scale), tab,
                                                  _stream()), "spadot_adamw_range_dev")

    def step_sharded(self, group=None):
        """step() for P data-parallel replicas that each own 1/P of the flat buffers (parallel.sharded_update: reduce-scatter of
        the gradient, the clip norm from one scalar all-reduce, this rank's slice updated by the same kernel as step(), all-gather
        of the parameters).  flat_grad holds THIS replica's gradient on entry.  The bf16 weight images are refreshed from the
        gathered parameters (the update kernel keeps only the rows inside this rank's slice current)."""
        from . import parallel as _par
        assert self.count % 4 == 0
        self.t += 1

        def local_sumsq(lo, hi):
            # (also advances the device step count: every rank calls this exactly once per step)
            if hi <= lo:                               # more ranks than slices: nothing to sum, the count still advances
                self.sumsq.zero_()
                self.step_dev += 1
                return self.sumsq
            _check(model_lib().spadot_grad_norm_step_dev(self.flat_grad[lo:hi].data_ptr(), hi - lo, _p(self.scratch), _p(self.sumsq),
                                                         _p(self.step_dev), _stream()), "spadot_grad_norm_step_dev")
            return self.sumsq

        def update_range(lo, hi, sumsq):
            if sumsq.data_ptr() != self.sumsq.data_ptr():
                self.sumsq.copy_(sumsq)
            self._update_range(lo, hi)

        _par.sharded_update(self.flat_grad, self.flat_param, local_sumsq, update_range, group=group)
        if self._images:
            self.refresh_images()

    def step_head(self):
        """First part of step() in two parts: gradient norm + step count, then the update of the `first` group alone.
        step_rest() must follow.  Together they leave the bits step() leaves (same arithmetic per element)."""
        assert self.head_count > 0 and self.count % 4 == 0
        self.t += 1
        _check(model_lib().spadot_grad_norm_step_dev(_p(self.flat_grad), self.count, _p(self.scratch), _p(self.sumsq),
                                                     _p(self.step_dev), _stream()), "spadot_grad_norm_step_dev")
        self._update_range(0, self.head_count)

    def step_rest(self):
        if self.head_count < self.count:
            self._update_range(self.head_count, self.count)
