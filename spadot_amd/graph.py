"""Spatial graph and mini-batch construction for the GAT branch (host side, one-off per run).

Replaces, without the dense N_t x N_t adjacency the reference builds:
  _Cal_Spatial_Net                /root/reference/SpaDOT/utils/_utils.py:52-100
  dense_to_sparse + NeighborLoader(num_neighbors=[f, f], batch_size=512, subgraph_type="induced")
                                  /root/reference/SpaDOT/utils/_train_utils.py:69-85
The loader is not shuffled in the reference, so its batches are the same every epoch: they are built
once here (node ids + CSR in both directions) and kept on the device.
"""
import numpy as np
import torch

from .ops import BatchGraph


def knn_graph(coords, k_cutoff, max_neigh=30, backend="sklearn", device=None):
    """Directed edges i -> j for j among the k_cutoff nearest neighbours of i (self excluded), plus one
    self loop per node (_utils.py:66-100: kNN over max_neigh+1 candidates, columns 1..k_cutoff kept;
    `G + eye`).  Returns int64 edge_index [2, E], row 0 = source i, row 1 = target j, sorted row-major
    like dense_to_sparse of the reference's adjacency.
    backend: 'sklearn' (the reference's host NearestNeighbors) or 'device' (ops.knn: brute force in HBM, same
    neighbours whenever no two candidates are exactly equidistant)."""
    coords = np.asarray(coords, dtype=np.float64)
    n = coords.shape[0]
    kk = min(max_neigh + 1, n)
    if backend == "device":
        from .ops import knn
        idx = knn(torch.as_tensor(coords).to(device or "cuda"), kk).cpu().numpy().astype(np.int64)
    else:
        from sklearn.neighbors import NearestNeighbors
        _, idx = NearestNeighbors(n_neighbors=kk, algorithm="auto").fit(coords).kneighbors(coords)
    nb = idx[:, 1:k_cutoff + 1]
    src = np.repeat(np.arange(n, dtype=np.int64), nb.shape[1])
    dst = nb.reshape(-1).astype(np.int64)
    keep = src != dst          # a duplicate point can return another index first; self loops are added once below
    src = np.concatenate([src[keep], np.arange(n, dtype=np.int64)])
    dst = np.concatenate([dst[keep], np.arange(n, dtype=np.int64)])
    key = np.unique(src * n + dst)
    return np.stack([key // n, key % n])


def _csr_both(src, dst, n, n_tgt=None):
    """CSR by target (targets 0..n_tgt-1) and its transpose (sources 0..n-1) for an edge list that already
    has exactly one self loop per target."""
    n_tgt = n if n_tgt is None else n_tgt
    order = np.argsort(dst, kind="stable")
    col = src[order].astype(np.int32)
    tgt = dst[order]
    rowptr = np.zeros(n_tgt + 1, dtype=np.int32)
    np.cumsum(np.bincount(tgt, minlength=n_tgt), out=rowptr[1:])
    perm = np.argsort(col, kind="stable").astype(np.int32)
    col_t = tgt[perm].astype(np.int32)
    rowptr_t = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(col, minlength=n), out=rowptr_t[1:])
    return rowptr, col, rowptr_t, col_t, perm


def build_batch_graph(edge_index, n, device, seeds=None, tiers=None, order_key=None, plans=False):
    """edge_index [2, E] (source, target) -> BatchGraph with GATConv's self-loop convention applied
    (existing self loops dropped, one per node appended: SURVEY App. A).
    seeds (optional int): also attach `.seed_graph`, the same graph with only the first `seeds` nodes as
    targets -- what the last GAT layer needs when only the seeds' rows of its output are used.
    tiers (optional, cumulative hop sizes [b, n1, n] of a 2-hop batch whose nodes are ordered seeds, hop 1,
    hop 2): also attach `.layer_graphs` = (g2, g3), the graphs the 2nd and 3rd GAT layer need when only the
    seeds' rows of the 3rd layer are used: g2 has the first n1 nodes as targets (the sources of g3), g3 has
    n1 sources and the b seeds as targets."""
    ei = edge_index.cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    src, dst = ei[0].astype(np.int64), ei[1].astype(np.int64)
    keep = src != dst
    loops = np.arange(n, dtype=np.int64)
    src = np.concatenate([src[keep], loops])
    dst = np.concatenate([dst[keep], loops])

    def dev(parts):
        return (torch.from_numpy(np.ascontiguousarray(p)).to(device) for p in parts)

    g = BatchGraph(n, *dev(_csr_both(src, dst, n)))
    if seeds is not None and 0 < seeds < n:
        sel = dst < seeds
        g.seed_graph = BatchGraph(n, *dev(_csr_both(src[sel], dst[sel], n, seeds)), n_tgt=seeds)
    if tiers is not None and len(tiers) == 3 and tiers[2] == n and 0 < tiers[0] <= tiers[1] <= n:
        b, n1 = int(tiers[0]), int(tiers[1])
        s2 = dst < n1
        s3 = dst < b
        if src[s3].max(initial=0) < n1:           # every in-neighbour of a seed is a seed or hop-1 node
            g2 = BatchGraph(n, *dev(_csr_both(src[s2], dst[s2], n, n1)), n_tgt=n1)
            g3 = BatchGraph(n1, *dev(_csr_both(src[s3], dst[s3], n1, b)), n_tgt=b)
            g.layer_graphs = (g2, g3)
            if plans:
                attach_plans(g2, order_key)       # (the seeds-only last layer stays on the per-edge kernels)
    if plans:                                     # plans=True / order_key: see attach_plans
        attach_plans(g, order_key)
    return g


def morton_key(coords):
    """Z-order key of 2-D coordinates (16 bits per axis): nodes that are close in space get close keys."""
    c = np.asarray(coords, dtype=np.float64)
    lo, hi = c.min(axis=0), c.max(axis=0)
    q = np.clip(((c - lo) / np.maximum(hi - lo, 1e-300) * 65535.0), 0, 65535).astype(np.uint64)

    def spread(v):
        v = (v | (v << 8)) & np.uint64(0x00FF00FF)
        v = (v | (v << 4)) & np.uint64(0x0F0F0F0F)
        v = (v | (v << 2)) & np.uint64(0x33333333)
        v = (v | (v << 1)) & np.uint64(0x55555555)
        return v

    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1))


def induced_batch(edge_index, n_nodes, seeds, hops=2, order_key=None, return_tiers=False):
    """Seeds + their `hops`-hop in-neighbourhood (sources of edges pointing at the frontier), seeds
    first; edges = every original edge with both ends inside (NeighborLoader 'induced', fan-out >=
    in-degree: SURVEY App. B).  Returns (n_id int64 [n_sub], sub_edge_index int64 [2, E_sub]).
    order_key (optional, one value per node): inside each hop the nodes are laid out by ascending key
    instead of by id -- the subgraph is the same up to relabelling, results on the seeds do not change.
    return_tiers: also return the cumulative node counts per hop (seeds, seeds + hop 1, ...)."""
    ei = edge_index.cpu().numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    src, dst = ei[0], ei[1]
    seen = np.zeros(n_nodes, dtype=bool)
    seeds = np.asarray(seeds, dtype=np.int64)
    seen[seeds] = True
    n_id = [seeds]
    frontier_mask = seen.copy()
    for _ in range(hops):
        hit = frontier_mask[dst]
        cand = np.unique(src[hit])
        new = cand[~seen[cand]]
        seen[new] = True
        n_id.append(new)
        frontier_mask = np.zeros(n_nodes, dtype=bool)
        frontier_mask[new] = True
    if order_key is not None:
        key = np.asarray(order_key)
        n_id = [n_id[0]] + [t[np.argsort(key[t], kind="stable")] for t in n_id[1:]]
    tiers = np.cumsum([t.size for t in n_id]).tolist()
    n_id = np.concatenate(n_id)
    relabel = np.full(n_nodes, -1, dtype=np.int64)
    relabel[n_id] = np.arange(n_id.size)
    keep = seen[src] & seen[dst]
    sub = np.stack([relabel[src[keep]], relabel[dst[keep]]])
    return (n_id, sub, tiers) if return_tiers else (n_id, sub)


class Batch:
    """One precomputed training batch of a time point (what the reference's loader yields each epoch)."""

    def __init__(self, n_id, graph, batch_size):
        self.n_id = n_id            # int64 device tensor, seeds first
        self.graph = graph          # BatchGraph on the device
        self.batch_size = int(batch_size)
        self.x = None               # optional cache: coordinates of the batch nodes [n_sub, 2]
        self.y = None               # optional cache: expression rows in the compute dtype, K padded to 128
        self.y_seed32 = None        # optional cache: the seeds' expression rows in fp32 (reconstruction target)


def precompute_batches(edge_index, n_nodes, batch_size, device, hops=2, coords=None, plans=False, only=None):
    """All batches of one time point in loader order (consecutive seed blocks, last one partial); with `only` (a set of
    batch indices) the other entries of the list are None (a data-parallel rank builds its own batches only).
    With `coords`, the nodes of each hop of every batch are stored in Z-order of their coordinates: the
    neighbours a GAT workgroup gathers are then close in memory and in launch order (L2 reuse per XCD).
    Each batch graph carries the per-layer graphs of build_batch_graph(tiers=...)."""
    key = morton_key(coords) if coords is not None else None
    out = []
    for s in range(0, n_nodes, batch_size):
        if only is not None and s // batch_size not in only:
            out.append(None)
            continue
        seeds = np.arange(s, min(n_nodes, s + batch_size))
        n_id, sub, tiers = induced_batch(edge_index, n_nodes, seeds, hops, order_key=key, return_tiers=True)
        g = build_batch_graph(sub, n_id.size, device, seeds=seeds.size, tiers=tiers if hops == 2 else None,
                              order_key=None if key is None else key[n_id], plans=plans)
        out.append(Batch(torch.from_numpy(n_id).to(device), g, seeds.size))
    return out


# ------------------------------------------------------------------------------------------------------------------
# Block plans for the matrix-core GAT edge kernels (csrc/model_kernels.hip: k_gat_agg / k_gat_edot).
#
# The edge phase of a GAT layer is out[i] = sum_j alpha[i, j] h[j]: per edge one 4 KB row gather.  Nodes that are
# close in space share most of their neighbours, so a block of 32 consecutive targets (in Z-order) touches ~60-130
# DISTINCT sources, not 32 x 31: the kernels therefore work on (32 rows) x (distinct columns of the block) tiles --
# every distinct row is fetched once per block and the weighted sum becomes a small dense product
# alpha_tile [32 x S] . h_rows [S x C] on the matrix cores, with zeros where there is no edge.
# A plan is the host-side (one-off) description of those tiles.

PLAN_ROWS = 32       # rows (targets, or sources on the transposed side) per block = the MFMA M dimension
PLAN_KPAD = 32       # a block's column list is padded to a multiple of this (two 16-wide k-steps)


class BlockPlan:
    """rows  int32 [nb * 32]      row node id of every block slot (-1 = padding)
       sptr  int32 [nb + 1]       start of each block's column list in `cols` (multiples of PLAN_KPAD)
       cols  int32 [sptr[nb]]     distinct column node ids of the block, ascending; the padding repeats a valid id
       cell  int32 [sptr[nb] * 32]  for chunk q = position // 16 and cell (r, kk) -> index r * 16 + kk:
                                  alpha row (edge position in the by-target CSR) of edge (column slot 16 q + kk -> row r),
                                  or -1 where the tile is zero (read by the target-side backward: d(alpha) scatter)
       cellq int32 [E]            indexed by the edge's position in the BY-TARGET CSR: q * 512 + position of the edge's
                                  cell inside the chunk's 32 x 16 weight tile in MFMA fragment order,
                                  ((kk >> 3) * 32 + r) * 8 + (kk & 7) -- where the per-node kernels put alpha into the
                                  dense weight image the aggregation kernel streams
       Built for a CSR in either direction: by target (rows = targets, cols = sources: forward and the target-side
       backward) or transposed (rows = sources, cols = targets: the source-side backward)."""
    __slots__ = ("rows", "sptr", "cols", "cell", "cellq", "nb", "n_rows", "max_cols", "avg_cols", "_acell")

    def to(self, device):
        p = BlockPlan()
        for k in ("rows", "sptr", "cols", "cell", "cellq"):
            setattr(p, k, getattr(self, k).to(device))
        p.nb, p.n_rows, p.max_cols, p.avg_cols = self.nb, self.n_rows, self.max_cols, self.avg_cols
        p._acell = {}
        return p

    def weight_image(self, H, owner=None):
        """The plan's dense weight image [chunks][H][hi | lo][512] (bf16), zero where a tile has no edge.  Allocated once
        per head count (and `owner`) and reused by every call on this plan: edges always overwrite the same cells, the rest
        stays 0.  owner: any hashable that tells apart the users whose images must stay alive side by side -- the
        source-side image is written by a layer's FORWARD pass and read by its backward pass, so two layers that share a
        graph (no per-layer tiers) each keep their own."""
        key = H if owner is None else (H, owner)
        img = self._acell.get(key)
        if img is None:
            img = torch.zeros(int(self.cols.numel()) // 16 * H * 1024, dtype=torch.bfloat16, device=self.cols.device)
            self._acell[key] = img
        return img


def build_block_plan(rowptr, col, n_cols, order=None, eid=None):
    """rowptr [n_rows + 1], col [E]: CSR by row; order: the rows in processing order (default 0..n_rows-1; pass a
    spatial order -- consecutive rows should share columns); eid [E]: alpha row of every CSR position (default: the
    position itself; the transposed CSR passes its eid_t).  Tensors on any device (the plan is built where they
    live); returns None if the CSR holds a duplicate (row, column) pair (the tile has one cell per pair)."""
    rowptr = torch.as_tensor(rowptr).long()
    col = torch.as_tensor(col).long()
    dev = col.device
    n_rows = rowptr.numel() - 1
    E = col.numel()
    deg = rowptr[1:] - rowptr[:-1]
    row_of = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg)
    if order is None:
        order = torch.arange(n_rows, device=dev)
    order = torch.as_tensor(order).long().to(dev)
    pos = torch.empty(n_rows, dtype=torch.long, device=dev)
    pos[order] = torch.arange(n_rows, device=dev)
    nb = (n_rows + PLAN_ROWS - 1) // PLAN_ROWS
    if torch.unique(row_of * n_cols + col).numel() != E:
        return None
    b_of = pos[row_of] // PLAN_ROWS
    r_of = pos[row_of] % PLAN_ROWS
    ukey, inv = torch.unique(b_of * n_cols + col, return_inverse=True)        # sorted by (block, column)
    ub, uc = ukey // n_cols, ukey % n_cols
    cnt = torch.bincount(ub, minlength=nb)
    pc = (cnt + PLAN_KPAD - 1) // PLAN_KPAD * PLAN_KPAD
    sptr = torch.zeros(nb + 1, dtype=torch.long, device=dev)
    sptr[1:] = torch.cumsum(pc, 0)
    first = torch.zeros(nb + 1, dtype=torch.long, device=dev)
    first[1:] = torch.cumsum(cnt, 0)
    slot = torch.arange(ukey.numel(), device=dev) - first[ub]
    gpos = sptr[ub] + slot
    last = uc[first[1:] - 1]                                                  # a valid column of every block (cnt >= 1)
    cols = torch.repeat_interleave(last, pc)
    cols[gpos] = uc
    rows = torch.full((nb * PLAN_ROWS,), -1, dtype=torch.long, device=dev)
    rows[:n_rows] = order
    egpos = gpos[inv]                                                         # global column position of every edge
    cell = torch.full((int(sptr[-1]) * PLAN_ROWS,), -1, dtype=torch.long, device=dev)
    eid = torch.arange(E, device=dev) if eid is None else torch.as_tensor(eid).long().to(dev)
    q, kk = egpos // 16, egpos % 16
    cell[q * (16 * PLAN_ROWS) + r_of * 16 + kk] = eid
    cellq = torch.empty(E, dtype=torch.long, device=dev)
    cellq[eid] = q * 512 + ((kk >> 3) * 32 + r_of) * 8 + (kk & 7)
    p = BlockPlan()
    p.rows, p.sptr, p.cols, p.cell, p.cellq = rows.int(), sptr.int(), cols.int(), cell.int(), cellq.int()
    p.nb, p.n_rows = int(nb), int(n_rows)
    p.max_cols, p.avg_cols = int(pc.max()), float(cnt.float().mean())
    p._acell = {}
    return p


def attach_plans(g, order_key=None):
    """Block plans of a BatchGraph for the matrix-core edge kernels (built on the device the graph lives on).
    order_key: one sortable value per node of the graph (e.g. the Morton key of its coordinates): blocks are runs of 32
    rows in that order, so that the rows of a block share columns.  Without it blocks follow the node numbering."""
    dev = g.col.device
    n, nt = g.n, g.n_tgt
    ot = os_ = None
    if order_key is not None:
        k = torch.as_tensor(np.asarray(order_key).astype(np.int64)).to(dev)
        ot = torch.argsort(k[:nt], stable=True)
        os_ = torch.argsort(k[:n], stable=True)
    g.plan_t = build_block_plan(g.rowptr, g.col, n, order=ot)
    g.plan_s = build_block_plan(g.rowptr_t, g.col_t, nt, order=os_, eid=g.eid_t)
    if g.plan_t is None or g.plan_s is None:
        g.plan_t = g.plan_s = None
    return g
