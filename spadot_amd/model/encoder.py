"""Encoders: mirror of /root/reference/SpaDOT/model/encoder.py (same class names, constructor
arguments and state_dict keys).

GATConv here is this package's own layer (torch_geometric is not a dependency): the dense map
x -> h = x W^T is a library GEMM on MFMA; everything after it -- attention logits h . att, scatter-softmax
over incoming edges, weighted scatter-add, bias, activation, head concat/mean -- is hand-written HIP
(spadot_amd.ops.gat_edge: two launches forward, four backward, no atomics).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graph import build_batch_graph
from ..ops import (stamp_if, BatchGraph, bn_act, cast_rows, dense_cd, encoder_mid, encoder_mid_ok, first_map_seeds, first_map_seeds_ok,
                   gat_edge, gat_tail, gat_tail_ok, head_fc, head_fc_ok, hidden_map, linear_bias, weight_image)


class SVGPEncoder(nn.Module):
    """encoder.py:7-34: [Linear, BatchNorm1d, LeakyReLU] per hidden size, then Linear -> (mu, var)."""

    def __init__(self, input_dim, SVGP_z_dim, hidden_dims, compute_dtype=torch.float32):
        super().__init__()
        self.compute_dtype = compute_dtype
        layers = [input_dim] + list(hidden_dims)
        net = []
        for i in range(1, len(layers)):
            lin = nn.Linear(layers[i - 1], layers[i])
            nn.init.xavier_uniform_(lin.weight)
            net += [lin, nn.BatchNorm1d(layers[i]), nn.LeakyReLU()]
        self.SVGP_encoder_net = nn.Sequential(*net)
        self.SVGP_fc = nn.Linear(hidden_dims[-1], SVGP_z_dim * 2)
        nn.init.xavier_uniform_(self.SVGP_fc.weight)

    def forward(self, x):
        mu, logvar = torch.chunk(self.pre_head(x), 2, dim=1)
        return mu, torch.exp(logvar)

    def pre_head(self, x, x_bf16=None, defer_fc=False):
        """SVGP_fc output (mu | logvar) [b, 2 z].  x_bf16 (optional, training in the bf16 compute dtype): the bf16 image
        [b, G padded] of the same rows -- the first map's weight gradient is then taken from it on the matrix cores
        (ops.first_map_seeds), the forward stays the fp32 product of x.
        defer_fc (training, round 5): the stages behind the first map run as two launches that leave SVGP_fc as partial
        products (ops.encoder_mid); with defer_fc the result tensor is NOT filled here -- it carries `_enc_partials` =
        (partials, their number, SVGP_fc's bias) and the SVGP stage's first kernel sums them, storing the result on the way
        (svgp.elbo_start(partials=...)).  Only callers that hand the result straight to that stage may ask for it."""
        net = list(self.SVGP_encoder_net)
        fused = self.training                          # eval mode (running statistics) takes the library modules
        if not fused:
            x = x[:, :net[0].in_features]
            if self.compute_dtype == torch.float32:
                h = self.SVGP_encoder_net(x.float())
            else:
                first, cd = net[0], self.compute_dtype
                h = F.linear(x.to(cd), first.weight.to(cd), first.bias.to(cd)).float()
                for layer in net[1:]:
                    h = layer(h)
            return self.SVGP_fc(h)
        # training: per hidden size  Linear (bias folded into the next kernel) -> BatchNorm + LeakyReLU in ONE launch;
        # the G -> hidden map is the only large GEMM of this branch: compute dtype on MFMA (fp32 accumulate)
        h = x
        start = 0
        if len(net) == 6 and h.dtype == torch.float32 and h.is_cuda:
            # two hidden stages on fp32 rows (the model's shape): first map, then -- where csrc/enc_fused.hip takes the sizes --
            # everything up to SVGP_fc in two launches
            lin, bn, act = net[0], net[1], net[2]
            if self.compute_dtype == torch.bfloat16 and first_map_seeds_ok(h, lin.weight, x_bf16):
                h1 = first_map_seeds(h, lin.weight, x_bf16)
            else:
                h1 = F.linear(h[:, :lin.in_features], lin.weight)
            stamp_if(19)
            if encoder_mid_ok(h1, net[3].weight, self.SVGP_fc.weight):
                z, pz, npz = encoder_mid(h1, lin.bias, bn, act.negative_slope, net[3].weight, net[3].bias, net[4],
                                         net[5].negative_slope, self.SVGP_fc.weight, self.SVGP_fc.bias, fill=not defer_fc)
                if defer_fc:
                    z._enc_partials = (pz, npz, self.SVGP_fc.bias.detach())
                return z
            h = bn_act(h1, lin.bias, bn, act.negative_slope)
            stamp_if(20)
            start = 3
        for i in range(start, len(net), 3):
            lin, bn, act = net[i], net[i + 1], net[i + 2]
            if i == 0 and self.compute_dtype != torch.float32 and h.dtype != torch.float32:
                h = dense_cd(h.to(self.compute_dtype), lin.weight, lin)
            elif i == 0 and self.compute_dtype == torch.bfloat16 and first_map_seeds_ok(h, lin.weight, x_bf16):
                h = first_map_seeds(h, lin.weight, x_bf16)
            elif i > 0 and h.dtype == torch.float32 and h.shape[1] == lin.in_features:
                h = hidden_map(h, lin.weight)              # (same product; its backward dodges a library tile that stalls)
            else:
                h = F.linear(h[:, :lin.in_features].float(), lin.weight)
            stamp_if(19 + i // 3 * 2)                      # (SPADOT_STAMPS=1 only: slots 19 / 21 behind the maps, 20 / 22 behind BN)
            h = bn_act(h, lin.bias, bn, act.negative_slope)
            stamp_if(20 + i // 3 * 2)
        return linear_bias(h, self.SVGP_fc.weight, self.SVGP_fc.bias)


class GATConv(nn.Module):
    """Graph attention layer with torch_geometric.nn.GATConv's parameters (lin.weight [H*C, in] without
    bias, att_src/att_dst [1, H, C], bias [H*C] or [C]) and forward semantics:
    negative_slope 0.2, self loops re-added, softmax over incoming edges, concat or head mean.

    forward(x, graph, act=False): `graph` is a BatchGraph (CSR both ways, on the device) or an
    edge_index tensor [2, E] (converted on the fly).  `act` fuses the following leaky_relu(0.01) of
    encoder.py:56-57 into the kernel's epilogue.  `compute_dtype` (fp32 or bf16) is the dtype of h and
    of the layer's output; parameters stay fp32."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, compute_dtype=torch.float32):
        super().__init__()
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, heads, concat
        self.compute_dtype = compute_dtype
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels if concat else out_channels))
        # PyG's reset_parameters: glorot on lin and on the attention vectors, zeros on bias.  Its glorot() draws from
        # U(-s, s) with s = sqrt(6 / (size(-2) + size(-1))): for att_* [1, H, C] that is sqrt(6 / (H + C)) -- NOT
        # nn.init.xavier_uniform_'s fan computation on a 3-D tensor (which would give sqrt(6 / (H C + C))).
        # (encoder.py:42-46 of the reference then re-draws lin.weight with xavier_uniform_: same distribution.)
        nn.init.xavier_uniform_(self.lin.weight)
        bound = (6.0 / (heads + out_channels)) ** 0.5
        nn.init.uniform_(self.att_src, -bound, bound)
        nn.init.uniform_(self.att_dst, -bound, bound)

    def forward(self, x, graph, act=False, fresh=False, taps=None, tap=None, in_cell=None, act_cell=None):
        """taps / tap (optional): the dense map's output is stored as taps[tap] -- where a backward pass issued in pieces cuts
        between this layer's edge phase and its dense map (GraphedStepper `defer_wgrad`)."""
        # head mean over few targets (the encoder's last layer for the seeds): aggregate first, map the n_tgt aggregated
        # rows instead of all source rows (ops.gat_tail; the same function, ~15x fewer flops at the benchmarked shape)
        if (not act and isinstance(graph, BatchGraph) and x.dtype == self.compute_dtype
                and gat_tail_ok(x, self.lin.weight, graph, self.heads, self.out_channels, self.concat)):
            wimg = None
            if self.compute_dtype != torch.float32:
                wimg = weight_image(self.lin.weight, x.shape[1], x.dtype, self)
                if not fresh:
                    wimg[:, :self.in_channels].copy_(self.lin.weight.detach())
            return gat_tail(x, self.lin.weight, wimg, self.att_src, self.att_dst, self.bias, graph, self.heads, self.out_channels,
                            in_cell=in_cell)
        d = self.dense(x, fresh, in_cell=in_cell)
        if taps is not None and tap is not None:
            taps[tap] = d
        return self.edge(d, graph, act, act_cell=act_cell)

    def dense(self, x, fresh=False, in_cell=None):
        """h = x W^T  [n, H*C] (MFMA GEMM in the compute dtype; x may be K-padded).  fresh: the compute-dtype image
        of the weight is already current (GATEncoder casts the three layers' weights in one launch)."""
        cd = self.compute_dtype
        if cd == torch.float32:
            return F.linear(x[:, :self.in_channels].float(), self.lin.weight)
        return dense_cd(x.to(cd), self.lin.weight, self, fresh=fresh, defer=getattr(self, "defer_wgrad", False), in_cell=in_cell)

    def edge(self, h, graph, act=False, act_cell=None):
        """Everything after the dense map (ops.gat_edge)."""
        if not isinstance(graph, BatchGraph):
            graph = build_batch_graph(graph, h.shape[0], h.device)
        return gat_edge(h, self.att_src, self.att_dst, self.bias, graph, self.heads, self.out_channels, self.concat, act,
                        defer=getattr(self, "defer_wgrad", False), act_cell=act_cell)


class GATEncoder(nn.Module):
    """encoder.py:37-61."""

    def __init__(self, input_dim, GAT_z_dim, hidden_dim=512, num_heads=4, compute_dtype=torch.float32):
        super().__init__()
        self.gat1 = GATConv(input_dim, hidden_dim, heads=num_heads, concat=True, compute_dtype=compute_dtype)
        self.gat2 = GATConv(hidden_dim * num_heads, hidden_dim, heads=num_heads, concat=True, compute_dtype=compute_dtype)
        self.gat3 = GATConv(hidden_dim * num_heads, hidden_dim, heads=num_heads, concat=False, compute_dtype=compute_dtype)
        self.GAT_fc = nn.Linear(hidden_dim, GAT_z_dim * 2)
        nn.init.xavier_uniform_(self.GAT_fc.weight)

    def above_second_dense(self):
        """The parameters whose gradients a backward pass has produced when it reaches the second layer's dense map: the head,
        layer 3 and the second layer's edge-phase parameters (everything but gat2.lin.weight and layer 1)."""
        return (list(self.GAT_fc.parameters()) + list(self.gat3.parameters())
                + [self.gat2.att_src, self.gat2.att_dst, self.gat2.bias])

    def first_layer_parameters(self):
        """The parameters whose gradients a backward pass produces last (ops.FlatAdamW `last`)."""
        return list(self.gat1.parameters())

    def forward(self, x, edge_index, rows=None):
        """`rows` (optional int): only the first `rows` nodes of the output are needed (the seeds)."""
        mu, logvar = torch.chunk(self.pre_head(x, edge_index, rows), 2, dim=1)
        return mu, torch.exp(logvar)

    def pre_head(self, x, edge_index, rows=None, after_first_dense=None, taps=None):
        """GAT_fc output (mu | logvar) [rows or n, 2 z]: what ops.latent_head consumes.
        after_first_dense: optional callable run right after the first (largest) GEMM has been issued -- the
        composite model issues the start of its SVGP branch there, on another stream.
        taps: optional dict; receives 'd2', the second layer's dense output (where a backward pass issued in pieces cuts:
        everything above the tap is differentiated first, what is below it afterwards)."""
        lg = getattr(edge_index, "layer_graphs", None) if rows is not None else None
        g3 = getattr(edge_index, "seed_graph", None) if rows is not None else None
        if not isinstance(edge_index, BatchGraph):
            edge_index = build_batch_graph(edge_index, x.shape[0], x.device)
        # compute-dtype images of the three layers' weights: one cast launch for all of them
        fresh = False
        if self.gat1.compute_dtype == torch.bfloat16 and x.is_cuda:
            HC = self.gat2.in_channels
            pairs = [(self.gat1.lin.weight, weight_image(self.gat1.lin.weight, x.shape[1], torch.bfloat16, self.gat1)),
                     (self.gat2.lin.weight, weight_image(self.gat2.lin.weight, HC, torch.bfloat16, self.gat2)),
                     (self.gat3.lin.weight, weight_image(self.gat3.lin.weight, HC, torch.bfloat16, self.gat3))]
            # training under an optimizer that keeps the images current (ops.FlatAdamW.maintain_image: the update kernel
            # stores the bf16 copy of every new weight): no cast launch at the head of the step
            # -- but only while that optimizer still owns the weights (a second FlatAdamW over this model re-points them:
            # the pin is dropped) and nothing else has written them since the images were cast (sync_images)
            opt = getattr(self, "_image_optimizer", None)
            if opt is not None and not all(opt.owns(W) for W, _ in pairs):
                object.__setattr__(self, "_image_optimizer", None)
                opt = None
            if opt is None or not self.training or not all(opt.maintain_image(W, im) for W, im in pairs):
                cast_rows([(W.detach(), im) for W, im in pairs])
            else:
                opt.sync_images()
            fresh = True
        h = self.gat1.dense(x, fresh)
        if after_first_dense is not None:
            after_first_dense()
        # c1 / c2: one-step mailboxes between a layer's fused activation and the op that consumes its output (ops._GATEdgeMFMA:
        # the consumer hands the gradient back already multiplied by the activation's derivative)
        c1, c2 = ({}, {}) if self.training else (None, None)
        h = self.gat1.edge(h, edge_index, act=True, act_cell=c1)
        if lg is not None and lg[1].n_tgt == rows:
            # only what the seeds' rows of layer 3 depend on: layer 2 for seeds + hop 1, layer 3 for the seeds
            h = self.gat2(h, lg[0], act=True, fresh=fresh, taps=taps, tap="d2", in_cell=c1, act_cell=c2)
            h = self.gat3(h, lg[1], act=False, fresh=fresh, in_cell=c2)
        elif g3 is not None and g3.n_tgt == rows:
            h = self.gat2(h, edge_index, act=True, fresh=fresh, taps=taps, tap="d2")
            h = self.gat3(h, g3, act=False, fresh=fresh)          # edge phase for the seeds only: same rows, ~n/rows less work
        else:
            h = self.gat2(h, edge_index, act=True, fresh=fresh, taps=taps, tap="d2")
            h = self.gat3(h, edge_index, act=False, fresh=fresh)
            if rows is not None:
                h = h[:rows]
        if head_fc_ok(h, self.GAT_fc.weight, self.GAT_fc.bias):
            return head_fc(h, self.GAT_fc.weight, self.GAT_fc.bias)              # bf16 rows in, fp32 out: no cast launches
        return linear_bias(h.float(), self.GAT_fc.weight, self.GAT_fc.bias)
