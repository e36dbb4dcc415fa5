"""Attention-diffusion conv operators with the reference's constructor / forward surface.

Drop-in for `GRAND_plusConv` (`src/GRAND_plus.py:40-347`), `GRAND_conv`
(`src/GRAND_plus.py:366-382`) and the stock `TransformerConv` that `get_conv(opt, 'TRANS', ...)` builds (`src/GNN.py:112-113`) in the configuration `get_conv` builds
(`src/GNN.py:115-119`): heads=1, root_weight=False, edge_dim=None, dropout=0,
identity value map.  Parameters and `state_dict` keys are the reference's
(`lin_key.{weight,bias}`, `lin_query.{weight,bias}`, `lin_skip.weight`, optional
`sm_temp_a`); the arithmetic runs in the fused HIP kernels through
`functional.grand_residual` - there is no torch/CPU implementation behind it.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from . import sparse_ops as Sp
from ._native import SUPPORTED_HIDDEN
from .graph import MeshGraph, graph_for


def _detached(alpha_t):
    """What `stored_alpha` keeps.  The reference stores the attention WITH its autograd graph (`src/GRAND_plus.py:253-256`,
    SURVEY appendix A: nothing consumes it), which keeps the whole previous step's graph - saved activations and the parameters'
    gradient accumulators, bound to the stream that step ran on - alive until the next forward overwrites it.  Besides the memory,
    that made a hipGraph capture on a side stream segfault inside hipStreamEndCapture after eager steps on the default stream
    (the stale accumulators pulled the default stream into the capture; tools/capture_bisect.sh).  The values are kept, the graph is not."""
    return alpha_t.detach() if torch.is_tensor(alpha_t) else alpha_t


class _AttentionDiffusionBase(nn.Module):
    def __init__(self, opt, in_channels, out_channels, heads=1, concat=False, beta=False, dropout=0.0,
                 edge_dim=None, bias=False, root_weight=False, skip_bias=False):
        super().__init__()
        if isinstance(in_channels, (tuple, list)):
            if in_channels[0] != in_channels[1]:
                raise NotImplementedError("bipartite in_channels are not used by the reference's get_conv")
            in_channels = in_channels[0]
        if in_channels != heads * out_channels:
            # value = Identity(x).view(-1, H, C) (GRAND_plus.py:150,227): the view needs in_channels == H C
            raise NotImplementedError(f"identity value map needs in_channels == heads * out_channels (src/GRAND_plus.py:150,227): "
                                      f"{in_channels} != {heads} * {out_channels}")
        if heads > 1 and not concat:
            raise ValueError("heads > 1 with concat=False: the reference's `out - x` (src/GRAND_plus.py:267) subtracts [N, H C] from "
                             "[N, C] - a shape error there too")
        self.opt = opt
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.root_weight = bool(root_weight)
        self.concat, self.beta = concat, bool(beta and root_weight)        # GRAND_plus.py:137
        self.dropout, self.edge_dim = float(dropout), edge_dim
        self.lin_key = nn.Linear(in_channels, heads * out_channels)        # GRAND_plus.py:146
        self.lin_query = nn.Linear(in_channels, heads * out_channels)      # GRAND_plus.py:147
        self.lin_value = nn.Identity()                                     # GRAND_plus.py:150
        skip_out = heads * out_channels if concat else out_channels        # GRAND_plus.py:168-181
        self.lin_skip = nn.Linear(in_channels, skip_out, bias=skip_bias)   # used only with root_weight (GRAND_plus.py:244-250)
        # GRAND_plus.py:165-168: a per-edge term on key and value (get_conv passes edge_dim=None, src/GNN.py:119)
        self.lin_edge = nn.Linear(edge_dim, heads * out_channels, bias=False) if edge_dim is not None else None
        self.lin_beta = nn.Linear(3 * skip_out, 1, bias=False) if self.beta else None
        # the fused kernels carry what get_conv builds: one head, no root term, no attention dropout
        self._general = heads > 1 or self.root_weight or self.dropout > 0.0 or edge_dim is not None
        self._alpha = None
        self._stored = None            # (graph, alpha in target-CSR order) of the last call
        self.stored_ei = None
        self.mesh_points = None
        self.mesh = None

    # -- attention of the last call, in the caller's edge order, shape [E, heads] like PyG
    @property
    def stored_alpha(self):
        if self._stored is None:
            return None
        graph, alpha_t = self._stored
        if alpha_t.dim() == 2:                                             # [H, E] of the general path -> [E, H]
            return graph.alpha_to_edge_order(alpha_t[:, :graph.num_edges]).t().contiguous()
        return graph.alpha_to_edge_order(alpha_t[:graph.num_edges]).unsqueeze(-1)

    @stored_alpha.setter
    def stored_alpha(self, value):
        self._stored = None if value is None else value

    def reset_parameters(self):
        self.lin_key.reset_parameters()
        self.lin_query.reset_parameters()
        self.lin_skip.reset_parameters()
        if self.lin_edge is not None:
            self.lin_edge.reset_parameters()
        if self.lin_beta is not None:
            self.lin_beta.reset_parameters()

    def _temperature(self):
        return None

    def _scale(self, device) -> torch.Tensor:
        s = torch.full((), 1.0 / math.sqrt(self.out_channels), device=device, dtype=torch.float32)   # GRAND_plus.py:279
        t = self._temperature()
        return s if t is None else s / t

    def _residual(self, x, edge_index, graph: Optional[MeshGraph], want_alpha: bool, edge_weight: Optional[torch.Tensor] = None,
                  edge_attr: Optional[torch.Tensor] = None):
        """A(x)x - x and the attention (target-CSR order).  The fused kernels carry the hidden sizes they are built for;
        any other width, and scores with a per-edge factor (`reg_skew`), take the same arithmetic through the generic
        primitives (Q/K projections as dense GEMMs, sddmm -> edge softmax -> spmm in HIP)."""
        if graph is None:
            graph = graph_for(edge_index, x.shape[0], x.device)
        if self._general or edge_attr is not None:
            return self._residual_general(x, graph, edge_weight, edge_attr)
        if edge_weight is None and x.shape[1] in SUPPORTED_HIDDEN:
            res, alpha_t = Fn.grand_residual(x, self.lin_query.weight, self.lin_query.bias, self.lin_key.weight,
                                             self.lin_key.bias, self._scale(x.device), graph, want_alpha)
            return res, alpha_t, graph
        q = torch.nn.functional.linear(x, self.lin_query.weight, self.lin_query.bias)      # GRAND_plus.py:225
        k = torch.nn.functional.linear(x, self.lin_key.weight, self.lin_key.bias)          # GRAND_plus.py:226
        m, alpha_t = Sp.attention_aggregate(graph, q, k, x, self._scale(x.device), edge_weight)
        return m - x, alpha_t, graph

    def _head_temperature(self, h: int):
        t = self._temperature()
        if torch.is_tensor(t) and t.numel() > 1:
            return t.reshape(-1)[h]
        return t

    def _residual_general(self, x, graph: MeshGraph, edge_weight: Optional[torch.Tensor], edge_attr: Optional[torch.Tensor] = None):
        """The constructor options get_conv never passes (`src/GRAND_plus.py:114-183,239-250,336`): H heads over slices of x
        (value = Identity(x).view(-1, H, C)), concatenation of the heads, `root_weight` / `beta`, attention dropout.  Per head the
        generic primitives (sddmm -> edge softmax -> spmm in HIP); projections, skip and gate are dense torch GEMMs."""
        n, hd, c = x.shape[0], self.heads, self.out_channels
        q = torch.nn.functional.linear(x, self.lin_query.weight, self.lin_query.bias).view(n, hd, c)    # GRAND_plus.py:225
        k = torch.nn.functional.linear(x, self.lin_key.weight, self.lin_key.bias).view(n, hd, c)        # :226
        v = x.view(n, hd, c)                                                                            # :227
        # edge features (GRAND_plus.py:273-277,338-340): caller's edge order -> the graph's target-CSR slots (a re-ordering of the
        # INPUT, once per call); lin_edge(edge_attr) joins the key and the value, a bare edge_attr (no lin_edge) the value only -
        # the reference tests `edge_attr is not None` at :339
        e_key = e_val = None
        if self.lin_edge is not None:
            assert edge_attr is not None                                                                # :274
        if edge_attr is not None:
            ea = edge_attr.to(x.dtype).index_select(0, graph.eid_t.long())
            if self.lin_edge is not None:
                e_key = e_val = self.lin_edge(ea).view(-1, hd, c)                                       # :275-277
            else:
                e_val = ea.reshape(-1, hd, c)
        pad = (-c) % 4                                                                                  # the per-edge-vector primitives take C % 4 == 0

        def _p(t):
            return torch.nn.functional.pad(t, (0, pad)).contiguous() if pad else t.contiguous()
        outs, alphas = [], []
        for h in range(hd):
            scale = 1.0 / math.sqrt(c)
            t = self._head_temperature(h)
            s_ = Sp.sddmm(graph, q[:, h].contiguous(), k[:, h].contiguous())
            if e_key is not None:
                s_ = s_ + Sp.edge_node_dot(graph, _p(q[:, h]), _p(e_key[:, h]))                         # <query_i, key_j + edge_e>
            s_ = s_ * scale                                                                             # :279
            if edge_weight is not None:
                s_ = s_ * edge_weight                                                                   # :324
            if t is not None:
                s_ = s_ / t                                                                             # :326-329
            a = Sp.edge_softmax(graph, s_)                                                              # :333
            alphas.append(a)
            a = torch.nn.functional.dropout(a, p=self.dropout, training=self.training)                  # :336
            o_h = Sp.spmm(graph, a, v[:, h].contiguous())                                               # :338-343, aggr='add'
            if e_val is not None:
                o_h = o_h + Sp.edge_weighted_rowsum(graph, a, _p(e_val[:, h]))[:, :c]                   # sum_e alpha_e edge_e
            outs.append(o_h)
        out = torch.cat(outs, dim=1) if self.concat else outs[0]                                        # :239-242 (mean over ONE head)
        if self.root_weight:                                                                            # :244-250
            x_r = self.lin_skip(x)
            if self.lin_beta is not None:
                beta = self.lin_beta(torch.cat([out, x_r, out - x_r], dim=-1)).sigmoid()
                out = beta * x_r + (1 - beta) * out
            else:
                out = out + x_r
        alpha_t = alphas[0] if hd == 1 else torch.stack(alphas)
        return out - x, alpha_t, graph                                                                  # :267

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads={self.heads})'


class GRAND_plusConv(_AttentionDiffusionBase):
    """`GRAND_plusConv(opt, in, out, global_feat_dim=..., heads=1, concat=False, ...)` (`src/GNN.py:118`)."""

    def __init__(self, opt, in_channels, out_channels, heads: int = 1, concat: bool = True, beta: bool = False,
                 dropout: float = 0., edge_dim: Optional[int] = None, bias: bool = True, root_weight: bool = True,
                 **kwargs):
        # extra kwargs (global_feat_dim, aggr, ...) are accepted and ignored, as MessagePassing.__init__ does
        super().__init__(opt, in_channels, out_channels, heads, concat, beta, dropout, edge_dim, bias, root_weight,
                         skip_bias=bias)
        self.dim = len(opt['mesh_dims'])
        t = opt.get('softmax_temp_type')
        if t == 'learnable_a':
            # the reference allocates torch.Tensor(1,heads,1) and never initialises it (GRAND_plus.py:154);
            # ones is the defined choice made here
            self.sm_temp_a = nn.Parameter(torch.ones(1, heads, 1))
        elif t == 'learnable_v':
            # GRAND_plus.py:158-160: Linear(in_channels, heads, bias=False); the parameter exists (state_dict key
            # `sm_temp_v.weight`), the forward cannot run - see forward()
            self.sm_temp_v = nn.Linear(in_channels, heads, bias=False)
        self._skew_cache = {}

    def _temperature(self):
        t = self.opt.get('softmax_temp_type')
        if t == 'fixed':
            return float(self.opt['softmax_temp'])                         # GRAND_plus.py:326-327
        if t == 'learnable_a':
            return self.sm_temp_a.reshape(()) if self.heads == 1 else self.sm_temp_a     # GRAND_plus.py:328-329 ([1,H,1]: one per head)
        return None

    def forward(self, x, edge_index, global_features=None, mesh=None, edge_attr=None,
                return_attention_weights=None, graph: Optional[MeshGraph] = None):
        if self.lin_edge is not None:
            assert edge_attr is not None                                   # GRAND_plus.py:274
        self.mesh_points, self.mesh = x, mesh                              # GRAND_plus.py:229-230
        if self.opt.get('softmax_temp_type') == 'learnable_v':
            # GRAND_plus.py:330-331 applies Linear(C, heads) to the [E, heads] score tensor and then .squeeze(2) a 2-D
            # tensor: a shape error for every hidden size (C != 1: matmul shapes; C == 1: squeeze dim out of range).
            # Same outcome here, said plainly.
            raise RuntimeError(f"softmax_temp_type='learnable_v': sm_temp_v = Linear({self.in_channels}, {self.heads}) cannot be applied "
                               f"to the [E, {self.heads}] attention scores (src/GRAND_plus.py:159,331) - the reference raises here too")
        store = isinstance(self.opt.get('show_mesh_evol_plots'), bool)     # GRAND_plus.py:253
        want_alpha = store or isinstance(return_attention_weights, bool)
        edge_weight = None
        if self.opt.get('reg_skew') and self.dim == 2:                     # GRAND_plus.py:280-324
            if graph is None:
                graph = graph_for(edge_index, x.shape[0], x.device)
            edge_weight = self._edge_area_sum(x, mesh, graph)
        res, alpha_t, graph = self._residual(x, edge_index, graph, want_alpha, edge_weight, edge_attr)
        if store:
            self.stored_ei, self._stored = edge_index, (graph, _detached(alpha_t))
        if isinstance(return_attention_weights, bool):                     # GRAND_plus.py:259-262
            alpha = (graph.alpha_to_edge_order(alpha_t[:, :graph.num_edges]).t().contiguous() if alpha_t.dim() == 2
                     else graph.alpha_to_edge_order(alpha_t[:graph.num_edges]).unsqueeze(-1))
            c = self.out_channels
            query = torch.nn.functional.linear(x, self.lin_query.weight, self.lin_query.bias).view(-1, self.heads, c)
            key = torch.nn.functional.linear(x, self.lin_key.weight, self.lin_key.bias).view(-1, self.heads, c)
            return res, (edge_index, (alpha, query, key))
        return res


    # ---- reg_skew: triangle-area weighted scores (GRAND_plus.py:280-324)
    def _edge_area_sum(self, x, mesh, graph: MeshGraph) -> torch.Tensor:
        """[E] in target-CSR order: for edge (a, b) the summed area of the triangles whose `cell_node_map` row [i,j,k] contains
        it as (i,j), (j,k) or (k,i) - 1 or 2 hits, else 0 (the reference's if/elif chain, :310-322) - with the areas taken
        from the first two columns of the layer's input (`self.mesh_points`, :229,283-288), differentiable.  The O(E*T)
        Python matching loop of the reference (:310-322) is a one-time hash join per topology here."""
        if mesh is None or not hasattr(mesh, 'coordinates'):
            raise ValueError("reg_skew needs the mesh (its coordinates.cell_node_map().values), as in src/GRAND_plus.py:281")
        cells = np.asarray(mesh.coordinates.cell_node_map().values).astype(np.int64)
        key = (id(graph), cells.shape, int(cells.sum()), cells.tobytes()[:64])
        ent = self._skew_cache.get(key)
        if ent is None:
            t = cells.shape[0]
            table = {}
            for col_a, col_b in ((0, 1), (1, 2), (2, 0)):              # tri_edges = [(i,j)] + [(j,k)] + [(k,i)], :295-299
                for tri, (a, b) in enumerate(zip(cells[:, col_a].tolist(), cells[:, col_b].tolist())):
                    table.setdefault((a, b), []).append(tri)
            ei = graph.edge_index.detach().cpu()
            t1 = np.full(ei.shape[1], -1, dtype=np.int64); t2 = np.full(ei.shape[1], -1, dtype=np.int64)
            for e, (a, b) in enumerate(zip(ei[0].tolist(), ei[1].tolist())):
                hits = table.get((a, b), ())
                if len(hits) == 1:
                    t1[e] = hits[0]
                elif len(hits) == 2:
                    t1[e], t2[e] = hits
            order = graph.eid_t.long().cpu().numpy()                   # target-CSR slot -> caller's edge id
            dev = x.device
            ent = (graph, torch.from_numpy(cells).to(dev),
                   torch.from_numpy(np.maximum(t1[order], 0)).to(dev), torch.from_numpy((t1[order] >= 0).astype(np.float32)).to(dev),
                   torch.from_numpy(np.maximum(t2[order], 0)).to(dev), torch.from_numpy((t2[order] >= 0).astype(np.float32)).to(dev))
            if len(self._skew_cache) > 4:
                self._skew_cache.clear()
            self._skew_cache[key] = ent
        _, cells_d, i1, m1, i2, m2 = ent
        px, py = x[:, 0], x[:, 1]
        xa, xb, xc = px[cells_d[:, 0]], px[cells_d[:, 1]], px[cells_d[:, 2]]
        ya, yb, yc = py[cells_d[:, 0]], py[cells_d[:, 1]], py[cells_d[:, 2]]
        area = 0.5 * torch.abs(xa * (yb - yc) + xb * (yc - ya) + xc * (ya - yb))        # :288
        return area[i1] * m1 + area[i2] * m2


class GRAND_conv(_AttentionDiffusionBase):
    """`GRAND_conv(opt, in, out, heads=1)` (`src/GNN.py:116`): TransformerConv with identity values,
    returning A x - x and always keeping (stored_ei, stored_alpha) (`src/GRAND_plus.py:380-382`)."""

    def __init__(self, opt, in_channels, out_channels, heads=1, concat=False, beta=False, dropout=0,
                 edge_dim=None, bias=False, root_weight=False):
        # the reference ignores these arguments and hard-codes the values below (GRAND_plus.py:371)
        super().__init__(opt, in_channels, out_channels, 1, False, False, 0.0, None, False, False, skip_bias=False)

    def forward(self, x, edge_index, graph: Optional[MeshGraph] = None):
        res, alpha_t, graph = self._residual(x, edge_index, graph, True)
        self.stored_ei, self._stored = edge_index, (graph, _detached(alpha_t))
        return res


class TRANS_conv(nn.Module):
    """`get_conv(opt, 'TRANS', in, out)` = PyG `TransformerConv(in, out, heads=1)` with its defaults (`src/GNN.py:112-113`):
    concat=True, beta=False, dropout=0, edge_dim=None, bias=True, root_weight=True, i.e.

        out_i = sum_j alpha_ij (W_v x_j + b_v) + W_s x_i + b_s,   alpha = softmax_j(<W_q x_i + b_q, W_k x_j + b_k> / sqrt(C)).

    The aggregation is linear in the values, so  sum_j alpha_ij (W_v x_j + b_v) = W_v m_i + (sum_j alpha_ij) b_v  with
    m = A(x) x: the graph part is exactly the GRAND residual op of the HIP kernels (m = residual + x), the value and skip
    projections are plain dense GEMMs (torch on the GPU: rocBLAS / hipBLASLt).  sum_j alpha_ij is 1 for a node with in-edges
    (up to the 1e-16 of PyG's softmax, below fp32 resolution) and 0 for a node without.  Parameters carry PyG's names
    (`lin_key`, `lin_query`, `lin_value`, `lin_skip`); hidden sizes are those of the fused kernels, in == out."""

    def __init__(self, opt, in_channels, out_channels, heads: int = 1):
        super().__init__()
        if heads != 1 or in_channels != out_channels:
            raise NotImplementedError("TRANS: get_conv builds heads=1 with in_dim == out_dim == hidden_dim (src/GNN.py:112-113,127-141)")
        self.opt = opt
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, 1
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_skip = nn.Linear(in_channels, out_channels, bias=True)

    def reset_parameters(self):
        for lin in (self.lin_key, self.lin_query, self.lin_value, self.lin_skip):
            lin.reset_parameters()

    def forward(self, x, edge_index, graph: Optional[MeshGraph] = None):
        if graph is None:
            graph = graph_for(edge_index, x.shape[0], x.device)
        scale = torch.full((), 1.0 / math.sqrt(self.out_channels), device=x.device, dtype=torch.float32)
        res, _ = Fn.grand_residual(x, self.lin_query.weight, self.lin_query.bias, self.lin_key.weight,
                                   self.lin_key.bias, scale, graph, False)
        m = res + x                                                    # A(x) x
        out = torch.nn.functional.linear(m, self.lin_value.weight) + graph.has_in * self.lin_value.bias
        return out + torch.nn.functional.linear(x, self.lin_skip.weight, self.lin_skip.bias)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads=1)'


def _glorot_(t: torch.Tensor) -> torch.Tensor:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))                 # torch_geometric.nn.inits.glorot
    with torch.no_grad():
        return t.uniform_(-a, a)


class _GATBase(nn.Module):
    """GATConv attention (PyG 2.4.0, heads=1, negative_slope=0.2, add_self_loops=True):
    alpha_ij = softmax_j(leaky_relu(<att_src, h_j> + <att_dst, h_i>)) over the edges j->i of the self-looped graph.
    Per-node sums and the leaky_relu are dense torch ops; gather / softmax / aggregation are the HIP primitives."""

    def __init__(self, opt, in_channels, out_channels):
        super().__init__()
        if in_channels != out_channels:
            raise NotImplementedError("get_conv builds in_dim == out_dim == hidden_dim (src/GNN.py:127-141)")
        self.opt = opt
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, 1
        self.negative_slope, self.add_self_loops = 0.2, True
        self.att_src = nn.Parameter(_glorot_(torch.empty(1, 1, out_channels)))
        self.att_dst = nn.Parameter(_glorot_(torch.empty(1, 1, out_channels)))
        self.stored_ei = None
        self._stored = None

    @property
    def stored_alpha(self):
        if self._stored is None:
            return None
        graph, alpha_t = self._stored
        return graph.alpha_to_edge_order(alpha_t).unsqueeze(-1)

    @stored_alpha.setter
    def stored_alpha(self, value):
        self._stored = None if value is None else value

    def _attention(self, h, edge_index, graph: Optional[MeshGraph]):
        if graph is None:
            graph = graph_for(edge_index, h.shape[0], h.device)
        looped = graph.with_self_loops()
        a_src = (h * self.att_src.view(1, -1)).sum(-1)
        a_dst = (h * self.att_dst.view(1, -1)).sum(-1)
        score = torch.nn.functional.leaky_relu(Sp.edge_add(looped, a_src, a_dst), self.negative_slope)
        return looped, Sp.edge_softmax(looped, score)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads=1)'


class GAT_plus(_GATBase):
    """`GAT_plus(opt, in, out)` (`src/GNN.py:120-121`, `src/GRAND_plus.py:386-416`): GATConv attention with identity source /
    target maps; the conv's own output is discarded and the attention re-applied as a sparse matrix,
    `Ax = sparse(alpha)^T x`; returns `Ax - x` (`gat_plus_type='GAT_res_lap'`) or `Ax` (`'GAT_lin'`).  Keeps
    `stored_ei` (the self-looped edge list GATConv returns) and `stored_alpha`.  Parameters: att_src, att_dst (bias=False)."""

    def __init__(self, opt, in_channels, out_channels, heads=1, concat=False, beta=False, dropout=0, edge_dim=None, bias=False,
                 root_weight=False):
        super().__init__(opt, in_channels, out_channels)
        self.lin_src = nn.Identity()                                   # GRAND_plus.py:394-395
        self.lin_dst = nn.Identity()

    def forward(self, x, edge_index, graph: Optional[MeshGraph] = None):
        kind = self.opt['gat_plus_type']
        if kind not in ('GAT_res_lap', 'GAT_lin'):
            # the reference's forward has no branch for the other declared choices (params.py:272) and returns None
            raise NotImplementedError(f"gat_plus_type={kind!r}: src/GRAND_plus.py:400-416 implements 'GAT_res_lap' and 'GAT_lin' only")
        looped, alpha_t = self._attention(x, edge_index, graph)
        self.stored_ei, self._stored = looped.edge_index, (looped, _detached(alpha_t))
        ax = Sp.spmm(looped, alpha_t, x)
        return ax - x if kind == 'GAT_res_lap' else ax


class GAT_conv(_GATBase):
    """`get_conv(opt, 'GAT', in, out)` = PyG `GATConv(in, out, heads=1)` (`src/GNN.py:110-111`) with its defaults:
    out_i = sum_j alpha_ij W x_j + bias.  PyG's parameter names: lin_src (= lin_dst), att_src, att_dst, bias."""

    def __init__(self, opt, in_channels, out_channels, heads: int = 1):
        if heads != 1:
            raise NotImplementedError("get_conv builds heads=1 (src/GNN.py:111)")
        super().__init__(opt, in_channels, out_channels)
        self.lin_src = nn.Linear(in_channels, out_channels, bias=False)
        _glorot_(self.lin_src.weight)
        self.lin_dst = self.lin_src
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, graph: Optional[MeshGraph] = None):
        h = self.lin_src(x)
        looped, alpha_t = self._attention(h, edge_index, graph)
        return Sp.spmm(looped, alpha_t, h) + self.bias


class GCN_conv(nn.Module):
    """`get_conv(opt, 'GCN', in, out)` = PyG `GCNConv(in, out)` (`src/GNN.py:109-110`): gcn_norm with
    `add_remaining_self_loops`, out = D^-1/2 (A + I) D^-1/2 (x W^T) + bias, degrees by target.  Parameters `lin.weight`, `bias`."""

    def __init__(self, opt, in_channels, out_channels):
        super().__init__()
        self.opt = opt
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        _glorot_(self.lin.weight)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, graph: Optional[MeshGraph] = None):
        if graph is None:
            graph = graph_for(edge_index, x.shape[0], x.device)
        looped = graph.with_self_loops()
        norm = getattr(looped, '_gcn_norm', None)
        if norm is None:                                               # constant of the topology: once per graph
            dis = Sp.in_degree(looped).pow(-0.5)
            dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
            norm = looped._gcn_norm = Sp.edge_mul(looped, dis, dis).detach()
        return Sp.spmm(looped, norm, self.lin(x)) + self.bias

    def __repr__(self):
        return f'{self.__class__.__name__}({self.in_channels}, {self.out_channels})'
