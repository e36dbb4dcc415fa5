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

import torch
import torch.nn as nn

from . import functional as Fn
from .graph import MeshGraph, graph_for


class _AttentionDiffusionBase(nn.Module):
    def __init__(self, opt, in_channels, out_channels, heads=1, concat=False, beta=False, dropout=0.0,
                 edge_dim=None, bias=False, root_weight=False, skip_bias=False):
        super().__init__()
        if isinstance(in_channels, (tuple, list)):
            if in_channels[0] != in_channels[1]:
                raise NotImplementedError("bipartite in_channels are not used by the reference's get_conv")
            in_channels = in_channels[0]
        if heads != 1:
            raise NotImplementedError("heads != 1: get_conv always passes heads=1 (src/GNN.py:116-118)")
        if root_weight or beta:
            raise NotImplementedError("root_weight/beta: get_conv passes root_weight=False (src/GNN.py:118-119)")
        if edge_dim is not None:
            raise NotImplementedError("edge_dim: get_conv passes edge_dim=None (src/GNN.py:119)")
        if dropout:
            raise NotImplementedError("attention dropout: get_conv passes dropout=0.0 (src/GNN.py:118)")
        if in_channels != out_channels:
            raise NotImplementedError("identity value map needs in_channels == out_channels (src/GRAND_plus.py:150)")
        self.opt = opt
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.beta, self.root_weight = concat, False, False
        self.dropout, self.edge_dim = 0.0, None
        self.lin_key = nn.Linear(in_channels, heads * out_channels)        # GRAND_plus.py:146
        self.lin_query = nn.Linear(in_channels, heads * out_channels)      # GRAND_plus.py:147
        self.lin_value = nn.Identity()                                     # GRAND_plus.py:150
        self.lin_skip = nn.Linear(in_channels, out_channels, bias=skip_bias)   # allocated, unused (GRAND_plus.py:178)
        self.lin_edge = None
        self.lin_beta = None
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
        return graph.alpha_to_edge_order(alpha_t[:graph.num_edges]).unsqueeze(-1)

    @stored_alpha.setter
    def stored_alpha(self, value):
        self._stored = None if value is None else value

    def reset_parameters(self):
        self.lin_key.reset_parameters()
        self.lin_query.reset_parameters()
        self.lin_skip.reset_parameters()

    def _temperature(self):
        return None

    def _scale(self, device) -> torch.Tensor:
        s = torch.full((), 1.0 / math.sqrt(self.out_channels), device=device, dtype=torch.float32)   # GRAND_plus.py:279
        t = self._temperature()
        return s if t is None else s / t

    def _residual(self, x, edge_index, graph: Optional[MeshGraph], want_alpha: bool):
        if graph is None:
            graph = graph_for(edge_index, x.shape[0], x.device)
        res, alpha_t = Fn.grand_residual(x, self.lin_query.weight, self.lin_query.bias, self.lin_key.weight,
                                         self.lin_key.bias, self._scale(x.device), graph, want_alpha)
        return res, alpha_t, graph

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
            raise NotImplementedError("softmax_temp_type='learnable_v' (src/GRAND_plus.py:158-160,330-331)")
        if opt.get('reg_skew') and self.dim == 2:
            raise NotImplementedError("reg_skew triangle-area weighting (src/GRAND_plus.py:280-324)")

    def _temperature(self):
        t = self.opt.get('softmax_temp_type')
        if t == 'fixed':
            return float(self.opt['softmax_temp'])                         # GRAND_plus.py:326-327
        if t == 'learnable_a':
            return self.sm_temp_a.reshape(())                              # GRAND_plus.py:328-329
        return None

    def forward(self, x, edge_index, global_features=None, mesh=None, edge_attr=None,
                return_attention_weights=None, graph: Optional[MeshGraph] = None):
        assert edge_attr is None, "edge_attr needs lin_edge (edge_dim), which get_conv never builds"
        self.mesh_points, self.mesh = x, mesh                              # GRAND_plus.py:229-230
        store = isinstance(self.opt.get('show_mesh_evol_plots'), bool)     # GRAND_plus.py:253
        want_alpha = store or isinstance(return_attention_weights, bool)
        res, alpha_t, graph = self._residual(x, edge_index, graph, want_alpha)
        if store:
            self.stored_ei, self._stored = edge_index, (graph, alpha_t)
        if isinstance(return_attention_weights, bool):                     # GRAND_plus.py:259-262
            alpha = graph.alpha_to_edge_order(alpha_t[:graph.num_edges]).unsqueeze(-1)
            c = self.out_channels
            query = torch.nn.functional.linear(x, self.lin_query.weight, self.lin_query.bias).view(-1, 1, c)
            key = torch.nn.functional.linear(x, self.lin_key.weight, self.lin_key.bias).view(-1, 1, c)
            return res, (edge_index, (alpha, query, key))
        return res


class GRAND_conv(_AttentionDiffusionBase):
    """`GRAND_conv(opt, in, out, heads=1)` (`src/GNN.py:116`): TransformerConv with identity values,
    returning A x - x and always keeping (stored_ei, stored_alpha) (`src/GRAND_plus.py:380-382`)."""

    def __init__(self, opt, in_channels, out_channels, heads=1, concat=False, beta=False, dropout=0,
                 edge_dim=None, bias=False, root_weight=False):
        # the reference ignores these arguments and hard-codes the values below (GRAND_plus.py:371)
        super().__init__(opt, in_channels, out_channels, 1, False, False, 0.0, None, False, False, skip_bias=False)

    def forward(self, x, edge_index, graph: Optional[MeshGraph] = None):
        res, alpha_t, graph = self._residual(x, edge_index, graph, True)
        self.stored_ei, self._stored = edge_index, (graph, alpha_t)
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
