"""ORACLE - test infrastructure only, never on the product path.  PARITY UNPINNED.

CPU restatement, in plain PyTorch ops, of the reference's message-passing hot
path, executing the op sequence `torch_geometric==2.4.0` (pinned,
`README.md:25`; NOT vendored under /root/reference and not installable here)
would execute.  The reference has no tests, golden vectors or fixtures for
this path and cannot be imported (ModuleNotFoundError:
torch_geometric / torch_scatter / firedrake), so this oracle is pinned only by
(i) an independent dense fp64 formulation (`oracle/dense_check.py`),
(ii) `torch.autograd.gradcheck`, (iii) analytic known answers - see
`tests/test_oracle.py`.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package.

What each function follows (paths under /root/reference):

* `pyg_softmax`            - `torch_geometric.utils.softmax` as called at
                             `src/GRAND_plus.py:333` (and `:35-37` with temperature):
                             scatter-max (detached) -> sub -> exp -> scatter-sum
                             + 1e-16 -> gather -> div.
* `grand_residual`         - `GRAND_plusConv.forward/.message`
                             (`src/GRAND_plus.py:204-267`, `:269-343`) with the
                             arguments `get_conv` passes (`src/GNN.py:117-119`):
                             heads=1, concat=False, root_weight=False, edge_dim=None,
                             dropout=0; identical maths to `GRAND_conv`
                             (`src/GRAND_plus.py:366-382`).
* `masked_edge_index`      - `GNN.forward` edge surgery (`src/GNN.py:206-218`).
* `node_features`          - feature concat (`src/GNN.py:225-239`).
* `identity_encoder_weight`- `get_enc` identity branch (`src/GNN.py:75-90`).
* `transformer_conv`       - PyG `TransformerConv(in, out, heads=1)` defaults, the
                             'TRANS' branch of `get_conv` (`src/GNN.py:112-113`).
* `gat_attention` / `gat_conv` / `gat_plus` - PyG 2.4.0 `GATConv(in, out, heads=1)` (`src/GNN.py:110-111`) and the
                             reference's `GAT_plus(GATConv)` with identity source / target maps
                             (`src/GRAND_plus.py:386-416`): additive score leaky_relu(a_s.x_j + a_d.x_i, 0.2),
                             `remove_self_loops` + `add_self_loops` (GATConv default), softmax by target.
* `gcn_conv`               - PyG 2.4.0 `GCNConv(in, out)` (`src/GNN.py:109-110`): gcn_norm with
                             `add_remaining_self_loops`, D^-1/2 (A+I) D^-1/2 (x W^T) + b.
* `triangle_edge_area_sum` - the `reg_skew` weights of `src/GRAND_plus.py:280-324`.
* `OracleGNN`              - `GNN.__init__/forward` (`src/GNN.py:144-306`) for
                             loss_type in {mesh_loss, modular}, enc='identity',
                             conv_type in {GRAND_plus, GRAND, TRANS, GAT_plus, GAT, GCN}.
"""
from __future__ import annotations

import math
import time
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


def pyg_softmax(src: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """Segment softmax over entries sharing `index` (PyG 2.4.0 `utils.softmax`)."""
    shape = [num_nodes] + list(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    src_max = torch.full(shape, float('-inf'), dtype=src.dtype, device=src.device)
    src_max = src_max.scatter_reduce(0, idx, src.detach(), reduce='amax', include_self=True)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = torch.zeros(shape, dtype=src.dtype, device=src.device).index_add_(0, index, out) + 1e-16
    return out / out_sum.index_select(0, index)


def grand_residual(x, edge_index, w_query, b_query, w_key, b_key,
                   temperature: Optional[torch.Tensor | float] = None, return_attention: bool = False,
                   edge_area_sum: Optional[torch.Tensor] = None):
    """One diffusion residual  A(x)x - x  (`src/GRAND_plus.py:225-267`).

    x [N,C]; edge_index [2,E] int64, row 0 = source j, row 1 = target i.
    """
    n, c = x.shape
    src, dst = edge_index[0], edge_index[1]
    query = F.linear(x, w_query, b_query).view(-1, 1, c)        # :225
    key = F.linear(x, w_key, b_key).view(-1, 1, c)              # :226
    value = x.view(-1, 1, c)                                    # :150,:227 identity
    query_i = query.index_select(0, dst)                        # propagate/_collect: _i <- edge_index[1]
    key_j = key.index_select(0, src)                            #                     _j <- edge_index[0]
    value_j = value.index_select(0, src)
    alpha = (query_i * key_j).sum(dim=-1) / math.sqrt(c)        # :279
    if edge_area_sum is not None:
        alpha = alpha * edge_area_sum.unsqueeze(-1)             # reg_skew :324
    if temperature is not None:
        alpha = alpha / temperature                             # :35-37,:326-329
    alpha = pyg_softmax(alpha, dst, n)                          # :333
    msg = value_j * alpha.view(-1, 1, 1)                        # :342
    out = torch.zeros(n, 1, c, dtype=x.dtype, device=x.device).index_add_(0, dst, msg)   # aggr='add' :127
    out = out.mean(dim=1)                                       # :242 (heads=1)
    res = out - x                                               # :267
    if return_attention:
        return res, (alpha, query, key)
    return res


def grand_plus_general(x, edge_index, w_query, b_query, w_key, b_key, heads: int = 1, concat: bool = True,
                       w_skip=None, b_skip=None, w_beta=None, temperature=None, dropout_mask=None,
                       return_attention: bool = False, w_edge=None, edge_attr=None):
    """`GRAND_plusConv.forward` with the constructor options `get_conv` never passes (`src/GRAND_plus.py:114-183,204-267,
    269-343`): H heads (value = Identity(x).view(-1, H, C): needs in_channels = H C), concat / mean over the heads,
    `root_weight` (`w_skip`, `b_skip` given: `lin_skip`), `beta` (`w_beta` given: `lin_beta`), attention dropout as an explicit
    per-edge, per-head mask of 0 / 1/(1-p) factors (`dropout_mask [E,H]`: what F.dropout multiplies with, :336).
    `temperature`: scalar or [1,H,1] (`sm_temp_a`).  `w_edge` [H C, edge_dim] (`lin_edge`, :165-166) with `edge_attr [E, edge_dim]`:
    the projected edge features are added to the key (:273-277) and to the value (:338-340); `edge_attr` WITHOUT `w_edge` is added
    to the value as it is (:339 tests `edge_attr is not None`, not `lin_edge`), so it must already be [E, H C].  Returns out - x (:267)."""
    n = x.shape[0]
    c = w_query.shape[0] // heads
    src, dst = edge_index[0], edge_index[1]
    query = F.linear(x, w_query, b_query).view(-1, heads, c)     # :225
    key = F.linear(x, w_key, b_key).view(-1, heads, c)           # :226
    value = x.view(-1, heads, c)                                 # :150,:227
    key_j = key.index_select(0, src)
    if w_edge is not None:
        assert edge_attr is not None                             # :274
        edge_attr = F.linear(edge_attr, w_edge).view(-1, heads, c)   # :275-276
        key_j = key_j + edge_attr                                # :277
    alpha = (query.index_select(0, dst) * key_j).sum(dim=-1) / math.sqrt(c)     # [E,H] :279
    if temperature is not None:
        t = temperature if not torch.is_tensor(temperature) else (temperature.squeeze(2) if temperature.dim() == 3 else temperature)
        alpha = alpha / t                                        # :35-37,:326-329 (sm_temp_a.squeeze(2): [1,H])
    alpha = pyg_softmax(alpha, dst, n)                           # :333
    kept = alpha
    if dropout_mask is not None:
        alpha = alpha * dropout_mask                             # :336
    value_j = value.index_select(0, src)
    if edge_attr is not None:
        value_j = value_j + edge_attr.view(-1, heads, c)         # :338-340
    msg = value_j * alpha.view(-1, heads, 1)                     # :342
    out = torch.zeros(n, heads, c, dtype=x.dtype, device=x.device).index_add_(0, dst, msg)
    out = out.view(-1, heads * c) if concat else out.mean(dim=1) # :239-242
    if w_skip is not None:                                       # root_weight :244-250
        x_r = F.linear(x, w_skip, b_skip)
        if w_beta is not None:
            beta = F.linear(torch.cat([out, x_r, out - x_r], dim=-1), w_beta).sigmoid()
            out = beta * x_r + (1 - beta) * out
        else:
            out = out + x_r
    res = out - x                                                # :267
    if return_attention:
        return res, (kept, query, key)
    return res


def transformer_conv(x, edge_index, w_query, b_query, w_key, b_key, w_value, b_value, w_skip, b_skip):
    """PyG 2.4.0 `TransformerConv(in, out, heads=1)` with its defaults (concat=True, beta=False, dropout=0, edge_dim=None,
    bias=True, root_weight=True) - what `get_conv(opt, 'TRANS', ...)` builds (`src/GNN.py:112-113`); same op sequence as
    `grand_residual` with a Linear value map and the root/skip term added (`out += lin_skip(x)`), no `- x`."""
    n, c = x.shape
    src, dst = edge_index[0], edge_index[1]
    query = F.linear(x, w_query, b_query).view(-1, 1, c)
    key = F.linear(x, w_key, b_key).view(-1, 1, c)
    value = F.linear(x, w_value, b_value).view(-1, 1, c)
    alpha = (query.index_select(0, dst) * key.index_select(0, src)).sum(dim=-1) / math.sqrt(c)
    alpha = pyg_softmax(alpha, dst, n)
    msg = value.index_select(0, src) * alpha.view(-1, 1, 1)
    out = torch.zeros(n, 1, c, dtype=x.dtype, device=x.device).index_add_(0, dst, msg)
    out = out.view(-1, c)                                       # concat=True, heads=1
    return out + F.linear(x, w_skip, b_skip)                    # root_weight=True


def triangle_edge_area_sum(mesh_points: torch.Tensor, cell_node_map, edge_index: torch.Tensor) -> torch.Tensor:
    """`reg_skew` (`src/GRAND_plus.py:280-324`): per directed edge (a, b) of `edge_index`, the summed area of the
    triangles that contain it in the orientation (i,j), (j,k), (k,i) of `cell_node_map` rows [i,j,k]; areas from the first
    two columns of the CURRENT node features (`self.mesh_points = x`, `:229`), differentiable.  An edge with 0 hits (or
    more than 2) gets 0, as the reference's if/elif chain leaves it (`:310-322`).  Literal O(E*T) matching, vectorised."""
    cells = torch.as_tensor(cell_node_map, dtype=torch.long)
    tri = mesh_points[cells]                                     # [T,3,C]  (:283)
    xx, yy = tri[:, :, 0], tri[:, :, 1]
    area = 0.5 * torch.abs(xx[:, 0] * (yy[:, 1] - yy[:, 2]) + xx[:, 1] * (yy[:, 2] - yy[:, 0]) + xx[:, 2] * (yy[:, 0] - yy[:, 1]))
    i_idx, j_idx, k_idx = cells[:, 0], cells[:, 1], cells[:, 2]
    tri_edges = torch.cat([torch.stack([i_idx, j_idx], 1), torch.stack([j_idx, k_idx], 1), torch.stack([k_idx, i_idx], 1)])   # [3T,2]
    edge_areas = torch.cat([area, area, area])
    out = torch.zeros(edge_index.shape[1], dtype=mesh_points.dtype)
    hit = (tri_edges[None, :, 0] == edge_index[0][:, None]) & (tri_edges[None, :, 1] == edge_index[1][:, None])   # [E,3T]
    count = hit.sum(1)
    summed = (hit.to(mesh_points.dtype) * edge_areas[None, :]).sum(1)
    return torch.where((count == 1) | (count == 2), summed, out)


def gat_attention(a_src: torch.Tensor, a_dst: torch.Tensor, edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """`GATConv.edge_update`: alpha = softmax_i(leaky_relu(alpha_j + alpha_i, 0.2)); a_src / a_dst are the per-node
    sums `(x * att).sum(-1)` [N,1]; edge_index already carries the self-loops."""
    alpha = a_src.index_select(0, edge_index[0]) + a_dst.index_select(0, edge_index[1])
    alpha = F.leaky_relu(alpha, 0.2)
    return pyg_softmax(alpha, edge_index[1], num_nodes)


def gat_conv(x, edge_index, w_lin, att_src, att_dst, bias):
    """PyG 2.4.0 `GATConv(in, out, heads=1)` forward with its defaults (concat=True, negative_slope=0.2,
    add_self_loops=True, bias=True); lin_src = lin_dst share `w_lin` for an int `in_channels`."""
    n = x.shape[0]
    h = F.linear(x, w_lin).view(n, 1, -1)                        # x_src = x_dst = lin_src(x)
    a_src = (h * att_src).sum(-1)
    a_dst = (h * att_dst).sum(-1)
    ei = with_self_loops(edge_index, n)                          # remove_self_loops + add_self_loops
    alpha = gat_attention(a_src, a_dst, ei, n)                   # [E',1]
    msg = h.index_select(0, ei[0]) * alpha.unsqueeze(-1)
    out = torch.zeros_like(h).index_add_(0, ei[1], msg).view(n, -1)
    return out + bias, (ei, alpha)


def gat_plus(x, edge_index, att_src, att_dst, kind: str = 'GAT_res_lap'):
    """`GAT_plus.forward` (`src/GRAND_plus.py:400-416`): GATConv attention with identity lin_src / lin_dst; the conv's own
    output is discarded, the attention is re-applied as a sparse matrix: `Ax = sparse(alpha)^T x`; returns Ax - x
    ('GAT_res_lap') or Ax ('GAT_lin')."""
    n = x.shape[0]
    h = x.view(n, 1, -1)
    a_src = (h * att_src).sum(-1)
    a_dst = (h * att_dst).sum(-1)
    ei = with_self_loops(edge_index, n)
    alpha = gat_attention(a_src, a_dst, ei, n)
    ax = torch.zeros_like(x).index_add_(0, ei[1], x.index_select(0, ei[0]) * alpha)   # (sparse_alpha.T @ x)_i = sum_j alpha_ij x_j
    if kind == 'GAT_res_lap':
        return ax - x, (ei, alpha)
    if kind == 'GAT_lin':
        return ax, (ei, alpha)
    raise NotImplementedError(kind)


def gcn_conv(x, edge_index, w_lin, bias):
    """PyG 2.4.0 `GCNConv(in, out)`: `gcn_norm` (add_remaining_self_loops, fill 1; deg by target; D^-1/2 A D^-1/2),
    `x = lin(x)`, sum-aggregate, `+ bias`."""
    n = x.shape[0]
    ei = with_self_loops(edge_index, n)                          # unweighted: add_remaining_self_loops == remove + add
    w = torch.ones(ei.shape[1], dtype=x.dtype)
    deg = torch.zeros(n, dtype=x.dtype).index_add_(0, ei[1], w)
    dis = deg.pow(-0.5)
    dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
    norm = dis.index_select(0, ei[0]) * w * dis.index_select(0, ei[1])
    h = F.linear(x, w_lin)
    out = torch.zeros_like(h).index_add_(0, ei[1], h.index_select(0, ei[0]) * norm.unsqueeze(-1))
    return out + bias


def with_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """`remove_self_loops` then `add_self_loops` (PyG utils; `src/GNN.py:220-223`): loops appended as arange(N)."""
    edge_index = edge_index[:, edge_index[0] != edge_index[1]]
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)


def masked_edge_index(data, dim: int, mesh_n: int, fix_boundary: bool = True) -> torch.Tensor:
    """`src/GNN.py:206-218`: drop masked edges, append boundary/corner self-loops."""
    edge_index = data.edge_index
    if not fix_boundary:
        return edge_index
    mask = ~data.to_boundary_edge_mask * ~data.to_corner_nodes_mask * ~data.diff_boundary_edges_mask
    edge_index = edge_index[:, mask]
    num_in_batch = int(data.batch.max().item()) + 1
    if dim == 1:
        ends = torch.cat([torch.tensor([0 + b * mesh_n, (1 + b) * mesh_n - 1]) for b in range(num_in_batch)])
        loops = ends.repeat(2, 1)
    else:
        import numpy as np
        corner = torch.stack([torch.from_numpy(np.asarray(a)) for a in data.corner_nodes])      # [B,4]
        counts = data.batch.unique(return_counts=True)[1]
        cum = torch.cumsum(counts, dim=0)
        corner = corner.clone()
        corner[1:] += cum[:-1].unsqueeze(-1)
        loops = corner.reshape(-1).repeat(2, 1)
    return torch.cat([edge_index, loops.to(edge_index.device)], dim=1)


def node_features(data, dim: int, inc_f: bool, inc_uu: bool, normalize: bool = False) -> torch.Tensor:
    """`src/GNN.py:225-239`."""
    x_comp = data.x_comp
    if dim == 1:
        x_comp = x_comp.unsqueeze(-1)
    feats = x_comp
    if inc_f:
        f = data.f_tensor
        if normalize:
            f = f / torch.max(f)
        feats = torch.cat([feats, f.unsqueeze(-1)], dim=1)
    if inc_uu:
        uu = data.uu_tensor
        if normalize:
            uu = uu / torch.max(uu)
        feats = torch.cat([feats, uu.unsqueeze(-1)], dim=1)
    return feats


def identity_encoder_weight(in_dim: int, out_dim: int) -> torch.Tensor:
    """Frozen `nn.Linear(in_dim,out_dim,bias=False).weight` of `src/GNN.py:75-90`."""
    w = torch.zeros(out_dim, in_dim)
    k = min(in_dim, out_dim)
    w[:k, :k] = torch.eye(k)
    return w


_NONLIN = {'relu': F.relu, 'elu': F.elu, 'selu': F.selu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
           'leaky_relu': F.leaky_relu, 'identity': lambda t: t}                           # get_nonlin, GNN.py:48-64


class _QKVS(nn.Module):
    """Parameter holder with PyG TransformerConv's state_dict names (heads=1, bias=True, root_weight=True)."""

    def __init__(self, c: int):
        super().__init__()
        self.lin_key = nn.Linear(c, c)
        self.lin_query = nn.Linear(c, c)
        self.lin_value = nn.Linear(c, c)
        self.lin_skip = nn.Linear(c, c, bias=True)


class _QK(nn.Module):
    """Parameter holder with the reference's state_dict names (`src/GRAND_plus.py:146-147,178`)."""

    def __init__(self, c: int, learnable_a: bool = False):
        super().__init__()
        self.lin_key = nn.Linear(c, c)
        self.lin_query = nn.Linear(c, c)
        self.lin_skip = nn.Linear(c, c, bias=False)           # allocated, never used (root_weight=False)
        if learnable_a:
            # `src/GRAND_plus.py:152-154`: Parameter(torch.Tensor(1, heads, 1)), never initialised there; ones is the defined
            # start value used on both sides of the parity tests (they load this state_dict into the HIP model)
            self.sm_temp_a = nn.Parameter(torch.ones(1, 1, 1))


def _glorot_(t: torch.Tensor) -> torch.Tensor:
    a = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))              # torch_geometric.nn.inits.glorot
    with torch.no_grad():
        return t.uniform_(-a, a)


class _GATPlusHolder(nn.Module):
    """Parameters of `GAT_plus(GATConv)` (`src/GRAND_plus.py:386-398`): att_src / att_dst [1,1,C]; lin_src / lin_dst are
    replaced by Identity, bias=False."""

    def __init__(self, c: int):
        super().__init__()
        self.att_src = nn.Parameter(_glorot_(torch.empty(1, 1, c)))
        self.att_dst = nn.Parameter(_glorot_(torch.empty(1, 1, c)))


class _GATHolder(nn.Module):
    """PyG `GATConv(c, c, heads=1)` parameter names: lin_src (== lin_dst), att_src, att_dst, bias."""

    def __init__(self, c: int):
        super().__init__()
        self.lin_src = nn.Linear(c, c, bias=False)
        _glorot_(self.lin_src.weight)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(_glorot_(torch.empty(1, 1, c)))
        self.att_dst = nn.Parameter(_glorot_(torch.empty(1, 1, c)))
        self.bias = nn.Parameter(torch.zeros(c))


class _GCNHolder(nn.Module):
    """PyG `GCNConv(c, c)` parameter names: lin.weight (glorot), bias (zeros)."""

    def __init__(self, c: int):
        super().__init__()
        self.lin = nn.Linear(c, c, bias=False)
        _glorot_(self.lin.weight)
        self.bias = nn.Parameter(torch.zeros(c))


class _GlobalCNN(nn.Module):
    """`GlobalFeatureExtractorCNN` (`src/feature_extractors.py:6-34`): 3x3 convs + SELU, global average pool."""

    def __init__(self, in_channels, mid_channels, out_channels, dim=2, num_layers=4):
        super().__init__()
        conv = nn.Conv1d if dim == 1 else nn.Conv2d
        chans = [in_channels] + [mid_channels] * (num_layers - 1) + [out_channels]
        self.convs = nn.ModuleList([conv(chans[k], chans[k + 1], kernel_size=3, stride=1, padding=1)
                                    for k in range(num_layers)])
        self.dim = dim

    def forward(self, u):
        u = u / u.abs().max()                                                              # :29
        for conv in self.convs:
            u = F.selu(conv(u))                                                            # :30-31
        return u.mean(dim=tuple(range(2, u.dim())))                                        # adaptive avg pool to 1 + flatten


def fd_tensor_to_grid(u, mapping, mesh_dims, batch_size, dim):
    """`reshape_fd_tensor_to_grid` (`src/utils_data.py:125-141`)."""
    ub = u.reshape(batch_size, -1)
    if dim == 1:
        return ub
    if mapping is not None:
        ub = torch.gather(ub, 1, mapping.unsqueeze(0).expand(batch_size, -1))
    g = ub.reshape(batch_size, mesh_dims[0], mesh_dims[1])
    return torch.flip(torch.transpose(g, 1, 2), [1])


class OracleGNN(nn.Module):
    """`GNN` (`src/GNN.py:144-306`) for enc='identity', GRAND/GRAND_plus, mesh_loss|modular."""

    def __init__(self, dataset, opt):
        super().__init__()
        self.opt = opt
        self.dim = dataset.num_x_comp_features
        in_dim = self.dim + int(bool(opt['gnn_inc_feat_f'])) + int(bool(opt['gnn_inc_feat_uu']))
        c = opt['hidden_dim']
        self.dataset = dataset
        for flag, name in (('gnn_inc_glob_feat_f', 'global_feature_extractor_cnn_f'),
                           ('gnn_inc_glob_feat_uu', 'global_feature_extractor_cnn_uu')):           # GNN.py:170-175
            if opt.get(flag):
                in_dim += opt['global_feat_dim']
                setattr(self, name, _GlobalCNN(1, c, opt['global_feat_dim'], dim=self.dim))
        assert opt['enc'] == 'identity' and opt['conv_type'] in ('GRAND', 'GRAND_plus', 'TRANS', 'GAT_plus', 'GAT', 'GCN')
        _QK_ = {'TRANS': _QKVS, 'GAT_plus': _GATPlusHolder, 'GAT': _GATHolder, 'GCN': _GCNHolder}.get(opt['conv_type'], _QK)
        self.enc = nn.Linear(in_dim, c, bias=False)
        self.enc.weight.data = identity_encoder_weight(in_dim, c)
        self.enc.weight.requires_grad = False
        make = _QK_
        if opt.get('softmax_temp_type') == 'learnable_a' and opt['conv_type'] == 'GRAND_plus':     # GRAND_plus.py:152-154
            make = lambda c_: _QK(c_, learnable_a=True)                                           # noqa: E731
        if opt['share_conv']:
            shared = make(c)
            self.conv_layers = nn.ModuleList([shared for _ in range(opt['num_layers'])])     # :131-141
        else:
            self.conv_layers = nn.ModuleList([make(c) for _ in range(opt['num_layers'])])
        if opt.get('learn_step'):
            self.steps = nn.ParameterList([nn.Parameter(torch.tensor([opt['time_step']]))
                                           for _ in range(opt['num_layers'])])              # :179-180
        self.end_MLmodel = None

    def temperature(self, layer=None):
        t = self.opt.get('softmax_temp_type')
        if t == 'learnable_a' and layer is not None and hasattr(layer, 'sm_temp_a'):
            return layer.sm_temp_a.squeeze(2)                                              # GRAND_plus.py:328-329: [1,H]
        return self.opt['softmax_temp'] if t == 'fixed' else None

    def forward(self, data, return_all: bool = False):
        opt = self.opt
        edge_index = masked_edge_index(data, self.dim, opt['mesh_dims'][0], opt['fix_boundary'])
        if opt.get('self_loops'):
            edge_index = with_self_loops(edge_index, data.x_comp.shape[0])                 # :220-223
        feats = node_features(data, self.dim, opt['gnn_inc_feat_f'], opt['gnn_inc_feat_uu'],
                              opt.get('gnn_normalize', False))
        for flag, field, name in (('gnn_inc_glob_feat_f', 'f_tensor', 'global_feature_extractor_cnn_f'),
                                  ('gnn_inc_glob_feat_uu', 'uu_tensor', 'global_feature_extractor_cnn_uu')):   # :240-268
            if opt.get(flag):
                u = getattr(data, field)
                if opt.get('gnn_normalize', False):
                    u = u / torch.max(u)
                nb = int(data.batch.max()) + 1
                grid = fd_tensor_to_grid(u, getattr(self.dataset, 'mapping_tensor', None), opt['mesh_dims'], nb, self.dim)
                per_mesh = getattr(self, name)(grid.unsqueeze(1).to(self.enc.weight.dtype))
                feats = torch.cat([feats.to(per_mesh.dtype), per_mesh.repeat_interleave(torch.bincount(data.batch), dim=0)], dim=-1)
        x = self.enc(feats.to(self.enc.weight.dtype))                                      # :270
        x = F.dropout(x, opt.get('dropout', 0.0), training=self.training)                  # :271 (identity at the shipped p = 0)
        alphas = []
        for i, layer in enumerate(self.conv_layers):                                       # :273
            if opt['conv_type'] == 'TRANS':                                                # GNN.py:284, stock TransformerConv
                res = transformer_conv(x, edge_index, layer.lin_query.weight, layer.lin_query.bias, layer.lin_key.weight,
                                       layer.lin_key.bias, layer.lin_value.weight, layer.lin_value.bias,
                                       layer.lin_skip.weight, layer.lin_skip.bias)
                alpha = None
            elif opt['conv_type'] == 'GAT_plus':                                           # GNN.py:120-121, GRAND_plus.py:400-416
                res, (_, alpha) = gat_plus(x, edge_index, layer.att_src, layer.att_dst, opt.get('gat_plus_type', 'GAT_res_lap'))
            elif opt['conv_type'] == 'GAT':                                                # GNN.py:110-111
                res, (_, alpha) = gat_conv(x, edge_index, layer.lin_src.weight, layer.att_src, layer.att_dst, layer.bias)
            elif opt['conv_type'] == 'GCN':                                                # GNN.py:109-110
                res, alpha = gcn_conv(x, edge_index, layer.lin.weight, layer.bias), None
            else:
                area = None
                if opt.get('reg_skew') and self.dim == 2:                                  # GRAND_plus.py:280-324
                    area = triangle_edge_area_sum(x, self.dataset.mesh.coordinates.cell_node_map().values, edge_index)
                res, (alpha, _, _) = grand_residual(x, edge_index, layer.lin_query.weight, layer.lin_query.bias,
                                                    layer.lin_key.weight, layer.lin_key.bias,
                                                    self.temperature(layer), return_attention=True, edge_area_sum=area)
            if not (opt['residual'] and opt['conv_type'] == 'GRAND_plus'):
                res = F.dropout(res, opt.get('dropout', 0.0), training=self.training)      # :285 / :295
                res = _NONLIN[opt['non_lin']](res)                                         # :286 / :296
            if opt['residual']:
                step = self.steps[i] if opt.get('learn_step') else opt['time_step']
                x = x + step * res                                                         # :288-291
            else:
                x = res                                                                    # :294-296
            alphas.append(alpha)
        x_phys = x[:, :self.dim]                                                           # :299
        self.end_MLmodel = time.time()
        if return_all:
            return x_phys, x, alphas, edge_index
        return x_phys
