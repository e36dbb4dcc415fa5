"""Differentiable message-passing primitives over a `MeshGraph` (C-ABI: `gadapt_spmm`, `gadapt_sddmm`,
`gadapt_edge_softmax_*`, `gadapt_edge_combine`, `gadapt_edge_rowsum`; include/gadapt_hip.h).

They are what PyG's `MessagePassing.propagate` + `utils.softmax` are to the reference's conv variants
(`get_conv`, `src/GNN.py:108-124`): gather by `edge_index[0]`, combine per edge, softmax grouped by
`edge_index[1]`, scatter-add to the targets - here as CSR-row kernels in HIP, with the transposed products of the
backward passes on the source CSR.  Per-edge tensors are in the graph's TARGET-CSR order
(`MeshGraph.alpha_to_edge_order` converts to the caller's edge order).  No CPU / torch fallback: a CPU tensor raises.

The fused GRAND kernels (functional.py) do not use these; the variants with other edge scores do:
`GAT_plus`, `GAT`, `GCN` (conv.py), `reg_skew`, and hidden sizes the fused kernels are not built for.
"""
from __future__ import annotations

from typing import Optional

import torch

from ._native import NativeError, check, current_stream, lib, ptr
from .graph import MeshGraph


def _dev(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise NativeError(f"{what}: the message-passing primitives run on the MI355X only (got a {t.device} tensor); "
                          "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{what}: fp32 expected, got {t.dtype}")
    return t.contiguous()


def _pad4(x: torch.Tensor):
    """[N,C] -> ([N,C4] zero-padded to a multiple of 4 columns, C)."""
    c = x.shape[1]
    if c % 4 == 0:
        return x, c
    return torch.nn.functional.pad(x, (0, 4 - c % 4)), c


def _spmm_raw(graph: MeshGraph, w: Optional[torch.Tensor], x: torch.Tensor, transpose: bool, self_scale: float = 0.0) -> torch.Tensor:
    xp, c = _pad4(x)
    xp = xp.contiguous()
    out = torch.empty_like(xp)
    check(lib().gadapt_spmm(graph.c_ref, int(transpose), ptr(w), ptr(xp), ptr(out), xp.shape[1], float(self_scale),
                            current_stream(x.device)), 'gadapt_spmm')
    return out if xp.shape[1] == c else out[:, :c]


def _sddmm_raw(graph: MeshGraph, a: torch.Tensor, b: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    ap, _ = _pad4(a)
    bp, _ = _pad4(b)
    out = torch.empty(max(graph.num_edges, 1), device=a.device, dtype=torch.float32)
    check(lib().gadapt_sddmm(graph.c_ref, 0, ptr(ap.contiguous()), ptr(bp.contiguous()), ptr(out), ap.shape[1], float(scale),
                             current_stream(a.device)), 'gadapt_sddmm')
    return out[:graph.num_edges]


class _SpMM(torch.autograd.Function):
    """out_i = sum_{e: j->i} w_e x_j.  d x_j = sum_{e: j->i} w_e g_i (source CSR);  d w_e = <g_i, x_j>."""

    @staticmethod
    def forward(ctx, w, x, graph: MeshGraph):
        x = _dev(x, 'spmm x')
        w = None if w is None else _dev(w, 'spmm weights')
        if x.shape[0] != graph.num_nodes or (w is not None and w.numel() != graph.num_edges):
            raise ValueError(f"spmm: x has {x.shape[0]} rows / w has {None if w is None else w.numel()} entries for a graph of "
                             f"{graph.num_nodes} nodes and {graph.num_edges} edges")
        ctx.graph = graph
        ctx.save_for_backward(w, x)
        return _spmm_raw(graph, w, x, False)

    @staticmethod
    def backward(ctx, g):
        w, x = ctx.saved_tensors
        g = g.contiguous()
        dw = _sddmm_raw(ctx.graph, g, x) if (w is not None and ctx.needs_input_grad[0]) else None
        dx = _spmm_raw(ctx.graph, w, g, True) if ctx.needs_input_grad[1] else None
        return dw, dx, None


class _SDDMM(torch.autograd.Function):
    """s_e = <a_i, b_j> for e: j->i.  d a_i = sum_in ds_e b_j;  d b_j = sum_out ds_e a_i."""

    @staticmethod
    def forward(ctx, a, b, graph: MeshGraph):
        a, b = _dev(a, 'sddmm a'), _dev(b, 'sddmm b')
        if a.shape != b.shape or a.shape[0] != graph.num_nodes:
            raise ValueError(f"sddmm: shapes {tuple(a.shape)} / {tuple(b.shape)} for a graph of {graph.num_nodes} nodes")
        ctx.graph = graph
        ctx.save_for_backward(a, b)
        return _sddmm_raw(graph, a, b)

    @staticmethod
    def backward(ctx, ds):
        a, b = ctx.saved_tensors
        ds = ds.contiguous()
        da = _spmm_raw(ctx.graph, ds, b, False) if ctx.needs_input_grad[0] else None
        db = _spmm_raw(ctx.graph, ds, a, True) if ctx.needs_input_grad[1] else None
        return da, db, None


class _EdgeSoftmax(torch.autograd.Function):
    """`torch_geometric.utils.softmax(s, index=edge_index[1])` (call site `src/GRAND_plus.py:333`), max detached, +1e-16."""

    @staticmethod
    def forward(ctx, s, graph: MeshGraph):
        s = _dev(s, 'edge scores')
        if s.numel() != graph.num_edges:
            raise ValueError(f"edge_softmax: {s.numel()} scores for {graph.num_edges} edges")
        alpha = torch.empty(max(graph.num_edges, 1), device=s.device, dtype=torch.float32)
        check(lib().gadapt_edge_softmax_forward(graph.c_ref, ptr(s.reshape(-1)), ptr(alpha), current_stream(s.device)),
              'gadapt_edge_softmax_forward')
        alpha = alpha[:graph.num_edges]
        ctx.graph = graph
        ctx.save_for_backward(alpha)
        return alpha

    @staticmethod
    def backward(ctx, d_alpha):
        (alpha,) = ctx.saved_tensors
        ds = torch.empty(max(ctx.graph.num_edges, 1), device=alpha.device, dtype=torch.float32)
        check(lib().gadapt_edge_softmax_backward(ctx.graph.c_ref, ptr(alpha.contiguous()), ptr(d_alpha.contiguous()), ptr(ds),
                                                 current_stream(alpha.device)), 'gadapt_edge_softmax_backward')
        return ds[:ctx.graph.num_edges], None


def _rowsum(graph: MeshGraph, vals: torch.Tensor, by_source: bool) -> torch.Tensor:
    out = torch.empty(graph.num_nodes, device=vals.device, dtype=torch.float32)
    check(lib().gadapt_edge_rowsum(graph.c_ref, int(by_source), ptr(vals.contiguous()), ptr(out), current_stream(vals.device)),
          'gadapt_edge_rowsum')
    return out


class _EdgeCombine(torch.autograd.Function):
    """o_e = u[j] + v[i] (op 0) or u[j] * v[i] (op 1) for e: j->i (GATConv's `alpha_j + alpha_i`, gcn_norm's product)."""

    @staticmethod
    def forward(ctx, u, v, graph: MeshGraph, op: int):
        u, v = _dev(u, 'edge_combine u'), _dev(v, 'edge_combine v')
        if u.shape != (graph.num_nodes,) or v.shape != (graph.num_nodes,):
            raise ValueError("edge_combine: per-node vectors of length num_nodes expected")
        out = torch.empty(max(graph.num_edges, 1), device=u.device, dtype=torch.float32)
        check(lib().gadapt_edge_combine(graph.c_ref, ptr(u), ptr(v), ptr(out), int(op), current_stream(u.device)), 'gadapt_edge_combine')
        ctx.graph, ctx.op = graph, int(op)
        ctx.save_for_backward(u, v)
        return out[:graph.num_edges]

    @staticmethod
    def backward(ctx, g):
        u, v = ctx.saved_tensors
        graph = ctx.graph
        g = g.contiguous()
        if ctx.op == 0:
            du = _rowsum(graph, g, True) if ctx.needs_input_grad[0] else None
            dv = _rowsum(graph, g, False) if ctx.needs_input_grad[1] else None
        else:   # d u_j = sum_out g_e v_i ; d v_i = sum_in g_e u_j: the same kernel with the other vector folded into g
            du = _rowsum(graph, g * _EdgeCombine.apply(torch.ones_like(u), v, graph, 1), True) if ctx.needs_input_grad[0] else None
            dv = _rowsum(graph, g * _EdgeCombine.apply(u, torch.ones_like(v), graph, 1), False) if ctx.needs_input_grad[1] else None
        return du, dv, None, None


def _edge_vec_raw(graph: MeshGraph, mode: int, w, a, b, c: int) -> torch.Tensor:
    dev = (b if b is not None else a).device
    e = max(graph.num_edges, 1)
    out = torch.empty((e,) if mode == 0 else ((graph.num_nodes, c) if mode == 1 else (e, c)), device=dev, dtype=torch.float32)
    check(lib().gadapt_edge_vector_op(graph.c_ref, mode, ptr(w), ptr(a), ptr(b), ptr(out), c, current_stream(dev)), 'gadapt_edge_vector_op')
    return out[:graph.num_edges] if mode != 1 else out


class _EdgeNodeDot(torch.autograd.Function):
    """s_e = <a_i, B_e>, i = target of e; B [E,C] per-edge vectors in target-CSR order."""

    @staticmethod
    def forward(ctx, a, b, graph: MeshGraph):
        a, b = _dev(a, 'edge_node_dot a'), _dev(b, 'edge_node_dot b')
        if a.shape[0] != graph.num_nodes or b.shape != (graph.num_edges, a.shape[1]) or a.shape[1] % 4:
            raise ValueError(f"edge_node_dot: a {tuple(a.shape)}, b {tuple(b.shape)} for {graph.num_nodes} nodes / {graph.num_edges} edges (C % 4 == 0)")
        ctx.graph = graph
        ctx.save_for_backward(a, b)
        return _edge_vec_raw(graph, 0, None, a, b, a.shape[1])

    @staticmethod
    def backward(ctx, ds):
        a, b = ctx.saved_tensors
        ds = ds.contiguous()
        da = _edge_vec_raw(ctx.graph, 1, ds, None, b, a.shape[1]) if ctx.needs_input_grad[0] else None
        db = _edge_vec_raw(ctx.graph, 2, ds, a, None, a.shape[1]) if ctx.needs_input_grad[1] else None
        return da, db, None


class _EdgeWeightedRowsum(torch.autograd.Function):
    """out_i = sum over the in-edges e of i of w_e B_e."""

    @staticmethod
    def forward(ctx, w, b, graph: MeshGraph):
        w, b = _dev(w, 'edge_weighted_rowsum w'), _dev(b, 'edge_weighted_rowsum b')
        if w.numel() != graph.num_edges or b.shape[0] != graph.num_edges or b.shape[1] % 4:
            raise ValueError("edge_weighted_rowsum: w [E], b [E,C] with C % 4 == 0 expected")
        ctx.graph = graph
        ctx.save_for_backward(w, b)
        return _edge_vec_raw(graph, 1, w.reshape(-1), None, b, b.shape[1])

    @staticmethod
    def backward(ctx, g):
        w, b = ctx.saved_tensors
        g = g.contiguous()
        dw = _edge_vec_raw(ctx.graph, 0, None, g, b, b.shape[1]).view_as(w) if ctx.needs_input_grad[0] else None
        db = _edge_vec_raw(ctx.graph, 2, w.reshape(-1), g, None, b.shape[1]) if ctx.needs_input_grad[1] else None
        return dw, db, None


def edge_node_dot(graph: MeshGraph, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[E] s_e = <a_i, b_e> for e: j -> i; `b` [E,C] in target-CSR order (the <query_i, lin_edge(edge_attr)_e> term, GRAND_plus.py:273-279)."""
    return _EdgeNodeDot.apply(a, b, graph)


def edge_weighted_rowsum(graph: MeshGraph, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[N,C] out_i = sum over in-edges of w_e b_e (the alpha-weighted edge term of the message, GRAND_plus.py:338-342)."""
    return _EdgeWeightedRowsum.apply(w, b, graph)


def spmm(graph: MeshGraph, w: Optional[torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """[N,C] aggregation out_i = sum over in-edges j->i of w_e x_j (`w=None`: plain sum); w in target-CSR order."""
    return _SpMM.apply(w, x, graph)


def sddmm(graph: MeshGraph, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[E] scores s_e = <a_i, b_j> for e: j->i, target-CSR order."""
    return _SDDMM.apply(a, b, graph)


def edge_softmax(graph: MeshGraph, scores: torch.Tensor) -> torch.Tensor:
    """Softmax of per-edge scores over the in-edges of each target (PyG `utils.softmax`, +1e-16 in the denominator)."""
    return _EdgeSoftmax.apply(scores, graph)


def edge_add(graph: MeshGraph, u_src: torch.Tensor, v_dst: torch.Tensor) -> torch.Tensor:
    return _EdgeCombine.apply(u_src, v_dst, graph, 0)


def edge_mul(graph: MeshGraph, u_src: torch.Tensor, v_dst: torch.Tensor) -> torch.Tensor:
    return _EdgeCombine.apply(u_src, v_dst, graph, 1)


def in_degree(graph: MeshGraph) -> torch.Tensor:
    """[N] float in-degree (row lengths of the target CSR)."""
    return (graph.rowptr_t[1:] - graph.rowptr_t[:-1]).to(torch.float32)


def attention_aggregate(graph: MeshGraph, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale,
                        edge_weight: Optional[torch.Tensor] = None):
    """sum_j softmax_j(scale <q_i,k_j> [* edge_weight_e]) v_j and the attention [E]: the generic form of the GRAND layer
    (`src/GRAND_plus.py:269-343`) for widths the fused kernels are not built for and for `reg_skew`."""
    s = sddmm(graph, q, k) * scale
    if edge_weight is not None:
        s = s * edge_weight
    alpha = edge_softmax(graph, s)
    return spmm(graph, alpha, v), alpha
