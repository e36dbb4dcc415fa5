"""Pins the CPU oracle (no GPU): independent dense fp64 formulation, gradcheck, analytic known answers,
golden fixtures.  The reference has no tests for this path, these stand in for them (§8(c))."""
import glob
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from g_adaptivity_amd import MeshDataset, collate, hot_path_opt, square_mesh
from oracle.dense_check import dense_attention, dense_euler_block, dense_residual
from oracle.pyg_restatement import OracleGNN, grand_residual, masked_edge_index, node_features, pyg_softmax

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*.npz')))


def _case(n=11, B=2, C=8, seed=0, dtype=torch.float64):
    ds = MeshDataset([n, n], B, seed=seed)
    data = collate(ds.samples)
    ei = masked_edge_index(data, 2, n)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(data.x_comp.shape[0], C, generator=g, dtype=dtype)
    k = 1.0 / math.sqrt(C)
    wq, wk = [(torch.rand(C, C, generator=g, dtype=dtype) * 2 - 1) * k for _ in range(2)]
    bq, bk = [(torch.rand(C, generator=g, dtype=dtype) * 2 - 1) * k for _ in range(2)]
    return data, ei, x, wq, bq, wk, bk


def test_segment_softmax_matches_dense_softmax():
    g = torch.Generator().manual_seed(1)
    src = torch.randn(40, 1, generator=g, dtype=torch.float64)
    index = torch.arange(8).repeat_interleave(5)
    out = pyg_softmax(src, index, 8).view(8, 5)
    ref = torch.softmax(src.view(8, 5), dim=1)
    assert torch.allclose(out, ref, atol=1e-14)
    assert torch.allclose(out.sum(1), torch.ones(8, dtype=torch.float64), atol=1e-14)


@pytest.mark.parametrize("temperature", [None, 2.0])
def test_sparse_restatement_matches_dense_fp64(temperature):
    data, ei, x, wq, bq, wk, bk = _case()
    res = grand_residual(x, ei, wq, bq, wk, bk, temperature)
    ref = dense_residual(x, ei, wq, bq, wk, bk, temperature)
    assert (res - ref).abs().max().item() < 1e-12


def test_euler_block_matches_dense_fp64():
    data, ei, x, wq, bq, wk, bk = _case(n=9)
    y = x
    for _ in range(4):
        y = y + 0.1 * grand_residual(y, ei, wq, bq, wk, bk)
    ref = dense_euler_block(x, ei, wq, bq, wk, bk, 4, 0.1)
    assert (y - ref).abs().max().item() < 1e-12


def test_gradcheck_fp64():
    data, ei, x, wq, bq, wk, bk = _case(n=5, B=1, C=4)
    args = [t.clone().requires_grad_(True) for t in (x, wq, bq, wk, bk)]
    assert torch.autograd.gradcheck(lambda x_, a, b, c, d: grand_residual(x_, ei, a, b, c, d), args, eps=1e-6, atol=1e-6)


def test_known_answer_zero_weights_is_neighbour_mean():
    data, ei, x, *_ = _case()
    C = x.shape[1]
    z, zb = torch.zeros(C, C, dtype=x.dtype), torch.zeros(C, dtype=x.dtype)
    res, (alpha, _, _) = grand_residual(x, ei, z, zb, z, zb, return_attention=True)
    deg = torch.bincount(ei[1], minlength=x.shape[0]).to(x.dtype)
    assert torch.allclose(alpha.view(-1), 1.0 / deg[ei[1]], atol=1e-15)
    mean = torch.zeros_like(x).index_add_(0, ei[1], x[ei[0]]) / deg[:, None]
    assert torch.allclose(res, mean - x, atol=1e-13)


def test_known_answers_boundary_and_corners():
    n, B = 11, 2
    ds = MeshDataset([n, n], B, seed=3)
    data = collate(ds.samples)
    opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=8, num_layers=4)
    torch.manual_seed(3)
    out = OracleGNN(ds, opt)(data)
    x0 = data.x_comp
    corners = torch.cat([torch.as_tensor(c) + b * n * n for b, c in enumerate(data.corner_nodes)])
    assert torch.equal(out[corners], x0[corners])                      # single self-loop: res = 0 up to the 1e-16 epsilon
    # a node on side x=0 only hears from nodes on x=0: its x coordinate cannot move (same for the other sides)
    for col, val in ((0, 0.0), (0, 1.0), (1, 0.0), (1, 1.0)):
        on_side = x0[:, col] == val
        assert torch.allclose(out[on_side, col], x0[on_side, col], atol=1e-7)
    interior = ~data.boundary_nodes
    assert (out[interior] - x0[interior]).abs().max() > 1e-5          # interior nodes do move


def test_alpha_rows_sum_to_one_and_dt_zero_is_identity():
    data, ei, x, wq, bq, wk, bk = _case()
    _, (alpha, _, _) = grand_residual(x, ei, wq, bq, wk, bk, return_attention=True)
    sums = torch.zeros(x.shape[0], dtype=x.dtype).index_add_(0, ei[1], alpha.view(-1))
    assert torch.allclose(sums, torch.ones_like(sums), atol=1e-14)
    assert torch.equal(x + 0.0 * grand_residual(x, ei, wq, bq, wk, bk), x)


def test_edge_permutation_invariance():
    data, ei, x, wq, bq, wk, bk = _case(dtype=torch.float32)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(5))
    a = grand_residual(x, ei, wq, bq, wk, bk)
    b = grand_residual(x, ei[:, perm], wq, bq, wk, bk)
    assert (a - b).abs().max().item() < 1e-5 * a.abs().max().item()


def test_key_bias_gradient_vanishes():
    """softmax is shift invariant per target, so dL/d(lin_key.bias) = 0 (the HIP path returns exact zeros)."""
    data, ei, x, wq, bq, wk, bk = _case()
    bk = bk.clone().requires_grad_(True)
    bq = bq.clone().requires_grad_(True)
    res = grand_residual(x, ei, wq, bq, wk, bk)
    (res * torch.randn_like(res)).sum().backward()
    assert bk.grad.abs().max().item() < 1e-12 * max(1.0, bq.grad.abs().max().item()) + 1e-13


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:2] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    g = np.load(path, allow_pickle=False)
    mesh_dims = [int(v) for v in g['mesh_dims']]
    opt = hot_path_opt(mesh_dims=mesh_dims, hidden_dim=int(g['hidden']), num_layers=int(g['layers']), conv_type=str(g['conv_type']),
                       gnn_inc_feat_f=bool(int(g['inc_f'])), gnn_inc_feat_uu=bool(int(g['inc_uu'])))
    ds = MeshDataset(mesh_dims, int(g['batch']), seed=0)
    data = collate(ds.samples)
    assert np.array_equal(masked_edge_index(data, len(mesh_dims), mesh_dims[0]).numpy(), g['edge_index'])
    m = OracleGNN(ds, dict(opt))
    lay = m.conv_layers[0]
    with torch.no_grad():
        lay.lin_query.weight.copy_(torch.from_numpy(g['wq'])); lay.lin_query.bias.copy_(torch.from_numpy(g['bq']))
        lay.lin_key.weight.copy_(torch.from_numpy(g['wk'])); lay.lin_key.bias.copy_(torch.from_numpy(g['bk']))
    out = m(data)
    tgt = data.x_phys if data.x_phys.dim() == 2 else data.x_phys.unsqueeze(-1)
    F.mse_loss(out, tgt).backward()
    # fp32 run agrees with its stored fp32 self to rounding and with the fp64 twin to 1e-6
    assert np.abs(out.detach().numpy() - g['x_phys_f32']).max() <= 2e-7
    assert np.abs(out.detach().numpy() - g['x_phys_f64']).max() <= 1e-6
    assert np.abs(out.detach().numpy() - g['x_phys_f64']).max() <= 1e-5 * np.abs(g['x_phys_f64']).max()   # the parity bar itself
    scale = np.abs(g['d_wq_f64']).max()
    assert np.abs(lay.lin_query.weight.grad.numpy() - g['d_wq_f64']).max() <= 1e-3 * scale


def test_transformer_conv_matches_dense_fp64():
    """The 'TRANS' branch of get_conv (stock TransformerConv, GNN.py:112-113): gather/softmax/scatter restatement against
    the dense masked-softmax formulation, fp64, including a node without in-edges (its aggregation is empty, the root term stays)."""
    import math
    from oracle.pyg_restatement import transformer_conv
    torch.manual_seed(0)
    n, c = 30, 8
    ei = torch.unique(torch.stack([torch.randint(0, n, (120,)), torch.randint(0, n, (120,))]), dim=1)
    ei = ei[:, ei[1] != 7]                                           # node 7: no in-edge
    x = torch.randn(n, c, dtype=torch.float64)
    W = [torch.randn(c, c, dtype=torch.float64) * 0.3 for _ in range(4)]
    b = [torch.randn(c, dtype=torch.float64) * 0.3 for _ in range(4)]
    out = transformer_conv(x, ei, W[0], b[0], W[1], b[1], W[2], b[2], W[3], b[3])
    Q, K, V = x @ W[0].T + b[0], x @ W[1].T + b[1], x @ W[2].T + b[2]
    mask = torch.zeros(n, n, dtype=torch.bool)
    mask[ei[1], ei[0]] = True
    A = torch.nan_to_num(torch.softmax(((Q @ K.T) / math.sqrt(c)).masked_fill(~mask, float('-inf')), dim=1), nan=0.0)
    ref = A @ V + x @ W[3].T + b[3]
    assert (out - ref).abs().max().item() <= 1e-13
    assert torch.allclose(out[7], x[7] @ W[3].T + b[3], rtol=0, atol=1e-14)


# ---------------------------------------------------------------------------------------------------------------
# conv variants behind get_conv (src/GNN.py:108-124): dense fp64 cross-checks of the gather / scatter restatements
# ---------------------------------------------------------------------------------------------------------------
def _looped_dense(ei, n):
    """[N,N] multiplicity matrix M[i,j] of edges j->i after remove_self_loops + add_self_loops."""
    keep = ei[0] != ei[1]
    m = torch.zeros(n, n, dtype=torch.float64)
    m.index_put_((ei[1][keep], ei[0][keep]), torch.ones(int(keep.sum()), dtype=torch.float64), accumulate=True)
    return m + torch.eye(n, dtype=torch.float64)


def test_gat_conv_and_gat_plus_match_dense_fp64():
    """GATConv (`src/GNN.py:110-111`) and GAT_plus (`src/GRAND_plus.py:386-416`): additive scores, leaky_relu(0.2), self-loops
    replaced / added for every node, softmax by target - against the dense masked-softmax formulation."""
    from oracle.pyg_restatement import gat_conv, gat_plus
    torch.manual_seed(1)
    n, c = 25, 8
    ei = torch.unique(torch.stack([torch.randint(0, n, (90,)), torch.randint(0, n, (90,))]), dim=1)
    ei = torch.cat([ei, torch.tensor([[3, 3], [3, 3]])], dim=1)          # an existing self-loop, twice: replaced by ONE loop
    x = torch.randn(n, c, dtype=torch.float64)
    w = torch.randn(c, c, dtype=torch.float64) * 0.4
    a_s, a_d = torch.randn(1, 1, c, dtype=torch.float64), torch.randn(1, 1, c, dtype=torch.float64)
    bias = torch.randn(c, dtype=torch.float64)
    mult = _looped_dense(ei, n)

    def dense(h):
        s = torch.nn.functional.leaky_relu((h @ a_d.view(-1))[:, None] + (h @ a_s.view(-1))[None, :], 0.2)   # [i,j]
        e = torch.where(mult > 0, torch.exp(s - torch.where(mult > 0, s, torch.full_like(s, -1e300)).max(1, keepdim=True).values), torch.zeros_like(s)) * mult
        return e / (e.sum(1, keepdim=True) + 1e-16)

    out, (ei2, alpha) = gat_conv(x, ei, w, a_s, a_d, bias)
    h = x @ w.T
    assert (out - (dense(h) @ h + bias)).abs().max().item() <= 1e-12
    assert ei2.shape[1] == int((ei[0] != ei[1]).sum()) + n and torch.equal(ei2[:, -n:], torch.arange(n).repeat(2, 1))
    res, _ = gat_plus(x, ei, a_s, a_d, 'GAT_res_lap')
    lin, _ = gat_plus(x, ei, a_s, a_d, 'GAT_lin')
    assert (res - (dense(x) @ x - x)).abs().max().item() <= 1e-12 and (lin - dense(x) @ x).abs().max().item() <= 1e-12


def test_gcn_conv_matches_dense_fp64():
    """GCNConv (`src/GNN.py:109-110`): D^-1/2 (A + I) D^-1/2 (x W^T) + b with degrees by target, multiplicities kept."""
    from oracle.pyg_restatement import gcn_conv
    torch.manual_seed(2)
    n, c = 20, 6
    ei = torch.stack([torch.randint(0, n, (70,)), torch.randint(0, n, (70,))])
    x = torch.randn(n, c, dtype=torch.float64)
    w, b = torch.randn(c, c, dtype=torch.float64), torch.randn(c, dtype=torch.float64)
    mult = _looped_dense(ei, n)
    dis = mult.sum(1).pow(-0.5)
    ref = (dis[:, None] * mult * dis[None, :]) @ (x @ w.T) + b
    assert (gcn_conv(x, ei, w, b) - ref).abs().max().item() <= 1e-12


def test_triangle_edge_area_sum_matches_the_reference_loop():
    """`reg_skew` weights (`src/GRAND_plus.py:280-324`): the vectorised restatement against the literal per-edge loop."""
    from g_adaptivity_amd import square_mesh
    from oracle.pyg_restatement import triangle_edge_area_sum
    m = square_mesh(6)
    torch.manual_seed(0)
    pts = torch.cat([m.x_comp.double() + 0.02 * torch.randn(36, 2, dtype=torch.float64), torch.randn(36, 3, dtype=torch.float64)], 1)
    cells = m.cells.numpy()
    got = triangle_edge_area_sum(pts, cells, m.edge_index)
    tri = pts[torch.from_numpy(cells)]
    xx, yy = tri[:, :, 0], tri[:, :, 1]
    area = 0.5 * torch.abs(xx[:, 0] * (yy[:, 1] - yy[:, 2]) + xx[:, 1] * (yy[:, 2] - yy[:, 0]) + xx[:, 2] * (yy[:, 0] - yy[:, 1]))
    tri_edges = [(int(a), int(b)) for a, b in zip(cells[:, 0], cells[:, 1])] + [(int(a), int(b)) for a, b in zip(cells[:, 1], cells[:, 2])] + \
                [(int(a), int(b)) for a, b in zip(cells[:, 2], cells[:, 0])]
    want = torch.zeros(m.edge_index.shape[1], dtype=torch.float64)
    for e, (a, b) in enumerate(m.edge_index.t().tolist()):
        hits = [k for k, te in enumerate(tri_edges) if te == (a, b)]
        if len(hits) in (1, 2):
            want[e] = sum(area[k % len(cells)] for k in hits)
    assert torch.allclose(got, want, rtol=0, atol=1e-15)
    assert (want > 0).any() and (want == 0).any()                        # oriented matching: some directed edges get no triangle


def test_edge_features_against_a_dense_loop():
    """`edge_dim` / `lin_edge` (`src/GRAND_plus.py:165-166,273-277,338-340`) in the oracle's general restatement against an
    explicit per-target loop: projected edge features join the key and the value; two heads, concatenated."""
    import math
    from oracle.pyg_restatement import grand_plus_general
    torch.manual_seed(0)
    n, c, H, ed = 7, 4, 2, 3
    x = torch.randn(n, H * c, dtype=torch.float64)
    ei = torch.tensor([[0, 1, 2, 3, 4, 5, 6, 0, 2, 4], [1, 2, 3, 4, 5, 6, 0, 3, 5, 1]])
    wq, wk = torch.randn(H * c, H * c, dtype=torch.float64), torch.randn(H * c, H * c, dtype=torch.float64)
    bq, bk = torch.randn(H * c, dtype=torch.float64), torch.randn(H * c, dtype=torch.float64)
    we, ea = torch.randn(H * c, ed, dtype=torch.float64), torch.randn(ei.shape[1], ed, dtype=torch.float64)
    res = grand_plus_general(x, ei, wq, bq, wk, bk, heads=H, concat=True, w_edge=we, edge_attr=ea)
    q, k, v = (x @ wq.T + bq).view(n, H, c), (x @ wk.T + bk).view(n, H, c), x.view(n, H, c)
    e = (ea @ we.T).view(-1, H, c)
    out = torch.zeros(n, H, c, dtype=torch.float64)
    for i in range(n):
        es = [t for t in range(ei.shape[1]) if ei[1, t] == i]
        for h in range(H):
            s = torch.stack([(q[i, h] * (k[ei[0, t], h] + e[t, h])).sum() / math.sqrt(c) for t in es])
            for w_, t in zip(torch.softmax(s, 0), es):
                out[i, h] += w_ * (v[ei[0, t], h] + e[t, h])
    assert (out.view(n, -1) - x - res).abs().max().item() <= 1e-12
    # a bare edge_attr (no lin_edge) joins the value only (:339)
    bare = torch.randn(ei.shape[1], H * c, dtype=torch.float64)
    res2 = grand_plus_general(x, ei, wq, bq, wk, bk, heads=H, concat=True, edge_attr=bare)
    out2 = torch.zeros(n, H, c, dtype=torch.float64)
    for i in range(n):
        es = [t for t in range(ei.shape[1]) if ei[1, t] == i]
        for h in range(H):
            s = torch.stack([(q[i, h] * k[ei[0, t], h]).sum() / math.sqrt(c) for t in es])
            for w_, t in zip(torch.softmax(s, 0), es):
                out2[i, h] += w_ * (v[ei[0, t], h] + bare[t].view(H, c)[h])
    assert (out2.view(n, -1) - x - res2).abs().max().item() <= 1e-12


def test_oracle_applies_the_input_dropout_of_the_reference():
    """`src/GNN.py:271`: `x = F.dropout(enc(features), opt['dropout'], training=self.training)` - the identity at the shipped p = 0 and
    in evaluation, active in training at p > 0 (the restatement was missing the op)."""
    from g_adaptivity_amd import MeshDataset, collate, hot_path_opt
    from oracle.pyg_restatement import OracleGNN
    ds = MeshDataset([6, 6], 2, seed=0)
    data = collate(ds.samples)
    outs = {}
    for p_drop, train in ((0.0, True), (0.5, False), (0.5, True)):
        opt = hot_path_opt(mesh_dims=[6, 6], hidden_dim=8, num_layers=2, dropout=p_drop, device='cpu')
        torch.manual_seed(0)
        m = OracleGNN(ds, opt)
        m.train(train)
        torch.manual_seed(1)
        outs[(p_drop, train)] = m(data).detach()
    assert torch.equal(outs[(0.0, True)], outs[(0.5, False)])            # p = 0 and evaluation: the identity
    assert not torch.equal(outs[(0.0, True)], outs[(0.5, True)])         # training at p = 0.5: the encoder output is dropped out

