"""GPU tests of the generic message-passing primitives and of the conv variants built on them (pytest -m gpu):
`GAT_plus`, `GAT`, `GCN` behind `get_conv` (`src/GNN.py:108-124`), `reg_skew`, `learnable_v`, hidden sizes the fused
kernels are not built for.  Reference = the CPU oracle (gather / scatter restatement of the PyG op sequence)."""
import math

import pytest
import torch
import torch.nn.functional as F

from g_adaptivity_amd import GNN, MeshDataset, MeshGraph, collate, get_conv, hot_path_opt
from g_adaptivity_amd import sparse_ops as Sp
from helpers import hip_model_like, make_case, oracle_fp64_twin, rel_err
from oracle.pyg_restatement import pyg_softmax

COORD_TOL, GRAD_TOL = 1e-5, 1e-4


def _graph(n, e, seed, dev):
    g = torch.Generator().manual_seed(seed)
    ei = torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])
    ei = torch.cat([ei, torch.tensor([[7] * 30, list(range(30))]), torch.tensor([list(range(40, 80)), [9] * 40])], 1)   # hub out / hub in
    ei = ei[:, ei[1] != 5]                                             # node 5: no in-edge
    return ei, MeshGraph(ei, n, dev)


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("C", [3, 8, 64, 100, 260])
def test_spmm_sddmm_forward_backward(gpu_device, C):
    """spmm / sddmm and their transposed backward products against index_select + index_add_ on the CPU (fp64), rows of
    0 / 1 / 40 entries, widths that are not a multiple of 4 and beyond one wave's 64 lanes x 4 floats."""
    n = 300
    ei, graph = _graph(n, 2000, 3, gpu_device)
    e = ei.shape[1]
    gen = torch.Generator().manual_seed(C)
    w = torch.randn(e, generator=gen, dtype=torch.float64)
    x, a = torch.randn(n, C, generator=gen, dtype=torch.float64), torch.randn(n, C, generator=gen, dtype=torch.float64)
    up, upe = torch.randn(n, C, generator=gen, dtype=torch.float64), torch.randn(e, generator=gen, dtype=torch.float64)
    order = graph.eid_t.long().cpu()                                   # target-CSR slot -> caller's edge id
    # reference (caller's edge order)
    wr, xr, ar = w.clone().requires_grad_(True), x.clone().requires_grad_(True), a.clone().requires_grad_(True)
    out_ref = torch.zeros(n, C, dtype=torch.float64).index_add_(0, ei[1], wr[:, None] * xr.index_select(0, ei[0]))
    s_ref = (ar.index_select(0, ei[1]) * xr.index_select(0, ei[0])).sum(-1)
    ((out_ref * up).sum() + (s_ref * upe).sum()).backward()
    # HIP (edge arrays in target-CSR order)
    wh = w[order].float().to(gpu_device).requires_grad_(True)
    xh, ah = x.float().to(gpu_device).requires_grad_(True), a.float().to(gpu_device).requires_grad_(True)
    out = Sp.spmm(graph, wh, xh)
    s = Sp.sddmm(graph, ah, xh)
    ((out * up.float().to(gpu_device)).sum() + (s * upe[order].float().to(gpu_device)).sum()).backward()
    torch.cuda.synchronize()
    assert rel_err(out, out_ref)[0] <= 2e-6 and rel_err(s, s_ref[order])[0] <= 2e-6
    assert rel_err(wh.grad, wr.grad[order])[0] <= 2e-6
    assert rel_err(xh.grad, xr.grad)[0] <= 2e-6 and rel_err(ah.grad, ar.grad)[0] <= 2e-6
    assert out[5].abs().max().item() == 0.0                            # empty row
    plain = Sp.spmm(graph, None, xh.detach())                          # w = None: plain neighbour sum
    assert rel_err(plain, torch.zeros(n, C, dtype=torch.float64).index_add_(0, ei[1], x.index_select(0, ei[0])))[0] <= 2e-6
    assert torch.equal(Sp.spmm(graph, wh.detach(), xh.detach()), out.detach())   # bit-reproducible


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_edge_softmax_and_edge_combine(gpu_device):
    n = 300
    ei, graph = _graph(n, 2000, 4, gpu_device)
    e = ei.shape[1]
    order = graph.eid_t.long().cpu()
    gen = torch.Generator().manual_seed(0)
    s = torch.randn(e, generator=gen, dtype=torch.float64) * 3
    s[ei[1] == 9] += torch.linspace(-40, 40, int((ei[1] == 9).sum()), dtype=torch.float64)   # a long row with a wide score range
    u, v = torch.randn(n, generator=gen, dtype=torch.float64), torch.randn(n, generator=gen, dtype=torch.float64)
    up = torch.randn(e, generator=gen, dtype=torch.float64)
    sr, ur, vr = s.clone().requires_grad_(True), u.clone().requires_grad_(True), v.clone().requires_grad_(True)
    a_ref = pyg_softmax(sr.unsqueeze(-1), ei[1], n).squeeze(-1)
    add_ref = ur.index_select(0, ei[0]) + vr.index_select(0, ei[1])
    mul_ref = ur.index_select(0, ei[0]) * vr.index_select(0, ei[1])
    ((a_ref * up).sum() + (add_ref * up).sum() + (mul_ref * up * up).sum()).backward()
    sh = s[order].float().to(gpu_device).requires_grad_(True)
    uh, vh = u.float().to(gpu_device).requires_grad_(True), v.float().to(gpu_device).requires_grad_(True)
    uph = up[order].float().to(gpu_device)
    a = Sp.edge_softmax(graph, sh)
    add, mul = Sp.edge_add(graph, uh, vh), Sp.edge_mul(graph, uh, vh)
    ((a * uph).sum() + (add * uph).sum() + (mul * uph * uph).sum()).backward()
    torch.cuda.synchronize()
    assert rel_err(a, a_ref[order])[0] <= 2e-6 and rel_err(add, add_ref[order])[0] <= 1e-6 and rel_err(mul, mul_ref[order])[0] <= 1e-6
    assert rel_err(sh.grad, sr.grad[order])[0] <= 5e-6
    assert rel_err(uh.grad, ur.grad)[0] <= 5e-6 and rel_err(vh.grad, vr.grad)[0] <= 5e-6
    rows = torch.zeros(n, device=gpu_device).index_add_(0, graph.edge_index[1].to(gpu_device), graph.alpha_to_edge_order(a.detach()))
    has_in = torch.bincount(ei[1], minlength=n) > 0
    assert (rows.cpu()[has_in] - 1).abs().max().item() <= 2e-6 and rows.cpu()[~has_in].abs().max().item() == 0.0


def _model_parity(gpu_device, mesh_dims, batch, hidden, layers, conv_type, **extra):
    opt, ds, data, oracle = make_case(mesh_dims, batch, hidden, layers, conv_type, **extra)
    model = hip_model_like(oracle, ds, opt, gpu_device)
    tgt = data.x_phys
    ref = oracle(data)
    F.mse_loss(ref, tgt).backward()
    o64, ref64 = oracle_fp64_twin(oracle, ds, opt, data, tgt)
    out = model(data.clone().to(gpu_device))
    F.mse_loss(out, tgt.to(gpu_device)).backward()
    torch.cuda.synchronize()
    assert rel_err(out, ref)[0] <= COORD_TOL and rel_err(out, ref64)[0] <= COORD_TOL, (rel_err(out, ref), rel_err(out, ref64))
    d32, d64 = dict(oracle.named_parameters()), dict(o64.named_parameters())
    checked = []
    for name, ph in model.named_parameters():
        if name not in d64 or d64[name].grad is None:
            assert ph.grad is None or name.endswith('lin_key.bias'), name
            continue
        if name.endswith('lin_key.bias'):
            continue
        e64, noise = rel_err(ph.grad, d64[name].grad)[0], rel_err(d32[name].grad, d64[name].grad)[0]
        assert e64 <= max(GRAD_TOL, 1.5 * noise), f"{name}.grad vs fp64 oracle: {e64:.2e} (fp32 oracle: {noise:.2e})"
        checked.append(name)
    return model, oracle, checked


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=['fused-block', 'per-layer-primitives'])
@pytest.mark.parametrize("kind,non_lin,share,hidden,residual", [
    ('GAT_res_lap', 'identity', True, 16, True), ('GAT_res_lap', 'tanh', False, 16, True), ('GAT_lin', 'relu', True, 16, True),
    ('GAT_res_lap', 'identity', True, 64, True), ('GAT_res_lap', 'selu', False, 64, True), ('GAT_lin', 'sigmoid', True, 128, True),
    ('GAT_res_lap', 'elu', True, 32, False), ('GAT_res_lap', 'leaky_relu', False, 8, True), ('GAT_res_lap', 'identity', True, 4, True)])
def test_gat_plus_parity(gpu_device, kind, non_lin, share, hidden, residual, fused):
    """`conv_type='GAT_plus'` (`src/GRAND_plus.py:386-416`): additive attention, default self-loops, A^T x - x.  Both routes
    against the oracle: the fused block op of csrc/gadapt_gat.inc (what `GNN` runs) and the per-layer generic primitives it
    replaced (dropout / learn_step still take them).  Coordinates and every parameter gradient."""
    model, oracle, checked = _model_parity(gpu_device, (12, 12), 3, hidden, 3, 'GAT_plus', gat_plus_type=kind, non_lin=non_lin, share_conv=share,
                                           residual=residual, fused_gat_plus=fused)
    assert model._gat_plus_fusable() == fused
    assert sorted(n.split('.')[-1] for n in checked if n.startswith('conv_layers.0.')) == ['att_dst', 'att_src']
    layer = model.conv_layers[0]
    n = 3 * 144
    assert layer.stored_ei.shape[1] == layer.stored_alpha.shape[0] and layer.stored_alpha.shape[1] == 1
    assert torch.equal(layer.stored_ei[:, -n:].cpu(), torch.arange(n).repeat(2, 1))      # add_self_loops appends one loop per node


@pytest.mark.gpu
@pytest.mark.parametrize("conv,hidden,non_lin", [('GAT', 16, 'relu'), ('GAT', 64, 'identity'), ('GCN', 8, 'tanh'), ('GCN', 64, 'relu')])
def test_stock_gat_and_gcn_parity(gpu_device, conv, hidden, non_lin):
    """`get_conv(opt, 'GAT' | 'GCN', ...)` = stock GATConv / GCNConv (`src/GNN.py:109-111`), layer by layer with non_lin and
    the residual update (`src/GNN.py:284-291`); coordinates and every parameter gradient against the oracle."""
    model, oracle, checked = _model_parity(gpu_device, (11, 11), 2, hidden, 3, conv, non_lin=non_lin)
    want = {'GAT': ['att_dst', 'att_src', 'bias', 'weight'], 'GCN': ['bias', 'weight']}[conv]
    assert sorted({n.split('.')[-1] for n in checked}) == want
    keys = set(model.state_dict())
    if conv == 'GAT':
        assert {'conv_layers.0.lin_src.weight', 'conv_layers.0.lin_dst.weight', 'conv_layers.0.att_src', 'conv_layers.0.bias'} <= keys
    else:
        assert {'conv_layers.0.lin.weight', 'conv_layers.0.bias'} <= keys


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [12, 24, 200])
def test_hidden_sizes_outside_the_fused_set(gpu_device, hidden):
    """The reference accepts any hidden_dim; widths the fused kernels are not built for run the same arithmetic through the
    generic primitives (GRAND_plus and GRAND)."""
    for conv in ('GRAND_plus', 'GRAND'):
        _model_parity(gpu_device, (10, 10), 2, hidden, 2, conv)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2])
def test_reg_skew_parity(gpu_device, batch):
    """`reg_skew` (`src/GRAND_plus.py:280-324`): scores times the summed area of the oriented triangles on the edge, areas from
    the layer's own input (gradients flow through them).  `mesh` is the dataset's single mesh, so in a batch only the first
    graph's edges find triangles and every other edge gets weight 0 - the reference's behaviour, reproduced."""
    model, oracle, checked = _model_parity(gpu_device, (9, 9), batch, 16, 3, 'GRAND_plus', reg_skew=True)
    assert any(n.endswith('lin_query.weight') for n in checked)
    # it does change the result
    opt, ds, data, plain = make_case((9, 9), batch, 16, 3, 'GRAND_plus', reg_skew=False)
    plain.load_state_dict(oracle.state_dict())
    with torch.no_grad():
        assert (plain(data) - oracle(data)).abs().max().item() > 1e-7


@pytest.mark.gpu
def test_learnable_v_fails_as_in_the_reference(gpu_device):
    """`softmax_temp_type='learnable_v'` (`src/GRAND_plus.py:158-160,330-331`) builds `sm_temp_v = Linear(C, heads)` and applies it
    to the [E, heads] scores: a shape error in the reference for every hidden size.  Same here: the parameter exists (same
    state_dict key), the forward raises."""
    opt = hot_path_opt(mesh_dims=[9, 9], hidden_dim=16, num_layers=2, softmax_temp_type='learnable_v', device=str(gpu_device))
    ds = MeshDataset([9, 9], 2, seed=0)
    model = GNN(ds, opt).to(gpu_device)
    assert 'conv_layers.0.sm_temp_v.weight' in model.state_dict() and model.state_dict()['conv_layers.0.sm_temp_v.weight'].shape == (1, 16)
    with pytest.raises(RuntimeError, match="learnable_v"):
        model(collate(ds.samples).to(gpu_device))
    with pytest.raises(NotImplementedError):
        get_conv(opt, 'Laplacian', 16, 16)                             # GNN.py:122-124


@pytest.mark.gpu
@pytest.mark.parametrize("heads,root,beta,temp", [(2, False, False, None), (4, True, False, None), (2, True, True, 'learnable_a'),
                                                  (1, True, True, None), (1, True, False, 'fixed')],
                         ids=['H2', 'H4-root', 'H2-root-beta-learnT', 'H1-root-beta', 'H1-root-fixedT'])
def test_grand_plus_conv_constructor_options(gpu_device, heads, root, beta, temp):
    """`GRAND_plusConv` with the constructor options `get_conv` never passes (`src/GRAND_plus.py:114-183,239-250`): several
    heads over slices of x, concatenation, `root_weight`, `beta`, a per-head learnable temperature.  Forward and every parameter
    gradient against the oracle's restatement in fp64, on a mesh batch and on a random graph with long rows."""
    import torch.nn.functional as F
    from g_adaptivity_amd import GRAND_plusConv, MeshDataset, collate, hot_path_opt
    from oracle.pyg_restatement import grand_plus_general, masked_edge_index
    from helpers import rel_err
    c = 16
    width = heads * c
    opt = hot_path_opt(mesh_dims=[11, 11], hidden_dim=width, softmax_temp_type=temp, softmax_temp=1.7, device=str(gpu_device))
    ds = MeshDataset([11, 11], 2, seed=3)
    g = torch.Generator().manual_seed(7)
    graphs = [masked_edge_index(collate(ds.samples), 2, 11), torch.stack([torch.randint(0, 300, (4000,), generator=g), torch.randint(0, 300, (4000,), generator=g)])]
    for ei, n in zip(graphs, (242, 300)):
        torch.manual_seed(5)
        conv = GRAND_plusConv(opt, width, c, heads=heads, concat=True, beta=beta, root_weight=root, bias=True).to(gpu_device)
        if temp == 'learnable_a':
            with torch.no_grad():
                conv.sm_temp_a.copy_(torch.linspace(0.8, 2.1, heads).view(1, heads, 1))
        x = torch.randn(n, width, generator=g)
        up = torch.randn(n, width, generator=g)
        xg = x.to(gpu_device).requires_grad_(True)
        res = conv(xg, ei.to(gpu_device))
        (res * up.to(gpu_device)).sum().backward()
        torch.cuda.synchronize()
        # oracle, fp64
        P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in conv.named_parameters()}
        x64 = x.double().requires_grad_(True)
        t = None
        if temp == 'fixed':
            t = 1.7
        elif temp == 'learnable_a':
            t = P['sm_temp_a']
        ref = grand_plus_general(x64, ei, P['lin_query.weight'], P['lin_query.bias'], P['lin_key.weight'], P['lin_key.bias'], heads=heads,
                                 concat=True, w_skip=P['lin_skip.weight'] if root else None, b_skip=P['lin_skip.bias'] if root else None,
                                 w_beta=P['lin_beta.weight'] if (beta and root) else None, temperature=t)
        (ref * up.double()).sum().backward()
        assert rel_err(res.detach().cpu(), ref.detach())[0] <= 1e-5
        assert rel_err(xg.grad.cpu(), x64.grad)[0] <= 2e-5
        for k, v in conv.named_parameters():
            if k.startswith('lin_skip') and not root:
                assert v.grad is None
                continue
            if k == 'lin_key.bias':                                 # vanishes analytically (softmax shift invariance): rounding noise on both sides
                assert v.grad.abs().max().item() <= 1e-5 * conv.lin_query.bias.grad.abs().max().item()
                continue
            assert rel_err(v.grad.cpu(), P[k].grad)[0] <= 1e-4, (k, rel_err(v.grad.cpu(), P[k].grad))
        assert conv.stored_alpha.shape == (ei.shape[1], heads)


@pytest.mark.gpu
def test_grand_plus_conv_attention_dropout(gpu_device):
    """Attention dropout (`src/GRAND_plus.py:336`): in eval mode it is the identity; in training mode the output equals the
    oracle's with the mask F.dropout drew - recovered from the kept attention (stored_alpha) and the aggregation itself is
    linear in the masked attention, so two draws differ while their common expectation is the eval output."""
    from g_adaptivity_amd import GRAND_plusConv, MeshDataset, collate, hot_path_opt
    from oracle.pyg_restatement import masked_edge_index
    from helpers import rel_err
    opt = hot_path_opt(mesh_dims=[11, 11], hidden_dim=32, device=str(gpu_device))
    ds = MeshDataset([11, 11], 2, seed=3)
    ei = masked_edge_index(collate(ds.samples), 2, 11).to(gpu_device)
    torch.manual_seed(2)
    drop = GRAND_plusConv(opt, 32, 32, heads=1, concat=False, dropout=0.5, root_weight=False, bias=False).to(gpu_device)
    plain = GRAND_plusConv(opt, 32, 32, heads=1, concat=False, dropout=0.0, root_weight=False, bias=False).to(gpu_device)
    plain.load_state_dict(drop.state_dict())
    x = torch.randn(242, 32, device=gpu_device)
    drop.eval(); plain.eval()
    assert rel_err(drop(x, ei), plain(x, ei))[0] <= 2e-6            # generic primitives vs the fused kernel, no dropout in eval
    drop.train()
    a, b = drop(x, ei), drop(x, ei)
    assert not torch.equal(a, b)
    mean = torch.stack([drop(x, ei) for _ in range(400)]).mean(0)
    assert rel_err(mean + x, plain(x, ei) + x)[0] <= 0.1            # E[dropout(alpha)] = alpha


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("conv", ['GAT_plus', 'GAT', 'GCN', 'GRAND'])
def test_step_captures_after_eager_steps_on_the_default_stream(gpu_device, conv):
    """Regression (VERDICT r3 missing #3): `bench.py` of the GAT_plus workload segfaulted inside hipStreamEndCapture.  Cause: the
    convs kept `stored_alpha` WITH its autograd graph, so the previous (eager, default-stream) step's graph - and with it the
    parameters' gradient accumulators, bound to the default stream - was still alive when the step was captured on a side stream.
    The exact flow (two eager steps on the default stream, side-stream warm-up, capture of zero_grad + forward + loss + backward +
    Adam with the preallocated root gradient, replay), in a child process: a regression is a segfault."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PROBE_LAYERS='3', PROBE_EAGER_FIRST='2', PROBE_WARM='1', PROBE_ROOT='1', PROBE_ZERO_OUTSIDE='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'capture_probe_convs.py'), conv, 'step', '16', '4'],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and 'replayed ok' in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
@pytest.mark.parametrize("C", [8, 64, 128])
def test_gat_plus_block_on_any_graph(gpu_device, C):
    """The fused GAT_plus block at operator level on a random graph with rows longer than 8 (the loop paths of all three kernels),
    isolated nodes and a ragged node count, against the same layers composed from the generic primitives in fp64-free torch
    autograd (`sparse_ops`): output, attention, d x0 and both parameter gradients, shared and per-layer vectors; bit-reproducible."""
    import math
    from g_adaptivity_amd import functional as Fn, sparse_ops as Sp
    from g_adaptivity_amd.graph import MeshGraph
    n, L, dt = 777, 3, 0.1
    gen = torch.Generator().manual_seed(5)
    ei = torch.randint(0, n, (2, 4000), generator=gen)
    ei = torch.cat([ei, torch.tensor([[3] * 40, list(range(100, 140))]), torch.tensor([list(range(200, 230)), [11] * 30])], dim=1)
    ei = ei[:, (ei[1] != 17) & (ei[0] != ei[1])]
    looped = MeshGraph(ei, n, gpu_device).with_self_loops()
    assert looped.max_in_degree > 8
    x0 = torch.randn(n, C, generator=gen).to(gpu_device)
    up = torch.randn(n, C, generator=gen).to(gpu_device)
    for S in (1, L):
        src0 = (torch.randn(S, C, generator=gen) / math.sqrt(C)).to(gpu_device)
        dst0 = (torch.randn(S, C, generator=gen) / math.sqrt(C)).to(gpu_device)
        for nl, res_lap in (('identity', True), ('tanh', True), ('relu', False)):
            def run(fused):
                xr = x0.clone().requires_grad_(True)
                s_, d_ = src0.clone().requires_grad_(True), dst0.clone().requires_grad_(True)
                if fused:
                    y, alpha = Fn.gat_plus_block(xr, s_, d_, looped, L, dt, True, res_lap, nl)
                    alpha_last = alpha[-1]
                else:
                    y = xr
                    for l in range(L):
                        k = l if S > 1 else 0
                        a_src, a_dst = (y * s_[k]).sum(-1), (y * d_[k]).sum(-1)
                        alpha_last = Sp.edge_softmax(looped, torch.nn.functional.leaky_relu(Sp.edge_add(looped, a_src, a_dst), 0.2))
                        m = Sp.spmm(looped, alpha_last, y)
                        r = {'identity': lambda t: t, 'tanh': torch.tanh, 'relu': torch.relu}[nl](m - y if res_lap else m)
                        y = y + dt * r
                (y * up).sum().backward()
                torch.cuda.synchronize()
                return y.detach(), alpha_last.detach(), xr.grad, s_.grad, d_.grad
            a, b = run(True), run(False)
            for name, u, v, tol in zip(('x_L', 'alpha', 'd x0', 'd att_src', 'd att_dst'), a, b, (2e-6, 2e-6, 2e-5, 5e-5, 5e-5)):
                assert rel_err(u, v)[0] <= tol, (S, nl, name, rel_err(u, v))
            again = run(True)
            assert all(torch.equal(u, v) for u, v in zip(a, again))


@pytest.mark.gpu
@pytest.mark.parametrize("heads,c,edge_dim,root", [(1, 16, 5, False), (2, 8, 3, True), (1, 6, 4, False), (1, 16, None, False)],
                         ids=['H1-C16-E5', 'H2-C8-E3-root', 'H1-C6-E4', 'bare-edge_attr'])
def test_grand_plus_conv_edge_dim(gpu_device, heads, c, edge_dim, root):
    """`GRAND_plusConv(edge_dim=...)` (`src/GRAND_plus.py:165-166`): `lin_edge(edge_attr)` is added to the key (:273-277) and to
    the value (:338-340); a bare `edge_attr` without `lin_edge` joins the value only (:339).  `get_conv` never passes either
    (`src/GNN.py:119`).  Forward and every gradient - parameters, x and edge_attr - against the oracle's restatement in fp64, on a
    mesh batch and on a random graph with long rows; a missing edge_attr raises AssertionError as at :274."""
    from g_adaptivity_amd import GRAND_plusConv, MeshDataset, collate, hot_path_opt
    from oracle.pyg_restatement import grand_plus_general, masked_edge_index
    from helpers import rel_err
    width = heads * c
    opt = hot_path_opt(mesh_dims=[11, 11], hidden_dim=width, device=str(gpu_device))
    ds = MeshDataset([11, 11], 2, seed=3)
    g = torch.Generator().manual_seed(11)
    graphs = [masked_edge_index(collate(ds.samples), 2, 11), torch.stack([torch.randint(0, 300, (4000,), generator=g), torch.randint(0, 300, (4000,), generator=g)])]
    for ei, n in zip(graphs, (242, 300)):
        torch.manual_seed(5)
        conv = GRAND_plusConv(opt, width, c, heads=heads, concat=True, beta=False, root_weight=root, bias=True, edge_dim=edge_dim).to(gpu_device)
        assert ('lin_edge.weight' in conv.state_dict()) == (edge_dim is not None)
        x = torch.randn(n, width, generator=g)
        up = torch.randn(n, width, generator=g)
        ea = 0.3 * torch.randn(ei.shape[1], edge_dim if edge_dim is not None else width, generator=g)
        xg, eg = x.to(gpu_device).requires_grad_(True), ea.to(gpu_device).requires_grad_(True)
        res = conv(xg, ei.to(gpu_device), edge_attr=eg)
        (res * up.to(gpu_device)).sum().backward()
        torch.cuda.synchronize()
        P = {k: v.detach().cpu().double().requires_grad_(True) for k, v in conv.named_parameters()}
        x64, e64 = x.double().requires_grad_(True), ea.double().requires_grad_(True)
        ref = grand_plus_general(x64, ei, P['lin_query.weight'], P['lin_query.bias'], P['lin_key.weight'], P['lin_key.bias'], heads=heads,
                                 concat=True, w_skip=P['lin_skip.weight'] if root else None, b_skip=P['lin_skip.bias'] if root else None,
                                 w_edge=P.get('lin_edge.weight'), edge_attr=e64)
        (ref * up.double()).sum().backward()
        assert rel_err(res.detach().cpu(), ref.detach())[0] <= 1e-5
        assert rel_err(xg.grad.cpu(), x64.grad)[0] <= 2e-5
        assert rel_err(eg.grad.cpu(), e64.grad)[0] <= 2e-5
        for k, v in conv.named_parameters():
            if k.startswith('lin_skip') and not root:
                assert v.grad is None
                continue
            if k == 'lin_key.bias':
                assert v.grad.abs().max().item() <= 1e-5 * conv.lin_query.bias.grad.abs().max().item()
                continue
            assert rel_err(v.grad.cpu(), P[k].grad)[0] <= 1e-4, (k, rel_err(v.grad.cpu(), P[k].grad))
        if edge_dim is not None:
            with pytest.raises(AssertionError):
                conv(xg.detach(), ei.to(gpu_device))
