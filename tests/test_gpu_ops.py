"""GPU tests of the operator surface, the small kernels, edge cases and full-size properties (pytest -m gpu)."""
import glob
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from g_adaptivity_amd import GNN, GRAND_conv, GRAND_plusConv, MeshDataset, MeshGraph, collate, hot_path_opt
from g_adaptivity_amd import functional as Fn
from g_adaptivity_amd.optim import FlatAdam
from helpers import hip_model_like, make_case, rel_err
from oracle.pyg_restatement import grand_residual

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', '*.npz')))


def _random_layer(C, seed, dev=None):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / math.sqrt(C)
    ws = [(torch.rand(C, C, generator=g) * 2 - 1) * k, (torch.rand(C, generator=g) * 2 - 1) * k,
          (torch.rand(C, C, generator=g) * 2 - 1) * k, (torch.rand(C, generator=g) * 2 - 1) * k]
    return [w.to(dev) if dev else w for w in ws]


def _random_graph(n, e, seed, max_in=None):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    return torch.stack([src, dst])


@pytest.mark.gpu
@pytest.mark.parametrize("C", [4, 8, 16, 32, 64, 128])
@pytest.mark.parametrize("kind", ["mesh", "random_long_rows", "ragged"])
def test_conv_residual_forward_backward_dense_x(gpu_device, C, kind):
    """Operator-level drop-in: GRAND_plusConv.forward(x, edge_index) on dense random x (all C columns live),
    including rows with > 8 in-edges (loop path), isolated nodes, and N not a multiple of the tile."""
    if kind == "mesh":
        ds = MeshDataset([13, 13], 3, seed=2)
        from oracle.pyg_restatement import masked_edge_index
        ei = masked_edge_index(collate(ds.samples), 2, 13)
        n = 3 * 169
    elif kind == "random_long_rows":
        n = 333
        ei = _random_graph(n, 5000, 7)                      # mean in-degree 15: every tile takes the loop path
    else:
        n = 1000
        ei = torch.cat([_random_graph(n, 2500, 9), torch.tensor([[5] * 40, list(range(40))])], dim=1)
        ei = ei[:, ei[1] != 17]                             # node 17 has no in-edge at all
    wq, bq, wk, bk = _random_layer(C, 3)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(4))
    up = torch.randn(n, C, generator=torch.Generator().manual_seed(5))

    xr = x.clone().requires_grad_(True)
    pr = [w.clone().requires_grad_(True) for w in (wq, bq, wk, bk)]
    ref, (alpha_ref, q_ref, k_ref) = grand_residual(xr, ei, pr[0], pr[1], pr[2], pr[3], return_attention=True)
    (ref * up).sum().backward()
    x64 = x.double().requires_grad_(True)
    p64 = [w.double().requires_grad_(True) for w in (wq, bq, wk, bk)]
    ref64 = grand_residual(x64, ei, *p64)
    (ref64 * up.double()).sum().backward()

    opt = hot_path_opt(hidden_dim=C, show_mesh_evol_plots=True, device=str(gpu_device))
    conv = GRAND_plusConv(opt, C, C, global_feat_dim=8, heads=1, concat=False, beta=False, dropout=0.0, edge_dim=None,
                          bias=False, root_weight=False).to(gpu_device)
    with torch.no_grad():
        conv.lin_query.weight.copy_(wq); conv.lin_query.bias.copy_(bq); conv.lin_key.weight.copy_(wk); conv.lin_key.bias.copy_(bk)
    xh = x.to(gpu_device).requires_grad_(True)
    res, (ei_out, (alpha, q, k)) = conv(xh, ei.to(gpu_device), None, None, return_attention_weights=True)
    (res * up.to(gpu_device)).sum().backward()
    torch.cuda.synchronize()

    assert rel_err(res, ref64)[0] <= 1e-5, rel_err(res, ref64)
    assert rel_err(alpha.view(-1), alpha_ref.view(-1))[0] <= 1e-5
    assert rel_err(conv.stored_alpha.view(-1), alpha_ref.view(-1))[0] <= 1e-5 and conv.stored_ei is not None
    assert rel_err(q, q_ref)[0] <= 1e-5 and rel_err(k, k_ref)[0] <= 1e-5
    assert rel_err(xh.grad, x64.grad)[0] <= 1e-4, rel_err(xh.grad, x64.grad)
    for got, want in ((conv.lin_query.weight.grad, p64[0].grad), (conv.lin_query.bias.grad, p64[1].grad),
                      (conv.lin_key.weight.grad, p64[2].grad)):
        assert rel_err(got, want)[0] <= 1e-4, rel_err(got, want)


@pytest.mark.gpu
def test_grand_conv_surface(gpu_device):
    C = 64
    opt = hot_path_opt(hidden_dim=C, device=str(gpu_device))
    conv = GRAND_conv(opt, C, C, heads=1).to(gpu_device)
    ei = _random_graph(200, 900, 1).to(gpu_device)
    x = torch.randn(200, C, device=gpu_device)
    res = conv(x, ei)
    ref = grand_residual(x.cpu(), ei.cpu(), conv.lin_query.weight.detach().cpu(), conv.lin_query.bias.detach().cpu(),
                         conv.lin_key.weight.detach().cpu(), conv.lin_key.bias.detach().cpu())
    assert rel_err(res, ref)[0] <= 1e-5
    assert conv.stored_alpha.shape == (900, 1) and conv.stored_ei is ei


@pytest.mark.gpu
def test_learnable_step_and_temperature_gradients(gpu_device):
    opt, ds, data, oracle = make_case((12, 12), 2, 64, 3, learn_step=True, softmax_temp_type='fixed', softmax_temp=1.5)
    model = hip_model_like(oracle, ds, opt, gpu_device)
    tgt = data.x_phys
    F.mse_loss(oracle(data), tgt).backward()
    F.mse_loss(model(data.clone().to(gpu_device)), tgt.to(gpu_device)).backward()
    for l in range(3):
        want, got = oracle.steps[l].grad, model.steps[l].grad
        assert rel_err(got, want)[0] <= 1e-4, (l, got, want)
    # temperature as a parameter: d/dT through the score scale
    C = 32
    wq, bq, wk, bk = _random_layer(C, 11)
    ei = _random_graph(300, 1500, 12)
    x = torch.randn(300, C, generator=torch.Generator().manual_seed(13))
    T = torch.tensor(1.7, dtype=torch.float64, requires_grad=True)
    ref = grand_residual(x.double(), ei, wq.double(), bq.double(), wk.double(), bk.double(), temperature=T)
    ref.square().sum().backward()
    o = hot_path_opt(hidden_dim=C, softmax_temp_type='learnable_a', device=str(gpu_device))
    conv = GRAND_plusConv(o, C, C, heads=1, concat=False, root_weight=False, bias=False).to(gpu_device)
    with torch.no_grad():
        conv.lin_query.weight.copy_(wq); conv.lin_query.bias.copy_(bq); conv.lin_key.weight.copy_(wk); conv.lin_key.bias.copy_(bk)
        conv.sm_temp_a.fill_(1.7)
    conv(x.to(gpu_device), ei.to(gpu_device)).square().sum().backward()
    assert rel_err(conv.sm_temp_a.grad.view(-1), T.grad.view(-1))[0] <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:2] for p in GOLDEN])
def test_golden_fixtures(gpu_device, path):
    g = np.load(path, allow_pickle=False)
    mesh_dims = [int(v) for v in g['mesh_dims']]
    C, L = int(g['hidden']), int(g['layers'])
    graph = MeshGraph(torch.from_numpy(g['edge_index'].astype(np.int64)), g['x_comp'].shape[0], gpu_device)
    x_comp = torch.from_numpy(g['x_comp'])
    extras = ([torch.from_numpy(g['f'])] if ('inc_f' not in g or int(g['inc_f'])) else []) + \
             ([torch.from_numpy(g['uu'])] if ('inc_uu' not in g or int(g['inc_uu'])) else [])
    feats = torch.stack(([x_comp] if x_comp.dim() == 1 else list(x_comp.T)) + extras, 1)
    enc = torch.zeros(C, feats.shape[1]); enc[torch.arange(feats.shape[1]), torch.arange(feats.shape[1])] = 1
    x0 = Fn.encode_linear(feats.to(gpu_device), enc.to(gpu_device))
    params = [torch.from_numpy(g[k]).to(gpu_device).unsqueeze(0).requires_grad_(True) for k in ('wq', 'bq', 'wk', 'bk')]
    lp = torch.tensor([[0.1, 1.0 / math.sqrt(C)]] * L, device=gpu_device)
    xL, alpha = Fn.grand_euler_block(x0, *params, lp, graph, L, want_alpha=True)
    d = len(mesh_dims)
    tgt = torch.from_numpy(g['target']).reshape(-1, d).to(gpu_device)
    F.mse_loss(xL[:, :d], tgt).backward()
    assert rel_err(xL[:, :d], torch.from_numpy(g['x_phys_f64']))[0] <= 1e-5
    assert rel_err(graph.alpha_to_edge_order(alpha[-1]), torch.from_numpy(g['alpha_last_f64']))[0] <= 1e-5
    for p, k in zip(params[:3], ('d_wq', 'd_bq', 'd_wk')):
        want64, want32 = torch.from_numpy(g[k + '_f64']), torch.from_numpy(g[k + '_f32'])
        # the fp32 oracle's own error against its fp64 twin: the worst over five edge orders of the fixture's batch (`band_*`,
        # written by make_golden.py) - the reference's run-to-run band, tests/test_gpu_parity.py docstring.  One rule, every fixture.
        noise = max(rel_err(want32, want64)[0], float(g['band_' + k]))
        assert rel_err(p.grad[0], want64)[0] <= max(1e-4, 1.5 * noise), (k, rel_err(p.grad[0], want64)[0], noise)


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("capturable", [False, True], ids=['host-step', 'device-step'])
def test_flat_adam_matches_torch_adam(gpu_device, capturable):
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(64, 64, device=gpu_device)), torch.nn.Parameter(torch.randn(64, device=gpu_device)),
          torch.nn.Parameter(torch.randn(3, device=gpu_device))]           # the last one never gets a gradient
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ours, ref = FlatAdam(ps, lr=1e-2, weight_decay=0.01, capturable=capturable), torch.optim.Adam(qs, lr=1e-2, weight_decay=0.01)
    for step in range(5):
        gs = [torch.randn_like(ps[0]), torch.randn_like(ps[1])]
        ours.zero_grad(); ref.zero_grad()
        for p, q, g in zip(ps, qs, gs):
            if p.grad is None: p.grad = g.clone()
            else: p.grad.add_(g)
            q.grad = g.clone()
        ours.step(); ref.step()
    torch.cuda.synchronize()
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    assert ours.grad_bucket.numel() == 64 * 64 + 64


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_mesh_loss_seed_kernel(gpu_device):
    from g_adaptivity_amd._native import check, current_stream, lib, ptr
    n, C, d = 1000, 64, 2
    x = torch.randn(n, C, device=gpu_device); tgt = torch.randn(n, d, device=gpu_device)
    for l1 in (0, 1):
        xp = torch.empty(n, d, device=gpu_device); g = torch.empty(n, C, device=gpu_device); loss = torch.zeros(1, device=gpu_device)
        check(lib().gadapt_mesh_loss_seed(ptr(x), ptr(tgt), ptr(xp), ptr(g), ptr(loss), n, d, C, l1, 1.0, current_stream(gpu_device)), 'seed')
        xr = x.clone().requires_grad_(True)
        ref = (F.l1_loss if l1 else F.mse_loss)(xr[:, :d], tgt)
        ref.backward()
        assert torch.equal(xp, x[:, :d]) and torch.allclose(loss[0], ref, rtol=1e-5)
        assert torch.allclose(g, xr.grad, rtol=1e-6, atol=1e-12)


# BASELINE.json configs at their full sizes: (mesh n, meshes, hidden, layers, conv_type, include f)
FULL_SIZES = [(64, 32, 64, 4, 'GRAND_plus', True),        # the metric workload = one rank's shard of config 3
              (32, 32, 64, 4, 'GRAND_plus', True),        # config 2
              (64, 32, 128, 6, 'GRAND', False),           # config 4: features [x, y, uu] (params.py:148,155)
              (128, 16, 64, 20, 'GRAND_plus', True)]      # config 5: 20 Euler steps, 128-node mesh rows (non-windowed tiles)
FULL_IDS = ['cfg3-64x64-b32-C64-L4', 'cfg2-32x32-b32-C64-L4', 'cfg4-64x64-b32-C128-L6-GRAND', 'cfg5-128x128-b16-C64-L20']


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("n,B,C,L,conv,inc_f", FULL_SIZES, ids=FULL_IDS)
def test_full_size_properties(gpu_device, n, B, C, L, conv, inc_f):
    """BASELINE config sizes, too big for the oracle to be quick: size-independent
    properties instead - attention rows sum to 1, corner nodes fixed, boundary nodes stay on their side,
    edge-order permutation invariance, dt = 0 is the identity, replay determinism."""
    opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=C, num_layers=L, conv_type=conv, gnn_inc_feat_f=inc_f, device=str(gpu_device),
                       show_mesh_evol_plots=True)
    ds = MeshDataset([n, n], B, seed=0)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(gpu_device).eval()
    with torch.no_grad():
        out = model(data)
        out2 = model(data)
    assert torch.equal(out, out2)                                        # deterministic
    x0 = data.x_comp
    graph = next(iter(model._graphs.values()))
    layer = model.conv_layers[0]
    alpha = layer.stored_alpha.view(-1)
    rows = torch.zeros(graph.num_nodes, device=gpu_device).index_add_(0, graph.edge_index[1].to(gpu_device), alpha)
    assert (rows - 1).abs().max().item() <= 2e-6
    assert alpha.min().item() >= 0.0 and alpha.max().item() <= 1.0 + 1e-6
    corners = torch.cat([torch.as_tensor(c) + b * n * n for b, c in enumerate(data.corner_nodes)]).to(gpu_device)
    assert torch.equal(out[corners], x0[corners])
    for col, val in ((0, 0.0), (0, 1.0), (1, 0.0), (1, 1.0)):
        side = x0[:, col] == val
        assert (out[side, col] - x0[side, col]).abs().max().item() <= 1e-6
    # permuted edge list -> same result within fp32 reassociation
    perm = torch.randperm(graph.num_edges, generator=torch.Generator().manual_seed(1))
    g2 = MeshGraph(graph.edge_index.cpu()[:, perm], graph.num_nodes, gpu_device)
    wq, bq, wk, bk = (layer.lin_query.weight.unsqueeze(0), layer.lin_query.bias.unsqueeze(0), layer.lin_key.weight.unsqueeze(0),
                      layer.lin_key.bias.unsqueeze(0))
    sc = 1.0 / math.sqrt(C)
    lp = torch.tensor([[0.1, sc]] * L, device=gpu_device)
    feats = torch.cat([x0] + ([data.f_tensor[:, None]] if inc_f else []) + [data.uu_tensor[:, None]], 1)
    xin = Fn.encode_linear(feats, model.enc.weight)
    with torch.no_grad():
        a, _ = Fn.grand_euler_block(xin, wq, bq, wk, bk, lp, graph, L)
        b, _ = Fn.grand_euler_block(xin, wq, bq, wk, bk, lp, g2, L)
        z, _ = Fn.grand_euler_block(xin, wq, bq, wk, bk, torch.tensor([[0.0, sc]] * L, device=gpu_device), graph, L)
    assert torch.equal(a[:, :2], out) and torch.equal(z, xin)
    assert torch.equal(a[:, feats.shape[1]:], torch.zeros_like(a[:, feats.shape[1]:]))   # zero-pad columns stay zero (convex combinations)
    for col in range(feats.shape[1]):                                    # live columns: x, y, (f,) uu
        assert rel_err(a[:, col], b[:, col])[0] <= 1e-5


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize('l1', [False, True])
def test_native_loss_matches_torch(gpu_device, l1):
    from g_adaptivity_amd import l1_loss, mse_loss
    torch.manual_seed(1)
    n, C, d = 4099, 16, 2
    x_top = torch.randn(n, C, device=gpu_device)
    tgt = torch.randn(n, d, device=gpu_device)
    for trial in range(3):                                  # the ticket counter must come back to zero every time
        a = x_top.clone().requires_grad_(True)
        b = x_top.clone().requires_grad_(True)
        ours = (l1_loss if l1 else mse_loss)(a[:, :d], tgt)             # strided view, as GNN returns it
        ref = (F.l1_loss if l1 else F.mse_loss)(b[:, :d], tgt)
        (3.0 * ours).backward(); (3.0 * ref).backward()
        assert torch.allclose(ours, ref, rtol=1e-5)
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-12)
    again = (l1_loss if l1 else mse_loss)(x_top[:, :d], tgt)
    assert torch.equal(again, ours.detach())                # fixed summation order: bit-identical repeats


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize('with_f,with_uu', [(True, True), (True, False), (False, True), (False, False)])
def test_encode_features_equals_concat_then_linear(gpu_device, with_f, with_uu):
    from g_adaptivity_amd import functional as Fn
    torch.manual_seed(2)
    n, dim, C = 1237, 2, 64
    xc = torch.randn(n, dim, device=gpu_device)
    f = torch.randn(n, device=gpu_device) if with_f else None
    uu = torch.randn(n, device=gpu_device) if with_uu else None
    cols = [xc] + [t.unsqueeze(-1) for t in (f, uu) if t is not None]
    feats = torch.cat(cols, dim=1)
    w = torch.randn(C, feats.shape[1], device=gpu_device)
    ours = Fn.encode_features(xc, f, uu, w)
    ref = Fn.encode_linear(feats, w)
    assert torch.equal(ours, ref)
    assert torch.allclose(ours, feats @ w.t(), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("C,L", [(64, 3), (64, 2), (16, 3), (128, 2), (8, 4)], ids=lambda v: str(v))
@pytest.mark.parametrize("kind", ["random_long_rows", "ragged_mesh"])
def test_compact_input_block_on_any_graph(gpu_device, C, L, kind):
    """The block op with the compact [N,4] layer-0 input through the C-ABI on graphs the model never builds: rows with more than
    8 in- and out-edges and isolated nodes (the loop paths of grand_bwd_target_compact_kernel and grand_bwd_source4_kernel, the
    slow tiles of the D4 target pass), N not a multiple of any tile.  Against the same op on the zero-padded dense [N,C] input:
    forward bit-identical, weight gradients to rounding; the top gradient is compact too (out_cols = 2)."""
    if kind == "random_long_rows":
        n = 1500
        ei = _random_graph(n, 15 * n, 21)
        ei = ei[:, ei[1] % 7 != 3]                                  # isolated targets
    else:
        from oracle.pyg_restatement import masked_edge_index
        ds = MeshDataset([19, 19], 3, seed=2)
        ei, n = masked_edge_index(collate(ds.samples), 2, 19), 3 * 361
    graph = MeshGraph(ei, n, gpu_device)
    gen = torch.Generator().manual_seed(5)
    feats = torch.rand(n, 4, generator=gen).to(gpu_device)
    wq, bq, wk, bk = [w.to(gpu_device).unsqueeze(0) for w in _random_layer(C, 31)]
    lp = torch.tensor([[0.1, 1.0 / math.sqrt(C)]] * L, device=gpu_device)
    tgt = torch.rand(n, 2, generator=gen).to(gpu_device)

    def run(compact):
        ps = [t.clone().requires_grad_(True) for t in (wq, bq, wk, bk)]
        x_all = torch.zeros(L + 1, n, C, device=gpu_device)
        if compact:
            x0 = x_all[0].view(-1)[:4 * n].view(n, 4)
            x0.copy_(feats)
            out, _ = Fn.grand_euler_block(x0, *ps, lp, graph, L, x_all=x_all, out_cols=2, x0_cols=4)
        else:
            x_all[0][:, :4] = feats
            out, _ = Fn.grand_euler_block(x_all[0], *ps, lp, graph, L, x_all=x_all, out_cols=2)
        F.mse_loss(out, tgt).backward()
        torch.cuda.synchronize()
        return out.detach().clone(), [p_.grad.clone() for p_ in ps[:3]]

    out_c, g_c = run(True)
    out_d, g_d = run(False)
    assert torch.equal(out_c, out_d)
    for name, a, b in zip(('d lin_query.weight', 'd lin_query.bias', 'd lin_key.weight'), g_c, g_d):
        assert rel_err(a, b)[0] <= 5e-6, (name, rel_err(a, b))
    out_c2, g_c2 = run(True)                                        # bit-reproducible
    assert torch.equal(out_c, out_c2) and all(torch.equal(a, b) for a, b in zip(g_c, g_c2))


@pytest.mark.gpu
def test_learn_step_gradients_ride_in_the_flat_tensor(gpu_device):
    """`learn_step` (GNN.py:179-180,288-289) with shared convs at hidden 64 - the shape of the metric workload: the step
    gradients come back as one-element slices of the tensor that carries the weight gradients ([dWq|dbq|dWk|dbk|d dt]), so
    FlatAdam adopts the whole range; compact slots stay on (compact layer-0 input, compact top gradient); values vs the oracle."""
    from g_adaptivity_amd import mse_loss
    opt, ds, data, oracle = make_case((16, 16), 3, 64, 4, learn_step=True)
    with torch.no_grad():
        for l, s_ in enumerate(oracle.steps):
            s_.fill_(0.05 + 0.03 * l)                                       # distinct steps per layer
    model = hip_model_like(oracle, ds, opt, gpu_device)
    F.mse_loss(oracle(data), data.x_phys).backward()
    ours = FlatAdam(model.parameters(), lr=1e-3)
    ours.zero_grad()
    dd = data.clone().to(gpu_device)
    mse_loss(model(dd), dd.x_phys).backward()
    for l in range(4):
        assert model.steps[l].grad.shape == model.steps[l].shape
        assert rel_err(model.steps[l].grad, oracle.steps[l].grad)[0] <= 1e-4, (l, model.steps[l].grad, oracle.steps[l].grad)
    for name in ('lin_query.weight', 'lin_query.bias', 'lin_key.weight'):
        want = dict(oracle.conv_layers[0].named_parameters())[name].grad
        got = dict(model.conv_layers[0].named_parameters())[name].grad
        assert rel_err(got, want)[0] <= 1e-4, name
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) == 4 + 4 and len({g.untyped_storage().data_ptr() for g in grads}) == 1
    ours.step()
    assert ours.grad_bucket.untyped_storage().data_ptr() == grads[0].untyped_storage().data_ptr()
    assert ours.grad_bucket.numel() == 2 * (64 * 64 + 64) + 4
    g = next(iter(model._graphs.values()))
    assert model.opt['compact_slots'] and g.num_nodes == 3 * 256
    # the step gradients are summed from per-workgroup partials in a fixed order (no float atomics): bit-reproducible
    first = [model.steps[l].grad.clone() for l in range(4)]
    ours.zero_grad()
    mse_loss(model(dd), dd.x_phys).backward()
    # (the Adam step in between moved the weights: compare against a third run from the same state instead)
    second = [model.steps[l].grad.clone() for l in range(4)]
    ours.zero_grad()
    mse_loss(model(dd), dd.x_phys).backward()
    assert all(torch.equal(a, model.steps[l].grad) for l, a in enumerate(second)) and not all(torch.equal(a, b) for a, b in zip(first, second))


@pytest.mark.gpu
def test_flat_adam_adopts_the_block_gradient_tensor(gpu_device):
    """After backward the four weight gradients are views of one tensor; FlatAdam steps on that memory directly."""
    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, mse_loss
    opt = hot_path_opt(mesh_dims=[9, 9], hidden_dim=16, num_layers=2, device=str(gpu_device), show_mesh_evol_plots='False')
    ds = MeshDataset([9, 9], 4, seed=0)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(gpu_device).train()
    twin = GNN(ds, opt).to(gpu_device).train()
    twin.load_state_dict(model.state_dict())
    ours = FlatAdam(model.parameters(), lr=1e-2)
    ref = torch.optim.Adam([p for p in twin.parameters() if p.requires_grad], lr=1e-2)
    for step in range(3):
        ours.zero_grad(); ref.zero_grad()
        mse_loss(model(data), data.x_phys).backward()
        F.mse_loss(twin(data), data.x_phys).backward()
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        assert len({g.untyped_storage().data_ptr() for g in grads}) == 1
        ours.step(); ref.step()
        assert ours.grad_bucket.untyped_storage().data_ptr() == grads[0].untyped_storage().data_ptr()
    torch.cuda.synchronize()
    for (k, p), q in zip(model.named_parameters(), twin.parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6), k
    # gradients accumulated over two backward calls: no longer one tensor -> gathered copy, same result as torch
    ours.zero_grad(); ref.zero_grad()
    for _ in range(2):
        mse_loss(model(data), data.x_phys).backward()
        F.mse_loss(twin(data), data.x_phys).backward()
    ours.step(); ref.step()
    torch.cuda.synchronize()
    for (k, p), q in zip(model.named_parameters(), twin.parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-6), k


@pytest.mark.gpu
@pytest.mark.parametrize("C", [32, 64])
def test_mixed_window_tiles(gpu_device, C):
    """A mesh batch (every tile's neighbours within +-1 tile: LDS-ring gathers) plus a few long-range edges and one
    hub node: windowed tiles, HBM-gather tiles and loop-path tiles in ONE launch, forward and backward."""
    ds = MeshDataset([16, 16], 4, seed=5)
    from oracle.pyg_restatement import masked_edge_index
    ei = masked_edge_index(collate(ds.samples), 2, 16)
    n = 4 * 256
    g = torch.Generator().manual_seed(11)
    far = torch.stack([torch.randint(0, n, (12,), generator=g), torch.randint(300, 420, (12,), generator=g)])   # into tiles 4..6
    hub = torch.stack([torch.randint(0, n, (20,), generator=g), torch.full((20,), 777)])                        # in-degree > 8
    ei = torch.cat([ei, far, hub], dim=1)
    wq, bq, wk, bk = _random_layer(C, 13)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(14))
    up = torch.randn(n, C, generator=torch.Generator().manual_seed(15))
    x64 = x.double().requires_grad_(True)
    p64 = [w.double().requires_grad_(True) for w in (wq, bq, wk, bk)]
    ref64 = grand_residual(x64, ei, *p64)
    (ref64 * up.double()).sum().backward()

    graph = MeshGraph(ei, n, gpu_device)
    tm = 128 if C == 32 else 64
    meta = graph._metas[('t', tm)].cpu().view(-1, 4)
    assert (meta[:, 3] == 1).any() and (meta[:, 3] == 0).any() and (meta[:, 2] > 8).any()   # all three kinds of tile present
    opt = hot_path_opt(hidden_dim=C, device=str(gpu_device))
    conv = GRAND_plusConv(opt, C, C, global_feat_dim=8, heads=1, concat=False, beta=False, dropout=0.0, edge_dim=None,
                          bias=False, root_weight=False).to(gpu_device)
    with torch.no_grad():
        conv.lin_query.weight.copy_(wq); conv.lin_query.bias.copy_(bq); conv.lin_key.weight.copy_(wk); conv.lin_key.bias.copy_(bk)
    xh = x.to(gpu_device).requires_grad_(True)
    res = conv(xh, ei.to(gpu_device), None, None)
    (res * up.to(gpu_device)).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(res, ref64)[0] <= 1e-5, rel_err(res, ref64)
    assert rel_err(xh.grad, x64.grad)[0] <= 1e-4, rel_err(xh.grad, x64.grad)
    for got, want in ((conv.lin_query.weight.grad, p64[0].grad), (conv.lin_query.bias.grad, p64[1].grad),
                      (conv.lin_key.weight.grad, p64[2].grad)):
        assert rel_err(got, want)[0] <= 1e-4, rel_err(got, want)


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("n,B,C,L,conv,inc_f", FULL_SIZES, ids=FULL_IDS)
def test_full_size_backward_properties(gpu_device, n, B, C, L, conv, inc_f):
    """Backward at the BASELINE sizes: gradients are linear in the upstream gradient,
    invariant to the caller's edge order (within fp32 reassociation) and bit-reproducible."""
    from g_adaptivity_amd import mse_loss
    opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=C, num_layers=L, conv_type=conv, gnn_inc_feat_f=inc_f, device=str(gpu_device),
                       show_mesh_evol_plots='False')
    ds = MeshDataset([n, n], B, seed=0)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(gpu_device).train()
    params = [p for p in model.parameters() if p.requires_grad]

    def grads(scale, d=data):
        for p in params:
            p.grad = None
        (scale * mse_loss(model(d), d.x_phys)).backward()
        return [p.grad.clone() for p in params if p.grad is not None]

    g1, g1b, g3 = grads(1.0), grads(1.0), grads(4.0)         # a power of two: scaling commutes with fp32 rounding
    assert len(g1) == 4 and all(torch.equal(a, b) for a, b in zip(g1, g1b))
    for a, b in zip(g1, g3):
        assert rel_err(4.0 * a, b)[0] <= 1e-6
    d2 = data.clone()
    perm = torch.randperm(d2.edge_index.shape[1], generator=torch.Generator().manual_seed(2)).to(gpu_device)
    d2.edge_index = d2.edge_index[:, perm]
    for k in ('to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask'):
        setattr(d2, k, getattr(d2, k)[perm])
    n_graphs = len(model._graphs)
    gp = grads(1.0, d2)                                          # a permuted edge list is another topology for the cache
    assert len(model._graphs) == n_graphs + 1
    for a, b in zip(g1, gp):
        if a.abs().max() > 0:
            # fp32 reassociation only; over 20 layers it compounds (the fp32 ORACLE is 5e-4 off its fp64 twin at that depth)
            assert rel_err(b, a)[0] <= (1e-4 if L <= 6 else 1e-3), rel_err(b, a)


def _conv_run(gpu_device, ei, x, up, layer, wide):
    """GRAND_plusConv residual forward + backward on dense x with the wide kernels on or off."""
    from g_adaptivity_amd import graph as graph_mod
    C = x.shape[1]
    old = graph_mod.WIDE_KERNELS
    graph_mod.WIDE_KERNELS = wide
    try:
        opt = hot_path_opt(hidden_dim=C, show_mesh_evol_plots=True, device=str(gpu_device))
        conv = GRAND_plusConv(opt, C, C, global_feat_dim=8, heads=1, concat=False, beta=False, dropout=0.0, edge_dim=None,
                              bias=False, root_weight=False).to(gpu_device)
        with torch.no_grad():
            conv.lin_query.weight.copy_(layer[0]); conv.lin_query.bias.copy_(layer[1])
            conv.lin_key.weight.copy_(layer[2]); conv.lin_key.bias.copy_(layer[3])
        xh = x.to(gpu_device).requires_grad_(True)
        res, (_, (alpha, _, _)) = conv(xh, ei.to(gpu_device), None, None, return_attention_weights=True)
        (res * up.to(gpu_device)).sum().backward()
        torch.cuda.synchronize()
        return (res.detach().cpu(), alpha.detach().cpu().view(-1), xh.grad.cpu(), conv.lin_query.weight.grad.cpu(),
                conv.lin_query.bias.grad.cpu(), conv.lin_key.weight.grad.cpu())
    finally:
        graph_mod.WIDE_KERNELS = old


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch", [(64, 32), (40, 45), (23, 7)], ids=['64x64-b32', '40x40-b45', '23x23-b7'])
def test_wide_kernels_match_tiled_kernels(gpu_device, mesh_n, batch):
    """Hidden 64 on row-major mesh batches runs the wide kernels (one wave per 32 nodes, 256-node workgroup steps);
    every other graph runs the tiled ones.  Same inputs through both: several steps per workgroup (64x64 b32: 512 steps
    on 256 workgroups), a ragged last step (40x40 b45: 72 000 nodes), a batch smaller than the grid (23x23 b7)."""
    from oracle.pyg_restatement import masked_edge_index
    from g_adaptivity_amd.graph import MeshGraph as MG
    C = 64
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=3)
    ei = masked_edge_index(collate(ds.samples), 2, mesh_n)
    n = batch * mesh_n * mesh_n
    assert MG(ei, n, gpu_device).wide_deg['t'] > 0                  # the mesh batch qualifies
    layer = _random_layer(C, 11)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(12))
    up = torch.randn(n, C, generator=torch.Generator().manual_seed(13))
    wide = _conv_run(gpu_device, ei, x, up, layer, True)
    tiled = _conv_run(gpu_device, ei, x, up, layer, False)
    names = ('residual', 'alpha', 'dx', 'd lin_query.weight', 'd lin_query.bias', 'd lin_key.weight')
    tols = (2e-6, 2e-6, 1e-5, 1e-5, 1e-5, 1e-5)
    for name, a, b, tol in zip(names, wide, tiled, tols):
        assert rel_err(a, b)[0] <= tol, (name, rel_err(a, b))


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch", [(32, 32), (48, 10), (23, 9), (64, 1)], ids=['32x32-b32', '48x48-b10', '23x23-b9', '64x64-b1'])
def test_wide_forward_four_wave_workgroups_are_the_eight_wave_ones(gpu_device, mesh_n, batch):
    """Batches of at most 32 768 nodes run the wide forward on FOUR-wave workgroups, steps of 128 nodes, a 256-row window
    (`wide::fwd_kernel<..., 4>`, `gadapt_graph::wide_half_deg_t`): twice the workgroups where the 256-node steps would leave half of the
    CUs idle.  A node's arithmetic is the same in both forms, so the full model - compact layer-0 input, dense layers, head-only last
    output, backward through the stored alpha - gives bit-identical results: one step per workgroup on every CU (32x32 b32), mesh rows
    that do not divide the step (48 wide), a ragged last step (23x23 b9), fewer steps than XCDs x 4 (64x64 b1)."""
    from g_adaptivity_amd import graph as graph_mod
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=64, num_layers=3, device=str(gpu_device), show_mesh_evol_plots=True)
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=7)
    data = collate(ds.samples).to(gpu_device)
    keep = graph_mod.WIDE_MIN_NODES, graph_mod.WIDE_HALF_MAX_NODES
    res = {}
    try:
        for half_max in (keep[1], 0):
            graph_mod.WIDE_MIN_NODES, graph_mod.WIDE_HALF_MAX_NODES = 0, half_max
            torch.manual_seed(3)
            model = GNN(ds, opt).to(gpu_device).train()
            g = model._graph(data, data.x_comp.shape[0], gpu_device)
            assert g.wide_deg['t'] > 0 and (g.wide_half_deg > 0) == (half_max > 0)
            out = model(data)
            F.mse_loss(out, data.x_phys).backward()
            torch.cuda.synchronize()
            res[half_max > 0] = [out.detach().clone()] + [p.grad.clone() for p in model.parameters() if p.grad is not None]
    finally:
        graph_mod.WIDE_MIN_NODES, graph_mod.WIDE_HALF_MAX_NODES = keep
    assert len(res[True]) == len(res[False]) >= 4
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch,layers", [(128, 16, 2), (128, 3, 3), (100, 5, 2), (65, 7, 2), (96, 2, 4)],
                         ids=['128x128-b16', '128x128-b3', '100x100-b5', '65x65-b7', '96x96-b2'])
def test_wide_forward_512_row_window_matches_tiled_kernels(gpu_device, mesh_n, batch, layers):
    """Meshes with 65..128 nodes per mesh row (BASELINE config 5: 128x128) miss the wide forward's 384-row window - neighbours
    sit up to 129 rows away - and take its 512-row variant (256-byte rows, XOR-swizzled chunks): full model, forward + backward,
    against the tiled kernels on the same inputs.  Covers several steps per workgroup in both walk directions (b16: 1024 steps
    on 256 workgroups), ragged last steps (100x100, 65x65), the compact layer-0 input and the head-only last output."""
    from g_adaptivity_amd import graph as graph_mod
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=64, num_layers=layers, device=str(gpu_device), show_mesh_evol_plots=True)
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=5)
    data = collate(ds.samples).to(gpu_device)
    res = {}
    for wide in (True, False):
        graph_mod.WIDE_KERNELS = wide
        try:
            torch.manual_seed(3)
            model = GNN(ds, opt).to(gpu_device).train()
            out = model(data)
            F.mse_loss(out, data.x_phys).backward()
            torch.cuda.synchronize()
            g = next(iter(model._graphs.values()))
            assert g.wide_deg['t'] == 0 and (g.wide_big_deg > 0) == wide
            lay = model.conv_layers[0]
            res[wide] = (out.detach().clone(), lay.stored_alpha.detach().clone(), lay.lin_query.weight.grad.clone(), lay.lin_key.weight.grad.clone())
        finally:
            graph_mod.WIDE_KERNELS = True
    for name, a, b, tol in zip(('x_phys', 'alpha (last layer)', 'd lin_query.weight', 'd lin_key.weight'), res[True], res[False], (2e-6, 1e-5, 1e-4, 1e-4)):   # alpha: after 1..3 layers of differently rounded inputs (measured 3.8e-6 .. 4.5e-6); weight gradients: cancelling sums, the 8e-5 of test_wide_kernels_size_sweep (measured 4.7e-5 .. 8.3e-5)
        assert rel_err(a, b)[0] <= tol, (name, rel_err(a, b))


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch", [(5, 1), (9, 3), (16, 1), (17, 2), (31, 1), (33, 9), (50, 11), (63, 17), (64, 1), (64, 5)],
                         ids=lambda v: str(v))
def test_wide_kernels_size_sweep(gpu_device, mesh_n, batch):
    """Node counts around the wave (32) and step (256) granularity of the wide kernels: 25 nodes, one node past a step,
    a single partial step, batches that end mid-wave.  Full model, forward + backward, against the tiled kernels."""
    from g_adaptivity_amd import graph as graph_mod
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=64, num_layers=3, device=str(gpu_device))
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=7)
    data = collate(ds.samples).to(gpu_device)
    res = {}
    for wide in (True, False):
        graph_mod.WIDE_KERNELS = wide
        try:
            torch.manual_seed(3)
            model = GNN(ds, opt).to(gpu_device).train()
            out = model(data)
            F.mse_loss(out, data.x_phys).backward()
            torch.cuda.synchronize()
            g = next(iter(model._graphs.values()))
            assert (g.wide_deg['t'] > 0) == wide
            res[wide] = (out.detach().clone(), model.conv_layers[0].lin_query.weight.grad.clone(), model.conv_layers[0].lin_key.weight.grad.clone())
        finally:
            graph_mod.WIDE_KERNELS = True
    # two fp32 summation orders of the same gradient (measured up to 4.05e-5 on the 64x64 single-mesh case)
    for a, b, tol in zip(res[True], res[False], (2e-6, 8e-5, 8e-5)):
        assert rel_err(a, b)[0] <= tol, rel_err(a, b)


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch", [(9, 3), (17, 2), (33, 9), (63, 5), (64, 3)], ids=lambda v: str(v))
def test_source_window_size_sweep(gpu_device, mesh_n, batch):
    """Hidden 128: the source pass with the LDS window of x rows (mesh-ordered graphs, `wide_deg['s'] > 0`) against the
    plain one (same graph declared not mesh-ordered): tiles that end mid-slab, chunks of one tile, a window clamped at
    both ends.  Full model, forward + backward."""
    from g_adaptivity_amd import graph as graph_mod
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=128, num_layers=3, conv_type='GRAND', device=str(gpu_device))
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=11)
    data = collate(ds.samples).to(gpu_device)
    res = {}
    for wide in (True, False):
        graph_mod.WIDE_KERNELS = wide
        try:
            torch.manual_seed(5)
            model = GNN(ds, opt).to(gpu_device).train()
            out = model(data)
            F.mse_loss(out, data.x_phys).backward()
            torch.cuda.synchronize()
            g = next(iter(model._graphs.values()))
            assert (g.wide_deg['s'] > 0) == wide
            conv0 = model.conv_layers[0]                                # its gradients pass through the source passes of the layers above
            res[wide] = (out.detach().clone(), conv0.lin_query.weight.grad.clone(), conv0.lin_key.weight.grad.clone(), conv0.lin_query.bias.grad.clone())
        finally:
            graph_mod.WIDE_KERNELS = True
    assert torch.equal(res[True][0], res[False][0])                    # the forward does not depend on it
    for a, b in zip(res[True][1:], res[False][1:]):                    # same sums, same order: the window only changes where rows are read
        assert rel_err(a, b)[0] <= 1e-6, rel_err(a, b)


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_wide_kernels_ragged_rows(gpu_device):
    """Rows of 0, 1, 7 and 8 in-edges inside the window (the ELL-8 limit), next to the mesh's 2..6: a node without
    in-edges gets res = -x (empty aggregation, GRAND_plus.py:338-343), unused ELL slots carry weight 0."""
    from oracle.pyg_restatement import masked_edge_index
    from g_adaptivity_amd.graph import MeshGraph as MG
    C, mesh_n, batch = 64, 20, 4
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=6)
    ei = masked_edge_index(collate(ds.samples), 2, mesh_n)
    n = batch * mesh_n * mesh_n
    keep = ~((ei[1] == 37) | (ei[1] == 611))                          # two nodes lose every in-edge
    keep &= ~((ei[1] == 90) & (ei[0] != 89))                          # one keeps a single in-edge
    ei = ei[:, keep]
    extra = torch.tensor([[204, 208, 300, 302, 299, 104], [206, 206, 301, 301, 301, 106]])   # interior rows of 6 -> 8, 8+, 7 entries
    ei = torch.cat([ei, extra, torch.tensor([[301], [301]])], dim=1)  # plus a self-loop: node 301 has 8+... in-edges
    ei = ei[:, ~((ei[1] == 301) & (torch.cumsum((ei[1] == 301).long(), 0) > 8))]    # cap node 301 at 8
    deg = torch.bincount(ei[1], minlength=n)
    assert deg.max().item() == 8 and deg.min().item() == 0 and (deg == 1).any() and (deg == 7).any()
    assert MG(ei, n, gpu_device).wide_deg['t'] == 8
    layer = _random_layer(C, 31)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(32))
    up = torch.randn(n, C, generator=torch.Generator().manual_seed(33))
    wide = _conv_run(gpu_device, ei, x, up, layer, True)
    tiled = _conv_run(gpu_device, ei, x, up, layer, False)
    ref = grand_residual(x.double(), ei, *[w.double() for w in layer])
    assert rel_err(wide[0], ref)[0] <= 1e-5
    assert torch.allclose(wide[0][37], -x[37], rtol=0, atol=1e-6)
    for name, a, b, tol in zip(('residual', 'alpha', 'dx'), wide, tiled, (2e-6, 2e-6, 1e-5)):
        assert rel_err(a, b)[0] <= tol, (name, rel_err(a, b))


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_wide_kernels_rebase_large_scores(gpu_device):
    """The wide forward takes softmax weights relative to the first score of a row and re-bases when a later score
    exceeds it by more than 16: scores spread over +-60 must still give the max-shifted softmax of the reference."""
    from oracle.pyg_restatement import masked_edge_index
    C, mesh_n, batch = 64, 16, 3
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=5)
    ei = masked_edge_index(collate(ds.samples), 2, mesh_n)
    n = batch * mesh_n * mesh_n
    wq, bq, wk, bk = _random_layer(C, 21)
    wq, wk = wq * 6.0, wk * 6.0                                     # scores of order +-60
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(22))
    up = torch.randn(n, C, generator=torch.Generator().manual_seed(23))
    ref, (alpha_ref, _, _) = grand_residual(x.double(), ei, wq.double(), bq.double(), wk.double(), bk.double(), return_attention=True)
    res, alpha, *_ = _conv_run(gpu_device, ei, x, up, (wq, bq, wk, bk), True)
    sc = (alpha_ref.view(-1) > 0.5).float().mean().item()
    assert sc > 0.12                                                # the case is what it claims: most rows (~6 edges) are one-hot
    assert rel_err(alpha, alpha_ref.view(-1))[0] <= 2e-5, rel_err(alpha, alpha_ref.view(-1))
    assert rel_err(res, ref)[0] <= 2e-5, rel_err(res, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,mesh_n,batch,layers,learn", [(64, 64, 3, 4, False), (64, 33, 5, 3, True), (32, 20, 4, 3, False), (128, 16, 2, 3, False), (8, 15, 3, 4, False)],
                         ids=['C64-64x64', 'C64-33x33-learn-step', 'C32', 'C128', 'C8'])
def test_block_backward_inplace_is_bit_identical(gpu_device, hidden, mesh_n, batch, layers, learn, monkeypatch):
    """gadapt_block_backward with the source pass writing g_out over the dxd rows it has read (the default) against separate
    buffers: the same launches on the same values, so every gradient is bit-identical."""
    import g_adaptivity_amd.functional as Fn_mod
    from g_adaptivity_amd._native import lib
    monkeypatch.setattr(Fn_mod, 'SMALL_MESH_FORWARD', False)
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=hidden, num_layers=layers, device=str(gpu_device), show_mesh_evol_plots='False')
    if learn:
        opt['learn_step'] = True
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=4)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(6)
    model = GNN(ds, opt).to(gpu_device).train()
    res = {}
    for on in (0, 1):
        lib().gadapt_debug_set_backward_inplace(on)
        try:
            model.zero_grad()
            F.mse_loss(model(data), data.x_phys).backward()
            torch.cuda.synchronize()
            res[on] = [p.grad.clone() for p in model.parameters() if p.grad is not None]
        finally:
            lib().gadapt_debug_set_backward_inplace(1)
    assert len(res[0]) >= 4
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)