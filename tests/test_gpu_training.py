"""End-to-end: a reference-style training loop on the HIP model tracks the CPU oracle trained with torch.optim.Adam."""
import copy

import pytest
import torch
import torch.nn.functional as F

from g_adaptivity_amd import GNN, MeshDataset, MeshLoader, collate, hot_path_opt
from g_adaptivity_amd.optim import FlatAdam
from helpers import rel_err
from oracle.pyg_restatement import OracleGNN


@pytest.mark.gpu
def test_training_steps_track_the_oracle(gpu_device):
    opt = hot_path_opt(mesh_dims=[12, 12], hidden_dim=64, num_layers=3, lr=1e-3, decay=1e-4, batch_size=4)
    ds = MeshDataset([12, 12], 16, seed=0)
    torch.manual_seed(0)
    oracle = OracleGNN(ds, dict(opt)).train()
    o = dict(opt); o['device'] = str(gpu_device)
    model = GNN(ds, o).to(gpu_device).train()
    model.load_state_dict(copy.deepcopy(oracle.state_dict()))
    opt_ref = torch.optim.Adam(oracle.parameters(), lr=opt['lr'], weight_decay=opt['decay'])
    opt_hip = FlatAdam(model.parameters(), lr=opt['lr'], weight_decay=opt['decay'])
    losses_ref, losses_hip = [], []
    for epoch in range(2):
        for data in MeshLoader(ds, batch_size=4, shuffle=False):
            opt_ref.zero_grad()
            l = F.mse_loss(oracle(data), data.x_phys); l.backward(); opt_ref.step(); losses_ref.append(l.item())
            opt_hip.zero_grad()
            d = data.clone().to(gpu_device)
            l2 = F.mse_loss(model(d), d.x_phys); l2.backward(); opt_hip.step(); losses_hip.append(l2.item())
    assert losses_ref[-1] < losses_ref[0]                              # it trains
    for a, b in zip(losses_hip, losses_ref):
        assert abs(a - b) <= 1e-4 * abs(b)
    lo, lh = oracle.conv_layers[0], model.conv_layers[0]
    # Adam turns gradient rounding into O(lr) parameter differences only where a gradient is ~0; compare the update
    for name in ('lin_query.weight', 'lin_key.weight', 'lin_query.bias'):
        w_ref, w_hip = dict(lo.named_parameters())[name], dict(lh.named_parameters())[name]
        assert (w_hip.detach().cpu() - w_ref.detach()).abs().max().item() <= 2e-3 * 8 * opt['lr'] + 1e-6
    assert torch.equal(lh.lin_skip.weight.cpu(), lo.lin_skip.weight)   # never receives a gradient: untouched (also by weight decay)
    # eval-mode forward: same result, end_MLmodel stamped after a stream sync
    model.eval()
    data = collate(ds.samples[:2])
    with torch.no_grad():
        out = model(data.clone().to(gpu_device))
    assert model.end_MLmodel is not None and rel_err(out, oracle.eval()(data))[0] <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,direct", [(8, True), (64, False)], ids=['C8-one-launch-direct', 'C64-hipgraph'])
def test_rollout_reuses_the_graph_and_hipgraph_replay_equals_eager(gpu_device, hidden, direct):
    """Burgers-style rollout (utils_eval_Burgers.py:282-300): same mesh, new uu_tensor every call.  The CSR is built once,
    and the captured forward gives bit-identical coordinates.  At the reference's own size (21 nodes, hidden 8) the forward is the
    one-launch small-mesh kernel and `GraphedForward` launches it directly on the caller's tensors (no capture, no copies); at
    hidden 64 it replays the captured per-layer launches."""
    from g_adaptivity_amd.inference import GraphedForward
    opt = hot_path_opt(mesh_dims=[21], hidden_dim=hidden, num_layers=3, device=str(gpu_device), conv_type='GRAND',
                       gnn_inc_feat_f=False, show_mesh_evol_plots='False')
    ds = MeshDataset([21], 1, seed=0)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(gpu_device).eval()
    runner = GraphedForward(model, data)
    assert runner.direct is direct
    outs_e, outs_g = [], []
    with torch.no_grad():
        for k in range(6):
            data.uu_tensor = torch.sin((k + 1) * data.x_comp.reshape(-1)).to(data.uu_tensor.dtype)
            outs_e.append(model(data).clone())
            outs_g.append(runner(data).clone())
    assert len(model._graphs) == 1
    assert all(torch.equal(a, b) for a, b in zip(outs_e, outs_g))
    assert not torch.equal(outs_e[0], outs_e[-1])
    assert model.end_MLmodel is not None
    # unchanged fields are not copied again (same tensor object, same version); an in-place update must be seen
    with torch.no_grad():
        before = runner(data).clone()
        data.uu_tensor.mul_(0.5)                                    # in place: same object, new version
        after_g, after_e = runner(data).clone(), model(data).clone()
        data.x_comp = data.x_comp.clone()                           # new object, same values
        again = runner(data).clone()
    assert torch.equal(after_g, after_e) and not torch.equal(before, after_g) and torch.equal(again, after_g)


@pytest.mark.gpu
@pytest.mark.parametrize("share", [True, False], ids=['shared-conv-direct-launch', 'per-layer-convs-captured-graph'])
def test_graphed_forward_follows_weight_changes(gpu_device, share):
    """`GraphedForward` on a small-mesh batch must never serve stale weights.  A shared conv is launched directly
    on the parameters' own storage - in-place optimizer updates are seen, and parameter storage that MOVES (FlatAdam lays its bucket
    out on its first step; `p.data = ...`) is noticed per call.  Per-layer convs are stacked copies in the one-launch plan, so they
    take the captured graph, which re-stacks from the live parameters on every replay."""
    from g_adaptivity_amd.inference import GraphedForward
    opt = hot_path_opt(mesh_dims=[11, 11], hidden_dim=8, num_layers=3, share_conv=share, device=str(gpu_device))
    ds = MeshDataset([11, 11], 2, seed=2)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(3)
    model = GNN(ds, opt).to(gpu_device).eval()
    runner = GraphedForward(model, data)
    assert runner.direct == share
    with torch.no_grad():
        assert torch.equal(runner(data).clone(), model(data))
        for p in model.conv_layers.parameters():                        # in place, raw (no version bump: what the Adam kernels do)
            p.data.mul_(1.25)
        assert torch.equal(runner(data).clone(), model(data))
        for p in model.conv_layers.parameters():                        # storage re-bound (what FlatAdam._build does)
            p.data = (p.data * 0.5).clone()
        assert torch.equal(runner(data).clone(), model(data))


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch,hidden,conv,f", [(16, 4, 64, 'GRAND_plus', True), (20, 3, 32, 'GRAND_plus', False), (64, 8, 64, 'GRAND_plus', True)],
                         ids=['16x16-C64', '20x20-C32-xyu', '64x64-b8-C64'])
def test_graphed_forward_issued_as_one_call(gpu_device, mesh_n, batch, hidden, conv, f):
    """Weight-shared blocks behind the identity encoder: `GraphedForward` issues the whole evaluation forward as ONE C-ABI call
    (`gadapt_block_forward_loss` without a target; layer 0 reads the caller's field tensors in place) instead of replaying a capture.
    Same kernels on the same values as `model(data)`: bit-identical output - for the batch it was built on, for new field tensors
    (a rollout: `src/utils_eval_Burgers.py:282-300`), after in-place weight updates, after parameter storage moved into FlatAdam's
    bucket (from where the wide layer-0 launch computes the coefficients itself), and with the tiled forward (`GADAPT_WIDE` off)."""
    from g_adaptivity_amd.inference import GraphedForward
    from g_adaptivity_amd import mse_loss
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=hidden, num_layers=3, conv_type=conv, gnn_inc_feat_f=f, device=str(gpu_device),
                       show_mesh_evol_plots='False')
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=2)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(3)
    model = GNN(ds, opt).to(gpu_device).eval()
    runner = GraphedForward(model, data)
    assert runner.issued and not runner.direct
    with torch.no_grad():
        assert torch.equal(runner(data).clone(), model(data))
        d2 = data.clone()
        d2.uu_tensor = data.uu_tensor * 0.5 + 0.1
        d2.x_comp = (data.x_comp + 0.01 * torch.sin(7 * data.x_comp)).contiguous()
        assert torch.equal(runner(d2).clone(), model(d2))
        assert torch.equal(runner(uu_tensor=data.uu_tensor, x_comp=data.x_comp).clone(), model(data))     # fields as keywords
        for p in model.conv_layers.parameters():                        # in place, raw (what the Adam kernels do)
            p.data.mul_(1.25)
        assert torch.equal(runner(data).clone(), model(data))
    # parameter storage moves into FlatAdam's bucket: [Wq | bq | Wk | bk] back to back
    model.train()
    optim = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
    optim.zero_grad(); mse_loss(model(data), data.x_phys).backward(); optim.step()
    model.eval()
    with torch.no_grad():
        assert torch.equal(runner(data).clone(), model(data))
        assert runner._flat_bucket() is not None


@pytest.mark.gpu
def test_compact_encoder_output_equals_dense(gpu_device):
    """Identity encoder = zero-pad (GNN.py:75-82): layer 0 reads the compact [N,4] features and the top layer's backward
    the compact [N,dim] gradient.  The forward gives exactly what the dense [N,C] matrices give (same kernels, same order).
    The weight gradients agree to rounding: layer 0's share is summed by the compact-input target kernel (one node per lane
    over the 4 live columns) instead of the tiled kernel's per-tile matrix products - the same terms in another order."""
    import copy
    import torch.nn.functional as F
    from helpers import rel_err
    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt
    for mesh, hidden in (([24, 24], 64), ([13, 13], 16)):          # wide kernels / tiled kernels
        opt = hot_path_opt(mesh_dims=mesh, hidden_dim=hidden, num_layers=3, device=str(gpu_device))
        ds = MeshDataset(mesh, 5, seed=4)
        data = collate(ds.samples).to(gpu_device)
        torch.manual_seed(1)
        model = GNN(ds, opt).to(gpu_device).train()
        assert model._enc_is_zero_pad()
        out_c = model(data)
        F.mse_loss(out_c, data.x_phys).backward()
        grads_c = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        model.zero_grad()
        out_c2 = model(data)                                        # the compact path itself is bit-reproducible
        F.mse_loss(out_c2, data.x_phys).backward()
        assert torch.equal(out_c, out_c2) and all(torch.equal(p.grad, grads_c[k]) for k, p in model.named_parameters() if p.grad is not None)
        model.zero_grad()
        model._enc_is_zero_pad = lambda: False                      # dense x0 [N,C] through encode + layer 0
        out_d = model(data)
        F.mse_loss(out_d, data.x_phys).backward()
        torch.cuda.synchronize()
        assert torch.equal(out_c, out_d)
        for k, p in model.named_parameters():
            if p.grad is not None:
                assert rel_err(p.grad, grads_c[k])[0] <= 2e-6, (k, rel_err(p.grad, grads_c[k]))
        # ... and the fully dense data flow (opt['compact_slots'] = False: padded x0, full x_L, padded top gradient)
        torch.manual_seed(1)
        dense = GNN(ds, dict(opt, compact_slots=False)).to(gpu_device).train()
        dense.load_state_dict(model.state_dict())
        out_f = dense(data)
        F.mse_loss(out_f, data.x_phys).backward()
        torch.cuda.synchronize()
        assert torch.equal(out_c, out_f)
        for k, p in dense.named_parameters():
            if p.grad is not None:
                assert rel_err(p.grad, grads_c[k])[0] <= 2e-6, (k, rel_err(p.grad, grads_c[k]))


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_bench_line_contract(gpu_device):
    """bench.py prints ONE JSON line with the driver's keys, the roofline object of the dominant kernel group and a
    bounded CPU baseline (run as the driver runs it: a child process, small K / W)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--cpu-seconds', '2'],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 2 and d['unit'] == 'meshes/s' and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic' and 'workload' in d['config']
    rf, cb = d['roofline'], d['cpu_baseline']
    assert rf['bound'] == 'hbm' and rf['peak'] == 8000.0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 and rf['unit'] == 'GB/s'
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and 'sample' in cb
    assert abs(d['value'] - 32 * 1000.0 / d['ms_per_step']) / d['value'] < 0.01
    # median of >= 5 windows of K steps, and the dense-slot flow timed in the same run
    wd = d['windows']
    # a window is the K-step block repeated until it lasts >= 50 ms
    assert wd['n'] >= 5 and wd['steps_per_window'] == 4 * wd['blocks_per_window'] and wd['ms_per_step_min'] <= wd['ms_per_step_median'] <= wd['ms_per_step_max']
    assert wd['window_ms_median'] >= 0.9 * wd['min_window_ms'] >= 45.0
    assert abs(wd['ms_per_step_median'] - d['ms_per_step']) < 1e-3
    assert d['config']['slots'] == 'compact' and d['value_dense_slots'] > 0 and d['ms_per_step_dense_slots'] >= wd['ms_per_step_min'] * 0.9
    assert d['windows_dense_slots']['n'] >= 5
    # ... and the reference's training loop on shuffled, changing batches (GraphedTrainStep): same order of speed as the static-batch
    # headline (with real step counts it is within 5 %: profiles/r04_bench.json; this run times 24 steps, so the bar is loose)
    tl = d['train_loop']
    if not tl['value'] > 0.6 * d['value']:                          # keep the evidence of a slow run
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        open(os.path.join(root, 'gpurun_out', 'bench_line_contract_slow.json'), 'w').write(lines[0] + '\n' + r.stderr[-4000:])
    assert tl['steps'] >= 16 and tl['value'] > 0.6 * d['value'], (tl, d['value'])
    # the other single-GPU BASELINE.json configurations ride on the same line, each with its own roofline object
    ow = d['other_workloads']
    assert set(ow) == {'poisson2d_32x32_b32_L4_C64', 'burgers2d_64x64_b32_L6_C128', 'euler20_128x128_b16_C64'}
    for name, o in ow.items():
        assert o['value'] > 0 and o['windows']['n'] == 2 and o['windows']['window_ms_median'] >= 45.0, (name, o)
        assert abs(o['value'] - o['config']['meshes_per_gpu'] * 1000.0 / o['ms_per_step']) / o['value'] < 0.01
        orf = o['roofline']
        for k in ('kernel', 'variant', 'frac', 'avg_launch_us', 'alg_bytes_per_launch', 'profile'):
            assert k in orf, (name, k)
        assert abs(orf['frac'] - orf['alg_bytes_per_launch'] / (orf['avg_launch_us'] * 1e-6) / 8e12) < 2e-3, (name, orf)


def _fresh_trainer(ds, opt, gpu_device, state, fused=True):
    from g_adaptivity_amd import GraphedTrainStep
    model = GNN(ds, opt).to(gpu_device).train()
    model.load_state_dict(copy.deepcopy(state))
    optim = FlatAdam(model.parameters(), lr=opt['lr'], weight_decay=opt['decay'], capturable=True)
    return model, optim, GraphedTrainStep(model, optim, fused=fused)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=['fused-13-launches', 'autograd-iteration'])
@pytest.mark.parametrize("into", [False, True], ids=['copy-in', 'loader-writes-static-buffers'])
def test_graphed_train_step_is_the_eager_loop(gpu_device, into, fused):
    """`GraphedTrainStep`: the reference's iteration (`src/run_GNN.py:99-131`) captured once per batch
    topology and replayed on CHANGING batches.  Two epochs over a shuffled `DeviceMeshLoader` whose last batch is short (a
    second topology -> a second capture): the replayed steps leave bit-identical parameters, moments and losses to the same
    steps issued eagerly, the capture's warm-up iterations leave no trace, and the whole run tracks the CPU oracle trained with
    torch.optim.Adam on the same batches.  fused: the captures are the 13-launch fused iteration
    (`training.FusedIteration`: node fields read by layer 0, loss in the last layer's launch, Adam + next-step coefficients in the
    tail) against the EAGER AUTOGRAD loop - parameters and moments still bit-identical, the loss value (another fixed summation
    order) within 1e-6 relative."""
    from g_adaptivity_amd import DeviceMeshLoader
    opt = hot_path_opt(mesh_dims=[12, 12], hidden_dim=64, num_layers=3, lr=1e-3, decay=1e-4, batch_size=4, device=str(gpu_device))
    ds = MeshDataset([12, 12], 14, seed=0)                              # 14 = 3 batches of 4 + one of 2
    torch.manual_seed(0)
    copt = dict(opt); copt['device'] = 'cpu'
    oracle = OracleGNN(ds, copt).train()
    state = copy.deepcopy(oracle.state_dict())
    opt_ref = torch.optim.Adam(oracle.parameters(), lr=opt['lr'], weight_decay=opt['decay'])
    m_e, o_e, step_e = _fresh_trainer(ds, opt, gpu_device, state)
    m_g, o_g, step_g = _fresh_trainer(ds, opt, gpu_device, state, fused=fused)
    fields = ('x_comp', 'x_phys', 'f_tensor', 'uu_tensor')

    def loader(step=None):
        gen = torch.Generator(device=gpu_device); gen.manual_seed(7)
        return DeviceMeshLoader(ds, batch_size=4, shuffle=True, device=gpu_device, generator=gen, fields=fields,
                                into=step.static_batch if step is not None else None)

    le, lg = loader(), loader(step_g if into else None)
    losses_e, losses_g, losses_ref = [], [], []
    for epoch in range(2):
        for d_e, d_g in zip(le, lg):
            assert torch.equal(d_e.idx, d_g.idx)
            losses_e.append(step_e.eager(d_e).clone())
            losses_g.append(step_g(d_g).clone())
            # the oracle on the same meshes (host collation)
            batch = collate([ds.samples[i] for i in d_e.idx.tolist()])
            opt_ref.zero_grad()
            l = F.mse_loss(oracle(batch), batch.x_phys); l.backward(); opt_ref.step(); losses_ref.append(l.item())
    torch.cuda.synchronize()
    assert len(step_g._captured) == 2                                   # batch of 4, batch of 2
    assert all((c.fused is not None) == fused for c in step_g._captured.values()), step_g.fused_reason
    for a, b in zip(losses_e, losses_g):
        if fused:
            assert abs(a.item() - b.item()) <= 1e-6 * abs(a.item()), (a.item(), b.item())
        else:
            assert torch.equal(a, b)
    for (n1, p1), (n2, p2) in zip(m_e.named_parameters(), m_g.named_parameters()):
        assert torch.equal(p1, p2), n1
    assert torch.equal(o_e.exp_avg, o_g.exp_avg) and torch.equal(o_e.exp_avg_sq, o_g.exp_avg_sq)
    assert o_g.state_dict()['step'] == o_e.state_dict()['step'] == len(losses_e) == 8
    for a, b in zip(losses_g, losses_ref):
        assert abs(a.item() - b) <= 1e-4 * abs(b)
    assert losses_ref[-1] < losses_ref[0]
    lo, lh = oracle.conv_layers[0], m_g.conv_layers[0]
    for name in ('lin_query.weight', 'lin_key.weight', 'lin_query.bias'):
        w_ref, w_hip = dict(lo.named_parameters())[name], dict(lh.named_parameters())[name]
        assert (w_hip.detach().cpu() - w_ref.detach()).abs().max().item() <= 2e-3 * 8 * opt['lr'] + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,batch,hidden,layers,conv,f,l1", [
    (20, 5, 64, 4, 'GRAND_plus', True, False), (64, 8, 64, 4, 'GRAND_plus', True, False), (16, 3, 128, 3, 'GRAND', False, True),
    (33, 5, 32, 2, 'GRAND_plus', True, False), (17, 4, 8, 5, 'GRAND_plus', True, True), (40, 2, 16, 3, 'GRAND', False, False)],
    ids=['20x20-C64', '64x64-b8-C64', '16x16-C128-GRAND-xyu-l1', '33x33-C32-L2', '17x17-C8-L5-l1', '40x40-C16-GRAND-xyu'])
def test_fused_iteration_is_the_autograd_iteration(gpu_device, mesh_n, batch, hidden, layers, conv, f, l1, monkeypatch):
    """`training.FusedIteration` issued EAGERLY (no capture) against the autograd iteration on the same batch: after three steps the
    parameters, both Adam moments and the device step count are bit-identical, so are the model output and the parameter gradients of
    every step; the loss value within 1e-6 relative; and the composite coefficients the next step's forward works with - left by the tail,
    or computed by the wide layer-0 launch itself (`coeffs_in_forward`) - are bit for bit the ones `gadapt_coeffs_forward` computes from the
    updated weights.  Tiled and wide forward, hidden 8 ... 128, 3 and 4 feature columns,
    MSE and L1 (`src/run_GNN.py:80-84`)."""
    import g_adaptivity_amd.functional as Fn_mod
    from g_adaptivity_amd import l1_loss, mse_loss, unit_gradient
    from g_adaptivity_amd.training import FusedIteration
    monkeypatch.setattr(Fn_mod, 'SMALL_MESH_FORWARD', False)         # (hidden 8 / 16: the per-layer kernels on both sides)
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=hidden, num_layers=layers, conv_type=conv, gnn_inc_feat_f=f, lr=1e-3, decay=1e-4,
                       device=str(gpu_device), show_mesh_evol_plots='False')
    ds = MeshDataset([mesh_n, mesh_n], batch, seed=3)
    data = collate(ds.samples).to(gpu_device)
    loss_fn = l1_loss if l1 else mse_loss
    torch.manual_seed(2)
    base = GNN(ds, opt).to(gpu_device).train()
    state = copy.deepcopy(base.state_dict())
    runs = {}
    for route in ('autograd', 'fused'):
        model = GNN(ds, opt).to(gpu_device).train(); model.load_state_dict(copy.deepcopy(state))
        optim = FlatAdam(model.parameters(), lr=opt['lr'], weight_decay=opt['decay'], capturable=True)

        def autograd_step():
            optim.zero_grad()
            out = model(data)
            loss = loss_fn(out, data.x_phys)
            loss.backward(gradient=unit_gradient(gpu_device))
            optim.step()
            grads = [p.grad.clone() for p in optim.active]               # [dWq, dbq, dWk, dbk]: the bucket's order
            return out.detach().clone(), loss.detach().clone(), grads

        rec = [autograd_step()]                                         # lays the optimizer's bucket out (both routes)
        if route == 'fused':
            assert FusedIteration.eligible(model, optim, loss_fn, data, 'x_phys') is None
            it = FusedIteration(model, optim, loss_fn, data, 'x_phys')
            it.refresh_coeffs()
        for _ in range(3):
            if route == 'fused':
                it.forward_backward()
                grads = None
                it.finish()
                rec.append((it.out.clone(), it.loss.clone(), [g.clone() for _, g in it.grads]))
            else:
                rec.append(autograd_step())
        if route == 'fused' and it.coeffs_in_forward:                  # (A, p0) come from the NEXT forward's layer-0 launch: issue one
            it.forward_backward()
        torch.cuda.synchronize()
        runs[route] = (rec, [p.detach().clone() for p in optim.active], optim.exp_avg.clone(), optim.exp_avg_sq.clone(), optim.state_dict()['step'],
                       it.coeffs if route == 'fused' else Fn_mod.composite_coeffs(*[p.detach() for p in optim.active[:3]]))
    (ra, pa, ma, va, sa, ca), (rf, pf, mf, vf, sf, cf) = runs['autograd'], runs['fused']
    assert sa == sf == 4
    for k, ((oa, la, ga), (of, lf, gf)) in enumerate(zip(ra, rf)):
        assert torch.equal(oa, of), k
        assert abs(la.item() - lf.item()) <= 1e-6 * abs(la.item()), (k, la.item(), lf.item())
        for x, y in zip(ga, gf):
            assert torch.equal(x.reshape(-1), y.reshape(-1)), k
    for x, y in zip(pa, pf):
        assert torch.equal(x, y)
    assert torch.equal(ma, mf) and torch.equal(va, vf)
    assert torch.equal(ca[0].reshape(-1), cf[0].reshape(-1)) and torch.equal(ca[1].reshape(-1), cf[1].reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("dims,batch,hidden,layers,conv,f,l1", [
    ([11, 11], 8, 8, 4, 'GRAND_plus', True, False), ([13, 13], 5, 16, 3, 'GRAND', False, True), ([23, 23], 6, 8, 4, 'GRAND_plus', True, False),
    ([15, 15], 4, 8, 2, 'GRAND_plus', True, False), ([9, 9], 70, 4, 3, 'GRAND_plus', False, True)],
    ids=['11x11-C8', '13x13-C16-GRAND-xyu-l1', '23x23-C8', '15x15-C8-L2', '9x9-b70-C4-l1'])
def test_fused_iteration_on_small_meshes_is_the_autograd_iteration(gpu_device, dims, batch, hidden, layers, conv, f, l1):
    """Small-mesh batches (the reference's own sizes: `src/params.py:37,130-134`): the fused iteration is FOUR launches - the one-launch
    forward with the loss inside (`gadapt_small_forward_loss`), the one-launch backward, slab sums, chain rule + Adam - against the
    autograd iteration over the same one-launch pair (seven): outputs, gradients, parameters and both Adam moments bit-identical after
    every one of three steps, the loss value within 1e-6 relative (another fixed summation order)."""
    from g_adaptivity_amd import l1_loss, mse_loss, unit_gradient
    from g_adaptivity_amd.training import FusedIteration
    opt = hot_path_opt(mesh_dims=dims, hidden_dim=hidden, num_layers=layers, conv_type=conv, gnn_inc_feat_f=f, lr=1e-3, decay=1e-4,
                       device=str(gpu_device), show_mesh_evol_plots='False')
    ds = MeshDataset(dims, batch, seed=3)
    data = collate(ds.samples).to(gpu_device)
    loss_fn = l1_loss if l1 else mse_loss
    torch.manual_seed(2)
    state = copy.deepcopy(GNN(ds, opt).to(gpu_device).state_dict())
    runs = {}
    for route in ('autograd', 'fused'):
        model = GNN(ds, opt).to(gpu_device).train(); model.load_state_dict(copy.deepcopy(state))
        optim = FlatAdam(model.parameters(), lr=opt['lr'], weight_decay=opt['decay'], capturable=True)

        def autograd_step():
            optim.zero_grad()
            out = model(data)
            loss = loss_fn(out, data.x_phys)
            loss.backward(gradient=unit_gradient(gpu_device))
            optim.step()
            return out.detach().clone(), loss.detach().clone(), [p.grad.clone() for p in optim.active]

        rec = [autograd_step()]                                         # lays the optimizer's bucket out (both routes)
        if route == 'fused':
            assert FusedIteration.eligible(model, optim, loss_fn, data, 'x_phys') is None
            it = FusedIteration(model, optim, loss_fn, data, 'x_phys')
            assert it.small is not None                                 # the one-launch pair, not the per-layer kernels
        for _ in range(3):
            if route == 'fused':
                it.run()
                rec.append((it.out.clone(), it.loss.clone(), [g.clone() for _, g in it.grads]))
            else:
                rec.append(autograd_step())
        torch.cuda.synchronize()
        runs[route] = (rec, [p.detach().clone() for p in optim.active], optim.exp_avg.clone(), optim.exp_avg_sq.clone(), optim.state_dict()['step'])
    (ra, pa, ma, va, sa), (rf, pf, mf, vf, sf) = runs['autograd'], runs['fused']
    assert sa == sf == 4
    for k, ((oa, la, ga), (of, lf, gf)) in enumerate(zip(ra, rf)):
        assert torch.equal(oa, of), k
        assert abs(la.item() - lf.item()) <= 1e-6 * abs(la.item()), (k, la.item(), lf.item())
        for x, y in zip(ga, gf):
            assert torch.equal(x.reshape(-1), y.reshape(-1)), k
    for x, y in zip(pa, pf):
        assert torch.equal(x, y)
    assert torch.equal(ma, mf) and torch.equal(va, vf)


@pytest.mark.gpu
def test_graphed_train_step_refreshes_coefficients_after_foreign_weight_changes(gpu_device):
    """The fused captures keep (A, p0) across steps; eager steps between replays and `refresh()` after `load_state_dict` must make the
    next replay recompute them: a run that mixes replays, eager steps and a reloaded checkpoint equals the all-eager run bit for bit."""
    opt = hot_path_opt(mesh_dims=[12, 12], hidden_dim=64, num_layers=3, lr=1e-3, decay=0.0, device=str(gpu_device))
    ds = MeshDataset([12, 12], 4, seed=5)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(4)
    state = copy.deepcopy(GNN(ds, opt).state_dict())
    torch.manual_seed(9)
    other = {k: (v + 0.01 * torch.randn_like(v) if 'lin_' in k and v.dtype.is_floating_point else v) for k, v in state.items()}
    res = []
    for mixed in (False, True):
        model, optim, step = _fresh_trainer(ds, opt, gpu_device, state)
        seq = ['e', 'e', 'e', 'load', 'e', 'e'] if not mixed else ['g', 'e', 'g', 'load', 'g', 'g']
        for s_ in seq:
            if s_ == 'load':
                model.load_state_dict(copy.deepcopy(other))            # (into the bucket the parameters are views of)
                step.refresh()
            elif s_ == 'g':
                step(data)
            else:
                step.eager(data)
        torch.cuda.synchronize()
        if mixed:
            assert all(c.fused is not None for c in step._captured.values()), step.fused_reason
        res.append(([p.detach().clone() for p in optim.active], optim.exp_avg.clone()))
    for x, y in zip(res[0][0], res[1][0]):
        assert torch.equal(x, y)
    assert torch.equal(res[0][1], res[1][1])


@pytest.mark.gpu
def test_graphed_train_step_with_eager_optimizer_step(gpu_device):
    """capture_optimizer=False (process groups whose collectives cannot be captured): forward + loss + backward replayed, the
    optimizer step issued eagerly on the gradients of the graph that just ran - also right after a capture of ANOTHER topology
    re-pointed the parameters' .grad."""
    from g_adaptivity_amd import DeviceMeshLoader, GraphedTrainStep
    opt = hot_path_opt(mesh_dims=[10, 10], hidden_dim=32, num_layers=2, lr=1e-3, decay=0.0, device=str(gpu_device))
    ds = MeshDataset([10, 10], 7, seed=1)
    torch.manual_seed(1)
    base = GNN(ds, opt).to(gpu_device).train()
    state = copy.deepcopy(base.state_dict())
    res = []
    for graphed in (False, True):
        model = GNN(ds, opt).to(gpu_device).train(); model.load_state_dict(copy.deepcopy(state))
        optim = FlatAdam(model.parameters(), lr=opt['lr'], capturable=True)
        step = GraphedTrainStep(model, optim, capture_optimizer=False)
        for epoch in range(2):
            for d in DeviceMeshLoader(ds, batch_size=3, shuffle=False, device=gpu_device):      # 3 + 3 + 1
                (step if graphed else step.eager)(d)
        torch.cuda.synchronize()
        res.append([p.detach().clone() for p in model.parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert not torch.equal(res[0][1], list(base.parameters())[1])


@pytest.mark.gpu
def test_graphed_train_step_with_global_cnn_features(gpu_device):
    """The captured iteration with `gnn_inc_glob_feat_f/uu` (GNN.py:240-268: the per-mesh CNN features, MIOpen convolutions,
    gradients through d/dx0 of the block op).  Two things used to synchronise with the host inside the forward and would have
    invalidated a capture: `batch.max().item()` (now the collated batch's own mesh count) and `repeat_interleave(bincount(batch))`
    (now the row gather `per_mesh[batch]`).  MIOpen's weight-gradient kernels accumulate with atomics, so two EAGER runs already
    differ in the last bits (measured 6e-8); the replayed run must sit in that band."""
    from g_adaptivity_amd import DeviceMeshLoader, GraphedTrainStep
    opt = hot_path_opt(mesh_dims=[12, 12], hidden_dim=32, num_layers=2, device=str(gpu_device), gnn_inc_glob_feat_f=True,
                       gnn_inc_glob_feat_uu=True, lr=1e-3)
    ds = MeshDataset([12, 12], 8, seed=0)
    torch.manual_seed(0)
    base = GNN(ds, opt).to(gpu_device).train()
    state = copy.deepcopy(base.state_dict())
    res = []
    for graphed in (False, True):
        m = GNN(ds, opt).to(gpu_device).train(); m.load_state_dict(copy.deepcopy(state))
        o = FlatAdam(m.parameters(), lr=opt['lr'], capturable=True)
        step = GraphedTrainStep(m, o)
        for epoch in range(2):
            for d in DeviceMeshLoader(ds, batch_size=4, shuffle=False, device=gpu_device):
                (step if graphed else step.eager)(d)
        torch.cuda.synchronize()
        res.append({n: p.detach().clone() for n, p in m.named_parameters()})
    moved = 0
    for n, p0 in base.named_parameters():
        assert (res[0][n] - res[1][n]).abs().max().item() <= 1e-6, n
        moved += int((res[0][n] - p0).abs().max().item() > 1e-4)
    assert moved >= 10                                               # conv weights / biases and the 16 CNN tensors were trained


@pytest.mark.gpu
@pytest.mark.parametrize("loss_name", ['l1_native', 'mse_torch'])
def test_graphed_train_step_other_losses_and_eviction(gpu_device, loss_name):
    """`GraphedTrainStep` with the native L1 loss (`F.l1_loss`, run_GNN.py:82) and with torch's own `F.mse_loss` (plain
    `loss.backward()`, no preallocated root gradient), and with room for ONE captured topology only (`max_graphs=1`): alternating
    batch sizes evict and re-capture every time - slower, never wrong.  Replayed training equals the eager loop bit for bit."""
    from g_adaptivity_amd import DeviceMeshLoader, GraphedTrainStep, l1_loss
    loss_fn = l1_loss if loss_name == 'l1_native' else F.mse_loss
    opt = hot_path_opt(mesh_dims=[10, 10], hidden_dim=32, num_layers=2, lr=1e-3, device=str(gpu_device))
    ds = MeshDataset([10, 10], 7, seed=4)                                # batches of 3, 3, 1
    torch.manual_seed(2)
    state = copy.deepcopy(GNN(ds, opt).state_dict())
    res = []
    for graphed in (False, True):
        m = GNN(ds, opt).to(gpu_device).train(); m.load_state_dict(copy.deepcopy(state))
        o = FlatAdam(m.parameters(), lr=opt['lr'], capturable=True)
        step = GraphedTrainStep(m, o, loss_fn=loss_fn, max_graphs=1)
        losses = []
        for epoch in range(2):
            for d in DeviceMeshLoader(ds, batch_size=3, shuffle=False, device=gpu_device):
                losses.append((step if graphed else step.eager)(d).clone())
        torch.cuda.synchronize()
        if graphed:
            assert len(step._captured) == 1
        res.append(([p.detach().clone() for p in m.parameters()], losses))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[0][1], res[1][1]):                               # (the fused captures sum the loss in another fixed order)
        assert torch.equal(a, b) if loss_name == 'mse_torch' else abs(a.item() - b.item()) <= 1e-6 * abs(a.item())
    assert not torch.equal(res[0][0][1], state['conv_layers.0.lin_key.weight'].to(gpu_device))


@pytest.mark.gpu
def test_graphed_train_step_1d_global_features_and_lr_change(gpu_device):
    """(a) a 1-D model with global CNN features has no corner list to count the meshes from - the static batch of a
    capture must carry the collation's graph count, or the captured forward would synchronise with the host (`batch.max().item()`)
    and invalidate the capture; (b) the captured Adam launch carries lr by value - an LR change must re-capture, not be ignored;
    (c) captures are kept in least-recently-used order.  Replayed training equals the eager loop throughout."""
    from g_adaptivity_amd import DeviceMeshLoader, GraphedTrainStep
    opt = hot_path_opt(mesh_dims=[21], hidden_dim=16, num_layers=2, device=str(gpu_device), gnn_inc_glob_feat_f=True,
                       gnn_inc_glob_feat_uu=True, lr=1e-3)
    ds = MeshDataset([21], 10, seed=3)                                   # batches of 4, 4, 2
    torch.manual_seed(0)
    base = GNN(ds, opt).to(gpu_device).train()
    state = copy.deepcopy(base.state_dict())
    res, steps = [], []
    for graphed in (False, True):
        m = GNN(ds, opt).to(gpu_device).train(); m.load_state_dict(copy.deepcopy(state))
        o = FlatAdam(m.parameters(), lr=opt['lr'], capturable=True)
        step = GraphedTrainStep(m, o, max_graphs=2, loss_fn=lambda out, t: F.mse_loss(out, t.reshape(out.shape)))   # 1-D: x_phys is [N]
        for epoch in range(3):
            if epoch == 2:
                o.param_groups[0]['lr'] = 5e-3                           # what an LR scheduler does
            for d in DeviceMeshLoader(ds, batch_size=4, shuffle=False, device=gpu_device):
                (step if graphed else step.eager)(d)
        torch.cuda.synchronize()
        res.append({n: p.detach().clone() for n, p in m.named_parameters()})
        steps.append(step)
    for n in res[0]:
        assert (res[0][n] - res[1][n]).abs().max().item() <= 5e-6, n    # (MIOpen's weight gradients use atomics: last-bit band, through nine Adam steps at lr up to 5e-3)
    g = steps[1]
    assert len(g._captured) == 2 and g._hyper_captured[0] == 5e-3        # re-captured at the new learning rate
    keys = list(g._captured)
    g(next(iter(DeviceMeshLoader(ds, batch_size=4, shuffle=False, device=gpu_device))))   # a hit moves the key to the recent end
    assert list(g._captured)[-1] == keys[0] and len(g._captured) == 2
