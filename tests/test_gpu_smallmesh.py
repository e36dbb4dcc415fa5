"""One-launch evaluation forward of small-mesh batches (csrc/gadapt_smallmesh.inc): the reference's own sizes
(`src/params.py:37,56,107,130-134`: 11x11 ... 23x23 meshes, hidden 8, four layers; 1-D Burgers on 21 nodes, `params.py:137-159`),
evaluated as `src/utils_eval.py:193-201` / `src/utils_eval_Burgers.py:282-300` do - against the CPU oracle and against the per-layer
launches on the same inputs."""
import ctypes as C

import pytest
import torch

from helpers import hip_model_like, make_case, rel_err

CASES = [
    # mesh_dims, batch, hidden, layers, conv_type, extra
    ((11, 11), 1, 8, 4, 'GRAND_plus', {}),                       # the shipped default, one sample per call
    ((11, 11), 5, 8, 4, 'GRAND_plus', {}),
    ((15, 15), 3, 16, 3, 'GRAND_plus', {}),
    ((21,), 1, 8, 3, 'GRAND', {'gnn_inc_feat_f': False}),        # Burgers features (params.py:148,155)
    ((21,), 4, 8, 1, 'GRAND', {}),
    ((23, 23), 2, 8, 4, 'GRAND_plus', {}),                       # the largest shipped mesh (1024 threads, one lane per node)
    ((20, 20), 2, 8, 3, 'GRAND_plus', {}),                       # 400 nodes: two lanes per node
    ((8, 8), 3, 8, 3, 'GRAND_plus', {}),                         # 64 nodes: four lanes per node in a 256-thread workgroup
    ((19, 19), 2, 32, 2, 'GRAND_plus', {}),                      # the widest rows the kernel takes (policy forced: see the test)
    ((13, 13), 3, 8, 3, 'GRAND_plus', {'share_conv': False, 'learn_step': True}),
    ((12, 12), 2, 4, 2, 'GRAND_plus', {'softmax_temp_type': 'fixed', 'softmax_temp': 2.0}),
    ((10, 10), 2, 16, 2, 'GRAND_plus', {'fix_boundary': False, 'self_loops': True}),
]
IDS = [f"{'x'.join(map(str, c[0]))}-b{c[1]}-C{c[2]}-L{c[3]}-{c[4]}" + ('-' + ','.join(c[5]) if c[5] else '') for c in CASES]


def _launches(kernel_id):
    from g_adaptivity_amd._native import lib
    tot, cnt = C.c_double(0.0), C.c_int(0)
    lib().gadapt_profile_read(kernel_id, C.byref(tot), C.byref(cnt))
    return cnt.value


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_dims,batch,hidden,layers,conv_type,extra", CASES, ids=IDS)
def test_small_mesh_forward_parity(gpu_device, mesh_dims, batch, hidden, layers, conv_type, extra, monkeypatch):
    import g_adaptivity_amd.functional as Fn
    from g_adaptivity_amd._native import lib
    opt, ds, data, oracle = make_case(mesh_dims, batch, hidden, layers, conv_type, **extra)
    model = hip_model_like(oracle, ds, opt, gpu_device).eval()
    oracle.eval()
    d = data.clone().to(gpu_device)
    monkeypatch.setattr(Fn, 'small_forward_policy', lambda c, max_nodes: True)   # every size the kernel takes, not only where it is the faster one
    with torch.no_grad():
        ref = oracle(data)
        lib().gadapt_profile_reset(); lib().gadapt_profile_enable(1)
        try:
            out = model(d)
            torch.cuda.synchronize()
            assert _launches(9) == 1 and _launches(0) == 0, "the evaluation forward of a small-mesh batch must be the one-launch kernel"
            alpha_small = [l.stored_alpha.clone() for l in model.conv_layers] if conv_type == 'GRAND' else None
            lib().gadapt_profile_reset()
            keep, Fn.SMALL_MESH_FORWARD = Fn.SMALL_MESH_FORWARD, False
            try:
                per_layer = model(d)
                torch.cuda.synchronize()
                assert _launches(9) == 0 and _launches(0) == layers
                alpha_layers = [l.stored_alpha.clone() for l in model.conv_layers] if conv_type == 'GRAND' else None
            finally:
                Fn.SMALL_MESH_FORWARD = keep
        finally:
            lib().gadapt_profile_enable(0); lib().gadapt_profile_reset()
    norm, elem = rel_err(out, ref)
    assert norm <= 1e-5 and elem <= 1e-5, f"x_phys vs fp32 oracle: normwise {norm:.2e} elementwise {elem:.2e}"
    assert rel_err(out, per_layer)[0] <= 2e-6, rel_err(out, per_layer)
    if alpha_small is not None:                                   # stored_alpha (GRAND_plus.py:253-256,381): same attention either way
        for a, b in zip(alpha_small, alpha_layers):
            assert rel_err(a, b)[0] <= 1e-5


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_small_mesh_dispatch_limits(gpu_device):
    """Training (autograd on) runs the one-launch forward + one-launch backward pair; learnable steps and meshes too large for a
    workgroup's LDS keep the per-layer launches; a batch whose `batch` vector is not what collation produces (meshes interleaved) is
    refused by the partition check, not mis-computed."""
    import g_adaptivity_amd.functional as Fn
    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt
    from g_adaptivity_amd._native import lib
    opt = hot_path_opt(mesh_dims=[11, 11], hidden_dim=8, num_layers=2, device=str(gpu_device))
    ds = MeshDataset([11, 11], 2, seed=0)
    data = collate(ds.samples).to(gpu_device)
    model = GNN(ds, opt).to(gpu_device)
    lib().gadapt_profile_reset(); lib().gadapt_profile_enable(1)
    try:
        model.train()
        model(data).sum().backward()                              # autograd on: forward keeps the layer inputs, one launch each way
        torch.cuda.synchronize()
        assert _launches(9) == 1 and _launches(10) == 1 and _launches(0) == 0 and _launches(1) == 0
        lib().gadapt_profile_reset()
        ls = GNN(ds, hot_path_opt(mesh_dims=[11, 11], hidden_dim=8, num_layers=2, device=str(gpu_device), learn_step=True)).to(gpu_device).train()
        ls(data).sum().backward()                                 # learnable steps: d dt comes from the per-layer kernels
        torch.cuda.synchronize()
        assert _launches(9) == 0 and _launches(10) == 0 and _launches(0) == 2
        lib().gadapt_profile_reset()
        big = hot_path_opt(mesh_dims=[64, 64], hidden_dim=8, num_layers=2, device=str(gpu_device))
        dsb = MeshDataset([64, 64], 1, seed=0)
        mb = GNN(dsb, big).to(gpu_device).eval()
        with torch.no_grad():
            mb(collate(dsb.samples).to(gpu_device))               # 4096 nodes: more than a workgroup takes
        torch.cuda.synchronize()
        assert _launches(9) == 0 and _launches(0) == 2
    finally:
        lib().gadapt_profile_enable(0); lib().gadapt_profile_reset()
    g = next(iter(model._graphs.values()))
    assert g.mesh_partition(data.batch) is not None
    shuffled = data.batch.clone(); shuffled[0], shuffled[-1] = 1, 0
    assert g.mesh_partition(shuffled) is None


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_small_mesh_training_step_graphed_equals_eager_and_tracks_the_oracle(gpu_device):
    """The reference's training loop at its own size (11 x 11 meshes, hidden 8, four layers; run_GNN.py:99-131) on the one-launch pair:
    the captured step (`GraphedTrainStep`) is bit-identical to the eager loop, the forward + backward are two launches, and three
    epochs of Adam stay on the CPU oracle's trajectory."""
    import copy
    import torch.nn.functional as F
    from g_adaptivity_amd import DeviceMeshLoader, GNN, GraphedTrainStep, MeshDataset, hot_path_opt
    from g_adaptivity_amd.optim import FlatAdam
    from g_adaptivity_amd._native import lib
    from oracle.pyg_restatement import OracleGNN
    from g_adaptivity_amd import collate
    opt = hot_path_opt(mesh_dims=[11, 11], hidden_dim=8, num_layers=4, lr=1e-3, decay=0.0, device=str(gpu_device))
    ds = MeshDataset([11, 11], 10, seed=2)                                  # batches of 4, 4, 2
    torch.manual_seed(0)
    base = GNN(ds, opt).to(gpu_device).train()
    state = copy.deepcopy(base.state_dict())
    res = []
    for graphed in (False, True):
        m = GNN(ds, opt).to(gpu_device).train(); m.load_state_dict(copy.deepcopy(state))
        o = FlatAdam(m.parameters(), lr=opt['lr'], capturable=True)
        step = GraphedTrainStep(m, o)
        lib().gadapt_profile_reset(); lib().gadapt_profile_enable(1)
        for epoch in range(3):
            for d in DeviceMeshLoader(ds, batch_size=4, shuffle=False, device=gpu_device):
                (step if graphed else step.eager)(d)
        torch.cuda.synchronize()
        lib().gadapt_profile_enable(0)
        if not graphed:
            assert _launches(9) == 9 and _launches(10) == 9 and _launches(0) == 0 and _launches(1) == 0
        lib().gadapt_profile_reset()
        res.append({n: p.detach().clone() for n, p in m.named_parameters()})
    for n in res[0]:
        assert torch.equal(res[0][n], res[1][n]), n
    oopt = dict(opt); oopt['device'] = 'cpu'
    oracle = OracleGNN(ds, oopt).train()
    oracle.load_state_dict({k: v.cpu() for k, v in state.items()})
    oo = torch.optim.Adam([p for p in oracle.parameters() if p.requires_grad], lr=opt['lr'])
    for epoch in range(3):
        for s in range(0, 10, 4):
            d = collate(ds.samples[s:s + 4])
            oo.zero_grad(); F.mse_loss(oracle(d), d.x_phys).backward(); oo.step()
    for n, p in oracle.named_parameters():
        if p.requires_grad and not n.endswith('lin_skip.weight'):
            assert (res[1][n].cpu() - p.detach()).abs().max().item() <= 2e-5, n


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,hidden,share", [(8, 8, True), (11, 8, True), (15, 8, True), (20, 8, True), (23, 8, True), (27, 4, True), (11, 16, True),
                                                   (15, 16, True), (20, 16, True), (13, 8, True), (13, 16, True), (13, 8, False), (13, 16, False)],
                         ids=lambda v: {True: 'shared-conv', False: 'per-layer-convs'}.get(v, str(v)) if isinstance(v, bool) else str(v))
def test_small_mesh_pair_matches_per_layer_kernels_at_every_lane_split(gpu_device, mesh_n, hidden, share, monkeypatch):
    """Training forward + backward through the one-launch pair against the per-layer kernels on the same model and batch, at mesh
    sizes that select every (threads, lanes per node) instantiation of csrc/gadapt_smallmesh.inc: 64 / 121 / 225 / 400 / 529 / 729
    nodes per mesh = four, four, two, one, one, one lanes in the backward and four, four, four, two, one, one in the forward.  Both
    are fp32 fma chains of the same formulation: parameter gradients agree to reassociation level.  per-layer-convs: the
    backward's `share_conv=False` branch in TRAINING - weights reloaded per layer, one slab set per conv, a slab reduction per conv -
    at two sizes (odd node counts: the LDS regions behind the index arrays must still be 16-byte aligned)."""
    import g_adaptivity_amd.functional as Fn
    import torch.nn.functional as F
    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt
    from g_adaptivity_amd._native import lib
    monkeypatch.setattr(Fn, 'small_forward_policy', lambda c, max_nodes: True)
    monkeypatch.setattr(Fn, 'small_training_policy', lambda c, max_nodes: True)
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=hidden, num_layers=3, share_conv=share, device=str(gpu_device))
    ds = MeshDataset([mesh_n, mesh_n], 3, seed=mesh_n)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(3)
    model = GNN(ds, opt).to(gpu_device).train()
    res = {}
    for small in (True, False):
        monkeypatch.setattr(Fn, 'SMALL_MESH_FORWARD', small)
        lib().gadapt_profile_reset(); lib().gadapt_profile_enable(1)
        try:
            model.zero_grad()
            out = model(data)
            F.mse_loss(out, data.x_phys).backward()
            torch.cuda.synchronize()
            assert (_launches(10) == 1) == small, "which backward ran is not what the loop variable says"
        finally:
            lib().gadapt_profile_enable(0); lib().gadapt_profile_reset()
        res[small] = (out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None])
    assert rel_err(res[True][0], res[False][0])[0] <= 2e-6
    assert len(res[True][1]) >= (3 if share else 9)
    for a, b in zip(res[True][1], res[False][1]):
        if b.abs().max() > 0:
            assert rel_err(a, b)[0] <= 5e-5, rel_err(a, b)


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_n,hidden", [(9, 8), (14, 8), (21, 8), (9, 16)], ids=lambda v: str(v))
def test_small_mesh_pair_takes_long_rows(gpu_device, mesh_n, hidden, monkeypatch):
    """Rows of more than eight in-neighbours (and nodes with more than eight out-edges) take the loop paths of the one-launch
    kernels, which the lanes of a node walk with a stride: every mesh of the batch gets a hub node with 14 extra in-edges and one
    with 12 extra out-edges (inside the mesh: the partition check must still pass).  Forward, attention weights and parameter
    gradients against the per-layer kernels on the same model and batch, at one / two / four lanes per node."""
    import g_adaptivity_amd.functional as Fn
    import torch.nn.functional as F
    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt
    from g_adaptivity_amd._native import lib
    monkeypatch.setattr(Fn, 'small_forward_policy', lambda c, max_nodes: True)
    monkeypatch.setattr(Fn, 'small_training_policy', lambda c, max_nodes: True)
    opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=hidden, num_layers=3, device=str(gpu_device), fix_boundary=False)
    ds = MeshDataset([mesh_n, mesh_n], 3, seed=1)
    data = collate(ds.samples)
    per = mesh_n * mesh_n
    gen = torch.Generator().manual_seed(11)
    extra = []
    for m in range(3):
        base = m * per
        hub_in, hub_out = base + per // 2, base + per // 3
        src = base + torch.randperm(per, generator=gen)[:14]
        dst = base + torch.randperm(per, generator=gen)[:12]
        extra += [torch.stack([src, torch.full_like(src, hub_in)]), torch.stack([torch.full_like(dst, hub_out), dst])]
    data.edge_index = torch.cat([data.edge_index] + extra, dim=1)
    data = data.to(gpu_device)
    torch.manual_seed(2)
    model = GNN(ds, opt).to(gpu_device).train()
    res = {}
    for small in (True, False):
        monkeypatch.setattr(Fn, 'SMALL_MESH_FORWARD', small)
        lib().gadapt_profile_reset(); lib().gadapt_profile_enable(1)
        try:
            model.zero_grad()
            out = model(data)
            F.mse_loss(out, data.x_phys).backward()
            torch.cuda.synchronize()
            assert (_launches(10) == 1) == small
        finally:
            lib().gadapt_profile_enable(0); lib().gadapt_profile_reset()
        res[small] = (out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None])
    g = next(iter(model._graphs.values()))
    assert g.max_in_degree > 8
    assert rel_err(res[True][0], res[False][0])[0] <= 2e-6
    for a, b in zip(res[True][1], res[False][1]):
        if b.abs().max() > 0:
            assert rel_err(a, b)[0] <= 5e-5, rel_err(a, b)
