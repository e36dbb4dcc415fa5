"""GPU tests of the callers either side of the path: device-resident collation, the graph
cache's content key, optimizer state, and the two-rank data-parallel step on real launches (pytest -m gpu)."""
import copy
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

from g_adaptivity_amd import DeviceMeshLoader, GNN, MeshDataset, MeshLoader, collate, hot_path_opt, mse_loss
from g_adaptivity_amd.optim import FlatAdam
from helpers import hip_model_like, make_case, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_device_mesh_loader_equals_host_collation(gpu_device):
    """`DeviceMeshLoader` batches == `collate(...).to(device)` field by field, the model gives bit-identical output on
    both, and every batch of the epoch hits ONE cached CSR (topology tensors are shared, `src/data.py:290`)."""
    ds = MeshDataset([14, 14], 10, seed=3)
    opt = hot_path_opt(mesh_dims=[14, 14], hidden_dim=64, num_layers=3, device=str(gpu_device), show_mesh_evol_plots='False')
    torch.manual_seed(0)
    model = GNN(ds, opt).to(gpu_device).eval()
    host = list(MeshLoader(ds, batch_size=4, shuffle=False))
    dev = list(DeviceMeshLoader(ds, batch_size=4, shuffle=False, device=gpu_device))
    assert len(host) == len(dev) == 3
    for h, d in zip(host, dev):
        h = h.to(gpu_device)
        for k in ('x_comp', 'x_phys', 'f_tensor', 'uu_tensor', 'u_true_tensor', 'edge_index', 'batch', 'to_boundary_edge_mask',
                  'to_corner_nodes_mask', 'diff_boundary_edges_mask'):
            assert torch.equal(getattr(h, k), getattr(d, k)), k
            assert getattr(d, k).device == gpu_device
        assert [list(map(int, c)) for c in h.corner_nodes] == [list(map(int, c)) for c in d.corner_nodes]
        with torch.no_grad():
            assert torch.equal(model(h), model(d))
    # 2 topologies (batches of 4 and the ragged last batch of 2), each built once for host AND device batches alike
    assert len(model._graphs) == 2
    # a shuffled device epoch covers every sample once
    seen = torch.cat([b.idx for b in DeviceMeshLoader(ds, batch_size=4, shuffle=True, device=gpu_device)]).sort().values
    assert seen.tolist() == list(range(10))


@pytest.mark.gpu
def test_graph_cache_is_keyed_on_content(gpu_device):
    """Batches with EQUAL node and edge counts but another edge order, another connectivity or other boundary masks each get
    their own CSR and the oracle's answer, without anyone clearing the cache (the reference redoes the edge surgery every
    forward, `src/GNN.py:206-218`); feeding a batch again reuses its graph."""
    opt, ds, data, oracle = make_case((10, 10), 2, 16, 2)
    model = hip_model_like(oracle, ds, opt, gpu_device).eval()

    def both(d):
        with torch.no_grad():
            return model(d.clone().to(gpu_device)), oracle(d)

    out_a, ref_a = both(data)
    assert len(model._graphs) == 1 and rel_err(out_a, ref_a)[0] <= 1e-5
    # (1) same edges, another order: same answer (up to fp32 reassociation), but its own cache entry
    perm = torch.randperm(data.edge_index.shape[1], generator=torch.Generator().manual_seed(0))
    d1 = data.clone()
    d1.edge_index = data.edge_index[:, perm]
    for k in ('to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask'):
        setattr(d1, k, getattr(data, k)[perm])
    out_1, ref_1 = both(d1)
    assert len(model._graphs) == 2 and rel_err(out_1, ref_1)[0] <= 1e-5 and rel_err(out_1, out_a)[0] <= 1e-5
    # (2) same counts, another connectivity: one interior edge pair re-targeted (an "edge flip")
    d2 = data.clone()
    ei = d2.edge_index.clone()
    free = (~(data.to_boundary_edge_mask | data.to_corner_nodes_mask | data.diff_boundary_edges_mask)).nonzero().flatten()
    e = int(free[len(free) // 2])
    old_src = int(ei[0, e])
    ei[0, e] = old_src + 1 if old_src % 10 < 8 else old_src - 1        # another interior source for that target
    d2.edge_index = ei
    out_2, ref_2 = both(d2)
    assert len(model._graphs) == 3 and rel_err(out_2, ref_2)[0] <= 1e-5
    assert not torch.equal(out_2, out_a)                                # a stale CSR would have reproduced out_a
    # (3) same edge list, other masks: fix one more edge (drop it from the diffusion)
    d3 = data.clone()
    m = d3.to_boundary_edge_mask.clone()
    m[e] = True
    d3.to_boundary_edge_mask = m
    out_3, ref_3 = both(d3)
    assert len(model._graphs) == 4 and rel_err(out_3, ref_3)[0] <= 1e-5 and not torch.equal(out_3, out_a)
    # feeding a known batch again builds nothing
    out_again, _ = both(data)
    assert len(model._graphs) == 4 and torch.equal(out_again, out_a)
    # a single un-collated sample (corner_nodes is one array, not a list) is handled without touching the caller's object
    single = ds.samples[0].clone()
    single.batch = torch.zeros(single.x_comp.shape[0], dtype=torch.int64)
    kept = single.corner_nodes
    with torch.no_grad():
        out_s = model(single.clone().to(gpu_device))
    one = collate([ds.samples[0]])
    assert rel_err(out_s, oracle(one))[0] <= 1e-5 and single.corner_nodes is kept


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_flat_adam_state_dict_param_groups_and_gradient_set_changes(gpu_device):
    torch.manual_seed(0)

    def make():
        return [torch.nn.Parameter(torch.randn(32, 32, device=gpu_device)), torch.nn.Parameter(torch.randn(32, device=gpu_device)),
                torch.nn.Parameter(torch.randn(5, device=gpu_device))]

    ps = make()
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ours, ref = FlatAdam(ps, lr=1e-2, weight_decay=0.01), torch.optim.Adam(qs, lr=1e-2, weight_decay=0.01)
    gens = torch.Generator(device=gpu_device).manual_seed(1)

    def step(optim_a, params_a, optim_b, params_b, which=(0, 1)):
        optim_a.zero_grad(); optim_b.zero_grad()
        for k in which:
            g = torch.randn(params_a[k].shape, device=gpu_device, generator=gens)
            params_a[k].grad = g.clone(); params_b[k].grad = g.clone()
        optim_a.step(); optim_b.step()

    for _ in range(3):
        step(ours, ps, ref, qs)
    # LR schedulers write param_groups[0]['lr']
    ours.param_groups[0]['lr'] = 5e-3; ref.param_groups[0]['lr'] = 5e-3
    assert ours.lr == 5e-3
    step(ours, ps, ref, qs)
    for p, q in zip(ps, qs):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    # checkpoint -> fresh optimizer over fresh parameter objects -> same trajectory as the uninterrupted pair
    state = copy.deepcopy(ours.state_dict())
    assert state['step'] == 4 and state['exp_avg'].numel() == 32 * 32 + 32
    ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    resumed = FlatAdam(ps2, lr=1.0)                                     # wrong lr on purpose: the state carries the right one
    resumed.load_state_dict(state)
    assert resumed.lr == 5e-3 and resumed.step_count == 4
    for _ in range(2):
        ours.zero_grad(); resumed.zero_grad(); ref.zero_grad()
        for k in (0, 1):
            g = torch.randn(ps[k].shape, device=gpu_device, generator=gens)
            ps[k].grad = g.clone(); ps2[k].grad = g.clone(); qs[k].grad = g.clone()
        ours.step(); resumed.step(); ref.step()
    torch.cuda.synchronize()
    for p, p2, q in zip(ps, ps2, qs):
        assert torch.equal(p, p2)
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-6)
    # a parameter that gets no gradient this time is skipped like torch.optim.Adam skips `grad is None` ...
    before = ps[1].detach().clone()
    step(ours, ps, ref, qs, which=(0,))
    torch.cuda.synchronize()
    assert torch.equal(ps[1], before) and torch.allclose(ps[0], qs[0], rtol=1e-5, atol=1e-6)
    # ... and a parameter whose FIRST gradient arrives after steps were taken is refused loudly (one bucket = one step count)
    with pytest.raises(RuntimeError, match="entered the gradient set"):
        step(ours, ps, ref, qs, which=(0, 2))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("graphed", [False, True], ids=['eager-step', 'graphed-step'])
def test_two_rank_step_equals_full_batch_step(gpu_device, graphed):
    """Two ranks (fresh child processes sharing the one GPU of the test box, gloo for the collective - RCCL needs a GPU
    per rank) each run forward + backward on THEIR shard through the HIP kernels, all-reduce the flat gradient bucket and
    take the fused Adam step; the result must equal the single-process step on the full batch (equal shards: the mean of
    the shard means is the full-batch mean).  This executes `FlatAdam.step()` behind a real reduce.  graphed-step: the same
    through `GraphedTrainStep(capture_optimizer=False)` - forward + loss + backward replayed from a hipGraph captured in
    thread_local mode inside a live process group, the gloo all-reduce and the device-stepped Adam issued eagerly on the
    gradients of the replay."""
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GADAPT_DDP_CHECK_OUT=os.path.join(ROOT, 'gpurun_out'),
               GADAPT_DDP_GRAPHED='1' if graphed else '0')
    os.makedirs(env['GADAPT_DDP_CHECK_OUT'], exist_ok=True)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'tools', 'ddp_step_check.py')],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['world'] == 2 and d['bucket_floats'] == 2 * (64 * 64 + 64)
    assert d['grad_rel_err'] <= 1e-5, d          # averaged shard gradients vs the full-batch gradient
    assert d['param_max_abs_diff'] <= 1e-6, d    # after 3 Adam steps (lr 1e-3)
    assert d['ranks_identical'] is True          # replicas stay bit-identical after the reduce
    assert d['graphed'] == graphed


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("capture_allreduce", [False, True], ids=['eager-collective', 'capture-allreduce-requested'])
def test_bench_two_ranks_rehearsal(gpu_device, capture_allreduce):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), rehearsed on the ONE
    GPU of the test box: GADAPT_BENCH_SHARE_GPU=1 maps both ranks onto it and GADAPT_BENCH_BACKEND=gloo carries the collectives
    (RCCL needs a GPU per rank).  The N > 1 launch mode runs: the fused iteration issued as its C-ABI calls - forward + loss + backward,
    chain rule, the all-reduce of the flat gradient, device-stepped Adam.  With GADAPT_BENCH_CAPTURE_ALLREDUCE=1 the collective would be
    captured too - only RCCL can be, so with gloo the request must be declined up front (an invalidated capture cannot be
    recovered from in-process on this ROCm: tools/capture_recovery_probe.py) and the run proceeds in the default mode.  One JSON
    line, n_gpus 2, finite numbers, weak scaling (64 meshes per step)."""
    import math
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GADAPT_BENCH_SHARE_GPU='1', GADAPT_BENCH_BACKEND='gloo')
    if capture_allreduce:
        env['GADAPT_BENCH_CAPTURE_ALLREDUCE'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    if r.returncode != 0:                                   # the ranks' tracebacks, where a truncated assertion message cannot hide them
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', f"bench_two_ranks_{'capture' if capture_allreduce else 'eager'}.err"), 'w') as fh:
            fh.write(r.stdout + '\n---- stderr ----\n' + r.stderr)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 5 and d['warmup'] == 2 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 64 and d['config']['parallelism'] == 'dp2'
    assert math.isfinite(d['value']) and d['value'] > 0 and math.isfinite(d['ms_per_step']) and d['ms_per_step'] > 0
    assert d['value'] == pytest.approx(64 / (d['ms_per_step'] * 1e-3), rel=1e-3)
    assert d['cpu_baseline'] is None                        # rank 0 at N = 1 only
    assert d['roofline'] is not None and math.isfinite(d['roofline']['frac'])
    # the fused step under data parallelism is ISSUED (3 C-ABI calls + the all-reduce per step: nothing to capture around a collective,
    # and a replay costs idle GPU time - docs/measurements.md K); --launch graph keeps the capture
    assert d['config']['launch'] == 'issued: 3 C-ABI calls (13 launches) + all-reduce per step', d['config']['launch']
    if capture_allreduce:
        assert 'GADAPT_BENCH_CAPTURE_ALLREDUCE=1 ignored' in r.stderr


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_ranks(gpu_device):
    """Plain `python bench.py --gpus 2` - how the driver invokes `--gpus 1` - must start the two ranks itself (a child
    torch.distributed.run before anything touches a GPU) and relay ONE JSON line with n_gpus 2, the all-reduce time per step and
    the collective's description.  Rehearsed on the one GPU of the test box (shared GPU, gloo collectives), as above."""
    import math
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(GADAPT_BENCH_SHARE_GPU='1', GADAPT_BENCH_BACKEND='gloo')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--windows', '2', '--no-companion'],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 64 and d['scaling'] == 'weak'
    assert math.isfinite(d['value']) and d['value'] > 0
    assert d['rccl']['world'] == 2 and d['rccl']['backend'] == 'gloo' and d['rccl']['captured'] is False and d['rccl']['bucket_bytes'] == 4 * (2 * 64 * 64 + 2 * 64)
    assert d['allreduce_us_per_step'] is not None and math.isfinite(d['allreduce_us_per_step']) and d['allreduce_us_per_step'] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("hidden", [16, 64])
def test_mixed_size_batch_parity(gpu_device, hidden):
    """`data_type='randg_mix'`: batches of 9x9 + 12x12 + 7x7 meshes from `Mixed_DataLoader` (excluded keys in `batch_dict`)
    through the HIP model, forward and gradients against the oracle; two batch compositions -> two cached graphs."""
    from g_adaptivity_amd import Mixed_DataLoader, MixedMeshDataset
    from oracle.pyg_restatement import OracleGNN
    ds = MixedMeshDataset([9, 12, 7], 6, seed=1)
    opt = hot_path_opt(mesh_dims=[9, 9], hidden_dim=hidden, num_layers=3, data_type='randg_mix')
    torch.manual_seed(0)
    oracle = OracleGNN(ds, dict(opt))
    model = hip_model_like(oracle, ds, opt, gpu_device)
    loader = Mixed_DataLoader(ds, batch_size=3, shuffle=False, follow_batch=[], exclude_keys=['pde_params'])
    for k, data in enumerate(loader):
        oracle.zero_grad(); model.zero_grad()
        ref = oracle(data)
        F.mse_loss(ref, data.x_phys).backward()
        out = model(data.clone().to(gpu_device))
        F.mse_loss(out, data.x_phys.to(gpu_device)).backward()
        torch.cuda.synchronize()
        assert rel_err(out, ref)[0] <= 1e-5
        for name in ('lin_query.weight', 'lin_query.bias', 'lin_key.weight'):
            want = dict(oracle.conv_layers[0].named_parameters())[name].grad
            got = dict(model.conv_layers[0].named_parameters())[name].grad
            assert rel_err(got, want)[0] <= 2e-4, (k, name, rel_err(got, want))
    assert len(model._graphs) == 1            # both batches: the same three meshes in the same order -> one topology


@pytest.mark.gpu
def test_unit_gradient_is_plain_backward(gpu_device):
    """`loss.backward(gradient=unit_gradient(dev))` gives bit-identical gradients to `loss.backward()` for the native
    losses (the derivative is handed on without the multiplication by 1), and still multiplies for any other root."""
    from g_adaptivity_amd import mse_loss, l1_loss, unit_gradient
    opt = hot_path_opt(mesh_dims=[12, 12], hidden_dim=32, num_layers=2, device=str(gpu_device))
    ds = MeshDataset([12, 12], 3, seed=2)
    data = collate(ds.samples).to(gpu_device)
    torch.manual_seed(1)
    model = GNN(ds, opt).to(gpu_device).train()
    params = [p for p in model.parameters() if p.requires_grad]

    def grads(fn, root):
        for p in params:
            p.grad = None
        loss = fn(model(data), data.x_phys)
        loss.backward() if root is None else loss.backward(gradient=root)
        return [p.grad.clone() for p in params if p.grad is not None]

    for fn in (mse_loss, l1_loss):
        plain, unit = grads(fn, None), grads(fn, unit_gradient(gpu_device))
        assert len(plain) == len(unit) > 0 and all(torch.equal(a, b) for a, b in zip(plain, unit))
        half = grads(fn, torch.full((), 0.5, device=gpu_device))
        assert all(torch.equal(0.5 * a, b) for a, b in zip(plain, half))       # a power of two: exact
    assert unit_gradient(gpu_device) is unit_gradient(gpu_device) and float(unit_gradient(gpu_device)) == 1.0


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_allreduce_flat_on_a_caller_owned_communicator(gpu_device):
    """`gadapt_allreduce_flat` with a communicator the caller created itself (RCCL through ctypes, one rank - what a one-GPU
    box offers): SUM and AVERAGE over one rank leave the bucket as it is, the call is enqueued on the given stream, and
    a null communicator is refused.  (Two ranks need two GPUs: the N > 1 path of the Python mirror is covered by
    test_two_rank_step_equals_full_batch_step with torch.distributed.)"""
    import ctypes as C
    from g_adaptivity_amd import _native
    rccl = C.CDLL('librccl.so')                                  # the copy torch has loaded already

    class UniqueId(C.Structure):
        _fields_ = [('internal', C.c_char * 128)]

    uid, comm = UniqueId(), C.c_void_p()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(gpu_device)
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        lib = _native.lib()
        bucket = torch.randn(2 * (64 * 64 + 64), device=gpu_device)
        ref = bucket.clone()
        stream = torch.cuda.current_stream(gpu_device).cuda_stream
        for average in (0, 1):
            assert lib.gadapt_allreduce_flat(comm, bucket.data_ptr(), bucket.numel(), average, stream) == 0
            torch.cuda.synchronize()
            assert torch.equal(bucket, ref)
        assert lib.gadapt_allreduce_flat(None, bucket.data_ptr(), bucket.numel(), 0, stream) < 0
        assert b'allreduce_flat' in lib.gadapt_last_error()
    finally:
        rccl.ncclCommDestroy(comm)


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_rccl_allreduce_captured_in_a_hipgraph_one_rank(gpu_device):
    """The rehearsal `bench.py --gpus N` (N > 1, RCCL) runs on every rank before it captures the gradient all-reduce with the step
    (`g_adaptivity_amd/rccl_probe.py`), here with ONE rank - what a one-GPU box can host: a child process creates an RCCL process
    group, captures copy -> ncclAllReduce -> scale in a hipGraph (thread_local capture mode, as the step does), replays it on
    four inputs and checks the sums."""
    from g_adaptivity_amd.rccl_probe import rehearse
    keep = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    try:
        assert rehearse(timeout=300.0)
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_allreduce_flat_captured_in_a_hipgraph_one_rank(gpu_device):
    """The C-ABI collective `gadapt_allreduce_flat` on a caller-owned one-rank communicator inside a captured hipGraph, replayed on
    changing inputs (child process: tools/rccl_cabi_capture_probe.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_cabi_capture_probe.py')], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0 and 'CABI_CAPTURE_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.one_dispatch
@pytest.mark.gpu
def test_rccl_rehearsal_from_torchrun_workers(gpu_device):
    """`rehearse()` as `bench.py --gpus N` calls it: from torch.distributed.run workers, before they touch the GPU.  The children
    must rendezvous on their own port (not on the elastic agent's store: TORCHELASTIC_* is stripped from their environment) and reach
    RCCL; on this one-GPU box two ranks cannot form an RCCL group (one GPU per rank), so the verdict is False on both ranks - and
    it must come quickly, not after a rendezvous timeout."""
    import time
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GADAPT_BENCH_SHARE_GPU='1', GADAPT_PROBE_TIMEOUT='60')
    t0 = time.time()
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'tools', 'rehearse_under_torchrun.py')],
                       capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    took = time.time() - t0
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    import re
    verdicts = re.findall(r'rank (\d+) rehearse -> (\w+)', r.stdout)     # (the two ranks' lines can share a line: unsynchronised stdout)
    assert sorted(verdicts) == [('0', 'False'), ('1', 'False')], r.stdout[-1500:]
    assert took < 120, f"rehearsal took {took:.0f} s: a rendezvous timeout, not an RCCL refusal"
