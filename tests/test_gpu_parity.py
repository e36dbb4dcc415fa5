"""HIP path vs the CPU oracle on the same seeded inputs (run on the MI355X: pytest -m gpu).

Tolerances (BASELINE.json north_star / BASELINE.md): predicted node coordinates within 1e-5 relative fp32; parameter gradients
within 1e-4 relative, measured against the oracle's fp64 twin.  ONE rule for every case (no per-case factors):

    error(HIP, fp64)  <=  max(1e-4, 1.5 x noise)

where `noise` is the fp32 ORACLE's own error against its fp64 twin (round 5: the band below is consulted only in a NOISY case - one where
that single-run noise reaches half the floor, 5e-5, on some parameter; a case whose oracle is quiet must meet the 1e-4 floor - and
every comparison is RECORDED:
`gpurun_out/r06_parity.json` -> committed as `profiles/r06_parity.json` (one per round, `PARITY_ROUND`): per case, kernel dispatch and parameter the error against
fp64, the single-run noise, the band if it was computed, and which rule admitted it).  Some shapes cannot reach 1e-4 in fp32 at all: the 1-D case
(|grad| ~ 1e-6) and above all BASELINE config 4 (64x64, 6 layers, hidden 128: |grad| ~ 1e-10, the remainder of sums that cancel to
1 part in 1e3..1e4).  There the fp32 result is a draw from a band, for the reference too: its edge order is the iteration order of
a Python set (`src/data.py:430-441`), so its scatter sums run in another order every run.  Measured (tools/diag_accumulation.py,
CPU): the fp32 oracle's error on config 4's `lin_query.weight` gradient ranges 0.76e-4 .. 2.1e-4 over six edge orders of the
SAME batch (2 meshes), 0.9e-4 .. 3.8e-4 at one mesh; accumulating the weight-gradient contraction in fp64 leaves it where it was
(7.57e-5 -> 7.58e-5: the error is rounding of the per-node terms - x, alpha and g through the layers - amplified by the
cancellation, not summation order), and so does taking scores and aggregation on differences x_j - x_i.  So when a gradient misses
the bound against ONE oracle run, `noise` is taken as the maximum over the oracle's runs on four more edge orders of the same batch
(`helpers.edge_order_band`) - the reference's own run-to-run band - and the same bound applies.  The softmax uses expf and a true
division (csrc GADAPT_PRECISE_SOFTMAX): with v_exp_f32 / v_rcp_f32 config 4 measured 3.4e-4.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import edge_order_band, hip_model_like, make_case, oracle_fp64_twin, rel_err

COORD_TOL = 1e-5
GRAD_TOL = 1e-4

import json   # noqa: E402
import os     # noqa: E402
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARITY_ROUND = 'r06'
PARITY_LOG = os.path.join(_ROOT, 'gpurun_out', f'{PARITY_ROUND}_parity.json')


def kernel_sources_sha16(root=_ROOT):
    """Fingerprint of the kernel sources the record was made with (csrc/ + the C-ABI header): tests/test_profiles.py compares it with
    the tree's - a parity record older than the kernels fails the CPU suite."""
    import glob, hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, 'g_adaptivity_amd', 'csrc', '*'))) + [os.path.join(root, 'include', 'gadapt_hip.h')]:
        if os.path.isfile(path) and not path.endswith(('.o', '.so')):
            h.update(os.path.basename(path).encode()); h.update(open(path, 'rb').read())
    return h.hexdigest()[:16]


def _record(case_id, rows, coord):
    """Append one case's margins to gpurun_out/<round>_parity.json (rewritten whole each time: the file stays valid JSON)."""
    os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
    try:
        log = json.load(open(PARITY_LOG))
    except Exception:
        log = {'rule': 'e64 <= max(1e-4, 1.5 * noise); noise = fp32 oracle vs its fp64 twin (one run); the worst of five edge orders '
                       '("band") only in a noisy case: one whose single-run noise reaches 5e-5 (half the floor) on some parameter', 'grad_tol': GRAD_TOL, 'coord_tol': COORD_TOL, 'cases': {}}
    log['cases'][case_id] = {'coordinates': coord, 'gradients': rows}
    log['kernel_sources_sha16'] = kernel_sources_sha16()
    with open(PARITY_LOG, 'w') as fh:
        json.dump(log, fh, indent=1, sort_keys=True)

CASES = [
    # mesh_dims, batch, hidden, layers, conv_type, extra opt
    ((32,), 8, 8, 1, 'GRAND', {}),            # BASELINE config 1 (1-D plumbing)
    ((11, 11), 2, 8, 4, 'GRAND_plus', {}),    # the reference's shipped default (params.py:37,130-134)
    ((11, 11), 3, 4, 2, 'GRAND_plus', {}),
    ((15, 15), 2, 16, 3, 'GRAND_plus', {}),
    ((12, 12), 3, 32, 2, 'GRAND_plus', {}),
    ((23, 23), 2, 8, 3, 'GRAND_plus', {}),    # the largest shipped mesh at the shipped width (params.py:64): per-layer kernels, and the one-launch pair forced
    ((32, 32), 2, 64, 4, 'GRAND_plus', {}),   # BASELINE config 2 shape (small batch)
    ((20, 20), 2, 128, 2, 'GRAND', {}),       # config 4 shape (hidden 128)
    ((13, 13), 5, 64, 3, 'GRAND_plus', {'share_conv': False}),
    ((13, 13), 2, 64, 2, 'GRAND_plus', {'softmax_temp_type': 'fixed', 'softmax_temp': 2.0}),
    ((10, 10), 2, 64, 2, 'GRAND_plus', {'fix_boundary': False, 'self_loops': True}),   # in-degree 7 rows
    ((21,), 3, 8, 3, 'GRAND', {'gnn_inc_feat_f': False}),                               # Burgers features (params.py:148,155)
    # option combinations of GNN.forward: no residual update (GNN.py:293-296), normalised raw features (GNN.py:230-238),
    # per-layer learnable steps with unshared convs (GNN.py:179-180,288-289), 1-D without the boundary surgery
    ((12, 12), 2, 16, 3, 'GRAND', {'residual': False, 'non_lin': 'tanh'}),
    ((12, 12), 2, 64, 2, 'GRAND_plus', {'gnn_normalize': True}),
    ((11, 11), 2, 32, 3, 'GRAND_plus', {'learn_step': True, 'share_conv': False}),
    ((16, 16), 2, 64, 4, 'GRAND_plus', {'learn_step': True}),                           # shared convs, compact slots + d dt sums
    ((12, 12), 2, 128, 2, 'GRAND', {'learn_step': True}),
    # learnable temperature sm_temp_a (GRAND_plus.py:152-154,328-329) through the fused block: compact slots + the SUMS = 2
    # instantiations (per-workgroup d score_scale partials, compact layer-0 kernel, 4-column and compact-gradient variants) - ADVICE r3
    ((16, 16), 2, 64, 2, 'GRAND_plus', {'softmax_temp_type': 'learnable_a'}),
    ((16, 16), 2, 64, 4, 'GRAND_plus', {'softmax_temp_type': 'learnable_a'}),
    ((14, 14), 3, 32, 3, 'GRAND_plus', {'softmax_temp_type': 'learnable_a', 'share_conv': False}),
    ((16, 16), 2, 64, 4, 'GRAND_plus', {'learn_step': True, 'softmax_temp_type': 'learnable_a'}),
    ((12, 12), 2, 32, 2, 'GRAND_plus', {'learn_step': True, 'softmax_temp_type': 'learnable_a', 'share_conv': False}),
    ((15, 15), 2, 16, 3, 'GRAND_plus', {'softmax_temp_type': 'learnable_a'}),
    ((64, 64), 8, 64, 3, 'GRAND_plus', {'learn_step': True, 'softmax_temp_type': 'learnable_a'}),    # fills the GPU: wide forward in production too
    ((17,), 4, 16, 2, 'GRAND_plus', {'fix_boundary': False}),
    # BASELINE config 4 shape: 64x64, 6 layers, hidden 128, GRAND, features [x, y, uu] (two meshes: the oracle stays quick)
    ((64, 64), 2, 128, 6, 'GRAND', {'gnn_inc_feat_f': False}),
    # BASELINE config 5 shape: 128x128 mesh, 20 Euler steps, hidden 64 (one mesh): 128-node mesh rows exceed the LDS window,
    # so this is the mesh-ordered NON-windowed tile path, and 20 layers of error growth in forward and backward
    # (under 'wide-any-size' the forward is the wide kernel's 512-row window.  One mesh, 20 layers: d lin_query.bias measured
    # 1.08e-4 against fp64 with it, 5.7e-5 with the tiled forward; the fp32 oracle's own error there is 1.8e-5 .. 8.8e-5 over six
    # edge orders of the batch - the band rule of the module docstring covers it)
    ((128, 128), 1, 64, 20, 'GRAND_plus', {}),
    # BASELINE config 2 at its full batch (32 meshes 32x32, 4 layers, hidden 64)
    ((32, 32), 32, 64, 4, 'GRAND_plus', {}),
    # The other BASELINE configs at their FULL sizes (the fp64 oracle takes 8 / 16 / 56 s on the GPU box's host cores): config 3's
    # per-GPU shard = the metric workload, config 4, config 5
    ((64, 64), 32, 64, 4, 'GRAND_plus', {}),
    ((64, 64), 32, 64, 4, 'GRAND_plus', {'learn_step': True}),       # the metric workload with learnable steps: SUMS kernels at full size
    ((64, 64), 32, 128, 6, 'GRAND', {'gnn_inc_feat_f': False}),
    ((128, 128), 16, 64, 20, 'GRAND_plus', {}),
]
TRANS_CASES = [((11, 11), 2, 8, 3, 'relu'), ((14, 14), 3, 64, 2, 'tanh'), ((12, 12), 2, 32, 2, 'identity')]
_META = ()
IDS = [f"{'x'.join(map(str, c[0]))}-b{c[1]}-C{c[2]}-L{c[3]}-{c[4]}" + ('-' + ','.join(k for k in c[5] if k not in _META) if [k for k in c[5] if k not in _META] else '') for c in CASES]


def _run(gpu_device, mesh_dims, batch, hidden, layers, conv_type, extra):
    extra = {k: v for k, v in extra.items() if k not in _META}
    opt, ds, data, oracle = make_case(mesh_dims, batch, hidden, layers, conv_type, **extra)
    if extra.get('softmax_temp_type') == 'learnable_a':         # distinct temperatures per distinct conv (the start value 1 hides scale errors)
        with torch.no_grad():
            for k, layer in enumerate({id(l): l for l in oracle.conv_layers}.values()):
                layer.sm_temp_a.fill_(1.3 + 0.25 * k)
    model = hip_model_like(oracle, ds, opt, gpu_device)
    tgt = data.x_phys if data.x_phys.dim() == 2 else data.x_phys.unsqueeze(-1)
    ref = oracle(data)
    F.mse_loss(ref, tgt).backward()
    o64, ref64 = oracle_fp64_twin(oracle, ds, opt, data, tgt)
    model._test_batch = (data.clone().to(gpu_device), tgt.to(gpu_device))
    out = model(model._test_batch[0])
    F.mse_loss(out, model._test_batch[1]).backward()
    torch.cuda.synchronize()
    return oracle, o64, model, ref, ref64, out


# Kernel choice.  tests/conftest.py sets graph.WIDE_MIN_NODES = 0 for the session so that graphs of every
# size go through the wide forward kernel; PRODUCTION keeps wide kernels for batches of >= 24 576 nodes only and runs the tiled,
# LDS-windowed forward below that.  Every hidden-64 mesh case below the limit is therefore run twice - once per forward kernel -
# so both are compared with the oracle directly (the other cases take the same kernels either way).
from g_adaptivity_amd import graph as _graph_mod   # noqa: E402
PRODUCTION_WIDE_MIN_NODES = 24576


def _n_nodes(c):
    n = c[1]
    for d in c[0]:
        n *= d
    return n


def _small_mesh_case(c):
    """Cases the one-launch small-mesh kernels take in training (csrc/gadapt_smallmesh.inc; functional.small_forward_policy): run
    once through them and once through the per-layer launches, so both are compared with the oracle directly."""
    per_mesh = 1
    for d in c[0]:
        per_mesh *= d
    ex = c[5]
    return (c[2] <= 8 or (c[2] == 16 and per_mesh <= 256)) and ex.get('residual', True) and not ex.get('learn_step') \
        and ex.get('softmax_temp_type') != 'learnable_a' and not (c[4] == 'GRAND' and ex.get('non_lin', 'identity') != 'identity')


PARAMS, PARAM_IDS = [], []
for _c, _id in zip(CASES, IDS):
    PARAMS.append(_c + (0, True)); PARAM_IDS.append(_id)
    if _c[2] == 64 and len(_c[0]) == 2 and _c[0][0] <= 64 and _n_nodes(_c) < PRODUCTION_WIDE_MIN_NODES:
        PARAMS.append(_c + (PRODUCTION_WIDE_MIN_NODES, True)); PARAM_IDS.append(_id + '-production-kernel-choice')
    if _small_mesh_case(_c):
        PARAMS.append(_c + (0, False)); PARAM_IDS.append(_id + '-per-layer-launches')
    elif _c[2] <= 16 and _n_nodes(_c) // _c[1] <= 1024 and _c[5].get('residual', True) and not _c[5].get('learn_step') \
            and _c[5].get('softmax_temp_type') != 'learnable_a' and not (_c[4] == 'GRAND' and _c[5].get('non_lin', 'identity') != 'identity'):
        PARAMS.append(_c + (0, None)); PARAM_IDS.append(_id + '-one-launch-forced')          # sizes the kernels take but the policy skips


@pytest.mark.one_dispatch
@pytest.mark.gpu
@pytest.mark.parametrize("mesh_dims,batch,hidden,layers,conv_type,extra,wide_min_nodes,small_mesh", PARAMS, ids=PARAM_IDS)
def test_forward_backward_parity(gpu_device, mesh_dims, batch, hidden, layers, conv_type, extra, wide_min_nodes, small_mesh, request):
    import ctypes as C
    import g_adaptivity_amd.functional as Fn
    from g_adaptivity_amd._native import lib
    keep, _graph_mod.WIDE_MIN_NODES = _graph_mod.WIDE_MIN_NODES, wide_min_nodes
    keep_small, Fn.SMALL_MESH_FORWARD = Fn.SMALL_MESH_FORWARD, small_mesh
    keep_pol = Fn.small_training_policy
    if small_mesh is None:                                  # larger small meshes: force the one-launch pair past its speed policy
        Fn.SMALL_MESH_FORWARD, Fn.small_training_policy = True, (lambda c, max_nodes: c <= 16)
    rows = []
    try:
        lib().gadapt_profile_reset(); lib().gadapt_profile_enable(1)
        try:
            oracle, o64, model, ref, ref64, out = _run(gpu_device, mesh_dims, batch, hidden, layers, conv_type, extra)
            tot, cnt = C.c_double(0.0), C.c_int(0)
            lib().gadapt_profile_read(10, C.byref(tot), C.byref(cnt))      # kernel id 10: the one-launch small-mesh backward
            took_small = cnt.value > 0
        finally:
            lib().gadapt_profile_enable(0); lib().gadapt_profile_reset()
        assert took_small == bool(small_mesh is None or (small_mesh and _small_mesh_case((mesh_dims, batch, hidden, layers, conv_type, extra)))), \
            "which kernels ran is not what the case's id says"
        g = next(iter(model._graphs.values()))
        if wide_min_nodes:                                  # production choice below the limit: the tiled forward must have run
            assert g.wide_deg['t'] == 0
        elif hidden == 64 and len(mesh_dims) == 2 and mesh_dims[0] <= 64 and extra.get('fix_boundary', True):
            assert g.wide_deg['t'] > 0                      # mesh-ordered hidden-64 batch: the wide forward ran
        if extra.get('learn_step'):                         # d L / d dt_l (GNN.py:288-289): the d dt sums of the target pass
            for l in range(layers):
                e64 = rel_err(model.steps[l].grad, o64.steps[l].grad)[0]
                noise = rel_err(oracle.steps[l].grad, o64.steps[l].grad)[0]
                assert e64 <= max(GRAD_TOL, 1.5 * noise), f"steps.{l}.grad vs fp64 oracle: {e64:.2e} (fp32 oracle: {noise:.2e})"
        if extra.get('softmax_temp_type') == 'learnable_a':  # d L / d sm_temp_a: the d score_scale sums, chained through 1/(sqrt(C) T)
            distinct = list({id(l): k for k, l in enumerate(model.conv_layers)}.values())
            for li in distinct:
                gh, g32, g64 = (m.conv_layers[li].sm_temp_a.grad for m in (model, oracle, o64))
                assert gh is not None and gh.shape == g64.shape
                e64, noise = rel_err(gh, g64)[0], rel_err(g32, g64)[0]
                assert e64 <= max(GRAD_TOL, 1.5 * noise), f"layer {li} sm_temp_a.grad vs fp64 oracle: {e64:.2e} (fp32 oracle: {noise:.2e})"
            # bit-reproducible: the per-workgroup partials are summed in a fixed order (no float atomics)
            first = {n_: p_.grad.clone() for n_, p_ in model.named_parameters() if p_.grad is not None}
            model.zero_grad(set_to_none=True)
            F.mse_loss(model(model._test_batch[0]), model._test_batch[1]).backward()
            torch.cuda.synchronize()
            for n_, p_ in model.named_parameters():
                if p_.grad is not None:
                    assert torch.equal(p_.grad, first[n_]), f"{n_}.grad differs between two identical runs"
    finally:
        _graph_mod.WIDE_MIN_NODES = keep
        Fn.SMALL_MESH_FORWARD = keep_small
        Fn.small_training_policy = keep_pol
    band = {}                                               # the oracle's edge-order band, computed at most once and only if needed
    norm, elem = rel_err(out, ref)
    if extra.get('residual', True):
        assert norm <= COORD_TOL and elem <= COORD_TOL, f"x_phys vs fp32 oracle: normwise {norm:.2e} elementwise {elem:.2e}"
    else:
        # without the residual update the output is non_lin(A(x)x - x) itself (GNN.py:293-296): small entries are what is left
        # of a cancellation, so elementwise the bar is the fp32 oracle's own error against fp64, as for the TRANS conv
        elem64, noise_elem = rel_err(out, ref64)[1], rel_err(ref, ref64)[1]
        assert norm <= COORD_TOL and elem64 <= max(COORD_TOL, 2.0 * noise_elem), \
            f"output vs oracle: normwise {norm:.2e}, elementwise vs fp64 {elem64:.2e} (fp32 oracle: {noise_elem:.2e})"
    assert rel_err(out, ref64)[0] <= COORD_TOL
    n_layers = len({id(l) for l in model.conv_layers})
    meas = []
    for li in range(n_layers):
        lo, l64, lh = oracle.conv_layers[li], o64.conv_layers[li], model.conv_layers[li]
        for name in ('lin_query.weight', 'lin_query.bias', 'lin_key.weight'):
            g32 = dict(lo.named_parameters())[name].grad
            g64 = dict(l64.named_parameters())[name].grad
            gh = dict(lh.named_parameters())[name].grad
            meas.append((li, name, rel_err(gh, g64)[0], rel_err(gh, g32)[0], rel_err(g32, g64)[0]))
    # A case is NOISY when the fp32 oracle's own single run misses fp64 by at least half the floor on some parameter: its gradients
    # are then the remainder of a cancellation and one fp32 run - the reference's too - is a draw from a band (module docstring;
    # tools/diag_wide_accuracy.py: on config 5's shape the SAME batch with another weight seed moves the oracle's error from 2e-7 to
    # 5e-4, and which forward kernel lands closer to fp64 changes from seed to seed).  Only a noisy case may consult the band of
    # the oracle's edge orders; a quiet case must meet the 1e-4 floor against one oracle run.
    noisy_case = max(m[4] for m in meas) >= 0.5 * GRAD_TOL
    coord = {'normwise': norm, 'elementwise': elem, 'vs_fp64_normwise': rel_err(out, ref64)[0]}
    failures = []
    for li, name, e64, e32, noise in meas:
        single, band_val = noise, None
        rule = 'floor' if e64 <= GRAD_TOL else 'noise'
        if e64 > max(GRAD_TOL, 1.5 * noise) and noisy_case:
            if not band:
                band.update(edge_order_band(oracle, o64, model._test_batch[0].to('cpu'), model._test_batch[1].cpu()))
            band_val = band[f'conv_layers.{li}.{name}']
            noise = max(noise, band_val)
            rule = 'band'
        ok = e64 <= max(GRAD_TOL, 1.5 * noise) and e32 <= GRAD_TOL + 2 * noise
        rows.append({'parameter': f'conv_layers.{li}.{name}', 'e64': e64, 'e32': e32, 'noise_single_run': single, 'band': band_val,
                     'bound': max(GRAD_TOL, 1.5 * noise), 'rule': rule, 'noisy_case': bool(noisy_case), 'passed': bool(ok)})
        if not ok:
            failures.append(f"layer {li} {name}.grad: {e64:.2e} vs fp64, {e32:.2e} vs fp32 oracle (fp32 oracle, {'worst of its edge orders' if rule == 'band' else 'one run'}: {noise:.2e})")
    _record(request.node.callspec.id, rows, coord)
    assert not failures, failures
    for li in range(n_layers):
        lh, l64 = model.conv_layers[li], o64.conv_layers[li]
        # d/d lin_key.bias vanishes analytically (softmax shift invariance); the oracle's is rounding noise
        assert lh.lin_key.bias.grad.abs().max().item() == 0.0
        assert l64.lin_key.bias.grad.abs().max().item() <= 1e-9 * max(l64.lin_query.bias.grad.abs().max().item(), 1e-30) + 1e-18


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_dims,hidden", [((9, 9), 16), ((12, 12), 64)], ids=['9x9-C16', '12x12-C64'])
def test_global_cnn_features_parity(gpu_device, mesh_dims, hidden):
    """gnn_inc_glob_feat_f/uu (GNN.py:240-268): the per-mesh CNN features ride on the node features; gradients reach the
    convolution weights through d/dx0 of the block op (source pass also run for layer 0)."""
    extra = dict(gnn_inc_glob_feat_f=True, gnn_inc_glob_feat_uu=True)
    oracle, o64, model, ref, ref64, out = _run(gpu_device, mesh_dims, 3, hidden, 2, 'GRAND_plus', extra)
    norm, elem = rel_err(out, ref)
    assert norm <= COORD_TOL and elem <= COORD_TOL, f"x_phys: normwise {norm:.2e} elementwise {elem:.2e}"
    assert rel_err(out, ref64)[0] <= COORD_TOL
    checked = 0
    d32, d64 = dict(oracle.named_parameters()), dict(o64.named_parameters())
    for name, ph in model.named_parameters():
        p32, p64 = d32[name], d64[name]
        if p64.grad is None:
            assert ph.grad is None, name
            continue
        if name.endswith('lin_key.bias'):
            continue                                            # analytically zero; the oracle's is rounding noise
        e64, noise = rel_err(ph.grad, p64.grad)[0], rel_err(p32.grad, p64.grad)[0]
        assert e64 <= max(GRAD_TOL, 1.5 * noise), f"{name}.grad vs fp64 oracle: {e64:.2e} (fp32 oracle: {noise:.2e})"
        checked += name.startswith('global_feature_extractor')
    assert checked == 16                                        # 2 extractors x 4 convs x (weight, bias)


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_dims,batch,hidden,layers,non_lin", TRANS_CASES, ids=[f"{c[0][0]}x{c[0][1]}-C{c[2]}-{c[4]}" for c in TRANS_CASES])
def test_trans_conv_parity(gpu_device, mesh_dims, batch, hidden, layers, non_lin):
    """conv_type='TRANS' (stock TransformerConv, GNN.py:112-113): graph part on the HIP kernels, value / skip projections as
    dense GEMMs; layer by layer with non_lin and the residual update (GNN.py:284-291).  Coordinates and every
    parameter gradient against the oracle."""
    oracle, o64, model, ref, ref64, out = _run(gpu_device, mesh_dims, batch, hidden, layers, 'TRANS', {'non_lin': non_lin})
    norm, elem = rel_err(out, ref64)
    _, noise_elem = rel_err(ref, ref64)                             # the fp32 oracle's own elementwise error (small coordinates
    assert norm <= COORD_TOL                                        #  after cancellation: x + dt * (value + skip terms))
    assert elem <= max(COORD_TOL, 2.0 * noise_elem), f"x_phys elementwise {elem:.2e} (fp32 oracle vs fp64: {noise_elem:.2e})"
    assert rel_err(out, ref)[0] <= COORD_TOL
    d32, d64 = dict(oracle.named_parameters()), dict(o64.named_parameters())
    checked = 0
    for name, ph in model.named_parameters():
        if name not in d64 or d64[name].grad is None:
            continue
        if name.endswith('lin_key.bias'):
            assert ph.grad.abs().max().item() == 0.0                # vanishes analytically (softmax shift invariance)
            continue
        e64, noise = rel_err(ph.grad, d64[name].grad)[0], rel_err(d32[name].grad, d64[name].grad)[0]
        assert e64 <= max(GRAD_TOL, 1.5 * noise), f"{name}.grad vs fp64 oracle: {e64:.2e} (fp32 oracle: {noise:.2e})"
        checked += 1
    assert checked == 7                                             # query w/b, key w, value w/b, skip w/b (shared conv)
