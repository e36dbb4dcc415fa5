#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ FROM THE ORACLE (oracle/pyg_restatement.py).

The reference has no fixtures for this path and cannot be imported here, so these
vectors pin the oracle against regressions and give the GPU tests a reference that does not need the
oracle at run time; they do not pin the oracle to the reference (parity unpinned).

    python tests/golden/make_golden.py            # rewrites every G*.npz (fp32 run + fp64 twin)
    python tests/golden/make_golden.py G4 G5      # only the named cases
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from g_adaptivity_amd import MeshDataset, collate, hot_path_opt          # noqa: E402
from oracle.pyg_restatement import OracleGNN, masked_edge_index           # noqa: E402

CASES = {
    # name: mesh_dims, batch, hidden, layers, conv_type, opt overrides   (G4/G5 = the shapes of
    # BASELINE.json configs 4 and 5 at one mesh per batch)
    'G1_1d_n32_b8_GRAND_L1_C8': ((32,), 8, 8, 1, 'GRAND', {}),
    'G2_2d_n11_b2_GRANDplus_L4_C8': ((11, 11), 2, 8, 4, 'GRAND_plus', {}),
    'G3_2d_n32_b2_GRANDplus_L4_C64': ((32, 32), 2, 64, 4, 'GRAND_plus', {}),
    'G4_2d_n64_b1_GRAND_L6_C128_xyuu': ((64, 64), 1, 128, 6, 'GRAND', {'gnn_inc_feat_f': False}),   # params.py:148,155 features
    'G5_2d_n128_b1_GRANDplus_L20_C64': ((128, 128), 1, 64, 20, 'GRAND_plus', {}),
}
ONLY = [a for a in sys.argv[1:]]


def permuted(data, seed):
    """Same batch, another edge order (edge_index and the per-edge masks together): the reference's edge order is the iteration
    order of a Python set (src/data.py:430-441) and changes from run to run, and with it the fp32 summation order of its scatters."""
    d = data.clone()
    perm = torch.randperm(d.edge_index.shape[1], generator=torch.Generator().manual_seed(seed))
    d.edge_index = d.edge_index[:, perm]
    for m in ('to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask'):
        setattr(d, m, getattr(d, m)[perm])
    return d


def relerr(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-300))


def run(dtype, ds, data, opt, state):
    m = OracleGNN(ds, dict(opt)).to(dtype)
    m.load_state_dict({k: v.to(dtype) for k, v in state.items()})
    d = data.clone()
    for k in ('x_comp', 'f_tensor', 'uu_tensor'):
        setattr(d, k, getattr(d, k).to(dtype))
    tgt = data.x_phys if data.x_phys.dim() == 2 else data.x_phys.unsqueeze(-1)
    x_phys, x_full, alphas, ei = m(d, return_all=True)
    F.mse_loss(x_phys, tgt.to(dtype)).backward()
    lay = m.conv_layers[0]
    return dict(x_phys=x_phys.detach().numpy(), alpha_last=alphas[-1].detach().numpy().reshape(-1),
                d_wq=lay.lin_query.weight.grad.numpy(), d_bq=lay.lin_query.bias.grad.numpy(),
                d_wk=lay.lin_key.weight.grad.numpy(), d_bk=lay.lin_key.bias.grad.numpy()), ei


for name, (mesh_dims, batch, hidden, layers, conv, over) in CASES.items():
    if ONLY and not any(name.startswith(o) for o in ONLY):
        continue
    opt = hot_path_opt(mesh_dims=list(mesh_dims), hidden_dim=hidden, num_layers=layers, conv_type=conv, **over)
    ds = MeshDataset(mesh_dims, batch, seed=0)
    data = collate(ds.samples)
    torch.manual_seed(0)
    state = OracleGNN(ds, dict(opt)).state_dict()
    out32, ei = run(torch.float32, ds, data, opt, state)
    out64, _ = run(torch.float64, ds, data, opt, state)
    # band_<grad>: the fp32 oracle's worst relative error against the fp64 twin over FIVE edge orders of this batch (the stored one
    # and four permutations) - the run-to-run band of the reference's own fp32 path, what tests/test_gpu_ops.py::test_golden_fixtures
    # measures the HIP path's error against (tests/test_gpu_parity.py docstring)
    band = {k: relerr(out32[k], out64[k]) for k in ('d_wq', 'd_bq', 'd_wk')}
    for seed in range(1, 5):
        outp, _ = run(torch.float32, ds, permuted(data, seed), opt, state)
        for k in band:
            band[k] = max(band[k], relerr(outp[k], out64[k]))
    lay = 'conv_layers.0.'
    np.savez_compressed(
        os.path.join(HERE, name + '.npz'),
        mesh_dims=np.array(mesh_dims), batch=batch, hidden=hidden, layers=layers, conv_type=conv,
        inc_f=int(bool(opt['gnn_inc_feat_f'])), inc_uu=int(bool(opt['gnn_inc_feat_uu'])),
        edge_index=ei.numpy().astype(np.int32), x_comp=data.x_comp.numpy(), f=data.f_tensor.numpy(), uu=data.uu_tensor.numpy(),
        target=data.x_phys.numpy(),
        wq=state[lay + 'lin_query.weight'].numpy(), bq=state[lay + 'lin_query.bias'].numpy(),
        wk=state[lay + 'lin_key.weight'].numpy(), bk=state[lay + 'lin_key.bias'].numpy(),
        **{k + '_f32': v for k, v in out32.items()}, **{k + '_f64': v for k, v in out64.items()},
        **{'band_' + k: np.float64(v) for k, v in band.items()})
    print(name, 'band', {k: '%.2e' % v for k, v in band.items()}, 'x_phys', out32['x_phys'].shape, 'max|f32-f64|', float(np.abs(out32['x_phys'] - out64['x_phys']).max()))
