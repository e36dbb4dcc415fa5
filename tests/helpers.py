"""Shared builders for the parity tests: same seeded inputs for the HIP path and the oracle."""
import copy

import torch

from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt
from oracle.pyg_restatement import OracleGNN


def make_case(mesh_dims, batch, hidden, layers, conv_type='GRAND_plus', seed=0, **opt_over):
    opt = hot_path_opt(mesh_dims=list(mesh_dims), hidden_dim=hidden, num_layers=layers, conv_type=conv_type, **opt_over)
    ds = MeshDataset(mesh_dims, batch, seed=seed)
    data = collate(ds.samples)
    torch.manual_seed(seed)
    oracle = OracleGNN(ds, dict(opt))
    return opt, ds, data, oracle


def hip_model_like(oracle, ds, opt, device):
    o = dict(opt)
    o['device'] = str(device)
    model = GNN(ds, o).to(device)
    missing, unexpected = model.load_state_dict(copy.deepcopy(oracle.state_dict()), strict=True)
    return model


def rel_err(a, b):
    """max |a-b| / max |b| (normwise) and the worst elementwise relative error where |b| > 1e-3 max|b|."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = b.abs().max().clamp_min(1e-300)
    norm = ((a - b).abs().max() / scale).item()
    big = b.abs() > 1e-3 * scale
    elem = ((a - b).abs()[big] / b.abs()[big]).max().item() if big.any() else 0.0
    return norm, elem


def oracle_fp64_twin(oracle, ds, opt, data, tgt):
    """Same weights and inputs in fp64, forward + mse backward: bounds the fp32 oracle's own rounding."""
    import torch.nn.functional as F
    o64 = OracleGNN(ds, dict(opt)).double()
    o64.load_state_dict({k: v.double() for k, v in oracle.state_dict().items()})
    d64 = data.clone()
    for k in ('x_comp', 'f_tensor', 'uu_tensor'):
        setattr(d64, k, getattr(d64, k).double())
    ref64 = o64(d64)
    F.mse_loss(ref64, tgt.double()).backward()
    return o64, ref64


def permute_edges(data, seed):
    """The same batch with its edge list in another order (edge_index and the three per-edge masks permuted together).  The
    reference builds edge_index from a Python set of tuples (`src/data.py:430-441`), so ITS edge order - and with it the fp32
    summation order of every scatter - changes from run to run."""
    d = data.clone()
    perm = torch.randperm(d.edge_index.shape[1], generator=torch.Generator().manual_seed(seed))
    d.edge_index = d.edge_index[:, perm]
    for m in ('to_boundary_edge_mask', 'to_corner_nodes_mask', 'diff_boundary_edges_mask'):
        if getattr(d, m, None) is not None:
            setattr(d, m, getattr(d, m)[perm])
    return d


def edge_order_band(oracle, o64, data, tgt, k=4):
    """{parameter name: max relative error against the fp64 twin} of the fp32 oracle over `k` edge orders of the same batch:
    the run-to-run band of the reference's own fp32 path (see permute_edges).  Parameters whose fp64 gradient is None are skipped."""
    import torch.nn.functional as F
    probe = copy.deepcopy(oracle)
    g64 = {n: p.grad for n, p in o64.named_parameters() if p.grad is not None}
    band = {n: 0.0 for n in g64}
    for seed in range(1, k + 1):
        probe.zero_grad(set_to_none=True)
        F.mse_loss(probe(permute_edges(data, seed)), tgt).backward()
        for n, p in probe.named_parameters():
            if n in band and p.grad is not None:
                band[n] = max(band[n], rel_err(p.grad, g64[n])[0])
    return band
