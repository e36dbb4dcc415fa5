"""HIP path vs the CPU oracle on the same seeded inputs (run on the MI355X: pytest -m gpu).

Tolerances (BASELINE.json north_star / BASELINE.md): predicted node coordinates within 1e-5
relative fp32; parameter gradients within 1e-4 relative.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import hip_model_like, make_case, rel_err

COORD_TOL = 1e-5
GRAD_TOL = 1e-4

CASES = [
    # mesh_dims, batch, hidden, layers, conv_type
    ((32,), 8, 8, 1, 'GRAND'),            # BASELINE config 1 (1-D plumbing)
    ((11, 11), 2, 8, 4, 'GRAND_plus'),    # the reference's shipped default (params.py:37,130-134)
    ((11, 11), 3, 4, 2, 'GRAND_plus'),
    ((15, 15), 2, 16, 3, 'GRAND_plus'),
    ((12, 12), 3, 32, 2, 'GRAND_plus'),
    ((32, 32), 2, 64, 4, 'GRAND_plus'),   # BASELINE config 2 shape (small batch)
    ((20, 20), 2, 128, 2, 'GRAND'),       # config 4 shape (hidden 128)
]


@pytest.mark.gpu
@pytest.mark.parametrize("mesh_dims,batch,hidden,layers,conv_type", CASES)
def test_forward_backward_parity(gpu_device, mesh_dims, batch, hidden, layers, conv_type):
    opt, ds, data, oracle = make_case(mesh_dims, batch, hidden, layers, conv_type)
    model = hip_model_like(oracle, ds, opt, gpu_device)
    tgt = data.x_phys if data.x_phys.dim() == 2 else data.x_phys.unsqueeze(-1)

    ref = oracle(data)
    loss_ref = F.mse_loss(ref, tgt)
    loss_ref.backward()

    out = model(data.clone().to(gpu_device))
    loss = F.mse_loss(out, tgt.to(gpu_device))
    loss.backward()
    torch.cuda.synchronize()

    norm, elem = rel_err(out, ref)
    assert norm <= COORD_TOL and elem <= COORD_TOL, f"x_phys rel err normwise {norm:.2e} elementwise {elem:.2e}"
    lo, lh = oracle.conv_layers[0], model.conv_layers[0]
    for name in ('lin_query.weight', 'lin_query.bias', 'lin_key.weight'):
        g_ref = dict(lo.named_parameters())[name].grad
        g_hip = dict(lh.named_parameters())[name].grad
        n, _ = rel_err(g_hip, g_ref)
        assert n <= GRAD_TOL, f"{name}.grad normwise rel err {n:.2e}"
    # d/d lin_key.bias vanishes analytically (softmax shift invariance); the oracle's is rounding noise
    gk_ref, gk_hip = lo.lin_key.bias.grad, lh.lin_key.bias.grad
    assert gk_hip.abs().max().item() == 0.0
    assert gk_ref.abs().max().item() <= 1e-4 * lo.lin_query.bias.grad.abs().max().item() + 1e-12
