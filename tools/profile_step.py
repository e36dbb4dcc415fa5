#!/usr/bin/env python
"""Runs K eager training steps of a bench workload; meant to sit behind `rocprofv3 ... -- python3 tools/profile_step.py`."""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS                                       # noqa: E402
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt   # noqa: E402
from g_adaptivity_amd.optim import FlatAdam                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='poisson2d_64x64_b32_L4_C64')
ap.add_argument('--steps', type=int, default=5)
a = ap.parse_args()
w = WORKLOADS[a.workload]
dev = torch.device('cuda:0')
opt = hot_path_opt(mesh_dims=[w['n'], w['n']], hidden_dim=w['hidden'], num_layers=w['layers'], conv_type=w['conv'],
                   gnn_inc_feat_f=w['f'], gnn_inc_feat_uu=w['uu'], device=str(dev), show_mesh_evol_plots='False')
ds = MeshDataset([w['n'], w['n']], w['batch'], seed=0)
data = collate(ds.samples).to(dev)
torch.manual_seed(0)
model = GNN(ds, opt).to(dev).train()
optim = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
from g_adaptivity_amd import mse_loss, unit_gradient               # noqa: E402
from g_adaptivity_amd.training import FusedIteration               # noqa: E402


def autograd_step():
    optim.zero_grad()
    mse_loss(model(data), data.x_phys).backward(gradient=unit_gradient(dev))
    optim.step()


autograd_step()                                                    # lays the optimizer's bucket out
fused = None
if FusedIteration.eligible(model, optim, mse_loss, data, 'x_phys') is None:      # the route bench.py's step takes
    fused = FusedIteration(model, optim, mse_loss, data, 'x_phys')
    fused.refresh_coeffs()
for _ in range(a.steps):
    if fused is not None:
        fused.run()
    else:
        autograd_step()
torch.cuda.synchronize()
print("profiled", a.steps, "steps of", a.workload, "(fused iteration)" if fused is not None else "(autograd iteration)")
