#!/usr/bin/env python
"""Diagnostic (GPU box): which parts of a step of the generic-primitive convs (GAT_plus / GAT / GCN) survive hipGraph capture.
    python tools/capture_probe_convs.py <conv_type> <what>     what: fwd | fwdbwd | step"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, mse_loss   # noqa: E402
from g_adaptivity_amd.optim import FlatAdam                                      # noqa: E402

conv, what = sys.argv[1], sys.argv[2]
mesh_n, batch = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16, 4)
dev = torch.device('cuda:0')
opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=64, num_layers=int(os.environ.get('PROBE_LAYERS', 2)), conv_type=conv, device=str(dev), show_mesh_evol_plots='False')
ds = MeshDataset([mesh_n, mesh_n], batch, seed=0)
data = collate(ds.samples).to(dev)
torch.manual_seed(0)
model = GNN(ds, opt).to(dev)
optim = FlatAdam(model.parameters(), lr=1e-3, capturable=True)


def run():
    if what == 'fwd':
        with torch.no_grad():
            return model(data)
    optim.zero_grad()
    loss = mse_loss(model(data), data.x_phys)
    loss.backward()
    if what == 'step':
        optim.step()
    return loss


model.train(what != 'fwd')
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print('capturing', conv, what, flush=True)
with torch.cuda.graph(g, stream=side):
    out = run()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('replayed ok', float(out.float().sum()), flush=True)
