#!/usr/bin/env python
"""Diagnostic (GPU box): which parts of a step of the generic-primitive convs (GAT_plus / GAT / GCN) survive hipGraph capture.
    python tools/capture_probe_convs.py <conv_type> <what> [mesh_n batch]     what: fwd | fwdbwd | step
Environment knobs that move the flow towards bench.py's (where GAT_plus crashed inside hipStreamEndCapture):
    PROBE_LAYERS (2)  PROBE_EAGER_FIRST (0: eager steps on the default stream before anything else)  PROBE_WARM (3: side-stream
    warm-up runs)  PROBE_ROOT (0: 1 = loss.backward(gradient=unit_gradient))  PROBE_FEATS (1: 0 = no f / uu features)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, mse_loss, unit_gradient   # noqa: E402
from g_adaptivity_amd.optim import FlatAdam                                      # noqa: E402

conv, what = sys.argv[1], sys.argv[2]
mesh_n, batch = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (16, 4)
E = lambda k, d: int(os.environ.get(k, d))                                       # noqa: E731
dev = torch.device('cuda:0')
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
opt = hot_path_opt(mesh_dims=[mesh_n, mesh_n], hidden_dim=64, num_layers=E('PROBE_LAYERS', 2), conv_type=conv, device=str(dev),
                   show_mesh_evol_plots='False', loss_type='mesh_loss')
ds = MeshDataset([mesh_n, mesh_n], batch, seed=0)
data = collate(ds.samples).to(dev)
torch.manual_seed(0)
model = GNN(ds, opt).to(dev)
optim = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
root = unit_gradient(dev) if E('PROBE_ROOT', 0) else None


def run():
    if what == 'fwd':
        with torch.no_grad():
            return model(data)
    optim.zero_grad()
    loss = mse_loss(model(data), data.x_phys)
    if root is None:
        loss.backward()
    else:
        loss.backward(gradient=root)
    if what == 'step':
        optim.step()
    return loss


model.train(what != 'fwd')
for _ in range(E('PROBE_EAGER_FIRST', 0)):
    run()
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(E('PROBE_WARM', 3)):
        run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
if what != 'fwd' and E('PROBE_ZERO_OUTSIDE', 0):
    optim.zero_grad()
print('capturing', conv, what, flush=True)
with torch.cuda.graph(g, stream=side):
    out = run()
print('captured', flush=True)
g.replay(); torch.cuda.synchronize()
print('replayed ok', float(out.float().sum()), flush=True)
