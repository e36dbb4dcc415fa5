#!/usr/bin/env python
"""Diagnostic: phase timeline of the wide backward main kernel (csrc/gadapt_wide_bwd.inc) from in-kernel s_memtime stamps.
Needs a -DGADAPT_STAMPS build:  GADAPT_LIB=build/diag/lib_stamps.so python tools/stamp_wide_bwd.py"""
import ctypes as C, os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native, MeshDataset, collate, hot_path_opt, GNN
handle = C.CDLL(_native.LIB_PATH)
dev = torch.device('cuda:0')
n, B, Cc, L = 64, 32, 64, 4
opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=Cc, num_layers=L, device='cuda:0', show_mesh_evol_plots='False')
ds = MeshDataset([n, n], B, seed=0); data = collate(ds.samples).to(dev)
model = GNN(ds, opt).to(dev).train()
buf = torch.zeros(3 * 1024 * 32, dtype=torch.int64, device=dev)
def step():
    model.zero_grad()
    F.mse_loss(model(data), data.x_phys).backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
step()
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().reshape(3, 1024, 32)[0].astype(np.float64)     # region 0: shared with the forward kernels, which ran earlier in the step
ok = s[:, 0] > 0
s = s[ok]
names = {0: 'start', 1: 'prologue (window + fragments)'}
for kk in range(4):
    names.update({2 + 6 * kk: 'in-row walk', 3 + 6 * kk: 'out-row walk', 4 + 6 * kk: 'projections', 5 + 6 * kk: 'staged stores + requests issued', 6 + 6 * kk: 'dA',
                  7 + 6 * kk: 'wait + barrier + commit + barrier'})
print(f"wide backward main kernel (LAST such launch of the step), wave 0 of {len(s)} workgroups: shader cycles between consecutive stamps (median, p10, p90)")
prev = s[:, 0]
for k in range(1, 30):
    cur = s[:, k]
    v = cur > 0
    if not v.any():
        continue
    d = (cur - prev)[v]
    print(f"  {k:2d} {names.get(k, ''):32s} {np.median(d):8.0f} {np.percentile(d, 10):8.0f} {np.percentile(d, 90):8.0f}   since start {np.median((cur - s[:, 0])[v]):8.0f}")
    prev = np.where(v, cur, prev)
