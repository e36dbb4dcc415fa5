#!/usr/bin/env python
"""Diagnostic: phase timeline of the wide backward target kernel from in-kernel s_memtime stamps (-DGADAPT_STAMPS build:
GADAPT_LIB=build/diag/lib_stamps.so python tools/stamp_wide_bwd.py)."""
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
    model.zero_grad(); F.mse_loss(model(data), data.x_phys).backward()
for _ in range(3): step()
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
step()
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().reshape(3, 1024, 32)[1].astype(np.float64)
s = s[s[:, 0] > 0]
names = {1: 'prologue(b)', 2: 'idx/offsets', 3: 'pass1', 4: 'pass2+edge_ws', 5: 'dA', 6: 'dP A', 7: 'touch+next loads', 8: 'barrier', 9: 'out+commit',
         10: 'barrier+idx', 11: 'pass1', 12: 'pass2+edge_ws', 13: 'dA', 14: 'dP A', 15: 'touch+next loads', 16: 'barrier', 17: 'out+commit', 26: 'barrier', 27: 'flush'}
print(f"wide target kernel (layer-0 launch), wave 0 of {len(s)} workgroups: shader cycles between consecutive stamps (median, p10, p90)")
prev = s[:, 0]
for k in sorted(names):
    cur = s[:, k]
    v = cur > 0
    if not v.any():
        continue
    d = (cur - prev)[v]
    print(f"  {k:2d} {names[k]:22s} {np.median(d):8.0f} {np.percentile(d, 10):8.0f} {np.percentile(d, 90):8.0f}   since start {np.median((cur - s[:, 0])[v]):8.0f}")
    prev = np.where(v, cur, prev)
rt = (s[:, 31] - s[:, 30])
print(f"  per-workgroup wall: median {np.median(rt) / 100:.2f} us; kernel span {(s[:, 31].max() - s[:, 30].min()) / 100:.2f} us")
