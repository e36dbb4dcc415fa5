#!/usr/bin/env python
"""Diagnostic: phase timeline of the wide forward kernel from in-kernel s_memtime stamps (needs a -DGADAPT_STAMPS build:
make EXTRA=-DGADAPT_STAMPS LIB=variants/stamps.so OBJDIR=build/obj_stamps; GADAPT_LIB=variants/stamps.so python tools/stamp_wide.py)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native, MeshDataset, collate, hot_path_opt, GNN
handle = C.CDLL(_native.LIB_PATH)
dev = torch.device('cuda:0')
n, B, Cc, L = int(os.environ.get("STAMP_N", "64")), int(os.environ.get("STAMP_B", "32")), 64, 4
opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=Cc, num_layers=L, device='cuda:0', show_mesh_evol_plots='False')
ds = MeshDataset([n, n], B, seed=0); data = collate(ds.samples).to(dev)
model = GNN(ds, opt).to(dev).train()
buf = torch.zeros(3 * 1024 * 32, dtype=torch.int64, device=dev)
for _ in range(3):
    with torch.no_grad(): model(data)
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
with torch.no_grad(): model(data)
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().reshape(3, 1024, 32)[0].astype(np.float64)
ok = s[:, 0] > 0
s = s[ok]
names = {0: 'start', 1: 'frags+prologue issued', 2: 'prologue committed(b)', 3: 'gemm', 4: 'edges', 5: 'finalize', 6: 'barrier', 7: 'commit+stores', 8: 'barrier',
         9: 'gemm', 10: 'edges', 11: 'finalize', 12: 'barrier', 13: 'commit+stores', 14: 'barrier'}
print(f"wide forward kernel (last layer launch), wave 0 of {len(s)} workgroups: shader cycles between consecutive stamps (median, p10, p90)")
prev = s[:, 0]
for k in range(1, 15):
    cur = s[:, k]
    v = cur > 0
    if not v.any():
        continue
    d = (cur - prev)[v]
    print(f"  {k:2d} {names[k]:28s} {np.median(d):8.0f} {np.percentile(d, 10):8.0f} {np.percentile(d, 90):8.0f}   since start {np.median((cur - s[:, 0])[v]):8.0f}")
    prev = np.where(v, cur, prev)
rt = (s[:, 31] - s[:, 30])
print(f"  per-workgroup wall (s_memrealtime, 100 MHz ticks): median {np.median(rt):.0f} = {np.median(rt) / 100:.2f} us; kernel span {(s[:, 31].max() - s[:, 30].min()) / 100:.2f} us")
print(f"  start skew: {(s[:, 30].max() - s[:, 30].min()) / 100:.2f} us; shader clock ~ {np.median((s[:, 14] - s[:, 0]) / np.maximum(rt, 1)) * 100 / 1e3:.2f} GHz")
