#!/usr/bin/env python
"""Diagnostic: phase timeline of grand_fwd_kernel from in-kernel s_memtime stamps (needs a -DGADAPT_STAMPS build:
make EXTRA=-DGADAPT_STAMPS LIB=variants/stamps.so OBJDIR=build/obj_stamps; GADAPT_LIB=variants/stamps.so python tools/stamp_fwd.py)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native, MeshDataset, collate, hot_path_opt, GNN
handle = C.CDLL(_native.LIB_PATH)
dev = torch.device('cuda:0')
n, B, Cc, L = int(os.environ.get("STAMP_N", "64")), int(os.environ.get("STAMP_B", "32")), int(os.environ.get("STAMP_C", "64")), 4
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
names = ['start', 'gemm', 'commit(b)+issue', 'slot0', 'slot1', 'slot2', 'slot3', 'end(b)']
print("forward kernel (last layer launch), compute wave 0, shader cycles between consecutive stamps (median over workgroups)")
for tile in range(4):
    seg = s[:, tile * 8:(tile + 1) * 8]
    ok = (seg[:, 7] > 0) & (seg[:, 0] > 0)
    if not ok.any():
        continue
    line = f"  tile {tile} (n={ok.sum()}):"
    prev = seg[ok, 0]
    for k in range(1, 8):
        cur = seg[ok, k]
        valid = cur > 0
        if not valid.any():
            continue
        d = np.where(valid, cur - prev, np.nan)
        line += f"  {names[k]} {np.nanmedian(d):.0f}"
        prev = np.where(valid, cur, prev)
    line += f"  | total {np.median(seg[ok, 7] - seg[ok, 0]):.0f}"
    print(line)
