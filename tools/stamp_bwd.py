#!/usr/bin/env python
"""Diagnostic: phase timeline of the two backward kernels from in-kernel s_memtime stamps (-DGADAPT_STAMPS build)."""
import ctypes as C, os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native, MeshDataset, collate, hot_path_opt, GNN
handle = C.CDLL(_native.LIB_PATH)
dev = torch.device('cuda:0')
n, B, Cc, L = int(os.environ.get("STAMP_N", "64")), int(os.environ.get("STAMP_B", "32")), int(os.environ.get("STAMP_C", "64")), 4
opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=Cc, num_layers=L, device='cuda:0', show_mesh_evol_plots='False')
ds = MeshDataset([n, n], B, seed=0); data = collate(ds.samples).to(dev)
model = GNN(ds, opt).to(dev).train()
model._enc_is_zero_pad = lambda: False      # dense x0: the last target launch (layer 0) is then the full kernel, like the middle layers
buf = torch.zeros(3 * 1024 * 32, dtype=torch.int64, device=dev)
for _ in range(2):
    model.zero_grad(); F.mse_loss(model(data), data.x_phys).backward()
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
model.zero_grad(); F.mse_loss(model(data), data.x_phys).backward()
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(None)
allb = buf.cpu().numpy().reshape(3, 1024, 32).astype(np.float64)
for name, s, names in (('backward_target (last launch = layer 0)', allb[1], ['start', 'commit', 'barrier+st', 'edge done', 'issue+barrier', 'dA done', 'gemm(b)', 'epilogue(b)']),
                       ('backward_source (last launch = layer 1)', allb[2], ['start', 'staged(b)', 'edge done', 'barrier', 'gemm(b)', 'epilogue', 'end(b)', '-'])):
    print(name)
    for tile in range(int(os.environ.get("STAMP_TILES", "4"))):
        seg = s[:, tile * 8:(tile + 1) * 8]
        ok = seg[:, 1] > 0
        if not ok.any():
            continue
        nst = 8 if 'target' in name else 7
        line = f"  tile {tile} (n={ok.sum()}):"
        for k in range(1, nst):
            d = seg[ok, k] - seg[ok, k - 1]
            line += f"  {names[k]} {np.median(d):.0f}"
        line += f"  | total {np.median(seg[ok, nst - 1] - seg[ok, 0]):.0f}"
        if tile < int(os.environ.get("STAMP_TILES", "4")) - 1:
            nxt = s[:, (tile + 1) * 8]
            ok2 = ok & (nxt > 0)
            if ok2.any():
                line += f"  | gap to next tile {np.median(nxt[ok2] - seg[ok2, nst - 1]):.0f}"
        print(line)
    if 'target' in name:
        ent, pro, end = s[:, 30], s[:, 29], s[:, 31]
        ok = (ent > 0) & (end > 0)
        first = s[:, 0]
        ntile = int(os.environ.get("STAMP_TILES", "4"))
        last = s[:, (ntile - 1) * 8 + 7]
        print(f"  workgroup: entry -> prologue done {np.median(pro[ok] - ent[ok]):.0f}  -> first tile start {np.median(first[ok] - pro[ok]):.0f}  "
              f"last stamped tile end -> slab row flushed {np.median(end[ok] - last[ok]):.0f}  | entry -> end {np.median(end[ok] - ent[ok]):.0f}")
        if ntile == 1:
            seq = [30, 8, 9, 10, 11, 12, 29, 0]
            lab = ['entry', 'chunk known', 'vector requests out', 'meta (scalar trip)', 'A fragments built', 'window commits', 'barrier', 'first tile loads in']
            print("  prologue: " + "  ".join(f"{lab[k]} +{np.median(s[ok, seq[k]] - s[ok, seq[k - 1]]):.0f}" for k in range(1, len(seq))))
        print(f"  launch: first entry -> last end {end[ok].max() - ent[ok].min():.0f} cycles; entry spread (max - min) {ent[ok].max() - ent[ok].min():.0f}; end spread {end[ok].max() - end[ok].min():.0f}")
