#!/usr/bin/env python
"""Diagnostic: phase timeline of grand_fwd_kernel from in-kernel s_memtime stamps (needs a -DGADAPT_STAMPS build:
GADAPT_LIB=build/variants/lib_stamps.so python tools/stamp_fwd.py)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import _native, MeshDataset, collate, hot_path_opt, GNN
from g_adaptivity_amd import functional as Fn
lib = _native.lib()
handle = C.CDLL(_native.LIB_PATH)
dev = torch.device('cuda:0')
n, B, Cc, L = 64, 32, 64, 4
opt = hot_path_opt(mesh_dims=[n, n], hidden_dim=Cc, num_layers=L, device='cuda:0', show_mesh_evol_plots='False')
ds = MeshDataset([n, n], B, seed=0); data = collate(ds.samples).to(dev)
model = GNN(ds, opt).to(dev).train()
buf = torch.zeros(1024 * 32, dtype=torch.int64, device=dev)
for _ in range(3): model(data)
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
with torch.no_grad(): model(data)
torch.cuda.synchronize()
handle.gadapt_debug_set_stamp_buffer(None)
s = buf.cpu().numpy().reshape(1024, 32).astype(np.float64)
names = ['tile start', 'gemm done', 'P written(b2)', 'node0', 'node1', 'node2', 'node3', 'tile end(b3)']
print("compute wave 0, shader cycles; deltas between consecutive stamps (median / p10 / p90 over workgroups)")
print(f"first barrier wait (kernel start -> tile 0 start): median {np.median(s[:,0]-s[:,31]):.0f}")
for tile in range(2):
    seg = s[:, tile * 8:(tile + 1) * 8]
    ok = seg[:, 7] > 0
    print(f"tile {tile}: workgroups with data {ok.sum()}")
    for k in range(1, 8):
        d = seg[ok, k] - seg[ok, k - 1]
        print(f"  {names[k]:<16} delta median {np.median(d):>8.0f}  p10 {np.percentile(d, 10):>8.0f}  p90 {np.percentile(d, 90):>8.0f}")
    print(f"  whole tile       median {np.median(seg[ok,7]-seg[ok,0]):>8.0f}")
    l = s[:, 16 + tile * 4:16 + tile * 4 + 3]
    okl = l[:, 2] > 0
    if okl.any():
        print(f"  loader: fetch+mid barriers+store {np.median(l[okl,1]-l[okl,0]):>8.0f}   wait at end barrier {np.median(l[okl,2]-l[okl,1]):>8.0f}")
okk = s[:, 15] > 0
print("kernel per-WG span (start -> last tile end), median:", np.median(s[okk, 15] - s[okk, 31]))
# residency: per XCD group (blockIdx % 8), how many workgroups start only after another one has ended?
for x in range(2):
    g = s[x::8]
    st, en = g[:, 31], g[:, 15]
    t0 = st.min()
    order = np.argsort(st)
    print(f"xcd-group {x}: starts (cycles after first) p0/p25/p50/p75/p100:",
          [int(v) for v in np.percentile(st - t0, [0, 25, 50, 75, 100])],
          " ends p0/p50/p100:", [int(v) for v in np.percentile(en - t0, [0, 50, 100])],
          " started after first end:", int((st > en.min()).sum()), "of", len(st))

raw = buf.cpu().numpy().reshape(1024, 32)
xcc = (raw[:, 30] & 0xF).astype(int)
hwid = (raw[:, 30] >> 32).astype(np.int64)
cu = ((hwid >> 8) & 0xF); se = ((hwid >> 13) & 0x7); sh = (hwid >> 12) & 1
print("xcc of blocks 0..31:", xcc[:32].tolist())
print("blocks per xcc:", np.bincount(xcc, minlength=8).tolist())
import collections
per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("distinct (xcc,se,sh,cu):", len(per_cu), " workgroups per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
for x in range(8):
    m = xcc == x
    st, en = s[m, 31], s[m, 15]
    t0 = st.min()
    print(f"xcc {x}: n={m.sum()} start spread p50/p100 {int(np.percentile(st-t0,50))}/{int((st-t0).max())}  end p50/p100 {int(np.percentile(en-t0,50))}/{int((en-t0).max())}  started after first end: {int((st > en.min()).sum())}")
groups = collections.defaultdict(list)
for b in range(1024):
    groups[(xcc[b], se[b], sh[b], cu[b])].append(b)
seq = collections.Counter(); spans = []
for k, bl in groups.items():
    st = np.array([s[b, 31] for b in bl]); en = np.array([s[b, 15] for b in bl])
    t0 = st.min()
    spans.append(en.max() - t0)
    # number of workgroups that started after some other workgroup on this CU ended
    seq[(len(bl), int((st > en.min()).sum()))] += 1
print("per-CU (n_wgs, n_started_after_first_end):", sorted(seq.items()))
print("per-CU busy span cycles: median %d  p90 %d  max %d" % (np.median(spans), np.percentile(spans, 90), max(spans)))
k0 = list(groups)[0]
for b in groups[k0]:
    print("  CU", k0, "block", b, "start", int(s[b,31]-min(s[x,31] for x in groups[k0])), "end", int(s[b,15]-min(s[x,31] for x in groups[k0])))
