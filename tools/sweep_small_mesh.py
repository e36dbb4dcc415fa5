#!/usr/bin/env python
"""Where does the one-launch small-mesh forward (csrc/gadapt_smallmesh.inc) beat the per-layer launches?  Evaluation forward as one
replayed hipGraph (inference.GraphedForward), both flows, a grid of mesh sizes / hidden sizes / batch sizes; per call: host-visible
latency (replay + synchronise) and the time between HIP events around the replay.

    python tools/sweep_small_mesh.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt          # noqa: E402
from g_adaptivity_amd.inference import GraphedForward                        # noqa: E402
import g_adaptivity_amd.functional as Fn                                     # noqa: E402

dev = torch.device('cuda:0')
Fn.small_forward_policy = lambda c, max_nodes: True                          # measure every size the kernel takes
print(f"{'mesh':>7s} {'batch':>5s} {'C':>3s} | per-layer: host us, event us | one-launch: host us, event us")
for dims in ([21], [11, 11], [15, 15], [19, 19], [23, 23], [31, 31]):
    for batch in (1, 16):
        for c in (8, 16, 32):
            opt = hot_path_opt(mesh_dims=dims, hidden_dim=c, num_layers=4, device=str(dev), conv_type='GRAND_plus', show_mesh_evol_plots='False')
            ds = MeshDataset(dims, batch, seed=0)
            data = collate(ds.samples).to(dev)
            torch.manual_seed(0)
            model = GNN(ds, opt).to(dev).eval()
            row = []
            for small in (False, True):
                Fn.SMALL_MESH_FORWARD = small
                g = next(iter(model._graphs.values()), None)
                with torch.no_grad():
                    runner = GraphedForward(model, data)
                    if small:
                        gr = next(iter(model._graphs.values()))
                        if not Fn.small_forward_fits(gr, gr.mesh_partition(data.batch), c):
                            row += [float('nan'), float('nan')]
                            continue
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for _ in range(20):
                        runner(sync=True)
                    host = dev_ms = 0.0
                    n = 200
                    for _ in range(n):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        e0.record(); runner(sync=False); e1.record()
                        torch.cuda.synchronize()
                        host += time.perf_counter() - t0
                        dev_ms += e0.elapsed_time(e1)
                row += [1e6 * host / n, 1e3 * dev_ms / n]
            print(f"{'x'.join(map(str, dims)):>7s} {batch:5d} {c:3d} | {row[0]:8.1f} {row[1]:8.1f} | {row[2]:8.1f} {row[3]:8.1f}", flush=True)
