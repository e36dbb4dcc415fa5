#!/usr/bin/env python
"""GPU box: evaluation forward (4 layers, hidden 64, one hipGraph replay per call) of 64x64 / 32x32 mesh batches on the tiled kernels and on
the wide forward (four-wave workgroups up to graph.WIDE_HALF_MAX_NODES nodes, eight-wave above): where should graph.WIDE_MIN_NODES sit?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt
from g_adaptivity_amd import graph as graph_mod
from g_adaptivity_amd.inference import GraphedForward
import g_adaptivity_amd.functional as Fn

dev = torch.device('cuda:0')
Fn.SMALL_MESH_FORWARD = False
for mesh in (64, 32):
    for B in ((1, 2, 3, 4, 6, 8) if mesh == 64 else (4, 8, 16, 24, 32)):
        row = []
        for label, mn, hm in (('tiled', 10**9, 0), ('wide, four waves', 0, 10**9), ('wide, eight waves', 0, 0)):
            graph_mod.WIDE_MIN_NODES, graph_mod.WIDE_HALF_MAX_NODES = mn, hm
            opt = hot_path_opt(mesh_dims=[mesh, mesh], hidden_dim=64, num_layers=4, device=str(dev), show_mesh_evol_plots='False')
            ds = MeshDataset([mesh, mesh], B, seed=0)
            data = collate(ds.samples).to(dev)
            torch.manual_seed(0)
            model = GNN(ds, opt).to(dev).eval()
            gf = GraphedForward(model, data)
            for _ in range(20):
                gf(sync=False)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(200):
                    gf(sync=False)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 200)
            row.append(f"{label} {best * 1e6:.1f} us")
        print(f"{mesh}x{mesh} batch {B} ({B * mesh * mesh} nodes): " + " | ".join(row), flush=True)
