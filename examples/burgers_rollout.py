#!/usr/bin/env python
"""Rollout driver in the shape of the reference's Burgers evaluation (`src/utils_eval_Burgers.py:262-300`): one mesh,
the model re-invoked every time step with the evolved field in `data.uu_tensor`, mesh time accumulated around the call.

The PDE step between two calls is Firedrake/FEM work and out of scope here; a travelling, steepening pulse stands in for
it so that the field - and therefore the predicted mesh - changes every step.

    python examples/burgers_rollout.py --mesh 21 --steps 20            # 1-D, the reference's Burgers size
    python examples/burgers_rollout.py --mesh 64 --dim 2 --steps 20
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt          # noqa: E402
from g_adaptivity_amd.inference import GraphedForward                        # noqa: E402


def pulse(coords: torch.Tensor, t: float) -> torch.Tensor:
    """Field on the computational nodes at time t: a pulse that travels and sharpens (stand-in for the Burgers solve)."""
    x = coords[:, 0] if coords.dim() == 2 else coords
    centre, width = 0.25 + 0.5 * t, 0.15 / (1.0 + 3.0 * t)
    u = torch.exp(-((x - centre) / width) ** 2)
    if coords.dim() == 2 and coords.shape[1] == 2:
        u = u * torch.exp(-((coords[:, 1] - 0.5) / 0.3) ** 2)
    return u


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mesh', type=int, default=21)
    ap.add_argument('--dim', type=int, default=1, choices=[1, 2])
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--hidden', type=int, default=64)
    ap.add_argument('--layers', type=int, default=4)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    dims = [args.mesh] * args.dim
    opt = hot_path_opt(mesh_dims=dims, hidden_dim=args.hidden, num_layers=args.layers, device=str(dev),
                       conv_type='GRAND' if args.dim == 1 else 'GRAND_plus', gnn_inc_feat_f=False,
                       show_mesh_evol_plots='False')
    ds = MeshDataset(dims, 1, seed=0)
    data = collate(ds.samples).to(dev)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(dev).eval()
    coords = data.x_comp if data.x_comp.dim() == 2 else data.x_comp.unsqueeze(-1)

    results = {}
    for mode in ('eager', 'hipgraph'):
        runner = GraphedForward(model, data) if mode == 'hipgraph' else None
        mesh_time, outs = 0.0, []
        with torch.no_grad():
            (runner(data) if runner is not None else model(data))                        # first call: graph build, kernel load
            for k in range(args.steps):
                data.uu_tensor = pulse(coords, k / args.steps).to(data.uu_tensor.dtype)   # the "PDE step"
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = runner(data) if runner is not None else model(data)
                mesh_time += time.perf_counter() - t0
                outs.append(out.clone())
        results[mode] = (mesh_time / args.steps, torch.stack(outs))
        print(f"{mode:9s}: {1e6 * mesh_time / args.steps:8.1f} us per model call ({args.steps} calls, "
              f"{data.x_comp.shape[0]} nodes, graphs built: {len(model._graphs)})")
    diff = (results['eager'][1] - results['hipgraph'][1]).abs().max().item()
    print(f"max |eager - hipgraph| over the rollout: {diff:.3e}")


if __name__ == '__main__':
    main()
