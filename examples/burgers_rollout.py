#!/usr/bin/env python
"""Rollout driver in the shape of the reference's Burgers evaluation (`src/utils_eval_Burgers.py:262-300`): one mesh,
the model re-invoked every time step with the evolved field in `data.uu_tensor`, mesh time accumulated around the call.

The PDE step between two calls is Firedrake/FEM work and out of scope here; a travelling, steepening pulse stands in for
it so that the field - and therefore the predicted mesh - changes every step.

    python examples/burgers_rollout.py --mesh 21 --steps 20            # 1-D, the reference's Burgers size
    python examples/burgers_rollout.py --mesh 64 --dim 2 --steps 20
    python examples/burgers_rollout.py --mesh 11 --dim 2 --hidden 8 --steps 50    # the reference's shipped 2-D size: one-launch forward

Meshes that fit a workgroup's LDS at hidden <= 32 run the ONE-LAUNCH forward (csrc/gadapt_smallmesh.inc); the script times that and
the per-layer launches (`functional.SMALL_MESH_FORWARD = False`) side by side: host-visible latency of a call (copy of the new field,
graph replay, synchronisation) and the device time of the replay (HIP events).
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

    import g_adaptivity_amd.functional as Fn
    results = {}
    flows = [('per-layer launches', False)]
    if args.hidden <= 32:
        flows.append(('one-launch forward', True))
    for flow, small in flows:
        Fn.SMALL_MESH_FORWARD = small
        for mode in ('eager', 'hipgraph'):
            runner = GraphedForward(model, data) if mode == 'hipgraph' else None
            mesh_time, dev_ms, outs = 0.0, 0.0, []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.no_grad():
                (runner(data) if runner is not None else model(data))                        # first call: graph build, kernel load
                for k in range(args.steps):                                                   # host-visible latency, as the reference stamps it
                    data.uu_tensor = pulse(coords, k / args.steps).to(data.uu_tensor.dtype)   # the "PDE step"
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = runner(data) if runner is not None else model(data)                 # (returns after a stream synchronisation)
                    mesh_time += time.perf_counter() - t0
                    outs.append(out.clone())
                for k in range(args.steps):                                                   # the same calls between two HIP events
                    data.uu_tensor = pulse(coords, k / args.steps).to(data.uu_tensor.dtype)
                    torch.cuda.synchronize()
                    e0.record()
                    out = runner(data, sync=False) if runner is not None else model(data)
                    e1.record()
                    torch.cuda.synchronize()
                    dev_ms += e0.elapsed_time(e1)
            results[(flow, mode)] = (mesh_time / args.steps, torch.stack(outs))
            how = '' if runner is None else (' [one kernel, launched directly]' if runner.direct else ' [one C-ABI call, issued]' if runner.issued else ' [replayed capture]')
            print(f"{flow:19s} {mode:9s}{how}: {1e6 * mesh_time / args.steps:8.1f} us host-visible per model call, {1e3 * dev_ms / args.steps:7.1f} us between "
                  f"HIP events ({args.steps} calls, {data.x_comp.shape[0]} nodes, graphs built: {len(model._graphs)})")
        diff = (results[(flow, 'eager')][1] - results[(flow, 'hipgraph')][1]).abs().max().item()
        print(f"{flow:19s} max |eager - hipgraph| over the rollout: {diff:.3e}")
    if len(flows) == 2:
        a, b = results[(flows[0][0], 'hipgraph')][1], results[(flows[1][0], 'hipgraph')][1]
        print(f"max |per-layer - one-launch| / max |out|: {((a - b).abs().max() / a.abs().max()).item():.3e}")


if __name__ == '__main__':
    main()
