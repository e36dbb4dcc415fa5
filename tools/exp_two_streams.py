#!/usr/bin/env python
"""Experiment (GPU box): the metric step as TWO independent half-batch chains replayed concurrently on two streams, against one
full-batch chain.  Meshes of a batch are independent, so the split is exact; the question is whether two co-resident, out-of-phase
chains overlap their memory and compute phases better than one chain whose workgroups run in lockstep.
    python tools/exp_two_streams.py [halves=2] [steps=200]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, mse_loss, unit_gradient
from g_adaptivity_amd.optim import FlatAdam

halves = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device('cuda:0')
torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
B = int(os.environ.get('EXP_B', 32))
EAGER = os.environ.get('EXP_MODE', 'graph') == 'eager'
TORCH_LOSS = os.environ.get('EXP_LOSS', 'native') == 'torch'
SKIP_ONE = os.environ.get('EXP_SKIP_ONE') == '1'


def chain(n_meshes, seed):
    opt = hot_path_opt(mesh_dims=[64, 64], hidden_dim=64, num_layers=4, device=str(dev), show_mesh_evol_plots='False')
    ds = MeshDataset([64, 64], n_meshes, seed=seed)
    data = collate(ds.samples).to(dev)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(dev).train()
    optim = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
    root = unit_gradient(dev)
    side = torch.cuda.Stream()

    def it():
        optim.zero_grad()
        if TORCH_LOSS:
            torch.nn.functional.mse_loss(model(data), data.x_phys).backward()
        else:
            mse_loss(model(data), data.x_phys).backward(gradient=root)
        optim.step()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            it()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if EAGER:
        class G:
            replay = staticmethod(it)
        return G, side
    g = torch.cuda.CUDAGraph()
    optim.zero_grad()
    with torch.cuda.graph(g, stream=side):
        it()
    return g, side


def timed(graphs):
    for _ in range(20):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


if not SKIP_ONE:
    one = [chain(B, 0)]
    t1 = timed(one)
    print(f"one chain, batch {B}: {t1 * 1e3:.4f} ms/step, {B / t1:,.0f} meshes/s", flush=True)
    del one
parts = [chain(B // halves, k) for k in range(halves)]
t2 = timed(parts)
print(f"{halves} concurrent chains, batch {B // halves} each: {t2 * 1e3:.4f} ms per {B} meshes, {B / t2:,.0f} meshes/s", flush=True)
t3 = timed(parts[:1])
print(f"one chain, batch {B // halves}: {t3 * 1e3:.4f} ms/step, {B // halves / t3:,.0f} meshes/s", flush=True)
