#!/usr/bin/env python
"""Two-rank data-parallel step on real launches (tests/test_gpu_callers.py::test_two_rank_step_equals_full_batch_step).

Launched with `python -m torch.distributed.run --nproc-per-node 2 tools/ddp_step_check.py`.  Every rank runs forward +
backward on ITS shard of the batch through the HIP kernels, `FlatAdam.step()` all-reduces the flat gradient bucket and
takes the fused Adam step.  Rank 0 also runs the same steps single-handedly on the FULL batch (an optimizer with
`data_parallel=False`) and prints one JSON line with the differences.  The ranks may share one GPU (the test box has one);
the collective then runs over gloo (RCCL needs one GPU per rank) - set GADAPT_DDP_BACKEND=nccl on a multi-GPU box.
"""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local = int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    backend = os.environ.get('GADAPT_DDP_BACKEND', 'gloo' if torch.cuda.device_count() < world else 'nccl')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=dev)
    else:
        dist.init_process_group(backend)

    from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, mse_loss
    from g_adaptivity_amd.optim import FlatAdam, shard_range

    mesh, n_meshes, hidden, layers, steps = [16, 16], 8, 64, 3, 3
    opt = hot_path_opt(mesh_dims=mesh, hidden_dim=hidden, num_layers=layers, device=str(dev), lr=1e-3, show_mesh_evol_plots='False')
    ds = MeshDataset(mesh, n_meshes, seed=0)                      # the same dataset on every rank; each takes its shard
    lo, hi = shard_range(n_meshes, rank, world)
    shard = collate(ds.samples[lo:hi]).to(dev)

    torch.manual_seed(0)                                          # identical replicas
    model = GNN(ds, opt).to(dev).train()
    graphed = os.environ.get('GADAPT_DDP_GRAPHED') == '1'       # the step as GraphedTrainStep: forward + loss + backward replayed,
    optim = FlatAdam(model.parameters(), lr=opt['lr'], capturable=graphed)   # all-reduce + Adam eager after each replay (gloo cannot be captured)
    first_sum = None
    if graphed:
        from g_adaptivity_amd import GraphedTrainStep
        step = GraphedTrainStep(model, optim, capture_optimizer=backend == 'nccl' and os.environ.get('GADAPT_DDP_CAPTURE_OPT') == '1')
    for _ in range(steps):
        if graphed:
            step(shard)
        else:
            optim.zero_grad()
            mse_loss(model(shard), shard.x_phys).backward()
            optim.step()                                          # all-reduce (SUM) + Adam with the 1/world scale folded in
        if first_sum is None:
            first_sum = optim.grad_bucket.detach().clone()        # the reduced bucket of step 1
    torch.cuda.synchronize()

    mine = optim.bucket.detach().clone()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    identical = all(torch.equal(gathered[0], g) for g in gathered)

    if rank == 0:
        full = collate(ds.samples).to(dev)
        torch.manual_seed(0)
        ref = GNN(ds, opt).to(dev).train()
        ref_optim = FlatAdam(ref.parameters(), lr=opt['lr'], data_parallel=False)
        ref_first = None
        for _ in range(steps):
            ref_optim.zero_grad()
            mse_loss(ref(full), full.x_phys).backward()
            ref_optim.step()
            if ref_first is None:
                ref_first = ref_optim.grad_bucket.detach().clone()
        torch.cuda.synchronize()
        avg = first_sum / world
        gerr = ((avg - ref_first).abs().max() / ref_first.abs().max()).item()
        pdiff = (optim.bucket - ref_optim.bucket).abs().max().item()
        print(json.dumps({'world': world, 'backend': backend, 'bucket_floats': int(optim.bucket.numel()), 'grad_rel_err': gerr,
                          'param_max_abs_diff': pdiff, 'ranks_identical': bool(identical), 'steps': steps, 'graphed': graphed}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
