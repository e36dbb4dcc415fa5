#!/usr/bin/env python
"""The reference's training loop (`src/run_GNN.py:66-154`, loss_type='mesh_loss') on the MI355X-native model.

Same shape as the reference: DataLoader -> for epoch / for batch: zero_grad, model(data), loss_fn(out, data.x_phys),
backward, optimizer.step; best-state tracking.  Differences: synthetic Firedrake-free dataset (mesh_graph.MeshDataset),
`GNN` from g_adaptivity_amd, and the flat-bucket Adam (drop-in for torch.optim.Adam).

    python examples/train_mesh_loss.py --mesh 32 --num_train 64 --batch_size 16 --epochs 3
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, DeviceMeshLoader, GraphedTrainStep, MeshDataset, MeshLoader, hot_path_opt, l1_loss, mse_loss, unit_gradient   # noqa: E402
from g_adaptivity_amd.optim import FlatAdam                                      # noqa: E402


def main(opt, dataset, log=print):
    """Returns (model, per-epoch losses, steady-state meshes/s of the last epoch).

    opt['graphed'] (default True): the iteration is a `GraphedTrainStep` - captured once per batch size, replayed on every new
    batch, the loader gathering straight into the captured step's input buffers.  False: the same iteration as eager launches,
    line by line the reference's loop."""
    shuffle = not opt.get('overfit_num')
    graphed = opt.get('graphed', True) and opt.get('device_loader', True) and opt.get('native_loss', True)
    torch.manual_seed(opt.get("seed", 0))             # reproducible initial weights (the reference draws a random seed: run_pipeline.py:52-54)
    model = GNN(dataset, opt).to(opt["device"])
    if opt.get('native_loss', True):       # loss and d loss/d out in one launch
        loss_fn = mse_loss if opt['loss_fn'] == 'mse' else l1_loss
    else:
        loss_fn = F.mse_loss if opt['loss_fn'] == 'mse' else F.l1_loss
    optimizer = FlatAdam(model.parameters(), lr=opt['lr'], weight_decay=opt['decay'], capturable=graphed)
    model.train()
    # (GADAPT_FUSED=0: the captured autograd iteration instead of the fused 13-launch one - A/B runs)
    step = GraphedTrainStep(model, optimizer, loss_fn=loss_fn, fused=os.environ.get('GADAPT_FUSED', '1') != '0') if graphed else None
    if opt.get('device_loader', True):     # samples stacked on the GPU, batches assembled there (no per-step host collation)
        loader = DeviceMeshLoader(dataset, batch_size=opt['batch_size'], shuffle=shuffle, device=opt['device'],
                                  fields=('x_comp', 'x_phys', 'f_tensor', 'uu_tensor'),
                                  into=step.static_batch if graphed else None)
    else:                                  # the reference's shape: CPU collation + .to(device) per step
        loader = MeshLoader(dataset, batch_size=opt['batch_size'], shuffle=shuffle)
    loss_list, best_loss, best_dict, rate = [], float('inf'), None, None
    for epoch in range(opt['epochs']):
        epoch_loss = torch.zeros((), device=opt['device'])
        model.epoch = epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i, data in enumerate(loader):
            data.idx = i
            if graphed:
                loss = step(data)                            # zero_grad + forward + loss + backward + Adam: one graph replay
            else:
                optimizer.zero_grad()
                data = data.to(opt['device'])
                out = model(data)
                loss = loss_fn(out, data.x_phys)
                if opt.get('native_loss', True):
                    loss.backward(gradient=unit_gradient(loss.device))   # = loss.backward(), root gradient not re-created per step
                else:
                    loss.backward()
                optimizer.step()
            epoch_loss += loss.detach()                      # no .item() per batch: one sync per epoch
        loss_list.append(float(epoch_loss))                  # (synchronises)
        rate = len(dataset) / (time.perf_counter() - t0)
        log(f"epoch {epoch} loss {loss_list[-1]:.6e}  {rate:,.0f} meshes/s")
        if loss_list[-1] < best_loss:
            best_loss = loss_list[-1]
            best_dict = {k: v.clone() for k, v in model.state_dict().items()}   # a real copy (the reference's is shallow)
    model.load_state_dict(best_dict)
    return model, loss_list, rate


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--mesh', type=int, default=32)
    ap.add_argument('--num_train', type=int, default=64)
    ap.add_argument('--batch_size', type=int, default=16)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--hidden_dim', type=int, default=64)
    ap.add_argument('--num_layers', type=int, default=4)
    ap.add_argument('--cpu_loader', action='store_true', help='collate on the host every step, as the reference does')
    ap.add_argument('--torch_loss', action='store_true')
    ap.add_argument('--eager', action='store_true', help='eager launches instead of the captured training step')
    a = ap.parse_args()
    opt = hot_path_opt(mesh_dims=[a.mesh, a.mesh], hidden_dim=a.hidden_dim, num_layers=a.num_layers, batch_size=a.batch_size,
                       epochs=a.epochs, device='cuda:0', loss_fn='mse', lr=1e-3, show_mesh_evol_plots='False',
                       device_loader=not a.cpu_loader, native_loss=not a.torch_loss, graphed=not a.eager)
    ds = MeshDataset(opt['mesh_dims'], a.num_train, seed=0)
    t0 = time.time()
    model, losses, rate = main(opt, ds)
    torch.cuda.synchronize()
    print(f"{a.epochs} epochs x {a.num_train} meshes in {time.time() - t0:.2f} s; last epoch {rate:,.0f} meshes/s "
          f"({'graphed step' if opt['graphed'] and not a.cpu_loader and not a.torch_loss else 'eager launches'}); losses {losses}")
