import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, GraphedTrainStep, mse_loss
from g_adaptivity_amd.optim import FlatAdam
dev = torch.device('cuda:0')
for mesh, hidden, B in ((11, 8, 8), (11, 8, 64), (23, 8, 16), (11, 16, 8), (16, 64, 8), (32, 64, 8)):
    opt = hot_path_opt(mesh_dims=[mesh, mesh], hidden_dim=hidden, num_layers=4, device=str(dev), show_mesh_evol_plots='False')
    ds = MeshDataset([mesh, mesh], B, seed=0); data = collate(ds.samples).to(dev)
    torch.manual_seed(0)
    model = GNN(ds, opt).to(dev).train()
    optim = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
    step = GraphedTrainStep(model, optim, loss_fn=mse_loss)
    step(data); torch.cuda.synchronize()
    c = next(iter(step._captured.values()))
    # host cost of an issued fused step: issue 300 without waiting
    t0 = time.perf_counter()
    for _ in range(300):
        c.fused.forward_backward(); c.fused.finish()
    host = (time.perf_counter() - t0) / 300
    torch.cuda.synchronize()
    print(f"{mesh}x{mesh} hidden {hidden} batch {B}: small={c.fused.small is not None} eager={c.eager} ab={c.replay_ab} host issue {host*1e6:.1f} us/step", flush=True)
