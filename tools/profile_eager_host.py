"""Diagnostic: where the HOST time of the eager training iteration goes (cProfile over 300 iterations of the metric workload)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, mse_loss, unit_gradient
from g_adaptivity_amd.optim import FlatAdam
dev = torch.device('cuda:0')
opt = hot_path_opt(mesh_dims=[64, 64], hidden_dim=64, num_layers=4, device=str(dev), show_mesh_evol_plots='False')
ds = MeshDataset([64, 64], 32, seed=0)
data = collate(ds.samples).to(dev)
torch.manual_seed(0)
model = GNN(ds, opt).to(dev).train()
optim = FlatAdam(model.parameters(), lr=1e-3)
root = unit_gradient(dev)
def it():
    optim.zero_grad()
    mse_loss(model(data), data.x_phys).backward(gradient=root)
    optim.step()
for _ in range(20): it()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(300): it()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host issue time {t_issue / 300 * 1e6:.1f} us per iteration; with sync {(time.perf_counter() - t0) / 300 * 1e6:.1f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): it()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
