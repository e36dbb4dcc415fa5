#!/usr/bin/env python
"""Device time of the one-launch small-mesh forward (kernel id 9) and of the training pair (id 9 with the layer inputs kept + id 10):
HIP event pairs on the launch stream (gadapt_profile_*; each pair includes the dispatch share of its launch, ~4 us), median of 300."""
import ctypes as C, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g_adaptivity_amd import GNN, MeshDataset, collate, hot_path_opt, _native   # noqa: E402
lib = _native.lib()
dev = torch.device('cuda:0')


def med(kid):
    buf = (C.c_double * 4096)()
    n = lib.gadapt_profile_samples(kid, buf, 4096)
    xs = sorted(buf[i] for i in range(max(n, 0)))
    return 1e3 * xs[len(xs) // 2] if xs else float('nan')


lib.gadapt_profile_calibrate(64, _native.current_stream(dev))
for dims, batch, c in (([11, 11], 1, 8), ([11, 11], 32, 8), ([21], 1, 8), ([23, 23], 1, 8), ([23, 23], 16, 8), ([15, 15], 1, 16)):
    opt = hot_path_opt(mesh_dims=dims, hidden_dim=c, num_layers=4, device=str(dev), show_mesh_evol_plots='False')
    ds = MeshDataset(dims, batch, seed=0)
    data = collate(ds.samples).to(dev)
    model = GNN(ds, opt).to(dev)
    import g_adaptivity_amd.functional as Fn
    Fn.small_training_policy = lambda c_, n_: True
    model.eval()
    with torch.no_grad():
        for _ in range(5):
            model(data)
        torch.cuda.synchronize()
        lib.gadapt_profile_reset(); lib.gadapt_profile_enable(1)
        for _ in range(300):
            model(data)
        torch.cuda.synchronize()
        lib.gadapt_profile_enable(0)
        fwd = med(9)
    model.train()
    tgt = data.x_phys.reshape(-1, model.dim)
    for _ in range(5):
        model.zero_grad(); F.mse_loss(model(data), tgt).backward()
    torch.cuda.synchronize()
    lib.gadapt_profile_reset(); lib.gadapt_profile_enable(1)
    for _ in range(300):
        model.zero_grad(); F.mse_loss(model(data), tgt).backward()
    torch.cuda.synchronize()
    lib.gadapt_profile_enable(0)
    print(f"{'x'.join(map(str, dims)):>6s} batch {batch:3d} hidden {c:2d}: evaluation forward {fwd:6.1f} us; training forward {med(9):6.1f} us, backward {med(10):6.1f} us (event pairs)", flush=True)
    lib.gadapt_profile_reset()
