#!/usr/bin/env python
"""Diagnostic: gradient error against the fp64 oracle of the one-launch small-mesh kernels vs the per-layer launches on BASELINE
config 1's shape (1-D, 32 nodes, 1 GRAND layer, batch 8, hidden 8) for several seeds (the parity case that is a remainder of
cancelling sums: |grad| ~ 1e-6)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import hip_model_like, make_case, oracle_fp64_twin, rel_err   # noqa: E402
import g_adaptivity_amd.functional as Fn                                   # noqa: E402

dev = torch.device('cuda:0')
shape = ((32,), 8, 8, 1, 'GRAND') if len(sys.argv) < 2 else eval(sys.argv[1])
names = ('lin_query.weight', 'lin_query.bias', 'lin_key.weight')
print(f"{'seed':>4s} {'parameter':18s} {'|grad| max':>11s} {'oracle fp32':>12s} {'one-launch':>12s} {'per-layer':>12s}   (relative error against the fp64 oracle)")
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
import math
log_ratio = []
for seed in range(n_seeds):
    opt, ds, data, oracle = make_case(*shape, seed=seed)
    tgt = data.x_phys if data.x_phys.dim() == 2 else data.x_phys.unsqueeze(-1)
    F.mse_loss(oracle(data), tgt).backward()
    o64, _ = oracle_fp64_twin(oracle, ds, opt, data, tgt)
    errs = {}
    for label, small in (('small', True), ('layers', False)):
        Fn.SMALL_MESH_FORWARD = small
        model = hip_model_like(oracle, ds, opt, dev)
        F.mse_loss(model(data.clone().to(dev)), tgt.to(dev)).backward()
        torch.cuda.synchronize()
        errs[label] = {n: rel_err(dict(model.conv_layers[0].named_parameters())[n].grad, dict(o64.conv_layers[0].named_parameters())[n].grad)[0] for n in names}
    for n in names:
        g64 = dict(o64.conv_layers[0].named_parameters())[n].grad
        e32 = rel_err(dict(oracle.conv_layers[0].named_parameters())[n].grad, g64)[0]
        if n_seeds <= 6:
            print(f"{seed:4d} {n:18s} {g64.abs().max().item():11.2e} {e32:12.2e} {errs['small'][n]:12.2e} {errs['layers'][n]:12.2e}")
        if n != 'lin_query.bias':
            log_ratio.append((math.log(errs['small'][n] / errs['layers'][n]), math.log(errs['small'][n] / e32), math.log(errs['layers'][n] / e32)))
import statistics
print(f"{n_seeds} seeds, weight gradients: geometric-mean error ratio one-launch / per-layer = {math.exp(statistics.mean(r[0] for r in log_ratio)):.2f}, "
      f"one-launch / fp32 oracle = {math.exp(statistics.mean(r[1] for r in log_ratio)):.2f}, per-layer / fp32 oracle = {math.exp(statistics.mean(r[2] for r in log_ratio)):.2f}")
