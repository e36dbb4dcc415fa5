#!/usr/bin/env python
"""Diagnostic (VERDICT r4 item 5 / ADVICE r3 #2): is the wide forward's 512-row-window kernel systematically further from fp64 than
the tiled forward on BASELINE config 5's shape (128x128 mesh, 20 Euler steps, hidden 64, one mesh)?  Same inputs through both forward
kernels for several weight / data seeds; error of every parameter gradient against the fp64 oracle, next to the fp32 oracle's own.

    python tools/diag_wide_accuracy.py [n_seeds]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import hip_model_like, make_case, oracle_fp64_twin, rel_err   # noqa: E402
import g_adaptivity_amd.graph as gm                                       # noqa: E402

torch.set_num_threads(min(16, torch.get_num_threads()))
dev = torch.device('cuda:0')
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
names = ('lin_query.weight', 'lin_query.bias', 'lin_key.weight')
print(f"{'seed':>4s} {'parameter':18s} {'oracle fp32':>12s} {'wide fwd':>12s} {'tiled fwd':>12s}   (relative error against the fp64 oracle)")
worse = {n: 0 for n in names}
for seed in range(n_seeds):
    opt, ds, data, oracle = make_case((128, 128), 1, 64, 20, 'GRAND_plus', seed=seed)
    tgt = data.x_phys
    F.mse_loss(oracle(data), tgt).backward()
    o64, _ = oracle_fp64_twin(oracle, ds, opt, data, tgt)
    errs = {}
    for label, (wk, wmin) in {'wide': (True, 0), 'tiled': (False, 0)}.items():
        gm.WIDE_KERNELS, gm.WIDE_MIN_NODES = wk, wmin
        model = hip_model_like(oracle, ds, opt, dev)
        F.mse_loss(model(data.clone().to(dev)), tgt.to(dev)).backward()
        torch.cuda.synchronize()
        g = next(iter(model._graphs.values()))
        assert (g.wide_big_deg > 0) == (label == 'wide')
        errs[label] = {n: rel_err(dict(model.conv_layers[0].named_parameters())[n].grad, dict(o64.conv_layers[0].named_parameters())[n].grad)[0] for n in names}
    for n in names:
        e32 = rel_err(dict(oracle.conv_layers[0].named_parameters())[n].grad, dict(o64.conv_layers[0].named_parameters())[n].grad)[0]
        worse[n] += errs['wide'][n] > errs['tiled'][n]
        print(f"{seed:4d} {n:18s} {e32:12.2e} {errs['wide'][n]:12.2e} {errs['tiled'][n]:12.2e}")
print("wide further from fp64 than tiled in", {n: f"{v}/{n_seeds}" for n, v in worse.items()}, "of the seeds")
