#!/usr/bin/env python
"""CPU diagnostic: what carries the weight-gradient error of an ill-conditioned parity case?

    python tools/diag_accumulation.py 64 2 128 6 GRAND 0      # mesh n, batch, hidden, layers, conv_type, include f

Prints, against the oracle's fp64 twin, the relative error of d lin_query.weight / d lin_query.bias / d lin_key.weight for
  fp32      the oracle as it is (PyG-equivalent op sequence, fp32);
  acc64     the same, but the weight-gradient contraction over the nodes (dW = dQ^T x, db = sum dQ) accumulated in fp64:
            if summation order / accumulation precision carried the error, this row would drop;
  diff      scores and aggregation taken on differences x_j - x_i (shift-invariant forms, exact subtraction of neighbours);
  order k   the fp32 oracle on other edge orders of the same batch (the reference's edge order is the iteration order of a Python
            set, src/data.py:430-441: its own results move from run to run by this much).
Result on BASELINE config 4's shape (64x64, 6 layers, hidden 128, 2 meshes; 8 threads): fp32 7.6e-5 / 6.2e-5 / 5.3e-5, acc64
7.6e-5 / 6.2e-5 / 5.4e-5 (unchanged), diff 8.5e-5 / 7.8e-5 / 8.8e-5, orders 0.76e-4 .. 2.1e-4: the error is fp32 rounding of the
per-node terms amplified by the cancellation (|g| ~ 1e-10), not the order or precision of the sums, and a single fp32 run is a draw
from a band about 3x wide.  docs/measurements.md has the table."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from helpers import make_case, oracle_fp64_twin, permute_edges, rel_err   # noqa: E402
import oracle.pyg_restatement as O                                        # noqa: E402

torch.set_num_threads(int(os.environ.get('OMP_NUM_THREADS', 8)))
n, b, c, l, conv, inc_f = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], bool(int(sys.argv[6]))
opt, ds, data, oracle = make_case((n, n), b, c, l, conv, gnn_inc_feat_f=inc_f)
tgt = data.x_phys
F.mse_loss(oracle(data), tgt).backward()
o64, r64 = oracle_fp64_twin(oracle, ds, opt, data, tgt)
NAMES = ('lin_query.weight', 'lin_query.bias', 'lin_key.weight')


def show(tag):
    errs = [rel_err(dict(oracle.conv_layers[0].named_parameters())[nm].grad, dict(o64.conv_layers[0].named_parameters())[nm].grad)[0] for nm in NAMES]
    print('%-8s ' % tag + '  '.join('%.2e' % e for e in errs), flush=True)


def rerun(d):
    oracle.zero_grad()
    F.mse_loss(oracle(d), tgt).backward()


print('         ' + '  '.join(NAMES) + '   (|g|max %s)' % ' '.join('%.1e' % dict(o64.conv_layers[0].named_parameters())[nm].grad.abs().max() for nm in NAMES))
show('fp32')

_linear = F.linear


class Lin64(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        return _linear(x, w, bias)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return g @ w, (g.double().t() @ x.double()).float(), g.double().sum(0).float()


O.F.linear = lambda x, w, bias=None: Lin64.apply(x, w, bias) if (w.requires_grad and x.dtype == torch.float32 and x.shape[0] > 1000) else _linear(x, w, bias)
rerun(data); show('acc64')
O.F.linear = _linear


def grand_residual_diff(x, edge_index, w_query, b_query, w_key, b_key, temperature=None, return_attention=False, edge_area_sum=None):
    nn_, c_ = x.shape
    src, dst = edge_index[0], edge_index[1]
    q = F.linear(x, w_query, b_query)
    d = x.index_select(0, src) - x.index_select(0, dst)
    s = (q.index_select(0, dst) * F.linear(d, w_key)).sum(-1, keepdim=True) / math.sqrt(c_)
    if temperature is not None:
        s = s / temperature
    alpha = O.pyg_softmax(s, dst, nn_)
    res = torch.zeros(nn_, c_, dtype=x.dtype).index_add_(0, dst, d * alpha)
    deg = torch.zeros(nn_, dtype=x.dtype).index_add_(0, dst, torch.ones_like(dst, dtype=x.dtype))
    res = torch.where((deg > 0).unsqueeze(-1), res, -x)
    return (res, (alpha, None, None)) if return_attention else res


keep, O.grand_residual = O.grand_residual, grand_residual_diff
rerun(data); show('diff')
O.grand_residual = keep
for k in range(1, 6):
    rerun(permute_edges(data, k)); show('order %d' % k)
