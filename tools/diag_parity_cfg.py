"""Gradient error of one parity case against the fp64 oracle, next to the fp32 oracle's own error (GPU box).
    python tools/diag_parity_cfg.py 64 2 128 6 GRAND 0      # mesh n, batch, hidden, layers, conv_type, include f"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from helpers import make_case, hip_model_like, rel_err, oracle_fp64_twin
n, b, c, l, conv, inc_f = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], bool(int(sys.argv[6]))
dev = torch.device('cuda:0')
import os
from g_adaptivity_amd import graph as _g
_g.WIDE_MIN_NODES = int(os.environ.get('GADAPT_DIAG_WIDE_MIN', _g.WIDE_MIN_NODES))   # 0: wide forward for every size (what the tests do)
opt, ds, data, oracle = make_case((n, n), b, c, l, conv, gnn_inc_feat_f=inc_f)
model = hip_model_like(oracle, ds, opt, dev)
tgt = data.x_phys
ref = oracle(data); F.mse_loss(ref, tgt).backward()
o64, r64 = oracle_fp64_twin(oracle, ds, opt, data, tgt)
out = model(data.clone().to(dev)); F.mse_loss(out, tgt.to(dev)).backward(); torch.cuda.synchronize()
print('x: hip-vs-64 %.2e o32-vs-64 %.2e' % (rel_err(out, r64)[0], rel_err(ref, r64)[0]))
for name in ('lin_query.weight', 'lin_query.bias', 'lin_key.weight'):
    g32 = dict(oracle.conv_layers[0].named_parameters())[name].grad
    g64 = dict(o64.conv_layers[0].named_parameters())[name].grad
    gh = dict(model.conv_layers[0].named_parameters())[name].grad
    print('  %-18s hip-vs-64 %.2e  o32-vs-64 %.2e  hip-vs-32 %.2e  |g|max %.2e' % (name, rel_err(gh, g64)[0], rel_err(g32, g64)[0], rel_err(gh, g32)[0], g64.abs().max()))
