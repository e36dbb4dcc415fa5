#!/usr/bin/env python
"""Diagnostic: dA / dp0 of the one-launch small backward for ONE layer against an fp64 evaluation of the same formulas on the SAME
fp32 activations (x, alpha, g) - separates arithmetic error inside the kernel from differences in the forward's activations."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import hip_model_like, make_case   # noqa: E402
import g_adaptivity_amd.functional as Fn       # noqa: E402
from g_adaptivity_amd._native import lib, check, ptr, current_stream   # noqa: E402

dev = torch.device('cuda:0')
for seed in range(3):
    opt, ds, data, oracle = make_case((32,), 8, 8, 1, 'GRAND', seed=seed)
    model = hip_model_like(oracle, ds, opt, dev).train()
    d = data.clone().to(dev)
    xc = d.x_comp.unsqueeze(-1) if d.x_comp.dim() == 1 else d.x_comp
    graph = model._graph(d, xc.shape[0], dev)
    with torch.enable_grad():
        plan = model._small_plan(d, graph, xc, d.f_tensor, d.uu_tensor)
    assert plan is not None and plan['train']
    wq, bq, wk, lp = (plan[k].detach().contiguous() for k in ('wq', 'bq', 'wk', 'lp'))
    out, alpha, x_all = Fn._small_launch(graph, plan['part'], xc.contiguous(), d.f_tensor, d.uu_tensor, plan['enc_w'], wq, bq, wk, lp, 1, 1, True, True)
    tgt = d.x_phys.reshape(out.shape)
    g_top = (2.0 * (out - tgt) / out.numel()).contiguous()
    mesh_ptr, n_meshes, mn, me = plan['part']
    c = 8
    slab = torch.empty(1, n_meshes, c * c + c, device=dev)
    dbg = torch.zeros(4096, device=dev)
    import ctypes as C_
    h = lib() if hasattr(lib(), 'gadapt_debug_set_stamp_buffer') else None     # a -DGADAPT_STAMPS build (make EXTRA=-DGADAPT_STAMPS LIB=variants/stamps.so OBJDIR=build/obj_stamps; GADAPT_LIB=variants/stamps.so)
    if h is not None:
        h.gadapt_debug_set_stamp_buffer(C_.c_void_p(dbg.data_ptr()))
    check(lib().gadapt_small_backward(graph.c_ref, ptr(mesh_ptr[0]), ptr(mesh_ptr[1]), n_meshes, mn, me, ptr(x_all), ptr(alpha), ptr(g_top), 1, ptr(wq), ptr(bq), ptr(wk), 0, 0,
                                      ptr(lp), 1, ptr(slab), c, current_stream(dev)), 'bwd')
    torch.cuda.synchronize()
    row = slab[0].double().sum(0).cpu()
    dA_k, dp0_k = row[:64].view(8, 8), row[64:]
    # fp64 evaluation on the same activations
    x = x_all[0].double().cpu(); a = alpha[0].double().cpu(); g = torch.zeros_like(x); g[:, :1] = g_top.double().cpu()
    rp, col = graph.rowptr_t.cpu().long(), graph.col_t.cpu().long()
    dt, sc = float(lp[0, 0]), float(lp[0, 1])
    n = x.shape[0]
    dst = torch.repeat_interleave(torch.arange(n), rp[1:] - rp[:-1])
    E = graph.num_edges
    da = dt * (g[dst] * x[col[:E]]).sum(1)
    D = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, a[:E] * da)
    dsv = a[:E] * (da - D[dst]) * sc
    dP = torch.zeros_like(x).index_add_(0, dst, dsv.unsqueeze(1) * x[col[:E]])
    dA_r, dp0_r = dP.t() @ x, dP.sum(0)
    rel = lambda u, v: ((u - v).abs().max() / v.abs().max()).item()
    print(f"seed {seed}: dA kernel vs fp64-on-same-activations {rel(dA_k, dA_r):.2e}   dp0 {rel(dp0_k, dp0_r):.2e}   |dA| max {dA_r.abs().max():.2e}  sum-of-|terms| / |dA| ~ {((dP.abs().t() @ x.abs()).max() / dA_r.abs().max()).item():.1e}")
    if h is not None:
        lib().gadapt_debug_set_stamp_buffer(None)
        nm = int(mesh_ptr[0][1])
        dP_k, x_k = dbg[:nm * 8].view(nm, 8).double().cpu(), dbg[nm * 8:2 * nm * 8].view(nm, 8).double().cpu()
        print(f"   mesh 0: dP tile vs fp64 {rel(dP_k, dP[:nm]):.2e}; x tile vs x_all {rel(x_k, x[:nm]):.2e}; dA from the dumped tiles (fp64 product) vs kernel's slab row of mesh 0 {rel((dP_k.t() @ x_k), slab[0, 0, :64].double().cpu().view(8, 8)):.2e}")
        print("   worst dP rows:", (dP_k - dP[:nm]).abs().max(1).values.topk(3))
    print("   rows sum alpha - 1 (max abs):", (torch.zeros(n, dtype=torch.float64).index_add_(0, dst, a[:E]) - 1).abs().max().item())
