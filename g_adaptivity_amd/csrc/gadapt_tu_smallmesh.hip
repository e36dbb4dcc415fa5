// gadapt_tu_smallmesh.hip - the reference's own sizes (src/params.py:37,56,107,130-134): encoder + L Euler steps + head of a batch of
// small meshes as ONE launch forward and ONE launch backward, one workgroup per mesh (gadapt_smallmesh.inc).  One translation
// unit of libgadapt_hip.so (see gadapt_internal.h).
#include "gadapt_internal.h"
#include "gadapt_smallmesh.inc"

// ------------------------------------------------------------------------------------------------
// one-launch evaluation forward of a batch of small meshes (gadapt_smallmesh.inc)
// ------------------------------------------------------------------------------------------------
extern "C" int64_t gadapt_small_forward_lds_bytes(int max_mesh_nodes, int max_mesh_edges, int c) {
    // one node per thread: 1024 nodes per mesh at most (512 at hidden 32: a row, its projection and its aggregate are 96 registers)
    if (max_mesh_nodes <= 0 || max_mesh_edges < 0 || max_mesh_nodes > (c == 32 ? 512 : 1024)) return -1;
    int64_t fl;
    switch (c) {
        case 4: fl = smallmesh::lds_floats<4>(max_mesh_nodes, max_mesh_edges); break;
        case 8: fl = smallmesh::lds_floats<8>(max_mesh_nodes, max_mesh_edges); break;
        case 16: fl = smallmesh::lds_floats<16>(max_mesh_nodes, max_mesh_edges); break;
        case 32: fl = smallmesh::lds_floats<32>(max_mesh_nodes, max_mesh_edges); break;
        default: return -1;
    }
    return 4 * fl <= 160 * 1024 ? 4 * fl : -1;
}
// returns the workgroup size it launched with (the fused loss leaves one partial per wave)
template <int C> static int launch_small(smallmesh::Args& p, int n_meshes, int lds, hipStream_t st, int* n_partials) {
    int nt_used = 0;
    auto go = [&](auto kern, int nt) {
        nt_used = nt;
        if (p.loss.target && (int64_t)n_meshes * (nt / 64) > GADAPT_LOSS_PARTIALS_MAX) { nt_used = -1; return; }
        if (n_partials) *n_partials = n_meshes * (nt / 64);
        allow_lds(kern, lds); hipLaunchKernelGGL(kern, dim3(n_meshes), dim3(nt), lds, st, p);
    };
    // lanes per node: as many as the workgroup has threads for (hidden 32: 512 threads at most - a row is 32 registers)
    const int nodes = p.max_nodes;
    if constexpr (C == 16) {                                    // 128 registers at 1024 threads would spill: 512 threads from 129 nodes on
        if (nodes <= 64) go(smallmesh::fwd_kernel<C, 256, 4>, 256);
        else if (nodes <= 128) go(smallmesh::fwd_kernel<C, 512, 4>, 512);
        else if (nodes <= 256) go(smallmesh::fwd_kernel<C, 512, 2>, 512);
        else if (nodes <= 512) go(smallmesh::fwd_kernel<C, 512, 1>, 512);
        else go(smallmesh::fwd_kernel<C, 1024, 1>, 1024);
    } else if constexpr (C < 32) {
        if (nodes <= 64) go(smallmesh::fwd_kernel<C, 256, 4>, 256);
        else if (nodes <= 128) go(smallmesh::fwd_kernel<C, 512, 4>, 512);
        else if (nodes <= 256) go(smallmesh::fwd_kernel<C, 1024, 4>, 1024);
        else if (nodes <= 512) go(smallmesh::fwd_kernel<C, 1024, 2>, 1024);
        else go(smallmesh::fwd_kernel<C, 1024, 1>, 1024);
    } else {
        if (nodes <= 128) go(smallmesh::fwd_kernel<C, 256, 2>, 256);
        else if (nodes <= 256) go(smallmesh::fwd_kernel<C, 512, 2>, 512);
        else go(smallmesh::fwd_kernel<C, 512, 1>, 512);
    }
    return nt_used;
}
static int small_forward_impl(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                                    const float* x_comp, int dim, const float* f, const float* uu, const float* enc_w, int n_feat,
                                    const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                                    const float* layer_params, int n_layers, float* out, int out_cols, float* alpha_all, float* x_all, int c, void* stream,
                                    const LossArgs& loss, int* n_partials) {
    if (!g || !g->rowptr_t || !g->col_t || !mesh_ptr || n_meshes <= 0 || !x_comp || !enc_w || !wq || !bq || !wk || !layer_params || !out)
        return fail(GADAPT_E_BADARG, "small_forward: null pointer");
    if (dim < 1 || dim > 4 || n_feat != dim + (f ? 1 : 0) + (uu ? 1 : 0) || n_feat > smallmesh::MAXF || n_layers <= 0 || n_layers > smallmesh::MAX_LAYERS
        || out_cols < 1 || out_cols > c)
        return fail(GADAPT_E_BADARG, "small_forward: 1..4 coordinates, encoder columns = coordinates + extras, at most 64 layers, 1 <= out_cols <= hidden");
    if (n_meshes == 1 && (max_mesh_nodes < g->n_nodes || max_mesh_edges < g->n_edges))
        return fail(GADAPT_E_BADARG, "small_forward: one mesh = the whole graph: max_mesh_nodes / max_mesh_edges below its node / edge count");
    const int64_t lds = gadapt_small_forward_lds_bytes(max_mesh_nodes, max_mesh_edges, c);
    if (lds < 0) return fail(GADAPT_E_BADARG, "small_forward: hidden in {4,8,16,32}, at most 1024 nodes per mesh (512 at hidden 32), rows + CSR slice within 160 KB of LDS");
    smallmesh::Args p{x_comp, f, uu, enc_w, dim, n_feat, wq, bq, wk, w_stride, b_stride, layer_params, g->rowptr_t, g->col_t, mesh_ptr, mesh_eptr, n_meshes,
                      out, out_cols, alpha_all, n_layers, g->n_edges, max_mesh_nodes, max_mesh_edges, x_all, g->n_nodes};
    p.loss = loss;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(9, st, x_all ? 32 : 0);
    int nt;
    switch (c) {
        case 4: nt = launch_small<4>(p, n_meshes, (int)lds, st, n_partials); break;
        case 8: nt = launch_small<8>(p, n_meshes, (int)lds, st, n_partials); break;
        case 16: nt = launch_small<16>(p, n_meshes, (int)lds, st, n_partials); break;
        default: nt = launch_small<32>(p, n_meshes, (int)lds, st, n_partials); break;
    }
    if (nt < 0) return fail(GADAPT_E_BADARG, "small_forward_loss: more loss partials (one per wave) than gadapt_loss_partials_max()");
    return check_launch("smallmesh::fwd_kernel");
}
extern "C" int gadapt_small_forward(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                                    const float* x_comp, int dim, const float* f, const float* uu, const float* enc_w, int n_feat,
                                    const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                                    const float* layer_params, int n_layers, float* out, int out_cols, float* alpha_all, float* x_all, int c, void* stream) {
    return small_forward_impl(g, mesh_ptr, mesh_eptr, n_meshes, max_mesh_nodes, max_mesh_edges, x_comp, dim, f, uu, enc_w, n_feat, wq, bq, wk, w_stride, b_stride,
                              layer_params, n_layers, out, out_cols, alpha_all, x_all, c, stream, LossArgs{}, nullptr);
}
// The same launch as the head of a fused training step (training.FusedIteration on small-mesh batches): it also writes
// seed = d loss / d out [N,d] (d = out_cols; mean squared error, l1 = 1: mean absolute error - run_GNN.py:80-84,106, the arithmetic of
// gadapt_loss_forward: bit-identical gradients) and one partial sum of the loss per wave to loss_partials.  Returns the number of
// partials (> 0; gadapt_step_tail sums them) or a negative error code.
extern "C" int gadapt_small_forward_loss(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                                         const float* x_comp, int dim, const float* f, const float* uu, const float* enc_w, int n_feat,
                                         const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                                         const float* layer_params, int n_layers, float* out, int out_cols, float* alpha_all, float* x_all,
                                         const float* target, int l1, float* seed, float* loss_partials, int c, void* stream) {
    if (!target || !seed || !loss_partials || !x_all || !alpha_all || out_cols < 1 || out_cols > 4 || !g)
        return fail(GADAPT_E_BADARG, "small_forward_loss: target, seed, loss_partials, x_all, alpha_all; 1 <= out_cols <= 4");
    LossArgs loss{target, seed, loss_partials, out_cols, l1 ? 1 : 0, 1.0f / (float)((int64_t)g->n_nodes * out_cols)};
    int n_partials = 0;
    const int rc = small_forward_impl(g, mesh_ptr, mesh_eptr, n_meshes, max_mesh_nodes, max_mesh_edges, x_comp, dim, f, uu, enc_w, n_feat, wq, bq, wk, w_stride, b_stride,
                                      layer_params, n_layers, out, out_cols, alpha_all, x_all, c, stream, loss, &n_partials);
    return rc != 0 ? rc : n_partials;
}

extern "C" int64_t gadapt_small_backward_lds_bytes(int max_mesh_nodes, int max_mesh_edges, int c) {
    if (max_mesh_nodes <= 0 || max_mesh_edges < 0 || max_mesh_edges > 65000 || max_mesh_nodes > (c == 32 ? 512 : 1024)) return -1;
    int64_t fl;
    switch (c) {
        case 4: fl = smallmesh::bwd_lds_floats<4>(max_mesh_nodes, max_mesh_edges); break;
        case 8: fl = smallmesh::bwd_lds_floats<8>(max_mesh_nodes, max_mesh_edges); break;
        case 16: fl = smallmesh::bwd_lds_floats<16>(max_mesh_nodes, max_mesh_edges); break;
        case 32: fl = smallmesh::bwd_lds_floats<32>(max_mesh_nodes, max_mesh_edges); break;
        default: return -1;
    }
    return 4 * fl <= 160 * 1024 ? 4 * fl : -1;
}
template <int C> static void launch_small_bwd(const smallmesh::BwdArgs& p, int n_meshes, int lds, hipStream_t st) {
    auto go = [&](auto kern, int nt) { allow_lds(kern, lds); hipLaunchKernelGGL(kern, dim3(n_meshes), dim3(nt), lds, st, p); };
    // lanes per node as in launch_small, within 512 threads (1024-thread workgroups have no registers for the layer-ahead requests)
    const int nodes = p.max_nodes;
    if constexpr (C < 32) {
        if (nodes <= 64) go(smallmesh::bwd_kernel<C, 256, 4>, 256);
        else if (nodes <= 128) go(smallmesh::bwd_kernel<C, 512, 4>, 512);
        else if (nodes <= 256) go(smallmesh::bwd_kernel<C, 512, 2>, 512);
        else if (nodes <= 512) go(smallmesh::bwd_kernel<C, 512, 1>, 512);
        else go(smallmesh::bwd_kernel<C, 1024, 1>, 1024);
    } else {
        if (nodes <= 256) go(smallmesh::bwd_kernel<C, 256, 1>, 256);
        else go(smallmesh::bwd_kernel<C, 512, 1>, 512);
    }
}
extern "C" int gadapt_small_backward(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                                     const float* x_all, const float* alpha_all, const float* g_top, int g_cols,
                                     const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                                     const float* layer_params, int n_layers, float* slab, int c, void* stream) {
    if (!g || !g->rowptr_t || !g->col_t || !g->rowptr_s || !g->col_s || !g->perm_s || !mesh_ptr || n_meshes <= 0 || !x_all || !alpha_all || !g_top
        || !wq || !bq || !wk || !layer_params || !slab)
        return fail(GADAPT_E_BADARG, "small_backward: null pointer");
    if (g_cols < 1 || g_cols > c || n_layers <= 0 || n_layers > smallmesh::MAX_LAYERS) return fail(GADAPT_E_BADARG, "small_backward: 1 <= g_cols <= hidden, at most 64 layers");
    if (n_meshes == 1 && (max_mesh_nodes < g->n_nodes || max_mesh_edges < g->n_edges))
        return fail(GADAPT_E_BADARG, "small_backward: one mesh = the whole graph: max_mesh_nodes / max_mesh_edges below its node / edge count");
    const int64_t lds = gadapt_small_backward_lds_bytes(max_mesh_nodes, max_mesh_edges, c);
    if (lds < 0) return fail(GADAPT_E_BADARG, "small_backward: hidden in {4,8,16,32}, at most 1024 nodes per mesh (512 at hidden 32), three row tiles + both CSR slices within 160 KB of LDS");
    // slab: [S][n_meshes][C*C + C] - one row set per conv, what gadapt_slab_reduce_coeffs_backward takes per conv
    smallmesh::BwdArgs p{x_all, alpha_all, g_top, g_cols, wq, bq, wk, w_stride, b_stride, layer_params, g->rowptr_t, g->col_t, g->rowptr_s, g->col_s, g->perm_s,
                         mesh_ptr, slab, w_stride ? (int64_t)n_meshes * (c * c + c) : 0, n_layers, g->n_edges, max_mesh_nodes, max_mesh_edges, g->n_nodes, mesh_eptr, n_meshes, nullptr};
#ifdef GADAPT_STAMPS
    p.dbg = reinterpret_cast<float*>(g_stamp_buf);
#endif
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(10, st, 0);
    switch (c) {
        case 4: launch_small_bwd<4>(p, n_meshes, (int)lds, st); break;
        case 8: launch_small_bwd<8>(p, n_meshes, (int)lds, st); break;
        case 16: launch_small_bwd<16>(p, n_meshes, (int)lds, st); break;
        default: launch_small_bwd<32>(p, n_meshes, (int)lds, st); break;
    }
    return check_launch("smallmesh::bwd_kernel");
}
