// gadapt_tu_bwd_target.hip - backward target pass launches (autograd of src/GRAND_plus.py:225-343: softmax backward, dP,
// weight-gradient partials, dxd): the tiled kernel and the compact layer-0 kernel (gadapt_bwd_target.inc).  One translation
// unit of libgadapt_hip.so (see gadapt_internal.h).
#include "gadapt_internal.h"
#include "gadapt_bwd_target.inc"

// out4: only columns 0..3 of g_out are wanted (dxd is [N,4]: D4 target pass, then grand_bwd_source4_kernel).
// g_stride: row pitch of g_in in floats for the compact-input launch (0 = C).
template <int C> static int launch_bwd_target(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha, const float* a,
                                              const float* lp, float* edge_ws, float* dxd, float* slab, int accumulate, float* sums_out,
                                              float* sums_sc_out, int want_source, int residual_only, int g_cols, int x_cols, int out4,
                                              int g_stride, int sums_partials, hipStream_t st) {
    using K = Cfg<C>;
    const int n_tiles = tiles_for<C>(g->n_nodes);
    if (g_cols < 0 || g_cols > 4) return fail(GADAPT_E_BADARG, "compact upstream gradient: 1..4 columns");
    if ((x_cols != 0 && x_cols != 4) || (x_cols && (g_cols || want_source)))
        return fail(GADAPT_E_BADARG, "compact layer input: 4 columns, layer 0 of a block of >= 2 layers, no d x0");
    if (x_cols && residual_only) return fail(GADAPT_E_BADARG, "compact layer input: Euler-step layers only");
    if (out4 && (!want_source || x_cols || residual_only || C < 8 || (g_cols && sums_out)))
        return fail(GADAPT_E_BADARG, "4-column backward: a layer with a gradient to pass on, hidden >= 8, not compact-g with d dt / d scale");
    BwdTArgs pt{x_in, g_in, alpha, a, lp, g->rowptr_t, g->col_t, g->tpos_s, meta_for<K::TM>(g->meta_t), reinterpret_cast<float2*>(edge_ws), dxd, slab, sums_out,
                g->n_nodes, n_tiles, accumulate, residual_only, g->n_edges, nullptr, g_cols};
    pt.sums_sc_out = sums_sc_out;
    pt.c = C;
    pt.g_stride = g_stride ? g_stride : C;
    pt.sums_partials = sums_partials;
#ifdef GADAPT_STAMPS
    pt.stamps = g_stamp_buf ? g_stamp_buf + 1024 * 32 : nullptr;
#endif
    // the slab holds one row per workgroup of the target pass: every launch of a block uses this grid, so every row is visited
    const dim3 grid(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS)));
    if (x_cols) {                                                // layer 0 on the compact [N,4] input: one node per lane
        ProfScope prof(1, st, 2);
        pt.ell = g->ell_t;
        if (sums_out && sums_sc_out) hipLaunchKernelGGL((grand_bwd_target_compact_kernel<2>), grid, dim3(256), 0, st, pt);
        else if (sums_out) hipLaunchKernelGGL((grand_bwd_target_compact_kernel<1>), grid, dim3(256), 0, st, pt);
        else if (pt.ell) hipLaunchKernelGGL((grand_bwd_target_compact_kernel<0, true>), grid, dim3(256), 0, st, pt);
        else hipLaunchKernelGGL((grand_bwd_target_compact_kernel<0>), grid, dim3(256), 0, st, pt);
        return check_launch("grand_bwd_target_compact_kernel");
    }
    constexpr int lds_t = K::lds_bytes(1, K::RING_T + 1, 1);
    ProfScope prof(1, st, (g_cols ? 1 : 0) | (out4 ? 8 : 0));
    // instantiations: SUMS 0 / 1 (d dt: learn_step) / 2 (d dt and d score_scale), each plain, with the compact upstream
    // gradient (GC) and - hidden >= 8 - with the 4-column dxd (D4)
    auto go = [&](auto kern) { allow_lds(kern, lds_t); hipLaunchKernelGGL(kern, grid, dim3(K::NT), lds_t, st, pt); };
    auto pick = [&](auto sums_tag) {
        constexpr int S = decltype(sums_tag)::value;
        if (out4) {
            if constexpr (C >= 8) {
                if (!g_cols) go(grand_bwd_target_kernel<C, S, false, true>);
                else if constexpr (S == 0) go(grand_bwd_target_kernel<C, 0, true, true>);   // GC + D4 + SUMS: not built (gadapt_block_backward)
            }
        } else if (g_cols) {
            go(grand_bwd_target_kernel<C, S, true>);
        } else {
            go(grand_bwd_target_kernel<C, S>);
        }
    };
    if (sums_out && sums_sc_out) pick(IntTag<2>{}); else if (sums_out) pick(IntTag<1>{}); else pick(IntTag<0>{});
    return check_launch("grand_bwd_target_kernel");
}

int gadapt_launch_bwd_target_c(int c, const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha, const float* a,
                               const float* lp, float* edge_ws, float* dxd, float* slab, int accumulate, float* sums_out, float* sums_sc_out,
                               int want_source, int residual_only, int g_cols, int x_cols, int out4, int g_stride, int sums_partials,
                               hipStream_t st) {
    GADAPT_DISPATCH_C(c, launch_bwd_target<CC>(g, x_in, g_in, alpha, a, lp, edge_ws, dxd, slab, accumulate, sums_out, sums_sc_out, want_source,
                                               residual_only, g_cols, x_cols, out4, g_stride, sums_partials, st));
}

int gadapt_slab_rows_c(int64_t n_nodes, int c) {
    GADAPT_DISPATCH_C(c, grid_for(tiles_for<CC>(n_nodes), resident_blocks_bwd_t<CC>(GADAPT_BWD_T_MAX_BLOCKS)));
}

template <int C> static int occupancy_bwd_target() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, grand_bwd_target_kernel<C, 0>, Cfg<C>::NT, Cfg<C>::lds_bytes(1, Cfg<C>::RING_T + 1, 1));
    return n;
}
int gadapt_occupancy_bwd_target_c(int c) { GADAPT_DISPATCH_C(c, occupancy_bwd_target<CC>()); }
