// gadapt_tu_bwd_source.hip - backward source pass launches: g_out = dxd + sum over out-edges (alpha dt g_i) + A y + sigma p0
// (autograd of PyG's MessagePassing.propagate at src/GRAND_plus.py:233-234, grouped by source), the dense kernel, its windowed
// form for hidden 128 and the 4-column form (gadapt_bwd_source.inc).  One translation unit of libgadapt_hip.so.
#include "gadapt_internal.h"
#include "gadapt_bwd_source.inc"

// Source pass with the LDS window of x rows (grand_bwd_source_kernel, WIN): hidden 128 only.  Measured (MI355X, same box): hidden
// 128, one wave per SIMD, every round trip exposed: 82.3 -> 79.3 us with whole-node steps (81.2 with half-node steps).  Hidden 64:
// 25.5 -> 32.2 us, with either step size - the window needs each workgroup on CONSECUTIVE tiles, and then the 64 workgroups of an
// XCD gather g rows from 192 distinct slabs at a time instead of 66 (interleaved order: 64 adjacent tiles): the first tile of a
// workgroup takes 10.4k cycles in the edge walk against 6.8k, and the L2 (4 MB per XCD) does not hold a slab until the
// neighbouring tile's step.  128-wide meshes (no tile windowed): 61 -> 80 us.
constexpr bool source_window(int c) { return c == 128; }

template <int C> static int launch_bwd_source(const gadapt_graph* g, const float* x_in, const float* g_in, const float* edge_ws, const float* dxd,
                                              const float* a, const float* p0, float* g_out, int g_cols, int out4, hipStream_t st) {
    using K = Cfg<C>;
    const int n_tiles = tiles_for<C>(g->n_nodes);
    BwdSArgs ps{x_in, g_in, edge_ws, dxd, a, p0, g->rowptr_s, g->col_s, meta_for<K::TM>(g->meta_s), g_out, g->n_nodes, n_tiles, g->n_edges, nullptr, g_cols};
#ifdef GADAPT_STAMPS
    ps.stamps = g_stamp_buf ? g_stamp_buf + 2 * 1024 * 32 : nullptr;
#endif
    ProfScope prof(2, st, (g_cols ? 1 : 0) | (out4 ? 8 : 0));
    if constexpr (C >= 8) {
        if (out4) {
            constexpr int lds_4 = K::lds_bytes(2, 0);
            constexpr int res4 = (C > 64 || K::NT != 256) ? 512 : 256 * GADAPT_WAVES_BWD_S4;   // resident workgroups: waves per SIMD x 256 CUs
            const dim3 grid4(grid_for(n_tiles, res4));
            if (g_cols) {
                allow_lds(grand_bwd_source4_kernel<C, true>, lds_4);
                hipLaunchKernelGGL((grand_bwd_source4_kernel<C, true>), grid4, dim3(K::NT), lds_4, st, ps);
            } else {
                allow_lds(grand_bwd_source4_kernel<C, false>, lds_4);
                hipLaunchKernelGGL((grand_bwd_source4_kernel<C, false>), grid4, dim3(K::NT), lds_4, st, ps);
            }
            return check_launch("grand_bwd_source4_kernel");
        }
    }
    if (out4) return fail(GADAPT_E_BADARG, "4-column source pass: hidden >= 8");
    const dim3 grid(grid_for(n_tiles, resident_blocks_bwd<C>(GADAPT_BWD_S_MAX_BLOCKS)));
    if constexpr (K::MFMA && source_window(C)) {
        // mesh-ordered graphs only (the wide kernels' test: every out-neighbour near its node): elsewhere no tile is windowed and
        // the slabs would be staged for nothing
        if (g->wide_deg_s > 0) {
            constexpr int lds_sw = K::lds_bytes(2, 4, 1);       // window (3 slabs) + y tile; ext = ring offsets
            if (g_cols) {
                allow_lds(grand_bwd_source_kernel<C, true, true>, lds_sw);
                hipLaunchKernelGGL((grand_bwd_source_kernel<C, true, true>), grid, dim3(K::NT), lds_sw, st, ps);
            } else {
                allow_lds(grand_bwd_source_kernel<C, false, true>, lds_sw);
                hipLaunchKernelGGL((grand_bwd_source_kernel<C, false, true>), grid, dim3(K::NT), lds_sw, st, ps);
            }
            return check_launch("grand_bwd_source_kernel");
        }
    }
    constexpr int lds_s = K::lds_bytes(2);
    if (g_cols) {
        allow_lds(grand_bwd_source_kernel<C, true>, lds_s);
        hipLaunchKernelGGL((grand_bwd_source_kernel<C, true>), grid, dim3(K::NT), lds_s, st, ps);
    } else {
        allow_lds(grand_bwd_source_kernel<C>, lds_s);
        hipLaunchKernelGGL(grand_bwd_source_kernel<C>, grid, dim3(K::NT), lds_s, st, ps);
    }
    return check_launch("grand_bwd_source_kernel");
}

int gadapt_launch_bwd_source_c(int c, const gadapt_graph* g, const float* x_in, const float* g_in, const float* edge_ws, const float* dxd,
                               const float* a, const float* p0, float* g_out, int g_cols, int out4, hipStream_t st) {
    GADAPT_DISPATCH_C(c, launch_bwd_source<CC>(g, x_in, g_in, edge_ws, dxd, a, p0, g_out, g_cols, out4, st));
}

template <int C> static int occupancy_bwd_source() {
    using K = Cfg<C>;
    int n = -1;
    if constexpr (K::MFMA && source_window(C))
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, grand_bwd_source_kernel<C, false, true>, K::NT, K::lds_bytes(2, 4, 1));
    else
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, grand_bwd_source_kernel<C>, K::NT, K::lds_bytes(2));
    return n;
}
int gadapt_occupancy_bwd_source_c(int c) { GADAPT_DISPATCH_C(c, occupancy_bwd_source<CC>()); }
