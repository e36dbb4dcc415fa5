// gadapt_tu_fwd.hip - forward layer launches: x' = x + dt (sum_j alpha_ij x_j - x) (src/GRAND_plus.py:225-343 + src/GNN.py:288-291),
// tiled kernel for every hidden size and graph (gadapt_fwd.inc), wide kernel for hidden 64 on row-major mesh batches
// (gadapt_wide.inc).  One translation unit of libgadapt_hip.so (see gadapt_internal.h).
#include "gadapt_internal.h"
#include "gadapt_fwd.inc"
#include "gadapt_wide.inc"

// The wide kernels take over for hidden size 64 when the graph qualifies; GADAPT_WIDE=0 in the environment keeps the tiled
// kernels (A/B runs and the tests of the tiled path).
static bool wide_enabled() {
    static const bool on = [] { const char* e = getenv("GADAPT_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}
static inline int wide_grid(int n_steps) {
    int g = (n_steps + 7) & ~7;
    if (g > 256) g = 256;                                       // one 512-thread workgroup per CU
    return g;
}
// Four-wave workgroups on steps of 128 nodes (wide::W<false, 4>): when the 256-node steps would leave half of the CUs without a
// workgroup and the graph qualifies for the narrower window (gadapt_graph::wide_half_deg_t)
static bool wide_half(const gadapt_graph* g) {
    return g->wide_deg_t > 0 && g->wide_half_deg_t > 0 && (g->n_nodes + wide::STEP - 1) / wide::STEP <= 128;
}
static int launch_wide_fwd(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                           const float* lp, float* alpha_out, int residual_only, int x_cols, float* x_top4, hipStream_t st, const FwdExtra* ex) {
    const bool big = g->wide_deg_t <= 0;                        // 512-row window (meshes with up to 128 nodes per row)
    const bool half = !big && wide_half(g);
    const int step = half ? wide::STEP_HALF : wide::STEP, nwv = half ? 4 : 8;
    const int n_steps = (g->n_nodes + step - 1) / step;
    wide::FwdArgs p{x_in, x_out, a, p0, lp, g->ell_t, g->rowptr_t, alpha_out, g->n_nodes, n_steps, residual_only,
                    big ? g->wide_big_deg_t : (half ? g->wide_half_deg_t : g->wide_deg_t), nullptr, x_top4};
#ifdef GADAPT_STAMPS
    p.stamps = g_stamp_buf;
#endif
    if (ex) {
        if (x_cols) {
            p.fs = ex->fs; p.x0c = ex->x0c;
            if (!half) { p.cw = ex->cw; p.a_out = ex->a_out; p.p0_out = ex->p0_out; }
            else if (ex->cw) return fail(GADAPT_E_BADARG, "wide forward on four-wave workgroups: the coefficients are given (gadapt_forward_computes_coeffs)");
        }
        if (!x_out && !x_cols && ex->loss.target) { p.loss = ex->loss; if (ex->n_partials_out) *ex->n_partials_out = nwv * wide_grid(n_steps); }
    }
    ProfScope prof(0, st, (x_cols ? 2 : 0) | (x_out ? 0 : 4));
    auto go = [&](auto kern, int lds) { allow_lds(kern, lds); hipLaunchKernelGGL(kern, dim3(wide_grid(n_steps)), dim3(64 * nwv), lds, st, p); };
    const bool head = !x_out && !x_cols;                        // head-only output: its own instantiation (aggregates one chunk)
    if (big) {
        if (x_cols) go(wide::fwd_kernel<true, true>, wide::fwd_lds_bytes<true>());
        else if (head) go(wide::fwd_kernel<false, true, true>, wide::fwd_lds_bytes<true>());
        else go(wide::fwd_kernel<false, true>, wide::fwd_lds_bytes<true>());
    } else if (half) {
        if (x_cols) go(wide::fwd_kernel<true, false, false, 4>, wide::fwd_lds_bytes<false, 4>());
        else if (head) go(wide::fwd_kernel<false, false, true, 4>, wide::fwd_lds_bytes<false, 4>());
        else go(wide::fwd_kernel<false, false, false, 4>, wide::fwd_lds_bytes<false, 4>());
    } else {
        if (x_cols) go(wide::fwd_kernel<true, false>, wide::fwd_lds_bytes<false>());
        else if (head) go(wide::fwd_kernel<false, false, true>, wide::fwd_lds_bytes<false>());
        else go(wide::fwd_kernel<false, false>, wide::fwd_lds_bytes<false>());
    }
    return check_launch("wide::fwd_kernel");
}

static bool wide_takes(const gadapt_graph* g) {
    return g->ell_t && (g->wide_deg_t > 0 || (g->wide_big_deg_t > 0 && g->wide_big_deg_t <= 7)) && wide_enabled();
}
// 1: the layer-0 launch of a fused training step on this graph at this hidden size is the wide kernel, which computes the composite
// coefficients itself (FwdExtra::cw); 0: they must be given (a coefficient launch ends the step)
int gadapt_forward_computes_coeffs_c(const gadapt_graph* g, int c) { return (c == 64 && g && wide_takes(g) && !wide_half(g)) ? 1 : 0; }

template <int C> static int launch_fwd(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                                       const float* lp, float* alpha_out, int residual_only, int x_cols, float* x_top4, hipStream_t st, const FwdExtra* ex) {
    using K = Cfg<C>;
    if (x_cols != 0 && x_cols != 4) return fail(GADAPT_E_BADARG, "compact layer input: 4 columns");
    if constexpr (C == 64) {
        if (wide_takes(g))
            return launch_wide_fwd(g, x_in, x_out, a, p0, lp, alpha_out, residual_only, x_cols, x_top4, st, ex);
    }
    FwdArgs p{x_in, x_out, a, p0, lp, g->rowptr_t, g->col_t, meta_for<K::TM>(g->meta_t), alpha_out, g->n_nodes,
              (g->n_nodes + K::TM - 1) / K::TM, residual_only, g->n_edges, nullptr, x_top4};
#ifdef GADAPT_STAMPS
    p.stamps = g_stamp_buf;
#endif
    const dim3 grid(grid_for(p.n_tiles, resident_blocks<C>(GADAPT_FWD_MAX_BLOCKS)));
    if (ex) {
        if (x_cols) { p.fs = ex->fs; p.x0c = ex->x0c; }
        if (!x_out && x_top4 && ex->loss.target) { p.loss = ex->loss; if (ex->n_partials_out) *ex->n_partials_out = (int)grid.x * K::NW; }
    }
    ProfScope prof(0, st, (x_cols ? 2 : 0) | (x_out ? 0 : 4));
    constexpr int lds = K::lds_bytes(0, K::RING + 1);
    if (x_cols) {
        allow_lds(grand_fwd_kernel<C, true>, lds);
        hipLaunchKernelGGL((grand_fwd_kernel<C, true>), grid, dim3(K::NT), lds, st, p);
    } else {
        allow_lds(grand_fwd_kernel<C>, lds);
        hipLaunchKernelGGL(grand_fwd_kernel<C>, grid, dim3(K::NT), lds, st, p);
    }
    return check_launch("grand_fwd_kernel");
}

int gadapt_launch_fwd_c(int c, const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0, const float* lp,
                        float* alpha_out, int residual_only, int x_cols, float* x_top4, hipStream_t st, const FwdExtra* extra) {
    GADAPT_DISPATCH_C(c, launch_fwd<CC>(g, x_in, x_out, a, p0, lp, alpha_out, residual_only, x_cols, x_top4, st, extra));
}

template <int C> static int occupancy_fwd() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, grand_fwd_kernel<C>, Cfg<C>::NT, Cfg<C>::lds_bytes(0, Cfg<C>::RING + 1));
    return n;
}
int gadapt_occupancy_fwd_c(int c) { GADAPT_DISPATCH_C(c, occupancy_fwd<CC>()); }
