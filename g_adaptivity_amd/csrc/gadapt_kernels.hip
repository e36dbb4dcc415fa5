// gadapt_kernels.hip - fused GRAND attention-diffusion layer for gfx950 (MI355X).
//
// Three kernels carry the hot path:
//   grand_fwd_kernel<C>        x' = x + dt (sum_j alpha_ij x_j - x)            (target-centric)
//   grand_bwd_target_kernel<C> d(score), dP, weight-gradient partials, dxd     (target-centric)
//   grand_bwd_source_kernel<C> g_out = dxd + sum over out-edges                (source-centric)
// All three share one shape: a 256-thread workgroup owns tiles of TM consecutive nodes, C/8 lanes
// cover one node (two float4 per lane for C >= 32, one for C < 32: a gathered neighbour row is one or
// two coalesced reads), the tile's CSR slice sits in LDS, and workgroups of one XCD walk a contiguous
// node range.  The forward and the target pass keep the x rows of tiles t-1..t+1 in an LDS ring and
// gather from it when the tile's neighbours all lie there (per-tile metadata built with the graph);
// the [TM,C]x[C,C] projections run on the bf16 matrix cores with a three-piece split of both operands
// (fp32 accuracy, C >= 32) and on the VALU for C < 32.  Hidden 64 on row-major mesh batches runs the wide forward
// kernel of gadapt_wide.inc instead of grand_fwd_kernel.
//
// Round 3 added the kernels of the two bottom backward layers behind the identity (zero-pad) encoder:
//   grand_bwd_target_compact_kernel  layer 0 on the compact [N,4] input: the 4 x 4 corner of dA, one node per lane
//   grand_bwd_target_kernel<..,D4>   the layer above it: dxd as [N,4], dP A[:, :4] on the vector ALU, no projection phase
//   grand_bwd_source4_kernel         ... and its source pass: g_out as [N,4], three waves per SIMD
// and the per-workgroup partial sums of d dt / d score_scale (layer_params_reduce_block: no float atomics).
//
// Arithmetic follows /root/reference/src/GRAND_plus.py:225-343 and src/GNN.py:273-291 in the
// (A, p0) formulation described in include/gadapt_hip.h.

#include "gadapt_internal.h"
#include <atomic>

// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
int gadapt_fail_(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
int gadapt_check_launch_(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e)); return GADAPT_E_LAUNCH; }
    return GADAPT_OK;
}
extern "C" const char* gadapt_last_error(void) { return g_err; }
extern "C" int gadapt_abi_version(void) { return 9; }   // 9: gadapt_graph::wide_half_deg_t, gadapt_wide_window_host(step); 8 (round 6): the wide backward, the strided tile walk and their graph fields / host helpers are gone (measured level twice: docs/measurements.md F, G); + fused training-step entry points
extern "C" int gadapt_clear_error(void) { g_err[0] = 0; return (int)hipGetLastError(); }
extern "C" int gadapt_supported_hidden_dim(int c) {
    return c == 4 || c == 8 || c == 16 || c == 32 || c == 64 || c == 128;
}

// ------------------------------------------------------------------------------------------------
// optional per-kernel timing (bench/roofline only): HIP events on the launch stream around every
// hot-kernel launch.  Off by default; when off the launch path touches none of this.
// ------------------------------------------------------------------------------------------------
struct ProfRec { int id, variant; hipEvent_t a, b; };
// Launches come from more than one host thread (forward: the Python thread, backward: autograd's worker thread), so the
// record list is guarded; the flag is read on every launch and stays a relaxed atomic.
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
ProfScope::ProfScope(int id, hipStream_t s, int variant) : st(s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    ProfRec r{id, variant, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    eb = r.b;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(r);
    idx = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() { if (idx >= 0) (void)hipEventRecord(eb, st); }
#ifdef GADAPT_STAMPS
unsigned long long* g_stamp_buf = nullptr;
extern "C" int gadapt_debug_set_stamp_buffer(void* p) { g_stamp_buf = static_cast<unsigned long long*>(p); return 0; }
#endif
extern "C" int gadapt_profile_enable(int on) { g_prof_on.store(on != 0, std::memory_order_relaxed); return GADAPT_OK; }
extern "C" int gadapt_profile_read(int kernel_id, double* total_ms, int* count) {
    if (!total_ms || !count) return fail(GADAPT_E_BADARG, "profile_read: null pointer");
    double tot = 0.0; int n = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        if (r.id != kernel_id) continue;
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { tot += ms; ++n; }
    }
    *total_ms = tot; *count = n;
    return GADAPT_OK;
}
extern "C" int gadapt_profile_samples(int kernel_id, double* out_ms, int cap) {
    if (!out_ms || cap < 0) return fail(GADAPT_E_BADARG, "profile_samples: bad argument");
    int n = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        if (r.id != kernel_id) continue;
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && n < cap) out_ms[n++] = ms;
    }
    return n;
}
extern "C" int gadapt_profile_variants(int kernel_id, int* out, int cap) {
    if (!out || cap < 0) return fail(GADAPT_E_BADARG, "profile_variants: bad argument");
    int n = 0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        if (r.id != kernel_id) continue;
        float ms = 0.f;                                         // same filter as gadapt_profile_samples: entries stay aligned
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && n < cap) out[n++] = r.variant;
    }
    return n;
}
// An event pair around a launch also times the dispatch of that launch.  gadapt_profile_calibrate brackets, the same
// way, n times ONE empty launch (kernel id 3) and n times TWO consecutive empty launches (id 4): with e = what one
// empty launch occupies, p1 = D + e and p2 = D + 2e, so the dispatch share of a pair is D = 2 p1 - p2.
__global__ void profile_empty_kernel() {}
extern "C" int gadapt_profile_calibrate(int n, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {
        {
            ProfScope prof(3, st);
            hipLaunchKernelGGL(profile_empty_kernel, dim3(1), dim3(64), 0, st);
        }
        {
            ProfScope prof(4, st);
            hipLaunchKernelGGL(profile_empty_kernel, dim3(1), dim3(64), 0, st);
            hipLaunchKernelGGL(profile_empty_kernel, dim3(1), dim3(64), 0, st);
        }
    }
    return check_launch("profile_empty_kernel");
}
extern "C" int gadapt_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    return GADAPT_OK;
}

#include "gadapt_small.inc"

// Block backward: the source pass writes g_out over the dxd rows it has just read (same row, same lanes: read-then-write), so a
// layer pair touches two [N,C] buffers instead of three.  GADAPT_BWD_INPLACE=0 / gadapt_debug_set_backward_inplace(0): separate buffers.
static std::atomic<int> g_bwd_inplace{-1};
static bool bwd_inplace_enabled() {
    int v = g_bwd_inplace.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("GADAPT_BWD_INPLACE");
        v = (e && e[0] == '0') ? 0 : 1;
        g_bwd_inplace.store(v, std::memory_order_relaxed);
    }
    return v == 1;
}
extern "C" int gadapt_debug_set_backward_inplace(int on) { g_bwd_inplace.store(on ? 1 : 0, std::memory_order_relaxed); return GADAPT_OK; }

// one layer's backward: target pass, then - when a gradient is to be passed on - the source pass
static int launch_bwd(int c, const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha, const float* a, const float* p0,
                      const float* lp, float* edge_ws, float* dxd, float* slab, int accumulate, float* sums_out, float* sums_sc_out, float* g_out,
                      int residual_only, int g_cols, int x_cols, int out4, int g_stride, int sums_partials, hipStream_t st) {
    if (int rc = gadapt_launch_bwd_target_c(c, g, x_in, g_in, alpha, a, lp, edge_ws, dxd, slab, accumulate, sums_out, sums_sc_out, g_out != nullptr,
                                            residual_only, g_cols, x_cols, out4, g_stride, sums_partials, st)) return rc;
    if (!g_out) return GADAPT_OK;
    return gadapt_launch_bwd_source_c(c, g, x_in, g_in, edge_ws, dxd, a, p0, g_out, g_cols, out4, st);
}

static int check_graph(const gadapt_graph* g, int c) {
    if (!g || g->n_nodes <= 0 || g->n_edges < 0 || !g->rowptr_t || !g->col_t) return fail(GADAPT_E_BADARG, "bad graph");
    for (int k = 0; k < 3; ++k) if (!g->meta_t[k] || !g->meta_s[k]) return fail(GADAPT_E_BADARG, "graph without tile metadata (gadapt_tile_meta_host)");
    if ((int64_t)g->n_nodes * c * 4 >= ((int64_t)1 << 32)) return fail(GADAPT_E_BADARG, "n_nodes*C*4 must stay below 4 GiB (32-bit row offsets)");
    return GADAPT_OK;
}

// Diagnostic: what the runtime says about residency (blocks per CU) of the three hot kernels for hidden size c.
extern "C" int gadapt_debug_occupancy(int c, int* out3) {
    if (!out3) return fail(GADAPT_E_BADARG, "occupancy: null");
    if (!gadapt_supported_hidden_dim(c)) return fail(GADAPT_E_BADARG, "occupancy: unsupported hidden_dim");
    out3[0] = gadapt_occupancy_fwd_c(c); out3[1] = gadapt_occupancy_bwd_target_c(c); out3[2] = gadapt_occupancy_bwd_source_c(c);
    return GADAPT_OK;
}

extern "C" int gadapt_layer_forward(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                                    const float* layer_params, float* alpha_out, int residual_only, int c, void* stream) {
    if (int rc = check_graph(g, c)) return rc;
    if (!x_in || !x_out || !a || !p0 || !layer_params || x_in == x_out) return fail(GADAPT_E_BADARG, "layer_forward: null or aliased pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return gadapt_launch_fwd_c(c, g, x_in, x_out, a, p0, layer_params, alpha_out, residual_only, 0, nullptr, st);
}

extern "C" int gadapt_backward_slab_rows(int64_t n_nodes, int c) {
    if (n_nodes <= 0) return fail(GADAPT_E_BADARG, "slab_rows: bad node count");
    return gadapt_slab_rows_c(n_nodes, c);
}
extern "C" int64_t gadapt_backward_slab_floats(int64_t n_nodes, int c) {
    const int rows = gadapt_backward_slab_rows(n_nodes, c);
    return rows < 0 ? rows : (int64_t)rows * (c * c + c);
}

extern "C" int gadapt_layer_backward(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha,
                                     const float* a, const float* p0, const float* layer_params, float* edge_ws, float* dxd_ws,
                                     float* slab, int accumulate, float* sums_out, float* g_out, int residual_only, int c, void* stream) {
    if (int rc = check_graph(g, c)) return rc;
    if (!x_in || !g_in || !alpha || !a || !p0 || !layer_params || !edge_ws || !dxd_ws || !slab)
        return fail(GADAPT_E_BADARG, "layer_backward: null pointer");
    if (!g->tpos_s || (g_out && (!g->rowptr_s || !g->col_s))) return fail(GADAPT_E_BADARG, "layer_backward: source CSR missing");
    if (g_out == g_in || g_out == dxd_ws) return fail(GADAPT_E_BADARG, "layer_backward: g_out aliases an input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // {d dt, d score_scale} side by side, atomically accumulated
    return launch_bwd(c, g, x_in, g_in, alpha, a, p0, layer_params, edge_ws, dxd_ws, slab, accumulate, sums_out, sums_out ? sums_out + 1 : nullptr, g_out,
                      residual_only, 0, 0, 0, 0, 0, st);
}

extern "C" int gadapt_layer_params_reduce(const float* partials, int n_rows, int n_layers, int want_d_scale, float* d_layer_params, void* stream) {
    if (!partials || !d_layer_params || n_rows <= 0 || n_layers <= 0) return fail(GADAPT_E_BADARG, "layer_params_reduce: bad argument");
    hipLaunchKernelGGL(layer_params_reduce_kernel, dim3(2 * n_layers), dim3(256), 0, static_cast<hipStream_t>(stream), partials, n_rows, n_layers,
                       want_d_scale, d_layer_params);
    return check_launch("layer_params_reduce_kernel");
}

extern "C" int gadapt_slab_reduce(const float* slab, int n_rows, float* scratch, float* d_a, float* d_p0, int c, void* stream) {
    if (!slab || n_rows <= 0 || !scratch || !d_a || !d_p0 || !gadapt_supported_hidden_dim(c)) return fail(GADAPT_E_BADARG, "slab_reduce: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int row_len = c * c + c;
    hipLaunchKernelGGL(slab_reduce1_kernel, dim3((row_len + 255) / 256, GADAPT_SLAB_CHUNKS), dim3(256), 0, st, slab, scratch,
                       n_rows, row_len);
    hipLaunchKernelGGL(slab_reduce2_kernel, dim3((row_len + 255) / 256), dim3(256), 0, st, scratch, d_a, d_p0, c);
    return check_launch("slab_reduce");
}

// slab -> d_wq | d_bq | d_wk | d_bk in two launches (first-level partial sums, then second level + chain rule together)
extern "C" int gadapt_slab_reduce_coeffs_backward(const float* slab, int n_rows, float* scratch, const float* wq, const float* bq,
                                                  const float* wk, float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c, void* stream,
                                                  const float* lp_partials, int n_layers, int want_d_scale, float* d_layer_params) {
    if (!slab || n_rows <= 0 || !scratch || !wq || !bq || !wk || !d_wq || !d_bq || !d_wk || !d_bk || !gadapt_supported_hidden_dim(c))
        return fail(GADAPT_E_BADARG, "slab_reduce_coeffs_backward: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int row_len = c * c + c;
    const int nbx = (row_len + 255) / 256;
    if (lp_partials && (n_layers <= 0 || !d_layer_params)) return fail(GADAPT_E_BADARG, "slab_reduce_coeffs_backward: layer-parameter partials without a destination");
    hipLaunchKernelGGL(slab_reduce1_kernel, dim3(nbx + (lp_partials ? 2 * n_layers : 0), GADAPT_SLAB_CHUNKS), dim3(256), 0, st, slab, scratch, n_rows, row_len,
                       nbx, lp_partials, n_layers, want_d_scale, d_layer_params);
    const int lds = (c * (c + 1) + c) * 4;
    int blocks = (2 * c * c + 2 * c + 1023) / 1024;
    if (blocks > 33) blocks = 33;                               // every workgroup repeats the second-level sums
#define CASE_C(CC) case CC: allow_lds(reduce2_coeffs_bwd_kernel<CC>, lds); \
        hipLaunchKernelGGL(reduce2_coeffs_bwd_kernel<CC>, dim3(blocks), dim3(1024), lds, st, scratch, wq, bq, wk, d_wq, d_bq, d_wk, d_bk); break;
    switch (c) { CASE_C(4) CASE_C(8) CASE_C(16) CASE_C(32) CASE_C(64) CASE_C(128) default: break; }
#undef CASE_C
    return check_launch("slab_reduce_coeffs_backward");
}

extern "C" int gadapt_coeffs_forward(const float* wq, const float* bq, const float* wk, float* a_out, float* p0_out, int c, void* stream) {
    if (!wq || !bq || !wk || !a_out || !p0_out || c <= 0) return fail(GADAPT_E_BADARG, "coeffs_forward: bad argument");
    if (!gadapt_supported_hidden_dim(c)) return fail(GADAPT_E_BADARG, "coeffs_forward: unsupported hidden_dim");
    const int n = c * c + c;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define CASE_C(CC) case CC: hipLaunchKernelGGL(coeffs_fwd_kernel<CC>, dim3((n + 255) / 256), dim3(256), 0, st, wq, bq, wk, a_out, p0_out); break;
    switch (c) { CASE_C(4) CASE_C(8) CASE_C(16) CASE_C(32) CASE_C(64) CASE_C(128) default: break; }
#undef CASE_C
    return check_launch("coeffs_fwd_kernel");
}
extern "C" int gadapt_coeffs_backward(const float* wq, const float* bq, const float* wk, const float* d_a, const float* d_p0,
                                      float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c, void* stream) {
    if (!wq || !bq || !wk || !d_a || !d_p0 || !d_wq || !d_bq || !d_wk || !d_bk || c <= 0) return fail(GADAPT_E_BADARG, "coeffs_backward: bad argument");
    if (!gadapt_supported_hidden_dim(c)) return fail(GADAPT_E_BADARG, "coeffs_backward: unsupported hidden_dim");
    const int n = 2 * c * c + 2 * c;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define CASE_C(CC) case CC: hipLaunchKernelGGL(coeffs_bwd_kernel<CC>, dim3((n + 255) / 256), dim3(256), 0, st, wq, bq, wk, d_a, d_p0, d_wq, d_bq, d_wk, d_bk); break;
    switch (c) { CASE_C(4) CASE_C(8) CASE_C(16) CASE_C(32) CASE_C(64) CASE_C(128) default: break; }
#undef CASE_C
    return check_launch("coeffs_bwd_kernel");
}

struct wq_t { const float* wq; const float* bq; const float* wk; float* a; float* p0; int c; };
static int launch_encode(const float* feats, int f0, const float* e1, const float* e2, const float* w, const float* b, float* x0,
                         int64_t n_nodes, int c, void* stream, const void* coeffs = nullptr) {
    const int f = f0 + (e1 ? 1 : 0) + (e2 ? 1 : 0);
    if (c % 4 || c > 256 || 256 % (c / 4) || (int64_t)c * f > GADAPT_ENC_MAX_WORDS)
        return fail(GADAPT_E_BADARG, "encode: need hidden_dim in {4,8,...,256} dividing 1024 and hidden_dim*in_dim <= 4096");
    const int rows_per_block = 256 / (c / 4);
    int64_t blocks = (n_nodes + 4 * rows_per_block - 1) / (4 * rows_per_block);    // four rows per thread and iteration
    if (blocks > 1024) blocks = 1024;                                              // resident set: the W^T table is staged once per block
    if (blocks < 1) blocks = 1;
    const wq_t* cf = static_cast<const wq_t*>(coeffs);
    EncArgs p{feats, f0, e1, e2, w, b, x0, n_nodes, f, c};
    if (!cf) {
        hipLaunchKernelGGL(encode_linear_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p);
        return check_launch("encode_linear_kernel");
    }
    const int cblocks = (cf->c * cf->c + cf->c + 255) / 256;
#define CASE_C(CC) case CC: hipLaunchKernelGGL(encode_coeffs_kernel<CC>, dim3((unsigned)blocks + cblocks), dim3(256), 0, static_cast<hipStream_t>(stream), \
                                                      p, (int)blocks, cf->wq, cf->bq, cf->wk, cf->a, cf->p0); break;
    switch (cf->c) { CASE_C(4) CASE_C(8) CASE_C(16) CASE_C(32) CASE_C(64) CASE_C(128)
                     default: return fail(GADAPT_E_BADARG, "encode_features_coeffs: unsupported hidden_dim"); }
#undef CASE_C
    return check_launch("encode_coeffs_kernel");
}
extern "C" int gadapt_encode_linear(const float* feats, const float* w, const float* b, float* x0, int64_t n_nodes, int f, int c, void* stream) {
    if (!feats || !w || !x0 || n_nodes <= 0 || f <= 0 || c <= 0) return fail(GADAPT_E_BADARG, "encode_linear: bad argument");
    return launch_encode(feats, f, nullptr, nullptr, w, b, x0, n_nodes, c, stream);
}
extern "C" int gadapt_encode_features(const float* x_comp, int dim, const float* f_tensor, const float* uu_tensor, const float* w,
                                      const float* b, float* x0, int64_t n_nodes, int c, void* stream) {
    if (!x_comp || !w || !x0 || n_nodes <= 0 || dim <= 0 || c <= 0) return fail(GADAPT_E_BADARG, "encode_features: bad argument");
    // the kernel reads "first extra" then "second extra": with only uu present it is the first one
    return launch_encode(x_comp, dim, f_tensor ? f_tensor : uu_tensor, f_tensor ? uu_tensor : nullptr, w, b, x0, n_nodes, c, stream);
}
extern "C" int gadapt_encode_features_coeffs(const float* x_comp, int dim, const float* f_tensor, const float* uu_tensor, const float* w,
                                             const float* b, float* x0, int64_t n_nodes, int c, const float* wq, const float* bq,
                                             const float* wk, float* a_out, float* p0_out, int c_conv, void* stream) {
    if (!x_comp || !w || !x0 || n_nodes <= 0 || dim <= 0 || c <= 0 || !wq || !bq || !wk || !a_out || !p0_out)
        return fail(GADAPT_E_BADARG, "encode_features_coeffs: bad argument");
    if (!gadapt_supported_hidden_dim(c_conv)) return fail(GADAPT_E_BADARG, "encode_features_coeffs: unsupported hidden_dim");
    const wq_t cf{wq, bq, wk, a_out, p0_out, c_conv};
    return launch_encode(x_comp, dim, f_tensor ? f_tensor : uu_tensor, f_tensor ? uu_tensor : nullptr, w, b, x0, n_nodes, c, stream, &cf);
}
extern "C" int gadapt_loss_forward(const float* pred, int64_t pred_stride, const float* target, int64_t n_rows, int d, int l1,
                                   float* seed, float* loss_out, float* scratch, void* stream) {
    if (!pred || !target || !seed || !loss_out || !scratch || n_rows <= 0 || d <= 0 || pred_stride < d)
        return fail(GADAPT_E_BADARG, "loss_forward: bad argument");
    int64_t blocks = (n_rows + 255) / 256;
    if (blocks > GADAPT_LOSS_BLOCKS) blocks = GADAPT_LOSS_BLOCKS;
    hipLaunchKernelGGL(loss_forward_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), pred, pred_stride, target,
                       n_rows, d, l1, seed, loss_out, scratch);
    return check_launch("loss_forward_kernel");
}
extern "C" int gadapt_loss_scratch_floats(void) { return GADAPT_LOSS_BLOCKS + 1; }

// ---- gradient exchange: the caller's RCCL communicator, ncclAllReduce resolved at run time (no link-time dependency: the process
// usually has an RCCL loaded already - torch ships one - and a second copy must not come in with this library)
#include <dlfcn.h>
extern "C" int gadapt_allreduce_flat(void* comm, float* bucket, int64_t n, int average, void* stream) {
    if (!comm || !bucket || n <= 0) return fail(GADAPT_E_BADARG, "allreduce_flat: null communicator / bucket or empty bucket");
    // ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static std::atomic<allreduce_fn> fn{nullptr};
    allreduce_fn f = fn.load(std::memory_order_acquire);
    if (!f) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // the copy the process already uses, if any
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return fail(GADAPT_E_RUNTIME, "allreduce_flat: no librccl.so in this process or on the library path");
        f = reinterpret_cast<allreduce_fn>(dlsym(h, "ncclAllReduce"));
        if (!f) return fail(GADAPT_E_RUNTIME, "allreduce_flat: librccl.so without ncclAllReduce");
        fn.store(f, std::memory_order_release);
    }
    constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0, NCCL_AVG = 4;          // rccl.h: ncclFloat, ncclSum, ncclAvg
    const int rc = f(bucket, bucket, (size_t)n, NCCL_FLOAT32, average ? NCCL_AVG : NCCL_SUM, comm, static_cast<hipStream_t>(stream));
    if (rc != 0) return fail(GADAPT_E_RUNTIME, "allreduce_flat: ncclAllReduce failed");
    return GADAPT_OK;
}

// ------------------------------------------------------------------------------------------------
// batch assembly on the device: sample rows of up to GADAPT_GATHER_MAX stacked per-sample fields -> the batch's node fields
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    const float* src[GADAPT_GATHER_MAX]; float* dst[GADAPT_GATHER_MAX]; int64_t row[GADAPT_GATHER_MAX];   // row: floats per sample
    const int64_t* idx; int n_fields, n_take;
};
__global__ __launch_bounds__(256) void gather_fields_kernel(GatherArgs p) {
    const int f = blockIdx.z, b = blockIdx.y;
    if (f >= p.n_fields) return;
    const int64_t row = p.row[f];
    const float* s = p.src[f] + p.idx[b] * row;
    float* d = p.dst[f] + (int64_t)b * row;
    if ((row & 3) == 0 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const int64_t n4 = row >> 2;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256)
            reinterpret_cast<float4*>(d)[e] = reinterpret_cast<const float4*>(s)[e];
    } else {
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < row; e += (int64_t)gridDim.x * 256) d[e] = s[e];
    }
}
extern "C" int gadapt_gather_fields(int n_fields, const float* const* src, float* const* dst, const int64_t* row_floats, const int64_t* idx,
                                    int n_take, void* stream) {
    if (n_fields <= 0 || n_fields > GADAPT_GATHER_MAX || !src || !dst || !row_floats || !idx || n_take <= 0)
        return fail(GADAPT_E_BADARG, "gather_fields: 1..GADAPT_GATHER_MAX fields, a sample index vector and a positive count");
    GatherArgs p{};
    int64_t longest = 0;
    for (int f = 0; f < n_fields; ++f) {
        if (!src[f] || !dst[f] || row_floats[f] <= 0) return fail(GADAPT_E_BADARG, "gather_fields: null field or empty row");
        p.src[f] = src[f]; p.dst[f] = dst[f]; p.row[f] = row_floats[f];
        longest = row_floats[f] > longest ? row_floats[f] : longest;
    }
    p.idx = idx; p.n_fields = n_fields; p.n_take = n_take;
    int bx = (int)((longest / 4 + 255) / 256);
    bx = bx < 1 ? 1 : (bx > 32 ? 32 : bx);
    hipLaunchKernelGGL(gather_fields_kernel, dim3(bx, n_take, n_fields), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("gather_fields_kernel");
}

extern "C" int gadapt_pad_columns(const float* g_phys, float* g_top, int64_t n_nodes, int d, int c, void* stream) {
    if (!g_phys || !g_top || n_nodes <= 0 || d <= 0 || d > c || c % 4) return fail(GADAPT_E_BADARG, "pad_columns: bad argument");
    const int64_t n = n_nodes * (c / 4);
    hipLaunchKernelGGL(pad_columns_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), g_phys, g_top, n_nodes, d, c);
    return check_launch("pad_columns_kernel");
}

extern "C" int gadapt_mesh_loss_seed(const float* x_top, const float* target, float* x_phys, float* g_top, float* loss_out,
                                     int64_t n_nodes, int d, int c, int l1, float grad_scale, void* stream) {
    if (!x_top || !target || !x_phys || !g_top || !loss_out || n_nodes <= 0 || d <= 0 || d > c) return fail(GADAPT_E_BADARG, "mesh_loss_seed: bad argument");
    const int64_t n = n_nodes * c;
    hipLaunchKernelGGL(mesh_loss_seed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x_top, target,
                       x_phys, g_top, loss_out, n_nodes, d, c, l1, grad_scale);
    return check_launch("mesh_loss_seed_kernel");
}

extern "C" int gadapt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return fail(GADAPT_E_BADARG, "adam_step: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
    return check_launch("adam_step_kernel");
}

extern "C" int gadapt_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int32_t* state, float grad_scale, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || !state) return fail(GADAPT_E_BADARG, "adam_step_dev: bad argument");
    hipLaunchKernelGGL(adam_step_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), param, grad,
                       exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, state, grad_scale);
    return check_launch("adam_step_dev_kernel");
}

// ------------------------------------------------------------------------------------------------
// L-step Euler block (GNN.py:273-291)
// ------------------------------------------------------------------------------------------------
static int block_forward(const gadapt_graph* g, float* x_all, int x0_cols, int n_layers, const float* a, int64_t a_stride,
                         const float* p0, int64_t p0_stride, const float* layer_params, float* alpha_all, float* x_top4,
                         int c, void* stream, const FwdExtra* extra) {
    if (int rc = check_graph(g, c)) return rc;
    if (!x_all || n_layers <= 0 || !a || !p0 || !layer_params) return fail(GADAPT_E_BADARG, "block_forward: bad argument");
    if (x0_cols != 0 && (x0_cols != 4 || n_layers < 2 || c < 8)) return fail(GADAPT_E_BADARG, "block_forward: compact x0 needs 4 columns, >= 2 layers, hidden >= 8");
    const size_t nc = (size_t)g->n_nodes * c;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        float* alpha_l = alpha_all ? alpha_all + (size_t)l * g->n_edges : nullptr;
        int rc;
        // extras of a fused training step, per layer: the node fields and the weights (in-kernel coefficients) for layer 0, the loss for
        // the last layer
        FwdExtra ex_l{};
        const FwdExtra* ex = nullptr;
        if (extra) {
            if (l == 0) { ex_l.fs = extra->fs; ex_l.x0c = extra->x0c; }
            if (l == 0) { ex_l.cw = extra->cw; ex_l.a_out = extra->a_out; ex_l.p0_out = extra->p0_out; }
            if (last) { ex_l.loss = extra->loss; ex_l.n_partials_out = extra->n_partials_out; }
            ex = &ex_l;
        }
        if ((l == 0 && x0_cols) || (last && x_top4))             // compact input and/or compact-only output
            rc = gadapt_launch_fwd_c(c, g, x_all + l * nc, (last && x_top4) ? nullptr : x_all + (l + 1) * nc, a + l * a_stride, p0 + l * p0_stride,
                                     layer_params + 2 * l, alpha_l, 0, l == 0 ? x0_cols : 0, last ? x_top4 : nullptr, st, ex);
        else
            rc = gadapt_layer_forward(g, x_all + l * nc, x_all + (l + 1) * nc, a + l * a_stride, p0 + l * p0_stride,
                                      layer_params + 2 * l, alpha_l, 0, c, stream);
        if (rc) return rc;
    }
    return GADAPT_OK;
}
extern "C" int gadapt_block_forward(const gadapt_graph* g, float* x_all, int x0_cols, int n_layers, const float* a, int64_t a_stride,
                                    const float* p0, int64_t p0_stride, const float* layer_params, float* alpha_all, float* x_top4,
                                    int c, void* stream) {
    return block_forward(g, x_all, x0_cols, n_layers, a, a_stride, p0, p0_stride, layer_params, alpha_all, x_top4, c, stream, nullptr);
}

// ------------------------------------------------------------------------------------------------
// fused training step (run_GNN.py:99-131 as 13 launches): forward with the node fields as layer-0 input and the loss in the last
// layer's launch; tail = slab sums + chain rule + Adam + the next step's composite coefficients
// ------------------------------------------------------------------------------------------------
extern "C" int gadapt_loss_partials_max(void) { return GADAPT_LOSS_PARTIALS_MAX; }
extern "C" int gadapt_forward_computes_coeffs(const gadapt_graph* g, int c) { return g ? gadapt_forward_computes_coeffs_c(g, c) : 0; }
extern "C" int gadapt_block_forward_loss(const gadapt_graph* g, float* x_all, const float* x_comp, int dim, const float* f_tensor, const float* uu_tensor,
                                         int n_layers, float* a, float* p0, const float* param, const float* layer_params, float* alpha_all, float* x_top4,
                                         const float* target, int d, int l1, float* seed, float* loss_partials, int c, void* stream) {
    if (!x_comp || dim < 1 || dim > 4 || dim + (f_tensor ? 1 : 0) + (uu_tensor ? 1 : 0) > 4)
        return fail(GADAPT_E_BADARG, "block_forward_loss: 1..4 coordinates, coordinates + extras <= 4 columns");
    // target == NULL: no loss - the evaluation forward on the node fields (inference.GraphedForward issues it as this one call)
    if (!x_top4 || (target && (!seed || !loss_partials || d < 1 || d > 4))) return fail(GADAPT_E_BADARG, "block_forward_loss: head rows; with a target: seed, partials, 1 <= d <= 4");
    if (!g || g->n_nodes <= 0) return fail(GADAPT_E_BADARG, "bad graph");
    int n_partials = 0;
    // x_all slot 0 starts with the compact [N,4] layer-0 input (written by the layer-0 launch for the layer-0 backward)
    if (param && !gadapt_forward_computes_coeffs_c(g, c))
        return fail(GADAPT_E_BADARG, "block_forward_loss: param given, but this graph / hidden size does not compute the coefficients in its layer-0 launch (gadapt_forward_computes_coeffs)");
    FwdExtra ex{FieldSrc{x_comp, f_tensor, uu_tensor, dim}, x_all,
                LossArgs{target, seed, loss_partials, d, l1 ? 1 : 0, 1.0f / (float)((int64_t)g->n_nodes * (d > 0 ? d : 1))}, &n_partials, param, a, p0};
    if (int rc = block_forward(g, x_all, 4, n_layers, a, 0, p0, 0, layer_params, alpha_all, x_top4, c, stream, &ex)) return rc;
    if (!target) return 0;
    if (n_partials <= 0 || n_partials > GADAPT_LOSS_PARTIALS_MAX) return fail(GADAPT_E_RUNTIME, "block_forward_loss: loss partial count out of range");
    return n_partials;
}

extern "C" int gadapt_step_tail(const float* slab, int n_rows, float* scratch, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int32_t* state, float grad_scale,
                                float* a_out, float* p0_out, const float* loss_partials, int n_loss_partials, float* loss_out, int64_t loss_count,
                                int c, void* stream) {
    const bool gradient_only = slab && !exp_avg && !exp_avg_sq;     // data parallel, first half: stop at the flat gradient
    if (!param || !grad || !gadapt_supported_hidden_dim(c) || (!gradient_only && (!state || !exp_avg || !exp_avg_sq || (!a_out != !p0_out))))
        return fail(GADAPT_E_BADARG, "step_tail: bad argument");
    if (slab && (n_rows <= 0 || !scratch)) return fail(GADAPT_E_BADARG, "step_tail: slab without row count / scratch");
    if (loss_partials && (n_loss_partials <= 0 || !loss_out || loss_count <= 0 || !slab))
        return fail(GADAPT_E_BADARG, "step_tail: loss partials need a count, a destination and the slab launch they ride in");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int row_len = c * c + c;
    int blocks = (2 * c * c + 2 * c + 1023) / 1024;                // one entry of the flat gradient / bucket per thread
    if (slab) {
        const int nbx = (row_len + 255) / 256;
        // (+ one workgroup: the step count for the Adam launch of this step, and the loss value)
        hipLaunchKernelGGL(slab_reduce1_kernel, dim3(nbx + 1, GADAPT_SLAB_CHUNKS), dim3(256), 0, st, slab, scratch, n_rows, row_len,
                           nbx, (const float*)nullptr, 0, 0, (float*)nullptr, loss_partials, n_loss_partials, loss_partials ? 1.0f / (float)loss_count : 0.f, loss_out,
                           gradient_only ? nullptr : state);     // (data parallel: the count advances with the Adam half)
    } else {
        hipLaunchKernelGGL(step_count_kernel, dim3(1), dim3(64), 0, st, state);
    }
    if (gradient_only) {                                            // second-level sums + chain rule, as gadapt_slab_reduce_coeffs_backward
        const int c2 = c * c, lds2 = (c * (c + 1) + c) * 4;
#define CASE_C(CC) case CC: allow_lds(reduce2_coeffs_bwd_kernel<CC>, lds2); \
        hipLaunchKernelGGL(reduce2_coeffs_bwd_kernel<CC>, dim3(blocks), dim3(1024), lds2, st, scratch, param, param + c2, param + c2 + c, \
                           grad, grad + c2, grad + c2 + c, grad + 2 * c2 + c); break;
        switch (c) { CASE_C(4) CASE_C(8) CASE_C(16) CASE_C(32) CASE_C(64) CASE_C(128) default: break; }
#undef CASE_C
        return check_launch("step_tail (gradient)");
    }
    TailArgs p{slab ? scratch : nullptr, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, grad_scale, state};
#define CASE_C(CC) case CC: { constexpr int lds = tail_lds_floats<CC>() * 4; allow_lds(step_tail_kernel<CC>, lds); \
        hipLaunchKernelGGL(step_tail_kernel<CC>, dim3(tail_blocks<CC>()), dim3(1024), lds, st, p); } break;
    switch (c) { CASE_C(4) CASE_C(8) CASE_C(16) CASE_C(32) CASE_C(64) CASE_C(128) default: break; }
#undef CASE_C
    if (int rc = check_launch("step_tail")) return rc;
    // the composite coefficients of the UPDATED weights for the next step's forward - unless that forward computes them itself
    // (a_out = p0_out = NULL: gadapt_forward_computes_coeffs)
    if (!a_out) return GADAPT_OK;
    const int c2 = c * c;
    return gadapt_coeffs_forward(param, param + c2, param + c2 + c, a_out, p0_out, c, stream);
}

extern "C" int gadapt_block_backward(const gadapt_graph* g, const float* x_all, int x0_cols, const float* alpha_all, const float* g_top, int g_top_cols, int n_layers,
                                     const float* a, int64_t a_stride, const float* p0, int64_t p0_stride, const float* layer_params,
                                     float* g_ws, float* dxd_ws, float* edge_ws, float* slab, float* d_layer_params, int want_d_scale, float* d_x0,
                                     int c, void* stream) {
    if (int rc = check_graph(g, c)) return rc;
    if (!x_all || !alpha_all || !g_top || n_layers <= 0 || !a || !p0 || !layer_params || !g_ws || !dxd_ws || !edge_ws || !slab)
        return fail(GADAPT_E_BADARG, "block_backward: bad argument");
    if (x0_cols != 0 && (x0_cols != 4 || n_layers < 2 || c < 8 || d_x0))
        return fail(GADAPT_E_BADARG, "block_backward: compact x0 needs 4 columns, >= 2 layers, hidden >= 8, no d_x0");
    const size_t nc = (size_t)g->n_nodes * c;
    const bool shared = (a_stride == 0);
    const int64_t slab_floats = gadapt_backward_slab_floats(g->n_nodes, c);
    if (slab_floats < 0) return (int)slab_floats;
    const int slab_rows = gadapt_backward_slab_rows(g->n_nodes, c);
    const float* g_cur = g_top;
    for (int l = n_layers - 1; l >= 0; --l) {
        float* g_next = (l == 0) ? d_x0 : g_ws + ((n_layers - 1 - l) & 1) * nc;
        float* slab_l = shared ? slab : slab + (size_t)l * slab_floats;
        const int accumulate = (shared && l != n_layers - 1) ? 1 : 0;
        // per-workgroup partials [2][L][G] (G = gadapt_backward_slab_rows: the grid of every target-pass launch), d dt block first
        float* d_dt = d_layer_params ? d_layer_params + (size_t)l * slab_rows : nullptr;
        float* d_sc = (d_layer_params && want_d_scale) ? d_layer_params + ((size_t)n_layers + l) * slab_rows : nullptr;
        const hipStream_t st = static_cast<hipStream_t>(stream);
        const int g_cols = (l == n_layers - 1) ? g_top_cols : 0, x_cols = (l == 0) ? x0_cols : 0;
        if (!g->tpos_s || (g_next && (!g->rowptr_s || !g->col_s))) return fail(GADAPT_E_BADARG, "block_backward: source CSR missing");
        // compact upstream gradient [N,g_top_cols] (top layer) / compact layer-0 input [N,4] (no d x0: checked in launch_bwd).
        // Layer 1 above a compact layer 0: that layer's backward contracts over the four live columns of its input, so all it
        // reads of this layer's g_out are columns 0..3 - this layer runs the 4-column pair (dxd and g_out as [N,4]).
        // (not for a two-layer block with learnable steps / temperature: layer 1 is then also the top layer, and the SUMS + GC + D4
        // instantiation spills at hidden 32 - that corner keeps the dense pair)
        const bool pair4 = x0_cols && c >= 8 && n_layers >= 2 && !(n_layers == 2 && g_top_cols > 0 && d_layer_params);
        const int out4 = (pair4 && l == 1) ? 1 : 0;
        const int g_stride = (pair4 && l == 0) ? 4 : 0;
        // dense layers that hand their result to the next layer of the block: dxd lives in the g_out buffer (see bwd_inplace_enabled)
        const bool inplace = bwd_inplace_enabled() && l > 0 && g_next && !out4;
        int rc = launch_bwd(c, g, x_all + l * nc, g_cur, alpha_all + (size_t)l * g->n_edges, a + l * a_stride, p0 + l * p0_stride, layer_params + 2 * l,
                            edge_ws, inplace ? g_next : dxd_ws, slab_l, accumulate, d_dt, d_sc, g_next, 0, g_cols, x_cols, out4, g_stride, 1, st);
        if (rc) return rc;
        g_cur = g_next;
    }
    return GADAPT_OK;
}
