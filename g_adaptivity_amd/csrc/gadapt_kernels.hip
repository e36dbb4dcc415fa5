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

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "gadapt_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GADAPT_MAXD 8           // in/out degree handled from registers; larger rows take the loop path
// minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument)
#ifndef GADAPT_WAVES_FWD
#define GADAPT_WAVES_FWD 2
#endif
#ifndef GADAPT_WAVES_BWD_T
#define GADAPT_WAVES_BWD_T 2
#endif
#ifndef GADAPT_WAVES_BWD_S
#define GADAPT_WAVES_BWD_S 2
#endif
// Hidden sizes from here on give the backward kernels one wave per SIMD (512 registers): at C = 128 the dA accumulators
// (64) + projection blocks (32) + row buffers do not fit 256 registers and the spill traffic costs more than the second
// resident workgroup brings.
#ifndef GADAPT_ONE_WAVE_C
#define GADAPT_ONE_WAVE_C 128
#endif
// Hidden sizes from here on would use 512-thread workgroups (Cfg::NT).  Measured at hidden 128 (-DGADAPT_WIDE_WG_C=128
// -DGADAPT_ONE_WAVE_C=256: the per-wave shares of the dA / projection blocks halve, two waves per SIMD): the backward kernels
// still spill (52 / 47 registers) and BASELINE config 4 ran 16.1k meshes/s against 17.8k with 256 threads at one wave per
// SIMD (forward 79 vs 59 us, target 142 vs 147, source 106 vs 93) - not adopted, the geometry stays parametrised.
#ifndef GADAPT_WIDE_WG_C
#define GADAPT_WIDE_WG_C 1024
#endif
// Forward at hidden 128 as ONE workgroup per CU at one wave per SIMD with resident weight fragments, the rolling LDS window
// (Cfg::RING) and staging two tiles ahead - what made the hidden-128 backward kernels faster.  Measured: 62.4 us against 60.4 us
// for two workgroups per CU gathering through L2 (without the window: 68 against 61).  Off.
#ifndef GADAPT_FWD_ONE_WAVE
#define GADAPT_FWD_ONE_WAVE 0
#endif
#ifndef GADAPT_SLAB_CHUNKS
#define GADAPT_SLAB_CHUNKS 8    // second-level partials of the slab reduction (<= 32: the scratch the callers allocate).
                                // 32 / 16 / 8 chunks: first level 5.1 / 5.5 / 5.0 us, second level + chain rule 12.1 / 8.8 / 6.6 us
#endif
#ifndef GADAPT_FPL
#define GADAPT_FPL 8            // floats per lane at hidden >= 32.  16 (four lanes per node at hidden 64) measured: the row buffers
                                // double, forward / target pass spill 36 / 117 registers (27.4 / 62.4 us), source pass 28.4 vs 26.2 us
#endif
#ifndef GADAPT_T_RING_MAX_C
#define GADAPT_T_RING_MAX_C 128   // target pass: largest hidden size that keeps the rolling LDS window of x rows
#endif
#ifndef GADAPT_BWD_JIT_B_C
#define GADAPT_BWD_JIT_B_C 1024  // backward kernels rebuild the projection fragments per tile from this hidden size on
#endif
#ifndef GADAPT_BWD_ONE_PER_CU
#define GADAPT_BWD_ONE_PER_CU 1
#endif
#ifndef GADAPT_T_PREFETCH_MAX_C
#define GADAPT_T_PREFETCH_MAX_C 128  // target pass: largest hidden size that requests the next tile one tile ahead
#endif
#ifndef GADAPT_DA_IN_SOURCE
#define GADAPT_DA_IN_SOURCE 0   // 1: at hidden 32 / 64 the source pass accumulates dA / dp0 when one follows the target pass
                                // (dA = sum_j x_j y_j^T).  Measured (64x64 b32 C64): target pass 39.1 -> 27.7 us dense / 32.9 ->
                                // 25.8 us compact-g (206 instead of 255 registers, no slab flush); source pass 28.3 -> 41.3 / 24.5
                                // -> 36.5 us (own x rows staged per tile, projection fragments rebuilt per tile to stay under 256
                                // registers; with resident fragments the dense variant spills 25 registers: 51 us).  Step 0.365
                                // against 0.356 ms: the dA phase costs more than it frees wherever it runs.
#endif
#ifndef GADAPT_T_STREAM
#define GADAPT_T_STREAM 0        /* target pass: own g rows and dxd rows non-temporal.  Measured: the target pass gains 0.3 us, the source pass
                                    that follows LOSES 6 us (24.9 -> 31.0 at hidden 64): it reads dxd, and a normally written dxd is still in
                                    the caches (L2 / Infinity Cache) when it does.  Off. */
#endif
#ifndef GADAPT_S_STREAM_DXD
#define GADAPT_S_STREAM_DXD 0
#endif
#ifndef GADAPT_S_STREAM
#define GADAPT_S_STREAM 2        /* windowed source pass: dxd reads and g_out writes (1), and the x slab reads (2), non-temporal: they pass through
                                    once, the g rows the other workgroups gather should stay in L2 (hidden 128: 70.6 -> 69.1 -> 66.9 us) */
#endif
#ifndef GADAPT_S_ALTERNATE
#define GADAPT_S_ALTERNATE 0
#endif
#ifndef GADAPT_T_ALTERNATE
#define GADAPT_T_ALTERNATE 1     /* target pass: every other workgroup walks its chunk backwards (see the kernel) */
#endif
#ifndef GADAPT_T_DIFF
#define GADAPT_T_DIFF 1           /* target pass: softmax backward on the differences x_k - x_i (see consume() in gadapt_bwd_target.inc) */
#endif
#ifndef GADAPT_DA_UNROLL
#define GADAPT_DA_UNROLL 2      // k-steps of the dA loop unrolled together
#endif
// Softmax arithmetic: 1 = expf / IEEE division (<= 1 ulp each), 0 = v_exp_f32 of a rounded product and v_rcp_f32.
// The approximate forms leave alpha with ~4x the rounding error of the reference's exp / true division; harmless for
// the coordinates (2e-7 either way) but visible in parameter gradients that are the small remainder of large cancelling
// sums (64x64, 6 layers, hidden 128: 3.4e-4 against the fp64 oracle with them, 1.4e-4 without, fp32 reference path 1.0e-4).
// Cost of the exact forms: +0.6 us per forward launch (17.1 -> 17.7 us), nothing measurable elsewhere.
#ifndef GADAPT_PRECISE_SOFTMAX
#define GADAPT_PRECISE_SOFTMAX 1
#endif
__device__ __forceinline__ float sm_exp(float x) {
#if GADAPT_PRECISE_SOFTMAX
    return expf(x);
#else
    return __expf(x);
#endif
}
__device__ __forceinline__ float sm_rcp(float x) {
#if GADAPT_PRECISE_SOFTMAX
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}

// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e)); return GADAPT_E_LAUNCH; }
    return GADAPT_OK;
}
extern "C" const char* gadapt_last_error(void) { return g_err; }
extern "C" int gadapt_abi_version(void) { return 7; }   // 7: + strided tile walk (gadapt_graph.t_strips, tile_meta_strided_host), small-mesh entry points with mesh_eptr; 6: + wide backward (gadapt_graph.xpos_t/xpos_s, ell_cross_host, block_backward_ws; round 5)
extern "C" int gadapt_clear_error(void) { g_err[0] = 0; return (int)hipGetLastError(); }
extern "C" int gadapt_supported_hidden_dim(int c) {
    return c == 4 || c == 8 || c == 16 || c == 32 || c == 64 || c == 128;
}

// ------------------------------------------------------------------------------------------------
// optional per-kernel timing (bench/roofline only): HIP events on the launch stream around every
// hot-kernel launch.  Off by default; when off the launch path touches none of this.
// ------------------------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>
struct ProfRec { int id, variant; hipEvent_t a, b; };
// Launches come from more than one host thread (forward: the Python thread, backward: autograd's worker thread), so the
// record list is guarded; the flag is read on every launch and stays a relaxed atomic.
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
struct ProfScope {
    hipStream_t st; int idx = -1;
    // variant: bit 0 = compact upstream gradient, bit 1 = compact layer input, bit 2 = head-only output (launchers below)
    hipEvent_t eb = nullptr;
    ProfScope(int id, hipStream_t s, int variant = 0) : st(s) {
        if (!g_prof_on.load(std::memory_order_relaxed)) return;
        ProfRec r{id, variant, nullptr, nullptr};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        (void)hipEventRecord(r.a, st);
        eb = r.b;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
        idx = (int)g_prof.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(eb, st); }
};
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

#include "gadapt_common.inc"
#include "gadapt_fwd.inc"
#include "gadapt_bwd_target.inc"
#include "gadapt_bwd_source.inc"
#include "gadapt_small.inc"
#include "gadapt_wide.inc"
#include "gadapt_wide_bwd.inc"
#include "gadapt_smallmesh.inc"

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

// Workgroups of a launch: a multiple of 8 (XCD groups), at most max_blocks (the resident set) unless that would
static inline int grid_for(int n_tiles, int max_blocks) {
    int g = (n_tiles + 7) & ~7;
    if (g > max_blocks) g = max_blocks;
    const int need = (((n_tiles + 63) / 64) + 7) & ~7;
    if (g < need) g = need;
    if (g < 8) g = 8;
    return g;
}
#ifndef GADAPT_FWD_MAX_BLOCKS
#define GADAPT_FWD_MAX_BLOCKS 512        /* 2 resident workgroups per CU (LDS ring: 4 tiles each) x 256 CUs */
#endif
#ifndef GADAPT_S_WINDOW
// Source pass with the LDS window of x rows (grand_bwd_source_kernel, WIN).  Measured (MI355X, same box): hidden 128, one wave
// per SIMD, every round trip exposed: 82.3 -> 79.3 us with whole-node steps (81.2 with half-node steps).  Hidden 64: 25.5 ->
// 32.2 us, with either step size - the window needs each workgroup on CONSECUTIVE tiles, and then the 64 workgroups of an XCD
// gather g rows from 192 distinct slabs at a time instead of 66 (interleaved order: 64 adjacent tiles): the first tile of a
// workgroup takes 10.4k cycles in the edge walk against 6.8k, and the L2 (4 MB per XCD) does not hold a slab until the
// neighbouring tile's step.  128-wide meshes (no tile windowed): 61 -> 80 us.  So: hidden 128 only.
#define GADAPT_S_WINDOW(C) ((C) == 128)
#endif
#ifndef GADAPT_BWD_S_MAX_BLOCKS
#define GADAPT_BWD_S_MAX_BLOCKS 512         /* the resident set (2 workgroups per CU): measured 27.8 vs 28.7 us with 1024 */
#endif
#ifndef GADAPT_BWD_T_MAX_BLOCKS
#define GADAPT_BWD_T_MAX_BLOCKS 512      /* target pass grid = slab row count */
#endif

// per-tile metadata pointer for this kernel's tile height (the graph carries one array per supported height)
template <int TM> static const int32_t* meta_for(const int32_t* const (&m)[3]) {
    static_assert(TM == 64 || TM == 128 || TM == 256, "tile heights with metadata");
    return m[TM == 64 ? 0 : (TM == 128 ? 1 : 2)];
}
// More than 48 KB of dynamic LDS needs the attribute set once per (device, kernel); the call costs several microseconds
// of host time, which is most of an eager small-graph forward, so it is remembered.
template <typename KernelT> static void allow_lds(KernelT k, int bytes) {
    if (bytes <= 48 * 1024) return;
    // forward launches come from the Python thread, backward launches from autograd's worker thread: the table is guarded
    static std::mutex mu;
    static std::vector<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void* f = reinterpret_cast<const void*>(k);
    std::lock_guard<std::mutex> lk(mu);
    for (auto& d : done) if (d.first == dev && d.second == f) return;
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) done.emplace_back(dev, f);
}

// The wide kernels (gadapt_wide.inc) take over for hidden size 64 when the graph qualifies; GADAPT_WIDE=0 in the
// environment keeps the tiled kernels (A/B runs and the tests of the tiled path).
static bool wide_enabled() {
    static const bool on = [] { const char* e = getenv("GADAPT_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}
static inline int wide_grid(int n_steps) {
    int g = (n_steps + 7) & ~7;
    if (g > 256) g = 256;                                       // one 512-thread workgroup per CU
    return g;
}
static int launch_wide_fwd(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                           const float* lp, float* alpha_out, int residual_only, int x_cols, float* x_top4, hipStream_t st) {
    const int n_steps = (g->n_nodes + wide::STEP - 1) / wide::STEP;
    const bool big = g->wide_deg_t <= 0;                        // 512-row window (meshes with up to 128 nodes per row)
    wide::FwdArgs p{x_in, x_out, a, p0, lp, g->ell_t, g->rowptr_t, alpha_out, g->n_nodes, n_steps, residual_only,
                    big ? g->wide_big_deg_t : g->wide_deg_t, nullptr, x_top4};
#ifdef GADAPT_STAMPS
    p.stamps = g_stamp_buf;
#endif
    ProfScope prof(0, st, (x_cols ? 2 : 0) | (x_out ? 0 : 4));
    auto go = [&](auto kern, int lds) { allow_lds(kern, lds); hipLaunchKernelGGL(kern, dim3(wide_grid(n_steps)), dim3(512), lds, st, p); };
    const bool head = !x_out && !x_cols;                        // head-only output: its own instantiation (aggregates one chunk)
    if (big) {
        if (x_cols) go(wide::fwd_kernel<true, true>, wide::fwd_lds_bytes<true>());
        else if (head) go(wide::fwd_kernel<false, true, true>, wide::fwd_lds_bytes<true>());
        else go(wide::fwd_kernel<false, true>, wide::fwd_lds_bytes<true>());
    } else {
        if (x_cols) go(wide::fwd_kernel<true, false>, wide::fwd_lds_bytes<false>());
        else if (head) go(wide::fwd_kernel<false, false, true>, wide::fwd_lds_bytes<false>());
        else go(wide::fwd_kernel<false, false>, wide::fwd_lds_bytes<false>());
    }
    return check_launch("wide::fwd_kernel");
}

// resident set of a launch: two 256-thread workgroups per CU, or one 512-thread workgroup (Cfg::NT)
template <int C> static constexpr int resident_blocks(int two_per_cu_default) { return Cfg<C>::NT == 512 ? 256 : two_per_cu_default; }
// ... and the backward kernels of hidden sizes that run one wave per SIMD (GADAPT_ONE_WAVE_C: 392 / 504 registers) fit ONE
// 256-thread workgroup per CU: 256 workgroups are the resident set, a 512-workgroup launch would run as two rounds (and
// flush twice as many slab rows).
template <int C> static constexpr int resident_blocks_fwd(int two_per_cu_default) {
    return (C >= GADAPT_ONE_WAVE_C && GADAPT_FWD_ONE_WAVE) ? 256 : resident_blocks<C>(two_per_cu_default);
}
template <int C> static constexpr int resident_blocks_bwd(int two_per_cu_default) {
    return (C >= GADAPT_ONE_WAVE_C && GADAPT_BWD_ONE_PER_CU) ? 256 : resident_blocks<C>(two_per_cu_default);
}
// ... and the target pass at hidden 32 (128-row tiles: ring + dP tile + slices = 87 KB of LDS) fits one workgroup per CU too
// (hipOccupancyMaxActiveBlocksPerMultiprocessor: forward / target / source = 2 / 1 / 2 at hidden 32, 2 / 2 / 2 at 64,
// 2 / 1 / 1 at 128, 3 / 3 / 3 at 8).
template <int C> static constexpr int resident_blocks_bwd_t(int two_per_cu_default) {
    return (C == 32 && GADAPT_BWD_ONE_PER_CU) ? 256 : resident_blocks_bwd<C>(two_per_cu_default);
}

template <int C> static int launch_fwd(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                                       const float* lp, float* alpha_out, int residual_only, int x_cols, float* x_top4, hipStream_t st) {
    using K = Cfg<C>;
    if (x_cols != 0 && x_cols != 4) return fail(GADAPT_E_BADARG, "compact layer input: 4 columns");
    if constexpr (C == 64) {
        if (g->ell_t && (g->wide_deg_t > 0 || (g->wide_big_deg_t > 0 && g->wide_big_deg_t <= 7)) && wide_enabled()) return launch_wide_fwd(g, x_in, x_out, a, p0, lp, alpha_out, residual_only, x_cols, x_top4, st);
    }
    FwdArgs p{x_in, x_out, a, p0, lp, g->rowptr_t, g->col_t, meta_for<K::TM>(g->meta_t), alpha_out, g->n_nodes,
              (g->n_nodes + K::TM - 1) / K::TM, residual_only, g->n_edges, nullptr, x_top4};
#ifdef GADAPT_STAMPS
    p.stamps = g_stamp_buf;
#endif
    ProfScope prof(0, st, (x_cols ? 2 : 0) | (x_out ? 0 : 4));
    constexpr int lds = K::lds_bytes(0, K::RING + 1);
    if (x_cols) {
        allow_lds(grand_fwd_kernel<C, true>, lds);
        hipLaunchKernelGGL((grand_fwd_kernel<C, true>), dim3(grid_for(p.n_tiles, resident_blocks_fwd<C>(GADAPT_FWD_MAX_BLOCKS))), dim3(K::NT), lds, st, p);
    } else {
        allow_lds(grand_fwd_kernel<C>, lds);
        hipLaunchKernelGGL(grand_fwd_kernel<C>, dim3(grid_for(p.n_tiles, resident_blocks_fwd<C>(GADAPT_FWD_MAX_BLOCKS))), dim3(K::NT), lds, st, p);
    }
    return check_launch("grand_fwd_kernel");
}
// The wide backward (gadapt_wide_bwd.inc) instead of the tiled target / source pair, for the layers it covers: hidden 64, a graph
// that qualifies for the wide kernels in both orientations and carries the ELL cross positions, a workspace from the caller.
// OFF by default: on the metric workload it measures level with the pair (edge + main kernel 12 + 46 us net against 31 + 23 us for the
// compact-gradient pair; docs/measurements.md F).  GADAPT_WIDE_BWD=1 in the environment or gadapt_debug_set_wide_backward(1)
// select it; tests/test_gpu_ops.py::test_wide_backward_matches_two_pass runs both on the same inputs.
static std::atomic<int> g_wide_bwd{-1};
static bool wide_bwd_enabled() {
    int v = g_wide_bwd.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("GADAPT_WIDE_BWD");
        v = (e && e[0] == '1') ? 1 : 0;
        g_wide_bwd.store(v, std::memory_order_relaxed);
    }
    return v == 1 && wide_enabled();
}
extern "C" int gadapt_debug_set_wide_backward(int on) { g_wide_bwd.store(on ? 1 : 0, std::memory_order_relaxed); return GADAPT_OK; }
extern "C" int64_t gadapt_wide_backward_ws_floats(int64_t n_nodes) {
    if (n_nodes <= 0) return fail(GADAPT_E_BADARG, "wide_backward_ws_floats: bad node count");
    return ((n_nodes + 255) / 256 * 256) * 8 * 2 * 2;            // adt + ads: [round_up(N,256)][8] float2 each
}
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
static bool wide_bwd_graph_ok(const gadapt_graph* g) {
    return g->ell_t && g->ell_s && g->xpos_t && g->xpos_s && g->wide_deg_t > 0 && g->wide_deg_s > 0 && g->rowptr_s && g->col_s && g->perm_s;
}
// top layer of a block whose caller takes x[:, :dim]: compact upstream gradient [N,g_cols]
static int launch_wide_bwd_compact(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha, const float* a,
                                   const float* p0, const float* lp, float* part4, float* slab, int slab_rows, int accumulate,
                                   float* g_out, int residual_only, int g_cols, float* wide_ws, hipStream_t st) {
    const int64_t n_pad = ((int64_t)g->n_nodes + 255) / 256 * 256;
    float* win = wide_ws;
    float* wout = wide_ws + n_pad * 8;
    const int n_steps = (g->n_nodes + wide::STEP - 1) / wide::STEP;
    {
        wide::BwdEArgs pe{x_in, g_in, alpha, lp, g->rowptr_t, g->ell_t, g->ell_s, g->xpos_t, g->xpos_s, win, wout, part4,
                          g->n_nodes, g->n_edges, n_steps, g_cols};
        ProfScope prof(7, st, 1);
        hipLaunchKernelGGL(wide::bwd_edge_compact_kernel, dim3(n_steps), dim3(256), 0, st, pe);
        if (int rc = check_launch("wide::bwd_edge_compact_kernel")) return rc;
    }
    wide::BwdMArgs pm{x_in, part4, win, wout, a, p0, lp, g->ell_t, g->ell_s, g_out, slab, slab_rows, accumulate,
                      g->n_nodes, n_steps, g->wide_deg_t, g->wide_deg_s, nullptr};
#ifdef GADAPT_STAMPS
    pm.stamps = g_stamp_buf;                                     // region 0 (the forward kernels' - they ran earlier in the step)
#endif
    ProfScope prof(8, st, 1);
    constexpr int lds = wide::bwd_main_lds_bytes();
    allow_lds(wide::bwd_main_kernel, lds);
    hipLaunchKernelGGL(wide::bwd_main_kernel, dim3(wide_grid(n_steps)), dim3(512), lds, st, pm);
    return check_launch("wide::bwd_main_kernel");
}
#ifndef GADAPT_BWD_OUT4
#define GADAPT_BWD_OUT4 1            /* 0: layer 1 above a compact layer 0 runs the dense pair (A/B) */
#endif
#ifndef GADAPT_XC_COMPACT_KERNEL
#define GADAPT_XC_COMPACT_KERNEL 1      /* 0: the tiled target kernel with the XC staging for the compact layer input (A/B) */
#endif
template <int C> static int launch_bwd(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha,
                                       const float* a, const float* p0, const float* lp, float* edge_ws, float* dxd, float* slab,
                                       int accumulate, float* sums_out, float* g_out, int residual_only, int g_cols, int x_cols, hipStream_t st, float* sums_sc_out,
                                       int out4 = 0, int g_stride = 0, int sums_partials = 0, float* wide_ws = nullptr) {
    // out4: only columns 0..3 of g_out are wanted (dxd and g_out are [N,4]: D4 target pass + grand_bwd_source4_kernel).
    // g_stride: row pitch of g_in in floats for the compact-input launch (0 = C).
    using K = Cfg<C>;
    const int n_tiles = (g->n_nodes + K::TM - 1) / K::TM;
    if (g_cols < 0 || g_cols > 4) return fail(GADAPT_E_BADARG, "compact upstream gradient: 1..4 columns");
    if ((x_cols != 0 && x_cols != 4) || (x_cols && (g_cols || g_out)))
        return fail(GADAPT_E_BADARG, "compact layer input: 4 columns, layer 0 of a block of >= 2 layers, no d x0");
    if (x_cols && residual_only) return fail(GADAPT_E_BADARG, "compact layer input: Euler-step layers only");
    BwdTArgs pt{x_in, g_in, alpha, a, lp, g->rowptr_t, g->col_t, g->tpos_s, meta_for<K::TM>(g->meta_t), reinterpret_cast<float2*>(edge_ws), dxd, slab, sums_out,
                g->n_nodes, n_tiles, accumulate, residual_only, g->n_edges, nullptr, g_cols};
    pt.sums_sc_out = sums_sc_out;
    pt.c = C;
    pt.g_stride = g_stride ? g_stride : C;
    pt.sums_partials = sums_partials;
    // strided tile walk of the hidden-64 dense target pass (128-node mesh rows and the like: gadapt_tile_meta_strided_host)
    const bool strided = C == 64 && g->t_strips >= 2 && n_tiles % g->t_strips == 0 && g->n_nodes % K::TM == 0;
    pt.t_strips = strided ? g->t_strips : 1;
    // (the strip-by-strip order applied inside each XCD group's share of the tiles instead - same nodes per XCD as in the source pass
    // that follows - was measured no better, and its share boundaries need the L2 path: not kept)
    pt.t_rows = strided ? n_tiles / g->t_strips : n_tiles;
    if (out4 && (!g_out || x_cols || residual_only || C < 8 || (g_cols && sums_out)))
        return fail(GADAPT_E_BADARG, "4-column backward: a layer with a gradient to pass on, hidden >= 8, not compact-g with d dt / d scale");
#ifdef GADAPT_STAMPS
    pt.stamps = g_stamp_buf ? g_stamp_buf + 1024 * 32 : nullptr;
#endif
    constexpr int lds_t = K::lds_bytes(1, K::RING_T + 1, 1), lds_s = K::lds_bytes(2);
    int rc;
    if constexpr (C == 64) {
        // (residual_only launches keep the pair: the edge kernel's part row would need the base factor of the caller's update)
        if (wide_ws && g_out && g_cols && !x_cols && !sums_out && !out4 && !residual_only && wide_bwd_graph_ok(g) && wide_bwd_enabled())
            return launch_wide_bwd_compact(g, x_in, g_in, alpha, a, p0, lp, dxd, slab, grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS)),
                                           accumulate, g_out, residual_only, g_cols, wide_ws, st);
    }
    // hidden 32 / 64 with a source pass to follow: the source pass accumulates dA / dp0 (see grand_bwd_source_kernel)
    constexpr bool CAN_MOVE_DA = (C == 32 || C == 64) && GADAPT_DA_IN_SOURCE;
    const bool da_in_s = CAN_MOVE_DA && g_out && !sums_out && !x_cols && !out4;
    if (GADAPT_XC_COMPACT_KERNEL && x_cols && !residual_only) {
        ProfScope prof(1, st, 2);
        // the slab holds one row per workgroup of the tiled target pass: same grid, so every row is visited
        const dim3 grid(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS)));
        if (sums_out && sums_sc_out) hipLaunchKernelGGL((grand_bwd_target_compact_kernel<2>), grid, dim3(256), 0, st, pt);
        else if (sums_out) hipLaunchKernelGGL((grand_bwd_target_compact_kernel<1>), grid, dim3(256), 0, st, pt);
        else hipLaunchKernelGGL((grand_bwd_target_compact_kernel<0>), grid, dim3(256), 0, st, pt);
        return check_launch("grand_bwd_target_compact_kernel");
    }
    {
        ProfScope prof(1, st, (g_cols ? 1 : 0) | (x_cols ? 2 : 0) | (out4 ? 8 : 0));
        if constexpr (CAN_MOVE_DA) {
            if (da_in_s && g_cols) {
                allow_lds(grand_bwd_target_kernel<C, false, true, false, false>, lds_t);
                hipLaunchKernelGGL((grand_bwd_target_kernel<C, false, true, false, false>), dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(K::NT), lds_t, st, pt);
            } else if (da_in_s) {
                allow_lds(grand_bwd_target_kernel<C, false, false, false, false>, lds_t);
                hipLaunchKernelGGL((grand_bwd_target_kernel<C, false, false, false, false>), dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(K::NT), lds_t, st, pt);
            }
        }
        if (da_in_s) {
        } else {
            // instantiations: SUMS 0 / 1 (d dt: learn_step) / 2 (d dt and d score_scale), each plain, with the compact upstream
            // gradient (GC), with the compact layer input on the tiled kernel (XC: only when the compact-input kernel is compiled
            // out) and - hidden >= 8 - with the 4-column dxd (D4)
            auto go = [&](auto kern) { allow_lds(kern, lds_t); hipLaunchKernelGGL(kern, dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(K::NT), lds_t, st, pt); };
            auto pick = [&](auto sums_tag) {
                constexpr int S = decltype(sums_tag)::value;
                if (x_cols) {
                    if constexpr (!GADAPT_XC_COMPACT_KERNEL) go(grand_bwd_target_kernel<C, S, false, true>);
                } else if (out4) {
                    if constexpr (C >= 8) {
                        if (!g_cols) go(grand_bwd_target_kernel<C, S, false, false, true, true>);
                        else if constexpr (S == 0) go(grand_bwd_target_kernel<C, 0, true, false, true, true>);   // GC + D4 + SUMS: not built (gadapt_block_backward)
                    }
                } else if (g_cols) {
                    go(grand_bwd_target_kernel<C, S, true>);
                } else if (strided && S == 0) {
                    if constexpr (C == 64 && S == 0) {
                        constexpr int lds_str = lds_t + K::RING_T * 2 * K::LD * 4;       // one halo row per slab side
                        auto kern = grand_bwd_target_kernel<C, 0, false, false, true, false, true>;
                        allow_lds(kern, lds_str);
                        hipLaunchKernelGGL(kern, dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(K::NT), lds_str, st, pt);
                    }
                } else {
                    go(grand_bwd_target_kernel<C, S>);
                }
            };
            if (sums_out && sums_sc_out) pick(IntTag<2>{}); else if (sums_out) pick(IntTag<1>{}); else pick(IntTag<0>{});
        }
        rc = check_launch("grand_bwd_target_kernel");
    }
    if (rc || !g_out) return rc;
    BwdSArgs ps{x_in, g_in, edge_ws, dxd, a, p0, g->rowptr_s, g->col_s, meta_for<K::TM>(g->meta_s), g_out, g->n_nodes, n_tiles, g->n_edges, nullptr, g_cols,
                slab, accumulate};
#ifdef GADAPT_STAMPS
    ps.stamps = g_stamp_buf ? g_stamp_buf + 2 * 1024 * 32 : nullptr;
#endif
    ProfScope prof(2, st, (g_cols ? 1 : 0) | (out4 ? 8 : 0));
    if constexpr (C >= 8) {
        if (out4) {
            constexpr int lds_4 = K::lds_bytes(2, 0);
            constexpr int res4 = (C > 64 || K::NT != 256) ? 512 : 256 * GADAPT_WAVES_BWD_S4;   // resident workgroups: waves per SIMD x 256 CUs
            if (g_cols) {
                allow_lds(grand_bwd_source4_kernel<C, true>, lds_4);
                hipLaunchKernelGGL((grand_bwd_source4_kernel<C, true>), dim3(grid_for(n_tiles, res4)), dim3(K::NT), lds_4, st, ps);
            } else {
                allow_lds(grand_bwd_source4_kernel<C, false>, lds_4);
                hipLaunchKernelGGL((grand_bwd_source4_kernel<C, false>), dim3(grid_for(n_tiles, res4)), dim3(K::NT), lds_4, st, ps);
            }
            return check_launch("grand_bwd_source4_kernel");
        }
    }
    if constexpr (CAN_MOVE_DA) {
        if (da_in_s) {
            // the grid EXPRESSION of the target pass (resident_blocks_bwd_t: what gadapt_backward_slab_rows sizes the slab with),
            // not the source pass's own: at hidden 32 the two differ (256 against 512 workgroups) and the slab holds one row per
            // workgroup of the pass that flushes it
            constexpr int lds_sd = K::lds_bytes(2, 3);
            if (g_cols) {
                allow_lds(grand_bwd_source_kernel<C, true, true>, lds_sd);
                hipLaunchKernelGGL((grand_bwd_source_kernel<C, true, true>), dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(K::NT), lds_sd, st, ps);
            } else {
                allow_lds(grand_bwd_source_kernel<C, false, true>, lds_sd);
                hipLaunchKernelGGL((grand_bwd_source_kernel<C, false, true>), dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(K::NT), lds_sd, st, ps);
            }
            return check_launch("grand_bwd_source_kernel");
        }
    }
    if constexpr (K::MFMA && GADAPT_S_WINDOW(C)) {
      // mesh-ordered graphs only (the wide kernels' test: every out-neighbour near its node): elsewhere no tile is windowed and
      // the slabs would be staged for nothing
      if (g->wide_deg_s > 0) {
        constexpr int lds_sw = K::lds_bytes(2, 4, 1);           // window (3 slabs) + y tile; ext = ring offsets
        if (g_cols) {
            allow_lds(grand_bwd_source_kernel<C, true, false, true>, lds_sw);
            hipLaunchKernelGGL((grand_bwd_source_kernel<C, true, false, true>), dim3(grid_for(n_tiles, resident_blocks_bwd<C>(GADAPT_BWD_S_MAX_BLOCKS))), dim3(K::NT), lds_sw, st, ps);
        } else {
            allow_lds(grand_bwd_source_kernel<C, false, false, true>, lds_sw);
            hipLaunchKernelGGL((grand_bwd_source_kernel<C, false, false, true>), dim3(grid_for(n_tiles, resident_blocks_bwd<C>(GADAPT_BWD_S_MAX_BLOCKS))), dim3(K::NT), lds_sw, st, ps);
        }
        return check_launch("grand_bwd_source_kernel");
      }
    }
    if (g_cols) {
        allow_lds(grand_bwd_source_kernel<C, true>, lds_s);
        hipLaunchKernelGGL((grand_bwd_source_kernel<C, true>), dim3(grid_for(n_tiles, resident_blocks_bwd<C>(GADAPT_BWD_S_MAX_BLOCKS))), dim3(K::NT), lds_s, st, ps);
    } else {
        allow_lds(grand_bwd_source_kernel<C>, lds_s);
        hipLaunchKernelGGL(grand_bwd_source_kernel<C>, dim3(grid_for(n_tiles, resident_blocks_bwd<C>(GADAPT_BWD_S_MAX_BLOCKS))), dim3(K::NT), lds_s, st, ps);
    }
    return check_launch("grand_bwd_source_kernel");
}

#define GADAPT_DISPATCH_C(c, CALL)                                                   \
    switch (c) {                                                                     \
        case 4:   { constexpr int CC = 4;   return CALL; }                           \
        case 8:   { constexpr int CC = 8;   return CALL; }                           \
        case 16:  { constexpr int CC = 16;  return CALL; }                           \
        case 32:  { constexpr int CC = 32;  return CALL; }                           \
        case 64:  { constexpr int CC = 64;  return CALL; }                           \
        case 128: { constexpr int CC = 128; return CALL; }                           \
        default: return fail(GADAPT_E_BADARG, "hidden_dim must be one of 4, 8, 16, 32, 64, 128");  \
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
    out3[0] = out3[1] = out3[2] = -1;
#define GADAPT_OCC(CC)                                                                                                  \
    case CC: {                                                                                                          \
        using K = Cfg<CC>;                                                                                              \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[0], grand_fwd_kernel<CC>, K::NT, K::lds_bytes(0, K::RING + 1)); \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[1], grand_bwd_target_kernel<CC, false>, K::NT, K::lds_bytes(1, K::RING_T + 1, 1)); \
        if constexpr (K::MFMA && GADAPT_S_WINDOW(CC))                                                                   \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[2], grand_bwd_source_kernel<CC, false, false, true>, K::NT, K::lds_bytes(2, 4, 1)); \
        else                                                                                                            \
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[2], grand_bwd_source_kernel<CC>, K::NT, K::lds_bytes(2)); \
        return GADAPT_OK;                                                                                               \
    }
    switch (c) { GADAPT_OCC(4) GADAPT_OCC(8) GADAPT_OCC(16) GADAPT_OCC(32) GADAPT_OCC(64) GADAPT_OCC(128) default: break; }
#undef GADAPT_OCC
    return fail(GADAPT_E_BADARG, "occupancy: unsupported hidden_dim");
}

extern "C" int gadapt_layer_forward(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                                    const float* layer_params, float* alpha_out, int residual_only, int c, void* stream) {
    if (int rc = check_graph(g, c)) return rc;
    if (!x_in || !x_out || !a || !p0 || !layer_params || x_in == x_out) return fail(GADAPT_E_BADARG, "layer_forward: null or aliased pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GADAPT_DISPATCH_C(c, launch_fwd<CC>(g, x_in, x_out, a, p0, layer_params, alpha_out, residual_only, 0, nullptr, st));
}

template <int C> static int tiles_for(int64_t n_nodes) { return (int)((n_nodes + Cfg<C>::TM - 1) / Cfg<C>::TM); }
extern "C" int gadapt_backward_slab_rows(int64_t n_nodes, int c) {
    if (n_nodes <= 0) return fail(GADAPT_E_BADARG, "slab_rows: bad node count");
    GADAPT_DISPATCH_C(c, grid_for(tiles_for<CC>(n_nodes), resident_blocks_bwd_t<CC>(GADAPT_BWD_T_MAX_BLOCKS)));
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
    GADAPT_DISPATCH_C(c, launch_bwd<CC>(g, x_in, g_in, alpha, a, p0, layer_params, edge_ws, dxd_ws, slab, accumulate, sums_out, g_out, residual_only, 0, 0, st,
                                        sums_out ? sums_out + 1 : nullptr));   // {d dt, d score_scale} side by side
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
#define GADAPT_R2CB(CC) case CC: allow_lds(reduce2_coeffs_bwd_kernel<CC>, lds); \
        hipLaunchKernelGGL(reduce2_coeffs_bwd_kernel<CC>, dim3(blocks), dim3(1024), lds, st, scratch, wq, bq, wk, d_wq, d_bq, d_wk, d_bk); break;
    switch (c) { GADAPT_R2CB(4) GADAPT_R2CB(8) GADAPT_R2CB(16) GADAPT_R2CB(32) GADAPT_R2CB(64) GADAPT_R2CB(128) default: break; }
#undef GADAPT_R2CB
    return check_launch("slab_reduce_coeffs_backward");
}

extern "C" int gadapt_coeffs_forward(const float* wq, const float* bq, const float* wk, float* a_out, float* p0_out, int c, void* stream) {
    if (!wq || !bq || !wk || !a_out || !p0_out || c <= 0) return fail(GADAPT_E_BADARG, "coeffs_forward: bad argument");
    if (!gadapt_supported_hidden_dim(c)) return fail(GADAPT_E_BADARG, "coeffs_forward: unsupported hidden_dim");
    const int n = c * c + c;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GADAPT_COEFFS_F(CC) case CC: hipLaunchKernelGGL(coeffs_fwd_kernel<CC>, dim3((n + 255) / 256), dim3(256), 0, st, wq, bq, wk, a_out, p0_out); break;
    switch (c) { GADAPT_COEFFS_F(4) GADAPT_COEFFS_F(8) GADAPT_COEFFS_F(16) GADAPT_COEFFS_F(32) GADAPT_COEFFS_F(64) GADAPT_COEFFS_F(128) default: break; }
#undef GADAPT_COEFFS_F
    return check_launch("coeffs_fwd_kernel");
}
extern "C" int gadapt_coeffs_backward(const float* wq, const float* bq, const float* wk, const float* d_a, const float* d_p0,
                                      float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c, void* stream) {
    if (!wq || !bq || !wk || !d_a || !d_p0 || !d_wq || !d_bq || !d_wk || !d_bk || c <= 0) return fail(GADAPT_E_BADARG, "coeffs_backward: bad argument");
    if (!gadapt_supported_hidden_dim(c)) return fail(GADAPT_E_BADARG, "coeffs_backward: unsupported hidden_dim");
    const int n = 2 * c * c + 2 * c;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GADAPT_COEFFS_B(CC) case CC: hipLaunchKernelGGL(coeffs_bwd_kernel<CC>, dim3((n + 255) / 256), dim3(256), 0, st, wq, bq, wk, d_a, d_p0, d_wq, d_bq, d_wk, d_bk); break;
    switch (c) { GADAPT_COEFFS_B(4) GADAPT_COEFFS_B(8) GADAPT_COEFFS_B(16) GADAPT_COEFFS_B(32) GADAPT_COEFFS_B(64) GADAPT_COEFFS_B(128) default: break; }
#undef GADAPT_COEFFS_B
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
#define GADAPT_ENC_CF(CC) case CC: hipLaunchKernelGGL(encode_coeffs_kernel<CC>, dim3((unsigned)blocks + cblocks), dim3(256), 0, static_cast<hipStream_t>(stream), \
                                                      p, (int)blocks, cf->wq, cf->bq, cf->wk, cf->a, cf->p0); break;
    switch (cf->c) { GADAPT_ENC_CF(4) GADAPT_ENC_CF(8) GADAPT_ENC_CF(16) GADAPT_ENC_CF(32) GADAPT_ENC_CF(64) GADAPT_ENC_CF(128)
                     default: return fail(GADAPT_E_BADARG, "encode_features_coeffs: unsupported hidden_dim"); }
#undef GADAPT_ENC_CF
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
static int layer_forward_cols(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0, const float* layer_params,
                              float* alpha_out, int x_cols, float* x_top4, int c, hipStream_t st) {
    GADAPT_DISPATCH_C(c, launch_fwd<CC>(g, x_in, x_out, a, p0, layer_params, alpha_out, 0, x_cols, x_top4, st));
}
extern "C" int gadapt_block_forward(const gadapt_graph* g, float* x_all, int x0_cols, int n_layers, const float* a, int64_t a_stride,
                                    const float* p0, int64_t p0_stride, const float* layer_params, float* alpha_all, float* x_top4,
                                    int c, void* stream) {
    if (int rc = check_graph(g, c)) return rc;
    if (!x_all || n_layers <= 0 || !a || !p0 || !layer_params) return fail(GADAPT_E_BADARG, "block_forward: bad argument");
    if (x0_cols != 0 && (x0_cols != 4 || n_layers < 2 || c < 8)) return fail(GADAPT_E_BADARG, "block_forward: compact x0 needs 4 columns, >= 2 layers, hidden >= 8");
    const size_t nc = (size_t)g->n_nodes * c;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int l = 0; l < n_layers; ++l) {
        const bool last = (l == n_layers - 1);
        float* alpha_l = alpha_all ? alpha_all + (size_t)l * g->n_edges : nullptr;
        int rc;
        if ((l == 0 && x0_cols) || (last && x_top4))             // compact input and/or compact-only output
            rc = layer_forward_cols(g, x_all + l * nc, (last && x_top4) ? nullptr : x_all + (l + 1) * nc, a + l * a_stride, p0 + l * p0_stride,
                                    layer_params + 2 * l, alpha_l, l == 0 ? x0_cols : 0, last ? x_top4 : nullptr, c, st);
        else
            rc = gadapt_layer_forward(g, x_all + l * nc, x_all + (l + 1) * nc, a + l * a_stride, p0 + l * p0_stride,
                                      layer_params + 2 * l, alpha_l, 0, c, stream);
        if (rc) return rc;
    }
    return GADAPT_OK;
}

static int layer_backward_cols(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha, const float* a,
                               const float* p0, const float* layer_params, float* edge_ws, float* dxd_ws, float* slab, int accumulate,
                               float* sums_out, float* sums_sc_out, float* g_out, int g_cols, int x_cols, int c, hipStream_t st,
                               int out4, int g_stride, float* wide_ws) {
    GADAPT_DISPATCH_C(c, launch_bwd<CC>(g, x_in, g_in, alpha, a, p0, layer_params, edge_ws, dxd_ws, slab, accumulate, sums_out, g_out, 0, g_cols, x_cols, st, sums_sc_out,
                                        out4, g_stride, 1, wide_ws));
}
extern "C" int gadapt_block_backward_ws(const gadapt_graph* g, const float* x_all, int x0_cols, const float* alpha_all, const float* g_top, int g_top_cols, int n_layers,
                                        const float* a, int64_t a_stride, const float* p0, int64_t p0_stride, const float* layer_params,
                                        float* g_ws, float* dxd_ws, float* edge_ws, float* slab, float* d_layer_params, int want_d_scale, float* d_x0,
                                        int c, void* stream, float* wide_ws) {
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
        const bool pair4 = GADAPT_BWD_OUT4 && x0_cols && c >= 8 && n_layers >= 2 && !(n_layers == 2 && g_top_cols > 0 && d_layer_params);
        const int out4 = (pair4 && l == 1) ? 1 : 0;
        const int g_stride = (pair4 && l == 0) ? 4 : 0;
        // dense layers that hand their result to the next layer of the block: dxd lives in the g_out buffer (see bwd_inplace_enabled)
        const bool inplace = bwd_inplace_enabled() && l > 0 && g_next && !out4 && !(wide_ws && wide_bwd_enabled());
        int rc = layer_backward_cols(g, x_all + l * nc, g_cur, alpha_all + (size_t)l * g->n_edges, a + l * a_stride, p0 + l * p0_stride,
                                     layer_params + 2 * l, edge_ws, inplace ? g_next : dxd_ws, slab_l, accumulate, d_dt, d_sc, g_next, g_cols, x_cols, c, st,
                                     out4, g_stride, wide_ws);
        if (rc) return rc;
        g_cur = g_next;
    }
    return GADAPT_OK;
}

extern "C" int gadapt_block_backward(const gadapt_graph* g, const float* x_all, int x0_cols, const float* alpha_all, const float* g_top, int g_top_cols, int n_layers,
                                     const float* a, int64_t a_stride, const float* p0, int64_t p0_stride, const float* layer_params,
                                     float* g_ws, float* dxd_ws, float* edge_ws, float* slab, float* d_layer_params, int want_d_scale, float* d_x0,
                                     int c, void* stream) {
    return gadapt_block_backward_ws(g, x_all, x0_cols, alpha_all, g_top, g_top_cols, n_layers, a, a_stride, p0, p0_stride, layer_params, g_ws, dxd_ws, edge_ws,
                                    slab, d_layer_params, want_d_scale, d_x0, c, stream, nullptr);
}

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
template <int C> static void launch_small(const smallmesh::Args& p, int n_meshes, int lds, hipStream_t st) {
    auto go = [&](auto kern, int nt) { allow_lds(kern, lds); hipLaunchKernelGGL(kern, dim3(n_meshes), dim3(nt), lds, st, p); };
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
}
extern "C" int gadapt_small_forward(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                                    const float* x_comp, int dim, const float* f, const float* uu, const float* enc_w, int n_feat,
                                    const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                                    const float* layer_params, int n_layers, float* out, int out_cols, float* alpha_all, float* x_all, int c, void* stream) {
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope prof(9, st, x_all ? 32 : 0);
    switch (c) {
        case 4: launch_small<4>(p, n_meshes, (int)lds, st); break;
        case 8: launch_small<8>(p, n_meshes, (int)lds, st); break;
        case 16: launch_small<16>(p, n_meshes, (int)lds, st); break;
        default: launch_small<32>(p, n_meshes, (int)lds, st); break;
    }
    return check_launch("smallmesh::fwd_kernel");
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

#include "gadapt_sparse.inc"

#include "gadapt_gat.inc"
