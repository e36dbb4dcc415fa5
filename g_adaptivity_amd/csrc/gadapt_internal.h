// gadapt_internal.h - what the translation units of libgadapt_hip.so share on the host side: build-time geometry parameters, error
// reporting, the optional per-launch event pairs, launch helpers, and the entry points one unit offers the others.
//
// The library is built from several .hip files so that `make -j` compiles the kernel families in parallel and an edit to one
// family rebuilds one unit (the hot kernels are large templates instantiated for six hidden sizes; as ONE unit the build took
// 2 min 20 s on one core):
//   gadapt_kernels.hip         C-ABI of layers / blocks / small per-step kernels, error + profiling state
//   gadapt_tu_fwd.hip          grand_fwd_kernel<C> (gadapt_fwd.inc), wide::fwd_kernel (gadapt_wide.inc)
//   gadapt_tu_bwd_target.hip   grand_bwd_target_kernel<C,...>, grand_bwd_target_compact_kernel (gadapt_bwd_target.inc)
//   gadapt_tu_bwd_source.hip   grand_bwd_source_kernel<C,...>, grand_bwd_source4_kernel (gadapt_bwd_source.inc)
//   gadapt_tu_smallmesh.hip    one-launch small-mesh pair (gadapt_smallmesh.inc)
//   gadapt_tu_sparse.hip       generic CSR primitives (gadapt_sparse.inc)
//   gadapt_tu_gat.hip          fused GAT_plus block (gadapt_gat.inc)
// Device code is never shared across units (no relocatable device code): every unit includes gadapt_common.inc itself.
#ifndef GADAPT_INTERNAL_H
#define GADAPT_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>
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
// (512-thread workgroups at hidden 128 - two waves per SIMD - were measured in rounds 2 and 6, per kernel family: the source pass then
// fits 206 registers without spills and runs 80 us against 68, the target pass is level, the forward 78 against 61: docs/measurements.md
// K.  Every tiled kernel is a 256-thread workgroup.)
#ifndef GADAPT_SLAB_CHUNKS
#define GADAPT_SLAB_CHUNKS 8    // second-level partials of the slab reduction (<= 32: the scratch the callers allocate).
                                // 32 / 16 / 8 chunks: first level 5.1 / 5.5 / 5.0 us, second level + chain rule 12.1 / 8.8 / 6.6 us
#endif
// Softmax arithmetic: expf / IEEE division (<= 1 ulp each).  The approximate forms (v_exp_f32 of a rounded product, v_rcp_f32)
// leave alpha with ~4x the rounding error of the reference's exp / true division; harmless for the coordinates (2e-7 either way)
// but visible in parameter gradients that are the small remainder of large cancelling sums (64x64, 6 layers, hidden 128: 3.4e-4
// against the fp64 oracle with them, 1.4e-4 without, fp32 reference path 1.0e-4).  Cost of the exact forms: +0.6 us per forward
// launch (17.1 -> 17.7 us), nothing measurable elsewhere.
__device__ __forceinline__ float sm_exp(float x) { return expf(x); }
__device__ __forceinline__ float sm_rcp(float x) { return 1.0f / x; }

// ------------------------------------------------------------------------------------------------
// error reporting (state in gadapt_kernels.hip)
// ------------------------------------------------------------------------------------------------
int gadapt_fail_(int code, const char* msg);
int gadapt_check_launch_(const char* what);
static inline int fail(int code, const char* msg) { return gadapt_fail_(code, msg); }
static inline int check_launch(const char* what) { return gadapt_check_launch_(what); }

// ------------------------------------------------------------------------------------------------
// optional per-kernel timing (bench/roofline only): HIP events on the launch stream around every hot-kernel launch.  Off by
// default; when off a launch reads one relaxed atomic and touches nothing else.  variant: bit 0 = compact upstream gradient,
// bit 1 = compact layer input, bit 2 = head-only output, bit 3 = 4-column backward output.
// ------------------------------------------------------------------------------------------------
struct ProfScope {
    hipStream_t st; int idx = -1; hipEvent_t eb = nullptr;
    ProfScope(int id, hipStream_t s, int variant = 0);
    ~ProfScope();
};

#ifdef GADAPT_STAMPS
extern unsigned long long* g_stamp_buf;                         // diagnostic builds: in-kernel cycle stamps (tools/stamp_*.py)
#endif

#include "gadapt_common.inc"

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// Workgroups of a launch: a multiple of 8 (XCD groups), at most max_blocks (the resident set) unless that would give a workgroup
// more than 64 tiles (Cfg::MAXM: its tile metadata must fit the LDS table).
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
#ifndef GADAPT_BWD_S_MAX_BLOCKS
#define GADAPT_BWD_S_MAX_BLOCKS 512      /* the resident set (2 workgroups per CU): measured 27.8 vs 28.7 us with 1024 */
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
// of host time, which is most of an eager small-graph forward, so it is remembered (per translation unit).
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

// resident set of a launch: two 256-thread workgroups per CU, or one 512-thread workgroup (Cfg::NT)
template <int C> static constexpr int resident_blocks(int two_per_cu_default) { return Cfg<C>::NT == 512 ? 256 : two_per_cu_default; }
// ... and the backward kernels of hidden sizes that run one wave per SIMD (GADAPT_ONE_WAVE_C: 392 / 504 registers) fit ONE
// 256-thread workgroup per CU: 256 workgroups are the resident set, a 512-workgroup launch would run as two rounds (and
// flush twice as many slab rows).
template <int C> static constexpr int resident_blocks_bwd(int two_per_cu_default) {
    return C >= GADAPT_ONE_WAVE_C ? 256 : resident_blocks<C>(two_per_cu_default);
}
// ... and the target pass at hidden 32 (128-row tiles: ring + dP tile + slices = 87 KB of LDS) fits one workgroup per CU too
// (hipOccupancyMaxActiveBlocksPerMultiprocessor: forward / target / source = 2 / 1 / 2 at hidden 32, 2 / 2 / 2 at 64,
// 2 / 1 / 1 at 128, 3 / 3 / 3 at 8).
template <int C> static constexpr int resident_blocks_bwd_t(int two_per_cu_default) {
    return C == 32 ? 256 : resident_blocks_bwd<C>(two_per_cu_default);
}
template <int C> static int tiles_for(int64_t n_nodes) { return (int)((n_nodes + Cfg<C>::TM - 1) / Cfg<C>::TM); }

// GADAPT_DEV_C=<hidden>: development builds that instantiate the tiled kernels for ONE hidden size only (never shipped: build()
// and the Makefile's default target compile every size)
#ifdef GADAPT_DEV_C
#define GADAPT_DISPATCH_C(c, CALL)                                                   \
    switch (c) {                                                                     \
        case GADAPT_DEV_C: { constexpr int CC = GADAPT_DEV_C; return CALL; }         \
        default: return fail(GADAPT_E_BADARG, "development build: one hidden size only (GADAPT_DEV_C)");  \
    }
#else
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
#endif

// ------------------------------------------------------------------------------------------------
// what the kernel units offer the C-ABI unit (validated arguments; one launch each)
// ------------------------------------------------------------------------------------------------
// Extras of the forward launches of a fused training step (gadapt_block_forward_loss): layer 0 assembles its compact [N,4] input
// from the caller's node fields (fs) and writes it to x0c for the layer-0 backward; the head-only launch of the last layer also
// produces the loss derivative and one loss partial per wave (loss; *n_partials_out = the grid of that launch).
// cw / a_out / p0_out (nullable): the flat parameter bucket - the layer-0 launch computes the composite coefficients itself when it is
// the wide kernel (gadapt_forward_computes_coeffs_c) and writes them to a_out / p0_out for the launches that follow.
struct FwdExtra { FieldSrc fs; float* x0c; LossArgs loss; int* n_partials_out; const float* cw; float* a_out; float* p0_out; };
int gadapt_forward_computes_coeffs_c(const gadapt_graph* g, int c);
#define GADAPT_LOSS_PARTIALS_MAX 4096  /* waves of a forward launch: GADAPT_FWD_MAX_BLOCKS workgroups of up to 8 waves */
int gadapt_launch_fwd_c(int c, const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0, const float* lp,
                        float* alpha_out, int residual_only, int x_cols, float* x_top4, hipStream_t st, const FwdExtra* extra = nullptr);
// target pass (always) ...
int gadapt_launch_bwd_target_c(int c, const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha, const float* a,
                               const float* lp, float* edge_ws, float* dxd, float* slab, int accumulate, float* sums_out, float* sums_sc_out,
                               int want_source, int residual_only, int g_cols, int x_cols, int out4, int g_stride, int sums_partials,
                               hipStream_t st);
// ... and the source pass of the same layer (g_out != NULL)
int gadapt_launch_bwd_source_c(int c, const gadapt_graph* g, const float* x_in, const float* g_in, const float* edge_ws, const float* dxd,
                               const float* a, const float* p0, float* g_out, int g_cols, int out4, hipStream_t st);
int gadapt_slab_rows_c(int64_t n_nodes, int c);
int gadapt_occupancy_fwd_c(int c);
int gadapt_occupancy_bwd_target_c(int c);
int gadapt_occupancy_bwd_source_c(int c);

#endif  // GADAPT_INTERNAL_H
