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
#ifndef GADAPT_STAGGER_FWD
#define GADAPT_STAGGER_FWD 0
#endif
#ifndef GADAPT_STAGGER_T
#define GADAPT_STAGGER_T 0
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
#ifndef GADAPT_T_TWO_BUFFERS
#define GADAPT_T_TWO_BUFFERS 0     // 1: one-wave target pass (hidden 128) with two row buffers in the edge walk; measured 128.0 vs 128.3 us
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
#ifndef GADAPT_T_MFMA_PRIO
#define GADAPT_T_MFMA_PRIO 0    // s_setprio level of the target pass's matrix phases (0: leave the priority alone)
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
#ifndef GADAPT_DA_BPREFETCH
#define GADAPT_DA_BPREFETCH 1
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
extern "C" int gadapt_abi_version(void) { return 4; }
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

// ------------------------------------------------------------------------------------------------
// compile-time geometry
// ------------------------------------------------------------------------------------------------
template <int C> struct Cfg {
    static constexpr bool MFMA = (C >= 32);
    static constexpr int FPL = MFMA ? GADAPT_FPL : 4;  // floats per lane
    static constexpr int NV = FPL / 4;                 // float4 chunks per lane
    static constexpr int LPN = C / FPL;                // lanes per node
    // threads per workgroup (512 from GADAPT_WIDE_WG_C on: see there)
    static constexpr int NT = (C >= GADAPT_WIDE_WG_C) ? 512 : 256;
    static constexpr int NW = NT / 64;                 // waves per workgroup
    static constexpr int SLOTS = NT / LPN;             // nodes in flight per workgroup
    static constexpr int TM = MFMA ? (C == 32 ? 128 : 64) : (SLOTS < 64 ? 64 : SLOTS);
    static constexpr int ITERS = TM / SLOTS;
    static constexpr int LD = C + 4;                   // padded LDS row (floats): conflict-free b128 rows
    static constexpr int TILE_FLOATS = TM * LD;
    static constexpr int CB = C / 32;                  // 32-wide column blocks (MFMA path)
    static constexpr int RB = TM / 32;                 // 32-high row blocks
    static constexpr int CAP = 7 * TM;                 // CSR entries of one tile staged in LDS (rest read from HBM)
    static constexpr int COLN = CAP + 64;              // + padding: reads up to MAXD past a row's end stay in
                                                       //   bounds and return a valid node id (weight 0)
    // LDS: `tiles` [TM][LD] tiles, rowptr[TM+1] (padded to TM+4), col[COLN], aux[AUXW*COLN]
    static constexpr int MAXM = 64;                    // tile-metadata words of this workgroup's tiles kept in LDS
    static constexpr int lds_bytes(int auxw, int tiles = 2, int ext = 0) { return (tiles * TILE_FLOATS + (TM + 4) + COLN + (auxw + ext) * COLN + 4 * MAXM) * 4; }
    static constexpr int RING = (C <= 64 || (C >= GADAPT_ONE_WAVE_C && GADAPT_FWD_ONE_WAVE)) ? 3 : 1;   // LDS slabs of x rows kept by the rolling-window kernels; C = 128: a
                                                       //   3-slab window would leave one workgroup per CU, so only the tile itself
    static constexpr int LEAD = (RING == 3) ? 1 : 0;   // the slab staged during tile t is slab t + LEAD
    // The target pass keeps the window at hidden 128 too: its 392 registers allow one workgroup per CU whatever the LDS
    // footprint is (4 tiles of 33 KB + slices = 145 KB), and at one wave per SIMD an L2 gather is fully exposed latency.
    static constexpr int RING_T = (C <= GADAPT_T_RING_MAX_C) ? 3 : 1;
    static constexpr int LEAD_T = (RING_T == 3) ? 1 : 0;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ void axpy4(float4& y, float a, const float4& x) {
    y.x = fmaf(a, x.x, y.x); y.y = fmaf(a, x.y, y.y); y.z = fmaf(a, x.z, y.z); y.w = fmaf(a, x.w, y.w);
}

// Sum over the LPN lanes that share a node, result in every lane.  DPP row operations (no LDS
// crossbar): quad_perm xor-1 / xor-2, row_half_mirror, row_mirror; the 32-lane case adds one
// ds_swizzle (xor 16).
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return v + __builtin_bit_cast(float, moved);
}
template <int LPN> __device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPN >= 2) v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    if constexpr (LPN >= 4) v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    if constexpr (LPN >= 8) v = dpp_add<0x141>(v);     // row_half_mirror
    if constexpr (LPN >= 16) v = dpp_add<0x140>(v);    // row_mirror
    if constexpr (LPN >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
    return v;
}

// Row gather with a 32-bit byte offset (the launcher checks N*C*4 < 4 GiB).
template <int C> __device__ __forceinline__ float4 ld_row4(const float* __restrict__ base, int row, int sub) {
    const uint32_t off = (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)sub * 16u;
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + off);
}
// streaming forms (read or written once per launch: keep them from displacing the rows other workgroups gather through L2)
typedef float f32x4nt __attribute__((ext_vector_type(4)));
template <int C> __device__ __forceinline__ float4 ld_row4_nt(const float* __restrict__ base, int row, int sub) {
    const uint32_t off = (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)sub * 16u;
    const f32x4nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(reinterpret_cast<const char*>(base) + off));
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int C> __device__ __forceinline__ void st_row4_nt(float* __restrict__ base, int row, int sub, const float4& v) {
    const uint32_t off = (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)sub * 16u;
    const f32x4nt t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4nt*>(reinterpret_cast<char*>(base) + off));
}
template <int C> __device__ __forceinline__ void st_row4(float* __restrict__ base, int row, int sub, const float4& v) {
    const uint32_t off = (uint32_t)row * (uint32_t)(C * 4) + (uint32_t)sub * 16u;
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + off) = v;
}

// Tiles [t, t_end) step `step` for this workgroup.  Workgroups with equal blockIdx%8 are observed to
// share an XCD (placement is a speed assumption only); each such group walks one contiguous eighth
// of the tiles so gathered neighbour rows hit that XCD's L2.
struct TileRange { int t, t_end, step; };
__device__ __forceinline__ TileRange tile_range(int n_tiles) {
    const int gx = gridDim.x >> 3, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int per = (n_tiles + 7) >> 3;
    const int t0 = xcd * per;
    const int t1 = min(n_tiles, t0 + per);
    return {t0 + bi, t1, gx};
}

// Phase stagger: the second half of the launch (the second workgroup of every CU when two are resident) starts
// `units` x 64 cycles late, so that co-resident workgroups are in different phases (edge walk: VALU/LDS, GEMM
// phases: MFMA) and the launch does not request its tiles from HBM in lockstep bursts.
template <int UNITS> __device__ __forceinline__ void stagger_start() {
    if constexpr (UNITS > 0) {
        if (blockIdx.x >= (gridDim.x >> 1)) {
#pragma unroll 1
            for (int k = 0; k < UNITS; k += 64) __builtin_amdgcn_s_sleep(64);
        }
    }
}

// Contiguous tiles [t0, t1) for this workgroup (rolling-window kernels): XCD group x walks the x-th eighth of the
// tiles, split evenly between that group's workgroups.
struct TileChunk { int t0, t1; };
__device__ __forceinline__ TileChunk tile_chunk(int n_tiles) {
    const int gx = gridDim.x >> 3, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int per = (n_tiles + 7) >> 3;
    const int x0 = xcd * per, x1 = min(n_tiles, x0 + per);
    if (x1 <= x0) return {0, 0};
    const int pw = (x1 - x0 + gx - 1) / gx;
    const int t0 = x0 + bi * pw;
    return {min(t0, x1), min(x1, t0 + pw)};
}

// ------------------------------------------------------------------------------------------------
// [TM,C] x [C,C] on the matrix cores.  D[n][j] = sum_k IN[n][k] * B[k][j] (+ bias[j]).
//   TRANS = false: B[k][j] = M[j*C + k]   (D = IN M^T : forward P = x A^T, source pass A y)
//   TRANS = true : B[k][j] = M[k*C + j]   (D = IN M   : target pass dP A)
// v_mfma_f32_32x32x2_f32: lane l feeds A[l&31][k=l>>5] and B[k=l>>5][l&31]; the k index is
// permuted so each lane-half reads 16 contiguous bytes of its IN row per 4 MFMAs (half h owns
// k in {8q+4h .. 8q+4h+3}); any permutation is valid as long as A and B agree.
// ------------------------------------------------------------------------------------------------
#ifndef GADAPT_GEMM_SPLIT
#define GADAPT_GEMM_SPLIT 1
#endif
#ifndef GADAPT_SPLIT_MAX_C
#define GADAPT_SPLIT_MAX_C 128
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// fp32 values on the bf16 matrix cores without losing fp32 accuracy: x = h + m + l exactly, three bf16 pieces of 8
// mantissa bits each (truncation; every subtraction is exact).  A product x*y keeps the six piece products down to
// 2^-16 relative (hh, hm, mh, mm, hl, lh); the dropped ones (ml, lm, ll) are below 2^-23 of |x*y|, the rounding
// level of an fp32 product, and every piece product is exact in the fp32 accumulator.  v_mfma_f32_32x32x16_bf16
// does 16x the work of v_mfma_f32_32x32x2_f32 in half its cycles, so six of them per 16 k cost 3/8 of the fp32 form.
struct Split3 { u32x4 h, m, l; };
#ifndef GADAPT_MFMA_INTERLEAVE
#define GADAPT_MFMA_INTERLEAVE 1
#endif
// (The same pipeline over the dA phase's (k-step, block) sequence was measured slower - target pass 32.5 -> 35.1 us at hidden
// 64, 97 -> 104.5 at hidden 128 - and its sched_group_barrier patterns took the build from 2 to 16 minutes: not kept.)
#ifndef GADAPT_SPLIT_PK
#define GADAPT_SPLIT_PK 0       // 1: residuals of the split on v_pk_add_f32 (two subtractions per instruction); measured 0.3597 ms
                                // per step against 0.3568 with scalar subtractions (three runs each): no gain
#endif
__device__ __forceinline__ Split3 split8(const float (&x)[8]) {
    Split3 s;
#if GADAPT_SPLIT_PK
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // element 2q in the low half, 2q+1 in the high half
        const f2 v = {x[2 * q], x[2 * q + 1]};
        const u2 vb = __builtin_bit_cast(u2, v);
        const f2 hf = __builtin_bit_cast(f2, vb & 0xFFFF0000u);
        const f2 r = v - hf;                            // exact
        const u2 rb = __builtin_bit_cast(u2, r);
        const f2 mf = __builtin_bit_cast(f2, rb & 0xFFFF0000u);
        const f2 r2 = r - mf;                           // exact
        const u2 r2b = __builtin_bit_cast(u2, r2);
        s.h[q] = __builtin_amdgcn_perm(vb.y, vb.x, 0x07060302u);
        s.m[q] = __builtin_amdgcn_perm(rb.y, rb.x, 0x07060302u);
        s.l[q] = __builtin_amdgcn_perm(r2b.y, r2b.x, 0x07060302u);
    }
#else
    float r[8], r2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        r[e] = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xFFFF0000u);
        r2[e] = r[e] - __uint_as_float(__float_as_uint(r[e]) & 0xFFFF0000u);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // element 2q in the low half, 2q+1 in the high half
        s.h[q] = __builtin_amdgcn_perm(__float_as_uint(x[2 * q + 1]), __float_as_uint(x[2 * q]), 0x07060302u);
        s.m[q] = __builtin_amdgcn_perm(__float_as_uint(r[2 * q + 1]), __float_as_uint(r[2 * q]), 0x07060302u);
        s.l[q] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * q + 1]), __float_as_uint(r2[2 * q]), 0x07060302u);
    }
#endif
    return s;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Two-piece f16 form (GADAPT_SPLIT_F16_*; where it is used: see below): the operand block is first multiplied by a power of two that brings its
// largest magnitude into [2^14, 2^15) - exact, and it puts the whole block into f16's range - then x = h + l with
// h = f16(x) and l = f16(x - h), round to nearest: 22 to 23 significant bits, |x - h - l| <= 2^-23 |x| or 2^-25 (f16's
// subnormal spacing, i.e. 2^-39 of the block's maximum), whichever is larger.  Products: hh + hl + lh (ll is below 2^-22
// of |x y| and 2^-24 on average, the rounding level of an fp32 product), every piece product exact in the fp32
// accumulator; the result is multiplied by the inverse powers of two.  Against the three-piece bf16 form (split8): half
// the matrix instructions (3 per 16 k instead of 6), 24 instead of 44 vector instructions per 8 values, fragments of 8
// instead of 12 registers per k-step; in exchange a maximum per scale group per use.  Scale groups follow the operand
// layout so that they cost two or three cross-lane steps and no wave-wide reduction (a first version with one scale per
// wave block - DPP reduction, readfirstlane - serialised every matrix phase behind it and was slower than the bf16 form):
// the A operand (lane (i, h): row i, half h of each k-step) shares a scale between FOUR consecutive rows - quad DPP and
// one half swap - because the 32x32 output layout gives a lane four consecutive rows per register quad, so it fetches four
// inverses (ds_bpermute) instead of sixteen; the B operand (lane (j, h): column j) has one scale per column, lane-local
// after the half swap.  An entry 2^-q below its group's maximum keeps min(23, 39 - q) bits.
// Measured (MI355X, same box, us per launch, bf16 three-piece -> f16 two-piece): source pass 26.8 -> 25.9 (hidden 64),
// 86.5 -> 83.4 (hidden 128); tiled forward 61.0 -> 60.3 (hidden 128); wide forward 17.9 -> 18.0; target pass 32.6 -> 34.6
// (hidden 64), 96.4 -> 98.6 (hidden 128): the f16 form has to hold a whole operand block until its maximum is known, where the
// bf16 form streams k-step by k-step, and the target pass has neither the registers (3 spills) nor the slack for that.  With
// the scales known in advance (diagnostic build, unit scale) the target pass would gain 0.9 / 7 us - not pursued.
// Tiled forward at hidden 64 (128-wide meshes): 50.1 -> 52.0; at hidden 32: 12.6 -> 12.4.
// So: on for the source pass and for the tiled forward at hidden 32 / 128, off for the target pass and the wide forward.
#ifndef GADAPT_SPLIT_F16_S
#define GADAPT_SPLIT_F16_S 1        // source pass A y
#endif
#ifndef GADAPT_SPLIT_F16_F
#define GADAPT_SPLIT_F16_F(C) ((C) != 64)   // tiled forward P = x A^T
#endif
#ifndef GADAPT_SPLIT_F16_T
#define GADAPT_SPLIT_F16_T 0        // TileGemm<C, true> and the dA phase: target pass
#endif
#ifndef GADAPT_DA_F16
#define GADAPT_DA_F16 0             // dA phase of the target pass in the f16 form (measured with GADAPT_SPLIT_F16_T, see above)
#endif
// PRE-SPLIT A operand (GADAPT_PRESPLIT_S / _T): the [TM,C] operand tile is written to LDS ONCE, already in the two-piece f16
// form, by the lanes that own its rows - one power-of-two scale per ROW (a lane group holds a whole row: its maximum is a DPP
// reduction over the group), h and l pieces of 8 consecutive k side by side in the 32 bytes the 8 fp32 values would take, so
// the tile keeps its footprint and a matrix-core lane fetches a k-step's fragment pair with two ds_read_b128 and NO vector
// arithmetic.  The per-wave splits this replaces are redundant across the waves that share rows: x4 at hidden 128 (every
// wave split the whole tile: 704 vector instructions per lane and tile in the target pass's projection, 1584 with its dA
// phase, a quarter of the kernel), x2 at hidden 64.  The row's inverse scale stays with the lanes that wrote the row - the
// same lanes read the product's row back and fold it into that read.
#ifndef GADAPT_PRESPLIT_S
#define GADAPT_PRESPLIT_S 1         // source pass: the y tile
#endif
#ifndef GADAPT_XC_ONE_KSTEP
#define GADAPT_XC_ONE_KSTEP 1       // compact layer input [N,4]: the projection's operand is zero beyond column 3 - one k-step of 16
#endif
#ifndef GADAPT_PRESPLIT_F
#define GADAPT_PRESPLIT_F 1         // forward without the LDS window (hidden 128): the x tile, at staging
#endif
#ifndef GADAPT_PRESPLIT_T
#define GADAPT_PRESPLIT_T 1         // target pass: dP for dP A (copy in the window slot that is free after the edge walk)
#endif
#ifndef GADAPT_SPLIT_F16_WIDE
#define GADAPT_SPLIT_F16_WIDE 0     // wide forward (gadapt_wide.inc)
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct Split2 { u32x4 h, l; };
struct Pow2 { float s, inv; };
// largest power-of-two scale that keeps mx below 2^15, and its inverse (both normal fp32 numbers for every mx)
__device__ __forceinline__ Pow2 pow2_scale(float mx) {
    int e = (int)((__float_as_uint(mx) >> 23) & 0xFFu);
    e = min(max(e, 15), 254);
    return {__uint_as_float((unsigned)(268 - e) << 23), __uint_as_float((unsigned)(e - 14) << 23)};
}
template <int CTRL> __device__ __forceinline__ float dpp_max(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true);
    return fmaxf(v, __builtin_bit_cast(float, moved));
}
// max(v(lane), v(lane ^ 32)), in both lanes
__device__ __forceinline__ float half_max(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
// scale of an A-operand lane (row r31, half h): maximum over the four rows 4 (r31 / 4) .. + 3 and both halves
__device__ __forceinline__ float quad_rows_max(float v) {
#ifdef GADAPT_ABL_FIXED_SCALE
    return 1.0f;
#endif
    v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v);
    return half_max(v);
}
// the four inverse scales a lane of the 32x32 output needs: register r holds row (r & 3) + 8 (r >> 2) + 4 h, i.e. row quad
// 2 (r >> 2) + h, whose scale sits in lane 8 (r >> 2) + 4 h of the A operand
__device__ __forceinline__ void quad_rows_inverse(float inv, int h, float (&out)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        out[q] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(32 * q + 16 * h, __builtin_bit_cast(int, inv)));
}
// maximum of a non-negative value over the wave, wave-uniform (scalar register)
__device__ __forceinline__ float wave_max(float v) {
#ifdef GADAPT_ABL_FIXED_SCALE
    return 1.0f;                                        // diagnostic build (timing only): no reduction, unit scale
#endif
    v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v); v = dpp_max<0x141>(v); v = dpp_max<0x140>(v);     // rows of 16 lanes
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F)));   // xor 16
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));                   // xor 32
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fmaxf(a, b))));
}
__device__ __forceinline__ float absmax8(const float (&x)[8], float m) {
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(x[e]));
    return m;
}
__device__ __forceinline__ Split2 split8h(const float (&x)[8], float s) {
    Split2 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {                       // element 2q in the low half, 2q+1 in the high half
        const float t0 = x[2 * q] * s, t1 = x[2 * q + 1] * s;
        f16x2 h; h.x = (_Float16)t0; h.y = (_Float16)t1;
        f16x2 l; l.x = (_Float16)(t0 - (float)h.x); l.y = (_Float16)(t1 - (float)h.y);   // the differences are exact
        r.h[q] = __builtin_bit_cast(unsigned, h);
        r.l[q] = __builtin_bit_cast(unsigned, l);
    }
    return r;
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mma3(const Split2& a, const Split2& b, f32x16 c) {
    c = mfma_f16(a.h, b.l, c);                          // small pieces first
    c = mfma_f16(a.l, b.h, c);
    return mfma_f16(a.h, b.h, c);
}

// Instruction order for the scheduler (sched_group_barrier): NDS LDS reads, then six times {one matrix instruction, NVALU
// vector instructions} - the shape of one software-pipelined k-step of the split products.
template <int NDS, int NVALU, int NM = 6> __device__ __forceinline__ void mfma_gap_pattern() {
    if constexpr (NDS > 0) __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
    for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);
    }
}

template <int C, bool TRANS, bool F16_ = (TRANS && GADAPT_SPLIT_F16_T)> struct TileGemm {
    using K = Cfg<C>;
    static constexpr int BPW = (K::CB * K::RB) / K::NW;   // 32x32 output blocks per wave
    static_assert(!K::MFMA || ((K::CB * K::RB) % K::NW == 0 && K::NW % K::CB == 0), "output blocks split evenly over the waves");
    static constexpr bool SPLIT = GADAPT_GEMM_SPLIT && C <= GADAPT_SPLIT_MAX_C;
    static constexpr int KS = C / 16;                  // k-steps of the bf16 form
    static constexpr bool F16 = SPLIT && F16_;
    float bf[SPLIT ? 1 : C / 2];
    Split3 bs[(SPLIT && !F16) ? KS : 1];
    Split2 bh[F16 ? KS : 1];
    float binv;                                        // F16: inverse scale of this lane's B column
    float oinv[BPW][4];                                // F16: inverse of (A scale x B scale) of the last accumulate: output block, row quad
    float bias;
    int cb, rb0, lane;

    __device__ __forceinline__ void init(int lane_, int wave) {
        lane = lane_;
        cb = wave % K::CB;
        rb0 = wave / K::CB;
    }
    // Issue the B-operand loads ([C,C] matrix: L2-resident, same lines for every workgroup).
    __device__ __forceinline__ void load(const float* __restrict__ M, const float* __restrict__ bias_vec) {
        const int h = lane >> 5, j = cb * 32 + (lane & 31);
        if constexpr (F16) {
            float v[KS][8];
            float mx = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (!TRANS) {
                    const float4 v0 = *reinterpret_cast<const float4*>(M + (size_t)j * C + 16 * ks + 8 * h);
                    const float4 v1 = *reinterpret_cast<const float4*>(M + (size_t)j * C + 16 * ks + 8 * h + 4);
                    v[ks][0] = v0.x; v[ks][1] = v0.y; v[ks][2] = v0.z; v[ks][3] = v0.w;
                    v[ks][4] = v1.x; v[ks][5] = v1.y; v[ks][6] = v1.z; v[ks][7] = v1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[ks][e] = M[(size_t)(16 * ks + 8 * h + e) * C + j];
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mx = absmax8(v[ks], mx);
            const Pow2 sb = pow2_scale(half_max(mx));
            binv = sb.inv;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bh[ks] = split8h(v[ks], sb.s);
        } else if constexpr (SPLIT) {
            // lane (j, h) holds B[k][j] for k = 16 ks + 8 h + e, e = 0..7 (both operands use this k order)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float v[8];
                if (!TRANS) {
                    const float4 v0 = *reinterpret_cast<const float4*>(M + (size_t)j * C + 16 * ks + 8 * h);
                    const float4 v1 = *reinterpret_cast<const float4*>(M + (size_t)j * C + 16 * ks + 8 * h + 4);
                    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = M[(size_t)(16 * ks + 8 * h + e) * C + j];
                }
                bs[ks] = split8(v);
            }
        } else {
#pragma unroll
            for (int q = 0; q < C / 8; ++q) {
                if (!TRANS) {
                    const float4 v = *reinterpret_cast<const float4*>(M + (size_t)j * C + 8 * q + 4 * h);
                    bf[4 * q + 0] = v.x; bf[4 * q + 1] = v.y; bf[4 * q + 2] = v.z; bf[4 * q + 3] = v.w;
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) bf[4 * q + t] = M[(size_t)(8 * q + 4 * h + t) * C + j];
                }
            }
        }
        bias = bias_vec ? bias_vec[j] : 0.f;
    }

    // The same in two halves for kernels that rebuild the fragments per tile (hidden 128 forward): the loads are requested
    // before other work of the tile (staging commits), the split runs after it - the L2 round trip is no longer exposed.
    struct BRaw { float v[(SPLIT && F16_) ? KS : 1][8]; };
    __device__ __forceinline__ void load_issue(const float* __restrict__ M, BRaw& raw) const {
        if constexpr (F16) {
            const int h = lane >> 5, j = cb * 32 + (lane & 31);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (!TRANS) {
                    const float4 v0 = *reinterpret_cast<const float4*>(M + (size_t)j * C + 16 * ks + 8 * h);
                    const float4 v1 = *reinterpret_cast<const float4*>(M + (size_t)j * C + 16 * ks + 8 * h + 4);
                    raw.v[ks][0] = v0.x; raw.v[ks][1] = v0.y; raw.v[ks][2] = v0.z; raw.v[ks][3] = v0.w;
                    raw.v[ks][4] = v1.x; raw.v[ks][5] = v1.y; raw.v[ks][6] = v1.z; raw.v[ks][7] = v1.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) raw.v[ks][e] = M[(size_t)(16 * ks + 8 * h + e) * C + j];
                }
            }
        }
    }
    __device__ __forceinline__ void load_finish(const float* __restrict__ M, const float* __restrict__ bias_vec, const BRaw& raw) {
        if constexpr (F16) {
            float mx = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) mx = absmax8(raw.v[ks], mx);
            const Pow2 sb = pow2_scale(half_max(mx));
            binv = sb.inv;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bh[ks] = split8h(raw.v[ks], sb.s);
            bias = bias_vec ? bias_vec[cb * 32 + (lane & 31)] : 0.f;
        } else {
            load(M, bias_vec);
        }
    }

    // in_tile/out_tile: LDS [TM][LD].  Caller synchronises around it.
    // KSU: k-steps that carry data (the caller knows the operand is zero beyond 16 KSU columns: compact layer input)
    template <int KSU = KS> __device__ __forceinline__ void accumulate(const float* in_tile, f32x16 (&acc)[BPW]) {
        const int h = lane >> 5, r31 = lane & 31;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int rb = rb0 + b * (K::NW / K::CB);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            if constexpr (F16) {
                const float* arow = in_tile + (rb * 32 + r31) * K::LD + 8 * h;
                auto rd = [&](int ks, float (&v)[8]) __attribute__((always_inline)) {
                    const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
                    const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ks + 4);
                    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
                };
                float mx = 0.f;
                if constexpr (KSU <= 4) {                               // the block's rows stay in registers between the two uses
                    float v[KSU][8];
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) rd(ks, v[ks]);
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) mx = absmax8(v[ks], mx);
                    const Pow2 sa = pow2_scale(quad_rows_max(mx));
                    quad_rows_inverse(sa.inv, h, oinv[b]);
#if GADAPT_MFMA_INTERLEAVE
                    Split2 cur = split8h(v[0], sa.s);
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) {                  // split of k-step ks+1 in the gaps of the three products of ks
                        Split2 nxt = cur;
                        if (ks + 1 < KSU) nxt = split8h(v[ks + 1], sa.s);
                        acc[b] = mma3(cur, bh[ks], acc[b]);
                        if (ks + 1 < KSU) mfma_gap_pattern<0, 8, 3>();
                        cur = nxt;
                    }
#else
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) acc[b] = mma3(split8h(v[ks], sa.s), bh[ks], acc[b]);
#endif
                } else {                                               // hidden 128: read twice rather than hold 64 registers
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) { float v[8]; rd(ks, v); mx = absmax8(v, mx); }
                    const Pow2 sa = pow2_scale(quad_rows_max(mx));
                    quad_rows_inverse(sa.inv, h, oinv[b]);
#if GADAPT_MFMA_INTERLEAVE
                    float v0[8];
                    rd(0, v0);
                    Split2 cur = split8h(v0, sa.s);
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) {
                        Split2 nxt = cur;
                        if (ks + 1 < KSU) { float v[8]; rd(ks + 1, v); nxt = split8h(v, sa.s); }
                        acc[b] = mma3(cur, bh[ks], acc[b]);
                        if (ks + 1 < KSU) mfma_gap_pattern<2, 8, 3>();
                        cur = nxt;
                    }
#else
#pragma unroll
                    for (int ks = 0; ks < KSU; ++ks) { float v[8]; rd(ks, v); acc[b] = mma3(split8h(v, sa.s), bh[ks], acc[b]); }
#endif
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) oinv[b][q] *= binv;
            } else if constexpr (SPLIT) {
                const float* arow = in_tile + (rb * 32 + r31) * K::LD + 8 * h;
                auto ldsplit = [&](int ks) __attribute__((always_inline)) {
                    const float4 a0 = *reinterpret_cast<const float4*>(arow + 16 * ks);
                    const float4 a1 = *reinterpret_cast<const float4*>(arow + 16 * ks + 4);
                    const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    return split8(v);
                };
#if GADAPT_MFMA_INTERLEAVE
                // Software pipeline: the split of k-step ks+1 (44 vector instructions) is issued in the gaps of the six matrix
                // instructions of k-step ks.  A wave issues in order, so six matrix instructions back to back (192 cycles of
                // pipe time, 8 cycles of issue each) followed by the next split leave the matrix pipe idle during the split and
                // the vector ALU idle during the six: 384 cycles per k-step measured at one wave per SIMD (hidden 128).
                Split3 cur = ldsplit(0);
#pragma unroll
                for (int ks = 0; ks < KSU; ++ks) {
                    Split3 nxt = cur;
                    if (ks + 1 < KSU) nxt = ldsplit(ks + 1);
                    acc[b] = mfma_bf16(cur.h, bs[ks].l, acc[b]);        // small pieces first
                    acc[b] = mfma_bf16(cur.l, bs[ks].h, acc[b]);
                    acc[b] = mfma_bf16(cur.m, bs[ks].m, acc[b]);
                    acc[b] = mfma_bf16(cur.h, bs[ks].m, acc[b]);
                    acc[b] = mfma_bf16(cur.m, bs[ks].h, acc[b]);
                    acc[b] = mfma_bf16(cur.h, bs[ks].h, acc[b]);
                    if (ks + 1 < KSU) mfma_gap_pattern<2, 8>();
                    cur = nxt;
                }
#else
#pragma unroll
                for (int ks = 0; ks < KSU; ++ks) {
                    const Split3 as = ldsplit(ks);
                    acc[b] = mfma_bf16(as.h, bs[ks].l, acc[b]);         // small pieces first
                    acc[b] = mfma_bf16(as.l, bs[ks].h, acc[b]);
                    acc[b] = mfma_bf16(as.m, bs[ks].m, acc[b]);
                    acc[b] = mfma_bf16(as.h, bs[ks].m, acc[b]);
                    acc[b] = mfma_bf16(as.m, bs[ks].h, acc[b]);
                    acc[b] = mfma_bf16(as.h, bs[ks].h, acc[b]);
                }
#endif
            } else {
                const float* arow = in_tile + (rb * 32 + r31) * K::LD + 4 * h;
#pragma unroll
                for (int q = 0; q < C / 8; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(arow + 8 * q);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bf[4 * q + 0], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bf[4 * q + 1], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bf[4 * q + 2], acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bf[4 * q + 3], acc[b], 0, 0, 0);
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* out_tile, const f32x16 (&acc)[BPW]) const {
        const int h = lane >> 5, r31 = lane & 31;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int rb = rb0 + b * (K::NW / K::CB);
            float* ocol = out_tile + cb * 32 + r31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                ocol[row * K::LD] = F16 ? fmaf(acc[b][r], oinv[b][r >> 2], bias) : acc[b][r] + bias;
            }
        }
    }
    // PRE-SPLIT A operand (lds_put_split): in_tile holds, per row and group of 8 k, the h pieces (16 bytes) then the l pieces
    __device__ __forceinline__ void accumulate_presplit(const float* in_tile, f32x16 (&acc)[BPW]) const {
        static_assert(F16, "pre-split operands are two-piece f16");
        const int h = lane >> 5, r31 = lane & 31;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int rb = rb0 + b * (K::NW / K::CB);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            const float* arow = in_tile + (rb * 32 + r31) * K::LD + 8 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Split2 a;
                a.h = *reinterpret_cast<const u32x4*>(arow + 16 * ks);
                a.l = *reinterpret_cast<const u32x4*>(arow + 16 * ks + 4);
                acc[b] = mma3(a, bh[ks], acc[b]);
            }
        }
    }
    // result of accumulate_presplit times the B column's inverse scale; the row's inverse scale is applied by the reader
    __device__ __forceinline__ void store_presplit(float* out_tile, const f32x16 (&acc)[BPW]) const {
        const int h = lane >> 5, r31 = lane & 31;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int rb = rb0 + b * (K::NW / K::CB);
            float* ocol = out_tile + cb * 32 + r31;
#pragma unroll
            for (int r = 0; r < 16; ++r) ocol[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * K::LD] = acc[b][r] * binv;
        }
    }
    template <int KSU = KS> __device__ __forceinline__ void run(const float* in_tile, float* out_tile) {
        f32x16 acc[BPW];
        accumulate<KSU>(in_tile, acc);
        store(out_tile, acc);
    }
    // out == in: every wave reads its operand rows before any wave overwrites them (two workgroup barriers inside)
    __device__ __forceinline__ void run_in_place(float* tile) {
        f32x16 acc[BPW];
        accumulate(tile, acc);
        __syncthreads();
        store(tile, acc);
    }
};

// The tile's slice of one CSR orientation in LDS: rowptr[TM+1], col[COLN] and AUXW per-edge words.  Where the slice
// starts, how long it is and the tile's longest row come from per-tile metadata built with the graph
// (gadapt_tile_meta_host), so every load of the stage is issued at once: one memory round trip per tile.
// A tile whose slice does not fit (more than CAP entries) or that has a row longer than MAXD is "slow":
// the whole workgroup walks the HBM copy with plain loops instead.
// EXT = 1: one more per-edge int32 array (separate in HBM) staged with the slice.  EXT = 2: no such array in HBM - the LDS
// array holds the ring offsets of the slice's columns when the tile is windowed, next to the node ids in col (a kernel that
// gathers one matrix from the ring and another one from HBM through the same edges: the source pass).
template <int C, int AUXW, int EXT = 0> struct TileCsr {
    using K = Cfg<C>;
    int* rp; int* col; float* aux; int* ext; int4* metas;
    const int32_t* rowptr_g; const int32_t* col_g; const float* aux_g; const int32_t* ext_g; const int4* meta_g;
    int ebase, n_edges_m1, m_first, m_step, m_ntiles;

    __device__ __forceinline__ void bind(float* lds_after_tiles, const int32_t* rowptr_g_, const int32_t* col_g_, const float* aux_g_,
                                         const int32_t* meta_g_, int n_edges, const int32_t* ext_g_ = nullptr) {
        n_edges_m1 = max(n_edges - 1, 0);
        rp = reinterpret_cast<int*>(lds_after_tiles);
        col = rp + (K::TM + 4);
        aux = reinterpret_cast<float*>(col + K::COLN);
        ext = reinterpret_cast<int*>(aux + AUXW * K::COLN);
        metas = reinterpret_cast<int4*>(ext + (EXT > 0 ? 1 : 0) * K::COLN);
        rowptr_g = rowptr_g_; col_g = col_g_; aux_g = aux_g_; ext_g = ext_g_; meta_g = reinterpret_cast<const int4*>(meta_g_);
    }
    // Metadata of this workgroup's tiles first, first+step, ... -> LDS, once per launch.  Read per tile with
    // meta_at(k): a per-tile global load would be moved to SGPRs by hipcc (v_readfirstlane) and its s_waitcnt,
    // vmcnt being in-order, would also wait for every older load - the GEMM's B fragments.  Caller barriers.
    __device__ __forceinline__ void load_metas(int first, int step, int n_tiles, int tid) {
        metas_commit(metas_issue(first, step, n_tiles, tid), tid);
    }
    // The same in two halves, so a kernel's prologue can put this load in ONE memory round trip with its other first loads
    // (window slabs, weight fragments): request everything, then consume - vmcnt is in-order, the first wait covers all.
    __device__ __forceinline__ int4 metas_issue(int first, int step, int n_tiles, int tid) {
        m_first = first; m_step = step; m_ntiles = n_tiles;
        return meta_g[max(min(first + min(tid, K::MAXM - 1) * step, n_tiles - 1), 0)];   // unconditional (see issue())
    }
    __device__ __forceinline__ void metas_commit(const int4& m, int tid) {
        if (tid < K::MAXM) metas[tid] = m;
    }
    // pure LDS read (the launch grid guarantees at most MAXM tiles per workgroup): a global-load fallback here,
    // even on a never-taken branch, makes hipcc wait vmcnt(0) at the join in every tile
    __device__ __forceinline__ int4 meta_at(int k) const { return metas[min(k, K::MAXM - 1)]; }
    // Staging is split so a kernel can request tile k+1 while it computes tile k: issue() only starts the
    // loads (results stay in registers), commit() writes them to LDS.  Nothing depends on an earlier load
    // except the slice position, which comes from the metadata word read one tile ahead.
    static constexpr int CQ = (K::CAP + GADAPT_MAXD + K::NT - 1) / K::NT;
    struct Regs {
        int4 meta;
        int rpv, rpv2;
        int colv[CQ];
        float auxv[AUXW > 0 ? AUXW * CQ : 1];
        int extv[EXT == 1 ? CQ : 1];
    };
    // Every load here is
    // UNCONDITIONAL with a clamped address (validity is applied in commit): a load under a branch or an exec mask
    // makes hipcc assume the worst at the join and emit s_waitcnt vmcnt(small) later, which drains the whole
    // prefetch right where the B fragments of the GEMM are waited for (vmcnt is in-order).
    // k = index of the tile within this workgroup's sequence (see load_metas)
    __device__ __forceinline__ void issue(Regs& r, int k, int node0, int n_nodes, int tid) const {
        r.meta = meta_at(k);
        const int eb = r.meta.x;
        const int nb = min(max(node0, 0), n_nodes);              // a request one tile past either end of the chunk stays in bounds
        r.rpv = rowptr_g[min(nb + min(tid, K::TM), n_nodes)];
        r.rpv2 = (K::TM >= K::NT) ? rowptr_g[min(nb + min(K::NT + tid, K::TM), n_nodes)] : 0;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int idx = q * K::NT + tid;
            r.colv[q] = col_g[min(eb + idx, n_edges_m1)];
            if constexpr (EXT == 1) r.extv[q] = ext_g[min(eb + idx, n_edges_m1)];
            if constexpr (AUXW > 0) {
#pragma unroll
                for (int w = 0; w < AUXW; ++w) {
                    const int ia = (q * AUXW + w) * K::NT + tid;
                    r.auxv[q * AUXW + w] = aux_g[min((size_t)AUXW * eb + ia, (size_t)AUXW * n_edges_m1 + (AUXW - 1))];
                }
            }
        }
    }
    // Float offset, inside a RING-slab LDS ring, of the row of node j when slabs t-1, t, t+1 are resident
    // (slab s lives in slot s % RING).
    template <int RINGN = K::RING> static __device__ __forceinline__ int ring_off(int j, int t) {
        int slot = (t + RINGN - 1) % RINGN + (j / K::TM - (t - 1));
        if (slot >= RINGN) slot -= RINGN;
        return slot * K::TILE_FLOATS + (j % K::TM) * K::LD;
    }
    // LDS writes of a previously issued tile; returns its row-length bound (-1: slow tile).  Caller barriers.
    // windowed_tile >= 0: the tile's neighbours all live in slabs t-1..t+1 (metadata word 3): store ring offsets
    // instead of node ids so the gathers read LDS directly.
    template <int RINGN = K::RING> __device__ __forceinline__ int commit(const Regs& r, int tid, int node0_pad, int windowed_tile = -1) {
        static_assert(K::TM + 1 <= 2 * K::NT, "rowptr slice: at most 2 entries per thread");
        ebase = r.meta.x;
        if (tid <= K::TM) rp[tid] = r.rpv;
        if (K::TM >= K::NT && K::NT + tid <= K::TM) rp[K::NT + tid] = r.rpv2;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int idx = q * K::NT + tid;
            const int cnt = min(r.meta.y, K::CAP);
            const int jv = (idx < cnt) ? r.colv[q] : node0_pad;       // padding entries: a valid node of this tile
            if constexpr (EXT == 2) {
                if (idx < K::CAP + GADAPT_MAXD) { col[idx] = jv; ext[idx] = (windowed_tile >= 0) ? ring_off<RINGN>(jv, windowed_tile) : 0; }
            } else {
                if (idx < K::CAP + GADAPT_MAXD) col[idx] = (windowed_tile >= 0) ? ring_off<RINGN>(jv, windowed_tile) : jv;
            }
            if constexpr (EXT == 1) { if (idx < K::CAP + GADAPT_MAXD) ext[idx] = r.extv[q]; }
            if constexpr (AUXW > 0) {
#pragma unroll
                for (int w = 0; w < AUXW; ++w) {
                    const int ia = (q * AUXW + w) * K::NT + tid;
                    if (ia < AUXW * (K::CAP + GADAPT_MAXD)) aux[ia] = (ia < AUXW * cnt) ? r.auxv[q * AUXW + w] : 0.f;
                }
            }
        }
        return (r.meta.y > K::CAP || r.meta.z > GADAPT_MAXD) ? -1 : r.meta.z;
    }
};

// Chunk c4 of row `row` of a matrix stored as [N,d] (d <= 4) but read as [N,C] with zeros beyond column d: the top
// layer's upstream gradient (backward of x[:, :dim], GNN.py:299).  Loads are unconditional (see TileCsr::issue).
__device__ __forceinline__ float4 ld_row4_compact(const float* __restrict__ base, int row, int c4, int d) {
    const float* r = base + (size_t)row * d;
    const float v0 = r[0], v1 = r[min(1, d - 1)], v2 = r[min(2, d - 1)], v3 = r[min(3, d - 1)];
    float4 o = make_float4(v0, d > 1 ? v1 : 0.f, d > 2 ? v2 : 0.f, d > 3 ? v3 : 0.f);
    return c4 == 0 ? o : f4zero();
}
// Chunk c4 of row `row` of x: dense [N,C], or (XC) the compact [N,4] whose columns 4.. are zero by construction - the
// output of the identity encoder (zero-pad, GNN.py:75-82) that layer 0 reads without it ever being materialised.
template <int C, bool XC> __device__ __forceinline__ float4 ld_row4x(const float* __restrict__ base, int row, int c4) {
    if constexpr (XC) {
        const float4 t = *reinterpret_cast<const float4*>(base + 4 * (size_t)row);   // unconditional (see TileCsr::issue)
        return c4 == 0 ? t : f4zero();
    } else {
        return ld_row4<C>(base, row, c4);
    }
}
// x-tile rows of a tile held in registers between issue and commit (same split as TileCsr)
template <int C> struct TileRows {
    using K = Cfg<C>;
    static constexpr int V = C / 4;
    static constexpr int XQ = (K::TM * V + K::NT - 1) / K::NT;
    float4 v[XQ];
    int node0_;
    // unconditional clamped loads (see TileCsr::issue); rows past N are zeroed at commit
    // NT: non-temporal loads (rows that pass through once, e.g. the target pass's own g rows)
    template <bool NT = false> __device__ __forceinline__ void issue(const float* __restrict__ src, int node0, int n_nodes, int tid) {
        node0_ = node0;
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = min(q * K::NT + tid, K::TM * V - 1), r = idx / V, c4 = idx % V;
            if constexpr (NT) v[q] = ld_row4_nt<C>(src, min(max(node0 + r, 0), n_nodes - 1), c4);
            else v[q] = ld_row4<C>(src, min(max(node0 + r, 0), n_nodes - 1), c4);
        }
    }
    // XC: src is the compact [N,4] matrix (see ld_row4x): one 16-byte load per row (thread t < TM takes row t, in v[0]);
    // commit_sel writes zeros everywhere else
    template <bool XC> __device__ __forceinline__ void issue_sel(const float* __restrict__ src, int node0, int n_nodes, int tid) {
        if constexpr (XC) {
            static_assert(K::TM <= K::NT, "one compact row per thread");
            node0_ = node0;
            v[0] = *reinterpret_cast<const float4*>(src + 4 * (size_t)min(max(node0 + min(tid, K::TM - 1), 0), n_nodes - 1));
        } else {
            issue(src, node0, n_nodes, tid);
        }
    }
    template <bool XC> __device__ __forceinline__ void commit_sel(float* tile, int n_nodes, int tid) const {
        if constexpr (XC) {
#pragma unroll
            for (int q = 0; q < XQ; ++q) {
                const int idx = q * K::NT + tid, r = idx / V, c4 = idx % V;
                if (idx < K::TM * V && c4 != 0) *reinterpret_cast<float4*>(tile + r * K::LD + 4 * c4) = f4zero();
            }
            if (tid < K::TM)
                *reinterpret_cast<float4*>(tile + tid * K::LD) = (node0_ + tid < n_nodes && node0_ + tid >= 0) ? v[0] : f4zero();
        } else {
            commit(tile, n_nodes, tid);
        }
    }
    // commit() plus a pre-split f16 copy of the rows (layout of lds_put_split) in `ptile`, the row's inverse scale in ptile's
    // first pad column: for a tile whose rows are projected right after staging (forward without the LDS window, hidden 128).
    // A row's V = 32 chunks sit in 32 consecutive threads - half a wave - so its maximum is four DPP steps and one swizzle.
    __device__ __forceinline__ void commit_presplit(float* tile, float* ptile, int n_nodes, int tid) const {
        static_assert(V == 32 && (K::TM * V) % K::NT == 0, "pre-split staging: rows of 32 chunks, whole rows per pass");
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = q * K::NT + tid, r = idx / V, c4 = idx % V;
            const float4 x = (node0_ + r < n_nodes && node0_ + r >= 0) ? v[q] : f4zero();
            *reinterpret_cast<float4*>(tile + r * K::LD + 4 * c4) = x;
            float m = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
            m = dpp_max<0xB1>(m); m = dpp_max<0x4E>(m); m = dpp_max<0x141>(m); m = dpp_max<0x140>(m);
            m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, m), 0x401F)));
            const Pow2 s = pow2_scale(m);
            const float t0 = x.x * s.s, t1 = x.y * s.s, t2 = x.z * s.s, t3 = x.w * s.s;
            f16x2 h01, h23, l01, l23;
            h01.x = (_Float16)t0; h01.y = (_Float16)t1; h23.x = (_Float16)t2; h23.y = (_Float16)t3;
            l01.x = (_Float16)(t0 - (float)h01.x); l01.y = (_Float16)(t1 - (float)h01.y);
            l23.x = (_Float16)(t2 - (float)h23.x); l23.y = (_Float16)(t3 - (float)h23.y);
            uint2* grp = reinterpret_cast<uint2*>(ptile + r * K::LD + 8 * (c4 >> 1));
            grp[c4 & 1] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
            grp[2 + (c4 & 1)] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
            if (c4 == 0) ptile[r * K::LD + C] = s.inv;
        }
    }
    // same for a compact [N,d] source (ld_row4_compact): like the XC form, thread t < TM takes row t (chunk 0, in v[0]) and the
    // commit (commit_sel<true>) writes zeros everywhere else - not XQ chunk loads per thread of which one in V carries data
    __device__ __forceinline__ void issue_compact(const float* __restrict__ src, int d, int node0, int n_nodes, int tid) {
        static_assert(K::TM <= K::NT, "one compact row per thread");
        node0_ = node0;
        v[0] = ld_row4_compact(src, min(max(node0 + min(tid, K::TM - 1), 0), n_nodes - 1), 0, d);
    }
    __device__ __forceinline__ void commit(float* tile, int n_nodes, int tid) const {
#pragma unroll
        for (int q = 0; q < XQ; ++q) {
            const int idx = q * K::NT + tid, r = idx / V, c4 = idx % V;
            if (idx < K::TM * V)
                *reinterpret_cast<float4*>(tile + r * K::LD + 4 * c4) = (node0_ + r < n_nodes && node0_ + r >= 0) ? v[q] : f4zero();
        }
    }
};

// value held for edge `sub` of this lane's node, picked from a group-uniform array (the asm keeps it a
// select chain: hipcc otherwise spills the array to scratch and indexes it)
template <int N> __device__ __forceinline__ float pick(const float (&v)[N], int sub) {
    float r = v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) { r = (sub == k) ? v[k] : r; asm volatile("" : "+v"(r)); }
    return r;
}
// Sum over the node's lanes for N independent values at once, stage by stage, so that consecutive DPP
// instructions are independent (a dependent DPP chain pays 2 wait states per step).
template <int LPN, int N> __device__ __forceinline__ void group_sum_n(float (&v)[N]) {
    if constexpr (LPN >= 2) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = dpp_add<0xB1>(v[k]);
    }
    if constexpr (LPN >= 4) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = dpp_add<0x4E>(v[k]);
    }
    if constexpr (LPN >= 8) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = dpp_add<0x141>(v[k]);
    }
    if constexpr (LPN >= 16) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = dpp_add<0x140>(v[k]);
    }
    if constexpr (LPN >= 32) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v[k]), 0x401F));
    }
}
template <int V> struct IntTag { static constexpr int value = V; };
// Software pipeline over the ITERS node slots of a tile: rows of slot it+1 are requested before slot it
// is consumed.  `before_first` runs after the first request (e.g. the MFMA phase).  The widest row
// bound keeps a single buffer: two would cost a wave of occupancy for every tile shape.
template <int ITERS, typename BufT, bool TWO_BUFFERS = false, typename Fetch, typename Consume, typename Mid>
__device__ __forceinline__ void run_pipeline(Fetch&& fetch, Consume&& consume, Mid&& before_first) {
    if constexpr (sizeof(BufT) <= 128 || TWO_BUFFERS) {         // two buffers only while one stays within 32 VGPRs (or on request)
        BufT b0, b1;
        fetch(b0, 0);
        before_first();
#pragma unroll
        for (int it = 0; it < ITERS; it += 2) {
            if (it + 1 < ITERS) fetch(b1, it + 1);
            __builtin_amdgcn_sched_barrier(0);
            consume(b0, it);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < ITERS) {
                if (it + 2 < ITERS) fetch(b0, it + 2);
                __builtin_amdgcn_sched_barrier(0);
                consume(b1, it + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        BufT b0;
        fetch(b0, 0);
        before_first();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            consume(b0, it);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < ITERS) fetch(b0, it + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// Run f with the smallest compiled row-length bound that covers dmax (block-uniform).
template <typename F> __device__ __forceinline__ void dispatch_dmax(int dmax, F&& f) {
    if (dmax <= 2) f(IntTag<2>{});
    else if (dmax <= 6) f(IntTag<6>{});
    else f(IntTag<GADAPT_MAXD>{});
}


// In-kernel phase stamps: diagnostic builds only (-DGADAPT_STAMPS); never compiled into the shipped library.
#ifdef GADAPT_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
extern "C" int gadapt_debug_set_stamp_buffer(void* p) { g_stamp_buf = static_cast<unsigned long long*>(p); return 0; }
#define GADAPT_STAMP_L(buf, slot_)                                                                 \
    do {                                                                                           \
        if ((buf) && threadIdx.x == 256) {                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            (buf)[(size_t)blockIdx.x * 32 + (slot_)] = t_;                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    } while (0)
#define GADAPT_STAMP(buf, slot_)                                                                   \
    do {                                                                                           \
        if ((buf) && threadIdx.x == 0 && (slot_) < 32) {                                           \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            (buf)[(size_t)blockIdx.x * 32 + (slot_)] = t_;                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    } while (0)
#define GADAPT_STAMP_RT(buf, slot_)                                                                \
    do {                                                                                           \
        if ((buf) && threadIdx.x == 0 && (slot_) < 32) {                                           \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            (buf)[(size_t)blockIdx.x * 32 + (slot_)] = t_;                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }                                                                                          \
    } while (0)
#else
#define GADAPT_STAMP(buf, slot_) do { } while (0)
#define GADAPT_STAMP_L(buf, slot_) do { } while (0)
#define GADAPT_STAMP_RT(buf, slot_) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Per-lane channel vector.  A node's C channels are spread over LPN = C/FPL lanes; a lane owns NV = FPL/4
// float4 chunks: chunk indices sub, sub+LPN, ... so that the LPN lanes of one load instruction read 16*LPN
// contiguous bytes of the row.  FPL = 8 for C >= 32: fewer lanes per node means the per-node scalar work
// (softmax, masks, address arithmetic) and the DPP reduction steps are shared by twice as many nodes per wave.
// ------------------------------------------------------------------------------------------------
template <int NV> struct Vec {
    float4 v[NV];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = f4zero();
    }
};
#ifndef GADAPT_PK_DOT
#define GADAPT_PK_DOT 1
#endif
typedef float pk2f __attribute__((ext_vector_type(2)));
template <int NV> __device__ __forceinline__ float vdot(const Vec<NV>& a, const Vec<NV>& b) {
#if GADAPT_PK_DOT
    // two interleaved partial sums on v_pk_fma_f32 (a scalar fma chain is one instruction per element)
    pk2f s = pk2f{a.v[0].x, a.v[0].y} * pk2f{b.v[0].x, b.v[0].y};
    s += pk2f{a.v[0].z, a.v[0].w} * pk2f{b.v[0].z, b.v[0].w};
#pragma unroll
    for (int i = 1; i < NV; ++i) {
        s += pk2f{a.v[i].x, a.v[i].y} * pk2f{b.v[i].x, b.v[i].y};
        s += pk2f{a.v[i].z, a.v[i].w} * pk2f{b.v[i].z, b.v[i].w};
    }
    return s.x + s.y;
#else
    float s = dot4(a.v[0], b.v[0]);
#pragma unroll
    for (int i = 1; i < NV; ++i) s += dot4(a.v[i], b.v[i]);
    return s;
#endif
}
template <int NV> __device__ __forceinline__ void vaxpy(Vec<NV>& y, float a, const Vec<NV>& x) {
#pragma unroll
    for (int i = 0; i < NV; ++i) axpy4(y.v[i], a, x.v[i]);
}
template <int C> __device__ __forceinline__ Vec<Cfg<C>::NV> ld_vec(const float* __restrict__ base, int row, int sub) {
    Vec<Cfg<C>::NV> r;
#pragma unroll
    for (int i = 0; i < Cfg<C>::NV; ++i) r.v[i] = ld_row4<C>(base, row, sub + i * Cfg<C>::LPN);
    return r;
}
template <int C, bool XC> __device__ __forceinline__ Vec<Cfg<C>::NV> ld_xsel(const float* __restrict__ base, int row, int sub) {
    Vec<Cfg<C>::NV> r;
#pragma unroll
    for (int i = 0; i < Cfg<C>::NV; ++i) r.v[i] = ld_row4x<C, XC>(base, row, sub + i * Cfg<C>::LPN);
    return r;
}
template <int C> __device__ __forceinline__ void st_vec(float* __restrict__ base, int row, int sub, const Vec<Cfg<C>::NV>& x) {
#pragma unroll
    for (int i = 0; i < Cfg<C>::NV; ++i) st_row4<C>(base, row, sub + i * Cfg<C>::LPN, x.v[i]);
}
// padded LDS tile [TM][LD]
template <int C> __device__ __forceinline__ Vec<Cfg<C>::NV> lds_vec(const float* tile, int li, int sub) {
    Vec<Cfg<C>::NV> r;
#pragma unroll
    for (int i = 0; i < Cfg<C>::NV; ++i) r.v[i] = *reinterpret_cast<const float4*>(tile + li * Cfg<C>::LD + 4 * (sub + i * Cfg<C>::LPN));
    return r;
}
template <int C> __device__ __forceinline__ void lds_put(float* tile, int li, int sub, const Vec<Cfg<C>::NV>& x) {
#pragma unroll
    for (int i = 0; i < Cfg<C>::NV; ++i) *reinterpret_cast<float4*>(tile + li * Cfg<C>::LD + 4 * (sub + i * Cfg<C>::LPN)) = x.v[i];
}
// Maximum over the LPN lanes that share a node (non-negative values), in every lane: the DPP steps of group_sum.
template <int LPN> __device__ __forceinline__ float group_max(float v) {
    if constexpr (LPN >= 2) v = dpp_max<0xB1>(v);
    if constexpr (LPN >= 4) v = dpp_max<0x4E>(v);
    if constexpr (LPN >= 8) v = dpp_max<0x141>(v);
    if constexpr (LPN >= 16) v = dpp_max<0x140>(v);
    static_assert(LPN <= 16, "group_max: lane groups of at most one DPP row");
    return v;
}
// A row's channels of this lane -> the pre-split f16 layout of the row in an LDS tile (see GADAPT_PRESPLIT_*): chunk c (4
// channels) is half c & 1 of k-group c >> 1, whose 32 bytes hold 8 h pieces then 8 l pieces.  Returns the row's inverse scale.
template <int C> __device__ __forceinline__ float lds_put_split(float* tile, int li, int sub, const Vec<Cfg<C>::NV>& x) {
    using K = Cfg<C>;
    float m = 0.f;
#pragma unroll
    for (int q = 0; q < K::NV; ++q) m = fmaxf(fmaxf(fmaxf(m, fabsf(x.v[q].x)), fabsf(x.v[q].y)), fmaxf(fabsf(x.v[q].z), fabsf(x.v[q].w)));
    const Pow2 s = pow2_scale(group_max<K::LPN>(m));
#pragma unroll
    for (int q = 0; q < K::NV; ++q) {
        const int c = sub + q * K::LPN;
        const float t0 = x.v[q].x * s.s, t1 = x.v[q].y * s.s, t2 = x.v[q].z * s.s, t3 = x.v[q].w * s.s;
        f16x2 h01, h23, l01, l23;
        h01.x = (_Float16)t0; h01.y = (_Float16)t1; h23.x = (_Float16)t2; h23.y = (_Float16)t3;
        l01.x = (_Float16)(t0 - (float)h01.x); l01.y = (_Float16)(t1 - (float)h01.y);
        l23.x = (_Float16)(t2 - (float)h23.x); l23.y = (_Float16)(t3 - (float)h23.y);
        uint2* grp = reinterpret_cast<uint2*>(tile + li * K::LD + 8 * (c >> 1));
        grp[c & 1] = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
        grp[2 + (c & 1)] = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
    }
    return s.inv;
}


// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* x_in; float* x_out;
    const float* A; const float* p0; const float* lp;
    const int32_t* rowptr; const int32_t* col; const int32_t* meta;
    float* alpha_out;
    int n_nodes, n_tiles, residual_only, n_edges;
    unsigned long long* stamps;
    float* x_top4;                                              // nullable: columns 0..3 of the output rows, [N,4]
};

template <int NROWS, int NV> struct RowBuf {
    static constexpr int N = NROWS;
    Vec<NV> r[NROWS];
    int deg, el0;
};

// XC: x_in is the compact [N,4] encoder output (layer 0, identity encoder): see ld_row4x
template <int C, bool XC = false>
__global__ __launch_bounds__(Cfg<C>::NT, ((C >= GADAPT_ONE_WAVE_C && GADAPT_FWD_ONE_WAVE) ? 1 : GADAPT_WAVES_FWD)) void grand_fwd_kernel(FwdArgs p) {
    using K = Cfg<C>;
    using V = Vec<K::NV>;
    extern __shared__ float4 smem4[];
    float* ring = reinterpret_cast<float*>(smem4);               // RING slabs of x rows: slab s in slot s % RING
    float* ps = ring + K::RING * K::TILE_FLOATS;
    TileCsr<C, 0> csr;
    csr.bind(ps + K::TILE_FLOATS, p.rowptr, p.col, nullptr, p.meta, p.n_edges);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid / K::LPN, sub = tid % K::LPN;
    const float dt = p.lp[0], sc = p.lp[1];
    float* xs = ring;                                            // slab of the current tile (set per tile)
    bool win = false;                                            // current tile gathers from the LDS ring

    TileGemm<C, false, GADAPT_SPLIT_F16_F(C)> gemm;
    float arow[K::MFMA ? 1 : 4][K::MFMA ? 1 : C];
    float4 p0v = f4zero();
    if constexpr (K::MFMA) {
        gemm.init(lane, wave);
    } else {
        static_assert(K::MFMA || K::NV == 1, "VALU projection assumes one float4 per lane");
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) arow[t][c] = p.A[(4 * sub + t) * C + c];
        p0v = *reinterpret_cast<const float4*>(p.p0 + 4 * sub);
    }

    // PREA: the tile's rows are staged twice - fp32 for the edge walk and pre-split f16 (per-row scale) as the projection's A
    // operand, which no wave then splits again (hidden 128 without the LDS window: every wave needed every row); the P tile
    // holds acc x (column scale) and the reader applies the row scale and p0
    constexpr bool PREA = K::MFMA && K::RING != 3 && !XC && GADAPT_PRESPLIT_F && C / 4 == 32 && TileGemm<C, false, GADAPT_SPLIT_F16_F(C)>::F16;
    V p0c;
    if constexpr (PREA) {
#pragma unroll
        for (int q = 0; q < K::NV; ++q) p0c.v[q] = *reinterpret_cast<const float4*>(p.p0 + 4 * (sub + q * K::LPN));
    }
    auto projected = [&](int li) __attribute__((always_inline)) {                               // P_i = A x_i + p0 for this lane's channels
        V Pi;
        if constexpr (PREA) {
            Pi = lds_vec<C>(ps, li, sub);
            const float ri = ps[li * K::LD + C];
#pragma unroll
            for (int q = 0; q < K::NV; ++q) {
                Pi.v[q].x = fmaf(Pi.v[q].x, ri, p0c.v[q].x); Pi.v[q].y = fmaf(Pi.v[q].y, ri, p0c.v[q].y);
                Pi.v[q].z = fmaf(Pi.v[q].z, ri, p0c.v[q].z); Pi.v[q].w = fmaf(Pi.v[q].w, ri, p0c.v[q].w);
            }
        } else if constexpr (K::MFMA) {
            Pi = lds_vec<C>(ps, li, sub);
        } else {
            Pi.v[0] = p0v;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float xc = xs[li * K::LD + c];
                Pi.v[0].x = fmaf(arow[0][c], xc, Pi.v[0].x); Pi.v[0].y = fmaf(arow[1][c], xc, Pi.v[0].y);
                Pi.v[0].z = fmaf(arow[2][c], xc, Pi.v[0].z); Pi.v[0].w = fmaf(arow[3][c], xc, Pi.v[0].w);
            }
        }
        return Pi;
    };
    auto finish = [&](int li, int i, const V& m) __attribute__((always_inline)) {               // res = m - x (GRAND_plus.py:267); x + dt*res (GNN.py:291)
        const V xi = lds_vec<C>(xs, li, sub);
        V o;
#pragma unroll
        for (int q = 0; q < K::NV; ++q) {
            float4 r;
            r.x = m.v[q].x - xi.v[q].x; r.y = m.v[q].y - xi.v[q].y; r.z = m.v[q].z - xi.v[q].z; r.w = m.v[q].w - xi.v[q].w;
            if (!p.residual_only) {
                r.x = fmaf(dt, r.x, xi.v[q].x); r.y = fmaf(dt, r.y, xi.v[q].y); r.z = fmaf(dt, r.z, xi.v[q].z); r.w = fmaf(dt, r.w, xi.v[q].w);
            }
            o.v[q] = r;
        }
        if (p.x_out) st_vec<C>(p.x_out, i, sub, o);
        if (p.x_top4 && sub == 0) *reinterpret_cast<float4*>(p.x_top4 + 4 * (size_t)i) = o.v[0];
    };

    // Fast path, row length bounded by the compile-time DM: every lane issues exactly DM gathers (slots past
    // its own row length read some valid row and get weight 0), so there is no divergence.
    // WIN (compile time): the tile gathers from the LDS ring (col holds ring offsets) or from HBM/L2.  The two
    // variants must not share a join point: a pending-global-load possibility on one side makes hipcc put
    // s_waitcnt vmcnt(0) in front of every use of the rows on the other side too.
    auto fetch = [&](auto& b, int node0, int it, auto win_tag) __attribute__((always_inline)) {
        constexpr int DM = std::remove_reference_t<decltype(b)>::N;
        constexpr bool WIN = decltype(win_tag)::value != 0;
        const int li = it * K::SLOTS + slot;
        b.el0 = csr.rp[li] - csr.ebase;
        b.deg = (node0 + li < p.n_nodes) ? csr.rp[li + 1] - csr.rp[li] : 0;
        if constexpr (WIN) {
#pragma unroll
            for (int k = 0; k < DM; ++k) {
                const float* row = ring + csr.col[b.el0 + k];
#pragma unroll
                for (int q = 0; q < K::NV; ++q) b.r[k].v[q] = *reinterpret_cast<const float4*>(row + 4 * (sub + q * K::LPN));
            }
        } else {
#pragma unroll
            for (int k = 0; k < DM; ++k) b.r[k] = ld_xsel<C, XC>(p.x_in, csr.col[b.el0 + k], sub);
        }
    };

    auto consume = [&](const auto& b, int node0, int it) __attribute__((always_inline)) {
        constexpr int DM = std::remove_reference_t<decltype(b)>::N;
        const int li = it * K::SLOTS + slot;
        const int i = node0 + li;
        const V Pi = projected(li);
        const int deg = b.deg;
        float s[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) s[k] = vdot(Pi, b.r[k]);
        group_sum_n<K::LPN>(s);
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            s[k] = (k < deg) ? s[k] * sc : -INFINITY;
            mx = fmaxf(mx, s[k]);
        }
        if (deg == 0) mx = 0.f;                                  // keeps exp(-inf - mx) = 0 instead of NaN
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            s[k] = sm_exp(s[k] - mx);                            // masked slots: exp(-inf) = 0
            den += s[k];
        }
        const float inv = sm_rcp(den + 1e-16f);                 // PyG softmax epsilon
        V m; m.zero();
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            s[k] *= inv;
            vaxpy(m, s[k], b.r[k]);
        }
        if (p.alpha_out) {
            if constexpr (K::LPN >= DM) {
                const float mine = pick(s, sub);
                if (sub < deg) p.alpha_out[csr.ebase + b.el0 + sub] = mine;
            } else {
#pragma unroll
                for (int k = 0; k < DM; ++k)
                    if (k < deg && (k % K::LPN) == sub) p.alpha_out[csr.ebase + b.el0 + k] = s[k];
            }
        }
        if (i < p.n_nodes) finish(li, i, m);
    };

    // any row length, CSR straight from HBM: three sweeps over the row
    auto slow_node = [&](int node0, int it) __attribute__((always_inline)) {
        const int li = it * K::SLOTS + slot;
        const int i = node0 + li;
        if (i >= p.n_nodes) return;
        const V Pi = projected(li);
        const int e0 = p.rowptr[i], deg = p.rowptr[i + 1] - e0;
        float mx = -INFINITY;
        for (int k = 0; k < deg; ++k)
            mx = fmaxf(mx, group_sum<K::LPN>(vdot(Pi, ld_xsel<C, XC>(p.x_in, p.col[e0 + k], sub))) * sc);
        float den = 0.f;
        for (int k = 0; k < deg; ++k)
            den += sm_exp(group_sum<K::LPN>(vdot(Pi, ld_xsel<C, XC>(p.x_in, p.col[e0 + k], sub))) * sc - mx);
        const float inv = 1.0f / (den + 1e-16f);
        V m; m.zero();
        for (int k = 0; k < deg; ++k) {
            const V v = ld_xsel<C, XC>(p.x_in, p.col[e0 + k], sub);
            const float a = sm_exp(group_sum<K::LPN>(vdot(Pi, v)) * sc - mx) * inv;
            vaxpy(m, a, v);
            if (p.alpha_out && (k % K::LPN) == sub) p.alpha_out[e0 + k] = a;
        }
        finish(li, i, m);
    };

    // ---- rolling window: this workgroup walks the consecutive tiles [t0, t1); when it works on tile t the slabs
    //      t-1, t, t+1 of x are resident in the ring (one new slab per tile, requested one tile ahead)
    const TileChunk ch = tile_chunk(p.n_tiles);
    if (ch.t0 >= ch.t1) return;
    stagger_start<GADAPT_STAGGER_FWD>();
    auto slab_ptr = [&](int s_) __attribute__((always_inline)) { return ring + ((s_ + K::RING) % K::RING) * K::TILE_FLOATS; };
    // Staging runs TWO tiles ahead (two register sets, used alternately): every workgroup of the launch requests
    // its next slab at the same moment, so one tile of compute does not cover that burst.
    // C = 128 does not have the registers for that (nor for resident B fragments): one set, one tile ahead.
    constexpr int AHEAD = (C <= 64 || (C >= GADAPT_ONE_WAVE_C && GADAPT_FWD_ONE_WAVE)) ? 2 : 1;
    constexpr bool RESIDENT_B = (C <= 64) || (C >= GADAPT_ONE_WAVE_C && GADAPT_FWD_ONE_WAVE);
    typename TileCsr<C, 0>::Regs srA, srB;
    TileRows<C> xrA, xrB;
    // prologue loads in ONE memory round trip: the two slabs of the window's start, the tile metadata, the weight
    // fragments - all requested before the first use (the fragment split) waits
    // walk direction: every other workgroup of an XCD walks its chunk backwards (see the target pass)
    const int dir = (GADAPT_T_ALTERNATE && K::RING == 3 && ((blockIdx.x >> 3) & 1)) ? -1 : 1;
    const int tb = dir > 0 ? ch.t0 : ch.t1 - 1, n_my = ch.t1 - ch.t0;
    if constexpr (K::RING == 3) {
        xrA.template issue_sel<XC>(p.x_in, (tb - dir) * K::TM, p.n_nodes, tid);
        xrB.template issue_sel<XC>(p.x_in, tb * K::TM, p.n_nodes, tid);
    }
    const int4 mreg = csr.metas_issue(tb, dir, p.n_tiles, tid);
    if constexpr (K::MFMA && RESIDENT_B) gemm.load(p.A, p.p0);   // B fragments stay in registers for the whole launch
    csr.metas_commit(mreg, tid);
    if constexpr (K::RING == 3) {
        xrA.template commit_sel<XC>(slab_ptr(tb - dir), p.n_nodes, tid);
        xrB.template commit_sel<XC>(slab_ptr(tb), p.n_nodes, tid);
    }
    __syncthreads();                                             // tile metadata visible
    xrA.template issue_sel<XC>(p.x_in, (tb + dir * K::LEAD) * K::TM, p.n_nodes, tid);   // rows past N come back as zeros
    csr.issue(srA, 0, tb * K::TM, p.n_nodes, tid);
    if constexpr (AHEAD == 2) {
        xrB.template issue_sel<XC>(p.x_in, (tb + dir * (1 + K::LEAD)) * K::TM, p.n_nodes, tid);
        csr.issue(srB, 1, (tb + dir) * K::TM, p.n_nodes, tid);
    }

    // Order inside a tile: GEMM first (its A operand, slab t, was committed during tile t-1; B fragments are
    // resident), THEN commit the prefetched slab t+1 / CSR slice of tile t.  vmcnt is in-order and hipcc waits
    // conservatively (vmcnt(0)) before the commit reads the prefetch registers, so anything still in flight -
    // the previous tile's output stores included - is paid for there: behind the MFMA phase it is (mostly) free.
    auto do_tile = [&](int k, TileRows<C>& xr, typename TileCsr<C, 0>::Regs& sr) __attribute__((always_inline)) {
        const int t = tb + k * dir;
        const int node0 = t * K::TM;
        const int tslot = k * 8;                                // stamps of the first 4 tiles of this workgroup
        GADAPT_STAMP(p.stamps, tslot + 0);
        xs = slab_ptr(t);
        int dmax;
        if constexpr (K::RING == 3) {
            if constexpr (K::MFMA && !RESIDENT_B) gemm.load(p.A, p.p0);
            if constexpr (K::MFMA) gemm.template run<(XC && GADAPT_XC_ONE_KSTEP) ? 1 : (C >= 16 ? C / 16 : 1)>(xs, ps);
            GADAPT_STAMP(p.stamps, tslot + 1);
            win = sr.meta.w != 0;
            xr.template commit_sel<XC>(slab_ptr(t + dir), p.n_nodes, tid);
            dmax = csr.commit(sr, tid, node0, win ? t : -1);
            __syncthreads();                                    // P tile, the next slab and the CSR slice are complete
        } else {                                                // no window: the tile itself is staged first, then projected
            win = false;
            typename decltype(gemm)::BRaw braw;
            if constexpr (K::MFMA && !RESIDENT_B) gemm.load_issue(p.A, braw);   // weight rows requested first: in flight under the commits
            if constexpr (PREA) xr.commit_presplit(xs, ps, p.n_nodes, tid); else
            xr.template commit_sel<XC>(xs, p.n_nodes, tid);
            dmax = csr.commit(sr, tid, node0, -1);
            if constexpr (K::MFMA && !RESIDENT_B) gemm.load_finish(p.A, p.p0, braw);
            __syncthreads();
        }
        {   // this register set's next job: tile t+AHEAD (slab t+AHEAD+LEAD).  Unconditional (clamped past the chunk end): see issue()
            csr.issue(sr, k + AHEAD, (t + AHEAD * dir) * K::TM, p.n_nodes, tid);
            xr.template issue_sel<XC>(p.x_in, (t + (AHEAD + K::LEAD) * dir) * K::TM, p.n_nodes, tid);
        }
        if constexpr (K::RING != 3) {
            if constexpr (PREA) {
                f32x16 acc[decltype(gemm)::BPW];
                gemm.accumulate_presplit(ps, acc);
                __syncthreads();                                // in place: every wave has read its operand rows
                gemm.store_presplit(ps, acc);
                __syncthreads();
            } else if constexpr (K::MFMA) { gemm.template run<(XC && GADAPT_XC_ONE_KSTEP) ? 1 : (C >= 16 ? C / 16 : 1)>(xs, ps); __syncthreads(); }
            GADAPT_STAMP(p.stamps, tslot + 1);
        }
        GADAPT_STAMP(p.stamps, tslot + 2);
        if (dmax >= 0) {
            dispatch_dmax(dmax, [&](auto tag) {
                auto walk = [&](auto win_tag) __attribute__((always_inline)) {
                    run_pipeline<K::ITERS, RowBuf<decltype(tag)::value, K::NV>>(
                        [&](auto& b, int it) { fetch(b, node0, it, win_tag); },
                        [&](const auto& b, int it) { consume(b, node0, it); GADAPT_STAMP(p.stamps, tslot + 3 + (it & 3)); },
                        [&]() {});
                };
                if (win) walk(IntTag<1>{}); else walk(IntTag<0>{});
            });
        } else {
#pragma unroll 1
            for (int it = 0; it < K::ITERS; ++it) slow_node(node0, it);
        }
        __syncthreads();                                        // edge walk done: P tile and ring slot t-1 may be rewritten
        GADAPT_STAMP(p.stamps, tslot + 7);
    };
    if constexpr (AHEAD == 2) {
        for (int k = 0; k < n_my; k += 2) {
            do_tile(k, xrA, srA);
            if (k + 1 < n_my) do_tile(k + 1, xrB, srB);
        }
    } else {
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) do_tile(k, xrA, srA);
    }
}

// ------------------------------------------------------------------------------------------------
// backward, target pass
// ------------------------------------------------------------------------------------------------
struct BwdTArgs {
    const float* x_in; const float* g_in; const float* alpha;
    const float* A; const float* lp;
    const int32_t* rowptr; const int32_t* col; const int32_t* tpos; const int32_t* meta;
    float2* edge_ws; float* dxd; float* slab; float* sums_out;   // sums_out: d dt of this layer (one float)
    int n_nodes, n_tiles, accumulate, residual_only, n_edges;
    unsigned long long* stamps;
    int g_cols;                                                 // GC kernels: g_in is [N,g_cols] (zero beyond), else unused
    float* sums_sc_out;                                         // d score_scale of this layer (one float); set with sums_out
    int c;                                                      // hidden size (read by grand_bwd_target_compact_kernel only)
    int g_stride;                                               // ... and the row pitch of its g_in, in floats
    int sums_partials;                                          // 1: sums_out / sums_sc_out are per-workgroup arrays [gridDim.x] (written, not added to)
};

template <int NROWS, int NV> struct TBuf {
    static constexpr int N = NROWS;
    Vec<NV> r[NROWS];
    int deg, el0;
};

// SUMS: 1 = also reduce d/d(dt) (learn_step), 2 = d/d(dt) and d/d(score_scale) (learnable temperature).  Separate instantiations:
// hipcc otherwise sinks the per-edge log terms behind the pipeline and keeps dozens of registers alive for them.
//
// Rolling window like the forward: a workgroup walks consecutive tiles with slabs t-1, t, t+1 of x in an LDS ring,
// so on mesh-ordered graphs the x_j gathers are LDS reads.  Per tile the fourth LDS tile holds g (staged with the
// ring slab), then dP (written over g row by row by the lanes that read it), then dP A (in place).
// GC: the upstream gradient is compact, [N,g_cols] (top layer: backward of the x[:, :dim] slice) - only its staging differs.
// XC: x_in is the compact [N,4] encoder output (layer 0): see ld_row4x.  No source pass follows such a launch (d x0 is
// not wanted), so it skips what only the source pass reads: the per-edge scratch, dP A and dxd.
// DA: this launch accumulates the weight-gradient partials (dA, dp0).  false when a source pass follows that does it
// instead (dA = sum_i dP_i x_i^T = sum_j x_j y_j^T with y_j = sum_i ds_ij x_i, the vector the source pass forms anyway;
// dp0 = sum_j sigma_j x_j): the target pass then has no dA phase, no accumulators and no slab flush.
// D4: only columns 0..3 of dxd are wanted (the layer BELOW reads the compact [N,4] encoder output, so its backward contracts
// d alpha = dt <g_i, x_k> over four columns and the source pass that follows this launch produces just those): dxd is written
// as [N,4] and dP A[:, :4] is four dot products per node on the vector ALU - no projection phase on the matrix cores, no
// pre-split copy, two barriers fewer per tile, and the dense dxd matrix (N C floats) is neither written nor read back.
template <int C, int SUMS, bool GC = false, bool XC = false, bool DA = true, bool D4 = false>
__global__ __launch_bounds__(Cfg<C>::NT, (C >= GADAPT_ONE_WAVE_C ? 1 : GADAPT_WAVES_BWD_T)) void grand_bwd_target_kernel(BwdTArgs p) {
    static_assert(DA || (!SUMS && !XC), "only the plain variants hand dA to the source pass");
    static_assert(!D4 || (DA && !XC), "D4: a source pass follows, the weight gradients stay here");
    using K = Cfg<C>;
    using V = Vec<K::NV>;
    extern __shared__ float4 smem4[];
    float* ring = reinterpret_cast<float*>(smem4);
    float* ds = ring + K::RING_T * K::TILE_FLOATS;              // g tile -> dP tile -> dP A
    using CsrT = TileCsr<C, 1, 1>;                              // aux = forward alpha (target order), ext = tpos
    CsrT csr;
    csr.bind(ds + K::TILE_FLOATS, p.rowptr, p.col, p.alpha, p.meta, p.n_edges, p.tpos);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid / K::LPN, sub = tid % K::LPN;
    // out = base*x + dt*(m - x): Euler step (base 1) or bare residual (base 0, dt 1)
    const float dt = p.residual_only ? 1.0f : p.lp[0], sc = p.lp[1];
    const float w1 = (p.residual_only ? 0.0f : 1.0f) - dt;
    constexpr int ROW = C * C + C;                              // slab row: dA then dp0
    float* xs = ring;                                           // slab of the current tile

    // PRE: dP A takes dP from a pre-split f16 copy (lds_put_split) made after the edge walk in the window slot of slab t-1,
    // which nothing reads between that barrier and the next tile's commit; dinv = its row scales
    constexpr bool PRE = K::MFMA && !XC && !D4 && GADAPT_PRESPLIT_T && K::RING_T == 3 && TileGemm<C, true>::SPLIT && C < GADAPT_BWD_JIT_B_C;
    float dinv[K::ITERS];
    TileGemm<C, true, GADAPT_SPLIT_F16_T || PRE> gemm;          // dxd = dP A
    float acol[(K::MFMA || D4) ? 1 : 4][(K::MFMA || D4) ? 1 : C];   // VALU: A[o][4sub+t]
    float a4[D4 ? K::NV : 1][4][4];                             // D4: A[o][c], o = this lane's channels, c = 0..3
    if constexpr (D4) {
#pragma unroll
        for (int q = 0; q < K::NV; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 r = *reinterpret_cast<const float4*>(p.A + (size_t)(4 * (sub + q * K::LPN) + e) * C);
                a4[q][e][0] = r.x; a4[q][e][1] = r.y; a4[q][e][2] = r.z; a4[q][e][3] = r.w;
            }
        if constexpr (K::MFMA) gemm.init(lane, wave);
    } else if constexpr (K::MFMA) {
        gemm.init(lane, wave);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < C; ++o) acol[t][o] = p.A[o * C + 4 * sub + t];
    }
    // dA partial accumulators.  MFMA: NB2 = CB*CB 32x32 blocks of dA; >= NW blocks -> NB2/NW per wave over
    // all TM nodes, 1 block -> the 4 waves split the node range.  VALU: thread owns element tid % C^2 for
    // node subset tid / C^2.
    constexpr int NB2 = K::CB * K::CB;
    constexpr int DPW = K::MFMA ? (NB2 >= K::NW ? NB2 / K::NW : 1) : 1;
    static_assert(!K::MFMA || NB2 >= K::NW || (NB2 == 1 && K::NW == 4), "dA blocks per wave");
    f32x16 dacc[DPW];
    float dav = 0.f;
    if constexpr (K::MFMA) {
#pragma unroll
        for (int b = 0; b < DPW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[b][r] = 0.f;
    }
    V dp0acc; dp0acc.zero();
    float sum_ddt = 0.f, sum_dsc = 0.f;
    V gk[K::ITERS];                                             // (base - dt) g_i of this lane's node slots, for the epilogue

    // Dg (SUMS): sum_k alpha_ik <g_i, x_k> of this node (group-uniform)
    auto finish = [&](int li, int i, int it, const V& gi, float Dg, const V& dP) __attribute__((always_inline)) {
        if (i < p.n_nodes) {
            if constexpr (SUMS) {
                // d dt = sum_i <g_i, m_i - x_i>   (GNN.py:288-289 learn_step) with <g_i, m_i> = sum_k alpha_ik <g_i, x_k> = Dg, which
                // the softmax backward has anyway: no aggregated row m_i is formed (it cost 8 registers and 48 fmas per node slot
                // and made these instantiations spill).  One lane of the group adds Dg, every lane its channels of -<g_i, x_i>.
                const V xi = lds_vec<C>(xs, li, sub);
                float own = (sub == 0) ? Dg : 0.f;
#pragma unroll
                for (int q = 0; q < K::NV; ++q)
                    own -= gi.v[q].x * xi.v[q].x + gi.v[q].y * xi.v[q].y + gi.v[q].z * xi.v[q].z + gi.v[q].w * xi.v[q].w;
                sum_ddt += own;
            }
            if constexpr (DA) {
#pragma unroll
                for (int q = 0; q < K::NV; ++q) {
                    dp0acc.v[q].x += dP.v[q].x; dp0acc.v[q].y += dP.v[q].y; dp0acc.v[q].z += dP.v[q].z; dp0acc.v[q].w += dP.v[q].w;
                }
            }
        }
        if constexpr (D4) {
            // dxd[i][0..3] = (base - dt) g_i[0..3] + sum_o dP_i[o] A[o][0..3]: this lane's channels, then the group
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < K::NV; ++q) {
                const float d[4] = {dP.v[q].x, dP.v[q].y, dP.v[q].z, dP.v[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int c = 0; c < 4; ++c) t4[c] = fmaf(d[e], a4[q][e][c], t4[c]);
            }
            group_sum_n<K::LPN>(t4);
            gk[it].v[0] = make_float4(fmaf(w1, gi.v[0].x, t4[0]), fmaf(w1, gi.v[0].y, t4[1]), fmaf(w1, gi.v[0].z, t4[2]), fmaf(w1, gi.v[0].w, t4[3]));   // lane sub == 0 holds columns 0..3
        } else {
#pragma unroll
            for (int q = 0; q < K::NV; ++q) {
                gk[it].v[q].x = w1 * gi.v[q].x; gk[it].v[q].y = w1 * gi.v[q].y; gk[it].v[q].z = w1 * gi.v[q].z; gk[it].v[q].w = w1 * gi.v[q].w;
            }
        }
        lds_put<C>(ds, li, sub, dP);                            // over g_i: only this lane group reads that row
    };
    // SUMS instantiations take the edge dot products with g_i itself and fold dt into the score scale afterwards
    // (d alpha_ik = dt <g_i, x_k>): Dg above is then free, and dt = 0 needs no special case
    const float scl = SUMS ? sc * dt : sc;

    // WIN (compile time): gathers from the LDS ring (col holds ring offsets) or from HBM/L2; see the forward kernel
    auto fetch = [&](auto& b, int node0, int it, auto win_tag) __attribute__((always_inline)) {
        constexpr int DM = std::remove_reference_t<decltype(b)>::N;
        constexpr bool WIN = decltype(win_tag)::value != 0;
        const int li = it * K::SLOTS + slot;
        b.el0 = csr.rp[li] - csr.ebase;
        b.deg = (node0 + li < p.n_nodes) ? csr.rp[li + 1] - csr.rp[li] : 0;
        if constexpr (WIN) {
#pragma unroll
            for (int k = 0; k < DM; ++k) {
                const float* row = ring + csr.col[b.el0 + k];
#pragma unroll
                for (int q = 0; q < K::NV; ++q) b.r[k].v[q] = *reinterpret_cast<const float4*>(row + 4 * (sub + q * K::LPN));
            }
        } else {
#pragma unroll
            for (int k = 0; k < DM; ++k) b.r[k] = ld_xsel<C, XC>(p.x_in, csr.col[b.el0 + k], sub);   // k >= deg: some valid row, weight 0
        }
    };

    auto consume = [&](const auto& b, int node0, int it) __attribute__((always_inline)) {
        constexpr int DM = std::remove_reference_t<decltype(b)>::N;
        const int li = it * K::SLOTS + slot;
        const int i = node0 + li;
        const int deg = b.deg, e0 = b.el0;
        const V gi = lds_vec<C>(ds, li, sub);                   // rows past N were staged as zeros
        V dm;
#pragma unroll
        for (int q = 0; q < K::NV; ++q) {
            if constexpr (SUMS) { dm.v[q] = gi.v[q]; continue; }
            dm.v[q].x = dt * gi.v[q].x; dm.v[q].y = dt * gi.v[q].y; dm.v[q].z = dt * gi.v[q].z; dm.v[q].w = dt * gi.v[q].w;
        }
        float a[DM], da[DM];
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const float av = csr.aux[e0 + k];
            a[k] = (k < deg) ? av : 0.f;
            da[k] = vdot(dm, b.r[k]);
        }
        group_sum_n<K::LPN>(da);
        float D = 0.f;
#pragma unroll
        for (int k = 0; k < DM; ++k) D = fmaf(a[k], da[k], D);
        V dP; dP.zero();
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            const float dsp = a[k] * (da[k] - D);                   // d(score') [/ dt: SUMS], score' = sc * <P_i,x_j>
            da[k] = dsp * scl;                                      // reuse: d<P_i,x_j>
            vaxpy(dP, da[k], b.r[k]);
        }
        // per-edge scratch for the source pass, one edge per lane of the group; SUMS: the same lane adds its edge's term of
        // d/d(score_scale) = (1/sc) sum_e d(score')_e log alpha_e  (the softmax backward sums to zero per target, so the
        // log-partition term drops out) - one log per lane and node slot instead of DM group-uniform ones
        if constexpr (XC && SUMS < 2) {
        } else if constexpr (K::LPN >= DM) {
            const float am = pick(a, sub), dm_ = pick(da, sub);
            if constexpr (SUMS > 1) { if (sub < deg && am > 0.f) sum_dsc = fmaf(dm_, __logf(am), sum_dsc); }
            if constexpr (!XC) {
#ifdef GADAPT_ABL_EDGEWS_LINEAR
                if (sub < deg) p.edge_ws[csr.ebase + e0 + sub] = make_float2(am * dt, dm_);
#elif !defined(GADAPT_ABL_NO_EDGEWS)
                if (sub < deg) p.edge_ws[csr.ext[e0 + sub]] = make_float2(am * dt, dm_);
#endif
            }
        } else {
#pragma unroll
            for (int k = 0; k < DM; ++k)
                if (k < deg && (k % K::LPN) == sub) {
                    if constexpr (!XC) p.edge_ws[csr.ext[e0 + k]] = make_float2(a[k] * dt, da[k]);
                    if constexpr (SUMS > 1) { if (a[k] > 0.f) sum_dsc = fmaf(da[k], __logf(a[k]), sum_dsc); }
                }
        }
        finish(li, i, it, gi, D, dP);
    };

    auto slow_node = [&](int node0, int it) __attribute__((always_inline)) {
        const int li = it * K::SLOTS + slot;
        const int i = node0 + li;
        V dP; dP.zero();
        float D = 0.f;
        const V gi = lds_vec<C>(ds, li, sub);
        if (i < p.n_nodes) {
            V dm;
#pragma unroll
            for (int q = 0; q < K::NV; ++q) {
                if constexpr (SUMS) { dm.v[q] = gi.v[q]; continue; }
                dm.v[q].x = dt * gi.v[q].x; dm.v[q].y = dt * gi.v[q].y; dm.v[q].z = dt * gi.v[q].z; dm.v[q].w = dt * gi.v[q].w;
            }
            const int e0 = p.rowptr[i], deg = p.rowptr[i + 1] - e0;
            for (int k = 0; k < deg; ++k)
                D = fmaf(p.alpha[e0 + k], group_sum<K::LPN>(vdot(dm, ld_xsel<C, XC>(p.x_in, p.col[e0 + k], sub))), D);
            for (int k = 0; k < deg; ++k) {
                const V v = ld_xsel<C, XC>(p.x_in, p.col[e0 + k], sub);
                const float ak = p.alpha[e0 + k];
                const float dsp = ak * (group_sum<K::LPN>(vdot(dm, v)) - D);
                const float dss = dsp * scl;
                vaxpy(dP, dss, v);
                if ((k % K::LPN) == sub) {
                    if constexpr (!XC) p.edge_ws[p.tpos[e0 + k]] = make_float2(ak * dt, dss);
                    if constexpr (SUMS > 1) { if (ak > 0.f) sum_dsc = fmaf(dss, __logf(ak), sum_dsc); }
                }
            }
        }
        finish(li, i, it, gi, D, dP);
    };

    // dxd rows of a finished tile leave the registers only after the NEXT tile's staging: vmcnt is in-order and hipcc
    // waits vmcnt(0) before the staging reads its prefetch registers, so stores issued just before it would be
    // waited for (a full write round trip per tile).
    auto store_dxd = [&](int node0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < K::ITERS; ++it) {
            const int i = node0 + it * K::SLOTS + slot;
            if constexpr (D4) {
                if (i < p.n_nodes && sub == 0) *reinterpret_cast<float4*>(p.dxd + 4 * (size_t)i) = gk[it].v[0];
            } else if (i < p.n_nodes) {
                if constexpr (GADAPT_T_STREAM) {
#pragma unroll
                    for (int q = 0; q < K::NV; ++q) st_row4_nt<C>(p.dxd, i, sub + q * K::LPN, gk[it].v[q]);
                } else {
                    st_vec<C>(p.dxd, i, sub, gk[it]);
                }
            }
        }
    };
    const TileChunk ch = tile_chunk(p.n_tiles);                 // an empty chunk still flushes its (zero) slab row
    if (ch.t0 < ch.t1) {
        stagger_start<GADAPT_STAGGER_T>();
        auto slab_ptr = [&](int s_) __attribute__((always_inline)) { return ring + ((s_ + K::RING_T) % K::RING_T) * K::TILE_FLOATS; };
        typename CsrT::Regs sr;
        TileRows<C> xr, gr;
        constexpr bool RESIDENT_B = TileGemm<C, true>::SPLIT && C < GADAPT_BWD_JIT_B_C;  // split fragments are built once per launch
        // prologue loads in ONE memory round trip: the two slabs of the window's start (gr is free until the first g tile),
        // the tile metadata, the weight fragments - all requested before the first use (the fragment split) waits
        // Walk direction: every other workgroup of an XCD walks its chunk BACKWARDS.  Neighbouring chunks then meet at their common
        // boundary at the same time - both at the start or both at the end of the launch - so the halo slab one of them
        // stages is the slab the other one stages for its own tile within the same few microseconds: an L2 hit instead of a
        // second trip to the fabric (with every chunk walked forwards the two reads are a whole launch apart: x came in 1.5x).
        const int dir = (GADAPT_T_ALTERNATE && K::RING_T == 3 && ((blockIdx.x >> 3) & 1)) ? -1 : 1;
        const int tb = dir > 0 ? ch.t0 : ch.t1 - 1, n_my = ch.t1 - ch.t0;
        if constexpr (K::RING_T == 3) {
            xr.template issue_sel<XC>(p.x_in, (tb - dir) * K::TM, p.n_nodes, tid);
            gr.template issue_sel<XC>(p.x_in, tb * K::TM, p.n_nodes, tid);
        }
        const int4 mreg = csr.metas_issue(tb, dir, p.n_tiles, tid);
        if constexpr (K::MFMA && RESIDENT_B && !XC && !D4) gemm.load(p.A, nullptr);   // B fragments stay in registers for the whole launch
        csr.metas_commit(mreg, tid);
        if constexpr (K::RING_T == 3) {
            xr.template commit_sel<XC>(slab_ptr(tb - dir), p.n_nodes, tid);
            gr.template commit_sel<XC>(slab_ptr(tb), p.n_nodes, tid);
        }
        __syncthreads();                                        // tile metadata visible
        // C = 128 has no registers to hold a tile across the edge walk: it stages at the top of the tile instead
        constexpr bool PREFETCH = (C <= GADAPT_T_PREFETCH_MAX_C);
        if constexpr (PREFETCH) {
            xr.template issue_sel<XC>(p.x_in, (tb + dir * K::LEAD_T) * K::TM, p.n_nodes, tid);
            if constexpr (GC) gr.issue_compact(p.g_in, p.g_cols, tb * K::TM, p.n_nodes, tid); else gr.template issue<GADAPT_T_STREAM != 0>(p.g_in, tb * K::TM, p.n_nodes, tid);
            csr.issue(sr, 0, tb * K::TM, p.n_nodes, tid);
        }
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) {
            const int t = tb + k * dir;
            const int node0 = t * K::TM;
            const int tslot = k * 8;
            if constexpr (!PREFETCH) {
                xr.template issue_sel<XC>(p.x_in, (t + dir * K::LEAD_T) * K::TM, p.n_nodes, tid);
                if constexpr (GC) gr.issue_compact(p.g_in, p.g_cols, t * K::TM, p.n_nodes, tid); else gr.template issue<GADAPT_T_STREAM != 0>(p.g_in, t * K::TM, p.n_nodes, tid);
                csr.issue(sr, k, t * K::TM, p.n_nodes, tid);
            }
#ifdef GADAPT_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // diagnostic: separates the prefetch wait from the LDS writes
#endif
            GADAPT_STAMP(p.stamps, tslot + 0);
            xs = slab_ptr(t);
            const bool win = K::RING_T == 3 && sr.meta.w != 0;
            xr.template commit_sel<XC>(slab_ptr(t + dir * K::LEAD_T), p.n_nodes, tid);
            gr.template commit_sel<GC>(ds, p.n_nodes, tid);
            const int dmax = csr.template commit<K::RING_T>(sr, tid, node0, win ? t : -1);
            GADAPT_STAMP(p.stamps, tslot + 1);
            __syncthreads();
            if constexpr (!XC) { if (k > 0) store_dxd((t - dir) * K::TM); }   // previous tile's result: see store_dxd
            GADAPT_STAMP(p.stamps, tslot + 2);
            // ---- edge phase: dP_i per node -> LDS
            if (dmax >= 0) {
                dispatch_dmax(dmax, [&](auto tag) {
                    auto walk = [&](auto win_tag) __attribute__((always_inline)) {
                        run_pipeline<K::ITERS, TBuf<decltype(tag)::value, K::NV>, (C >= GADAPT_ONE_WAVE_C && GADAPT_T_TWO_BUFFERS)>(
                            [&](auto& b, int it) { fetch(b, node0, it, win_tag); },
                            [&](const auto& b, int it) { consume(b, node0, it); },
                            [&]() {});
                    };
                    if (win) walk(IntTag<1>{}); else walk(IntTag<0>{});
                });
            } else {
#pragma unroll
                for (int it = 0; it < K::ITERS; ++it) slow_node(node0, it);
            }
            GADAPT_STAMP(p.stamps, tslot + 3);
            if constexpr (PREFETCH) {
                // request the next tile (unconditional, clamped past the end: see issue()).  Here rather than at the top
                // of the tile: the edge phase needs every register, and the MFMA phases below cover the round trip.
                csr.issue(sr, k + 1, (t + dir) * K::TM, p.n_nodes, tid);
                xr.template issue_sel<XC>(p.x_in, (t + dir * (1 + K::LEAD_T)) * K::TM, p.n_nodes, tid);
                if constexpr (GC) gr.issue_compact(p.g_in, p.g_cols, (t + dir) * K::TM, p.n_nodes, tid); else gr.template issue<GADAPT_T_STREAM != 0>(p.g_in, (t + dir) * K::TM, p.n_nodes, tid);
            }
            __syncthreads();
            GADAPT_STAMP(p.stamps, tslot + 4);
            if constexpr (PRE) {
                float* cv = slab_ptr(t - dir);                  // the slab behind: dead until the next tile's commit puts the slab two ahead there
#pragma unroll
                for (int it = 0; it < K::ITERS; ++it) {
                    const int li = it * K::SLOTS + slot;
                    dinv[it] = lds_put_split<C>(cv, li, sub, lds_vec<C>(ds, li, sub));   // own rows: written by this lane group in finish()
                }
            }
            // ---- dA partial:  dA[o][c] += sum_node dP[node][o] x[node][c]
#if GADAPT_T_MFMA_PRIO
            __builtin_amdgcn_s_setprio(GADAPT_T_MFMA_PRIO);
#endif
#ifdef GADAPT_ABL_NO_DA
            if (p.n_nodes >= 0) {} else                          // diagnostic build: the dA phase never runs
#endif
            if constexpr (!DA) {
            } else if constexpr (K::MFMA && TileGemm<C, true>::SPLIT && GADAPT_DA_F16) {
                // two-piece f16 form (see split8h): k = node, 16 nodes per step; lane (i, h) feeds nodes 8h..8h+7 of the step.
                // Scales: per quad of dP channels and per x channel over this wave's nodes; fresh accumulator per tile.
                const int h = lane >> 5, r31 = lane & 31;
                constexpr int NODES = (NB2 >= K::NW) ? K::TM : K::TM / 4;   // one 32x32 block: the waves split the nodes
                constexpr int KSN = NODES / 16;
                const int nbase = (NB2 >= K::NW) ? 0 : wave * NODES;
                const int ob = (NB2 >= K::NW) ? (wave * DPW) / K::CB : 0, cb0 = (NB2 >= K::NW) ? (wave * DPW) % K::CB : 0;
                Split2 as[KSN];
                float ainv[4];
                {
                    float av[KSN][8];
                    float mx = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                        for (int e = 0; e < 8; ++e) av[ks][e] = ds[(nbase + 16 * ks + 8 * h + e) * K::LD + ob * 32 + r31];
#pragma unroll
                    for (int ks = 0; ks < KSN; ++ks) mx = absmax8(av[ks], mx);
                    const Pow2 sa = pow2_scale(quad_rows_max(mx));
                    quad_rows_inverse(sa.inv, h, ainv);
#pragma unroll
                    for (int ks = 0; ks < KSN; ++ks) as[ks] = split8h(av[ks], sa.s);
                }
#pragma unroll
                for (int b = 0; b < DPW; ++b) {
                    float bv[KSN][8];
                    float mx = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                        for (int e = 0; e < 8; ++e) bv[ks][e] = xs[(nbase + 16 * ks + 8 * h + e) * K::LD + (cb0 + b) * 32 + r31];
#pragma unroll
                    for (int ks = 0; ks < KSN; ++ks) mx = absmax8(bv[ks], mx);
                    const Pow2 sb = pow2_scale(half_max(mx));
                    f32x16 t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KSN; ++ks) t = mma3(as[ks], split8h(bv[ks], sb.s), t);
                    const float u[4] = {ainv[0] * sb.inv, ainv[1] * sb.inv, ainv[2] * sb.inv, ainv[3] * sb.inv};
#pragma unroll
                    for (int r = 0; r < 16; ++r) dacc[b][r] = fmaf(t[r], u[r >> 2], dacc[b][r]);
                }
            } else if constexpr (K::MFMA && TileGemm<C, true>::SPLIT) {
                // bf16 three-piece form (see split8): k = node, 16 nodes per step; lane (i, h) feeds nodes 8h..8h+7 of the step
                const int h = lane >> 5, r31 = lane & 31;
                constexpr int NODES = (NB2 >= K::NW) ? K::TM : K::TM / 4;   // one 32x32 block: the waves split the nodes
                const int nbase = (NB2 >= K::NW) ? 0 : wave * NODES;
                const int ob = (NB2 >= K::NW) ? (wave * DPW) / K::CB : 0, cb0 = (NB2 >= K::NW) ? (wave * DPW) % K::CB : 0;
#pragma unroll GADAPT_DA_UNROLL
                for (int ks = 0; ks < NODES / 16; ++ks) {
                    const int n0 = nbase + 16 * ks + 8 * h;
                    float av[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) av[e] = ds[(n0 + e) * K::LD + ob * 32 + r31];
                    const Split3 as = split8(av);
#if GADAPT_DA_BPREFETCH
                    // several blocks per wave (hidden 128): the x columns of block b + 1 are requested before block b is split
                    // and multiplied
                    float bv[2][8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[0][e] = xs[(n0 + e) * K::LD + cb0 * 32 + r31];
#pragma unroll
                    for (int b = 0; b < DPW; ++b) {
                        // compact layer input: x is zero beyond column 3, so only column block 0 of dA gets anything
                        if (XC && GADAPT_XC_ONE_KSTEP && cb0 + b > 0) continue;
                        if (b + 1 < DPW) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) bv[(b + 1) & 1][e] = xs[(n0 + e) * K::LD + (cb0 + b + 1) * 32 + r31];
                        }
                        const Split3 bs = split8(bv[b & 1]);
                        dacc[b] = mfma_bf16(as.h, bs.l, dacc[b]);
                        dacc[b] = mfma_bf16(as.l, bs.h, dacc[b]);
                        dacc[b] = mfma_bf16(as.m, bs.m, dacc[b]);
                        dacc[b] = mfma_bf16(as.h, bs.m, dacc[b]);
                        dacc[b] = mfma_bf16(as.m, bs.h, dacc[b]);
                        dacc[b] = mfma_bf16(as.h, bs.h, dacc[b]);
                    }
#else
#pragma unroll
                    for (int b = 0; b < DPW; ++b) {
                        float bv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) bv[e] = xs[(n0 + e) * K::LD + (cb0 + b) * 32 + r31];
                        const Split3 bs = split8(bv);
                        dacc[b] = mfma_bf16(as.h, bs.l, dacc[b]);
                        dacc[b] = mfma_bf16(as.l, bs.h, dacc[b]);
                        dacc[b] = mfma_bf16(as.m, bs.m, dacc[b]);
                        dacc[b] = mfma_bf16(as.h, bs.m, dacc[b]);
                        dacc[b] = mfma_bf16(as.m, bs.h, dacc[b]);
                        dacc[b] = mfma_bf16(as.h, bs.h, dacc[b]);
                    }
#endif
                }
            } else if constexpr (K::MFMA) {
                const int h = lane >> 5, r31 = lane & 31;
                if constexpr (NB2 >= K::NW) {
                    // wave owns o-block `ob` and DPW consecutive c-blocks
                    const int ob = (wave * DPW) / K::CB, cb0 = (wave * DPW) % K::CB;
#pragma unroll 4
                    for (int st = 0; st < K::TM / 2; ++st) {
                        const int node = 2 * st + h;
                        const float a = ds[node * K::LD + ob * 32 + r31];
#pragma unroll
                        for (int b = 0; b < DPW; ++b) {
                            const float bv = xs[node * K::LD + (cb0 + b) * 32 + r31];
                            dacc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, dacc[b], 0, 0, 0);
                        }
                    }
                } else {
                    // one 32x32 block: waves split the TM nodes
                    constexpr int NPW = K::TM / 4;
#pragma unroll 4
                    for (int st = 0; st < NPW / 2; ++st) {
                        const int node = wave * NPW + 2 * st + h;
                        dacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[node * K::LD + r31], xs[node * K::LD + r31],
                                                                       dacc[0], 0, 0, 0);
                    }
                }
            } else {
                constexpr int C2 = C * C;
                constexpr int NSUB = (256 / C2) > 0 ? (256 / C2) : 1;       // node subsets
                const int el = tid % C2, sb = tid / C2;
                if (sb < NSUB) {
                    const int o = el / C, c = el % C;
                    for (int node = sb; node < K::TM; node += NSUB) dav = fmaf(ds[node * K::LD + o], xs[node * K::LD + c], dav);
                }
            }
            GADAPT_STAMP(p.stamps, tslot + 5);
            // ---- dxd = (base-dt) g + dP A
            if constexpr (!XC && !D4) {
            if constexpr (K::MFMA && !RESIDENT_B) gemm.load(p.A, nullptr);
#ifndef GADAPT_ABL_NO_GEMM
            if constexpr (PRE) {
                __syncthreads();                                // the pre-split copy is complete; every wave is done with dP in ds (dA)
                f32x16 acc[decltype(gemm)::BPW];
                gemm.accumulate_presplit(slab_ptr(t - dir), acc);
                gemm.store_presplit(ds, acc);
                __syncthreads();
            } else if constexpr (K::MFMA) {
                gemm.run_in_place(ds);                          // reads dP (like the dA pass), barrier, writes dP A
                __syncthreads();
            }
#endif
#if GADAPT_T_MFMA_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            GADAPT_STAMP(p.stamps, tslot + 6);
#pragma unroll
            for (int it = 0; it < K::ITERS; ++it) {
                const int li = it * K::SLOTS + slot;
                const int i = node0 + li;
                V r;
                if constexpr (K::MFMA) {
                    r = lds_vec<C>(ds, li, sub);
                } else {
                    r.zero();
#pragma unroll
                    for (int o = 0; o < C; ++o) {
                        const float d = ds[li * K::LD + o];
                        r.v[0].x = fmaf(d, acol[0][o], r.v[0].x); r.v[0].y = fmaf(d, acol[1][o], r.v[0].y);
                        r.v[0].z = fmaf(d, acol[2][o], r.v[0].z); r.v[0].w = fmaf(d, acol[3][o], r.v[0].w);
                    }
                }
                const float ri = PRE ? dinv[it] : 1.0f;         // pre-split operand: the row's inverse scale
#pragma unroll
                for (int q = 0; q < K::NV; ++q) {
                    gk[it].v[q].x = fmaf(r.v[q].x, ri, gk[it].v[q].x); gk[it].v[q].y = fmaf(r.v[q].y, ri, gk[it].v[q].y);
                    gk[it].v[q].z = fmaf(r.v[q].z, ri, gk[it].v[q].z); gk[it].v[q].w = fmaf(r.v[q].w, ri, gk[it].v[q].w);
                }
            }
            }
            __syncthreads();
            GADAPT_STAMP(p.stamps, tslot + 7);
        }
        if constexpr (!XC) store_dxd((tb + (n_my - 1) * dir) * K::TM);
    }
    if constexpr (!DA) return;                                  // the source pass owns the weight-gradient partials
    xs = ring;                                                  // scratch for the flush below
#ifdef GADAPT_ABL_NO_ACCUM
    p.accumulate = 0;                                           // diagnostic build: slab rows written, never read
#endif

    // ---- flush partials into this workgroup's slab row (deterministic: one owner per element)
    float* row = p.slab + (size_t)blockIdx.x * ROW;
    if constexpr (K::MFMA) {
        const int h = lane >> 5, r31 = lane & 31;
        if constexpr (NB2 >= K::NW) {
            const int ob = (wave * DPW) / K::CB, cb0 = (wave * DPW) % K::CB;
            // read-modify-write of the slab row in two phases: all loads of the old partials in flight at once, then the
            // stores (an interleaved load / add / store per element is one memory round trip per element at the kernel's tail:
            // measured 3.7 us of a 39 us launch at hidden 64, 28 of 128 us at hidden 128)
            float old[DPW][16];
#pragma unroll
            for (int b = 0; b < DPW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, c = (cb0 + b) * 32 + r31;
                    old[b][r] = p.accumulate ? __builtin_nontemporal_load(row + o * C + c) : 0.f;
                }
#pragma unroll
            for (int b = 0; b < DPW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, c = (cb0 + b) * 32 + r31;
                    row[o * C + c] = dacc[b][r] + old[b][r];
                }
        } else {
            float* red = xs;                                    // [4][32*32]
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r31] = dacc[0][r];
            __syncthreads();
            for (int e = tid; e < 1024; e += K::NT) {
                float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
                if (p.accumulate) v += row[e];
                row[e] = v;
            }
            __syncthreads();
        }
    } else {
        constexpr int C2 = C * C;
        constexpr int NSUB = (256 / C2) > 0 ? (256 / C2) : 1;
        float* red = xs;
        if (tid < C2 * NSUB) red[tid] = dav;
        __syncthreads();
        if (tid < C2) {
            float v = 0.f;
            for (int sb = 0; sb < NSUB; ++sb) v += red[sb * C2 + tid];
            if (p.accumulate) v += row[tid];
            row[tid] = v;
        }
        __syncthreads();
    }
    {   // dp0 and the two scalars: tree over the node slots.  Thread (slot, sub) holds channels 4*(sub+q*LPN)+comp.
        constexpr int W = 4 * K::NV;
        float* red = xs;                                        // [NT][W], spans into the dP tile for small C
#pragma unroll
        for (int q = 0; q < K::NV; ++q) {
            red[tid * W + 4 * q + 0] = dp0acc.v[q].x; red[tid * W + 4 * q + 1] = dp0acc.v[q].y;
            red[tid * W + 4 * q + 2] = dp0acc.v[q].z; red[tid * W + 4 * q + 3] = dp0acc.v[q].w;
        }
        __syncthreads();
        if (tid < C) {
            const int c4 = tid / 4, comp = tid % 4;
            const int sb = c4 % K::LPN, q = c4 / K::LPN;
            float v = 0.f;
            for (int s = 0; s < K::SLOTS; ++s) v += red[(s * K::LPN + sb) * W + 4 * q + comp];
            if (p.accumulate) v += row[C * C + tid];
            row[C * C + tid] = v;
        }
        if constexpr (SUMS != 0) {
            // the two scalars: butterfly inside each wave, then the wave sums in order (a serial walk over the NT threads'
            // values by two threads was 256 dependent LDS reads at the tail of every workgroup: +3.6 us per launch)
            float v0 = sum_ddt, v1 = sum_dsc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_xor(v0, off, 64); if (SUMS > 1) v1 += __shfl_xor(v1, off, 64); }
            __syncthreads();                                    // the dp0 tree has read red[]
            if (lane == 0) { red[2 * wave] = v0; red[2 * wave + 1] = v1; }
            __syncthreads();
            if (p.sums_out && tid < (SUMS > 1 ? 2 : 1)) {
                float v = 0.f;
                for (int w_ = 0; w_ < K::NW; ++w_) v += red[2 * w_ + tid];
                if (tid == 1) v = v / (sc * sc);                // d/d(score_scale) = sum d(score') <P,x> = (1/sc) sum d(score') log alpha; the terms carry one more sc
                float* dst = tid == 0 ? p.sums_out : p.sums_sc_out;
                if (p.sums_partials) dst[blockIdx.x] = v;       // one slot per workgroup, summed in fixed order by layer_params_reduce_kernel
                else atomicAdd(dst, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, target pass of a layer whose INPUT is the compact [N,4] encoder output (layer 0 behind the identity encoder,
// GNN.py:75-82: x0 = [features | 0]).  No source pass follows such a launch (d x0 is not wanted), so all it owes is the
// weight-gradient partial - and with x zero beyond column 3 that partial lives in a 4 x 4 corner:
//   d alpha_ik = dt <g_i, x_k>          touches the first four columns of g_i only,
//   dP_i = sum_k ds_ik x_k              is zero beyond column 3,
//   dA[o][c] += dP_i[o] x_i[c], dp0[o] += dP_i[o]     are non-zero for o, c < 4 only.
// So the launch reads 16 bytes of each g row, the [N,4] matrix and alpha, and adds 20 numbers per workgroup to the slab: one
// NODE PER LANE, no LDS tiles, no matrix cores.  (The generic kernel with the XC staging did the full-width edge walk, the dA
// phase and the flush of a 64 x 64 block for the same 20 numbers: 25.6 us per launch at hidden 64 on the metric workload.)
// Same arithmetic as grand_bwd_target_kernel, SUMS included; summation order: per lane over its nodes (grid-stride, ascending),
// wave butterfly, the four waves in order - fixed for a given grid, so runs stay bit-reproducible.
// ------------------------------------------------------------------------------------------------
template <int SUMS>
__global__ __launch_bounds__(256) void grand_bwd_target_compact_kernel(BwdTArgs p) {
    __shared__ float red[4][24];
    const int C = p.c;
    const int gs = p.g_stride;                                  // floats between g rows: C, or 4 when the layer above ran the D4 / source4 pair
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float dt = p.lp[0], sc = p.lp[1];
    const float scl = SUMS ? sc * dt : sc;
    float acc[22];                                              // dA[o][c] (16), dp0[o] (4), d dt, d score_scale
#pragma unroll
    for (int k = 0; k < 22; ++k) acc[k] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < p.n_nodes; i += (int64_t)gridDim.x * 256) {
        const float4 g4 = *reinterpret_cast<const float4*>(p.g_in + (size_t)i * gs);
        const float4 xi = *reinterpret_cast<const float4*>(p.x_in + 4 * (size_t)i);
        const float4 dm = SUMS ? g4 : make_float4(dt * g4.x, dt * g4.y, dt * g4.z, dt * g4.w);
        const int e0 = p.rowptr[i], e1 = p.rowptr[i + 1], deg = e1 - e0;
        float D = 0.f;
        float4 dP = f4zero();
        if (deg <= 8) {
            // the usual case, three memory round trips per node: (g row head, x row, row bounds) -> (8 column / alpha pairs,
            // unconditional with clamped indices) -> (8 neighbour rows); a loop over the row would chain them per edge
            const int last = max(p.n_edges - 1, 0);
            int cj[8]; float ak[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = min(e0 + k, last);
                cj[k] = p.col[e];
                ak[k] = p.alpha[e];
            }
            float4 xk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xk[k] = *reinterpret_cast<const float4*>(p.x_in + 4 * (size_t)cj[k]);
            float da[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                ak[k] = (k < deg) ? ak[k] : 0.f;
                da[k] = dot4(dm, xk[k]);
                D = fmaf(ak[k], da[k], D);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float ds = ak[k] * (da[k] - D) * scl;
                axpy4(dP, ds, xk[k]);
                if constexpr (SUMS > 1) { if (ak[k] > 0.f) acc[21] = fmaf(ds, __logf(ak[k]), acc[21]); }
            }
        } else {
            for (int e = e0; e < e1; ++e) {
                const float4 xk = *reinterpret_cast<const float4*>(p.x_in + 4 * (size_t)p.col[e]);
                D = fmaf(p.alpha[e], dot4(dm, xk), D);
            }
            for (int e = e0; e < e1; ++e) {
                const float4 xk = *reinterpret_cast<const float4*>(p.x_in + 4 * (size_t)p.col[e]);
                const float ak = p.alpha[e];
                const float ds = ak * (dot4(dm, xk) - D) * scl;
                axpy4(dP, ds, xk);
                if constexpr (SUMS > 1) { if (ak > 0.f) acc[21] = fmaf(ds, __logf(ak), acc[21]); }
            }
        }
        if constexpr (SUMS != 0) acc[20] += D - dot4(g4, xi);    // d dt = sum_i <g_i, m_i - x_i>, <g_i, m_i> = D (see the tiled kernel)
        const float dp[4] = {dP.x, dP.y, dP.z, dP.w}, xv[4] = {xi.x, xi.y, xi.z, xi.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[4 * o + c] = fmaf(dp[o], xv[c], acc[4 * o + c]);
            acc[16 + o] += dp[o];
        }
    }
#pragma unroll
    for (int k = 0; k < 22; ++k) {
        if (k >= 20 && (SUMS == 0 || (k == 21 && SUMS < 2))) continue;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
        if (lane == 0) red[wave][k] = acc[k];
    }
    __syncthreads();
    // this workgroup's slab row (layout of the tiled kernel: dA [C][C], then dp0 [C]); accumulate = 0: the row is written whole
    float* row = p.slab + (size_t)blockIdx.x * (C * C + C);
    if (!p.accumulate) {
        for (int e = tid; e < C * C + C; e += 256) {
            const int o = e / C, c = e % C;
            const bool live = (e < C * C) ? (o < 4 && c < 4) : (e - C * C < 4);
            if (!live) row[e] = 0.f;
        }
    }
    if (tid < 20) {
        const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        const int e = tid < 16 ? (tid >> 2) * C + (tid & 3) : C * C + (tid - 16);
        row[e] = p.accumulate ? row[e] + v : v;
    }
    if constexpr (SUMS != 0) {
        if (p.sums_out && tid >= 20 && tid < (SUMS > 1 ? 22 : 21)) {
            float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            if (tid == 21) v = v / (sc * sc);
            float* dst = tid == 20 ? p.sums_out : p.sums_sc_out;
            if (p.sums_partials) dst[blockIdx.x] = v; else atomicAdd(dst, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, source pass
// ------------------------------------------------------------------------------------------------
struct BwdSArgs {
    const float* x_in; const float* g_in; const float* edge_ws; const float* dxd;
    const float* A; const float* p0;
    const int32_t* rowptr; const int32_t* col; const int32_t* meta;
    float* g_out;
    int n_nodes, n_tiles, n_edges;
    unsigned long long* stamps;
    int g_cols;                                                 // GC kernels: g_in is [N,g_cols]
    float* slab; int accumulate;                                // DA kernels: this workgroup's slab row (layout of the target pass)
};

// Half of one node's out-edge rows: g_i and x_i of HN targets (two halves cover DM edges)
template <int HN_, int NV> struct SBuf {
    static constexpr int N = HN_;
    Vec<NV> g[HN_], x[HN_];
    float2 ev[HN_];
};

// One node's out-edge rows of g (windowed source pass: the x rows come from the LDS ring)
template <int DM_, int NV> struct SBufG {
    static constexpr int N = DM_;
    Vec<NV> g[DM_];
    float2 ev[DM_];
    int e0;
};

// DA: also accumulate the weight-gradient partials here instead of in the target pass:
//   dA[o][c] = sum_i dP_i[o] x_i[c] = sum_j x_j[o] y_j[c],   dp0[o] = sum_i dP_i[o] = sum_j sigma_j x_j[o]
// with y_j / sigma_j the per-source sums this pass forms for A y_j + sigma_j p0 anyway (exchange the two sums over the
// edges).  The y tile is in LDS for the projection; the node's own x rows join it in a third tile.
// WIN: rolling LDS window of x rows like the target pass (slabs t-1, t, t+1; a workgroup walks consecutive tiles): on
// mesh-ordered graphs the x_i gathers of the out-edge walk are LDS reads and only the g_i rows go through L1 / L2 - the walk is
// bound by the 64 bytes per clock of the CU's vector-memory path (512 B per edge), which this halves.  The projection then
// runs in place in the y tile (window + two tiles would leave one workgroup per CU).
template <int C, bool GC = false, bool DA = false, bool WIN = false>
__global__ __launch_bounds__(Cfg<C>::NT, (C >= GADAPT_ONE_WAVE_C ? 1 : GADAPT_WAVES_BWD_S)) void grand_bwd_source_kernel(BwdSArgs p) {
    using K = Cfg<C>;
    using V = Vec<K::NV>;
    static_assert(!DA || (K::MFMA && TileGemm<C, false>::SPLIT && K::NW == 4), "dA in the source pass: matrix-core sizes only");
    static_assert(!WIN || (!DA && K::MFMA), "windowed source pass: matrix-core sizes, weight gradients in the target pass");
    extern __shared__ float4 smem4[];
    float* ring = reinterpret_cast<float*>(smem4);              // WIN: three slabs of x rows
    float* ys = ring + (WIN ? 3 * K::TILE_FLOATS : 0);
    float* os = WIN ? ys : ys + K::TILE_FLOATS;
    float* xt = os + K::TILE_FLOATS;                            // DA: own x rows of the tile
    using CsrT = TileCsr<C, 2, WIN ? 2 : 0>;                    // aux = {alpha*dt, d<P,x>} per out-edge (source order); WIN: ext = ring offsets
    CsrT csr;
    csr.bind(os + (DA ? 2 : 1) * K::TILE_FLOATS, p.rowptr, p.col, p.edge_ws, p.meta, p.n_edges);
    constexpr int NB2 = K::CB * K::CB;
    constexpr int DPW = DA ? (NB2 >= 4 ? NB2 / 4 : 1) : 1;
    f32x16 dacc[DPW];
    V dp0acc; dp0acc.zero();
    float sigs[K::ITERS];
    if constexpr (DA) {
#pragma unroll
        for (int b = 0; b < DPW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[b][r] = 0.f;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid / K::LPN, sub = tid % K::LPN;

    // row i of the upstream gradient, this lane's chunks (GC: compact [N,g_cols] source)
    auto ld_g = [&](int i) __attribute__((always_inline)) {
        if constexpr (GC) {
            V r;
            r.v[0] = ld_row4_compact(p.g_in, i, sub, p.g_cols);  // chunk `sub` of the row: data in chunk 0 only, zeros elsewhere
#pragma unroll
            for (int q = 1; q < K::NV; ++q) r.v[q] = f4zero();   // chunks sub + q LPN > 0: no load
            return r;
        } else {
            return ld_vec<C>(p.g_in, i, sub);
        }
    };
    // PRE: the y tile is written in the pre-split f16 form by the lanes that sum its rows (lds_put_split); yinv = row scales
    constexpr bool PRE = K::MFMA && !DA && GADAPT_PRESPLIT_S && TileGemm<C, false>::SPLIT;
    float yinv[K::ITERS];
    TileGemm<C, false, GADAPT_SPLIT_F16_S || PRE> gemm;         // os = y A^T
    float arow[K::MFMA ? 1 : 4][K::MFMA ? 1 : C];
    if constexpr (K::MFMA) {
        gemm.init(lane, wave);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) arow[t][c] = p.A[(4 * sub + t) * C + c];
    }
    V p0v;
#pragma unroll
    for (int q = 0; q < K::NV; ++q) p0v.v[q] = *reinterpret_cast<const float4*>(p.p0 + 4 * (sub + q * K::LPN));

#ifndef GADAPT_S_RESIDENT_B
#define GADAPT_S_RESIDENT_B 1
#endif
    // split fragments: built once per launch - unless the dA accumulators need their 48 registers (DA: rebuilt per tile, in
    // flight under the barrier)
    constexpr bool RESIDENT_B = K::MFMA && TileGemm<C, false>::SPLIT && GADAPT_S_RESIDENT_B && !DA && C < GADAPT_BWD_JIT_B_C;
    // this workgroup's tiles: every step-th one (XCD-interleaved), or - WIN - a run of consecutive ones
    // (tb, stp, n_my): first tile, stride, count.  (WIN: walking every other workgroup's run backwards, as the target pass
    // does, was measured at hidden 128: 70.5 -> 72.1 us - the rows this pass shares between workgroups are gathered g rows, not
    // halo slabs.  GADAPT_S_ALTERNATE=1 builds it.)
    int tb, stp, n_my;
    if constexpr (WIN) {
        const TileChunk ch = tile_chunk(p.n_tiles);
        stp = (GADAPT_S_ALTERNATE && ((blockIdx.x >> 3) & 1)) ? -1 : 1;
        tb = stp > 0 ? ch.t0 : ch.t1 - 1;
        n_my = max(ch.t1 - ch.t0, 0);
    } else {
        const TileRange tr = tile_range(p.n_tiles);
        tb = tr.t; stp = tr.step;
        n_my = tr.t < tr.t_end ? (tr.t_end - tr.t + tr.step - 1) / tr.step : 0;
    }
    typename CsrT::Regs sr;
    auto slab_ptr = [&](int s_) __attribute__((always_inline)) { return ring + ((s_ + 3) % 3) * K::TILE_FLOATS; };
    TileRows<C> xr;
    if constexpr (WIN) {
        // prologue loads in one memory round trip (see the target pass): the window's first two slabs, metadata, fragments
        TileRows<C> xr2;
        xr.template issue<GADAPT_S_STREAM == 2>(p.x_in, (tb - stp) * K::TM, p.n_nodes, tid);
        xr2.template issue<GADAPT_S_STREAM == 2>(p.x_in, tb * K::TM, p.n_nodes, tid);
        const int4 mreg = csr.metas_issue(tb, stp, p.n_tiles, tid);
        if constexpr (RESIDENT_B) gemm.load(p.A, nullptr);
        csr.metas_commit(mreg, tid);
        xr.commit(slab_ptr(tb - stp), p.n_nodes, tid);
        xr2.commit(slab_ptr(tb), p.n_nodes, tid);
    } else {
        const int4 mreg = csr.metas_issue(tb, stp, p.n_tiles, tid);   // one round trip with the fragment loads
        if constexpr (RESIDENT_B) gemm.load(p.A, nullptr);
        csr.metas_commit(mreg, tid);
    }
    __syncthreads();
    int kt = 0;                                                 // index of the tile in this workgroup's sequence
    if (n_my > 0) {
        csr.issue(sr, 0, tb * K::TM, p.n_nodes, tid);
        if constexpr (WIN) xr.template issue<GADAPT_S_STREAM == 2>(p.x_in, (tb + stp) * K::TM, p.n_nodes, tid);
    }
    V zr[K::ITERS];                                             // sum(alpha dt g_i) + sigma p0, kept across the GEMM; then the result rows
    // g_out rows of a finished tile leave the registers only after the NEXT tile's staging (see the target pass)
    auto store_out = [&](int node0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < K::ITERS; ++it) {
            const int j = node0 + it * K::SLOTS + slot;
            if (j < p.n_nodes) {
                if constexpr (WIN && GADAPT_S_STREAM) {
#pragma unroll
                    for (int q = 0; q < K::NV; ++q) st_row4_nt<C>(p.g_out, j, sub + q * K::LPN, zr[it].v[q]);
                } else {
                    st_vec<C>(p.g_out, j, sub, zr[it]);
                }
            }
        }
    };
    int prev_node0 = -1;
    for (int k = 0; k < n_my; ++k) {
        const int t = tb + k * stp;
        const int node0 = t * K::TM;
        const int tslot = k * 8;
        GADAPT_STAMP(p.stamps, tslot + 0);
        const bool win = WIN && sr.meta.w != 0;
        if constexpr (WIN) xr.commit(slab_ptr(t + stp), p.n_nodes, tid);
        const int dmax = WIN ? csr.template commit<3>(sr, tid, node0, win ? t : -1) : csr.commit(sr, tid, node0);
        __syncthreads();
        if (prev_node0 >= 0) store_out(prev_node0);
        prev_node0 = node0;
        {   // request the next tile now (unconditional, clamped past the end: see issue()): it lands during this one
            csr.issue(sr, ++kt, (t + stp) * K::TM, p.n_nodes, tid);
        }
        if constexpr (DA) {                                     // own x rows of this tile -> xt (zeros past N); not held across the edge walk
            TileRows<C> xrows;
            xrows.issue(p.x_in, node0, p.n_nodes, tid);
            xrows.commit(xt, p.n_nodes, tid);
        }
        GADAPT_STAMP(p.stamps, tslot + 1);
        if (dmax >= 0) {
            dispatch_dmax(dmax, [&](auto tag) {
                constexpr int DM = decltype(tag)::value;
                constexpr int HN = (DM + 1) / 2;
                // pipeline steps = (node slot, half): rows of the next half are requested before this one is summed
                auto fetch_w = [&](SBuf<HN, K::NV>& b, int step, auto win_tag) __attribute__((always_inline)) {
                    const int it = step >> 1, half = step & 1;
                    const int li = it * K::SLOTS + slot;
                    const int e0 = csr.rp[li] - csr.ebase + half * HN;
                    const int deg = (node0 + li < p.n_nodes) ? csr.rp[li + 1] - csr.rp[li] : 0;
#pragma unroll
                    for (int k = 0; k < HN; ++k) {
                        const int i = csr.col[e0 + k];               // past the row end: some valid row, weight 0
                        b.ev[k] = *reinterpret_cast<const float2*>(csr.aux + 2 * (e0 + k));
                        if (half * HN + k >= deg) b.ev[k] = make_float2(0.f, 0.f);
                        b.g[k] = ld_g(i);
#ifdef GADAPT_ABL_S_NO_X
                        b.x[k] = b.g[k];
#else
                        if constexpr (decltype(win_tag)::value != 0) {
                            const float* row = ring + csr.ext[e0 + k];
#pragma unroll
                            for (int q = 0; q < K::NV; ++q) b.x[k].v[q] = *reinterpret_cast<const float4*>(row + 4 * (sub + q * K::LPN));
                        } else {
                            b.x[k] = ld_vec<C>(p.x_in, i, sub);
                        }
#endif
                    }
                };
                V z, y; float sig = 0.f;
                auto consume = [&](const SBuf<HN, K::NV>& b, int step) __attribute__((always_inline)) {
                    const int it = step >> 1, half = step & 1;
                    if (half == 0) { z.zero(); y.zero(); sig = 0.f; }
#pragma unroll
                    for (int k = 0; k < HN; ++k) { vaxpy(z, b.ev[k].x, b.g[k]); vaxpy(y, b.ev[k].y, b.x[k]); sig += b.ev[k].y; }
                    if (half == 1) {
                        const int li = it * K::SLOTS + slot;
                        if constexpr (PRE) yinv[it] = lds_put_split<C>(ys, li, sub, y); else lds_put<C>(ys, li, sub, y);
                        vaxpy(z, sig, p0v);
                        zr[it] = z;
                        sigs[it] = sig;
                    }
                };
                if constexpr (WIN) {
                    if (win) {
                        // windowed tile: the buffer holds g rows only (x rows are LDS reads at the point of use), so one request
                        // covers a whole node - half as many exposed round trips per tile as the two-matrix half-node steps
                        auto fetch_n = [&](SBufG<DM, K::NV>& b, int it) __attribute__((always_inline)) {
                            const int li = it * K::SLOTS + slot;
                            const int e0 = csr.rp[li] - csr.ebase;
                            const int deg = (node0 + li < p.n_nodes) ? csr.rp[li + 1] - csr.rp[li] : 0;
                            b.e0 = e0;
#pragma unroll
                            for (int k = 0; k < DM; ++k) {
                                const int i = csr.col[e0 + k];
                                b.ev[k] = *reinterpret_cast<const float2*>(csr.aux + 2 * (e0 + k));
                                if (k >= deg) b.ev[k] = make_float2(0.f, 0.f);
                                b.g[k] = ld_g(i);
                            }
                        };
                        auto consume_n = [&](const SBufG<DM, K::NV>& b, int it) __attribute__((always_inline)) {
                            V zz, yy; zz.zero(); yy.zero();
                            float sg = 0.f;
#pragma unroll
                            for (int k = 0; k < DM; ++k) {
                                const float* row = ring + csr.ext[b.e0 + k];
                                V xv;
#pragma unroll
                                for (int q = 0; q < K::NV; ++q) xv.v[q] = *reinterpret_cast<const float4*>(row + 4 * (sub + q * K::LPN));
                                vaxpy(zz, b.ev[k].x, b.g[k]); vaxpy(yy, b.ev[k].y, xv); sg += b.ev[k].y;
                            }
                            const int li = it * K::SLOTS + slot;
                            if constexpr (PRE) yinv[it] = lds_put_split<C>(ys, li, sub, yy); else lds_put<C>(ys, li, sub, yy);
                            vaxpy(zz, sg, p0v);
                            zr[it] = zz;
                            sigs[it] = sg;
                        };
                        run_pipeline<K::ITERS, SBufG<DM, K::NV>>(fetch_n, consume_n, [&]() {});
                    } else {
                        run_pipeline<2 * K::ITERS, SBuf<HN, K::NV>>([&](auto& b, int step) { fetch_w(b, step, IntTag<0>{}); }, consume, [&]() {});
                    }
                } else {
                    run_pipeline<2 * K::ITERS, SBuf<HN, K::NV>>([&](auto& b, int step) { fetch_w(b, step, IntTag<0>{}); }, consume, [&]() {});
                }
            });
        } else {
#pragma unroll
            for (int it = 0; it < K::ITERS; ++it) {              // any row length, CSR straight from HBM
                const int li = it * K::SLOTS + slot;
                const int j = node0 + li;
                V z, y; z.zero(); y.zero();
                float sig = 0.f;
                if (j < p.n_nodes) {
                    const int e0 = p.rowptr[j], deg = p.rowptr[j + 1] - e0;
                    for (int k = 0; k < deg; ++k) {
                        const int i = p.col[e0 + k];
                        const float2 ev = *reinterpret_cast<const float2*>(p.edge_ws + 2 * (size_t)(e0 + k));
                        vaxpy(z, ev.x, ld_g(i));
                        vaxpy(y, ev.y, ld_vec<C>(p.x_in, i, sub));
                        sig += ev.y;
                    }
                }
                if constexpr (PRE) yinv[it] = lds_put_split<C>(ys, li, sub, y); else lds_put<C>(ys, li, sub, y);
                vaxpy(z, sig, p0v);
                zr[it] = z;
                sigs[it] = sig;
            }
        }
        GADAPT_STAMP(p.stamps, tslot + 2);
        if constexpr (K::MFMA && !RESIDENT_B) gemm.load(p.A, nullptr);   // in flight under the barrier
        if constexpr (WIN) xr.template issue<GADAPT_S_STREAM == 2>(p.x_in, (t + 2 * stp) * K::TM, p.n_nodes, tid);   // slab of the next tile's window (clamped past the end)
        V dpre[K::ITERS];                                       // own dxd rows: requested here, used after the GEMM
#pragma unroll
        for (int it = 0; it < K::ITERS; ++it) {
            const int jr = min(node0 + it * K::SLOTS + slot, p.n_nodes - 1);
            if constexpr ((WIN && GADAPT_S_STREAM) || GADAPT_S_STREAM_DXD) {
#pragma unroll
                for (int q = 0; q < K::NV; ++q) dpre[it].v[q] = ld_row4_nt<C>(p.dxd, jr, sub + q * K::LPN);
            } else {
                dpre[it] = ld_vec<C>(p.dxd, jr, sub);
            }
        }
        __syncthreads();
        GADAPT_STAMP(p.stamps, tslot + 3);
#ifndef GADAPT_ABL_S_NO_GEMM
        if constexpr (K::MFMA) {
            if constexpr (PRE) {
                f32x16 acc[decltype(gemm)::BPW];
                gemm.accumulate_presplit(ys, acc);
                if constexpr (WIN) __syncthreads();             // in place: every wave has read its operand rows
                gemm.store_presplit(os, acc);
            } else if constexpr (WIN) {
                gemm.run_in_place(ys);
            } else {
                gemm.run(ys, os);
            }
            if constexpr (DA) {
#pragma unroll
                for (int it = 0; it < K::ITERS; ++it)           // dp0 += sigma_j x_j
                    vaxpy(dp0acc, sigs[it], lds_vec<C>(xt, it * K::SLOTS + slot, sub));
            }
            __syncthreads();
        }
#endif
        if constexpr (DA) {
            // dA[o][c] += sum_node x[node][o] y[node][c]   (three-piece bf16 split, k = node: see the target pass)
            const int h = lane >> 5, r31 = lane & 31;
            constexpr int NODES = (NB2 >= 4) ? K::TM : K::TM / 4;
            const int nbase = (NB2 >= 4) ? 0 : wave * NODES;
            const int ob = (NB2 >= 4) ? (wave * DPW) / K::CB : 0, cb0 = (NB2 >= 4) ? (wave * DPW) % K::CB : 0;
#pragma unroll GADAPT_DA_UNROLL
            for (int ks = 0; ks < NODES / 16; ++ks) {
                const int n0 = nbase + 16 * ks + 8 * h;
                float av[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) av[e] = xt[(n0 + e) * K::LD + ob * 32 + r31];
                const Split3 as = split8(av);
#pragma unroll
                for (int b = 0; b < DPW; ++b) {
                    float bv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) bv[e] = ys[(n0 + e) * K::LD + (cb0 + b) * 32 + r31];
                    const Split3 bs = split8(bv);
                    dacc[b] = mfma_bf16(as.h, bs.l, dacc[b]);
                    dacc[b] = mfma_bf16(as.l, bs.h, dacc[b]);
                    dacc[b] = mfma_bf16(as.m, bs.m, dacc[b]);
                    dacc[b] = mfma_bf16(as.h, bs.m, dacc[b]);
                    dacc[b] = mfma_bf16(as.m, bs.h, dacc[b]);
                    dacc[b] = mfma_bf16(as.h, bs.h, dacc[b]);
                }
            }
        }
        GADAPT_STAMP(p.stamps, tslot + 4);
#pragma unroll
        for (int it = 0; it < K::ITERS; ++it) {
            const int li = it * K::SLOTS + slot;
            const int j = node0 + li;
            if (j >= p.n_nodes) continue;
            V r;
            if constexpr (K::MFMA) {
                r = lds_vec<C>(os, li, sub);
            } else {
                r.zero();
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float yc = ys[li * K::LD + c];
                    r.v[0].x = fmaf(arow[0][c], yc, r.v[0].x); r.v[0].y = fmaf(arow[1][c], yc, r.v[0].y);
                    r.v[0].z = fmaf(arow[2][c], yc, r.v[0].z); r.v[0].w = fmaf(arow[3][c], yc, r.v[0].w);
                }
            }
            const V d = dpre[it];
            const float ri = PRE ? yinv[it] : 1.0f;             // pre-split operand: the row's inverse scale (1: exact either way)
#pragma unroll
            for (int q = 0; q < K::NV; ++q) {
                zr[it].v[q].x = fmaf(r.v[q].x, ri, zr[it].v[q].x) + d.v[q].x; zr[it].v[q].y = fmaf(r.v[q].y, ri, zr[it].v[q].y) + d.v[q].y;
                zr[it].v[q].z = fmaf(r.v[q].z, ri, zr[it].v[q].z) + d.v[q].z; zr[it].v[q].w = fmaf(r.v[q].w, ri, zr[it].v[q].w) + d.v[q].w;
            }
        }
        GADAPT_STAMP(p.stamps, tslot + 5);
        __syncthreads();
        GADAPT_STAMP(p.stamps, tslot + 6);
    }
    if (prev_node0 >= 0) store_out(prev_node0);
    if constexpr (DA) {
        // ---- flush the partials into this workgroup's slab row (layout and order of the target pass's flush)
        constexpr int ROW = C * C + C;
        float* row = p.slab + (size_t)blockIdx.x * ROW;
        const int h = lane >> 5, r31 = lane & 31;
        __syncthreads();
        if constexpr (NB2 >= 4) {
            const int ob = (wave * DPW) / K::CB, cb0 = (wave * DPW) % K::CB;
#pragma unroll
            for (int b = 0; b < DPW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, c = (cb0 + b) * 32 + r31;
                    float v = dacc[b][r];
                    if (p.accumulate) v += row[o * C + c];
                    row[o * C + c] = v;
                }
        } else {
            float* red = ys;                                    // [4][32*32]
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r31] = dacc[0][r];
            __syncthreads();
            for (int e = tid; e < 1024; e += K::NT) {
                float v = (red[e] + red[1024 + e]) + (red[2048 + e] + red[3072 + e]);
                if (p.accumulate) v += row[e];
                row[e] = v;
            }
            __syncthreads();
        }
        constexpr int W = 4 * K::NV;
        float* red = ys;                                        // [NT][W]
#pragma unroll
        for (int q = 0; q < K::NV; ++q) {
            red[tid * W + 4 * q + 0] = dp0acc.v[q].x; red[tid * W + 4 * q + 1] = dp0acc.v[q].y;
            red[tid * W + 4 * q + 2] = dp0acc.v[q].z; red[tid * W + 4 * q + 3] = dp0acc.v[q].w;
        }
        __syncthreads();
        if (tid < C) {
            const int c4 = tid / 4, comp = tid % 4;
            const int sb = c4 % K::LPN, q = c4 / K::LPN;
            float v = 0.f;
            for (int s = 0; s < K::SLOTS; ++s) v += red[(s * K::LPN + sb) * W + 4 * q + comp];
            if (p.accumulate) v += row[C * C + tid];
            row[C * C + tid] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// C is a compile-time trip count so that every load of a dot product is in flight at once (a runtime-length loop
// pays one L2 round trip per unrolled group).
template <int C>
__device__ __forceinline__ void coeffs_fwd_body(const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wk,
                                                float* __restrict__ a, float* __restrict__ p0, int e) {
    if (e < C * C) {                    // A[o][cc] = sum_r wk[r][o] wq[r][cc]
        const int o = e / C, cc = e % C;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < C; ++r) v[r & 3] = fmaf(wk[r * C + o], wq[r * C + cc], v[r & 3]);
        a[e] = (v[0] + v[1]) + (v[2] + v[3]);
    } else if (e < C * C + C) {
        const int o = e - C * C;
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < C; ++r) v = fmaf(wk[r * C + o], bq[r], v);
        p0[o] = v;
    }
}
template <int C>
__global__ void coeffs_fwd_kernel(const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wk,
                                  float* __restrict__ a, float* __restrict__ p0) {
    coeffs_fwd_body<C>(wq, bq, wk, a, p0, blockIdx.x * blockDim.x + threadIdx.x);
}

template <int C>
__global__ void coeffs_bwd_kernel(const float* __restrict__ wq, const float* __restrict__ bq, const float* __restrict__ wk,
                                  const float* __restrict__ d_a, const float* __restrict__ d_p0,
                                  float* __restrict__ d_wq, float* __restrict__ d_bq, float* __restrict__ d_wk, float* __restrict__ d_bk) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int c2 = C * C;
    if (e < c2) {                       // d_wq[r][cc] = sum_o wk[r][o] dA[o][cc]
        const int r = e / C, cc = e % C;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < C; ++o) v[o & 3] = fmaf(wk[r * C + o], d_a[o * C + cc], v[o & 3]);
        d_wq[e] = (v[0] + v[1]) + (v[2] + v[3]);
    } else if (e < 2 * c2) {            // d_wk[r][o] = sum_cc wq[r][cc] dA[o][cc] + bq[r] dp0[o]
        const int f = e - c2, r = f / C, o = f % C;
        float v[4] = {bq[r] * d_p0[o], 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < C; ++cc) v[cc & 3] = fmaf(wq[r * C + cc], d_a[o * C + cc], v[cc & 3]);
        d_wk[f] = (v[0] + v[1]) + (v[2] + v[3]);
    } else if (e < 2 * c2 + C) {        // d_bq[r] = sum_o wk[r][o] dp0[o]
        const int r = e - 2 * c2;
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < C; ++o) v = fmaf(wk[r * C + o], d_p0[o], v);
        d_bq[r] = v;
    } else if (e < 2 * c2 + 2 * C) {
        d_bk[e - 2 * c2 - C] = 0.f;
    }
}

// x0[i][4q..4q+3] = sum_k feats[i][k] w[4q+t][k] (+ b).  W^T sits in LDS as float4 per (k, q); C/4 consecutive
// threads write one 4*C-byte row, and a thread keeps its q while it strides over nodes.
#define GADAPT_ENC_MAX_WORDS 4096      /* C * F floats of LDS */
struct EncArgs {
    const float* feats; int f0; const float* e1; const float* e2; const float* w; const float* b; float* x0;
    int64_t n_nodes; int f, c;
};
// block `bid` of `nblocks` encoder blocks (the merged encoder + coefficient launch runs more blocks than that)
__device__ __forceinline__ void encode_body(const EncArgs& p, float4* wl, int bid, int nblocks) {
    const float* __restrict__ feats = p.feats; const float* __restrict__ e1 = p.e1; const float* __restrict__ e2 = p.e2;
    const float* __restrict__ w = p.w; const float* __restrict__ b = p.b; float* __restrict__ x0 = p.x0;
    const int f0 = p.f0, f = p.f, c = p.c; const int64_t n_nodes = p.n_nodes;
    const int c4 = c >> 2;
    for (int idx = threadIdx.x; idx < f * c4; idx += blockDim.x) {
        const int k = idx / c4, q = idx % c4;
        wl[idx] = make_float4(w[(4 * q + 0) * f + k], w[(4 * q + 1) * f + k], w[(4 * q + 2) * f + k], w[(4 * q + 3) * f + k]);
    }
    __syncthreads();
    const int q = threadIdx.x % c4;
    const int rows_per_block = blockDim.x / c4;
    const float4 bias = b ? *reinterpret_cast<const float4*>(b + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    // features of node i: the f0 columns of feats [N,f0], then the per-node scalars e1, e2 (GNN.py:227-239 concat).
    // Four rows per thread and iteration, every load of the four issued before the first use (the kernel is a pure
    // HBM write stream: what limits it is how many stores a wave keeps in flight).
    constexpr int R = 4;
    const int64_t stride = (int64_t)nblocks * rows_per_block;
    for (int64_t i0 = (int64_t)bid * rows_per_block + threadIdx.x / c4; i0 < n_nodes; i0 += R * stride) {
        float xv[R][4], x1[R], x2[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = min(i0 + r * stride, n_nodes - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[r][k] = feats[i * f0 + min(k, f0 - 1)];
            x1[r] = e1 ? e1[i] : 0.f;
            x2[r] = e2 ? e2[i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r * stride;
            if (i >= n_nodes) break;
            float4 v = bias;
            if (f0 <= 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < f0) {
                        const float4 wv = wl[k * c4 + q];
                        v.x = fmaf(xv[r][k], wv.x, v.x); v.y = fmaf(xv[r][k], wv.y, v.y); v.z = fmaf(xv[r][k], wv.z, v.z); v.w = fmaf(xv[r][k], wv.w, v.w);
                    }
                }
            } else {
                for (int k = 0; k < f0; ++k) {
                    const float xk = feats[i * f0 + k];
                    const float4 wv = wl[k * c4 + q];
                    v.x = fmaf(xk, wv.x, v.x); v.y = fmaf(xk, wv.y, v.y); v.z = fmaf(xk, wv.z, v.z); v.w = fmaf(xk, wv.w, v.w);
                }
            }
            if (e1) {
                const float4 wv = wl[f0 * c4 + q];
                v.x = fmaf(x1[r], wv.x, v.x); v.y = fmaf(x1[r], wv.y, v.y); v.z = fmaf(x1[r], wv.z, v.z); v.w = fmaf(x1[r], wv.w, v.w);
            }
            if (e2) {
                const float4 wv = wl[(f - 1) * c4 + q];
                v.x = fmaf(x2[r], wv.x, v.x); v.y = fmaf(x2[r], wv.y, v.y); v.z = fmaf(x2[r], wv.z, v.z); v.w = fmaf(x2[r], wv.w, v.w);
            }
            *reinterpret_cast<float4*>(x0 + i * c + 4 * q) = v;
        }
    }
}
__global__ __launch_bounds__(256) void encode_linear_kernel(EncArgs p) {
    __shared__ float4 wl[GADAPT_ENC_MAX_WORDS / 4];
    encode_body(p, wl, blockIdx.x, gridDim.x);
}
// The encoder and the composite coefficients (A, p0) of the shared conv are independent and both precede layer 0: one
// launch, the first `enc_blocks` workgroups encode, the rest compute coefficients (a dependent dispatch costs ~4.5 us).
template <int C>
__global__ __launch_bounds__(256) void encode_coeffs_kernel(EncArgs p, int enc_blocks, const float* __restrict__ wq, const float* __restrict__ bq,
                                                            const float* __restrict__ wk, float* __restrict__ a, float* __restrict__ p0) {
    __shared__ float4 wl[GADAPT_ENC_MAX_WORDS / 4];
    if ((int)blockIdx.x < enc_blocks) { encode_body(p, wl, blockIdx.x, enc_blocks); return; }
    coeffs_fwd_body<C>(wq, bq, wk, a, p0, ((int)blockIdx.x - enc_blocks) * 256 + threadIdx.x);
}

// g_top[i][:] = {g_phys[i][0..d), 0, ...}: backward of the x[:, :dim] slice (GNN.py:299) in one pass
__global__ void pad_columns_kernel(const float* __restrict__ g_phys, float* __restrict__ g_top, int64_t n_nodes, int d, int c) {
    const int c4 = c >> 2;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_nodes * c4) return;
    const int64_t i = e / c4;
    const int o = (int)(e % c4) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o < d) {
        v.x = g_phys[i * d + o];
        if (o + 1 < d) v.y = g_phys[i * d + o + 1];
        if (o + 2 < d) v.z = g_phys[i * d + o + 2];
        if (o + 3 < d) v.w = g_phys[i * d + o + 3];
    }
    *reinterpret_cast<float4*>(g_top + i * c + o) = v;
}

// slab [n_rows][row_len] -> part [CHUNKS][row_len]
// d dt_l / d score_scale_l: the per-workgroup partials of the SUMS target-pass launches (gadapt_block_backward) summed in a
// fixed order - one workgroup per (kind, layer): four interleaved partial sums per thread, wave butterfly, the four waves in
// order.  Replaces float atomics on one address per layer (512 of them per launch: +5 us on the compact-input launch) and
// makes the step / temperature gradients bit-reproducible like everything else.
__device__ __forceinline__ void layer_params_reduce_block(const float* __restrict__ partials, int n_rows, int n_layers, int want_scale,
                                                          float* __restrict__ out, int slot /* kind * L + l */) {
    __shared__ float red[4];
    float v = 0.f;
    if (slot < n_layers || want_scale) {
        const float* src = partials + (size_t)slot * n_rows;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        for (int r = threadIdx.x; r < n_rows; r += 1024) {
            v0 += src[r];
            if (r + 256 < n_rows) v1 += src[r + 256];
            if (r + 512 < n_rows) v2 += src[r + 512];
            if (r + 768 < n_rows) v3 += src[r + 768];
        }
        v = (v0 + v1) + (v2 + v3);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[slot] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void layer_params_reduce_kernel(const float* __restrict__ partials, int n_rows, int n_layers, int want_scale,
                                                                  float* __restrict__ out) {
    layer_params_reduce_block(partials, n_rows, n_layers, want_scale, out, blockIdx.x);
}
// Workgroups past the row's column blocks (blockIdx.x >= nbx, first chunk row only) sum the d dt / d score_scale partials of the
// same backward instead (lp_*: optional; see layer_params_reduce_block) - they ride in this launch rather than in one of their own.
__global__ void slab_reduce1_kernel(const float* slab, float* part, int n_rows, int row_len, int nbx = 1 << 30,
                                    const float* lp_partials = nullptr, int n_layers = 0, int want_scale = 0, float* lp_out = nullptr) {
    if ((int)blockIdx.x >= nbx) {
        if (blockIdx.y == 0 && lp_partials) layer_params_reduce_block(lp_partials, n_rows, n_layers, want_scale, lp_out, (int)blockIdx.x - nbx);
        return;
    }
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= row_len) return;
    const int per = (n_rows + GADAPT_SLAB_CHUNKS - 1) / GADAPT_SLAB_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(n_rows, r0 + per);
    // 32 rows in flight per thread, four partial sums (fixed order): the kernel is a handful of memory round trips long
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int r = r0;
    for (; r + 32 <= r1; r += 32) {
        float t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) t[k] = slab[(size_t)(r + k) * row_len + e];
#pragma unroll
        for (int k = 0; k < 32; k += 4) { v0 += t[k]; v1 += t[k + 1]; v2 += t[k + 2]; v3 += t[k + 3]; }
    }
    for (; r < r1; ++r) v0 += slab[(size_t)r * row_len + e];
    part[(size_t)blockIdx.y * row_len + e] = (v0 + v1) + (v2 + v3);
}
__global__ void slab_reduce2_kernel(const float* part, float* d_a, float* d_p0, int c) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int row_len = c * c + c;
    if (e >= row_len) return;
    float v = 0.f;
    for (int k = 0; k < GADAPT_SLAB_CHUNKS; ++k) v += part[(size_t)k * row_len + e];
    if (e < c * c) d_a[e] = v; else d_p0[e - c * c] = v;
}

// part [CHUNKS][C*C+C] -> (dA, dp0) in LDS -> dWq | dbq | dWk | dbk: the second level of the slab reduction and the chain
// rule of coeffs_bwd_kernel in ONE launch.  Every workgroup adds the CHUNKS partial rows itself (fixed order, so all
// workgroups hold bit-identical sums; 0.5 MB of L2-resident reads per workgroup at C = 64) and then produces its share
// of the 2 C^2 + 2 C outputs.  dA rows are padded by one float: the dWk sum walks a column of dA^T.
template <int C>
__global__ __launch_bounds__(1024) void reduce2_coeffs_bwd_kernel(const float* __restrict__ part, const float* __restrict__ wq,
                                                                  const float* __restrict__ bq, const float* __restrict__ wk,
                                                                  float* __restrict__ d_wq, float* __restrict__ d_bq,
                                                                  float* __restrict__ d_wk, float* __restrict__ d_bk) {
    extern __shared__ float4 smem4[];
    float* da = reinterpret_cast<float*>(smem4);                 // [C][C+1]
    float* dp = da + C * (C + 1);                                // [C]
    constexpr int ROW = C * C + C, c2 = C * C;
    // 1024 threads, the CHUNKS loads of an element all in flight: the sums are an L2 round trip per element and thread,
    // not per load (a 256-thread version with an 8-wide unroll took 26.8 us at C = 64: 64 dependent round trips)
    for (int e = threadIdx.x; e < ROW; e += 1024) {
        float t[GADAPT_SLAB_CHUNKS];
#pragma unroll
        for (int k = 0; k < GADAPT_SLAB_CHUNKS; ++k) t[k] = part[(size_t)k * ROW + e];
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < GADAPT_SLAB_CHUNKS; ++k) v += t[k];  // same order as slab_reduce2_kernel
        if (e < c2) da[(e / C) * (C + 1) + (e % C)] = v; else dp[e - c2] = v;
    }
    __syncthreads();
    for (int e = blockIdx.x * 1024 + threadIdx.x; e < 2 * c2 + 2 * C; e += gridDim.x * 1024) {
        if (e < c2) {                       // d_wq[r][cc] = sum_o wk[r][o] dA[o][cc]
            const int r = e / C, cc = e % C;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int o = 0; o < C; ++o) v[o & 3] = fmaf(wk[r * C + o], da[o * (C + 1) + cc], v[o & 3]);
            d_wq[e] = (v[0] + v[1]) + (v[2] + v[3]);
        } else if (e < 2 * c2) {            // d_wk[r][o] = sum_cc wq[r][cc] dA[o][cc] + bq[r] dp0[o]
            const int f = e - c2, r = f / C, o = f % C;
            float v[4] = {bq[r] * dp[o], 0.f, 0.f, 0.f};
#pragma unroll
            for (int cc = 0; cc < C; ++cc) v[cc & 3] = fmaf(wq[r * C + cc], da[o * (C + 1) + cc], v[cc & 3]);
            d_wk[f] = (v[0] + v[1]) + (v[2] + v[3]);
        } else if (e < 2 * c2 + C) {        // d_bq[r] = sum_o wk[r][o] dp0[o]
            const int r = e - 2 * c2;
            float v = 0.f;
#pragma unroll
            for (int o = 0; o < C; ++o) v = fmaf(wk[r * C + o], dp[o], v);
            d_bq[r] = v;
        } else {
            d_bk[e - 2 * c2 - C] = 0.f;     // softmax shift invariance: d/d lin_key.bias vanishes identically
        }
    }
}

__global__ void mesh_loss_seed_kernel(const float* x_top, const float* target, float* x_phys, float* g_top, float* loss_out,
                                      int64_t n_nodes, int d, int c, int l1, float gscale) {
    __shared__ float red[256];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float lv = 0.f;
    if (e < n_nodes * c) {
        const int64_t i = e / c;
        const int k = (int)(e % c);
        float g = 0.f;
        if (k < d) {
            const float xv = x_top[e];
            const float diff = xv - target[i * d + k];
            x_phys[i * d + k] = xv;
            const float inv = 1.0f / (float)(n_nodes * d);
            if (l1) { lv = fabsf(diff) * inv; g = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * inv * gscale; }
            else    { lv = diff * diff * inv; g = 2.0f * diff * inv * gscale; }
        }
        g_top[e] = g;
    }
    red[threadIdx.x] = lv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0 && red[0] != 0.f) atomicAdd(loss_out, red[0]);
}

// loss = mean((pred - target)^2) or mean(|pred - target|) and seed = d loss / d pred in one launch.  Deterministic:
// per-block partials, then the block that takes the last ticket adds them in index order.  scratch[0] is the ticket
// counter (left at zero), scratch[1..] the partials.
#ifndef GADAPT_LOSS_BLOCKS
#define GADAPT_LOSS_BLOCKS 128      /* every block costs one ticket atomic: 64 / 128 / 256 / 512 blocks measured 10.1 / 8.9 / 10.5 / 15.3 us */
#endif
__global__ __launch_bounds__(256) void loss_forward_kernel(const float* __restrict__ pred, int64_t pred_stride,
                                                           const float* __restrict__ target, int64_t n_rows, int d, int l1,
                                                           float* __restrict__ seed, float* __restrict__ loss_out, float* scratch) {
    __shared__ float red[4];
    __shared__ unsigned ticket;
    const int64_t total = n_rows * d;
    const float inv = 1.0f / (float)total;
    float lv = 0.f;
    // one row per thread and slot, four slots per iteration with every load issued before the first use (rows of `pred`
    // are pred_stride floats apart: each read is its own cache line, so the kernel lives on loads in flight)
    constexpr int R = 4;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n_rows; i0 += R * stride) {
        float pv[R][4], tv[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = min(i0 + r * stride, n_rows - 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < d) { pv[r][k] = pred[i * pred_stride + k]; tv[r][k] = target[i * d + k]; }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r * stride;
            if (i >= n_rows) break;
            if (d <= 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < d) {
                        const float diff = pv[r][k] - tv[r][k];
                        if (l1) { lv += fabsf(diff); seed[i * d + k] = (diff > 0.f ? inv : (diff < 0.f ? -inv : 0.f)); }
                        else    { lv = fmaf(diff, diff, lv); seed[i * d + k] = 2.0f * diff * inv; }
                    }
                }
            } else {
                for (int k = 0; k < d; ++k) {
                    const float diff = pred[i * pred_stride + k] - target[i * d + k];
                    if (l1) { lv += fabsf(diff); seed[i * d + k] = (diff > 0.f ? inv : (diff < 0.f ? -inv : 0.f)); }
                    else    { lv = fmaf(diff, diff, lv); seed[i * d + k] = 2.0f * diff * inv; }
                }
            }
        }
    }
    // fixed-shape tree: butterfly inside each wave, then the four wave sums in order (one barrier instead of eight)
    auto block_sum = [&](float v) __attribute__((always_inline)) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();                                         // red[] of an earlier call has been read
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float bsum = block_sum(lv);
    if (threadIdx.x == 0) {
        __hip_atomic_store(scratch + 1 + blockIdx.x, bsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        ticket = atomicAdd(reinterpret_cast<unsigned*>(scratch), 1u);
    }
    __syncthreads();
    if (ticket != gridDim.x - 1) return;
    __threadfence();
    float v = 0.f;
    for (unsigned k = threadIdx.x; k < gridDim.x; k += 256)      // fixed order per thread, then the fixed tree
        v += __hip_atomic_load(scratch + 1 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float total_sum = block_sum(v);
    if (threadIdx.x == 0) {
        loss_out[0] = total_sum * inv;
        *reinterpret_cast<unsigned*>(scratch) = 0u;            // ready for the next launch
    }
}

__global__ void adam_step_kernel(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float b1, float b2,
                                 float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float g = grad[e] * gscale;
    const float pv = param[e];
    if (wd != 0.f) g = fmaf(wd, pv, g);
    const float mn = fmaf(b1, m[e], (1.f - b1) * g);
    const float vn = fmaf(b2, v[e], (1.f - b2) * g * g);
    m[e] = mn; v[e] = vn;
    const float denom = sqrtf(vn) / bc2_sqrt + eps;
    param[e] = pv - (lr / bc1) * (mn / denom);
}

#include "gadapt_wide.inc"
#include "gadapt_fused_bwd.inc"

// Same update with the step count kept on the device (state[0] = steps taken, state[1] = exit ticket), so that the
// launch carries no host-side value that changes from step to step and can sit inside a captured hipGraph.  Every
// workgroup reads the count when it starts; the workgroup that exits last (all others have read by then) advances it.
__global__ __launch_bounds__(256) void adam_step_dev_kernel(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float b1,
                                                            float b2, float eps, float wd, int* state, float gscale) {
    __shared__ int s_step;
    if (threadIdx.x == 0) s_step = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __syncthreads();
    const float stepf = (float)s_step;
    const float bc1 = 1.0f - powf(b1, stepf), bc2_sqrt = sqrtf(1.0f - powf(b2, stepf));
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) {
        float g = grad[e] * gscale;
        const float pv = param[e];
        if (wd != 0.f) g = fmaf(wd, pv, g);
        const float mn = fmaf(b1, m[e], (1.f - b1) * g);
        const float vn = fmaf(b2, v[e], (1.f - b2) * g * g);
        m[e] = mn; v[e] = vn;
        const float denom = sqrtf(vn) / bc2_sqrt + eps;
        param[e] = pv - (lr / bc1) * (mn / denom);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(state + 1, 1) == (int)gridDim.x - 1) {
            __hip_atomic_store(state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(state, s_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

// Workgroups of a launch: a multiple of 8 (XCD groups), at most max_blocks (the resident set) unless that would
// give a workgroup more than 64 tiles (Cfg::MAXM: its tile metadata must fit the LDS table).
// ------------------------------------------------------------------------------------------------
// backward, source pass when only columns 0..3 of g_out are wanted (the layer below reads the compact [N,4] encoder output: see
// grand_bwd_target_compact_kernel; the target pass before this launch was the D4 variant and left dxd as [N,4]):
//   g_out[j][c] = dxd[j][c] + sum_i alpha_ij dt g_i[c] + sum_o A[c][o] y_j[o] + sigma_j p0[c],   c = 0..3,
// with y_j = sum_i ds_ij x_i over the out-edges as in grand_bwd_source_kernel.  Of the g rows only the first 16 bytes are
// gathered, A y is four dot products per node on the vector ALU (no y tile, no matrix cores, no barrier between edge walk and
// result), and N C floats of dxd reads and g_out writes become N * 4 each.  GC: the upstream gradient is compact [N,g_cols].
// ------------------------------------------------------------------------------------------------
template <int HN_, int NV> struct SBuf4 {
    static constexpr int N = HN_;
    float4 g4[HN_];
    Vec<NV> x[HN_];
    float2 ev[HN_];
    float4 d4;
};
#ifndef GADAPT_WAVES_BWD_S4
#define GADAPT_WAVES_BWD_S4 3      /* hidden <= 64: 157..166 registers, three workgroups per CU (the launch uses a grid of 768) */
#endif
template <int C, bool GC = false>
__global__ __launch_bounds__(Cfg<C>::NT, (C > 64 ? 2 : GADAPT_WAVES_BWD_S4)) void grand_bwd_source4_kernel(BwdSArgs p) {
    using K = Cfg<C>;
    using V = Vec<K::NV>;
    extern __shared__ float4 smem4[];
    using CsrT = TileCsr<C, 2, 0>;                              // aux = {alpha*dt, d<P,x>} per out-edge (source order)
    CsrT csr;
    csr.bind(reinterpret_cast<float*>(smem4), p.rowptr, p.col, p.edge_ws, p.meta, p.n_edges);
    const int tid = threadIdx.x;
    const int slot = tid / K::LPN, sub = tid % K::LPN;
    float a4[K::NV][4][4];                                      // A[c][o], o = this lane's channels, c = 0..3
#pragma unroll
    for (int q = 0; q < K::NV; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 r = *reinterpret_cast<const float4*>(p.A + (size_t)c * C + 4 * (sub + q * K::LPN));
            a4[q][0][c] = r.x; a4[q][1][c] = r.y; a4[q][2][c] = r.z; a4[q][3][c] = r.w;
        }
    const float4 p04 = *reinterpret_cast<const float4*>(p.p0);
    auto ld_g4 = [&](int i) __attribute__((always_inline)) {   // columns 0..3 of row i of the upstream gradient (same address in the group)
        if constexpr (GC) return ld_row4_compact(p.g_in, i, 0, p.g_cols);
        else return *reinterpret_cast<const float4*>(p.g_in + (size_t)i * C);
    };
    // result of one node: the group's sum of the four dot products, lane sub == 0 writes
    auto finish = [&](int j, const float4& d4, const float4& z4, const V& y, float sig) __attribute__((always_inline)) {
        float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < K::NV; ++q) {
            const float yv[4] = {y.v[q].x, y.v[q].y, y.v[q].z, y.v[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) t4[c] = fmaf(yv[e], a4[q][e][c], t4[c]);
        }
        group_sum_n<K::LPN>(t4);
        if (sub == 0 && j < p.n_nodes) {
            const float4 o = make_float4(d4.x + z4.x + t4[0] + sig * p04.x, d4.y + z4.y + t4[1] + sig * p04.y,
                                         d4.z + z4.z + t4[2] + sig * p04.z, d4.w + z4.w + t4[3] + sig * p04.w);
            *reinterpret_cast<float4*>(p.g_out + 4 * (size_t)j) = o;
        }
    };
    const TileRange tr = tile_range(p.n_tiles);
    const int tb = tr.t, stp = tr.step;
    const int n_my = tr.t < tr.t_end ? (tr.t_end - tr.t + tr.step - 1) / tr.step : 0;
    typename CsrT::Regs sr;
    csr.load_metas(tb, stp, p.n_tiles, tid);
    __syncthreads();
    if (n_my > 0) csr.issue(sr, 0, tb * K::TM, p.n_nodes, tid);
    for (int k = 0; k < n_my; ++k) {
        const int t = tb + k * stp;
        const int node0 = t * K::TM;
        const int dmax = csr.commit(sr, tid, node0);
        __syncthreads();
        csr.issue(sr, k + 1, (t + stp) * K::TM, p.n_nodes, tid);   // unconditional, clamped past the end: lands during this tile
        if (dmax >= 0) {
            dispatch_dmax(dmax, [&](auto tag) {
                constexpr int DM = decltype(tag)::value;
                constexpr int HN = (DM + 1) / 2;
                auto fetch = [&](SBuf4<HN, K::NV>& b, int step) __attribute__((always_inline)) {
                    const int it = step >> 1, half = step & 1;
                    const int li = it * K::SLOTS + slot;
                    const int e0 = csr.rp[li] - csr.ebase + half * HN;
                    const int deg = (node0 + li < p.n_nodes) ? csr.rp[li + 1] - csr.rp[li] : 0;
#pragma unroll
                    for (int kk = 0; kk < HN; ++kk) {
                        const int i = csr.col[e0 + kk];              // past the row end: some valid row, weight 0
                        b.ev[kk] = *reinterpret_cast<const float2*>(csr.aux + 2 * (e0 + kk));
                        if (half * HN + kk >= deg) b.ev[kk] = make_float2(0.f, 0.f);
                        b.g4[kk] = ld_g4(i);
                        b.x[kk] = ld_vec<C>(p.x_in, i, sub);
                    }
                    b.d4 = *reinterpret_cast<const float4*>(p.dxd + 4 * (size_t)min(node0 + li, p.n_nodes - 1));
                };
                float4 z4; V y; float sig = 0.f;
                auto consume = [&](const SBuf4<HN, K::NV>& b, int step) __attribute__((always_inline)) {
                    const int it = step >> 1, half = step & 1;
                    if (half == 0) { z4 = f4zero(); y.zero(); sig = 0.f; }
#pragma unroll
                    for (int kk = 0; kk < HN; ++kk) { axpy4(z4, b.ev[kk].x, b.g4[kk]); vaxpy(y, b.ev[kk].y, b.x[kk]); sig += b.ev[kk].y; }
                    if (half == 1) finish(node0 + it * K::SLOTS + slot, b.d4, z4, y, sig);
                };
                run_pipeline<2 * K::ITERS, SBuf4<HN, K::NV>>(fetch, consume, [&]() {});
            });
        } else {
#pragma unroll 1
            for (int it = 0; it < K::ITERS; ++it) {              // any row length, CSR straight from HBM
                const int j = node0 + it * K::SLOTS + slot;
                float4 z4 = f4zero(); V y; y.zero();
                float sig = 0.f;
                float4 d4 = f4zero();
                if (j < p.n_nodes) {
                    d4 = *reinterpret_cast<const float4*>(p.dxd + 4 * (size_t)j);
                    const int e0 = p.rowptr[j], deg = p.rowptr[j + 1] - e0;
                    for (int kk = 0; kk < deg; ++kk) {
                        const int i = p.col[e0 + kk];
                        const float2 ev = *reinterpret_cast<const float2*>(p.edge_ws + 2 * (size_t)(e0 + kk));
                        axpy4(z4, ev.x, ld_g4(i));
                        vaxpy(y, ev.y, ld_vec<C>(p.x_in, i, sub));
                        sig += ev.y;
                    }
                }
                finish(j, d4, z4, y, sig);
            }
        }
        __syncthreads();                                        // every wave is done with the CSR slice of this tile
    }
}

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
// The node pass + fused kernel of gadapt_fused_bwd.inc instead of the target / source pair for dense layers.  OFF by
// default: measured on MI355X (64x64 b32 C64) the fused kernel takes 79.8 us and the node pass 16.4 us against 42.1 + 29.9 us for
// the pair.  GADAPT_FUSED_BWD=1 in the environment or gadapt_debug_set_fused_backward(1) turn it on
// (tests/test_gpu_ops.py::test_fused_backward_matches_two_pass keeps it correct).
static std::atomic<int> g_fused_bwd{-1};
static bool fused_bwd_enabled() {
    int v = g_fused_bwd.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("GADAPT_FUSED_BWD");
        v = (e && e[0] == '1') ? 1 : 0;
        g_fused_bwd.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}
extern "C" int gadapt_debug_set_fused_backward(int on) { g_fused_bwd.store(on ? 1 : 0, std::memory_order_relaxed); return GADAPT_OK; }
#ifndef GADAPT_BWD_D_MAX_BLOCKS
#define GADAPT_BWD_D_MAX_BLOCKS 1024
#endif
#ifndef GADAPT_BWD_OUT4
#define GADAPT_BWD_OUT4 1            /* 0: layer 1 above a compact layer 0 runs the dense pair (A/B) */
#endif
#ifndef GADAPT_XC_COMPACT_KERNEL
#define GADAPT_XC_COMPACT_KERNEL 1      /* 0: the tiled target kernel with the XC staging for the compact layer input (A/B) */
#endif
template <int C> static int launch_bwd(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha,
                                       const float* a, const float* p0, const float* lp, float* edge_ws, float* dxd, float* slab,
                                       int accumulate, float* sums_out, float* g_out, int residual_only, int g_cols, int x_cols, hipStream_t st, float* sums_sc_out,
                                       int out4 = 0, int g_stride = 0, int sums_partials = 0) {
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
    if (out4 && (!g_out || x_cols || residual_only || C < 8 || (g_cols && sums_out)))
        return fail(GADAPT_E_BADARG, "4-column backward: a layer with a gradient to pass on, hidden >= 8, not compact-g with d dt / d scale");
#ifdef GADAPT_STAMPS
    pt.stamps = g_stamp_buf ? g_stamp_buf + 1024 * 32 : nullptr;
#endif
    constexpr int lds_t = K::lds_bytes(1, K::RING_T + 1, 1), lds_s = K::lds_bytes(2);
    int rc;
    if constexpr (C == 32 || C == 64) {
        // dense layer with a gradient to pass on: node pass for D + ONE fused kernel (gadapt_fused_bwd.inc) instead of the
        // target / source pair; D [N] lives at the start of the (otherwise unused) dxd workspace
        if (fused_bwd_enabled() && !g_cols && !x_cols && !sums_out && !out4 && g_out && g->rowptr_s && g->col_s && g->perm_s) {
            {
                BwdDArgs pd{x_in, g_in, alpha, lp, g->rowptr_t, g->col_t, meta_for<K::TM>(g->meta_t), dxd, g->n_nodes, n_tiles, residual_only, g->n_edges};
                ProfScope prof(5, st, 0);
                constexpr int lds_d = K::lds_bytes(1, 0);
                allow_lds(grand_bwd_dnode_kernel<C>, lds_d);
                hipLaunchKernelGGL(grand_bwd_dnode_kernel<C>, dim3(grid_for(n_tiles, GADAPT_BWD_D_MAX_BLOCKS)), dim3(256), lds_d, st, pd);
                if ((rc = check_launch("grand_bwd_dnode_kernel"))) return rc;
            }
            BwdFArgs pf{x_in, g_in, alpha, dxd, a, p0, lp, g->rowptr_t, g->col_t, meta_for<K::TM>(g->meta_t),
                        g->rowptr_s, g->col_s, g->perm_s, meta_for<K::TM>(g->meta_s), g_out, slab,
                        g->n_nodes, n_tiles, accumulate, residual_only, g->n_edges};
            ProfScope prof(6, st, 0);
            constexpr int lds_f = fused_bwd_lds_bytes<C>();
            allow_lds(grand_bwd_fused_kernel<C>, lds_f);
            hipLaunchKernelGGL(grand_bwd_fused_kernel<C>, dim3(grid_for(n_tiles, resident_blocks_bwd_t<C>(GADAPT_BWD_T_MAX_BLOCKS))), dim3(256), lds_f, st, pf);
            return check_launch("grand_bwd_fused_kernel");
        }
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
                               int out4, int g_stride) {
    GADAPT_DISPATCH_C(c, launch_bwd<CC>(g, x_in, g_in, alpha, a, p0, layer_params, edge_ws, dxd_ws, slab, accumulate, sums_out, g_out, 0, g_cols, x_cols, st, sums_sc_out,
                                        out4, g_stride, 1));
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
        const bool pair4 = GADAPT_BWD_OUT4 && x0_cols && c >= 8 && n_layers >= 2 && !(n_layers == 2 && g_top_cols > 0 && d_layer_params);
        const int out4 = (pair4 && l == 1) ? 1 : 0;
        const int g_stride = (pair4 && l == 0) ? 4 : 0;
        int rc = layer_backward_cols(g, x_all + l * nc, g_cur, alpha_all + (size_t)l * g->n_edges, a + l * a_stride, p0 + l * p0_stride,
                                     layer_params + 2 * l, edge_ws, dxd_ws, slab_l, accumulate, d_dt, d_sc, g_next, g_cols, x_cols, c, st,
                                     out4, g_stride);
        if (rc) return rc;
        g_cur = g_next;
    }
    return GADAPT_OK;
}

#include "gadapt_sparse.inc"
