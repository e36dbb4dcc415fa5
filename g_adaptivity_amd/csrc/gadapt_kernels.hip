// gadapt_kernels.hip - fused GRAND attention-diffusion layer for gfx950 (MI355X).
//
// Three kernels carry the hot path:
//   grand_fwd_kernel<C>        x' = x + dt (sum_j alpha_ij x_j - x)            (target-centric)
//   grand_bwd_target_kernel<C> d(score), dP, weight-gradient partials, dxd     (target-centric)
//   grand_bwd_source_kernel<C> g_out = dxd + sum over out-edges                (source-centric)
// All three share one shape: a 256-thread workgroup owns a tile of TM consecutive nodes,
// C/4 lanes cover one node (float4 per lane => a gathered neighbour row is one coalesced
// 4*C-byte read), the [TM,C]x[C,C] projection runs on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32, B operand resident in registers) for C >= 32 and on the VALU
// for C < 32, and tiles are dealt to workgroups so that one XCD walks a contiguous node
// range (neighbour rows are shared through that XCD's L2).
//
// Arithmetic follows /root/reference/src/GRAND_plus.py:225-343 and src/GNN.py:273-291 in the
// (A, p0) formulation described in include/gadapt_hip.h.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "gadapt_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GADAPT_MAXD 8           // in/out degree handled from registers; larger rows take the loop path
#define GADAPT_SLAB_CHUNKS 32   // second-level partials of the slab reduction

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
extern "C" int gadapt_abi_version(void) { return 1; }
extern "C" int gadapt_supported_hidden_dim(int c) {
    return c == 4 || c == 8 || c == 16 || c == 32 || c == 64 || c == 128;
}

// ------------------------------------------------------------------------------------------------
// optional per-kernel timing (bench/roofline only): HIP events on the launch stream around every
// hot-kernel launch.  Off by default; when off the launch path touches none of this.
// ------------------------------------------------------------------------------------------------
#include <vector>
struct ProfRec { int id; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
struct ProfScope {
    hipStream_t st; int idx = -1;
    ProfScope(int id, hipStream_t s) : st(s) {
        if (!g_prof_on) return;
        ProfRec r{id, nullptr, nullptr};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        (void)hipEventRecord(r.a, st);
        g_prof.push_back(r);
        idx = (int)g_prof.size() - 1;
    }
    ~ProfScope() { if (idx >= 0) (void)hipEventRecord(g_prof[idx].b, st); }
};
extern "C" int gadapt_profile_enable(int on) { g_prof_on = (on != 0); return GADAPT_OK; }
extern "C" int gadapt_profile_read(int kernel_id, double* total_ms, int* count) {
    if (!total_ms || !count) return fail(GADAPT_E_BADARG, "profile_read: null pointer");
    double tot = 0.0; int n = 0;
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
    for (auto& r : g_prof) {
        if (r.id != kernel_id) continue;
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && n < cap) out_ms[n++] = ms;
    }
    return n;
}
extern "C" int gadapt_profile_reset(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    return GADAPT_OK;
}

// ------------------------------------------------------------------------------------------------
// compile-time geometry
// ------------------------------------------------------------------------------------------------
template <int C> struct Cfg {
    static constexpr int LPN = C / 4;                  // lanes per node (float4 each)
    static constexpr int SLOTS = 256 / LPN;            // nodes in flight per workgroup
    static constexpr bool MFMA = (C >= 32);
    static constexpr int TM = MFMA ? (C == 32 ? 128 : 64) : (SLOTS < 64 ? 64 : SLOTS);
    static constexpr int ITERS = TM / SLOTS;
    static constexpr int LD = C + 4;                   // padded LDS row (floats): conflict-free b128 rows
    static constexpr int TILE_FLOATS = TM * LD;
    static constexpr int CB = C / 32;                  // 32-wide column blocks (MFMA path)
    static constexpr int RB = TM / 32;                 // 32-high row blocks
    static constexpr int LDS_BYTES = 2 * TILE_FLOATS * 4;
};

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ void axpy4(float4& y, float a, const float4& x) {
    y.x = fmaf(a, x.x, y.x); y.y = fmaf(a, x.y, y.y); y.z = fmaf(a, x.z, y.z); y.w = fmaf(a, x.w, y.w);
}
template <int LPN> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = 1; m < LPN; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
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

// ------------------------------------------------------------------------------------------------
// [TM,C] x [C,C] on the matrix cores.  D[n][j] = sum_k IN[n][k] * B[k][j] (+ bias[j]).
//   TRANS = false: B[k][j] = M[j*C + k]   (D = IN M^T : forward P = x A^T, source pass A y)
//   TRANS = true : B[k][j] = M[k*C + j]   (D = IN M   : target pass dP A)
// v_mfma_f32_32x32x2_f32: lane l feeds A[l&31][k=l>>5] and B[k=l>>5][l&31]; the k index is
// permuted so each lane-half reads 16 contiguous bytes of its IN row per 4 MFMAs (half h owns
// k in {8q+4h .. 8q+4h+3}); any permutation is valid as long as A and B agree.
// ------------------------------------------------------------------------------------------------
template <int C, bool TRANS> struct TileGemm {
    using K = Cfg<C>;
    static constexpr int BPW = (K::CB * K::RB) / 4;    // 32x32 output blocks per wave
    float bf[C / 2];
    float bias;
    int cb, rb0, lane;

    __device__ __forceinline__ void load(const float* __restrict__ M, const float* __restrict__ bias_vec,
                                         int lane_, int wave) {
        lane = lane_;
        cb = wave % K::CB;
        rb0 = wave / K::CB;
        const int h = lane >> 5, j = cb * 32 + (lane & 31);
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
        bias = bias_vec ? bias_vec[j] : 0.f;
    }

    // in_tile/out_tile: LDS [TM][LD].  Caller synchronises around it.
    __device__ __forceinline__ void run(const float* in_tile, float* out_tile) const {
        const int h = lane >> 5, r31 = lane & 31;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const int rb = rb0 + b * (4 / K::CB);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* arow = in_tile + (rb * 32 + r31) * K::LD + 4 * h;
#pragma unroll
            for (int q = 0; q < C / 8; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(arow + 8 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bf[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bf[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bf[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bf[4 * q + 3], acc, 0, 0, 0);
            }
            float* ocol = out_tile + cb * 32 + r31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                ocol[row * K::LD] = acc[r] + bias;
            }
        }
    }
};

// Stage TM rows of a [N,C] matrix into a padded LDS tile (zero rows past N).
template <int C> __device__ __forceinline__ void stage_tile(const float* __restrict__ src, float* tile,
                                                            int node0, int n_nodes, int tid) {
    using K = Cfg<C>;
    constexpr int V = C / 4;
    const float4* src4 = reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int idx = tid; idx < K::TM * V; idx += 256) {
        const int r = idx / V, c4 = idx % V;
        const int node = node0 + r;
        float4 v = f4zero();
        if (node < n_nodes) v = src4[(size_t)node * V + c4];
        *reinterpret_cast<float4*>(tile + r * K::LD + 4 * c4) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct FwdArgs {
    const float* x_in; float* x_out;
    const float* A; const float* p0; const float* lp;
    const int32_t* rowptr; const int32_t* col;
    float* alpha_out;
    int n_nodes, n_tiles, residual_only;
};

template <int C>
__global__ __launch_bounds__(256) void grand_fwd_kernel(FwdArgs p) {
    using K = Cfg<C>;
    extern __shared__ float4 smem4[];
    float* xs = reinterpret_cast<float*>(smem4);
    float* ps = xs + K::TILE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid / K::LPN, sub = tid % K::LPN;
    const float dt = p.lp[0], sc = p.lp[1];
    const float4* xin4 = reinterpret_cast<const float4*>(p.x_in);
    float4* xout4 = reinterpret_cast<float4*>(p.x_out);
    constexpr int V = C / 4;

    TileGemm<C, false> gemm;
    float arow[K::MFMA ? 1 : 4][K::MFMA ? 1 : C];
    float4 p0v = f4zero();
    if constexpr (K::MFMA) {
        gemm.load(p.A, p.p0, lane, wave);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) arow[t][c] = p.A[(4 * sub + t) * C + c];
        p0v = *reinterpret_cast<const float4*>(p.p0 + 4 * sub);
    }

    const TileRange tr = tile_range(p.n_tiles);
    for (int t = tr.t; t < tr.t_end; t += tr.step) {
        const int node0 = t * K::TM;
        stage_tile<C>(p.x_in, xs, node0, p.n_nodes, tid);
        __syncthreads();
        if constexpr (K::MFMA) {
            gemm.run(xs, ps);
            __syncthreads();
        }
#pragma unroll 1
        for (int it = 0; it < K::ITERS; ++it) {
            const int li = it * K::SLOTS + slot;
            const int i = node0 + li;
            if (i >= p.n_nodes) continue;
            float4 Pi;
            if constexpr (K::MFMA) {
                Pi = *reinterpret_cast<const float4*>(ps + li * K::LD + 4 * sub);
            } else {
                Pi = p0v;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float xc = xs[li * K::LD + c];
                    Pi.x = fmaf(arow[0][c], xc, Pi.x); Pi.y = fmaf(arow[1][c], xc, Pi.y);
                    Pi.z = fmaf(arow[2][c], xc, Pi.z); Pi.w = fmaf(arow[3][c], xc, Pi.w);
                }
            }
            const int e0 = p.rowptr[i], deg = p.rowptr[i + 1] - e0;
            float4 m = f4zero();
            if (deg <= GADAPT_MAXD) {
                float4 xj[GADAPT_MAXD];
                float s[GADAPT_MAXD];
#pragma unroll
                for (int k = 0; k < GADAPT_MAXD; ++k) {
                    xj[k] = f4zero();
                    if (k < deg) xj[k] = xin4[(size_t)p.col[e0 + k] * V + sub];
                }
                float mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < GADAPT_MAXD; ++k) {
                    s[k] = group_sum<K::LPN>(dot4(Pi, xj[k])) * sc;
                    if (k < deg) mx = fmaxf(mx, s[k]);
                }
                float den = 0.f;
#pragma unroll
                for (int k = 0; k < GADAPT_MAXD; ++k) {
                    s[k] = (k < deg) ? __expf(s[k] - mx) : 0.f;
                    den += s[k];
                }
                const float inv = 1.0f / (den + 1e-16f);       // PyG softmax epsilon
#pragma unroll
                for (int k = 0; k < GADAPT_MAXD; ++k) {
                    const float a = s[k] * inv;
                    axpy4(m, a, xj[k]);
                    if (p.alpha_out && k < deg && (k % K::LPN) == sub) p.alpha_out[e0 + k] = a;
                }
            } else {
                float mx = -INFINITY;
                for (int k = 0; k < deg; ++k) {
                    const float4 v = xin4[(size_t)p.col[e0 + k] * V + sub];
                    mx = fmaxf(mx, group_sum<K::LPN>(dot4(Pi, v)) * sc);
                }
                float den = 0.f;
                for (int k = 0; k < deg; ++k) {
                    const float4 v = xin4[(size_t)p.col[e0 + k] * V + sub];
                    den += __expf(group_sum<K::LPN>(dot4(Pi, v)) * sc - mx);
                }
                const float inv = 1.0f / (den + 1e-16f);
                for (int k = 0; k < deg; ++k) {
                    const float4 v = xin4[(size_t)p.col[e0 + k] * V + sub];
                    const float a = __expf(group_sum<K::LPN>(dot4(Pi, v)) * sc - mx) * inv;
                    axpy4(m, a, v);
                    if (p.alpha_out && (k % K::LPN) == sub) p.alpha_out[e0 + k] = a;
                }
            }
            const float4 xi = *reinterpret_cast<const float4*>(xs + li * K::LD + 4 * sub);
            float4 o;                                            // res = m - x (GRAND_plus.py:267); x + dt*res (GNN.py:291)
            o.x = m.x - xi.x; o.y = m.y - xi.y; o.z = m.z - xi.z; o.w = m.w - xi.w;
            if (!p.residual_only) {
                o.x = fmaf(dt, o.x, xi.x); o.y = fmaf(dt, o.y, xi.y); o.z = fmaf(dt, o.z, xi.z); o.w = fmaf(dt, o.w, xi.w);
            }
            xout4[(size_t)i * V + sub] = o;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// backward, target pass
// ------------------------------------------------------------------------------------------------
struct BwdTArgs {
    const float* x_in; const float* g_in; const float* alpha;
    const float* A; const float* lp;
    const int32_t* rowptr; const int32_t* col;
    float2* edge_ws; float* dxd; float* slab; float* sums_out;
    int n_nodes, n_tiles, accumulate, residual_only;
};

template <int C>
__global__ __launch_bounds__(256) void grand_bwd_target_kernel(BwdTArgs p) {
    using K = Cfg<C>;
    extern __shared__ float4 smem4[];
    float* xs = reinterpret_cast<float*>(smem4);
    float* ds = xs + K::TILE_FLOATS;                            // dP tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid / K::LPN, sub = tid % K::LPN;
    // out = base*x + dt*(m - x): Euler step (base 1) or bare residual (base 0, dt 1)
    const float dt = p.residual_only ? 1.0f : p.lp[0], sc = p.lp[1];
    const float w1 = (p.residual_only ? 0.0f : 1.0f) - dt;
    const float4* xin4 = reinterpret_cast<const float4*>(p.x_in);
    const float4* gin4 = reinterpret_cast<const float4*>(p.g_in);
    float4* dxd4 = reinterpret_cast<float4*>(p.dxd);
    constexpr int V = C / 4;
    constexpr int ROW = C * C + C;                              // slab row: dA then dp0

    // ---- per-kernel resident state
    TileGemm<C, true> gemm;                                     // dxd = dP A
    float acol[K::MFMA ? 1 : 4][K::MFMA ? 1 : C];               // VALU: A[o][4sub+t]
    if constexpr (K::MFMA) {
        gemm.load(p.A, nullptr, lane, wave);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < C; ++o) acol[t][o] = p.A[o * C + 4 * sub + t];
    }
    // dA partial accumulators.  MFMA: NB2 = CB*CB 32x32 blocks of dA; >= 4 blocks -> NB2/4 per wave over
    // all TM nodes, 1 block -> the 4 waves split the node range.  VALU: thread owns element tid % C^2 for
    // node subset tid / C^2.
    constexpr int NB2 = K::CB * K::CB;
    constexpr int DPW = K::MFMA ? (NB2 >= 4 ? NB2 / 4 : 1) : 1;
    f32x16 dacc[DPW];
    float dav = 0.f;
    if constexpr (K::MFMA) {
#pragma unroll
        for (int b = 0; b < DPW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[b][r] = 0.f;
    }
    float4 dp0acc = f4zero();
    float sum_ddt = 0.f, sum_dsc = 0.f;

    const TileRange tr = tile_range(p.n_tiles);
    for (int t = tr.t; t < tr.t_end; t += tr.step) {
        const int node0 = t * K::TM;
        stage_tile<C>(p.x_in, xs, node0, p.n_nodes, tid);
        // ---- edge phase: dP_i per node -> LDS
#pragma unroll 1
        for (int it = 0; it < K::ITERS; ++it) {
            const int li = it * K::SLOTS + slot;
            const int i = node0 + li;
            float4 dP = f4zero();
            if (i < p.n_nodes) {
                const float4 gi = gin4[(size_t)i * V + sub];
                float4 dm; dm.x = dt * gi.x; dm.y = dt * gi.y; dm.z = dt * gi.z; dm.w = dt * gi.w;
                const int e0 = p.rowptr[i], deg = p.rowptr[i + 1] - e0;
                float4 m = f4zero();
                if (deg <= GADAPT_MAXD) {
                    float4 xj[GADAPT_MAXD];
                    float a[GADAPT_MAXD], da[GADAPT_MAXD];
#pragma unroll
                    for (int k = 0; k < GADAPT_MAXD; ++k) {
                        xj[k] = f4zero(); a[k] = 0.f;
                        if (k < deg) { xj[k] = xin4[(size_t)p.col[e0 + k] * V + sub]; a[k] = p.alpha[e0 + k]; }
                    }
                    float D = 0.f;
#pragma unroll
                    for (int k = 0; k < GADAPT_MAXD; ++k) {
                        da[k] = group_sum<K::LPN>(dot4(dm, xj[k]));
                        D = fmaf(a[k], da[k], D);
                    }
#pragma unroll
                    for (int k = 0; k < GADAPT_MAXD; ++k) {
                        const float dsp = a[k] * (da[k] - D);           // d(score'), score' = sc * <P_i,x_j>
                        const float dss = dsp * sc;
                        axpy4(dP, dss, xj[k]);
                        axpy4(m, a[k], xj[k]);
                        if (a[k] > 0.f) sum_dsc = fmaf(dsp, __logf(a[k]), sum_dsc);
                        if (k < deg && (k % K::LPN) == sub) p.edge_ws[e0 + k] = make_float2(a[k] * dt, dss);
                    }
                } else {
                    float D = 0.f;
                    for (int k = 0; k < deg; ++k) {
                        const float4 v = xin4[(size_t)p.col[e0 + k] * V + sub];
                        D = fmaf(p.alpha[e0 + k], group_sum<K::LPN>(dot4(dm, v)), D);
                    }
                    for (int k = 0; k < deg; ++k) {
                        const float4 v = xin4[(size_t)p.col[e0 + k] * V + sub];
                        const float ak = p.alpha[e0 + k];
                        const float dsp = ak * (group_sum<K::LPN>(dot4(dm, v)) - D);
                        const float dss = dsp * sc;
                        axpy4(dP, dss, v);
                        axpy4(m, ak, v);
                        if (ak > 0.f) sum_dsc = fmaf(dsp, __logf(ak), sum_dsc);
                        if ((k % K::LPN) == sub) p.edge_ws[e0 + k] = make_float2(ak * dt, dss);
                    }
                }
                // d dt = sum_i <g_i, m_i - x_i>   (GNN.py:288-289 learn_step); every lane adds its 4 channels
                const float4 xi = xin4[(size_t)i * V + sub];
                sum_ddt += gi.x * (m.x - xi.x) + gi.y * (m.y - xi.y) + gi.z * (m.z - xi.z) + gi.w * (m.w - xi.w);
                dp0acc.x += dP.x; dp0acc.y += dP.y; dp0acc.z += dP.z; dp0acc.w += dP.w;
            }
            *reinterpret_cast<float4*>(ds + li * K::LD + 4 * sub) = dP;
        }
        __syncthreads();
        // ---- dA partial:  dA[o][c] += sum_node dP[node][o] x[node][c]
        if constexpr (K::MFMA) {
            const int h = lane >> 5, r31 = lane & 31;
            if constexpr (NB2 >= 4) {
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
        __syncthreads();
        // ---- dxd = (base-dt) g + dP A
        if constexpr (K::MFMA) {
            gemm.run(ds, xs);                                   // xs is dead after the dA pass
            __syncthreads();
        }
#pragma unroll 1
        for (int it = 0; it < K::ITERS; ++it) {
            const int li = it * K::SLOTS + slot;
            const int i = node0 + li;
            if (i >= p.n_nodes) continue;
            float4 r;
            if constexpr (K::MFMA) {
                r = *reinterpret_cast<const float4*>(xs + li * K::LD + 4 * sub);
            } else {
                r = f4zero();
#pragma unroll
                for (int o = 0; o < C; ++o) {
                    const float d = ds[li * K::LD + o];
                    r.x = fmaf(d, acol[0][o], r.x); r.y = fmaf(d, acol[1][o], r.y);
                    r.z = fmaf(d, acol[2][o], r.z); r.w = fmaf(d, acol[3][o], r.w);
                }
            }
            const float4 gi = gin4[(size_t)i * V + sub];
            r.x = fmaf(w1, gi.x, r.x); r.y = fmaf(w1, gi.y, r.y); r.z = fmaf(w1, gi.z, r.z); r.w = fmaf(w1, gi.w, r.w);
            dxd4[(size_t)i * V + sub] = r;
        }
        __syncthreads();
    }

    // ---- flush partials into this workgroup's slab row (deterministic: one owner per element)
    float* row = p.slab + (size_t)blockIdx.x * ROW;
    if constexpr (K::MFMA) {
        const int h = lane >> 5, r31 = lane & 31;
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
            float* red = xs;                                    // [4][32*32]
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + r31] = dacc[0][r];
            __syncthreads();
            for (int e = tid; e < 1024; e += 256) {
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
    {   // dp0 and the two scalars: tree over the node slots
        float* red = xs;                                        // [256][6], spans into the dP tile for small C
        red[tid * 6 + 0] = dp0acc.x; red[tid * 6 + 1] = dp0acc.y; red[tid * 6 + 2] = dp0acc.z; red[tid * 6 + 3] = dp0acc.w;
        red[tid * 6 + 4] = sum_ddt;
        red[tid * 6 + 5] = (sub == 0) ? sum_dsc : 0.f;          // d(score) sums are group-uniform: count once
        __syncthreads();
        if (tid < C) {
            const int sb = tid / 4, comp = tid % 4;
            float v = 0.f;
            for (int s = 0; s < K::SLOTS; ++s) v += red[(s * K::LPN + sb) * 6 + comp];
            if (p.accumulate) v += row[C * C + tid];
            row[C * C + tid] = v;
        }
        if (p.sums_out && tid < 2) {
            float v = 0.f;
            for (int s = 0; s < 256; ++s) v += red[s * 6 + 4 + tid];
            if (tid == 1) v = v / sc;                           // d/d(score_scale) = sum d(score') * <P,x>
            atomicAdd(p.sums_out + tid, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, source pass
// ------------------------------------------------------------------------------------------------
struct BwdSArgs {
    const float* x_in; const float* g_in; const float2* edge_ws; const float* dxd;
    const float* A; const float* p0;
    const int32_t* rowptr; const int32_t* col; const int32_t* perm;
    float* g_out;
    int n_nodes, n_tiles;
};

template <int C>
__global__ __launch_bounds__(256) void grand_bwd_source_kernel(BwdSArgs p) {
    using K = Cfg<C>;
    extern __shared__ float4 smem4[];
    float* ys = reinterpret_cast<float*>(smem4);
    float* os = ys + K::TILE_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = tid / K::LPN, sub = tid % K::LPN;
    const float4* xin4 = reinterpret_cast<const float4*>(p.x_in);
    const float4* gin4 = reinterpret_cast<const float4*>(p.g_in);
    const float4* dxd4 = reinterpret_cast<const float4*>(p.dxd);
    float4* gout4 = reinterpret_cast<float4*>(p.g_out);
    constexpr int V = C / 4;

    TileGemm<C, false> gemm;                                    // os = y A^T
    float arow[K::MFMA ? 1 : 4][K::MFMA ? 1 : C];
    if constexpr (K::MFMA) {
        gemm.load(p.A, nullptr, lane, wave);
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) arow[t][c] = p.A[(4 * sub + t) * C + c];
    }
    const float4 p0v = *reinterpret_cast<const float4*>(p.p0 + 4 * sub);

    const TileRange tr = tile_range(p.n_tiles);
    for (int t = tr.t; t < tr.t_end; t += tr.step) {
        const int node0 = t * K::TM;
        float4 zr[K::ITERS];
        float sg[K::ITERS];
#pragma unroll
        for (int it = 0; it < K::ITERS; ++it) {
            const int li = it * K::SLOTS + slot;
            const int j = node0 + li;
            float4 z = f4zero(), y = f4zero();
            float sig = 0.f;
            if (j < p.n_nodes) {
                const int e0 = p.rowptr[j], deg = p.rowptr[j + 1] - e0;
                if (deg <= GADAPT_MAXD) {
                    float4 gi[GADAPT_MAXD], xi[GADAPT_MAXD];
                    float2 ev[GADAPT_MAXD];
#pragma unroll
                    for (int k = 0; k < GADAPT_MAXD; ++k) {
                        gi[k] = f4zero(); xi[k] = f4zero(); ev[k] = make_float2(0.f, 0.f);
                        if (k < deg) {
                            const int i = p.col[e0 + k];
                            ev[k] = p.edge_ws[p.perm[e0 + k]];
                            gi[k] = gin4[(size_t)i * V + sub];
                            xi[k] = xin4[(size_t)i * V + sub];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < GADAPT_MAXD; ++k) { axpy4(z, ev[k].x, gi[k]); axpy4(y, ev[k].y, xi[k]); sig += ev[k].y; }
                } else {
                    for (int k = 0; k < deg; ++k) {
                        const int i = p.col[e0 + k];
                        const float2 ev = p.edge_ws[p.perm[e0 + k]];
                        axpy4(z, ev.x, gin4[(size_t)i * V + sub]);
                        axpy4(y, ev.y, xin4[(size_t)i * V + sub]);
                        sig += ev.y;
                    }
                }
            }
            *reinterpret_cast<float4*>(ys + li * K::LD + 4 * sub) = y;
            zr[it] = z; sg[it] = sig;
        }
        __syncthreads();
        if constexpr (K::MFMA) {
            gemm.run(ys, os);
            __syncthreads();
        }
#pragma unroll
        for (int it = 0; it < K::ITERS; ++it) {
            const int li = it * K::SLOTS + slot;
            const int j = node0 + li;
            if (j >= p.n_nodes) continue;
            float4 r;
            if constexpr (K::MFMA) {
                r = *reinterpret_cast<const float4*>(os + li * K::LD + 4 * sub);
            } else {
                r = f4zero();
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float yc = ys[li * K::LD + c];
                    r.x = fmaf(arow[0][c], yc, r.x); r.y = fmaf(arow[1][c], yc, r.y);
                    r.z = fmaf(arow[2][c], yc, r.z); r.w = fmaf(arow[3][c], yc, r.w);
                }
            }
            const float4 d = dxd4[(size_t)j * V + sub];
            const float4 z = zr[it];
            const float s = sg[it];
            r.x = (r.x + fmaf(s, p0v.x, z.x)) + d.x; r.y = (r.y + fmaf(s, p0v.y, z.y)) + d.y;
            r.z = (r.z + fmaf(s, p0v.z, z.z)) + d.z; r.w = (r.w + fmaf(s, p0v.w, z.w)) + d.w;
            gout4[(size_t)j * V + sub] = r;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void coeffs_fwd_kernel(const float* wq, const float* bq, const float* wk, float* a, float* p0, int c) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < c * c) {
        const int o = e / c, cc = e % c;
        float v = 0.f;
        for (int r = 0; r < c; ++r) v = fmaf(wk[r * c + o], wq[r * c + cc], v);
        a[e] = v;
    } else if (e < c * c + c) {
        const int o = e - c * c;
        float v = 0.f;
        for (int r = 0; r < c; ++r) v = fmaf(wk[r * c + o], bq[r], v);
        p0[o] = v;
    }
}

__global__ void coeffs_bwd_kernel(const float* wq, const float* bq, const float* wk, const float* d_a, const float* d_p0,
                                  float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int c2 = c * c;
    if (e < c2) {                       // d_wq[r][cc] = sum_o wk[r][o] dA[o][cc]
        const int r = e / c, cc = e % c;
        float v = 0.f;
        for (int o = 0; o < c; ++o) v = fmaf(wk[r * c + o], d_a[o * c + cc], v);
        d_wq[e] = v;
    } else if (e < 2 * c2) {            // d_wk[r][o] = sum_cc wq[r][cc] dA[o][cc] + bq[r] dp0[o]
        const int f = e - c2, r = f / c, o = f % c;
        float v = bq[r] * d_p0[o];
        for (int cc = 0; cc < c; ++cc) v = fmaf(wq[r * c + cc], d_a[o * c + cc], v);
        d_wk[f] = v;
    } else if (e < 2 * c2 + c) {        // d_bq[r] = sum_o wk[r][o] dp0[o]
        const int r = e - 2 * c2;
        float v = 0.f;
        for (int o = 0; o < c; ++o) v = fmaf(wk[r * c + o], d_p0[o], v);
        d_bq[r] = v;
    } else if (e < 2 * c2 + 2 * c) {
        d_bk[e - 2 * c2 - c] = 0.f;
    }
}

__global__ void encode_linear_kernel(const float* feats, const float* w, const float* b, float* x0,
                                     int64_t n_nodes, int f, int c) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_nodes * c) return;
    const int64_t i = e / c;
    const int o = (int)(e % c);
    float v = b ? b[o] : 0.f;
    for (int k = 0; k < f; ++k) v = fmaf(feats[i * f + k], w[o * f + k], v);
    x0[e] = v;
}

// slab [n_rows][row_len] -> part [CHUNKS][row_len]
__global__ void slab_reduce1_kernel(const float* slab, float* part, int n_rows, int row_len) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= row_len) return;
    const int per = (n_rows + GADAPT_SLAB_CHUNKS - 1) / GADAPT_SLAB_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(n_rows, r0 + per);
    float v = 0.f;
    for (int r = r0; r < r1; ++r) v += slab[(size_t)r * row_len + e];
    part[(size_t)blockIdx.y * row_len + e] = v;
}
__global__ void slab_reduce2_kernel(const float* part, float* d_a, float* d_p0, int c) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int row_len = c * c + c;
    if (e >= row_len) return;
    float v = 0.f;
    for (int k = 0; k < GADAPT_SLAB_CHUNKS; ++k) v += part[(size_t)k * row_len + e];
    if (e < c * c) d_a[e] = v; else d_p0[e - c * c] = v;
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int grid_for(int n_tiles, int max_blocks) {
    int g = (n_tiles + 7) & ~7;
    if (g > max_blocks) g = max_blocks;
    if (g < 8) g = 8;
    return g;
}
#define GADAPT_FWD_MAX_BLOCKS 1024
#define GADAPT_BWD_T_MAX_BLOCKS 512      /* target pass grid = slab row count */

template <typename KernelT> static void allow_lds(KernelT k, int bytes) {
    if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int C> static int launch_fwd(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                                       const float* lp, float* alpha_out, int residual_only, hipStream_t st) {
    using K = Cfg<C>;
    FwdArgs p{x_in, x_out, a, p0, lp, g->rowptr_t, g->col_t, alpha_out, g->n_nodes, (g->n_nodes + K::TM - 1) / K::TM, residual_only};
    allow_lds(grand_fwd_kernel<C>, K::LDS_BYTES);
    ProfScope prof(0, st);
    hipLaunchKernelGGL(grand_fwd_kernel<C>, dim3(grid_for(p.n_tiles, GADAPT_FWD_MAX_BLOCKS)), dim3(256), K::LDS_BYTES, st, p);
    return check_launch("grand_fwd_kernel");
}
template <int C> static int launch_bwd(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha,
                                       const float* a, const float* p0, const float* lp, float* edge_ws, float* dxd, float* slab,
                                       int accumulate, float* sums_out, float* g_out, int residual_only, hipStream_t st) {
    using K = Cfg<C>;
    const int n_tiles = (g->n_nodes + K::TM - 1) / K::TM;
    BwdTArgs pt{x_in, g_in, alpha, a, lp, g->rowptr_t, g->col_t, reinterpret_cast<float2*>(edge_ws), dxd, slab, sums_out,
                g->n_nodes, n_tiles, accumulate, residual_only};
    allow_lds(grand_bwd_target_kernel<C>, K::LDS_BYTES);
    int rc;
    {
        ProfScope prof(1, st);
        hipLaunchKernelGGL(grand_bwd_target_kernel<C>, dim3(grid_for(n_tiles, GADAPT_BWD_T_MAX_BLOCKS)), dim3(256), K::LDS_BYTES, st, pt);
        rc = check_launch("grand_bwd_target_kernel");
    }
    if (rc || !g_out) return rc;
    BwdSArgs ps{x_in, g_in, reinterpret_cast<const float2*>(edge_ws), dxd, a, p0, g->rowptr_s, g->col_s, g->perm_s, g_out,
                g->n_nodes, n_tiles};
    allow_lds(grand_bwd_source_kernel<C>, K::LDS_BYTES);
    ProfScope prof(2, st);
    hipLaunchKernelGGL(grand_bwd_source_kernel<C>, dim3(grid_for(n_tiles, GADAPT_FWD_MAX_BLOCKS)), dim3(256), K::LDS_BYTES, st, ps);
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

static int check_graph(const gadapt_graph* g) {
    if (!g || g->n_nodes <= 0 || g->n_edges < 0 || !g->rowptr_t || !g->col_t) return fail(GADAPT_E_BADARG, "bad graph");
    return GADAPT_OK;
}

extern "C" int gadapt_layer_forward(const gadapt_graph* g, const float* x_in, float* x_out, const float* a, const float* p0,
                                    const float* layer_params, float* alpha_out, int residual_only, int c, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (!x_in || !x_out || !a || !p0 || !layer_params || x_in == x_out) return fail(GADAPT_E_BADARG, "layer_forward: null or aliased pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GADAPT_DISPATCH_C(c, launch_fwd<CC>(g, x_in, x_out, a, p0, layer_params, alpha_out, residual_only, st));
}

template <int C> static int tiles_for(int64_t n_nodes) { return (int)((n_nodes + Cfg<C>::TM - 1) / Cfg<C>::TM); }
extern "C" int gadapt_backward_slab_rows(int64_t n_nodes, int c) {
    if (n_nodes <= 0) return fail(GADAPT_E_BADARG, "slab_rows: bad node count");
    GADAPT_DISPATCH_C(c, grid_for(tiles_for<CC>(n_nodes), GADAPT_BWD_T_MAX_BLOCKS));
}
extern "C" int64_t gadapt_backward_slab_floats(int64_t n_nodes, int c) {
    const int rows = gadapt_backward_slab_rows(n_nodes, c);
    return rows < 0 ? rows : (int64_t)rows * (c * c + c);
}

extern "C" int gadapt_layer_backward(const gadapt_graph* g, const float* x_in, const float* g_in, const float* alpha,
                                     const float* a, const float* p0, const float* layer_params, float* edge_ws, float* dxd_ws,
                                     float* slab, int accumulate, float* sums_out, float* g_out, int residual_only, int c, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (!x_in || !g_in || !alpha || !a || !p0 || !layer_params || !edge_ws || !dxd_ws || !slab)
        return fail(GADAPT_E_BADARG, "layer_backward: null pointer");
    if (g_out && (!g->rowptr_s || !g->col_s || !g->perm_s)) return fail(GADAPT_E_BADARG, "layer_backward: source CSR missing");
    if (g_out == g_in || g_out == dxd_ws) return fail(GADAPT_E_BADARG, "layer_backward: g_out aliases an input");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GADAPT_DISPATCH_C(c, launch_bwd<CC>(g, x_in, g_in, alpha, a, p0, layer_params, edge_ws, dxd_ws, slab, accumulate, sums_out, g_out, residual_only, st));
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

extern "C" int gadapt_coeffs_forward(const float* wq, const float* bq, const float* wk, float* a_out, float* p0_out, int c, void* stream) {
    if (!wq || !bq || !wk || !a_out || !p0_out || c <= 0) return fail(GADAPT_E_BADARG, "coeffs_forward: bad argument");
    const int n = c * c + c;
    hipLaunchKernelGGL(coeffs_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), wq, bq, wk, a_out, p0_out, c);
    return check_launch("coeffs_fwd_kernel");
}
extern "C" int gadapt_coeffs_backward(const float* wq, const float* bq, const float* wk, const float* d_a, const float* d_p0,
                                      float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c, void* stream) {
    if (!wq || !bq || !wk || !d_a || !d_p0 || !d_wq || !d_bq || !d_wk || !d_bk || c <= 0) return fail(GADAPT_E_BADARG, "coeffs_backward: bad argument");
    const int n = 2 * c * c + 2 * c;
    hipLaunchKernelGGL(coeffs_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), wq, bq, wk, d_a, d_p0,
                       d_wq, d_bq, d_wk, d_bk, c);
    return check_launch("coeffs_bwd_kernel");
}

extern "C" int gadapt_encode_linear(const float* feats, const float* w, const float* b, float* x0, int64_t n_nodes, int f, int c, void* stream) {
    if (!feats || !w || !x0 || n_nodes <= 0 || f <= 0 || c <= 0) return fail(GADAPT_E_BADARG, "encode_linear: bad argument");
    const int64_t n = n_nodes * c;
    hipLaunchKernelGGL(encode_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), feats, w, b, x0, n_nodes, f, c);
    return check_launch("encode_linear_kernel");
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

// ------------------------------------------------------------------------------------------------
// L-step Euler block (GNN.py:273-291)
// ------------------------------------------------------------------------------------------------
extern "C" int gadapt_block_forward(const gadapt_graph* g, float* x_all, int n_layers, const float* a, int64_t a_stride,
                                    const float* p0, int64_t p0_stride, const float* layer_params, float* alpha_all, int c, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (!x_all || n_layers <= 0 || !a || !p0 || !layer_params) return fail(GADAPT_E_BADARG, "block_forward: bad argument");
    const size_t nc = (size_t)g->n_nodes * c;
    for (int l = 0; l < n_layers; ++l) {
        int rc = gadapt_layer_forward(g, x_all + l * nc, x_all + (l + 1) * nc, a + l * a_stride, p0 + l * p0_stride,
                                      layer_params + 2 * l, alpha_all ? alpha_all + (size_t)l * g->n_edges : nullptr, 0, c, stream);
        if (rc) return rc;
    }
    return GADAPT_OK;
}

extern "C" int gadapt_block_backward(const gadapt_graph* g, const float* x_all, const float* alpha_all, const float* g_top, int n_layers,
                                     const float* a, int64_t a_stride, const float* p0, int64_t p0_stride, const float* layer_params,
                                     float* g_ws, float* dxd_ws, float* edge_ws, float* slab, float* d_layer_params, float* d_x0,
                                     int c, void* stream) {
    if (int rc = check_graph(g)) return rc;
    if (!x_all || !alpha_all || !g_top || n_layers <= 0 || !a || !p0 || !layer_params || !g_ws || !dxd_ws || !edge_ws || !slab)
        return fail(GADAPT_E_BADARG, "block_backward: bad argument");
    const size_t nc = (size_t)g->n_nodes * c;
    const bool shared = (a_stride == 0);
    const int64_t slab_floats = gadapt_backward_slab_floats(g->n_nodes, c);
    if (slab_floats < 0) return (int)slab_floats;
    const float* g_cur = g_top;
    for (int l = n_layers - 1; l >= 0; --l) {
        float* g_next = (l == 0) ? d_x0 : g_ws + ((n_layers - 1 - l) & 1) * nc;
        float* slab_l = shared ? slab : slab + (size_t)l * slab_floats;
        const int accumulate = (shared && l != n_layers - 1) ? 1 : 0;
        int rc = gadapt_layer_backward(g, x_all + l * nc, g_cur, alpha_all + (size_t)l * g->n_edges, a + l * a_stride, p0 + l * p0_stride,
                                       layer_params + 2 * l, edge_ws, dxd_ws, slab_l, accumulate,
                                       d_layer_params ? d_layer_params + 2 * l : nullptr, g_next, 0, c, stream);
        if (rc) return rc;
        g_cur = g_next;
    }
    return GADAPT_OK;
}
