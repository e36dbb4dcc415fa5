/*
 * gadapt_hip.h - C-ABI of the MI355X-native g-adaptivity message-passing path.
 *
 * Plain pointers and sizes only: no torch types cross this boundary.  Every device
 * pointer is fp32 / int32 HBM memory owned by the caller; `stream` is a hipStream_t
 * passed as void*.  All entry points return 0 on success or a negative GADAPT_E_*
 * code and never abort; `gadapt_last_error()` gives the message.  Nothing here
 * allocates, frees or synchronises (graph-capture safe) except the *_host helpers.
 *
 * The reference (/root/reference, 100 % Python) has no FFI; each entry point cites
 * the reference code whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer adds on the reference side.
 *
 * Formulation.  One GRAND layer of the reference is
 *     Q = x Wq^T + bq,  K = x Wk^T + bk,  s_ij = <Q_i,K_j>/sqrt(C)   (GRAND_plus.py:225-226,279)
 *     alpha_ij = softmax_j(s_ij / T)  over edges j->i                 (GRAND_plus.py:326-333)
 *     x'_i = x_i + dt * (sum_j alpha_ij x_j - x_i)                    (GRAND_plus.py:338-343,267; GNN.py:288-291)
 * <Q_i,K_j> = P_i . x_j + <Q_i,bk>  with  P_i = A x_i + p0,  A = Wk^T Wq,  p0 = Wk^T bq;
 * the second term does not depend on j and cancels in the softmax, so the kernels
 * work from (A, p0): one [C,C] projection per node instead of two, and the edge walk
 * gathers x_j only.
 *
 * Arithmetic.  Everything is fp32 in and out.  At hidden sizes >= 32 the [nodes,C] x [C,C] products (P = x A^T, dP A, A y and
 * the dA partial) run on the 16-bit matrix cores from operands split into pieces that restore fp32 accuracy: three bf16
 * pieces (x = h + m + l exactly, six piece products, the dropped ones below 2^-23 |x y|) or, where the operand tile is prepared
 * once per tile, two f16 pieces after an exact power-of-two scaling per row / column group (hh + hl + lh, dropped term below
 * 2^-22 |x y|); piece products are exact in the fp32 accumulators.  The softmax uses expf and a true
 * division.  Measured against an fp64 evaluation: coordinates 1e-7 relative, parameter gradients at the fp32 reference
 * path's own rounding level (tests/test_gpu_parity.py, full BASELINE sizes included).
 */
#ifndef GADAPT_HIP_H
#define GADAPT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GADAPT_OK            0
#define GADAPT_E_BADARG     -1   /* null pointer, negative size, unsupported hidden_dim ... */
#define GADAPT_E_LAUNCH     -2   /* hipGetLastError() after a launch */
#define GADAPT_E_RANGE      -3   /* edge endpoint outside [0, n_nodes) */
#define GADAPT_E_RUNTIME    -4   /* a run-time dependency is missing or failed (gadapt_allreduce_flat: RCCL) */

/* Hidden sizes the fused kernels are built for; others -> GADAPT_E_BADARG. */
int  gadapt_supported_hidden_dim(int c);
const char* gadapt_last_error(void);
int  gadapt_abi_version(void);
/* Forget a pending error of the calling thread: the message of gadapt_last_error() and HIP's own sticky per-thread last error
 * (hipGetLastError).  For callers that recover from a failure that did not come through this library - e.g. a hipGraph capture
 * that was invalidated: every later launch check would otherwise report that stale error once.  Returns the HIP error code that
 * was pending (0: none). */
int  gadapt_clear_error(void);

/* ------------------------------------------------------------------ graph
 * Replaces the per-forward edge bookkeeping of PyG's MessagePassing.propagate
 * (called at GRAND_plus.py:233-234) with a one-time CSR build.  Input is the
 * edge list AFTER GNN.py:206-223 surgery, host memory, int64 as PyG holds it:
 * src[e] -> dst[e].  Outputs (host, int32, caller-allocated):
 *   rowptr_t[N+1], col_t[E] : in-edges grouped by target, col_t = source;
 *   eid_t[E]                : original edge id of each target-ordered slot
 *                             (alpha_out[slot] belongs to edge eid_t[slot]);
 *   rowptr_s[N+1], col_s[E] : out-edges grouped by source, col_s = target;
 *   perm_s[E]               : target-ordered slot of each source-ordered slot;
 *   tpos_s[E]               : its inverse - source-ordered slot of each target-ordered slot.
 * Order inside a row is the input order (stable), so results are reproducible. */
int gadapt_csr_build_host(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes,
                          int32_t* rowptr_t, int32_t* col_t, int32_t* eid_t,
                          int32_t* rowptr_s, int32_t* col_s, int32_t* perm_s, int32_t* tpos_s);

/* Per-tile staging metadata for the kernels: for tiles of `tile_rows` consecutive rows of one CSR orientation,
 * meta[4t+0] = first entry of the tile's slice, [4t+1] = its length, [4t+2] = the longest row,
 * [4t+3] = 1 if every neighbour of the tile's rows lies in tiles t-1..t+1 (the kernels then gather from an LDS
 * window instead of L2), else 0.  The kernels use tile heights 64, 128 and 256 (by hidden size): build all three
 * for both orientations (host pointers). */
int gadapt_tile_meta_host(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int tile_rows, int32_t* meta_out);

typedef struct gadapt_graph {
    int32_t n_nodes;
    int32_t n_edges;
    const int32_t* rowptr_t;  /* device */
    const int32_t* col_t;
    const int32_t* rowptr_s;
    const int32_t* col_s;
    const int32_t* perm_s;
    const int32_t* tpos_s;
    const int32_t* meta_t[3];  /* device: tile metadata of the target CSR for tile heights 64, 128, 256 */
    const int32_t* meta_s[3];  /* device: same for the source CSR */
    /* optional (NULL / 0 = absent): ELL-8 copies for the wide kernels (gadapt_ell_build_host) */
    const int32_t* ell_t;      /* device [round_up(N,256)][8]: in-neighbours of node i, -1 = unused */
    const int32_t* ell_s;      /* device: out-neighbours (targets), same layout */
    int32_t wide_deg_t;        /* longest in-row if the target orientation qualifies for the wide kernels, else 0 */
    int32_t wide_deg_s;        /* same for the source orientation */
    int32_t wide_big_deg_t;    /* longest in-row if the target orientation qualifies for the 512-row window of the wide forward
                                  (gadapt_wide_window_host(step 256, halo 128, rows <= 7): meshes with up to 128 nodes per row), else 0 */
    int32_t wide_half_deg_t;   /* ABI 9: longest in-row if the target orientation also qualifies for steps of 128 nodes with a 64-row halo
                                  (gadapt_wide_window_host(step 128, halo 64, rows <= 8)), else 0: batches of at most 32 768 nodes then run
                                  the wide forward on four-wave workgroups, twice as many of them */
} gadapt_graph;

/* ELL-8 copy of one CSR orientation (host pointers).  The wide kernels (hidden size 64: one wave owns 32 consecutive
 * nodes, two lanes per node, a 256-node workgroup step gathers from a 384-row LDS window) run when every row has at
 * most 8 entries and every neighbour of node i lies in rows [256*(i/256) - 64, 256*(i/256) + 320): row-major mesh
 * batches do; any other graph takes the tiled kernels.  ell_out: round_up(N,256)*8 int32; *max_deg_out = longest
 * row when the orientation qualifies, else 0. */
int gadapt_ell_build_host(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t* ell_out, int32_t* max_deg_out);
/* The same locality test for steps of `step` nodes, a window of step + 2*halo rows and rows of at most max_row entries: *max_deg_out =
 * longest row when every neighbour of node i lies in rows [step*(i/step) - halo, step*(i/step) + step + halo), else 0.  step 256 / halo
 * 128 / max_row 7 is what the wide forward's 512-row window takes (row-major meshes with up to 128 nodes per mesh row: BASELINE config
 * 5); step 128 / halo 64 / max_row 8 what its four-wave form takes (gadapt_graph::wide_half_deg_t). */
int gadapt_wide_window_host(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int step, int halo, int max_row, int32_t* max_deg_out);

/* ------------------------------------------------------------------ weights
 * A[o][c] = sum_r Wk[r][o] Wq[r][c],  p0[o] = sum_r Wk[r][o] bq[r].
 * Wq,Wk are torch Linear weights [out,in] (GRAND_plus.py:146-147). */
int gadapt_coeffs_forward(const float* wq, const float* bq, const float* wk,
                          float* a_out, float* p0_out, int c, void* stream);
/* Chain rule back to the Linear parameters.  dbk is written as exact zeros:
 * d/dbk vanishes because softmax is shift-invariant per target. */
int gadapt_coeffs_backward(const float* wq, const float* bq, const float* wk,
                           const float* d_a, const float* d_p0,
                           float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c, void* stream);

/* ------------------------------------------------------------------ encoder
 * x0 = feats @ W^T (+ b): get_enc (GNN.py:72-98, call :270).  feats [N,F], W [C,F]. */
int gadapt_encode_linear(const float* feats, const float* w, const float* b /*nullable*/,
                         float* x0, int64_t n_nodes, int f, int c, void* stream);
/* Same with the feature assembly of GNN.py:225-239 folded in: the node feature vector is
 * [x_comp[i, 0..dim), f_tensor[i] (if given), uu_tensor[i] (if given)], W [C, dim + extras]. */
int gadapt_encode_features(const float* x_comp, int dim, const float* f_tensor /*nullable*/,
                           const float* uu_tensor /*nullable*/, const float* w, const float* b /*nullable*/,
                           float* x0, int64_t n_nodes, int c, void* stream);

/* gadapt_encode_features and gadapt_coeffs_forward (of ONE conv: the weight-shared layer of GNN.py:131-140) in a single
 * launch: both precede layer 0 and are independent of each other.  c = encoder output width, c_conv = conv width. */
int gadapt_encode_features_coeffs(const float* x_comp, int dim, const float* f_tensor /*nullable*/,
                                  const float* uu_tensor /*nullable*/, const float* w, const float* b /*nullable*/,
                                  float* x0, int64_t n_nodes, int c,
                                  const float* wq, const float* bq, const float* wk, float* a_out, float* p0_out, int c_conv,
                                  void* stream);

/* ------------------------------------------------------------------ one layer
 * layer_params (device, 2 floats): {dt, score_scale} with score_scale = 1/(sqrt(C)*T).
 * alpha_out (nullable) [E] in target order.  x_out must not alias x_in.
 * residual_only = 0: x_out = x + dt*(A(x)x - x)   (conv + Euler update, GNN.py:288-291)
 * residual_only = 1: x_out = A(x)x - x            (what GRAND_plusConv.forward returns,
 *                                                  GRAND_plus.py:267; dt is ignored)   */
int gadapt_layer_forward(const gadapt_graph* g, const float* x_in, float* x_out,
                         const float* a, const float* p0, const float* layer_params,
                         float* alpha_out, int residual_only, int c, void* stream);

/* Workspace sizing for the backward: the target pass leaves one partial-sum row of
 * (C*C + C) floats per workgroup ("slab"); rows depends on the node count only. */
int     gadapt_backward_slab_rows(int64_t n_nodes, int c);
int64_t gadapt_backward_slab_floats(int64_t n_nodes, int c);   /* rows * (C*C + C) */

/* Backward of one layer, split in two launches:
 *   target pass: per in-edge d(score), per node dP, weight-gradient partials, and
 *                dxd = (1-dt) g + A^T dP;
 *   source pass: g_out = dxd + sum over out-edges (alpha dt g_i + dscore A x_i ...) .
 * g_in = dL/dx_out [N,C]; x_in = the layer's input; alpha = forward's alpha_out.
 * edge_ws [E] float2 (filled in source order through tpos_s), dxd_ws [N,C] scratch.  slab accumulates (accumulate!=0) or is
 * overwritten.  sums_out (nullable, 2 floats per call, atomically accumulated):
 * {d dt, d score_scale}.  Pass g_out = NULL to skip the source pass (layer 0 with a
 * frozen encoder: GNN.py:82). */
int gadapt_layer_backward(const gadapt_graph* g, const float* x_in, const float* g_in,
                          const float* alpha, const float* a, const float* p0,
                          const float* layer_params,
                          float* edge_ws, float* dxd_ws, float* slab, int accumulate,
                          float* sums_out, float* g_out, int residual_only, int c, void* stream);

/* slab [n_rows][C*C+C] -> d_a [C*C], d_p0 [C].  scratch: 32*(C*C+C) floats. */
int gadapt_slab_reduce(const float* slab, int n_rows, float* scratch, float* d_a, float* d_p0, int c, void* stream);

/* gadapt_slab_reduce followed by gadapt_coeffs_backward in two launches instead of three: the second-level sums and the
 * chain rule share a kernel (d_a / d_p0 only ever exist in LDS).  Same results bit for bit. */
int gadapt_slab_reduce_coeffs_backward(const float* slab, int n_rows, float* scratch,
                                       const float* wq, const float* bq, const float* wk,
                                       float* d_wq, float* d_bq, float* d_wk, float* d_bk, int c, void* stream,
                                       const float* lp_partials /*nullable*/, int n_layers, int want_d_scale,
                                       float* d_layer_params /*nullable*/);
/* lp_partials (nullable): the [2][L][n_rows] workspace gadapt_block_backward filled - the first launch then also does what
 * gadapt_layer_params_reduce does (a few extra workgroups instead of a launch of its own). */

/* ------------------------------------------------------------------ L-step Euler block
 * The loop of GNN.py:273-291 with weight sharing (GNN.py:131-141): x_all is
 * [(L+1),N,C] with x_all[0] = encoder output on entry; layer l reads x_all[l],
 * writes x_all[l+1].  a/p0/layer_params advance by *_stride floats per layer
 * (0 = shared).  alpha_all (nullable) [L,E].
 * x0_cols = 0: slot 0 holds the dense [N,C] encoder output.  x0_cols = 4: slot 0 holds, at its start, the COMPACT
 * [N,4] output of the identity encoder (zero-pad to C columns, GNN.py:75-82: columns 4.. are zero by construction);
 * layer 0 reads it without the padded matrix ever being written (needs >= 2 layers, hidden >= 8). */
int gadapt_block_forward(const gadapt_graph* g, float* x_all, int x0_cols, int n_layers,
                         const float* a, int64_t a_stride, const float* p0, int64_t p0_stride,
                         const float* layer_params, float* alpha_all, float* x_top4, int c, void* stream);
/* x_top4 (nullable) [N,4]: when given, the last layer writes ONLY columns 0..3 of its output rows, to x_top4, and slot L
 * of x_all is left unwritten: the caller wants x_phys = x[:, :dim] (GNN.py:299) and nothing downstream reads the rest. */

/* Backward of the block.  g_top = dL/dx_all[L] (not modified): [N,C] when g_top_cols = 0, or the compact [N,g_top_cols]
 * (1..4 columns, zero beyond: the backward of x_phys = x[:, :dim], GNN.py:299, without materialising the padded
 * matrix).  x0_cols as in gadapt_block_forward (then d_x0 must be NULL).  g_ws: 2*N*C floats,
 * dxd_ws: N*C, edge_ws: 2*E, slab: n_slots*gadapt_backward_slab_floats(N,c) with
 * n_slots = 1 (shared weights) or L.  d_layer_params (nullable): WORKSPACE of 2 * L * gadapt_backward_slab_rows(N,c) floats -
 * every target-pass workgroup writes its partial of d dt_l (learn_step, GNN.py:179-180,288-289) and, when want_d_scale != 0, of
 * d score_scale_l into its own slot; gadapt_layer_params_reduce then sums them in a fixed order (no float atomics: the step and
 * temperature gradients are bit-reproducible).  want_d_scale = 0: the per-edge log terms of d score_scale are not computed.
 * d_x0 (nullable) [N,C] receives dL/dx_all[0]. */
int gadapt_block_backward(const gadapt_graph* g, const float* x_all, int x0_cols, const float* alpha_all,
                          const float* g_top, int g_top_cols, int n_layers,
                          const float* a, int64_t a_stride, const float* p0, int64_t p0_stride,
                          const float* layer_params,
                          float* g_ws, float* dxd_ws, float* edge_ws, float* slab,
                          float* d_layer_params, int want_d_scale, float* d_x0, int c, void* stream);
/* ------------------------------------------------------------------ fused training step (ABI 8)
 * The reference's iteration - zero_grad, model(data), F.mse_loss / F.l1_loss(out, data.x_phys), backward, Adam step
 * (run_GNN.py:99-131; loss :80-84,106; Adam :88,128-131) - for a weight-shared GRAND / GRAND_plus block behind the identity
 * encoder, as 14 launches instead of 16 (and without the two that cost most: the loss and the encoder): gadapt_block_forward_loss, gadapt_block_backward (g_top = seed, g_top_cols = d),
 * gadapt_step_tail.
 *
 * gadapt_block_forward_loss = gadapt_block_forward(x0_cols = 4, x_top4) with two things folded into its launches:
 *   - layer 0 assembles its compact [N,4] input [x_comp | f | uu | 0] from the caller's node fields (the feature concatenation of
 *     GNN.py:225-239 through the zero-pad encoder GNN.py:75-82; f_tensor / uu_tensor nullable, dim + extras <= 4) and writes it to
 *     the start of x_all's slot 0, where the layer-0 backward reads it: no encoder launch;
 *   - the last layer's head-only launch also writes seed [N,d] = d loss / d x_top4[:, :d] (same arithmetic as gadapt_loss_forward)
 *     and one partial of sum |x_top4[:, :d] - target|^p per wave to loss_partials (gadapt_loss_partials_max() floats): no loss
 *     launch.  Returns the number of partials written (> 0) or a negative error code.  target == NULL (seed, loss_partials unused): no
 *     loss, returns 0 - the evaluation forward on the node fields as one call (the layer-0 launch reads them, x_top4 is the result).
 * (a, p0): the composite coefficients of the shared conv.  param == NULL: INPUTS - from gadapt_coeffs_forward before the first
 * step, from the previous step's gadapt_step_tail afterwards.  param != NULL (only where gadapt_forward_computes_coeffs(g, c) = 1:
 * hidden 64 on a graph the wide forward kernel takes): the flat bucket [Wq | bq | Wk | bk] the step trains - the layer-0 launch
 * computes A = Wk^T Wq, p0 = Wk^T bq itself (gadapt_coeffs_forward's arithmetic; every workgroup for its own operand fragments,
 * while its window of node rows is in flight) and writes them to a / p0 for the launches that follow: OUTPUTS, and no coefficient
 * launch anywhere in the step. */
int gadapt_loss_partials_max(void);
int gadapt_forward_computes_coeffs(const gadapt_graph* g, int c);
int gadapt_block_forward_loss(const gadapt_graph* g, float* x_all, const float* x_comp, int dim, const float* f_tensor /*nullable*/,
                              const float* uu_tensor /*nullable*/, int n_layers, float* a, float* p0, const float* param /*nullable*/,
                              const float* layer_params, float* alpha_all, float* x_top4,
                              const float* target, int d, int l1, float* seed, float* loss_partials, int c, void* stream);
/* Tail of the step.  slab != NULL (one GPU): the first-level slab sums (as gadapt_slab_reduce_coeffs_backward's first launch; one
 * extra workgroup advances the optimizer's device step count and sums the loss partials in a fixed order: loss_out[0] = sum /
 * loss_count); ONE launch that finishes the sums, applies the chain rule to the flat gradient grad = [dWq | dbq | dWk | dbk]
 * (gadapt_slab_reduce_coeffs_backward's arithmetic) and runs Adam on the flat bucket param = [Wq | bq | Wk | bk]
 * (gadapt_adam_step_dev's arithmetic and state; a workgroup owns whole rows r of Wq / Wk with bq[r], bk[r] - the chain rule of those
 * entries reads only entries the same workgroup owns, so the launch needs no exchange between workgroups); and, when a_out / p0_out
 * are given, gadapt_coeffs_forward of the UPDATED weights -> (a_out, p0_out) for the next step (NULL where the next forward computes
 * them itself: gadapt_forward_computes_coeffs) - 14 or 13 launches per step.  slab == NULL (data parallel): grad is given - the
 * caller all-reduced it - and the launches are step count, Adam (+ coefficients); its first half is the same call with slab given and
 * exp_avg = exp_avg_sq = NULL: the two sum launches (loss included), stopping at the flat gradient.
 * state = int32[2] {steps taken, 0} as for gadapt_adam_step_dev.
 * Bit-identical parameters, moments and coefficients to the separate launches (tests/test_gpu_training.py). */
int gadapt_step_tail(const float* slab /*nullable*/, int n_rows, float* scratch, float* param, float* grad, float* exp_avg,
                     float* exp_avg_sq, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t* state,
                     float grad_scale, float* a_out, float* p0_out,
                     const float* loss_partials /*nullable*/, int n_loss_partials, float* loss_out, int64_t loss_count,
                     int c, void* stream);

/* ------------------------------------------------------------------ small meshes: the whole evaluation forward in ONE launch
 * The reference's own sizes (params.py:37,56,107,130-134: 11x11 ... 23x23 meshes, hidden 8, 4 layers; evaluation one sample per call,
 * utils_eval.py:128-130,193-201; the Burgers rollout, utils_eval_Burgers.py:282-300): encoder (GNN.py:225-239,270), composite
 * coefficients, the L Euler steps (GNN.py:273-291 on GRAND_plus.py:225-343) and the head x[:, :out_cols] (GNN.py:299) by ONE workgroup
 * per mesh - the meshes of a batch are disconnected components, so only workgroup barriers are needed.  mesh_ptr [n_meshes+1]
 * (device): first node of each mesh (PyG's Batch.ptr); every edge must stay inside its mesh.  mesh_eptr (nullable, device)
 * [n_meshes+1]: first in-edge of each mesh = rowptr_t[mesh_ptr[m]] - given, a workgroup reads its node and edge bounds side by
 * side instead of one after the other (a dependent round trip of a ~10 us launch); n_meshes == 1 reads neither.
 * max_mesh_nodes / max_mesh_edges: the largest mesh
 * (<= 1024 nodes, 512 at hidden 32: one to four lanes per node; rows + CSR slice must fit 160 KB: gadapt_small_forward_lds_bytes returns the bytes, or -1).  hidden c in
 * {4, 8, 16, 32}.  x_comp [N,dim], f / uu [N] nullable, enc_w [c, n_feat] the frozen bias-free encoder weight; wq / bq / wk advance by
 * w_stride / b_stride floats per layer (0 = shared); layer_params [L,2] = (dt, score scale).  out [N,out_cols]; alpha_all nullable
 * [L,E] (target-CSR order).  No activations are kept: inference only. */
int64_t gadapt_small_forward_lds_bytes(int max_mesh_nodes, int max_mesh_edges, int c);
int gadapt_small_forward(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                         const float* x_comp, int dim, const float* f, const float* uu, const float* enc_w, int n_feat,
                         const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                         const float* layer_params, int n_layers, float* out, int out_cols, float* alpha_all, float* x_all, int c, void* stream);
/* The same launch as the head of a fused training step on a small-mesh batch (ABI 9): x_all / alpha_all are kept for
 * gadapt_small_backward, and the launch also writes seed = d loss / d out [N,out_cols] (loss = mean squared error over all N*out_cols
 * entries against target [N,out_cols], or mean absolute error when l1 != 0: run_GNN.py:80-84,106; the arithmetic of
 * gadapt_loss_forward, bit-identical gradients) and one partial sum of the loss per wave to loss_partials (gadapt_loss_partials_max()
 * floats; gadapt_step_tail sums them).  Returns the number of partials (> 0) or a negative error code.  1 <= out_cols <= 4. */
int gadapt_small_forward_loss(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                              const float* x_comp, int dim, const float* f, const float* uu, const float* enc_w, int n_feat,
                              const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                              const float* layer_params, int n_layers, float* out, int out_cols, float* alpha_all, float* x_all,
                              const float* target, int l1, float* seed, float* loss_partials, int c, void* stream);
/* x_all (nullable) [L,N,c]: when given, the input rows of every layer are kept for gadapt_small_backward (training; alpha_all must
 * be given too).
 * Backward of the same block in ONE launch, one workgroup per mesh (autograd of GRAND_plus.py:225-343 + GNN.py:288-291 through the
 * L layers; the encoder is frozen, GNN.py:82,89, so nothing is owed below layer 0): g_top = dL/dx_L[:, :g_cols] (zero beyond).
 * slab: [S][n_meshes][c*c + c] floats - per conv (S = 1 shared, L per-layer) one partial row per mesh (dA then dp0), the layout
 * gadapt_slab_reduce_coeffs_backward(slab + s * n_meshes * (c*c + c), n_meshes, ...) sums and chains to the Linear parameters. */
int64_t gadapt_small_backward_lds_bytes(int max_mesh_nodes, int max_mesh_edges, int c);
int gadapt_small_backward(const gadapt_graph* g, const int32_t* mesh_ptr, const int32_t* mesh_eptr, int n_meshes, int max_mesh_nodes, int max_mesh_edges,
                          const float* x_all, const float* alpha_all, const float* g_top, int g_cols,
                          const float* wq, const float* bq, const float* wk, int64_t w_stride, int64_t b_stride,
                          const float* layer_params, int n_layers, float* slab, int c, void* stream);

/* partials: the workspace gadapt_block_backward filled ([2][L][n_rows], n_rows = gadapt_backward_slab_rows).  d_layer_params
 * [2,L]: row 0 = d dt_l, row 1 = d score_scale_l (zeros when want_d_scale = 0) - two contiguous rows, so a caller can hand out
 * the d dt row as the gradients of L one-element step parameters laid side by side. */
int gadapt_layer_params_reduce(const float* partials, int n_rows, int n_layers, int want_d_scale, float* d_layer_params, void* stream);

/* ------------------------------------------------------------------ generic message-passing primitives
 * What PyG's MessagePassing.propagate / utils.softmax do with index_select + scatter, per CSR row, for the conv variants
 * besides GRAND / GRAND_plus that get_conv builds (GNN.py:108-124: GAT_plus GRAND_plus.py:386-416, GATConv, GCNConv), the
 * reg_skew scores (GRAND_plus.py:280-324) and hidden sizes the fused kernels are not built for.  Per-edge arrays are in
 * target-CSR order (like alpha_out).  transpose / by_source = 1 walks the source CSR (the transposed product a backward
 * pass needs).  C must be a multiple of 4.  Rows are summed in the caller's edge order: bit-reproducible.
 *   spmm:         out_i = sum_{e: j->i} w_e x_j (+ self_scale * x_i)      [transpose: out_j = sum_{e: j->i} w_e x_i]
 *                 w = NULL means w_e = 1.
 *   sddmm:        s_e = scale * <a_i, b_j> for e: j->i                    [transpose: scale * <a_j, b_i>]
 *   edge_softmax: alpha_e = exp(s_e - max_i) / (sum_i exp(.) + 1e-16) over the in-edges of i (PyG utils.softmax);
 *                 backward d_s_e = alpha_e (d_alpha_e - sum_i alpha d_alpha)
 *   edge_combine: o_e = u_src[j] + v_dst[i] (op 0) or u_src[j] * v_dst[i] (op 1) for e: j->i
 *   edge_rowsum:  r_i = sum of edge_vals over the in-edges of i           [by_source: over the out-edges of j]        */
int gadapt_spmm(const gadapt_graph* g, int transpose, const float* w /*nullable*/, const float* x, float* out, int c,
                float self_scale, void* stream);
int gadapt_sddmm(const gadapt_graph* g, int transpose, const float* a, const float* b, float* out, int c, float scale, void* stream);
int gadapt_edge_softmax_forward(const gadapt_graph* g, const float* scores, float* alpha, void* stream);
int gadapt_edge_softmax_backward(const gadapt_graph* g, const float* alpha, const float* d_alpha, float* d_scores, void* stream);
int gadapt_edge_combine(const gadapt_graph* g, const float* u_src, const float* v_dst, float* out, int op, void* stream);
int gadapt_edge_rowsum(const gadapt_graph* g, int by_source, const float* edge_vals, float* out, void* stream);
/* Per-edge VECTORS b [E,C] in target-CSR order (the conv option edge_dim / lin_edge, GRAND_plus.py:165-166: a per-edge term on
 * key and value, :273-277,338-340).  i = the target (row) of edge e.  C a multiple of 4.
 *   mode 0  s_e   = <a_i, b_e>                   out [E]      (score term <query_i, edge_e>; d w of mode 1)
 *   mode 1  out_i = sum_{e in row i} w_e b_e     out [N,C]    (aggregated edge term; d a of mode 0)
 *   mode 2  out_e = w_e a_i                      out [E,C]    (d b of modes 0 and 1)                                      */
int gadapt_edge_vector_op(const gadapt_graph* g, int mode, const float* w, const float* a, const float* b, float* out, int c, void* stream);

/* ------------------------------------------------------------------ GAT_plus block (fused)
 * Replaces, for conv_type = 'GAT_plus' (get_conv, GNN.py:120-121), the L iterations of GNN.forward's layer loop (GNN.py:273-296)
 * around GAT_plus.forward (GRAND_plus.py:400-416: GATConv attention with identity maps, re-applied as sparse(alpha)^T x):
 *     a = <x, att_src>, b = <x, att_dst>;  alpha_e = softmax_i(leaky_relu(a_j + b_i, 0.2)) over the in-edges e: j -> i of the
 *     SELF-LOOPED graph (GATConv: remove_self_loops + add_self_loops - `g` must be that graph);
 *     res_i = sum_e alpha_e x_j - x_i  (res_lap = 1, 'GAT_res_lap')  |  sum_e alpha_e x_j  (res_lap = 0, 'GAT_lin');
 *     x <- x + dt * non_lin(res)  (residual = 1)  |  x <- non_lin(res)  (residual = 0).
 * non_lin: 0 identity, 1 relu, 2 tanh, 3 sigmoid, 4 leaky_relu(0.01), 5 elu, 6 selu (get_nonlin, GNN.py:48-64).  Dropout is not
 * part of it (callers with dropout > 0 in training mode use the per-layer primitives below).
 * x_all [(L+1),N,C]: slot 0 = input, slot l+1 = output of layer l.  att_src / att_dst [C] per layer, att_stride floats apart
 * (0: one shared conv).  ab [L,2,N] (the per-node score halves of every layer) and alpha [L,max(E,1)] (target-CSR order of `g`)
 * are outputs the backward reads.  C in {4,...,128} like the GRAND kernels.
 * backward: g_top [N,C] = dL/dx_L; workspaces g_ws [2,N,C], gr_ws [N,C], dz_ws [max(E,1)], db_ws [N],
 * part_ws [gadapt_gat_plus_partial_rows(N,C), 2C]; d_att [S,2,C] = (d att_src | d att_dst) per distinct conv, written whole
 * (summed over the layers in a fixed order when shared: bit-reproducible); d_x0 [N,C] nullable. */
int gadapt_gat_plus_block_forward(const gadapt_graph* g, float* x_all, int n_layers, const float* att_src, const float* att_dst,
                                  int64_t att_stride, float dt, int residual, int res_lap, int non_lin, float* ab, float* alpha,
                                  int c, void* stream);
int gadapt_gat_plus_block_backward(const gadapt_graph* g, const float* x_all, const float* alpha, const float* ab, const float* g_top,
                                   int n_layers, const float* att_src, const float* att_dst, int64_t att_stride, float dt, int residual,
                                   int res_lap, int non_lin, float* g_ws, float* gr_ws, float* dz_ws, float* db_ws, float* part_ws,
                                   float* d_att, float* d_x0 /*nullable*/, int c, void* stream);
int gadapt_gat_plus_partial_rows(int64_t n_nodes, int c);

/* ------------------------------------------------------------------ loss seed
 * mesh_loss (run_GNN.py:80-84,106): loss = mean |x_phys - target|^p, p = 2 (mse) or 1 (l1),
 * x_phys = x_top[:, :d] (GNN.py:299).  Writes x_phys [N,d], g_top [N,C] (zero outside
 * the first d columns, scaled by grad_scale/(N*d)) and atomically adds the summed loss
 * (already divided by N*d) into loss_out[0] (caller zeroes it). */
int gadapt_mesh_loss_seed(const float* x_top, const float* target, float* x_phys, float* g_top,
                          float* loss_out, int64_t n_nodes, int d, int c, int l1, float grad_scale,
                          void* stream);

/* The training loop's loss, F.mse_loss / F.l1_loss(out, data.x_phys) with reduction 'mean'
 * (run_GNN.py:80-84,106), forward AND its derivative in one launch: pred [N,d] with a row stride of
 * pred_stride floats (the x[:, :dim] view of GNN.py:299 is not dense), target [N,d] dense;
 * loss_out[0] = the loss (fixed summation order), seed [N,d] = d loss / d pred.
 * scratch: gadapt_loss_scratch_floats() floats, zero-filled once by the caller; left zero-filled. */
int gadapt_loss_forward(const float* pred, int64_t pred_stride, const float* target, int64_t n_rows, int d,
                        int l1, float* seed, float* loss_out, float* scratch, void* stream);
int gadapt_loss_scratch_floats(void);

/* Batch assembly on the device (what PyG's DataLoader collation does on the host for the node fields, run_GNN.py:72-76, for a
 * dataset whose samples share one mesh): for up to GADAPT_GATHER_MAX fields k, dst[k][b, :] = src[k][idx[b], :] for b < n_take, rows
 * of row_floats[k] floats (x_comp, x_phys, f_tensor, uu_tensor ... stacked per sample).  idx: int64 on the device.  src / dst /
 * row_floats are host arrays of n_fields entries (device pointers inside).  One launch per batch. */
#define GADAPT_GATHER_MAX 8
int gadapt_gather_fields(int n_fields, const float* const* src, float* const* dst, const int64_t* row_floats, const int64_t* idx,
                         int n_take, void* stream);

/* Backward of the slice x_phys = x_top[:, :d] (GNN.py:299): g_top [N,C] = g_phys [N,d] zero-padded. */
int gadapt_pad_columns(const float* g_phys, float* g_top, int64_t n_nodes, int d, int c, void* stream);

/* ------------------------------------------------------------------ optimizer
 * torch.optim.Adam(lr, weight_decay) step on a flat fp32 bucket (run_GNN.py:88,128-131):
 * L2 weight decay added to the gradient, bias-corrected moments, eps outside the sqrt.
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce SUM). */
int gadapt_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, float grad_scale, void* stream);

/* Same step with the step count on the device: state = int32[2] {steps taken so far, 0}, zero-filled once by the
 * caller and advanced by the launch itself - nothing in the argument list changes from step to step, so the launch can be
 * part of a captured hipGraph together with forward and backward. */
int gadapt_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                         int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int32_t* state, float grad_scale, void* stream);

/* ------------------------------------------------------------------ data-parallel gradient exchange
 * SUM (average = 0) or AVERAGE (average = 1) of a flat fp32 bucket over the ranks of an RCCL communicator, in place, enqueued
 * on `stream` (capturable like any RCCL collective).  Replaces what `torch.distributed` / DDP would do for the reference's
 * `loss.backward()` under data parallelism (src/run_GNN.py:106-131 runs single-process; SURVEY.md section 8(b) item 6 and 8(e): one
 * all-reduce of the 2(C^2+C)-float bucket per step).  `comm` is the CALLER's ncclComm_t (ncclCommInitRank, or the one a
 * framework hands out); the library owns no communicator and links no RCCL: it resolves ncclAllReduce from the librccl.so
 * already loaded in the process (or loads it), GADAPT_E_RUNTIME if there is none.  The Python mirror keeps using
 * torch.distributed (its process group owns the communicator); this entry point is for callers that hold their own. */
int gadapt_allreduce_flat(void* comm, float* bucket, int64_t n, int average, void* stream);

/* ------------------------------------------------------------------ timing (bench only)
 * When enabled, every launch of the three hot kernels is bracketed by hipEvents on its
 * stream; kernel_id 0 = forward, 1 = backward target pass, 2 = backward source pass.
 * gadapt_profile_read synchronises on the recorded events (host helper, not capturable). */
int gadapt_profile_enable(int on);
int gadapt_profile_read(int kernel_id, double* total_ms, int* count);
int gadapt_profile_samples(int kernel_id, double* out_ms, int cap);   /* returns the number written */
/* Variant of each recorded launch, aligned with gadapt_profile_samples: bit 0 = compact upstream gradient (g_top_cols),
 * bit 1 = compact layer input (x0_cols), bit 2 = head-only output (x_top4). */
int gadapt_profile_variants(int kernel_id, int* out, int cap);
int gadapt_profile_reset(void);
/* Dispatch share D of an event pair: n x one empty launch (kernel id 3, p1) and n x two empty launches (id 4, p2),
 * bracketed like the hot kernels; D = 2 p1 - p2. */
int gadapt_profile_calibrate(int n, void* stream);
/* Switch: 1 (the default; GADAPT_BWD_INPLACE=0 in the environment starts with 0) lets gadapt_block_backward keep a dense
 * layer's dxd rows in the buffer its source pass then writes g_out to - every lane group reads its dxd row before it stores the
 * same row of g_out - so a layer pair touches two [N,C] work buffers instead of three (dxd_ws then only serves layer 0 and the
 * 4-column pair).  Results are bit-identical either way (tests/test_gpu_ops.py::test_block_backward_inplace_is_bit_identical). */
int gadapt_debug_set_backward_inplace(int on);
/* Diagnostic: runtime-reported workgroups per CU of {forward, backward target, backward source}. */
int gadapt_debug_occupancy(int c, int* out3);

#ifdef __cplusplus
}
#endif
#endif /* GADAPT_HIP_H */
