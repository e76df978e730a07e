/* a3vt.h — C ABI of the MI355X (gfx950) mesh-reconstruction hot path.
 *
 * Drop-in boundary for the reconstruction path of facebookresearch/Active-3D-Vision-and-Touch
 * (`pterotactyl`).  The reference has no native/FFI layer: its boundary is the
 * Python module API of pterotactyl.reconstruction.vision.{model,train} and pterotactyl.utility.utils,
 * which reach third-party kernels through torch / PyTorch3D.  Every entry point below names the
 * reference call site (file:line under /root/reference) whose arithmetic it replaces; the Python
 * façade in active-3d-vision-and-touch_amd/pterotactyl/ binds them with ctypes.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  Return 0 = OK, <0 = error
 *    (a3vt_last_error() gives a thread-local message).
 *  - All data pointers are caller-owned DEVICE pointers unless a parameter says "host".
 *    The library never allocates or frees user tensors and never synchronises the host;
 *    scratch memory comes from the caller (query with the *_bytes functions).
 *  - `stream` is a hipStream_t (NULL = default stream).  Everything is fp32; indices are int32.
 *  - Dense row-major layouts.  M = batch * n_vert rows of per-vertex data, sample-major.
 */
#ifndef A3VT_H
#define A3VT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3VT_VERSION 163 /* 163: a3vt_adam_step (the trainer's Adam step over all parameter tensors in one launch); 162: a3vt_conv5_weight_grad (the weight gradients of the image pyramid's 5 x 5 layers: fixed-order sums, no fill / cast launches); 161: a3vt_bnrelu_fwd / _bwd (training BatchNorm2d + ReLU of the image pyramid on channels-last bf16 maps), a3vt_cast_weights_bf16, a3vt_image_pool_fwd_add, a3vt_conv5_nhwc; 160: a3vt_adj_split, a3vt_adj_split_validate, a3vt_gcn_stack_fwd_adj / _bwd_adj (the fused vision + touch adjacency as P + a complete bipartite block); 150: a3vt_dbg_path_counts, larger a3vt_chamfer_workspace_bytes (oriented boxes of the pruned search), a3vt_gcn_stack_scratch_bytes covers every gemm mode, gemm_bf16 outside 0..3 refused; 140: gemm_bf16 = 3 ("fp32x3": fp32 products as six bf16 MFMA passes on exactly split operands), a3vt_split3_bf16; 0.2.x: bf16 storage mode (gemm_bf16 = 2), a3vt_gcn_stack_stash_bytes / _scratch_bytes_mode, deterministic backward scatters; 120: a3vt_chamfer_fwd_ws (pruned exact search); 130: a3vt_posenc_wide_* (448-wide vertex-feature encoder), larger a3vt_gcn_stack_mask_bytes / _scratch_bytes (channel-sliced aggregation) */

int a3vt_version(void);
const char *a3vt_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Adjacency.  Replaces the dense (N,N) float adjacency of utility/utils.py:47-52,134-148 with CSR
 * (row-normalised, self loops included).  Host-side structural check: monotone rowptr, columns in
 * range.  Pointers are HOST pointers.  */
int a3vt_csr_validate(const int32_t *rowptr_host, const int32_t *col_host, int n_vert, int nnz);

/* ---------------------------------------------------------------------------------------------
 * GCN.  Replaces GCN_layer.forward, reconstruction/vision/model.py:351-363
 *   Z = X W ;  Y[:, :c] = A Z[:, :c] + b[:c] ;  Y[:, c:] = Z[:, c:] ;  ReLU     (hidden layers, do_cut)
 *   Y = A (X W) + b                                                              (last layer, 3 channels)
 * and GCN.forward, model.py:316-331 (the layer loop; the NaN trap at :326 becomes a3vt_check_finite).
 *
 * Layer dims follow model.py:297-301: [in_features, hidden x (L-1), 3].  `weights[i]` is the
 * reference parameter layout (1, in_i, out_i) = row-major [in_i][out_i]; `biases[i]` is [out_i].
 * `weights`, `biases`, `grad_weights`, `grad_biases` are HOST arrays of L DEVICE pointers.
 * `cut_len` = round(hidden * cut) (model.py:355; 99 for 300 * 0.33).
 * csr_* is A (row-normalised), csrT_* is its transpose (A is not symmetric after normalisation).
 * gemm_bf16 (every GCN entry point and a3vt_rowgemm; 2 = bf16 storage, stack entry points only, see below): 0 = the
 * per-vertex products run as exact fp32 MFMA (the parity
 * mode, BASELINE configs[1]/[2]); 1 = "bf16 + MFMA feature MLP" (configs[3]/[4]): the operands of X W, dZ W^T and
 * X^T dZ are rounded to bf16 on their way into the matrix pipe (v_mfma_f32_16x16x16_bf16, fp32 accumulation); every
 * stored tensor stays fp32.  Vertex positions then agree with the fp32 reference to ~2e-3 relative instead of 3e-7.
 * 3 = "fp32x3" (stack entry points only; csrc/gcn_gemm3.hip): every stored tensor stays fp32 exactly as in mode 0, and
 * the three products of the hidden layers (K, n in (288, 304], >= 12 288 rows) are formed on the bf16 matrix pipe from
 * operands split EXACTLY into three bf16 pieces (x == hi + mid + lo, a3vt_split3_bf16), the six partial products down
 * to 2^-16 of the largest kept, fp32 accumulation: fp32-level error (measured against the fp64 oracle: DESIGN.md), not
 * bit-identical to mode 0; all other shapes of such a call run the mode-0 kernels — and so does the WHOLE stack when its
 * ReLU-sign rows are longer than 128 bytes (pad4(cut_len)/4 + ceil(hidden/4) > 128, i.e. cut_len > 212 at hidden 300): a
 * shape that works in mode 0 works in mode 3.  Mode 3 keeps three bf16 weight images per hidden layer in `scratch`, a
 * larger slot than mode 0's: size the scratch with a3vt_gcn_stack_scratch_bytes (enough for every mode) or with
 * a3vt_gcn_stack_scratch_bytes_mode(..., 3).  Values outside 0..3 are refused (A3VT error).  Mode 0 stays the default.
 * csr_max_degree / csrT_max_degree: the largest number of entries in a row of that matrix, or 0 if unknown.  The
 * fused vision + touch graphs have hub rows (chart centres linked to every seam vertex, ~1150 entries,
 * utility/utils.py:119-128); rows above 64 entries are aggregated by a whole workgroup in a second launch, which a
 * known small maximum lets the library skip.  Results do not depend on the value.  With every row <= 8 entries (the vision
 * templates), fp32 storage, hidden 273-304, <= 3072 vertices per mesh and >= 12 288 rows the forward uses the channel-sliced
 * aggregation (csrc/gcn_csrq.hip) and leaves hybrid rows in the stash; the library remembers which layout it wrote into a
 * stash (keyed by the `masks` pointer) and the backward call that receives that stash follows it, whatever its own hint.
 *
 * feats  : [M][ld_feats] with ld_feats == in_features rounded up to a multiple of 4; pad columns must be zero.
 * acts   : saved inputs of layers 1..L-1, (L-1) * M * hidden floats (needed by the backward pass).  OPAQUE to the caller:
 *          row-major [L-1][M][hidden] on the half-wave aggregation path, "hybrid" rows on the channel-sliced one (per layer
 *          columns [0,160) quad-major [batch][40][n_vert] float4, then columns [160,hidden) row-major) — same size either way
 * masks  : ReLU sign bytes of those activations (1 byte per 4 channels), a3vt_gcn_stack_mask_bytes() bytes (opaque);
 *          the backward pass reads these 16 MB per layer instead of re-reading the 197 MB activation
 * update : [M][3]
 * Forward-only callers (policy scoring, environment.py:221-257) may pass acts = masks = NULL: the
 * layer outputs then ping-pong inside `scratch`.
 * a3vt_gcn_stack_scratch_bytes: enough for ANY gemm_bf16 (the maximum over the modes); a3vt_gcn_stack_scratch_bytes_mode
 * (below) is the exact figure of one mode.  The stack entry points take no scratch size: an undersized scratch is
 * undefined behaviour, so size it with one of the two. */
size_t a3vt_gcn_stack_scratch_bytes(int batch, int n_vert, int in_features, int hidden, int num_layers,
                                    int cut_len, int need_backward);
size_t a3vt_gcn_stack_mask_bytes(int batch, int n_vert, int hidden, int num_layers, int cut_len);
/* gemm_bf16 == 2, "bf16 storage" (BASELINE configs[3]/[4]): activations, their gradients and the weight images are kept
 * as bf16 in HBM (fp32 accumulation; fp32 weights, gradients of the weights, features and update at this boundary).
 * `acts` then holds bf16 rows of pad8(hidden) elements — and, behind them, the stack's input rows in bf16, which the backward
 * reads instead of converting `feats` again — and the sign bytes use another row length: query both sizes
 * with a3vt_gcn_stack_stash_bytes, and the scratch size with a3vt_gcn_stack_scratch_bytes_mode (or the all-modes
 * maximum a3vt_gcn_stack_scratch_bytes).  Needs >= 2 layers. */
size_t a3vt_gcn_stack_scratch_bytes_mode(int batch, int n_vert, int in_features, int hidden, int num_layers,
                                         int cut_len, int need_backward, int gemm_bf16);
int a3vt_gcn_stack_stash_bytes(int batch, int n_vert, int hidden, int num_layers, int cut_len, int gemm_bf16,
                               size_t *acts_bytes, size_t *mask_bytes);

int a3vt_gcn_stack_fwd(const float *feats, int ld_feats, int in_features,
                       const float *const *weights, const float *const *biases,
                       int num_layers, int hidden, int cut_len,
                       const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val, int csr_max_degree,
                       int n_vert, int batch, int gemm_bf16,
                       void *acts, uint8_t *masks, float *scratch, float *update, void *stream);

/* Backward of the stack.  grad_update [M][3] -> grad_feats [M][ld_feats] (pad columns written as 0),
 * grad_weights[i] [in_i][out_i], grad_biases[i] [out_i] (overwritten; channels >= cut_len of hidden
 * biases get exact zeros: they are dead parameters in the reference, model.py:358). */
int a3vt_gcn_stack_bwd(const float *feats, int ld_feats, int in_features,
                       const float *const *weights, const float *const *biases,
                       int num_layers, int hidden, int cut_len,
                       const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val,
                       const int32_t *csrT_rowptr, const int32_t *csrT_col, const float *csrT_val, int csrT_max_degree,
                       int n_vert, int batch, int gemm_bf16,
                       const void *acts, const uint8_t *masks, const float *grad_update,
                       float *const *grad_weights, float *const *grad_biases, float *grad_feats,
                       float *scratch, void *stream);

/* The same with `accumulate` != 0: the weight and bias gradients are ADDED to what grad_weights[i] / grad_biases[i] hold
 * (each sum is formed first, in the usual fixed order, then added once) — for trainers that let the library write every
 * gradient where it lives (a flat all-reduce bucket) and whose stages share weights (mesh_deform_2, model.py:268,281:
 * the first backward call of a step overwrites, the second accumulates).  grad_feats is always overwritten. */
int a3vt_gcn_stack_bwd_acc(const float *feats, int ld_feats, int in_features,
                           const float *const *weights, const float *const *biases,
                           int num_layers, int hidden, int cut_len,
                           const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val,
                           const int32_t *csrT_rowptr, const int32_t *csrT_col, const float *csrT_val, int csrT_max_degree,
                           int n_vert, int batch, int gemm_bf16,
                           const void *acts, const uint8_t *masks, const float *grad_update,
                           float *const *grad_weights, float *const *grad_biases, float *grad_feats,
                           float *scratch, int accumulate, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Structured adjacency (round 6).  The fused vision + touch matrix of utility/utils.py:75-130 links EVERY seam vertex of the
 * chart atlas (:80-84,119-123) to EVERY touch-chart centre (:95-98,124-128), so the matrix of model.py:356,360 is
 *     A^ = D^-1 (P + J):  P a sparse symmetric 0/1 pattern (<= 10 entries per row on the reference's graphs),
 *                         J the COMPLETE bipartite block S x C and C x S (S = seam vertices, C = chart centres, disjoint),
 *                         D = the row sums of P + J   (every non-zero of row i of A^ is 1 / d_i).
 * (A^ Z)_i = (sum_{j in P(i)} Z_j + [i in S] sum_{c in C} Z_c + [i in C] sum_{s in S} Z_s) / d_i: two sums per mesh replace
 * the ~1150-entry centre rows and the 26-30-entry seam rows (75 % of the non-zeros of the 20-chart graph).  A caller that
 * knows the decomposition passes it NEXT TO the full CSR (which the output layer and every fallback path keep using):
 * the hidden layers of a3vt_gcn_stack_fwd_adj / _bwd_adj then aggregate with it wherever the shape allows (fp32 storage: the
 * channel-sliced kernels of csrc/gcn_csrqs.hip under the conditions listed above with max_degree <= 12; bf16 storage: the
 * row walk over P plus the two sums).  Same values up to fp32 rounding (another association of the same sums), not bit for
 * bit.  All pointers DEVICE pointers; the struct itself lives on the host.  Vision-only templates are the special case
 * n_seam = n_centre = 0. */
typedef struct a3vt_adj_split {
  const int32_t *rowptr; /* [n_vert + 1]  CSR of P */
  const int32_t *col;    /* [rowptr[n_vert]]  ascending within a row */
  const float *scale;    /* [n_vert]  1 / d_i = the value of every non-zero of row i of the full matrix */
  const uint8_t *cls;    /* [n_vert]  0 = neither, 1 = in S, 2 = in C */
  int32_t max_degree;    /* longest row of P */
  int32_t n_seam, n_centre;
} a3vt_adj_split;

/* Host-side proof that (P, S, C, scale) IS the matrix (csr_rowptr, csr_col, csr_val): for every row the sorted union of its
 * P entries and of C (rows in S) or S (rows in C) equals the row's columns, every value equals scale[row] bit for bit, P is
 * symmetric, S and C are disjoint and J is not already part of P.  All pointers HOST pointers.  0 = proven, -1 = refuted
 * (a3vt_last_error says where). */
int a3vt_adj_split_validate(const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val, int n_vert,
                            const int32_t *p_rowptr, const int32_t *p_col, const float *scale, const uint8_t *cls);

/* a3vt_gcn_stack_fwd / a3vt_gcn_stack_bwd_acc with the decomposition (`split` may be NULL: exactly the calls above).  The
 * backward must receive the split the forward received (the stash layout follows the forward's choice). */
int a3vt_gcn_stack_fwd_adj(const float *feats, int ld_feats, int in_features,
                           const float *const *weights, const float *const *biases,
                           int num_layers, int hidden, int cut_len,
                           const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val, int csr_max_degree,
                           const a3vt_adj_split *split,
                           int n_vert, int batch, int gemm_bf16,
                           void *acts, uint8_t *masks, float *scratch, float *update, void *stream);
int a3vt_gcn_stack_bwd_adj(const float *feats, int ld_feats, int in_features,
                           const float *const *weights, const float *const *biases,
                           int num_layers, int hidden, int cut_len,
                           const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val,
                           const int32_t *csrT_rowptr, const int32_t *csrT_col, const float *csrT_val, int csrT_max_degree,
                           const a3vt_adj_split *split,
                           int n_vert, int batch, int gemm_bf16,
                           const void *acts, const uint8_t *masks, const float *grad_update,
                           float *const *grad_weights, float *const *grad_biases, float *grad_feats,
                           float *scratch, int accumulate, void *stream);

/* One GCN layer on its own — GCN_layer.forward(features, adj, activation), model.py:351-363, for callers
 * that run their own layer loop (the copies in reconstruction/autoencoder/model.py:96-137 and
 * policies/DDQN/model.py:132-168 have layer shapes the stack entry points do not cover:
 * hidden -> hidden without the cut, hidden_dim -> num_actions).
 *   Z = X W ;  Y[:, :c] = A Z[:, :c] + b[:c] ;  Y[:, c:] = Z[:, c:] ;  Y = relu ? max(Y, 0) : Y
 * c = cut_len = round(out * cut) when the layer cuts (do_cut), c = out_features otherwise (all channels
 * aggregated, bias on all).  in_features <= ld_x <= 600, out_features <= 304.
 * x [M][ld_x] (ld_x % 4 == 0, pad columns zero), weight [in_features][out_features] (reference layout
 * (1,in,out)), bias [out_features], y [M][ld_y] (ld_y % 4 == 0, ld_y >= out_features; pad columns untouched).
 * Backward: grad_y [M][ld_gy] and the forward output y (the ReLU mask is y > 0) -> grad_x [M][ld_x] (pad
 * columns zero), grad_weight [in][out], grad_bias [out] (all overwritten; bias channels >= c get zeros). */
size_t a3vt_gcn_layer_scratch_bytes(int batch, int n_vert, int ld_x, int out_features, int cut_len,
                                    int need_backward);
int a3vt_gcn_layer_fwd(const float *x, int ld_x, int in_features, const float *weight, const float *bias,
                       int out_features, int cut_len, int relu,
                       const int32_t *csr_rowptr, const int32_t *csr_col, const float *csr_val, int csr_max_degree,
                       int n_vert, int batch, int gemm_bf16, float *y, int ld_y, float *scratch, void *stream);
int a3vt_gcn_layer_bwd(const float *x, int ld_x, int in_features, const float *weight,
                       int out_features, int cut_len, int relu,
                       const int32_t *csrT_rowptr, const int32_t *csrT_col, const float *csrT_val, int csrT_max_degree,
                       int n_vert, int batch, int gemm_bf16, const float *y, int ld_y, const float *grad_y, int ld_gy,
                       float *grad_weight, float *grad_bias, float *grad_x, float *scratch, void *stream);

/* The dense per-vertex product alone (torch.matmul(features, self.weight), model.py:352) on the
 * fp32 MFMA path: C[M][n_out] = A[M][k] * W[k][n_out], k % 4 == 0, n_out <= 304.  `wt` is W transposed
 * and zero padded to [a3vt_wt_rows(n_out)][a3vt_wt_ld(k)] floats (a3vt_transpose_weight builds it).
 * Used by tests and by bench.py to time the dominant kernel in isolation. */
int a3vt_wt_rows(int n_out);
int a3vt_wt_ld(int k);
int a3vt_transpose_weight(const float *w, int k, int n_out, float *wt, void *stream);
int a3vt_rowgemm(const float *a, int lda, int m, int k, const float *wt, int n_out, int gemm_bf16,
                 float *c, int ldc, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Vertex features.  Replaces Positional_Encoder.forward (model.py:381-399) + Mask_Encoder.forward
 * (model.py:410-414) + their sum (model.py:232,240,262,275):
 *   feats = MLP(nerf_embedding(p) ++ p) + Embedding[mask]
 * pe_params: packed fp32 [W1 (I/4 x 63), b1, W2 (I/2 x I/4), b2, W3 (I x I/2), b3, E (4 x I)],
 * each in torch (out,in) row-major layout (state-dict order).  Only I = 50 (the image-free model,
 * model.py:193) is supported by the fused kernel.  feats is [M][ld_feats], pad columns zeroed. */
size_t a3vt_posenc_param_count(int input_size);
size_t a3vt_posenc_scratch_bytes(int m, int input_size);
int a3vt_posenc_mask_fwd(const float *verts, const float *mask, int m, int input_size,
                         const float *pe_params, float *feats, int ld_feats, void *stream);
/* grad_params has a3vt_posenc_param_count floats (overwritten); grad_verts [M][3] (overwritten). */
int a3vt_posenc_mask_bwd(const float *verts, const float *mask, int m, int input_size,
                         const float *pe_params, const float *grad_feats, int ld_feats,
                         float *grad_verts, float *grad_params, float *scratch, void *stream);

/* The same encoder for wide inputs — input_size = 448 of the image models (model.py:180-190: 64 + 128 + 256 pooled channels)
 * — whose 133 k parameters do not fit the LDS-resident kernel above: every layer runs as one product on the fp32 matrix
 * pipe (csrc/posenc_wide.hip).  a3vt_posenc_wide_supported: input_size % 8 == 0, 16 <= input_size <= 630.  ld_feats must
 * equal input_size.  `acts` (a3vt_posenc_wide_acts_bytes) receives the layer activations in the forward and is read by the
 * backward — the caller keeps it between the two calls; `scratch` (a3vt_posenc_wide_scratch_bytes) is free afterwards.
 * pe_params / grad_params: the packing of a3vt_posenc_mask_fwd.  Replaces the same reference lines.
 * gemm_bf16: 0 = exact fp32 products (the default path), 1 = the products after the embedding layer, forward and
 * backward, take bf16-rounded operands (fp32 storage and accumulation; the bf16 configurations' knob). */
int a3vt_posenc_wide_supported(int input_size);
size_t a3vt_posenc_wide_acts_bytes(int m, int input_size);
size_t a3vt_posenc_wide_scratch_bytes(int m, int input_size, int need_backward);
int a3vt_posenc_wide_fwd(const float *verts, const float *mask, int m, int input_size, const float *pe_params,
                         float *feats, int ld_feats, float *acts, float *scratch, int gemm_bf16, void *stream);
int a3vt_posenc_wide_bwd(const float *verts, const float *mask, int m, int input_size, const float *pe_params,
                         const float *grad_feats, int ld_feats, const float *acts, float *grad_verts,
                         float *grad_params, float *scratch, int gemm_bf16, void *stream);

/* Per-vertex image features.  Replaces Image_Encoder.pooling (model.py:70-103): project the vertices with the fixed
 * camera matrix `proj` = K.RT (row-major 3 x 4, model.py:50-67; HOST pointer), z == 0 -> 0.1,
 * xs = P1/P2/256, ys = P0/P2/256, inf -> 0.5, then grid_sample(bilinear, zeros, align_corners=True) of every map at
 * (2 ys - 1, 2 xs - 1) and concatenation:  feats[b][v][off_k + c] for map k, channel c.
 * maps[k] is CHANNELS-LAST, [B][H_k][W_k][C_k] (torch channels_last memory of a (B,C,H,W) tensor), C_k % 4 == 0;
 * `maps`, `chans`, `heights`, `widths`, `grad_maps` are HOST arrays of n_maps (<= 4) entries.
 * feats / grad_feats are [B*N][ld_feats], ld_feats % 4 == 0, ld_feats >= sum C_k (other columns untouched).
 * Backward overwrites grad_maps[k] (same layout as maps[k]) and grad_verts [B*N][3] (the reference's autograd
 * reaches the vertex positions through the sampling grid; its in-place patches cut the gradient where they fire). */
int a3vt_image_pool_fwd(const float *verts, int batch, int n_vert, const float *proj_host, int n_maps,
                        const float *const *maps, const int *chans, const int *heights, const int *widths,
                        float *feats, int ld_feats, void *stream);
/* The same with the sum of model.py:243,265,277 folded in: feats = base + pooled, `base` = the positional + mask features
 * [B*N][ld_feats] (a3vt_posenc_mask_fwd); ld_feats must equal sum C_k.  One pass instead of the pooling's store plus a
 * separate 3 x 220 MB element-wise add per refinement stage of configs[3].  feats may not alias base. */
int a3vt_image_pool_fwd_add(const float *verts, int batch, int n_vert, const float *proj_host, int n_maps,
                            const float *const *maps, const int *chans, const int *heights, const int *widths,
                            const float *base, float *feats, int ld_feats, void *stream);
int a3vt_image_pool_bwd(const float *verts, int batch, int n_vert, const float *proj_host, int n_maps,
                        const float *const *maps, const int *chans, const int *heights, const int *widths,
                        const float *grad_feats, int ld_feats, float *const *grad_maps, float *grad_verts,
                        void *stream);

/* Bias gradient of a convolution whose output gradient is stored channels-last (the bias gradients of the nn.Conv2d
 * layers of the image pyramid, model.py:15-47, in the channels-last bf16 branch):
 *   out[c] = sum over rows of grad[row][c],  grad = [rows = N*H*W][channels] contiguous, fp32 (bf16 == 0) or bf16 (== 1),
 * 16-byte aligned; out fp32 [channels], overwritten.  Fixed summation order (repeatable bit for bit).
 * scratch: a3vt_bias_grad_scratch_bytes(rows, channels) bytes (0 = unsupported channel count). */
size_t a3vt_bias_grad_scratch_bytes(long long rows, int channels);
int a3vt_bias_grad_nhwc(const void *grad, int bf16, long long rows, int channels, float *out, void *scratch,
                        size_t scratch_bytes, void *stream);

/* Training-mode BatchNorm2d + ReLU of a `CNN_layer` (model.py:15-23: BatchNorm2d -> ReLU -> Conv2d; 13 per encoder and
 * step, model.py:147-164) on a channels-last bf16 map, three launches each way instead of 3 + 1 (+ the counter increment):
 *   y = relu(gamma * (x - mean_c) / sqrt(var_c + eps) + beta),  statistics over the rows per channel (biased variance);
 *   running_mean / running_var (unbiased variance, `momentum`) and *num_batches_tracked (+1) are updated as nn.BatchNorm2d
 *   does (pass NULL to skip either).  pre_bias (optional, fp32 [channels]): a per-channel constant the producer of x left
 *   out — the bias of the Conv2d in front: batch statistics remove any such shift, so y is the same without the 28 bias-add
 *   launches of a step, and only running_mean takes it (mean + pre_bias).  x, y, dy, dx: [rows = N*H*W][channels] bf16, 16-byte aligned; gamma, beta, dgamma,
 *   dbeta: fp32 [channels]; save: fp32 [4][channels] written by the forward (mean, 1/std, scale, shift), read by the backward.
 * Backward: dgamma = sum g * xhat, dbeta = sum g, dx = scale * (g - dbeta / rows - xhat * dgamma / rows), g = dy where y > 0
 * (the mask is recomputed from x with the forward's own coefficients).  dx_colsum (optional, fp32 [channels]): the column sums
 * of dx as stored in bf16 — when x is the output of a Conv2d this is that layer's bias gradient, which a3vt_bias_grad_nhwc would
 * otherwise compute by reading dx again.  Fixed summation order, float64 final sums: repeatable
 * bit for bit.  rows >= 2.  scratch: a3vt_bnrelu_scratch_bytes(channels) bytes; one buffer may serve every layer of a stream. */
size_t a3vt_bnrelu_scratch_bytes(int channels);
int a3vt_bnrelu_fwd(const void *x, long long rows, int channels, const float *gamma, const float *beta, const float *pre_bias,
                    float eps, float momentum, float *running_mean, float *running_var, long long *num_batches_tracked, void *y,
                    float *save, void *scratch, size_t scratch_bytes, void *stream);
int a3vt_bnrelu_bwd(const void *dy, const void *x, long long rows, int channels, const float *save, void *dx, float *dgamma,
                    float *dbeta, float *dx_colsum, void *scratch, size_t scratch_bytes, void *stream);

/* bf16 copies of the pyramid's fp32 convolution weights and biases (the nn.Conv2d parameters of model.py:15-47) for MIOpen's
 * NHWC bf16 kernels, all in one launch per optimizer step instead of two or three per tensor: tensor k is
 * src[k] fp32 [outer[k]][inner[k]][hw[k]] (a weight: O x I x (KH KW), contiguous; a bias: inner = hw = 1) and becomes
 * dst[k] bf16 [outer][hw][inner] (channels-last), round to nearest even.  n <= 96; the five arrays are HOST arrays of n
 * entries holding device pointers / sizes. */
int a3vt_cast_weights_bf16(int n, const float *const *src, void *const *dst, const long long *outer, const int *inner,
                           const int *hw, void *stream);

/* The 5 x 5 convolutions of the image pyramid's small-channel layers (`CNN_layer`'s nn.Conv2d(kernel_size = 5, padding = 1),
 * model.py:15-47) on channels-last bf16 maps, which MIOpen runs at a tenth of the rate their bytes allow.  Shapes taken
 * (a3vt_conv5_supported): (cin, cout, stride) = (16, 16, 1), (32, 32, 1) — layers 2-3 and 5-6 of Image_Encoder —, (16, 32, 2),
 * layer 4, and the 3-channel layers 0 and 1: (3, 3, 1) and (3, 16, 2), forward only (flip = 0).   y[b][oy][ox][co] = bias[co] + sum over (ky, kx, ci) of x[b][oy stride + ky - pad][ox stride + kx - pad][ci] * Wm[co][ky][kx][ci]
 * (pixels outside the map are zeros); x: [batch][height][width][cin] bf16, y: [batch][Ho][Wo][cout] bf16 with
 * Ho = (height + 2 pad - 5) / stride + 1; fp32 accumulation, one rounding.  `image` (a3vt_conv5_image_bytes bytes, 16-byte
 * aligned) is written by a3vt_conv5_weight_image from the fp32 weight [cout][cin][5][5] (OIHW): flip = 0 with pad = 1 is the
 * layer's forward; for the stride-1 shapes flip = 1 (Wm transposed and flipped: call the convolution with cin and cout swapped)
 * with pad = 3 applied to the OUTPUT gradient is the gradient with respect to the layer's input.  bias: fp32 [cout] or NULL.
 * (Weight gradients, and the input gradient of the stride-2 layer, stay on MIOpen.) */
int a3vt_conv5_supported(int cin, int cout, int stride);
size_t a3vt_conv5_image_bytes(int cout, int cin, int flip);
int a3vt_conv5_weight_image(const float *weight, int cout, int cin, int flip, void *image, void *stream);
int a3vt_conv5_nhwc(const void *x, int batch, int height, int width, int cin, int cout, int stride, int pad, const void *image,
                    const float *bias, void *y, void *stream);

/* The input gradient of the pyramid's layer 1 (nn.Conv2d(3, 16, 5, stride 2, padding 1)): grad_out [batch][out_height][out_width][16]
 * bf16 -> grad_in [batch][2 out_height + 2][2 out_width + 2][3] bf16 (= the layer's input size when (H + 2 - 5) is odd, as for the
 * 254-pixel map), computed as the stride-1 convolution of a3vt_conv5_nhwc on grad_out read as if upsampled by two with zeros;
 * `image` = a3vt_conv5_weight_image(weight [16][3][5][5], cout 16, cin 3, flip 1). */
int a3vt_conv5_input_grad_3x16s2(const void *grad_out, int batch, int out_height, int out_width, const void *image, void *grad_in,
                                 void *stream);

/* The weight gradient of the same five layers ((cin, cout, stride) = (3, 3, 1), (3, 16, 2), (16, 16, 1), (32, 32, 1), (16, 32, 2);
 * padding 1) — autograd's third product of nn.Conv2d at vision/model.py:15-23:
 *   grad_weight[co][ci][ky][kx] = sum over (b, oy, ox) of grad_out[b][oy][ox][co] * x[b][oy stride + ky - 1][ox stride + kx - 1][ci],
 * x: [batch][height][width][cin] bf16 (the layer's input), grad_out: [batch][Ho][Wo][cout] bf16, grad_weight: fp32 [cout][cin][5][5]
 * (OIHW), overwritten.  fp32 accumulation on the matrix pipe, partial sums per workgroup added in a fixed order (repeatable bit
 * for bit; MIOpen's kernel accumulates with atomics into an fp32 workspace it first fills and then casts).  scratch:
 * a3vt_conv5_wrw_scratch_bytes(cin, cout) bytes, 16-byte aligned; x / grad_out 16-byte aligned for 16 / 32 channels, 2-byte for 3. */
size_t a3vt_conv5_wrw_scratch_bytes(int cin, int cout);
int a3vt_conv5_weight_grad(const void *x, const void *grad_out, int batch, int height, int width, int cin, int cout, int stride,
                           float *grad_weight, void *scratch, size_t scratch_bytes, void *stream);

/* The optimizer step of the trainer, `optim.Adam(params, lr, weight_decay=0)` + `optimizer.step()` (vision/train.py:64,148), over ALL
 * parameter tensors in one launch — torch's Adam (amsgrad off, maximize off) in its own order of operations per element:
 *   g' = g + weight_decay p;  m += (g' - m)(1 - beta1);  v = v beta2 + ((1 - beta2) g') g';
 *   p -= lr / (1 - beta1^step) * (m / (sqrt(v) / sqrt(1 - beta2^step) + eps)),      step = 1 for the first call.
 * The tensors (fp32, contiguous, any alignment; 16-byte aligned ones take the vector path) are described by DEVICE tables the caller
 * builds once: param / grad / exp_avg / exp_avg_sq [tensors] pointers, numel [tensors], and a chunk list — chunk k covers elements
 * [chunk_off[k], chunk_off[k] + a3vt_adam_chunk_elems()) of tensor chunk_tensor[k] (cut off at numel); every element must be covered
 * exactly once.  Element-wise and without reductions: bit-repeatable. */
int a3vt_adam_chunk_elems(void);
int a3vt_adam_step(void *const *param, const void *const *grad, void *const *exp_avg, void *const *exp_avg_sq,
                   const long long *numel, const int *chunk_tensor, const long long *chunk_off, int n_chunks, double lr, double beta1,
                   double beta2, double eps, double weight_decay, long long step, void *stream);

/* Vertex update, model.py:250,270,283:  out[b][v] = in[b][v] + (v < n_vision ? update[b][v] : 0). */
int a3vt_vertex_update(const float *verts_in, const float *update, int batch, int n_vert, int n_vision,
                       float *verts_out, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Surface sampling.  Replaces batch_sample, utility/utils.py:152-187 (PyTorch3D
 * mesh_face_areas_normals + torch.multinomial + _rand_barycentric_coords + gathers).
 * faces [F][3] int32 is shared by the whole batch (adj_info['faces']).
 * a3vt_face_cdf: per-mesh inclusive CDF of p_f = |area_f / sum(area)| with the reference's NaN
 * scrubs (utils.py:165-168): cdf [batch][F].
 * a3vt_sample_points_fwd: `draws` independent clouds of `num` points per mesh (utils.chamfer_distance
 * draws 3, utils.py:204-217).  If face_idx_in/u_in/v_in are non-NULL ([draws][batch][num]) they are
 * used as the samples (parity mode); otherwise samples come from Philox4x32-10(seed, offset) through
 * the CDF.  Outputs: points [draws][batch][num][3] and the samples actually used (face_idx/u/v out,
 * needed by the backward pass). */
int a3vt_face_cdf(const float *verts, const int32_t *faces, int batch, int n_vert, int n_faces,
                  float *cdf, void *stream);
int a3vt_sample_points_fwd(const float *verts, const int32_t *faces, const float *cdf,
                           int batch, int n_vert, int n_faces, int draws, int num,
                           const int32_t *face_idx_in, const float *u_in, const float *v_in,
                           uint64_t seed, uint64_t offset,
                           float *points, int32_t *face_idx_out, float *u_out, float *v_out, void *stream);
/* grad_verts [batch][n_vert][3] is overwritten with the scatter-add of w_k * grad_points. */
int a3vt_sample_points_bwd(const int32_t *faces, int batch, int n_vert, int n_faces, int draws, int num,
                           const int32_t *face_idx, const float *u, const float *v,
                           const float *grad_points, float *grad_verts, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Chamfer.  Replaces pytorch3d.loss.chamfer_distance(x, y, batch_reduction=None) as called at
 * utility/utils.py:207,212 (squared-L2 K=1 nearest neighbour both ways, mean over each cloud, summed)
 * and the mean over draws at utils.py:214-215.
 * x [draws][batch][p][3] (predicted clouds), y [batch][q][3] (ground truth, shared by the draws).
 * Outputs: dist_xy/idx_xy [draws][batch][p], dist_yx/idx_yx [draws][batch][q],
 *          cd [batch] = (1/draws) sum_r ( mean_i dist_xy + mean_j dist_yx ).
 * scratch: a3vt_chamfer_scratch_bytes() bytes (8 per ground-truth point and cloud), contents irrelevant: with it both
 *          directions come out of ONE pass over the distance matrix; NULL selects the two-pass search (same results,
 *          bit for bit — every distance is the same fma chain and ties go to the lowest index either way). */
size_t a3vt_chamfer_scratch_bytes(int draws, int batch, int p, int q);
int a3vt_chamfer_fwd(const float *x, const float *y, int draws, int batch, int p, int q,
                     float *dist_xy, int32_t *idx_xy, float *dist_yx, int32_t *idx_yx, float *cd,
                     void *scratch, void *stream);
/* The same with a sized workspace and a choice of search.  a3vt_chamfer_workspace_bytes() holds every algorithm.
 * algo: 0 = automatic (the pruned search from 2048 points per cloud on, else the sweep), 1 = brute force, one launch
 *       per direction (workspace may be NULL), 2 = brute force, one sweep, 3 = pruned exact search: each cloud is
 *       sorted into blocks of 64 neighbouring points with bounding boxes, and a wave of 64 neighbouring queries only
 *       evaluates the blocks whose box can still hold a nearer (or equally near) point.
 * All algorithms return the same distances and indices bit for bit (same fma chain, ties to the lowest index).
 * <0 if the workspace is too small for the algorithm asked for (or, for the pruned search, not 16-byte aligned). */
size_t a3vt_chamfer_workspace_bytes(int draws, int batch, int p, int q);
int a3vt_chamfer_fwd_ws(const float *x, const float *y, int draws, int batch, int p, int q,
                        float *dist_xy, int32_t *idx_xy, float *dist_yx, int32_t *idx_yx, float *cd,
                        void *workspace, size_t workspace_bytes, int algo, void *stream);
/* Forward-only callers that score K candidate meshes per ground-truth cloud (policies/environment.py:174-180,252-257: the
 * greedy search evaluates up to 50 candidate touches against the same object): y holds y_batch clouds, batch is a multiple
 * of y_batch and mesh b is compared with y[b % y_batch] — the ground truth is uploaded, sorted and boxed once instead of
 * once per candidate.  Outputs as a3vt_chamfer_fwd_ws (dist_yx / idx_yx per mesh: [draws][batch][q]). */
int a3vt_chamfer_fwd_shared(const float *x, const float *y, int draws, int batch, int y_batch, int p, int q,
                            float *dist_xy, int32_t *idx_xy, float *dist_yx, int32_t *idx_yx, float *cd,
                            void *workspace, size_t workspace_bytes, int algo, void *stream);
/* grad_cd [batch].  grad_x [draws][batch][p][3] overwritten; grad_y [batch][q][3] overwritten, may be
 * NULL (the trainer's ground truth needs no gradient, vision/train.py:141-143). */
int a3vt_chamfer_bwd(const float *x, const float *y, int draws, int batch, int p, int q,
                     const int32_t *idx_xy, const int32_t *idx_yx, const float *grad_cd,
                     float *grad_x, float *grad_y, void *stream);

/* Test hook (tests/test_gpu_csr_sliced.py): which neighbour-aggregation kernels the fp32 stacks use — 0 = chosen by shape
 * (default), 1 = the half-wave-per-vertex kernels everywhere, 2 = the channel-sliced kernels wherever a mesh slice fits
 * LDS, graphs with long rows included.  In the bf16 storage mode 1 also keeps the row walk where the tiled aggregation
 * (neighbour rows from LDS, bit-identical sums) would run.  Every choice gives the same outputs; process-wide; not a tuning
 * knob.  The library reads NO environment variable. */
int a3vt_dbg_csr_algo(int algo);

/* Test hook: how many launch decisions of this process took each kernel family since the last reset — so that a parity test
 * can assert that its fixture REACHES the kernels it claims to pin (tests/test_gpu_golden.py::test_g12_*).  counts[0..n):
 *   [0] stack forward calls on the channel-sliced aggregation (hybrid rows)   [1] ... on the half-wave row-walk kernels
 *   [2] rowgemm_kernel<19,...,ADIRECT> launches (exact fp32 hidden-layer products: the headline kernel)
 *   [3] rowgemm3_kernel launches (gemm mode 3)   [4] dw3_kernel launches   [5] dw_kernel launches on quad-major operands
 *   [6] rowgemm16_kernel launches (bf16 storage)  [7] bf16-storage stack forward calls on the channel-sliced aggregation
 *   [8] rowgemmw_kernel launches  [9] dww_kernel launches (exact fp32 hidden-layer products, round 6)
 *   [10] stack forward calls that aggregated through the P + bipartite split (struct a3vt_adj_split)
 *   [11] csr16t forward launches (bf16 storage: aggregation from LDS tiles)
 * Returns the number of counters the library keeps (entries beyond it are written as 0); reset != 0 clears them. */
int a3vt_dbg_path_counts(long long *counts, int n, int reset);

/* Test hook: WORK counters of the pruned nearest-neighbour search (a3vt_chamfer_fwd_ws and friends) — values cannot show a
 * search that does too much work (round 5: pad lanes that asked about the origin cost 1.85 ms of a 1.5 ms launch for three
 * rounds, every result bit-exact).  enable != 0 clears the counters and switches them on for the searches that follow (a
 * handful of atomics per wave: not for timing), enable == 0 switches them off; out8 != NULL receives what was counted so
 * far: [0] waves of 64 queries, [1] groups of 16 candidates evaluated, [2] blocks of 64 evaluated, [3] point-to-box tests,
 * [4] the most groups any one wave evaluated, [5..7] 0.  Synchronises the device.  Off by default (the product touches no
 * counter). */
int a3vt_dbg_nn_work(int enable, unsigned long long *out8);

/* The operand split of gemm mode 3 (replaces nothing in the reference: it is how torch.matmul(features, self.weight),
 * model.py:352, is fed to the bf16 matrix pipe without losing fp32 bits).  hi / mid / lo receive bf16 bit patterns with
 * float(hi[i]) + float(mid[i]) + float(lo[i]) == x[i] exactly for every finite x[i] whose lowest set bit is >= 2^-133
 * (all normal floats down to 2^-110; FLT_MAX included: hi is a truncation and cannot overflow); smaller magnitudes
 * differ by < 2^-133.  Device pointers, n elements each. */
int a3vt_split3_bf16(const float *x, size_t n, uint16_t *hi, uint16_t *mid, uint16_t *lo, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Deferred finite check (replaces the blocking NaN trap of model.py:326-329): sets *flag (device
 * int32, caller-zeroed) to 1 if any of the n floats is NaN/Inf.  No host sync. */
int a3vt_check_finite(const float *data, size_t n, int32_t *flag, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): when enabled, every fp32-MFMA launch of the GCN stack is
 * bracketed by a HIP event pair on its launch stream.  a3vt_profile_read synchronises on those events
 * and returns, per kernel class, the summed device time and launch count since the last read:
 *   [0] forward product Z = X W (hidden layers, 300x300 and the first layer)
 *   [1] backward product dX = dZ W^T (hidden layers)     [2] backward product dW = X^T dZ
 * Only hidden x hidden launches are of the headline shape; bench.py divides by the counts it expects. */
int a3vt_profile_enable(int on);
int a3vt_profile_read(double *total_ms /*[3]*/, int *count /*[3]*/);
/* The same for n classes (round 6; every call of this library on a step, so that a bench line can say where a step's time
 * goes without a profiler): [0..2] as above, [3] neighbour aggregation of the hidden layers (forward and A^T backward),
 * [4] output layer (300 -> 3 product, its aggregation, their backward), [5] Chamfer forward (sort, boxes, exact search,
 * reduction), [6] surface sampling forward / backward + Chamfer backward, [7] vertex-feature encoders, image pooling and the
 * image pyramid's library kernels (forward and backward), [8] the optimizer step (a3vt_adam_step).  A class's time is the span from the first to the last launch of each call, summed over calls.
 * Returns the number of classes the library keeps. */
int a3vt_profile_read_classes(double *total_ms, int *count, int n);

#ifdef __cplusplus
}
#endif
#endif /* A3VT_H */
