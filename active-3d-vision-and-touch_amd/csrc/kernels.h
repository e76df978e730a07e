// kernels.h — internal launcher declarations shared by the .hip translation units and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace a3vt {

enum { EPI_PLAIN = 0, EPI_FWD_HIDDEN = 1, EPI_DX_MASK = 2 };

// C[M][*] = A[M][K] * Bt[*][K]^T.  A columns [0,ksplit) come from a0, [ksplit,K) from a1 (same column index).
struct RowGemmArgs {
  const float *a0;
  const float *a1;
  const float *bt;     // [rowgemm_bt_rows(n_store)][ldb], zero padded (rows >= n_out, cols >= k)
  const float *zeros;  // >= 16 B of zeros
  // ReLU sign bytes, one per (row, 4 columns): [Mpad32][mld]; bytes [0, moff) cover the aggregated channels
  // (written by csr_fwd), bytes [moff, mld) all channels (written by the EPI_FWD_HIDDEN epilogue for columns
  // >= csplit).  EPI_FWD_HIDDEN writes them, EPI_DX_MASK reads them (instead of re-reading the 197 MB activation).
  uint8_t *maskb;
  float *c;            // main output [M][ldc]
  float *c2;           // EPI_FWD_HIDDEN: raw output for cols < csplit, [M][ldc2]
  int lda0, lda1, ksplit, ldb;
  int bt_rows;  // rows present in the Bt buffer (set by launch_rowgemm)
  int col0;     // first output column handled by blockIdx.y == 0 (set by launch_rowgemm)
  int m, k, n_store;
  int ldc, ldc2, csplit, mld, moff;
  int no_relu;  // EPI_FWD_HIDDEN: store the pass-through channels without the ReLU (identity activation)
  int plain_relu;  // EPI_PLAIN: C = max(A Bt^T, 0) (the wide vertex-feature encoder's layers, posenc_wide.hip)
  // 0: exact fp32 MFMA.  1: fp32 storage, operands rounded to bf16 on the way into v_mfma_f32_16x16x16_bf16.
  // 2: bf16 STORAGE — a0 / a1 / bt point at bf16 rows (lda*, ldb, k, ksplit still in 4-byte units = pairs of bf16),
  //    v_mfma_f32_16x16x32_bf16; EPI_FWD_HIDDEN / EPI_DX_MASK write bf16 (c, c2 as bf16 with ldc, ldc2 in ELEMENTS, all
  //    columns < ldc written), EPI_PLAIN writes fp32.  fp32 accumulation in every mode.
  // 3: "fp32x3" (gcn_gemm3.hip) — fp32 storage as mode 0, the product as six bf16 MFMA passes on exactly split operands;
  //    bt points at the THREE bf16 images of the layer (launch_weight_images3; ldb is ignored).  Hidden-layer shapes only
  //    (rowgemm3_ok): the CALLER decides per stack (rowgemm3_stack_ok) and passes bf16 = 3 with the three images only for
  //    the layers the kernel takes; every other product of a mode-3 stack is issued as a mode-0 call with an fp32 image.
  //    A bf16 = 3 call the kernel does not take is an error (launch_rowgemm3 does not fall back: it has no fp32 image).
  int bf16;
  // rows [rem_row0, rem_row0 + rem_rows) beyond the m rows of the main loop: the few leftover tiles of the load-balanced
  // split, done by the tail of the same launch (set by launch_rowgemm; 0 = none)
  int rem_row0, rem_rows;
  // Quad-major side output (round 3, fp32 storage only; 0 = off).  With zq_nvert = vertices per mesh and zq_quads = Q,
  // c2 is the quad-major array [m / zq_nvert][Q][zq_nvert] float4 of columns [0, 4 Q), Q = pad4(csplit) / 4.  Those
  // columns then belong to the channel-sliced aggregation kernels (gcn_csrq.hip) alone and are NOT written to c, and no
  // sign byte is written / read for them:
  //   EPI_FWD_HIDDEN: raw Z (csrq_kernel<0> aggregates, adds the bias, applies the ReLU — also to the pass-through columns
  //                   [csplit, 4 Q) — and keeps the signs); c receives columns >= 4 Q only;
  //   EPI_DX_MASK   : the UNMASKED gradient columns (csrq_kernel<1> applies the signs); c receives columns >= 4 Q, masked.
  int zq_nvert, zq_quads;
  // EPI_FWD_HIDDEN with zq_nvert > 0: the activations' quad-major region may extend past the aggregated columns, up to the
  // end of the epilogue's first column group (160 columns): columns [4 zq_quads, 4 yq_quads) leave activated (ReLU, sign
  // bytes as usual) into yq, the quad-major array [m / zq_nvert][yq_quads][zq_nvert] float4 whose first zq_quads planes the
  // aggregation kernel fills; c then receives columns >= 4 yq_quads only.  0 = off.
  float *yq;
  int yq_quads;
  // Quad-major a0 (0 = row-major): a0 is [m / a0q_nvert][a0q_quads][a0q_nvert] float4 holding A columns [0, ksplit),
  // ksplit = 4 a0q_quads (the aggregation kernels' outputs: activations / dZa); lda0 is ignored.
  int a0q_nvert, a0q_quads;
  // >= 1 KiB of device memory nobody reads (rowgemmw_kernel: lanes whose columns fall behind n_store store there instead of
  // branching around the store); nullptr = the caller has none (the kernel is then not used)
  float *trash;
};
int rowgemm_bt_rows(int n_store);
int launch_rowgemm(const RowGemmArgs &a, int epi, hipStream_t s);
// gcn_gemmw.hip: the exact fp32 hidden layers on hybrid rows with the weights resident in registers (round 6)
bool rowgemmw_ok(const RowGemmArgs &a, int epi);
int launch_rowgemmw(const RowGemmArgs &a, int epi, hipStream_t s);
// gcn_gemm16.hip: the hidden layers of the bf16 storage mode with the weight image held in registers
bool rowgemm16_ok(const RowGemmArgs &a, int epi);
int launch_rowgemm16(const RowGemmArgs &a, int epi, hipStream_t s);
int launch_transpose_pad(const float *w, int k, int n, float *wt, int rows, int ld, hipStream_t s);
int launch_copy_pad(const float *w, int rows_in, int cols_in, float *out, int rows, int ld, hipStream_t s);
// Weight images of up to kMaxImages layers in one launch.  transpose = 1: dst_l [rows_l][ld_l] = W_l^T (W_l is
// [k_l][n]); transpose = 0: dst_l [rows_l][ld_l] = W_l zero padded.  dst_l = dst + l * dst_stride.
constexpr int kMaxImages = 32;
struct WeightImages {
  const float *w[kMaxImages];
  float *dst;
  size_t dst_stride;
  int k[kMaxImages], rows[kMaxImages], ld[kMaxImages];
  int n, count, transpose;
};
int launch_weight_images(const WeightImages &w, int max_rows, int max_ld, hipStream_t s);

// Bias gradients of up to kMaxImages layers in one launch: out_l[i] = sum over the nslab partial rows of layer l
// (slab + l * layer_stride + s * stride + i), i < n; entries [n, n_out) are written as zeros; accumulate as slab_reduce.
// The channel-sliced backward leaves one partial row per (layer, mesh); reduced layer by layer these were 19 launches of
// 3.8 us per stack call for 25 KB each.
struct SlabReduceBatch {
  const float *slab;
  float *out[kMaxImages];
  size_t layer_stride, stride, n, n_out;
  int count, nslab, accumulate;
};
int launch_slab_reduce_batch(const SlabReduceBatch &b, hipStream_t s);

// gcn_gemm3.hip — gemm mode 3 ("fp32x3"): the hidden-layer products as six bf16 MFMA passes on exactly split fp32 operands
constexpr int kX3ImageRows = 320;                               // Bt rows of a weight image (n <= 304, zero padded)
constexpr int kX3ImageLd = 160;                                 // 4-byte units per image row (320 bf16: K <= 320, zero padded)
constexpr int kX3ImageFloats = kX3ImageRows * kX3ImageLd;       // one image; a layer owns three (hi, mid, lo), back to back
bool rowgemm3_dims_ok(long long m, int k, int n_store);
bool rowgemm3_stack_ok(long long m, int hidden, int mld);   // what a3vt_gcn_stack_fwd / _bwd decide mode 3 with
bool rowgemm3_ok(const RowGemmArgs &a, int epi);
int launch_rowgemm3(const RowGemmArgs &a, int epi, hipStream_t s);
// dst_l + p * kX3ImageFloats (p = 0, 1, 2: hi, mid, lo) = piece p of W_l^T (transpose = 1) or W_l (transpose = 0) as
// [kX3ImageRows][2 kX3ImageLd] bf16, dst_l = w.dst + l * w.dst_stride (floats); w.rows / w.ld are ignored
int launch_weight_images3(const WeightImages &w, hipStream_t s);
// x[i] = hi[i] + mid[i] + lo[i] (bf16 bit patterns): the split every mode-3 operand goes through
int launch_split3(const float *x, size_t n, unsigned short *hi, unsigned short *mid, unsigned short *lo, hipStream_t s);

// slab[wg][k_in][n_out] = X[rows of wg]^T * dZ[rows of wg];  dZ cols [0,zsplit) from z0, rest from z1.
struct DwArgs {
  const float *x;   // [M][ldx], ldx % 4 == 0, rows contiguous
  const float *z0;  // [M][ldz0], ldz0 >= zsplit (may be a 4-float dummy row when zsplit == 0)
  const float *z1;  // [M][ldz1], ldz1 >= n_out
  const float *zeros;
  float *slab;  // [dw_num_slabs(n_out)][k_in][n_out]
  int ldx, ldz0, ldz1, zsplit;
  int m, k_in, n_out;
  int bf16;  // as RowGemmArgs::bf16 (0, 1; 3 = the split-operand kernel dw3_kernel, hidden-layer shapes only: dw3_ok)
  int nstage;  // LDS ring depth (set by launch_dw)
  // Quad-major operands (0 = off; the outputs of the channel-sliced aggregation kernels, gcn_csrq.hip):
  //   xq_nvert > 0 : X columns [0, 4 xq_quads) come from xq [m / xq_nvert][xq_quads][xq_nvert] float4; the other columns
  //                  from x with row stride ldx_src (x is pre-offset so that x + row * ldx_src + col addresses column col);
  //                  ldx stays the width of the staged image (>= k_in).  ldx_src = 0 means ldx.
  //   z0q_nvert > 0: z0 is quad-major [m / z0q_nvert][z0q_quads][z0q_nvert] float4, zsplit = 4 z0q_quads.
  const float *xq;
  int ldx_src, xq_nvert, xq_quads, z0q_nvert, z0q_quads;
};
int dw_num_slabs(int n_out);
// slab images the launch that takes `a` will write (launch_dw picks the kernel): what launch_slab_reduce* must sum
int dw_images(const DwArgs &a);
int dw_slab_capacity(int n_out);   // images to allocate: enough for either kernel
// dW = X^T dZ of a hidden layer in mode 3 (DwArgs::bf16 == 3, gcn_gemm3.hip; same slabs as launch_dw); dw3_ok: the shapes it takes
bool dw3_ok(const DwArgs &a);
int launch_dw3(const DwArgs &a, hipStream_t s);
// True when dw_kernel can take the first 4 * quads columns of a k_in-wide X quad-major (they must end where a wave's input tiles end).
bool dw_quad_major_ok(int k_in, int quads);
int launch_dw(const DwArgs &a, hipStream_t s);
// gcn_dww.hip: the hidden layers' dW on hybrid rows, waves specialised (round 6); same slab images as launch_dw
bool dww_ok(const DwArgs &a);
int dww_images();   // slab images a dww launch writes (>= dw_num_slabs(300): size the slab array with dw_slab_capacity)
int launch_dww(const DwArgs &a, hipStream_t s);
int launch_copy_cols(const float *src, int ld_src, int c0, int w, float *dst, long long m, hipStream_t s);
int launch_slab_reduce(const float *slab, int nslab, size_t stride, size_t n, float *out, hipStream_t s);
// same, and out[n .. n_out) = 0
int launch_slab_reduce_z(const float *slab, int nslab, size_t stride, size_t n, size_t n_out, float *out, hipStream_t s);
// same; accumulate != 0: out += the sum (the sum is formed first, in the same fixed order, then added once)
int launch_slab_reduce_za(const float *slab, int nslab, size_t stride, size_t n, size_t n_out, float *out, int accumulate,
                          hipStream_t s);

// CSR neighbour aggregation on the first c channels (+ bias + ReLU), model.py:356-358,363.
int launch_csr_fwd(const float *za, int ldza, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                   const float *val, const int32_t *heavy, int n_vert, int batch, float *y, int ldy, uint8_t *maskb,
                   int mld, int relu, hipStream_t s);
// Rows with many neighbours (hub rows of the fused touch graph) are listed once per call into `heavy`
// (csr_heavy_scratch_ints(n_vert) ints of scratch) and handled by a workgroup each; launch_csr_fwd/bwd take the list.
size_t csr_heavy_scratch_ints(int n_vert);
int csr_heavy_degree();   // rows with more neighbours than this go to the heavy pass; heavy == nullptr disables it
int launch_csr_heavy_list(const int32_t *rowptr, int n_vert, int32_t *heavy, hipStream_t s);
// Stand-alone layer backward entry: G = grad_y (* (y > 0) when relu); columns [0, cpad) -> ga [M][cpad] (input of the
// A^T gather), columns [cpad, npad) -> dz [M][npad] (pad columns zero).
int launch_relu_split(const float *gy, int ldgy, const float *y, int ldy, int relu, int n_out, int cpad, int npad,
                      long long m, float *ga, float *dz, hipStream_t s);
// dZa[:, :c] = A^T G[:, :c];  dZa[:, c:cpad] = G[:, c:cpad];  db partial sums of G[:, :c] -> slab [nslab][cpad].
int csr_bwd_num_slabs(int batch, int n_vert);
int launch_csr_bwd(const float *g, int ldg, int c, const int32_t *rowptrT, const int32_t *colT, const float *valT,
                   const int32_t *heavyT, int n_vert, int batch, float *dza, int lddza, float *db_slab, hipStream_t s);

// Channel-sliced aggregation, whole mesh resident in LDS (gcn_csrq.hip): inputs AND outputs quad-major
// [batch][Q = pad4(c)/4][n_vert] float4 (columns [0, 4 Q) of the activations / gradients), ReLU signs of those columns
// quad-major bytes [batch][Q][n_vert] (written by the forward, applied by the backward).  db_slab: [batch][pad4(c)].
// `ell`: the slot-major index image launch_csrq_ell builds from the same CSR (csrq_ell_ints(n_vert) ints, 256-B aligned).
bool csrq_fits(int n_vert, int cut_len);
int csrq_max_degree();   // rows up to this many edges are served entirely from the index entries a thread keeps in registers
size_t csrq_ell_ints(int n_vert);
int launch_csrq_ell(const int32_t *rowptr, const int32_t *col, const float *val, int n_vert, int32_t *ell, hipStream_t s);
int launch_csrq_fwd(const float *zq, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                    const float *val, const int32_t *heavy, const int32_t *ell, int n_vert, int batch, float *yq,
                    int yq_quads /* planes per mesh of yq (>= pad4(c) / 4; the first pad4(c) / 4 are written) */,
                    uint8_t *signq, int relu, hipStream_t s);
int launch_csrq_bwd(const float *gq, int c, const int32_t *rowptrT, const int32_t *colT, const float *valT,
                    const int32_t *heavyT, const int32_t *ellT, int n_vert, int batch, float *dzaq, const uint8_t *signq,
                    float *db_slab, hipStream_t s);
// D^-1 (P + J) as device pointers (a3vt_adj_split without the counts): the 3-channel aggregations of the output layer take it
struct SplitRef {
  const int32_t *rowptr, *col;
  const float *scale;
  const uint8_t *cls;
};
// The same for structured adjacencies D^-1 (P + J), J = a complete bipartite block (gcn_csrqs.hip, a3vt_adj_split): `img` =
// the index image launch_csrqs_image builds from the split (csrqs_image_ints(n_vert) ints); same layouts as csrq.
bool csrqs_fits(int n_vert, int cut_len);
int csrqs_max_degree();   // longest row of P the kernel takes
size_t csrqs_image_ints(int n_vert);
int launch_csrqs_image(const int32_t *rowptr, const int32_t *col, const float *scale, const uint8_t *cls, int n_vert,
                       int32_t *img, hipStream_t s);
int launch_csrqs_fwd(const float *zq, const float *bias, int c, const int32_t *img, int n_vert, int batch, float *yq,
                     int yq_quads, uint8_t *signq, int relu, hipStream_t s);
int launch_csrqs_bwd(const float *gq, int c, const int32_t *img, int n_vert, int batch, float *dzaq, const uint8_t *signq,
                     float *db_slab, hipStream_t s);
// True when launch_rowgemm will run (m, n_store) as ONE column block of 19-tile rows whose first epilogue column group
// holds the cpad aggregated columns — the shape for which the epilogues can write quad-major (RowGemmArgs::zq_nvert).
bool rowgemm_quad_major_ok(int m, int n_store, int cpad);

// Last layer (out = 3 channels, all aggregated, no activation), model.py:359-361.
int thin_num_slabs();
int launch_thin_fwd(const float *x, int ldx, int k, const float *w /*[k][3]*/, const float *bias /*[3]*/,
                    const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *heavy, int n_vert,
                    int batch, float *z3 /*[M][4] scratch*/, float *update /*[M][3]*/,
                    const float *xq /*quad-major X columns [0, 4 xq_quads) or nullptr (then x holds them all)*/, int xq_quads,
                    hipStream_t s, const SplitRef *sp = nullptr);
int launch_thin_bwd(const float *x, int ldx, int k, const float *w, const int32_t *rowptrT, const int32_t *colT,
                    const float *valT, const int32_t *heavyT, int n_vert, int batch,
                    const float *grad_update /*[M][3]*/, float *dz3 /*[2][M][4] scratch*/, int apply_mask,
                    float *g_prev /*[M][ldg]*/, int ldg, int n_store, float *dw_slab /*[thin_num_slabs()][k*3]*/,
                    float *db_slab /*[thin_num_slabs()][3]*/, float *gq /*quad-major columns [0, 4 nq) or nullptr*/, int nq,
                    const float *xq /*as launch_thin_fwd*/, int xq_quads, hipStream_t s, const SplitRef *sp = nullptr);

int launch_vertex_update(const float *vin, const float *upd, int batch, int n_vert, int n_vision, float *vout,
                         hipStream_t s);
int launch_check_finite(const float *d, size_t n, int32_t *flag, hipStream_t s);
int launch_fill_zero(float *d, size_t n, hipStream_t s);

// gcn_bf16s.hip — bf16-storage variants (gemm mode 2); `void *` rows are bf16, leading dimensions in elements
int launch_cvt_rows(const float *in, int ld_in, int n, void *out, int ld_out, long long m, hipStream_t s);
int launch_weight_images16(const WeightImages &w, int max_rows, int max_ld, hipStream_t s);  // ld / rows in bf16 elements
struct Dw16Args {
  const unsigned short *x;   // [M][ldx] bf16; input channels [xc0, xc0 + xw) are used (xw = k_in rounded up to 8)
  const unsigned short *z0;  // dZa [M][ldz0]: columns [0, zsplit)
  const unsigned short *z1;  // G   [M][ldz1]: columns [zsplit, n_out)
  const float *zeros;
  float *slab;               // [dw16_num_slabs(n_out)][k_in][n_out] fp32
  int ldx, xc0, xw, ldz0, ldz1, zsplit;
  int m, k_in, n_out;
  int nstage;                // set by launch_dw16
};
int dw16_num_slabs(int n_out);
int launch_dw16(const Dw16Args &a, hipStream_t s);
int launch_csr16_fwd(const void *za, int ldza, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                     const float *val, const int32_t *heavy, int n_vert, int batch, void *y, int ldy, uint8_t *maskb,
                     int mld, int relu, hipStream_t s);
int launch_csr16_bwd(const void *g, int ldg, int c, int cpad, const int32_t *rowptrT, const int32_t *colT,
                     const float *valT, const int32_t *heavyT, int n_vert, int batch, void *dza, int lddza,
                     float *db_slab, hipStream_t s);
// tiled aggregation on bf16 rows (gcn_bf16s.hip, round 6): plan = csr16t_plan_ints(n_vert) ints built by launch_csr16t_build
size_t csr16t_plan_ints(int n_vert);
bool csr16t_ok(int n_vert, int c, int max_degree, long long m);
int launch_csr16t_build(const int32_t *rowptr, const int32_t *col, const float *val, int n_vert, int32_t *plan, hipStream_t s);
int launch_csr16t_fwd(const void *za, int ldza, const float *bias, int c, const int32_t *plan, const int32_t *rowptr,
                      const int32_t *col, const float *val, int n_vert, int batch, void *y, int ldy, uint8_t *maskb, int mld,
                      int relu, hipStream_t s);
int launch_csr16t_bwd(const void *g, int ldg, int c, int cpad, const int32_t *planT, const int32_t *rowptrT,
                      const int32_t *colT, const float *valT, int n_vert, int batch, void *dza, int lddza, float *db_slab,
                      hipStream_t s);
int launch_thin16_fwd_product(const void *x, int ldx, int k, const float *w, long long m, float *z3, hipStream_t s);
int launch_thin16_bwd_main(const void *x, int ldx, int k, const float *w, const float *dz3, const float *du, long long m,
                           int apply_mask, void *gprev, int ldg, float *dw_slab, float *db_slab, hipStream_t s);
// 3-channel aggregation / padding helpers of the output layer (gcn_csr.hip), shared with the bf16-storage path
int launch_csr3(const float *z, const float *bias, const int32_t *rowptr, const int32_t *col, const float *val,
                const int32_t *heavy, int n_vert, int batch, float *out, int ldo, hipStream_t s,
                const SplitRef *sp = nullptr /* the split of THIS matrix, or of its transpose with transposed = true */,
                bool transposed = false);
int launch_pad3to4(const float *in, long long m, float *out, hipStream_t s);

// posenc.hip
size_t posenc_param_count(int input_size);
int posenc_num_slabs(int m);
int launch_posenc_fwd(const float *verts, const float *mask, int m, int input_size, const float *params, float *feats,
                      int ld, hipStream_t s);
int launch_posenc_bwd(const float *verts, const float *mask, int m, int input_size, const float *params,
                      const float *gfeats, int ld, float *gverts, float *gparams, float *scratch, hipStream_t s);

// posenc_wide.hip — the same encoder for wide inputs (I = 448 of the image models): three augmented products on rowgemm /
// dw_kernel; `acts` keeps E', H1', H2' for the backward (posenc_wide_acts_floats), `scratch` the operand images and the
// backward's intermediates (posenc_wide_scratch_floats).  ld must equal input_size.  gemm_bf16 != 0: the products of layers 2
// and 3 (and every backward product) take bf16-rounded operands (the bf16 configurations); the embedding layer stays exact.
bool posenc_wide_supported(int input_size);
size_t posenc_wide_acts_floats(int m, int input_size);
size_t posenc_wide_scratch_floats(int m, int input_size, int need_backward);
int launch_posenc_wide_fwd(const float *verts, const float *mask, int m, int input_size, const float *params, float *feats,
                           int ld, float *acts, float *scratch, const float *zeros, int gemm_bf16, hipStream_t s);
int launch_posenc_wide_bwd(const float *verts, const float *mask, int m, int input_size, const float *params,
                           const float *gfeats, int ld, const float *acts, float *gverts, float *gparams, float *scratch,
                           const float *zeros, int gemm_bf16, hipStream_t s);

// sample.hip
int launch_face_cdf(const float *verts, const int32_t *faces, int batch, int n_vert, int n_faces, float *cdf,
                    hipStream_t s);
int launch_sample_fwd(const float *verts, const int32_t *faces, const float *cdf, int batch, int n_vert, int n_faces,
                      int draws, int num, const int32_t *fi_in, const float *u_in, const float *v_in, uint64_t seed,
                      uint64_t offset, float *points, int32_t *fi_out, float *u_out, float *v_out, hipStream_t s);
int launch_sample_bwd(const int32_t *faces, int batch, int n_vert, int n_faces, int draws, int num,
                      const int32_t *fi, const float *u, const float *v, const float *gpoints, float *gverts,
                      hipStream_t s);

// pooling.hip — Image_Encoder.pooling (model.py:70-103): projection + bilinear gather from channels-last maps
constexpr int kMaxMaps = 4;
struct PoolArgs {
  const float *verts;   // [B][N][3]
  float proj[12];       // K.RT, row-major 3 x 4
  int batch, n_vert, n_maps;
  const float *maps[kMaxMaps];   // channels-last [B][H][W][C]
  float *gmaps[kMaxMaps];        // backward: gradient of the maps, same layout (overwritten)
  int C[kMaxMaps], H[kMaxMaps], W[kMaxMaps], off[kMaxMaps];  // off = first output channel of the map
  float *feats;         // fwd out  [B*N][ld]
  const float *base;    // fwd, optional [B*N][ld]: feats = base + pooled (the reference's `positional + mask + image` sum)
  const float *gfeats;  // bwd in   [B*N][ld]
  int ld;
  float *gverts;        // bwd out  [B*N][3]
};
int launch_pool_fwd(PoolArgs a, hipStream_t s);
int launch_pool_bwd(PoolArgs a, hipStream_t s);

// bias_grad.hip: column sums of a channels-last gradient map (slab: bias_grad_wgs(rows * c, c) * c floats)
int bias_grad_wgs(long long n_elem, int c);
int launch_bias_grad(const void *g, int bf16, long long rows, int c, float *out, float *slab, hipStream_t s);

// bnrelu.hip: training BatchNorm2d + ReLU on channels-last bf16 maps (save: [4][c] mean, invstd, scale, shift)
int bnrelu_wgs(long long n_elem, int c, int per_thread, int cap);
size_t bnrelu_scratch_bytes(int c);
int launch_bnrelu_fwd(const void *x, long long rows, int c, const float *gamma, const float *beta, const float *pre_bias, float eps,
                      float momentum, float *running_mean, float *running_var, long long *num_batches, void *y, float *save,
                      void *scratch, hipStream_t s);
int launch_bnrelu_bwd(const void *dy, const void *x, long long rows, int c, const float *save, void *dx, float *dgamma,
                      float *dbeta, float *dx_colsum, void *scratch, hipStream_t s);

// bnrelu.hip: up to kCastBatchMax fp32 tensors [outer][inner][hw] -> bf16 [outer][hw][inner] in one launch
constexpr int kCastBatchMax = 96;
struct CastBatch {
  int n;
  const float *src[kCastBatchMax];
  uint16_t *dst[kCastBatchMax];
  int inner[kCastBatchMax], hw[kCastBatchMax];
  long long start[kCastBatchMax + 1];   // first destination element of tensor k in the launch's index space
};
int launch_cast_weights(const CastBatch &b, hipStream_t s);

// conv5.hip: 5 x 5 convolutions of the image pyramid on channels-last bf16 maps, (cin, cout, stride) in {(16,16,1), (32,32,1),
// (16,32,2)}; with flip = 1 / pad = 3 (stride 1 only) the input gradient
bool conv5_shape_ok(int cin, int cout, int stride);
size_t conv5_weight_image_bytes(int rows, int inner);
int launch_conv5_weight_image(const float *w, int flip, int cout, int cin, void *image, hipStream_t s);
int launch_conv5(const void *x, int batch, int h, int w, int cin, int cout, int stride, int pad, const void *image, const float *bias,
                 void *y, hipStream_t s);
int launch_conv5_up3(const void *gy, int batch, int ho, int wo, const void *image, void *gx, hipStream_t s);
size_t conv5_wrw_scratch_bytes(int cin, int cout);
int launch_conv5_wrw(const void *x, const void *gy, int batch, int h, int w, int cin, int cout, int stride, float *gw, void *scratch,
                     hipStream_t s);

// adam.hip: torch's Adam step over a table of fp32 tensors cut into chunks of adam_chunk_elems() elements (device tables)
int adam_chunk_elems();
int launch_adam(void *const *param, const void *const *grad, void *const *exp_avg, void *const *exp_avg_sq,
                const long long *numel, const int *chunk_tensor, const long long *chunk_off, int n_chunks, double lr, double beta1,
                double beta2, double eps, double weight_decay, long long step, hipStream_t s);

// chamfer.hip
size_t chamfer_scratch_bytes(int draws, int batch, int q);
// algo: 0 = choose (pruned search when the workspace holds it and the clouds are large enough to pay for the sort),
// 1 = brute force, one launch per direction (no scratch), 2 = brute force, one sweep for both directions, 3 = pruned search
enum { NN_AUTO = 0, NN_BRUTE_TWO_PASS = 1, NN_BRUTE_SWEEP = 2, NN_PRUNED = 3 };
size_t chamfer_workspace_bytes(int draws, int batch, int p, int q);
int launch_chamfer_fwd(const float *x, const float *y, int draws, int batch, int p, int q, float *dxy, int32_t *ixy,
                       float *dyx, int32_t *iyx, float *cd, void *scratch, size_t scratch_bytes, int algo, hipStream_t s,
                       int y_batch = 0 /* clouds in y (mesh b asks y[b % y_batch]); 0 = batch */);
// nn_prune.hip
size_t nn_pruned_workspace_bytes(int draws, int batch, int p, int q);
int launch_nn_pruned(const float *x, const float *y, int draws, int batch, int p, int q, float *dxy, int32_t *ixy,
                     float *dyx, int32_t *iyx, void *ws, hipStream_t s, int y_batch = 0);
int launch_chamfer_bwd(const float *x, const float *y, int draws, int batch, int p, int q, const int32_t *ixy,
                       const int32_t *iyx, const float *gcd, float *gx, float *gy, hipStream_t s);

}  // namespace a3vt
