// posenc_wide.hip — vertex-feature encoder for wide inputs (input_size 448 of the image models), forward and backward.
//
// Replaces, for input sizes whose parameters do not fit the LDS-resident kernel of posenc.hip (133 k parameters at 448):
//   Positional_Encoder.nerf_embedding / .forward  reconstruction/vision/model.py:381-399   63 -> I/4 -> I/2 -> I, ReLU between
//   Mask_Encoder.forward                          :410-414                                 Embedding(4, I)[mask]
//   and their sum                                 :243-246, 261-266, 274-278
// which the mirror used to run as ~30 torch / rocBLAS launches per stage (5-6 ms of the configs[3] step).
//
// Formulation: every layer is ONE product on the fp32 matrix pipe (rowgemm_kernel, exact v_mfma_f32_16x16x4_f32 chains)
// of an AUGMENTED activation row with an augmented weight image, so that biases, the mask embedding and their gradients
// need no kernels of their own:
//   E'  [M][68]        = [ e(p) (63) | 1 | onehot(mask) (4) ]
//   H1' [M][LH1]       = relu(E'  B1'),  B1' = [ W1^T | 0 ; b1 | 1 ; 0 | I4 ]  -> [ relu(W1 e + b1) | 1 | onehot | 0.. ]
//   H2' [M][LH2]       = relu(H1' B2'),  likewise
//   OUT [M][I]         = H2' B3',        B3' = [ W3^T ; b3 ; Emb ]             -> W3 h2 + b3 + Emb[mask]
// (the constant 1 and the one-hot columns ride through the ReLUs unchanged).  Backward: dB' = X'^T dZ on dw_kernel —
// its bias row IS the bias gradient, its four one-hot rows ARE the embedding gradient — and dX' = dZ B'^T on rowgemm,
// the ReLU masks from the saved activations, the sin / cos chain rule for the vertex positions at the end.
// Every sum runs in a fixed order (slab reduces): bit-reproducible.
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

struct WideDims {
  int I, H1, H2;        // layer widths
  int LE, LH1, LH2;     // augmented row lengths (multiples of 4): E' 68, H1' = pad4(H1 + 5), H2' = pad4(H2 + 5)
  // packed parameter offsets (torch (out, in) row-major, state-dict order — the same packing as posenc.hip)
  size_t oW1, ob1, oW2, ob2, oW3, ob3, oE, nparam;
  __host__ __device__ WideDims(int input_size) {
    I = input_size;
    H1 = I / 4;
    H2 = I / 2;
    LE = 68;
    LH1 = (H1 + 5 + 3) & ~3;
    LH2 = (H2 + 5 + 3) & ~3;
    oW1 = 0;
    ob1 = oW1 + (size_t)H1 * 63;
    oW2 = ob1 + H1;
    ob2 = oW2 + (size_t)H2 * H1;
    oW3 = ob2 + H2;
    ob3 = oW3 + (size_t)I * H2;
    oE = ob3 + I;
    nparam = oE + 4 * (size_t)I;
  }
};

__device__ __forceinline__ float pew_freq(int i) {
  // model.py:385-389: np.pi for i == 0 else np.pi * 2 * i (python double), multiplied into a float32 tensor
  return i == 0 ? (float)3.141592653589793 : (float)(3.141592653589793 * 2.0 * i);
}

// E' rows: one thread per vertex.
__global__ void pew_embed_kernel(const float *__restrict__ verts, const float *__restrict__ mask, int m,
                                 float *__restrict__ e) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= m) return;
  const float p[3] = {verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2]};
  float *row = e + (size_t)v * 68;
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    const float f = pew_freq(i);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      row[6 * i + c] = sinf(f * p[c]);
      row[6 * i + 3 + c] = cosf(f * p[c]);
    }
  }
  row[60] = p[0];
  row[61] = p[1];
  row[62] = p[2];
  row[63] = 1.f;
  int tok = (int)mask[v];  // mask.long() (model.py:413)
  tok = tok < 0 ? 0 : (tok > 3 ? 3 : tok);
#pragma unroll
  for (int k = 0; k < 4; ++k) row[64 + k] = tok == k ? 1.f : 0.f;
}

// One layer's augmented weight as the two operand images rowgemm needs:
//   fwd  Bt [rows_f][ld_f]: Bt[n][k] = B'[k][n]   (output column n of the layer, input column k)
//   bwd  Bt [rows_b][ld_b]: Bt[k][n] = B'[k][n]   (dX' = dZ B'^T: output column k, reduction over n)
// B' [kin_aug][nout_aug]: rows < kin = W^T, row kin = bias (+ 1 -> column nout), rows kin+1..kin+4 = identity into columns
// nout+1..nout+4 — or, for the last layer (`last`), the four embedding rows and no pass-through columns.
__global__ void pew_weight_image_kernel(const float *__restrict__ w /*(nout, kin)*/, const float *__restrict__ b,
                                        const float *__restrict__ emb /*(4, nout) or nullptr*/, int kin, int nout,
                                        int last, float *__restrict__ btf, int rows_f, int ld_f,
                                        float *__restrict__ btb, int rows_b, int ld_b) {
  const int total_f = rows_f * ld_f, total_b = rows_b * ld_b;
  auto bprime = [&](int k, int n) -> float {   // B'[k][n]
    if (n < nout) {
      if (k < kin) return w[(size_t)n * kin + k];
      if (k == kin) return b[n];
      if (last && k <= kin + 4) return emb[(size_t)(k - kin - 1) * nout + n];
      return 0.f;
    }
    if (last) return 0.f;
    const int j = n - nout;   // pass-through columns: 0 -> the constant one, 1..4 -> the one-hot token
    return (j <= 4 && k == kin + j) ? 1.f : 0.f;
  };
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total_f + total_b; i += gridDim.x * blockDim.x) {
    if (i < total_f) {
      const int n = i / ld_f, k = i - n * ld_f;
      btf[i] = bprime(k, n);
    } else {
      const int ii = i - total_f;
      const int k = ii / ld_b, n = ii - k * ld_b;
      btb[ii] = bprime(k, n);
    }
  }
}

// g *= (y > 0), 16 bytes per thread (rows of g and y have the same length)
__global__ void pew_relu_bwd_kernel(float *__restrict__ g, const float *__restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 gv = reinterpret_cast<f32x4 *>(g)[i];
    const f32x4 yv = reinterpret_cast<const f32x4 *>(y)[i];
#pragma unroll
    for (int t = 0; t < 4; ++t) gv[t] = yv[t] > 0.f ? gv[t] : 0.f;
    reinterpret_cast<f32x4 *>(g)[i] = gv;
  }
}

// gverts from dE' and E' (d sin(f p)/dp = f cos(f p), d cos(f p)/dp = -f sin(f p)); one thread per vertex
__global__ void pew_embed_bwd_kernel(const float *__restrict__ e, const float *__restrict__ de, int m,
                                     float *__restrict__ gverts) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= m) return;
  const float *er = e + (size_t)v * 68, *dr = de + (size_t)v * 68;
  float gp[3] = {dr[60], dr[61], dr[62]};
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    const float f = pew_freq(i);
#pragma unroll
    for (int c = 0; c < 3; ++c) gp[c] += f * (er[6 * i + 3 + c] * dr[6 * i + c] - er[6 * i + c] * dr[6 * i + 3 + c]);
  }
  gverts[3 * (size_t)v + 0] = gp[0];
  gverts[3 * (size_t)v + 1] = gp[1];
  gverts[3 * (size_t)v + 2] = gp[2];
}

// dB' [kin_aug][ldn] (row = augmented input column, col = output column) -> packed gradients: dW (nout, kin) = dB'^T, db = row
// kin, and for the last layer dEmb (4, nout) = rows kin+1..kin+4.
__global__ void pew_unpack_kernel(const float *__restrict__ db, int ldn, int kin, int nout, int last,
                                  float *__restrict__ gw, float *__restrict__ gb, float *__restrict__ gemb) {
  const int total = nout * kin + nout + (last ? 4 * nout : 0);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (i < nout * kin) {
      const int n = i / kin, k = i - n * kin;
      gw[i] = db[(size_t)k * ldn + n];
    } else if (i < nout * kin + nout) {
      const int n = i - nout * kin;
      gb[n] = db[(size_t)kin * ldn + n];
    } else {
      const int j = i - nout * kin - nout, t = j / nout, n = j - t * nout;
      gemb[(size_t)t * nout + n] = db[(size_t)(kin + 1 + t) * ldn + n];
    }
  }
}

struct WideLayout {   // float offsets into the caller's scratch, 256-B aligned regions
  size_t btf[3], btb[3];      // operand images
  size_t dh2, dh1, de;        // backward: gradients of the augmented rows
  size_t dbp;                 // backward: dB' of the layer in flight [kin_aug][nout]
  size_t slab;                // backward: dw slabs
  size_t total;
};
struct WideActs {   // float offsets into the saved-activation buffer
  size_t e, h1, h2, total;
};

static size_t up64(size_t x) { return (x + 63) / 64 * 64; }

static WideActs wide_acts(const WideDims &d, size_t m) {
  WideActs a{};
  a.e = 0;
  a.h1 = up64(a.e + m * d.LE);
  a.h2 = up64(a.h1 + m * d.LH1);
  a.total = up64(a.h2 + m * d.LH2);
  return a;
}

// layer l: input row length / real width, output row length / real width
static void layer_dims(const WideDims &d, int l, int &lin, int &kin, int &lout, int &nout) {
  if (l == 0) { lin = d.LE; kin = 63; lout = d.LH1; nout = d.H1; }
  else if (l == 1) { lin = d.LH1; kin = d.H1; lout = d.LH2; nout = d.H2; }
  else { lin = d.LH2; kin = d.H2; lout = d.I; nout = d.I; }
}

static WideLayout wide_layout(const WideDims &d, size_t m, int need_backward) {
  WideLayout L{};
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off = up64(off + n); return o; };
  for (int l = 0; l < 3; ++l) {
    int lin, kin, lout, nout;
    layer_dims(d, l, lin, kin, lout, nout);
    L.btf[l] = take((size_t)rowgemm_bt_rows(lout) * pad16(lin));
    L.btb[l] = take((size_t)rowgemm_bt_rows(lin) * pad16(lout));
  }
  if (need_backward) {
    L.dh2 = take(m * d.LH2);
    L.dh1 = take(m * d.LH1);
    L.de = take(m * d.LE);
    L.dbp = take((size_t)d.LH2 * d.I);
    size_t slab = 0;
    for (int l = 0; l < 3; ++l) {
      int lin, kin, lout, nout;
      layer_dims(d, l, lin, kin, lout, nout);
      const size_t s = (size_t)dw_num_slabs(lout) * lin * lout;
      slab = s > slab ? s : slab;
    }
    L.slab = take(slab);
  }
  L.total = off;
  return L;
}

static int build_images(const WideDims &d, const float *params, float *scratch, const WideLayout &L, hipStream_t s) {
  const float *w[3] = {params + d.oW1, params + d.oW2, params + d.oW3};
  const float *b[3] = {params + d.ob1, params + d.ob2, params + d.ob3};
  for (int l = 0; l < 3; ++l) {
    int lin, kin, lout, nout;
    layer_dims(d, l, lin, kin, lout, nout);
    const int rows_f = rowgemm_bt_rows(lout), ld_f = pad16(lin), rows_b = rowgemm_bt_rows(lin), ld_b = pad16(lout);
    const int total = rows_f * ld_f + rows_b * ld_b;
    A3VT_LAUNCH(pew_weight_image_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024), dim3(256), 0, s, w[l], b[l],
                l == 2 ? params + d.oE : nullptr, kin, nout, l == 2 ? 1 : 0, scratch + L.btf[l], rows_f, ld_f,
                scratch + L.btb[l], rows_b, ld_b);
    A3VT_CHECK_LAUNCH();
  }
  return 0;
}

static RowGemmArgs plain_gemm(const float *a, int lda, int k, const float *bt, int n_store, float *c, int ldc, size_t m,
                              const float *zeros, int relu, int bf16) {
  RowGemmArgs g{};
  g.bf16 = bf16 ? 1 : 0;   // bf16 OPERAND mode of the bf16 configurations: fp32 rows in memory, operands rounded into the matrix pipe
  g.a0 = g.a1 = a;
  g.lda0 = g.lda1 = lda;
  g.ksplit = k;
  g.bt = bt;
  g.ldb = pad16(k);
  g.zeros = zeros;
  g.m = (int)m;
  g.k = k;
  g.n_store = n_store;
  g.c = c;
  g.ldc = ldc;
  g.plain_relu = relu;
  return g;
}

}  // namespace

bool posenc_wide_supported(int input_size) {
  // I / 4 and I / 2 whole, rows 16-byte aligned, and the widest augmented row within dw_kernel's 320 input channels
  return input_size >= 16 && input_size % 8 == 0 && WideDims(input_size).LH2 <= 320 && input_size <= 1024;
}
size_t posenc_wide_acts_floats(int m, int input_size) { return wide_acts(WideDims(input_size), (size_t)m).total; }
size_t posenc_wide_scratch_floats(int m, int input_size, int need_backward) {
  return wide_layout(WideDims(input_size), (size_t)m, need_backward).total;
}

int launch_posenc_wide_fwd(const float *verts, const float *mask, int m, int input_size, const float *params, float *feats,
                           int ld, float *acts, float *scratch, const float *zeros, int gemm_bf16, hipStream_t s) {
  const WideDims d(input_size);
  if (!posenc_wide_supported(input_size) || ld != input_size) {   // (input_size % 8 == 0: the feature rows have no pad columns)
    set_error("posenc_wide: input_size=%d ld=%d unsupported", input_size, ld);
    return -1;
  }
  const WideActs A = wide_acts(d, (size_t)m);
  const WideLayout L = wide_layout(d, (size_t)m, 0);
  if (int rc = build_images(d, params, scratch, L, s)) return rc;
  A3VT_LAUNCH(pew_embed_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, verts, mask, m, acts + A.e);
  A3VT_CHECK_LAUNCH();
  if (int rc = launch_rowgemm(plain_gemm(acts + A.e, d.LE, d.LE, scratch + L.btf[0], d.LH1, acts + A.h1, d.LH1, m, zeros, 1, 0),   // the embedding layer stays exact in every mode (positions, sin / cos)
                              EPI_PLAIN, s))
    return rc;
  if (int rc = launch_rowgemm(plain_gemm(acts + A.h1, d.LH1, d.LH1, scratch + L.btf[1], d.LH2, acts + A.h2, d.LH2, m, zeros, 1, gemm_bf16),
                              EPI_PLAIN, s))
    return rc;
  if (int rc = launch_rowgemm(plain_gemm(acts + A.h2, d.LH2, d.LH2, scratch + L.btf[2], d.I, feats, ld, m, zeros, 0, gemm_bf16), EPI_PLAIN, s))
    return rc;
  return 0;
}

int launch_posenc_wide_bwd(const float *verts, const float *mask, int m, int input_size, const float *params,
                           const float *gfeats, int ld, const float *acts, float *gverts, float *gparams, float *scratch,
                           const float *zeros, int gemm_bf16, hipStream_t s) {
  (void)verts; (void)mask;
  const WideDims d(input_size);
  if (!posenc_wide_supported(input_size) || ld != input_size) {
    set_error("posenc_wide: input_size=%d ld=%d unsupported", input_size, ld);
    return -1;
  }
  const WideActs A = wide_acts(d, (size_t)m);
  const WideLayout L = wide_layout(d, (size_t)m, 1);
  if (int rc = build_images(d, params, scratch, L, s)) return rc;
  const float *x[3] = {acts + A.e, acts + A.h1, acts + A.h2};
  float *dx[3] = {scratch + L.de, scratch + L.dh1, scratch + L.dh2};
  float *gw[3] = {gparams + d.oW1, gparams + d.oW2, gparams + d.oW3};
  float *gb[3] = {gparams + d.ob1, gparams + d.ob2, gparams + d.ob3};
  const float *dz = gfeats;
  int ldz = ld;
  for (int l = 2; l >= 0; --l) {
    int lin, kin, lout, nout;
    layer_dims(d, l, lin, kin, lout, nout);
    const int nz = l == 2 ? d.I : lout;   // columns of dZ that exist
    // dB' = X'^T dZ
    DwArgs w{};
    w.x = x[l];
    w.ldx = lin;
    w.z0 = zeros;   // zsplit = 0: never consumed
    w.ldz0 = 4;
    w.z1 = dz;
    w.ldz1 = ldz;
    w.zsplit = 0;
    w.zeros = zeros;
    w.slab = scratch + L.slab;
    w.m = m;
    w.k_in = lin;
    w.n_out = nz;
    w.bf16 = (gemm_bf16 && l > 0) ? 1 : 0;   // layer 0 (its weight gradient and the gradient that goes on to the positions) exact
    if (int rc = launch_dw(w, s)) return rc;
    if (int rc = launch_slab_reduce(scratch + L.slab, dw_num_slabs(nz), (size_t)lin * nz, (size_t)lin * nz, scratch + L.dbp, s))
      return rc;
    {
      const int total = nout * kin + nout + (l == 2 ? 4 * nout : 0);
      A3VT_LAUNCH(pew_unpack_kernel, dim3(cdiv(total, 256) < 1024 ? cdiv(total, 256) : 1024), dim3(256), 0, s, scratch + L.dbp, nz,
                  kin, nout, l == 2 ? 1 : 0, gw[l], gb[l], l == 2 ? gparams + d.oE : nullptr);
      A3VT_CHECK_LAUNCH();
    }
    // dX' = dZ B'^T, then through the ReLU of the layer below (E' has none)
    RowGemmArgs g = plain_gemm(dz, ldz, nz, scratch + L.btb[l], lin, dx[l], lin, m, zeros, 0, l > 0 ? gemm_bf16 : 0);
    g.ldb = pad16(lout);
    if (int rc = launch_rowgemm(g, EPI_PLAIN, s)) return rc;
    if (l > 0) {
      const size_t n4 = (size_t)m * lin / 4;
      A3VT_LAUNCH(pew_relu_bwd_kernel, dim3(n4 / 256 + 1 < 4096 ? (unsigned)(n4 / 256 + 1) : 4096u), dim3(256), 0, s, dx[l], x[l], n4);
      A3VT_CHECK_LAUNCH();
    }
    dz = dx[l];
    ldz = lin;
  }
  A3VT_LAUNCH(pew_embed_bwd_kernel, dim3(cdiv(m, 256)), dim3(256), 0, s, acts + A.e, scratch + L.de, m, gverts);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
