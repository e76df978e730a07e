// posenc.hip — fused vertex-feature encoder, forward and backward (gfx950).
//
// Replaces ~30 small launches per stage of reconstruction/vision/model.py:
//   Positional_Encoder.nerf_embedding (:381-391)  e = [sin(f_i p), cos(f_i p)]_{i=0..9} ++ p,  f = pi*{1,2,4,...,18}
//   Positional_Encoder.forward        (:393-399)  63 -> I/4 -> I/2 -> I MLP, ReLU between
//   Mask_Encoder.forward              (:410-414)  Embedding(4, I)[mask]
//   and their sum                     (:232,240,262,275)
// One thread per vertex; the 2.6k parameters sit in LDS (transposed so that each k-step reads one
// contiguous row).  Backward recomputes the activations, back-propagates to the vertex position
// (the path that links refinement stage s+1 to stage s) and reduces the parameter gradients per
// workgroup through LDS into slabs that slab_reduce sums in a fixed order.
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int I>
struct PE {
  static constexpr int H1 = I / 4, H2 = I / 2, E = 63;
  // packed parameter offsets, torch (out,in) row-major, state-dict order
  static constexpr int oW1 = 0, ob1 = oW1 + H1 * E, oW2 = ob1 + H1, ob2 = oW2 + H2 * H1, oW3 = ob2 + H2,
                       ob3 = oW3 + I * H2, oE = ob3 + I, N = oE + 4 * I;
};

size_t posenc_param_count(int input_size) {
  const int h1 = input_size / 4, h2 = input_size / 2;
  return (size_t)h1 * 63 + h1 + (size_t)h2 * h1 + h2 + (size_t)input_size * h2 + input_size + 4 * input_size;
}

__device__ __forceinline__ float pe_freq(int i) {
  // model.py:385-389: np.pi for i == 0 else np.pi * 2 * i (python double), multiplied into a float32 tensor
  return i == 0 ? (float)3.141592653589793 : (float)(3.141592653589793 * 2.0 * i);
}

template <int I>
__device__ __forceinline__ void pe_embed(const float p[3], float e[63]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const float f = pe_freq(i);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e[6 * i + c] = sinf(f * p[c]);
      e[6 * i + 3 + c] = cosf(f * p[c]);
    }
  }
  e[60] = p[0];
  e[61] = p[1];
  e[62] = p[2];
}

// LDS parameter image: W1t [63][H1], W2t [H1][H2], W3t [H2][I] (transposed), b1, b2, b3, E [4][I]
template <int I>
struct PELds {
  using P = PE<I>;
  static constexpr int oW1t = 0, oW2t = oW1t + 63 * P::H1, oW3t = oW2t + P::H1 * P::H2, ob1 = oW3t + P::H2 * I,
                       ob2 = ob1 + P::H1, ob3 = ob2 + P::H2, oE = ob3 + I, N = oE + 4 * I;
};

template <int I>
__device__ __forceinline__ void pe_load_params(const float *__restrict__ params, float *__restrict__ sp) {
  using P = PE<I>;
  using L = PELds<I>;
  for (int i = threadIdx.x; i < P::H1 * 63; i += blockDim.x) sp[L::oW1t + (i % 63) * P::H1 + i / 63] = params[P::oW1 + i];
  for (int i = threadIdx.x; i < P::H2 * P::H1; i += blockDim.x)
    sp[L::oW2t + (i % P::H1) * P::H2 + i / P::H1] = params[P::oW2 + i];
  for (int i = threadIdx.x; i < I * P::H2; i += blockDim.x) sp[L::oW3t + (i % P::H2) * I + i / P::H2] = params[P::oW3 + i];
  for (int i = threadIdx.x; i < P::H1; i += blockDim.x) sp[L::ob1 + i] = params[P::ob1 + i];
  for (int i = threadIdx.x; i < P::H2; i += blockDim.x) sp[L::ob2 + i] = params[P::ob2 + i];
  for (int i = threadIdx.x; i < I; i += blockDim.x) sp[L::ob3 + i] = params[P::ob3 + i];
  for (int i = threadIdx.x; i < 4 * I; i += blockDim.x) sp[L::oE + i] = params[P::oE + i];
}

template <int I>
__device__ __forceinline__ void pe_mlp(const float *__restrict__ sp, const float e[63], float h1[PE<I>::H1],
                                       float h2[PE<I>::H2]) {
  using P = PE<I>;
  using L = PELds<I>;
#pragma unroll
  for (int j = 0; j < P::H1; ++j) h1[j] = sp[L::ob1 + j];
#pragma unroll
  for (int k = 0; k < 63; ++k)
#pragma unroll
    for (int j = 0; j < P::H1; ++j) h1[j] += sp[L::oW1t + k * P::H1 + j] * e[k];
#pragma unroll
  for (int j = 0; j < P::H1; ++j) h1[j] = h1[j] > 0.f ? h1[j] : 0.f;
#pragma unroll
  for (int j = 0; j < P::H2; ++j) h2[j] = sp[L::ob2 + j];
#pragma unroll
  for (int k = 0; k < P::H1; ++k)
#pragma unroll
    for (int j = 0; j < P::H2; ++j) h2[j] += sp[L::oW2t + k * P::H2 + j] * h1[k];
#pragma unroll
  for (int j = 0; j < P::H2; ++j) h2[j] = h2[j] > 0.f ? h2[j] : 0.f;
}

template <int I>
__global__ __launch_bounds__(256) void posenc_fwd_kernel(const float *__restrict__ verts,
                                                         const float *__restrict__ mask, int m,
                                                         const float *__restrict__ params,
                                                         float *__restrict__ feats, int ld) {
  using P = PE<I>;
  using L = PELds<I>;
  __shared__ float sp[L::N];
  pe_load_params<I>(params, sp);
  __syncthreads();
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= m) return;
  const float p[3] = {verts[3 * (long long)v], verts[3 * (long long)v + 1], verts[3 * (long long)v + 2]};
  float e[63], h1[P::H1], h2[P::H2];
  pe_embed<I>(p, e);
  pe_mlp<I>(sp, e, h1, h2);
  int tok = (int)mask[v];  // mask.long() (model.py:413)
  tok = tok < 0 ? 0 : (tok > 3 ? 3 : tok);
  float *out = feats + (long long)v * ld;
  // output channels in groups of 10 to bound live registers
#pragma unroll
  for (int o0 = 0; o0 < I; o0 += 10) {
    float acc[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = o0 + j < I ? sp[L::ob3 + o0 + j] : 0.f;
#pragma unroll
    for (int k = 0; k < P::H2; ++k)
#pragma unroll
      for (int j = 0; j < 10; ++j)
        if (o0 + j < I) acc[j] += sp[L::oW3t + k * I + o0 + j] * h2[k];
#pragma unroll
    for (int j = 0; j < 10; ++j)
      if (o0 + j < I) out[o0 + j] = acc[j] + sp[L::oE + tok * I + o0 + j];
  }
  for (int o = I; o < ld; ++o) out[o] = 0.f;
}

int launch_posenc_fwd(const float *verts, const float *mask, int m, int input_size, const float *params, float *feats,
                      int ld, hipStream_t s) {
  if (input_size != 50) {
    set_error("posenc: fused kernel supports input_size == 50 only (got %d)", input_size);
    return -1;
  }
  A3VT_LAUNCH((posenc_fwd_kernel<50>), dim3(cdiv(m, 256)), dim3(256), 0, s, verts, mask, m, params, feats, ld);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward.  Workgroup = 128 vertices, one per thread.  Every per-vertex vector the weight-gradient outer
// products need (e, gout, h2, dh2, h1, dh1) is parked in LDS rows; the thread keeps only the small hidden
// vectors in registers and walks the long ones with rolled loops (no spills).  After one barrier the
// workgroup reduces the outer products over its 128 vertices and writes one slab of partial gradients.
// ------------------------------------------------------------------------------------------------
constexpr int kPEBwdThreads = 128;

int posenc_num_slabs(int m) { return cdiv(m, kPEBwdThreads); }

template <int I>
__global__ __launch_bounds__(kPEBwdThreads) void posenc_bwd_kernel(const float *__restrict__ verts,
                                                                   const float *__restrict__ mask, int m,
                                                                   const float *__restrict__ params,
                                                                   const float *__restrict__ gfeats, int ld,
                                                                   float *__restrict__ gverts,
                                                                   float *__restrict__ slab) {
  using P = PE<I>;
  using L = PELds<I>;
  constexpr int T = kPEBwdThreads;
  constexpr int H1 = P::H1, H2 = P::H2;
  extern __shared__ float smem[];
  // per-vertex vectors, row = vertex.  E, H2 and H1 carry a trailing 1 so that the bias gradients fall out of the same
  // outer-product tiles as the weight gradients; OH is the one-hot mask token (embedding gradient).
  constexpr int LE = 64, LG = I, LH2 = H2 + 1, LD2 = H2, LH1 = H1 + 1, LD1 = H1, LOH = 4;
  float *sp = smem;                  // parameters (transposed image)
  float *sE = sp + L::N;             // [T][64]  e | 1
  float *sG = sE + T * LE;           // [T][I]   gout
  float *sH2 = sG + T * LG;          // [T][H2+1] h2 | 1
  float *sD2 = sH2 + T * LH2;        // [T][H2]  dh2
  float *sH1 = sD2 + T * LD2;        // [T][H1+1] h1 | 1
  float *sD1 = sH1 + T * LH1;        // [T][H1]  dh1
  float *sOH = sD1 + T * LD1;        // [T][4]   one-hot token (all zero for padding rows); + 64 floats of slack
  pe_load_params<I>(params, sp);
  __syncthreads();
  const int t = threadIdx.x;
  const int v = blockIdx.x * T + t;
  const bool live = v < m;
  float *out = slab + (size_t)blockIdx.x * P::N;

  float p[3] = {0.f, 0.f, 0.f};
  if (live) {
    p[0] = verts[3 * (long long)v];
    p[1] = verts[3 * (long long)v + 1];
    p[2] = verts[3 * (long long)v + 2];
  }
  // ---- forward recompute: e -> LDS, h1 / h2 in registers
  float *myE = sE + t * LE;
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    const float f = pe_freq(i);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      myE[6 * i + c] = sinf(f * p[c]);
      myE[6 * i + 3 + c] = cosf(f * p[c]);
    }
  }
  myE[60] = p[0];
  myE[61] = p[1];
  myE[62] = p[2];
  myE[63] = 1.f;
  float h1[H1], h2[H2];
#pragma unroll
  for (int j = 0; j < H1; ++j) h1[j] = sp[L::ob1 + j];
#pragma unroll 1
  for (int k = 0; k < 63; ++k) {
    const float ek = myE[k];
#pragma unroll
    for (int j = 0; j < H1; ++j) h1[j] += sp[L::oW1t + k * H1 + j] * ek;
  }
#pragma unroll
  for (int j = 0; j < H1; ++j) {
    h1[j] = h1[j] > 0.f ? h1[j] : 0.f;
    sH1[t * LH1 + j] = h1[j];
  }
  sH1[t * LH1 + H1] = 1.f;
#pragma unroll
  for (int j = 0; j < H2; ++j) h2[j] = sp[L::ob2 + j];
#pragma unroll
  for (int k = 0; k < H1; ++k)
#pragma unroll
    for (int j = 0; j < H2; ++j) h2[j] += sp[L::oW2t + k * H2 + j] * h1[k];
#pragma unroll
  for (int j = 0; j < H2; ++j) {
    h2[j] = h2[j] > 0.f ? h2[j] : 0.f;
    sH2[t * LH2 + j] = h2[j];
  }
  sH2[t * LH2 + H2] = 1.f;
  int tok = live ? (int)mask[v] : 0;
  tok = tok < 0 ? 0 : (tok > 3 ? 3 : tok);
#pragma unroll
  for (int k = 0; k < 4; ++k) sOH[t * LOH + k] = (live && tok == k) ? 1.f : 0.f;

  // ---- backward through the MLP
  float dh2[H2];
#pragma unroll
  for (int k = 0; k < H2; ++k) dh2[k] = 0.f;
#pragma unroll 1
  for (int o = 0; o < I; ++o) {
    const float g = live ? gfeats[(long long)v * ld + o] : 0.f;
    sG[t * LG + o] = g;
#pragma unroll
    for (int k = 0; k < H2; ++k) dh2[k] += sp[L::oW3t + k * I + o] * g;
  }
#pragma unroll
  for (int k = 0; k < H2; ++k) {
    dh2[k] = h2[k] > 0.f ? dh2[k] : 0.f;
    sD2[t * LD2 + k] = dh2[k];
  }
  float dh1[H1];
#pragma unroll
  for (int k = 0; k < H1; ++k) dh1[k] = 0.f;
#pragma unroll
  for (int j = 0; j < H2; ++j)
#pragma unroll
    for (int k = 0; k < H1; ++k) dh1[k] += sp[L::oW2t + k * H2 + j] * dh2[j];
#pragma unroll
  for (int k = 0; k < H1; ++k) {
    dh1[k] = h1[k] > 0.f ? dh1[k] : 0.f;
    sD1[t * LD1 + k] = dh1[k];
  }
  // gradient w.r.t. the position: de_k = sum_j W1[j][k] dh1[j];
  // d sin(f p)/dp = f cos(f p) = f e[6i+3+c],  d cos(f p)/dp = -f sin(f p) = -f e[6i+c]
  float gp[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    const float f = pe_freq(i);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float ds = 0.f, dc = 0.f;
#pragma unroll
      for (int j = 0; j < H1; ++j) {
        ds += sp[L::oW1t + (6 * i + c) * H1 + j] * dh1[j];
        dc += sp[L::oW1t + (6 * i + 3 + c) * H1 + j] * dh1[j];
      }
      gp[c] += f * (myE[6 * i + 3 + c] * ds - myE[6 * i + c] * dc);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < H1; ++j) d += sp[L::oW1t + (60 + c) * H1 + j] * dh1[j];
    gp[c] += d;
  }
  if (live) {
    gverts[3 * (long long)v + 0] = gp[0];
    gverts[3 * (long long)v + 1] = gp[1];
    gverts[3 * (long long)v + 2] = gp[2];
  }
  __syncthreads();

  // ---- workgroup reductions: every parameter gradient of the block is an outer-product sum over its T vertices,
  // out[a][b] = sum_r A[r][a] B[r][b].  Each 16 x 16 tile of each product is one chain of T/4 exact-fp32 MFMAs
  // (v_mfma_f32_16x16x4_f32, fixed order) fed straight from the LDS rows above; rows of padding vertices are zero in
  // G / D2 / D1 / OH, columns past an array's width read the neighbouring row and only reach discarded outputs.
  struct Job { const float *A; int lda, a0, na; const float *B; int ldb, b0; int kind; };
  const int lane = t & 63, wave = t >> 6, l16 = lane & 15, kq = lane >> 4;
  constexpr int NA3 = (I + 15) / 16, NB3 = (H2 + 1 + 15) / 16, NA2 = (H2 + 15) / 16, NB1 = 4;
  constexpr int J3 = NA3 * NB3, JE = NA3, J2 = NA2, J1 = NB1, NJOBS = J3 + JE + J2 + J1;
  for (int job = wave; job < NJOBS; job += T / 64) {
    Job jb;
    if (job < J3) jb = Job{sG, LG, (job / NB3) * 16, I, sH2, LH2, (job % NB3) * 16, 0};                 // dW3 | db3
    else if (job < J3 + JE) jb = Job{sG, LG, (job - J3) * 16, I, sOH, LOH, 0, 1};                        // dE
    else if (job < J3 + JE + J2) jb = Job{sD2, LD2, (job - J3 - JE) * 16, H2, sH1, LH1, 0, 2};           // dW2 | db2
    else jb = Job{sD1, LD1, 0, H1, sE, LE, (job - J3 - JE - J2) * 16, 3};                                // dW1 | db1
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *pa = jb.A + kq * jb.lda + jb.a0 + l16, *pb = jb.B + kq * jb.ldb + jb.b0 + l16;
#pragma unroll 8
    for (int st = 0; st < T / 4; ++st)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[st * 4 * jb.lda], pb[st * 4 * jb.ldb], acc, 0, 0, 0);
    const int col = jb.b0 + l16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = jb.a0 + kq * 4 + r;
      if (row >= jb.na) continue;
      const float val = acc[r];
      if (jb.kind == 0) {          // row = o, col = k | bias column
        if (col < H2) out[P::oW3 + row * H2 + col] = val;
        else if (col == H2) out[P::ob3 + row] = val;
      } else if (jb.kind == 1) {   // row = o, col = token
        if (col < 4) out[P::oE + col * I + row] = val;
      } else if (jb.kind == 2) {   // row = j, col = k | bias column
        if (col < H1) out[P::oW2 + row * H1 + col] = val;
        else if (col == H1) out[P::ob2 + row] = val;
      } else {                     // row = j, col = k (63 = bias column)
        if (col < 63) out[P::oW1 + row * 63 + col] = val;
        else if (col == 63) out[P::ob1 + row] = val;
      }
    }
  }
}

template <int I>
static size_t posenc_bwd_smem() {
  using P = PE<I>;
  constexpr int T = kPEBwdThreads;
  return (size_t)(PELds<I>::N + T * (64 + I + (2 * P::H2 + 1) + (2 * P::H1 + 1) + 4) + 64) * sizeof(float);
}

int launch_posenc_bwd(const float *verts, const float *mask, int m, int input_size, const float *params,
                      const float *gfeats, int ld, float *gverts, float *gparams, float *scratch, hipStream_t s) {
  if (input_size != 50) {
    set_error("posenc: fused kernel supports input_size == 50 only (got %d)", input_size);
    return -1;
  }
  const int nslab = posenc_num_slabs(m);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)posenc_bwd_kernel<50>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)posenc_bwd_smem<50>());
    attr_set = true;
  }
  A3VT_LAUNCH((posenc_bwd_kernel<50>), dim3(nslab), dim3(kPEBwdThreads), posenc_bwd_smem<50>(), s, verts, mask, m,
              params, gfeats, ld, gverts, scratch);
  A3VT_CHECK_LAUNCH();
  return launch_slab_reduce(scratch, nslab, PE<50>::N, PE<50>::N, gparams, s);
}

}  // namespace a3vt
