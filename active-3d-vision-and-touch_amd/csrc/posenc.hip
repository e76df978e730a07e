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
  float *sp = smem;                  // parameters (transposed image)
  float *sE = sp + L::N;             // [T][63]
  float *sG = sE + T * 63;           // [T][I]   gout
  float *sH2 = sG + T * I;           // [T][H2]
  float *sD2 = sH2 + T * H2;         // [T][H2]  dh2
  float *sH1 = sD2 + T * H2;         // [T][H1]
  float *sD1 = sH1 + T * H1;         // [T][H1]  dh1
  int *stok = reinterpret_cast<int *>(sD1 + T * H1);
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
  float *myE = sE + t * 63;
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
    sH1[t * H1 + j] = h1[j];
  }
#pragma unroll
  for (int j = 0; j < H2; ++j) h2[j] = sp[L::ob2 + j];
#pragma unroll
  for (int k = 0; k < H1; ++k)
#pragma unroll
    for (int j = 0; j < H2; ++j) h2[j] += sp[L::oW2t + k * H2 + j] * h1[k];
#pragma unroll
  for (int j = 0; j < H2; ++j) {
    h2[j] = h2[j] > 0.f ? h2[j] : 0.f;
    sH2[t * H2 + j] = h2[j];
  }
  int tok = live ? (int)mask[v] : 0;
  tok = tok < 0 ? 0 : (tok > 3 ? 3 : tok);
  stok[t] = live ? tok : -1;

  // ---- backward through the MLP
  float dh2[H2];
#pragma unroll
  for (int k = 0; k < H2; ++k) dh2[k] = 0.f;
#pragma unroll 1
  for (int o = 0; o < I; ++o) {
    const float g = live ? gfeats[(long long)v * ld + o] : 0.f;
    sG[t * I + o] = g;
#pragma unroll
    for (int k = 0; k < H2; ++k) dh2[k] += sp[L::oW3t + k * I + o] * g;
  }
#pragma unroll
  for (int k = 0; k < H2; ++k) {
    dh2[k] = h2[k] > 0.f ? dh2[k] : 0.f;
    sD2[t * H2 + k] = dh2[k];
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
    sD1[t * H1 + k] = dh1[k];
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

  // ---- workgroup reductions of the outer products (fixed order over the 128 vertices)
  for (int idx = t; idx < I * H2; idx += T) {  // dW3[o][k] = sum_v gout[v][o] h2[v][k]
    const int o = idx / H2, k = idx % H2;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < T; ++r) s += sG[r * I + o] * sH2[r * H2 + k];
    out[P::oW3 + idx] = s;
  }
  for (int o = t; o < I; o += T) {  // db3, dE
    float s = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int r = 0; r < T; ++r) {
      const float g = sG[r * I + o];
      const int tk = stok[r];
      s += g;
      s0 += tk == 0 ? g : 0.f;
      s1 += tk == 1 ? g : 0.f;
      s2 += tk == 2 ? g : 0.f;
      s3 += tk == 3 ? g : 0.f;
    }
    out[P::ob3 + o] = s;
    out[P::oE + 0 * I + o] = s0;
    out[P::oE + 1 * I + o] = s1;
    out[P::oE + 2 * I + o] = s2;
    out[P::oE + 3 * I + o] = s3;
  }
  for (int idx = t; idx < H2 * H1; idx += T) {  // dW2[j][k] = sum_v dh2[v][j] h1[v][k]
    const int j = idx / H1, k = idx % H1;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < T; ++r) s += sD2[r * H2 + j] * sH1[r * H1 + k];
    out[P::oW2 + idx] = s;
  }
  for (int j = t; j < H2; j += T) {
    float s = 0.f;
    for (int r = 0; r < T; ++r) s += sD2[r * H2 + j];
    out[P::ob2 + j] = s;
  }
  for (int idx = t; idx < H1 * 63; idx += T) {  // dW1[j][k] = sum_v dh1[v][j] e[v][k]
    const int j = idx / 63, k = idx % 63;
    float s = 0.f;
#pragma unroll 8
    for (int r = 0; r < T; ++r) s += sD1[r * H1 + j] * sE[r * 63 + k];
    out[P::oW1 + idx] = s;
  }
  for (int j = t; j < H1; j += T) {
    float s = 0.f;
    for (int r = 0; r < T; ++r) s += sD1[r * H1 + j];
    out[P::ob1 + j] = s;
  }
}

template <int I>
static size_t posenc_bwd_smem() {
  using P = PE<I>;
  constexpr int T = kPEBwdThreads;
  return (size_t)(PELds<I>::N + T * (63 + I + 2 * P::H2 + 2 * P::H1)) * sizeof(float) + T * sizeof(int);
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
