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
  // The 63-wide embedding is never held whole (63 + 12 + 25 live floats put the kernel at 256 VGPRs, one wave per SIMD):
  // each frequency's six values go straight into the first layer's accumulators, in the same k order as pe_mlp.
  float h1[P::H1], h2[P::H2];
#pragma unroll
  for (int j = 0; j < P::H1; ++j) h1[j] = sp[L::ob1 + j];
#pragma unroll 1
  for (int i = 0; i < 10; ++i) {
    const float f = pe_freq(i);
    float e6[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      e6[c] = sinf(f * p[c]);
      e6[3 + c] = cosf(f * p[c]);
    }
#pragma unroll
    for (int kk = 0; kk < 6; ++kk)
#pragma unroll
      for (int j = 0; j < P::H1; ++j) h1[j] += sp[L::oW1t + (6 * i + kk) * P::H1 + j] * e6[kk];
  }
#pragma unroll
  for (int kk = 0; kk < 3; ++kk)
#pragma unroll
    for (int j = 0; j < P::H1; ++j) h1[j] += sp[L::oW1t + (60 + kk) * P::H1 + j] * p[kk];
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
// Backward.  Workgroup = 128 vertices (256 threads).  The per-vertex work is a chain of small dense products over the
// block's vertices — the forward recompute (63 -> 12 -> 25), the back-propagation (50 -> 25 -> 12 -> 63) — and the
// parameter gradients are outer-product sums over the same vertices, so everything runs as 16 x 16 tiles of
// v_mfma_f32_16x16x4_f32 (exact fp32) on vertex rows parked in LDS; threads only compute sin / cos, the ReLU masks and
// the 63-term position gradient.  (The first version did all of it with per-thread scalar FMAs fed by LDS reads at
// two waves per CU: 445 us per call against ~70 us here.)
// LDS: weight images as B operands [k][n] with the bias as an extra k row against a 1 in the vertex row, zero rows
// where K is padded to a multiple of 4; vertex rows E|1, G, H1|1, H2|1, D2, D1, DE, one-hot token.
// ------------------------------------------------------------------------------------------------
constexpr int kPEBwdVerts = 128;
constexpr int kPEBwdThreads = 256;

int posenc_num_slabs(int m) { return cdiv(m, kPEBwdVerts); }

template <int I>
struct PEB {
  using P = PE<I>;
  static constexpr int T = kPEBwdVerts, H1 = P::H1, H2 = P::H2;
  // B-operand weight images
  static constexpr int oB1 = 0;                      // [64][H1]   rows 0..62 = W1^T, row 63 = b1          (E|1 -> h1)
  static constexpr int oB2 = oB1 + 64 * H1;          // [16][H2]   rows 0..11 = W2^T, row 12 = b2, rest 0  (H1|1 -> h2)
  static constexpr int oB3 = oB2 + 16 * H2;          // [52][H2]   = W3 (o, k), rows 50, 51 = 0            (G -> dh2)
  static constexpr int oB4 = oB3 + 52 * H2;          // [28][H1]   = W2 (j, k), rows 25..27 = 0            (D2 -> dh1)
  static constexpr int oB5 = oB4 + 28 * H1;          // [12][63]   = W1 (j, k)                             (D1 -> de)
  static constexpr int NW = oB5 + H1 * 63 + 16;      // + slack for the 16-wide reads of the last row
  // vertex rows
  static constexpr int LE = 64, LG = 52, LH1 = 16, LH2 = 28, LD2 = 28, LD1 = 16, LDE = 64, LOH = 4;
  static constexpr int oE = NW, oG = oE + T * LE, oH1 = oG + T * LG, oH2 = oH1 + T * LH1, oD2 = oH2 + T * LH2,
                       oD1 = oD2 + T * LD2, oDE = oD1 + T * LD1, oOH = oDE + T * LDE, N = oOH + T * LOH + 64;
  static_assert(I == 50, "layout written for I = 50 (H1 = 12, H2 = 25)");
};

// 16 x 16 tile of A[rows m0.., K] * B[K, cols n0..]: A rows in LDS (stride lda), B rows in LDS (stride ldb), K % 4 == 0.
// The lane ends up with column n0 + l16 of rows m0 + 4 kq .. + 3.
__device__ __forceinline__ f32x4 pe_tile(const float *A, int lda, int m0, const float *B, int ldb, int n0, int K, int l16,
                                         int kq) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float *pa = A + (m0 + l16) * lda + kq, *pb = B + kq * ldb + n0 + l16;
#pragma unroll 4
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[k], pb[k * ldb], acc, 0, 0, 0);
  return acc;
}

template <int I>
__global__ __launch_bounds__(kPEBwdThreads) void posenc_bwd_kernel(const float *__restrict__ verts,
                                                                   const float *__restrict__ mask, int m,
                                                                   const float *__restrict__ params,
                                                                   const float *__restrict__ gfeats, int ld,
                                                                   float *__restrict__ gverts,
                                                                   float *__restrict__ slab) {
  using P = PE<I>;
  using B = PEB<I>;
  constexpr int T = B::T, H1 = B::H1, H2 = B::H2;
  extern __shared__ float smem[];
  float *sE = smem + B::oE, *sG = smem + B::oG, *sH1 = smem + B::oH1, *sH2 = smem + B::oH2, *sD2 = smem + B::oD2,
        *sD1 = smem + B::oD1, *sDE = smem + B::oDE, *sOH = smem + B::oOH;
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6, l16 = lane & 15, kq = lane >> 4;
  float *out = slab + (size_t)blockIdx.x * P::N;

  // ---- phase 0: weight images (all threads) and the vertex rows E|1, G, one-hot, constant columns (threads < T)
  for (int i = t; i < B::NW; i += kPEBwdThreads) smem[i] = 0.f;
  __syncthreads();
  for (int i = t; i < H1 * 63; i += kPEBwdThreads) {  // W1 (j, k): transposed into B1, as is into B5
    const int j = i / 63, k = i - j * 63;
    const float w = params[P::oW1 + i];
    smem[B::oB1 + k * H1 + j] = w;
    smem[B::oB5 + i] = w;
  }
  for (int i = t; i < H1; i += kPEBwdThreads) smem[B::oB1 + 63 * H1 + i] = params[P::ob1 + i];
  for (int i = t; i < H2 * H1; i += kPEBwdThreads) {  // W2 (j, k)
    const int j = i / H1, k = i - j * H1;
    const float w = params[P::oW2 + i];
    smem[B::oB2 + k * H2 + j] = w;
    smem[B::oB4 + i] = w;
  }
  for (int i = t; i < H2; i += kPEBwdThreads) smem[B::oB2 + H1 * H2 + i] = params[P::ob2 + i];
  for (int i = t; i < I * H2; i += kPEBwdThreads) smem[B::oB3 + i] = params[P::oW3 + i];  // W3 (o, k)
  {  // embedding: two threads per vertex, five frequencies each
    const int tv = t >> 1, half = t & 1;
    const int v = blockIdx.x * T + tv;
    float p[3] = {0.f, 0.f, 0.f};
    if (v < m) {
      p[0] = verts[3 * (long long)v];
      p[1] = verts[3 * (long long)v + 1];
      p[2] = verts[3 * (long long)v + 2];
    }
    float *myE = sE + tv * B::LE;
#pragma unroll 1
    for (int i = half * 5; i < half * 5 + 5; ++i) {
      const float f = pe_freq(i);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        myE[6 * i + c] = sinf(f * p[c]);
        myE[6 * i + 3 + c] = cosf(f * p[c]);
      }
    }
    if (half == 0) {
      myE[60] = p[0];
      myE[61] = p[1];
      myE[62] = p[2];
      myE[63] = 1.f;
    }
  }
  // upstream gradient rows: coalesced 16-byte pieces of the [T][ld] tile (columns >= I are zero by contract when ld >= 52)
  for (int i = t; i < T * (B::LG / 4); i += kPEBwdThreads) {
    const int r = i / (B::LG / 4), c4 = i - r * (B::LG / 4);
    const int v = blockIdx.x * T + r;
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (v < m) {
      if (c4 * 4 + 3 < ld) {
        g = *reinterpret_cast<const f32x4 *>(gfeats + (long long)v * ld + c4 * 4);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c4 * 4 + k < ld) g[k] = gfeats[(long long)v * ld + c4 * 4 + k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c4 * 4 + k >= I) g[k] = 0.f;
    }
    *reinterpret_cast<f32x4 *>(sG + r * B::LG + c4 * 4) = g;
  }
  if (t < T) {
    const int v = blockIdx.x * T + t;
    const bool live = v < m;
    int tok = live ? (int)mask[v] : 0;
    tok = tok < 0 ? 0 : (tok > 3 ? 3 : tok);
#pragma unroll
    for (int k = 0; k < 4; ++k) sOH[t * B::LOH + k] = (live && tok == k) ? 1.f : 0.f;
    sH1[t * B::LH1 + H1] = 1.f;   // bias column; the other pad columns stay finite zeros
#pragma unroll
    for (int k = H1 + 1; k < B::LH1; ++k) sH1[t * B::LH1 + k] = 0.f;
    sH2[t * B::LH2 + H2] = 1.f;
#pragma unroll
    for (int k = H2 + 1; k < B::LH2; ++k) sH2[t * B::LH2 + k] = 0.f;
#pragma unroll
    for (int k = H2; k < B::LD2; ++k) sD2[t * B::LD2 + k] = 0.f;
#pragma unroll
    for (int k = H1; k < B::LD1; ++k) sD1[t * B::LD1 + k] = 0.f;
  }
  __syncthreads();

  constexpr int NW_ = kPEBwdThreads / 64, MT = T / 16;
  // ---- phase 1: h1 = relu(E|1 . [W1^T; b1])
  for (int tile = wave; tile < MT; tile += NW_) {
    const f32x4 acc = pe_tile(sE, B::LE, tile * 16, smem + B::oB1, H1, 0, 64, l16, kq);
    if (l16 < H1)
#pragma unroll
      for (int r = 0; r < 4; ++r) sH1[(tile * 16 + kq * 4 + r) * B::LH1 + l16] = acc[r] > 0.f ? acc[r] : 0.f;
  }
  __syncthreads();
  // ---- phase 2: h2 = relu(H1|1 . [W2^T; b2])
  for (int tile = wave; tile < MT * 2; tile += NW_) {
    const int mt = tile >> 1, n0 = (tile & 1) * 16;
    const f32x4 acc = pe_tile(sH1, B::LH1, mt * 16, smem + B::oB2, H2, n0, 16, l16, kq);
    if (n0 + l16 < H2)
#pragma unroll
      for (int r = 0; r < 4; ++r) sH2[(mt * 16 + kq * 4 + r) * B::LH2 + n0 + l16] = acc[r] > 0.f ? acc[r] : 0.f;
  }
  __syncthreads();
  // ---- phase 3: dh2 = (G . W3) * (h2 > 0)
  for (int tile = wave; tile < MT * 2; tile += NW_) {
    const int mt = tile >> 1, n0 = (tile & 1) * 16;
    const f32x4 acc = pe_tile(sG, B::LG, mt * 16, smem + B::oB3, H2, n0, 52, l16, kq);
    if (n0 + l16 < H2)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + kq * 4 + r;
        sD2[row * B::LD2 + n0 + l16] = sH2[row * B::LH2 + n0 + l16] > 0.f ? acc[r] : 0.f;
      }
  }
  __syncthreads();
  // ---- phase 4: dh1 = (D2 . W2) * (h1 > 0)
  for (int tile = wave; tile < MT; tile += NW_) {
    const f32x4 acc = pe_tile(sD2, B::LD2, tile * 16, smem + B::oB4, H1, 0, 28, l16, kq);
    if (l16 < H1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tile * 16 + kq * 4 + r;
        sD1[row * B::LD1 + l16] = sH1[row * B::LH1 + l16] > 0.f ? acc[r] : 0.f;
      }
  }
  __syncthreads();
  // ---- phase 5: de = D1 . W1
  for (int tile = wave; tile < MT * 4; tile += NW_) {
    const int mt = tile >> 2, n0 = (tile & 3) * 16;
    const f32x4 acc = pe_tile(sD1, B::LD1, mt * 16, smem + B::oB5, 63, n0, H1, l16, kq);
#pragma unroll
    for (int r = 0; r < 4; ++r) sDE[(mt * 16 + kq * 4 + r) * B::LDE + n0 + l16] = acc[r];
  }
  __syncthreads();
  // ---- phase 6: gradient w.r.t. the position.  d sin(f p)/dp = f cos(f p) = f e[6i+3+c], d cos(f p)/dp = -f e[6i+c]
  if (t < T) {
    const int v = blockIdx.x * T + t;
    const float *e = sE + t * B::LE, *de = sDE + t * B::LDE;
    float gp[3] = {de[60], de[61], de[62]};
#pragma unroll 1
    for (int i = 0; i < 10; ++i) {
      const float f = pe_freq(i);
#pragma unroll
      for (int c = 0; c < 3; ++c) gp[c] += f * (e[6 * i + 3 + c] * de[6 * i + c] - e[6 * i + c] * de[6 * i + 3 + c]);
    }
    if (v < m) {
      gverts[3 * (long long)v + 0] = gp[0];
      gverts[3 * (long long)v + 1] = gp[1];
      gverts[3 * (long long)v + 2] = gp[2];
    }
  }

  // ---- phase 7: parameter gradients of the block, out[a][b] = sum_r A[r][a] B[r][b] over its T vertex rows: each
  // 16 x 16 tile of each product is one chain of T/4 MFMAs (fixed order).  Rows of padding vertices are zero in
  // G / D2 / D1 / OH; columns past an array's width read the neighbouring row and only reach discarded outputs.
  struct Job { const float *A; int lda, a0, na; const float *Bm; int ldb, b0; int kind; };
  constexpr int NA3 = (I + 15) / 16, NB3 = (H2 + 1 + 15) / 16, NA2 = (H2 + 15) / 16, NB1 = 4;
  constexpr int J3 = NA3 * NB3, JE = NA3, J2 = NA2, J1 = NB1, NJOBS = J3 + JE + J2 + J1;
  for (int job = wave; job < NJOBS; job += NW_) {
    Job jb;
    if (job < J3) jb = Job{sG, B::LG, (job / NB3) * 16, I, sH2, B::LH2, (job % NB3) * 16, 0};              // dW3 | db3
    else if (job < J3 + JE) jb = Job{sG, B::LG, (job - J3) * 16, I, sOH, B::LOH, 0, 1};                    // dE
    else if (job < J3 + JE + J2) jb = Job{sD2, B::LD2, (job - J3 - JE) * 16, H2, sH1, B::LH1, 0, 2};       // dW2 | db2
    else jb = Job{sD1, B::LD1, 0, H1, sE, B::LE, (job - J3 - JE - J2) * 16, 3};                            // dW1 | db1
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float *pa = jb.A + kq * jb.lda + jb.a0 + l16, *pb = jb.Bm + kq * jb.ldb + jb.b0 + l16;
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
  return (size_t)PEB<I>::N * sizeof(float);
}

int launch_posenc_bwd(const float *verts, const float *mask, int m, int input_size, const float *params,
                      const float *gfeats, int ld, float *gverts, float *gparams, float *scratch, hipStream_t s) {
  if (input_size != 50) {
    set_error("posenc: fused kernel supports input_size == 50 only (got %d)", input_size);
    return -1;
  }
  const int nslab = posenc_num_slabs(m);
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)posenc_bwd_kernel<50>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)posenc_bwd_smem<50>());
  });
  A3VT_LAUNCH((posenc_bwd_kernel<50>), dim3(nslab), dim3(kPEBwdThreads), posenc_bwd_smem<50>(), s, verts, mask, m,
              params, gfeats, ld, gverts, scratch);
  A3VT_CHECK_LAUNCH();
  return launch_slab_reduce(scratch, nslab, PE<50>::N, PE<50>::N, gparams, s);
}

}  // namespace a3vt
