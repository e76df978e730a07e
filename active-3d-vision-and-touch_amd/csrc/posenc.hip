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
// Backward (round 4 form).  The per-vertex work is a chain of small dense products — the forward recompute (63 -> 12 -> 25),
// the back-propagation (50 -> 25 -> 12 -> 63) — and the parameter gradients are outer-product sums over vertices, so
// everything runs as 16 x 16 tiles of v_mfma_f32_16x16x4_f32 (exact fp32); threads only compute sin / cos, the ReLU masks
// and the 63-term position gradient.
// A WAVE owns 16 vertices at a time and walks the whole chain on them alone, in its own 15 KB of LDS: no workgroup barrier
// inside the loop.  An MFMA of this shape fed by two ds_read_b32 is bound by the LDS port (512 B per 32 matrix cycles and
// SIMD = half the port before any bank conflict), so
//   * the WEIGHT operands of the five chain products (69 k-steps) live in the wave's registers for the whole launch, read
//     once from the packed parameters: those MFMAs take one LDS read each;
//   * the parameter-gradient tiles (14 accumulators) stay in registers over all the wave's vertex tiles; their products run
//     over the tile's vertices in the order (lane group kq, step st) -> vertex 4 kq + st, which is how an MFMA result lies
//     in registers: dh2 / dh1 enter them without touching LDS and the other operand rows sit 4 rows (= 16 banks) apart;
//   * the next tile's inputs are requested before the current tile's chain.
// Persistent grid, at most 512 workgroups of 4 waves (two per CU); a workgroup's four partials are summed through LDS in a
// fixed order into its slab.  (Round 3: one 128-vertex workgroup per CU with a barrier between phases, 154 KB of LDS, a slab
// per 128 vertices — 1281 workgroups on 256 CUs ran as six rounds of ~30 us: 182 us per call.  First version: per-thread
// scalar FMAs, 445 us.)
// Against the round-3 kernel (tools/experiments/posenc_bwd_bits.py): the forward is untouched; grad_verts agrees to 2e-6 of
// its largest element (same products and chains, but the compiler contracts the 63-term position sum differently in the
// unrolled loop), the parameter gradients — summed in a different, still fixed order — to 3e-7.
// Per wave in LDS: the vertex rows E|1, G (later dE), H1|1, H2|1|one-hot token, D2, D1.  Row strides are = 4 mod 16 floats:
// the A-operand read of a chain tile (lane = row l16, k = kq) touches 64 different banks, and so do the row reads of the
// outer products.
// ------------------------------------------------------------------------------------------------
constexpr int kPEBwdThreads = 256;
constexpr int kPEBwdWaves = kPEBwdThreads / 64;
constexpr int kPEBwdMaxWgs = 512;

int posenc_num_slabs(int m) {
  const int wgs = cdiv(cdiv(m, 16), kPEBwdWaves);
  return wgs < 1 ? 1 : wgs > kPEBwdMaxWgs ? kPEBwdMaxWgs : wgs;
}

template <int I>
struct PEB {
  using P = PE<I>;
  static constexpr int H1 = P::H1, H2 = P::H2;
  // a wave's vertex rows (16 vertices)
  static constexpr int LE = 68, LG = 68, LH1 = 20, LH2 = 36, LD2 = 36, LD1 = 20;
  static constexpr int oE = 0, oG = oE + 16 * LE, oH1 = oG + 16 * LG, oH2 = oH1 + 16 * LH1, oD2 = oH2 + 16 * LH2,
                       oD1 = oD2 + 16 * LD2, WAVE = oD1 + 16 * LD1;
  static constexpr int N = kPEBwdWaves * WAVE;
  static_assert(I == 50, "layout written for I = 50 (H1 = 12, H2 = 25)");
  static_assert(WAVE >= P::N && WAVE % 4 == 0, "a wave's partial parameter gradients are parked in its own rows at the end");
};

// LDS operations of one wave execute in order; this only keeps the COMPILER from moving a phase's reads above the
// previous phase's writes of other lanes
__device__ __forceinline__ void pe_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// 16 x 16 tile of A[16 rows, 4 KS] * W: A rows in LDS (stride lda; lane = row l16, k = kq + 4 j), W's k-steps in registers.
// The lane ends up with its column of rows 4 kq .. + 3.
template <int KS>
__device__ __forceinline__ f32x4 pe_chain(const float *A, int lda, const float (&w)[KS], int l16, int kq) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float *pa = A + l16 * lda + kq;
#pragma unroll
  for (int j = 0; j < KS; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * j], w[j], acc, 0, 0, 0);
  return acc;
}

template <int I>
__global__ __launch_bounds__(kPEBwdThreads, 2) void posenc_bwd_kernel(const float *__restrict__ verts,
                                                                      const float *__restrict__ mask, int m,
                                                                      const float *__restrict__ params,
                                                                      const float *__restrict__ gfeats, int ld,
                                                                      float *__restrict__ gverts,
                                                                      float *__restrict__ slab) {
  using P = PE<I>;
  using B = PEB<I>;
  constexpr int H1 = B::H1, H2 = B::H2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, kq = lane >> 4;
  float *wv = smem + wave * B::WAVE;
  float *sE = wv + B::oE, *sG = wv + B::oG, *sH1 = wv + B::oH1, *sH2 = wv + B::oH2, *sD2 = wv + B::oD2, *sD1 = wv + B::oD1;
  float *sDE = sG;   // the position-embedding gradient takes G's rows once the products on G are done

  // ---- weight operands, B[k = kq + 4 j][n = n0 + l16] of each chain product; rows / columns of padding are zeros
  float w1[16];       // [W1^T; b1]  (E|1 -> h1)      k < 63: W1[n][k], k = 63: b1[n]
  float w2[2][4];     // [W2^T; b2]  (H1|1 -> h2)     k < 12: W2[n][k], k = 12: b2[n]
  float w3[2][13];    // W3 (o, k)   (G -> dh2)       B[o][n] = W3[o][n], o < 50
  float w4[7];        // W2 (j, k)   (D2 -> dh1)      B[j][n] = W2[j][n], j < 25
  float w5[4][3];     // W1 (j, k)   (D1 -> de)       B[j][n] = W1[j][n], j < 12, n < 63
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = kq + 4 * j;
    w1[j] = l16 < H1 ? (k < 63 ? params[P::oW1 + l16 * 63 + k] : params[P::ob1 + l16]) : 0.f;
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = n * 16 + l16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = kq + 4 * j;
      w2[n][j] = col < H2 && k <= H1 ? (k < H1 ? params[P::oW2 + col * H1 + k] : params[P::ob2 + col]) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const int o = kq + 4 * j;
      w3[n][j] = col < H2 && o < I ? params[P::oW3 + o * H2 + col] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int r = kq + 4 * j;
    w4[j] = l16 < H1 && r < H2 ? params[P::oW2 + r * H1 + l16] : 0.f;
  }
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int col = n * 16 + l16;
#pragma unroll
    for (int j = 0; j < 3; ++j) w5[n][j] = col < 63 ? params[P::oW1 + (kq + 4 * j) * 63 + col] : 0.f;
  }
  // this wave's rows zeroed (pad columns stay zero: nothing below writes them), then the bias columns
  for (int i = lane; i < B::WAVE / 4; i += 64) reinterpret_cast<f32x4 *>(wv)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  pe_wave_sync();
  if (kq == 0) {
    sE[l16 * B::LE + 63] = 1.f;
    sH1[l16 * B::LH1 + H1] = 1.f;
    sH2[l16 * B::LH2 + H2] = 1.f;
  }

  // parameter-gradient tiles of this wave, summed over all its vertex tiles
  f32x4 aW3[4][2], aW2[2], aW1[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) aW3[a][0] = aW3[a][1] = aW1[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  aW2[0] = aW2[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ntile = (m + 15) >> 4, nwave = gridDim.x * kPEBwdWaves;
  // a tile's inputs: the 16 x 52 gradient rows as 208 16-byte pieces (four per lane; columns >= I are zeroed), the
  // lane's vertex l16 (its position and token)
  struct Inputs { f32x4 g[4]; float p[3]; int tok; bool live; };
  auto fetch = [&](int tile) {
    Inputs in;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = lane + 64 * j, r = i / 13, c4 = i - r * 13;
      const long long v = (long long)tile * 16 + r;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (i < 208 && v < m) {
        if (c4 * 4 + 3 < ld) {
          g = *reinterpret_cast<const f32x4 *>(gfeats + v * ld + c4 * 4);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c4 * 4 + k < ld) g[k] = gfeats[v * ld + c4 * 4 + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (c4 * 4 + k >= I) g[k] = 0.f;
      }
      in.g[j] = g;
    }
    const long long v = (long long)tile * 16 + l16;
    in.live = v < m;
    in.p[0] = in.live ? verts[3 * v] : 0.f;
    in.p[1] = in.live ? verts[3 * v + 1] : 0.f;
    in.p[2] = in.live ? verts[3 * v + 2] : 0.f;
    int tok = in.live ? (int)mask[v] : 0;
    in.tok = tok < 0 ? 0 : (tok > 3 ? 3 : tok);
    return in;
  };

  int tile = blockIdx.x * kPEBwdWaves + wave;
  Inputs in;
  if (tile < ntile) in = fetch(tile);
#ifdef A3VT_DBG_PE_NOLOOP   // timing-only: set-up and the final reduction alone
  tile = ntile;
#endif
  for (; tile < ntile; tile += nwave) {
    // ---- park the inputs: G rows, the embedding E (lane = vertex l16, a quarter of the 30 (frequency, axis) pairs), token
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = lane + 64 * j, r = i / 13, c4 = i - r * 13;
      if (i < 208) *reinterpret_cast<f32x4 *>(sG + r * B::LG + c4 * 4) = in.g[j];
    }
    {
      float *myE = sE + l16 * B::LE;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = kq + 4 * u;
        if (idx < 30) {
          const int i = idx / 3, c = idx - 3 * i;
          // pe_freq(i) for a runtime i: the same double product, rounded once
          const float f = i == 0 ? (float)3.141592653589793 : (float)(3.141592653589793 * 2.0 * (double)i);
          const float pc = c == 0 ? in.p[0] : c == 1 ? in.p[1] : in.p[2];
#ifdef A3VT_DBG_PE_NOSINCOS   // timing-only
          myE[6 * i + c] = f * pc;
          myE[6 * i + 3 + c] = f + pc;
#else
          myE[6 * i + c] = sinf(f * pc);
          myE[6 * i + 3 + c] = cosf(f * pc);
#endif
        }
      }
      if (kq < 3) myE[60 + kq] = kq == 0 ? in.p[0] : kq == 1 ? in.p[1] : in.p[2];
      sH2[l16 * B::LH2 + H2 + 1 + kq] = (in.live && in.tok == kq) ? 1.f : 0.f;   // one-hot token behind the bias column
    }
    const int vtile = tile;
    if (tile + nwave < ntile) in = fetch(tile + nwave);   // the next tile's inputs travel during this tile's chain
    pe_wave_sync();

    // ---- h1 = relu(E|1 . [W1^T; b1])
    bool h1pos[4];
    {
      const f32x4 acc = pe_chain<16>(sE, B::LE, w1, l16, kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h1pos[r] = acc[r] > 0.f;
        if (l16 < H1) sH1[(kq * 4 + r) * B::LH1 + l16] = h1pos[r] ? acc[r] : 0.f;
      }
    }
    pe_wave_sync();
    // ---- h2 = relu(H1|1 . [W2^T; b2])
    bool h2pos[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f32x4 acc = pe_chain<4>(sH1, B::LH1, w2[n], l16, kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h2pos[n][r] = acc[r] > 0.f;
        if (n * 16 + l16 < H2) sH2[(kq * 4 + r) * B::LH2 + n * 16 + l16] = h2pos[n][r] ? acc[r] : 0.f;
      }
    }
    // ---- dh2 = (G . W3) * (h2 > 0): kept in registers (columns >= 25 are exact zeros: zero weights), rows for the next product
    f32x4 d2[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f32x4 acc = pe_chain<13>(sG, B::LG, w3[n], l16, kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2[n][r] = h2pos[n][r] ? acc[r] : 0.f;
        sD2[(kq * 4 + r) * B::LD2 + n * 16 + l16] = d2[n][r];
      }
    }
    pe_wave_sync();
    // The outer products run over the tile's vertices as (kq, st) -> vertex 4 kq + st: operand rows 4 kq + st of G / H2 /
    // H1 / E from LDS, dh2 / dh1 straight from the registers above.
    // ---- dW3 | db3 | dE += G^T . [H2 | 1 | one-hot]
#ifndef A3VT_DBG_PE_NOOUTER
    {
      float hb[2][4];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int st = 0; st < 4; ++st) hb[n][st] = sH2[(kq * 4 + st) * B::LH2 + n * 16 + l16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        float ga[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) ga[st] = sG[(kq * 4 + st) * B::LG + a * 16 + l16];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int st = 0; st < 4; ++st) aW3[a][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[st], hb[n][st], aW3[a][n], 0, 0, 0);
      }
    }
#endif
    // ---- dh1 = (D2 . W2) * (h1 > 0)
    f32x4 d1;
    {
      const f32x4 acc = pe_chain<7>(sD2, B::LD2, w4, l16, kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d1[r] = h1pos[r] ? acc[r] : 0.f;
        sD1[(kq * 4 + r) * B::LD1 + l16] = d1[r];
      }
    }
    // ---- dW2 | db2 += D2^T . [H1 | 1]
#ifndef A3VT_DBG_PE_NOOUTER
    {
      float hb[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) hb[st] = sH1[(kq * 4 + st) * B::LH1 + l16];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int st = 0; st < 4; ++st) aW2[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(d2[a][st], hb[st], aW2[a], 0, 0, 0);
    }
#endif
    pe_wave_sync();
    // ---- de = D1 . W1 (into G's rows: every product on G has been issued, LDS runs in order)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const f32x4 acc = pe_chain<3>(sD1, B::LD1, w5[n], l16, kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) sDE[(kq * 4 + r) * B::LG + n * 16 + l16] = acc[r];
    }
    // ---- dW1 | db1 += D1^T . [E | 1]
#ifndef A3VT_DBG_PE_NOOUTER
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float eb[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) eb[st] = sE[(kq * 4 + st) * B::LE + n * 16 + l16];
#pragma unroll
      for (int st = 0; st < 4; ++st) aW1[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(d1[st], eb[st], aW1[n], 0, 0, 0);
    }
#endif
    pe_wave_sync();
    // ---- gradient w.r.t. the position (lane = vertex l16, axis kq).  d sin(f p)/dp = f cos(f p) = f e[6i+3+c],
    // d cos(f p)/dp = -f e[6i+c]
    if (kq < 3) {
      const long long v = (long long)vtile * 16 + l16;
      const float *e = sE + l16 * B::LE, *de = sDE + l16 * B::LG;
      float gp = de[60 + kq];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const float f = pe_freq(i);
        gp += f * (e[6 * i + 3 + kq] * de[6 * i + kq] - e[6 * i + kq] * de[6 * i + 3 + kq]);
      }
      if (v < m) gverts[3 * v + kq] = gp;
    }
    pe_wave_sync();
  }

  // ---- the wave's partials into its own rows in the packed parameter order, then the workgroup's four in a fixed order
  pe_wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = kq * 4 + r;
#pragma unroll
    for (int a = 0; a < 4; ++a) {          // row = output channel o, column = k | bias | token
      const int o = a * 16 + row;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int col = n * 16 + l16;
        if (o >= I) continue;
        if (col < H2) wv[P::oW3 + o * H2 + col] = aW3[a][n][r];
        else if (col == H2) wv[P::ob3 + o] = aW3[a][n][r];
        else if (col < H2 + 5) wv[P::oE + (col - H2 - 1) * I + o] = aW3[a][n][r];
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {          // row = j of W2 (j, k), column = k | bias
      const int j = a * 16 + row;
      if (j >= H2) continue;
      if (l16 < H1) wv[P::oW2 + j * H1 + l16] = aW2[a][r];
      else if (l16 == H1) wv[P::ob2 + j] = aW2[a][r];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {          // row = j of W1 (j, k), column = k | bias (63)
      const int col = n * 16 + l16;
      if (row >= H1) continue;
      if (col < 63) wv[P::oW1 + row * 63 + col] = aW1[n][r];
      else wv[P::ob1 + row] = aW1[n][r];
    }
  }
  __syncthreads();
  float *out = slab + (size_t)blockIdx.x * P::N;
  for (int i = t; i < P::N; i += kPEBwdThreads) {
    float s = smem[i];
#pragma unroll
    for (int w = 1; w < kPEBwdWaves; ++w) s += smem[w * B::WAVE + i];
    out[i] = s;
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
  if (m <= 0) {
    set_error("posenc: m=%d", m);
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
