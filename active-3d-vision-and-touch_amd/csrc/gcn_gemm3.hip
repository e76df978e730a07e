// gcn_gemm3.hip — gemm mode 3 ("fp32x3"): the fp32 per-vertex products of the hidden layers on the bf16 matrix pipe.
//
//   C[M][n] = A[M][K] * Bt[n][K]^T,  A / C fp32 in HBM exactly as in mode 0,  288 < K <= 320, 288 < n <= 304
//   (torch.matmul(features, W) of GCN_layer.forward, vision/model.py:352, and autograd's dX = dZ W^T)
//
// Every fp32 operand x is cut into three bf16 pieces with  x == hi + mid + lo  bit for bit (split3 below), and the
// product is formed from the six partial products down to 2^-16 of the largest one,
//     a b ~= ah bh + ah bm + am bh + ah bl + al bh + am bm        (dropped: am bl, al bm, al bl <= 2^-23 |a b|),
// each a v_mfma_f32_16x16x32_bf16 into the same fp32 accumulator: 6 x 16 cycles per 16 x 16 x 32 block where the exact
// v_mfma_f32_16x16x4_f32 path takes 8 x 32 — 3/8 of the matrix-pipe cycles, at fp32-level error (the products of bf16
// pieces are exact in fp32; what remains is the accumulation, rounded once per 32 k instead of once per k; measured
// against the fp64 oracle in tests/test_gpu_fp32x3.py and tabulated in DESIGN.md).  NOT bit-identical to mode 0, which
// stays the default and the parity mode.
//
// Kernel shape (rowgemm3_kernel): the round structure of rowgemm_kernel<19, EPI, ..., ADIRECT> (gcn_gemm.hip) —
// persistent 8-wave workgroups, 16-row tiles dealt evenly, two tiles x 19 column tiles of accumulators per wave and round.
//   A : each lane loads the 32 bytes that are ITS operand elements of a 32-wide k chunk (row l16, k = 8 q .. 8 q + 7)
//       straight into registers, two global_load_dwordx4 (row-major rows or the quad-major planes of the hybrid layout),
//       and splits them once (11 VALU per pair of elements): three 16-byte MFMA operands per tile.
//   Bt: pre-split ONCE per stack call into three bf16 images [320][320] (weight_images3_kernel); a chunk stages
//       3 x 19 pieces of 16 rows x 64 bytes through a 2-stage LDS ring with LDS-DMA (global_load_lds_dwordx4), bank
//       swizzled as in rowgemm_kernel; a fragment read is one ds_read_b128 = the operand, three per column tile serve
//       twelve MFMAs.
//   one s_barrier per chunk (25 per launch at M = 163,968 where the fp32 kernel has 47.5), the chunk after next is never
//   needed: a chunk is 228 MFMAs = 3,650 matrix-pipe cycles per wave.
// Epilogues: those of rowgemm_kernel, value for value (forward: raw aggregated columns / ReLU'd pass-through columns /
// sign bytes, row-major or the quad-major hybrid layout; backward: ReLU-sign multiply from the bytes DMA'd into LDS).
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kNT = 19;                 // column tiles of 16
constexpr int kWaves = 8;
constexpr int kMT = 2;                  // row tiles per wave and round
constexpr int kPieces = 3 * kNT;        // LDS-DMA pieces (1 KiB: 16 Bt rows x 64 B) per chunk: three images x 19 tiles
constexpr int kStage = kPieces * 256;   // floats per ring stage (58,368 B)
constexpr int kStages = 2;
constexpr int kMSlot = 1024;            // floats (4 KiB) of ReLU-sign bytes per wave
#ifdef A3VT_DBG_RG3_DMA4   // variant (measured 5-8 % slower): waves 0-3, one per SIMD, carry all of the Bt staging
constexpr int kDmaWaves = 4;
#else
constexpr int kDmaWaves = 8;            // the Bt pieces are dealt to all eight waves (8 / 7 per wave and chunk)
#endif
constexpr int kBPer = (kPieces + kDmaWaves - 1) / kDmaWaves;   // Bt pieces per DMA wave and chunk
constexpr size_t kLdsBytes = (size_t)(kStages * kStage + kWaves * kMSlot) * sizeof(float);   // 149,504 B
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void glds16(const float *gsrc, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ f32x4 mfma(f32x4 a, f32x4 b, f32x4 c) {
#ifdef A3VT_DBG_X3_NOMFMA   // timing-only ablations of this file (tools/build_variants.sh x3): results are wrong by design
  asm volatile("" ::"v"(a), "v"(b));
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bfloat(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {   // v_cvt_pk_bf16_f32: round to nearest even, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}

// The exact three-way split of two floats (a -> low halves, b -> high halves of the packed words):
//   hi  = the upper 16 bits of x (truncation: never overflows, x - hi is exact and has the sign of x),
//   mid = RNE_bf16(x - hi), lo = (x - hi) - mid  — exact again, and lo has at most 8 significant bits, so it IS a bf16.
// hi + mid + lo == x for every finite x whose lowest set bit is >= 2^-133 (the bf16 subnormal grid; all normal floats
// down to 2^-110 qualify); below that the difference is < 2^-133.  11 VALU instructions per pair.
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned ua = fbits(a), ub = fbits(b);
#ifdef A3VT_DBG_X3_NOSPLIT
  h = ua; m = ub; l = ua ^ ub;
  return;
#endif
  h = __builtin_amdgcn_perm(ub, ua, 0x07060302u);            // (ua >> 16) | (ub & 0xffff0000)
  const float ra = a - bfloat(ua & 0xffff0000u), rb = b - bfloat(ub & 0xffff0000u);
  m = cvt_pk(ra, rb);
  const float sa = ra - bfloat(m << 16), sb = rb - bfloat(m & 0xffff0000u);
  l = cvt_pk(sa, sb);
}
struct Pieces {
  f32x4 h, m, l;   // 8 bf16 each, element j = k-offset j of the lane's 8 operand elements
};
__device__ __forceinline__ Pieces split3_x8(f32x4 x0, f32x4 x1) {
  unsigned h[4], m[4], l[4];
  split3_pair(x0[0], x0[1], h[0], m[0], l[0]);
  split3_pair(x0[2], x0[3], h[1], m[1], l[1]);
  split3_pair(x1[0], x1[1], h[2], m[2], l[2]);
  split3_pair(x1[2], x1[3], h[3], m[3], l[3]);
  return Pieces{__builtin_bit_cast(f32x4, (u32x4){h[0], h[1], h[2], h[3]}), __builtin_bit_cast(f32x4, (u32x4){m[0], m[1], m[2], m[3]}),
                __builtin_bit_cast(f32x4, (u32x4){l[0], l[1], l[2], l[3]})};
}


// One 16 x 16 output tile with its operands pulled straight from global memory — the few rows the load-balanced split
// leaves over (launch_rowgemm3; rowtile_unit of gcn_gemm.hip in this mode).  Same split, same six products in the same
// order per 32-wide k chunk as rowgemm3_kernel's main loop, so a row gives the same bits wherever it lands.
template <int EPI>
__device__ __forceinline__ void rowtile3_unit(const RowGemmArgs &p, int row_base, int row_end, int mt, int n0, int lane) {
  constexpr int KB = 5;   // k chunks per register block
  const int l16 = lane & 15, q = lane >> 4;
  const int ar = min(row_base + mt * 16 + l16, row_end - 1);   // A row of this lane (ragged tail: duplicate, never stored)
  unsigned a0off = (unsigned)ar * (unsigned)p.lda0;
  const unsigned a0mul = p.a0q_nvert > 0 ? (unsigned)p.a0q_nvert : 1u;
  if (p.a0q_nvert > 0) {
    const int bq = ar / p.a0q_nvert;
    a0off = ((unsigned)bq * (unsigned)(p.a0q_quads * p.a0q_nvert) + (unsigned)(ar - bq * p.a0q_nvert)) * 4u;
  }
  const unsigned a1off = (unsigned)ar * (unsigned)p.lda1;
  const float *brow = p.bt + (size_t)(n0 + l16) * kX3ImageLd + q * 4;   // hi image; mid / lo one / two images further
  const int nch = (p.k + 31) >> 5;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nch; c0 += KB) {
    f32x4 ra[KB][2], rb[KB][3];
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const int ch = c0 + c < nch ? c0 + c : nch - 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int kk = ch * 32 + q * 8 + h * 4;
        kk = kk < p.k ? kk : p.k - 4;
        const float *src = kk < p.ksplit ? p.a0 + (size_t)(a0off + (unsigned)kk * a0mul) : p.a1 + (size_t)(a1off + (unsigned)kk);
        ra[c][h] = *reinterpret_cast<const f32x4 *>(src);
      }
#pragma unroll
      for (int im = 0; im < 3; ++im) rb[c][im] = *reinterpret_cast<const f32x4 *>(brow + (size_t)im * kX3ImageFloats + ch * 16);
    }
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      if (c0 + c >= nch) break;
      const Pieces ap = split3_x8(ra[c][0], ra[c][1]);
      acc = mfma(ap.h, rb[c][2], acc);
      acc = mfma(ap.m, rb[c][1], acc);
      acc = mfma(ap.h, rb[c][1], acc);
      acc = mfma(ap.l, rb[c][0], acc);
      acc = mfma(ap.m, rb[c][0], acc);
      acc = mfma(ap.h, rb[c][0], acc);
    }
  }
  // C/D layout: this lane holds column n0 + l16 of rows 4 q .. 4 q + 3 of the tile
  const int col = n0 + l16;
  const bool col_ok = col < p.n_store;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row_base + mt * 16 + q * 4 + r;
    const bool row_ok = row < row_end;
    const float v = acc[r];
    if (EPI == EPI_FWD_HIDDEN) {
      // ReLU-sign byte of 4 consecutive pass-through columns: gathered from the 4 lanes that hold them
      const unsigned mybit = (col_ok && col >= p.csplit && v > 0.f) ? 1u << (l16 & 3) : 0u;
      unsigned bits = mybit;
      bits |= __shfl_xor(bits, 1, 64);
      bits |= __shfl_xor(bits, 2, 64);
      if (row_ok && p.maskb && (l16 & 3) == 0 && (col | 3) >= p.csplit && col < p.n_store)
        p.maskb[(size_t)row * p.mld + p.moff + (col >> 2)] = (uint8_t)bits;
      if (row_ok && col_ok) {
        if (p.zq_nvert > 0 && col < p.zq_quads * 4) {   // quad-major raw columns
          const int bq = row / p.zq_nvert;
          p.c2[(((size_t)bq * p.zq_quads + (col >> 2)) * p.zq_nvert + (row - bq * p.zq_nvert)) * 4 + (col & 3)] = v;
        } else if (p.zq_nvert > 0 && col < p.yq_quads * 4) {   // quad-major pass-through columns of the activations
          const int bq = row / p.zq_nvert;
          p.yq[(((size_t)bq * p.yq_quads + (col >> 2)) * p.zq_nvert + (row - bq * p.zq_nvert)) * 4 + (col & 3)] =
              (v > 0.f || p.no_relu) ? v : 0.f;
        } else {
          // (the quad that straddles the cut: raw to c2 AND activated to c, as the main epilogue does — the raw copy's
          // columns >= csplit are never gathered, the activation's columns < csplit are overwritten by the aggregation)
          if (p.zq_nvert == 0 && col < ((p.csplit + 3) & ~3) && col < p.ldc2) p.c2[(size_t)row * p.ldc2 + col] = v;
          if (col >= (p.csplit & ~3)) p.c[(size_t)row * p.ldc + col] = (v > 0.f || p.no_relu) ? v : 0.f;
        }
      }
    } else {   // EPI_DX_MASK
      if (row_ok && col_ok) {
        if (p.zq_nvert > 0 && col < p.zq_quads * 4) {   // quad-major gradient columns: unmasked
          const int bq = row / p.zq_nvert;
          p.c2[(((size_t)bq * p.zq_quads + (col >> 2)) * p.zq_nvert + (row - bq * p.zq_nvert)) * 4 + (col & 3)] = v;
        } else {
          const unsigned byte = p.maskb[(size_t)row * p.mld + (col < p.csplit ? 0 : p.moff) + (col >> 2)];
          p.c[(size_t)row * p.ldc + col] = ((byte >> (col & 3)) & 1u) ? v : 0.f;
        }
      }
    }
  }
}

#ifdef A3VT_DBG_RG3_STAMPS   // diagnostic build (tools/build_variants.sh stamps3): s_memrealtime (100 MHz) + s_memtime at the phase boundaries
__device__ unsigned long long g_rg3_stamps[2 * 2 * 256 * 4 * 8];   // [real time | shader cycles][epilogue][workgroup][round (< 4)][8]
#define RG3_STAMP(round, k)                                                                                                  \
  do {                                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 256 && (round) < 4) {                                                               \
      const int i_ = (((EPI == EPI_DX_MASK ? 1 : 0) * 256 + blockIdx.x) * 4 + (round)) * 8 + (k);                            \
      g_rg3_stamps[i_] = __builtin_amdgcn_s_memrealtime();                                                                   \
      g_rg3_stamps[2 * 256 * 4 * 8 + i_] = __builtin_amdgcn_s_memtime();                                                     \
    }                                                                                                                        \
  } while (0)
#else
#define RG3_STAMP(round, k) do { } while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(64 * kWaves, 2) void rowgemm3_kernel(RowGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, q = lane >> 4;
  const int kpiece = ((lane & 3) ^ ((lane >> 3) & 2)) * 4;   // DMA side of the bank swizzle (rowgemm_kernel), 4-byte units
  const int qs = q ^ ((l16 >> 1) & 2);                       // reader side
  const int nchunks = (p.k + 31) >> 5;

  // 16-row tiles, dealt evenly to the workgroups
  const int tiles = (p.m + 15) >> 4;
  const int tbase = tiles / gridDim.x, trem = tiles % gridDim.x;
  const int t0 = blockIdx.x * tbase + ((int)blockIdx.x < trem ? blockIdx.x : trem);
  const int t1 = t0 + tbase + ((int)blockIdx.x < trem ? 1 : 0);

  // Bt pieces of this wave: piece pc = wave + 8 j = (image pc / 19, column tile pc % 19) — a wave-uniform offset — plus the
  // per-lane part of the source address: row (lane >> 2) of the tile, 16-byte k-quad kpiece of the chunk's 64 bytes.
  const int blane = (lane >> 2) * kX3ImageLd + kpiece;
  const int nbp = wave < kDmaWaves ? (kPieces - wave + kDmaWaves - 1) / kDmaWaves : 0;   // wave-uniform

  // This lane's A rows of a round (ragged tail: duplicate the last row, never stored), as float offsets from a0 / a1, and the
  // fp32 operand elements of the chunk in flight.  Past the last k (K = 300 of a 320-wide chunk row) the lane re-reads the
  // row's last quad: those elements meet zero rows of the weight images.  An absent second tile (partial round) reads the
  // clamped row and is never multiplied or stored.
  unsigned a0off[kMT], a1off[kMT];
  const unsigned a0mul = p.a0q_nvert > 0 ? (unsigned)p.a0q_nvert : 1u;   // quad-major a0: k -> k * N floats past the row's base
  f32x4 raw[kMT][2];
  auto round_first_tile = [&](int tb) {   // first tile of this wave in the round that starts at tile tb
    const int cnt = t1 - tb < kMT * kWaves ? t1 - tb : kMT * kWaves;
    const int base = cnt / kWaves, extra = cnt % kWaves;
    return tb + wave * base + (wave < extra ? wave : extra);
  };
  auto set_rows = [&](int tb) {
    const int row0 = round_first_tile(tb) * 16;
#pragma unroll
    for (int i = 0; i < kMT; ++i) {
      int r = row0 + i * 16 + l16;
      r = r < p.m ? r : p.m - 1;
      a0off[i] = (unsigned)r * (unsigned)p.lda0;
      if (p.a0q_nvert > 0) {
        const int bq = r / p.a0q_nvert;
        a0off[i] = ((unsigned)bq * (unsigned)(p.a0q_quads * p.a0q_nvert) + (unsigned)(r - bq * p.a0q_nvert)) * 4u;
      }
      a1off[i] = (unsigned)r * (unsigned)p.lda1;
    }
  };
  auto issue_a = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < kMT; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int kk = chunk * 32 + q * 8 + h * 4;
        kk = kk < p.k ? kk : p.k - 4;
        const float *src = kk < p.ksplit ? p.a0 + (size_t)(a0off[i] + (unsigned)kk * a0mul) : p.a1 + (size_t)(a1off[i] + (unsigned)kk);
#ifdef A3VT_DBG_RG3_NOA
        raw[i][h] = f32x4{1.f + kk, 2.f + i, 3.f + lane, (float)(size_t)src};
#else
        raw[i][h] = *reinterpret_cast<const f32x4 *>(src);
#endif
      }
  };

  int rnd_ = 0;
  bool a_ahead = false;   // the first A elements of this round were requested before the previous round's epilogue
  for (int tb = t0; tb < t1; tb += kMT * kWaves, ++rnd_) {
    RG3_STAMP(rnd_, 0);
    // tiles of this round for this wave: two each when the round is full, an even split otherwise
    const int cnt = t1 - tb < kMT * kWaves ? t1 - tb : kMT * kWaves;
    const int base = cnt / kWaves, extra = cnt % kWaves;
    const int nm = base + (wave < extra ? 1 : 0);   // 0..2, wave-uniform
    const int first = tb + wave * base + (wave < extra ? wave : extra);
    const bool active = nm > 0;
    const int row0 = first * 16;

    f32x4 acc[kMT][kNT];
#pragma unroll
    for (int i = 0; i < kMT; ++i)
#pragma unroll
      for (int j = 0; j < kNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue_b = [&](int chunk, int buf, int j) {
      if (j >= nbp) return;   // wave-uniform
#ifdef A3VT_DBG_RG3_NOB
      return;
#endif
      const int pc = wave + kDmaWaves * j, img = pc / kNT, tile = pc - img * kNT;   // scalar
      glds16(p.bt + ((size_t)img * kX3ImageFloats + (size_t)tile * (16 * kX3ImageLd) + chunk * 16) + blane, lds + buf * kStage + pc * 256);
    };

    if (EPI == EPI_DX_MASK) {
      // this round's ReLU-sign bytes (16 nm rows x mld contiguous bytes) ride along with the first chunk
      float *ms = lds + kStages * kStage + wave * kMSlot;
      const uint8_t *src0 = p.maskb + (size_t)row0 * p.mld;
      const int nbytes = 16 * nm * p.mld;
      for (int o = 0; o < 32 * p.mld; o += 1024) {
        const int b = o + lane * 16;
        const void *src = b < nbytes ? (const void *)(src0 + b) : (const void *)p.zeros;
        glds16(reinterpret_cast<const float *>(src), ms + o / 4);
      }
    }
    if (!a_ahead) {
      set_rows(tb);
      issue_a(0);
    }
#pragma unroll
    for (int j = 0; j < kBPer; ++j) issue_b(0, 0, j);

    // One chunk for a wave that owns NM (1 or 2, compile-time) tiles this round.
    auto chunk_step = [&](auto nmc, int t) __attribute__((always_inline)) {
      constexpr int NM = decltype(nmc)::value;
      const int buf = t & 1;
      const bool prefetch = t + 1 < nchunks;
      Pieces ap[NM];
#pragma unroll
      for (int i = 0; i < NM; ++i) ap[i] = split3_x8(raw[i][0], raw[i][1]);
      __builtin_amdgcn_sched_barrier(0);
      if (prefetch) issue_a(t + 1);     // HBM-streamed: as early as the registers are free
      __builtin_amdgcn_sched_barrier(0);
      const float *sB = lds + buf * kStage + l16 * 16 + qs * 4;
      // Fragments of column tile j: lo first, then mid, then hi; each register set is refilled for tile j + 1 as soon as
      // its last MFMA of tile j has issued, so the LDS latency hides under the rest of the tile (no second register set).
      f32x4 bl = *reinterpret_cast<const f32x4 *>(sB + 2 * kNT * 256);
      f32x4 bm = *reinterpret_cast<const f32x4 *>(sB + kNT * 256);
      f32x4 bh = *reinterpret_cast<const f32x4 *>(sB);
#pragma unroll
      for (int j = 0; j < kNT; ++j) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i][j] = mfma(ap[i].h, bl, acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < kNT) bl = *reinterpret_cast<const f32x4 *>(sB + (2 * kNT + j + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          acc[i][j] = mfma(ap[i].m, bm, acc[i][j]);
          acc[i][j] = mfma(ap[i].h, bm, acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < kNT) bm = *reinterpret_cast<const f32x4 *>(sB + (kNT + j + 1) * 256);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          acc[i][j] = mfma(ap[i].l, bh, acc[i][j]);
          acc[i][j] = mfma(ap[i].m, bh, acc[i][j]);
          acc[i][j] = mfma(ap[i].h, bh, acc[i][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < kNT) bh = *reinterpret_cast<const f32x4 *>(sB + (j + 1) * 256);
        if (kDmaWaves == 8 ? ((j & 1) && (j >> 1) < kBPer) : j < kBPer) {   // the next chunk's Bt pieces, spread over the column tiles
          if (prefetch) issue_b(t + 1, buf ^ 1, kDmaWaves == 8 ? j >> 1 : j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // (one loop per tile count: with both bodies inside one loop the register allocator spills the accumulators)
    if (nm == 2) {
      for (int t = 0; t < nchunks; ++t) {
        wait_vm0();                       // chunk t (this wave's share) has landed
        __builtin_amdgcn_s_barrier();     // ... everyone's has; everyone is done with chunk t - 1's stage
        if (t == 0) RG3_STAMP(rnd_, 1);
        if (t == 1) RG3_STAMP(rnd_, 5);
        chunk_step(std::integral_constant<int, 2>{}, t);
      }
    } else if (nm == 1) {
      for (int t = 0; t < nchunks; ++t) {
        wait_vm0();
        __builtin_amdgcn_s_barrier();
        chunk_step(std::integral_constant<int, 1>{}, t);
      }
    } else {                              // idle wave of a partial round: it still carries its share of the Bt staging
      for (int t = 0; t < nchunks; ++t) {
        wait_vm0();
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nchunks) {
#pragma unroll
          for (int j = 0; j < kBPer; ++j) issue_b(t + 1, (t & 1) ^ 1, j);
        }
      }
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();   // all waves finished reading the ring -> reuse it for the epilogue
    RG3_STAMP(rnd_, 2);
#ifdef A3VT_DBG_RG3_AHEAD
    // Measured, not shipped: the next round's first A elements (HBM-streamed) requested here, into the
    // free operand registers, land under the epilogue — the first-chunk wait of the next round falls from 6 to 1.2 us and
    // its K loop grows by the same 5 us (the loads share the CU's memory path with the epilogue's stores).
    a_ahead = tb + kMT * kWaves < t1;
    if (a_ahead) {
      set_rows(tb + kMT * kWaves);
      issue_a(0);
    }
#endif

    // ---- epilogue (rowgemm_kernel's, fp32 rows; launch_rowgemm3 guarantees ldc % 4 == 0, n_store % 4 == 0 and, for the
    // forward without the quad-major side output, ldc2 % 4 == 0: every column quad leaves with one 16-byte store)
#ifdef A3VT_DBG_RG3_NOEPI
    if (active && acc[0][0][0] == 1.2345e-33f) {   // never true in practice: keeps the accumulators alive, skips the epilogue
#else
    if (active) {
#endif
      // The epilogue's per-lane address arithmetic depends on the lane alone, so the compiler hoists it in front of the round
      // loop and then spills it across the K loop (where all 256 registers are taken): the reloads sat between the
      // epilogue's global stores, and a scratch reload waits — on the one vmcnt counter — for every store issued before it.
      // An opaque copy of the lane index keeps that arithmetic here, where registers are free again.
      int le = lane;
#ifndef A3VT_DBG_RG_HOISTED_EPI   // variant build (tools/build_variants.sh epi): without the fix, for A/B timing
      asm volatile("" : "+v"(le));
#endif
      const int l16 = le & 15, q = le >> 4;
      float *ep = lds + wave * ((kStages * kStage) / kWaves);
      constexpr int G0 = (kNT + 1) / 2;
      static_assert(16 * (G0 * 16 + 4) <= (kStages * kStage) / kWaves, "epilogue slice too small");
      uint8_t *mslot = reinterpret_cast<uint8_t *>(lds + kStages * kStage + wave * kMSlot);
      const bool mask_rows = EPI == EPI_FWD_HIDDEN && p.maskb != nullptr && 32 * p.mld <= 4 * kMSlot;
      const bool zq_mode = p.zq_nvert > 0;
#pragma unroll
      for (int i = 0; i < kMT; ++i) {
        if (i >= nm) continue;
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
          const int j0 = grp == 0 ? 0 : G0;
          const int tiles_g = grp == 0 ? G0 : kNT - G0;
          const int ncols = tiles_g * 16, stride = ncols + 4, f4row = ncols / 4;
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int jj = 0; jj < G0; ++jj) {
            if (jj < tiles_g) {
#pragma unroll
              for (int r = 0; r < 4; ++r) ep[(q * 4 + r) * stride + jj * 16 + l16] = acc[i][j0 + jj][r];
            }
          }
          __builtin_amdgcn_wave_barrier();
          int qlo = 0;
          if (zq_mode && grp == 0) {
            // quad-major side output: the first zq_quads column quads leave as 16 consecutive rows x 16 B per quad
            const int nqz = p.zq_quads;
            const int rl = le & 15;
            const int row = row0 + i * 16 + rl;
            const int bq = row / p.zq_nvert;
            float *qbase = p.c2 + ((size_t)bq * nqz * p.zq_nvert + (size_t)(row - bq * p.zq_nvert)) * 4;
            for (int c4 = le >> 4; c4 < nqz; c4 += 4) {
              const f32x4 v = *reinterpret_cast<const f32x4 *>(ep + rl * stride + c4 * 4);
              if (row < p.m) *reinterpret_cast<f32x4 *>(qbase + (size_t)c4 * p.zq_nvert * 4) = v;
            }
            qlo = nqz;
            if (EPI == EPI_FWD_HIDDEN && p.yq_quads > nqz) {
              // the pass-through columns [4 Q, 4 yq_quads) of the hybrid activations: ReLU, sign bits, row-fastest stores
              float *ybase = p.yq + ((size_t)bq * p.yq_quads * p.zq_nvert + (size_t)(row - bq * p.zq_nvert)) * 4;
              for (int c4 = nqz + (le >> 4); c4 < p.yq_quads; c4 += 4) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(ep + rl * stride + c4 * 4);
                unsigned bits = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                  bits |= (v[t] > 0.f ? 1u : 0u) << t;
                  v[t] = (v[t] > 0.f || p.no_relu) ? v[t] : 0.f;
                }
                if (row >= p.m) continue;
                if (p.maskb) {
                  if (mask_rows) mslot[(i * 16 + rl) * p.mld + p.moff + c4] = (uint8_t)bits;
                  else p.maskb[(size_t)row * p.mld + p.moff + c4] = (uint8_t)bits;
                }
                *reinterpret_cast<f32x4 *>(ybase + (size_t)c4 * p.zq_nvert * 4) = v;
              }
              qlo = p.yq_quads;
            }
          }
          const int wq = f4row - qlo, nf4 = 16 * wq;
          for (int f = le; f < nf4; f += 64) {
            const int rl = f / wq, c4 = qlo + (f - rl * wq);
            const int row = row0 + i * 16 + rl;
            const int col = j0 * 16 + c4 * 4;
            if (row >= p.m || col >= p.n_store) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(ep + rl * stride + c4 * 4);
            if (EPI == EPI_FWD_HIDDEN) {
              if (p.maskb && col + 3 >= p.csplit) {   // ReLU sign of the pass-through channels, 1 byte per 4 columns
                unsigned bits = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) bits |= ((col + t >= p.csplit && v[t] > 0.f) ? 1u : 0u) << t;
                if (mask_rows) mslot[(i * 16 + rl) * p.mld + p.moff + (col >> 2)] = (uint8_t)bits;
                else p.maskb[(size_t)row * p.mld + p.moff + (col >> 2)] = (uint8_t)bits;
              }
              if (!zq_mode && col + 3 < p.csplit) {   // aggregated channels: raw Z for the neighbour gather
                *reinterpret_cast<f32x4 *>(p.c2 + (size_t)row * p.ldc2 + col) = v;
              } else {
                // (the quad that straddles the cut goes to BOTH outputs whole, as in rowgemm_kernel)
                if (!zq_mode && col < p.csplit) *reinterpret_cast<f32x4 *>(p.c2 + (size_t)row * p.ldc2 + col) = v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (v[t] > 0.f || p.no_relu) ? v[t] : 0.f;
                *reinterpret_cast<f32x4 *>(p.c + (size_t)row * p.ldc + col) = v;
              }
            } else {   // EPI_DX_MASK: gradient through the ReLU of the producing layer (sign bytes from LDS)
              const int ur = i * 16 + rl;
              const unsigned ba = mslot[ur * p.mld + (col >> 2)];
              const unsigned bb = mslot[ur * p.mld + p.moff + (col >> 2)];
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const unsigned bit = ((col + t < p.csplit ? ba : bb) >> t) & 1u;
                v[t] = bit ? v[t] : 0.f;
              }
              *reinterpret_cast<f32x4 *>(p.c + (size_t)row * p.ldc + col) = v;
            }
          }
        }
      }
      if (mask_rows) {   // sign bytes of this wave's 16 nm rows: one contiguous block, 16 bytes per lane
        wait_lgkm0();
        __builtin_amdgcn_wave_barrier();
        const int nbytes = 16 * nm * p.mld;
        uint8_t *dstm = p.maskb + (size_t)row0 * p.mld;
        for (int o = le * 16; o < nbytes; o += 1024)
          *reinterpret_cast<f32x4 *>(dstm + o) = *reinterpret_cast<const f32x4 *>(mslot + o);
      }
    }
    RG3_STAMP(rnd_, 3);
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();   // epilogue slices are free again before the next round's DMA
    RG3_STAMP(rnd_, 4);
  }
  // Leftover rows of the load-balanced split (launch_rowgemm3): one 16 x 16 output tile per wave, dealt across the
  // workgroups, operands straight from global memory — a few microseconds at the end of this launch instead of a lone
  // extra tile that would make eight workgroups run a longer last round.
  if (p.rem_rows > 0) {
    const int units = ((p.rem_rows + 15) >> 4) * kNT;
    for (int u = wave * gridDim.x + blockIdx.x; u < units; u += kWaves * gridDim.x)
      rowtile3_unit<EPI>(p, p.rem_row0, p.rem_row0 + p.rem_rows, u / kNT, (u % kNT) * 16, lane);
  }
}

// The three bf16 images of up to kMaxImages weight matrices in one launch (blockIdx.z = layer): image p of layer l is
// [kX3ImageRows][2 kX3ImageLd] bf16 at dst + l * dst_stride + p * kX3ImageFloats (floats), zero padded;
// transpose = 1: row n, column k = W[k][n] (forward operand Bt = W^T), transpose = 0: row r, column c = W[r][c] (dX).
__global__ void weight_images3_kernel(WeightImages w) {
  __shared__ float tile[32][33];
  const int l = blockIdx.z;
  const float *src = w.w[l];
  unsigned short *dst = reinterpret_cast<unsigned short *>(w.dst + (size_t)l * w.dst_stride);
  const int k = w.k[l], n = w.n;
  constexpr int ld = 2 * kX3ImageLd;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx over image columns, by over image rows (32 x 32 tiles of 320 x 320)
  for (int i = threadIdx.y; i < 32; i += 8) {
    float v = 0.f;
    if (w.transpose) {   // image (row r, column c) = W[c][r]: read W rows (c fixed) along r
      const int c = bx + i, r = by + threadIdx.x;
      if (c < k && r < n) v = src[(size_t)c * n + r];
      tile[i][threadIdx.x] = v;
    } else {             // image (row r, column c) = W[r][c]
      const int r = by + i, c = bx + threadIdx.x;
      if (r < k && c < n) v = src[(size_t)r * n + c];
      tile[threadIdx.x][i] = v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = by + i, c = bx + threadIdx.x;   // consecutive threads -> consecutive columns of one image row
    unsigned h, m, lo;
    split3_pair(tile[threadIdx.x][i], 0.f, h, m, lo);
    const size_t idx = (size_t)r * ld + c;
    dst[idx] = (unsigned short)(h & 0xffffu);
    dst[idx + 2 * kX3ImageFloats] = (unsigned short)(m & 0xffffu);
    dst[idx + 4 * kX3ImageFloats] = (unsigned short)(lo & 0xffffu);
  }
}

__global__ void split3_kernel(const float *__restrict__ x, size_t n, unsigned short *__restrict__ hi,
                              unsigned short *__restrict__ mid, unsigned short *__restrict__ lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h, m, l;
  split3_pair(x[i], 0.f, h, m, l);
  hi[i] = (unsigned short)(h & 0xffffu);
  mid[i] = (unsigned short)(m & 0xffffu);
  lo[i] = (unsigned short)(l & 0xffffu);
}


// ------------------------------------------------------------------------------------------------
// dW = X^T dZ in mode 3 (autograd's weight gradient of model.py:352).  Same decomposition as dw_kernel / dw16_kernel:
// grid = (128 row slabs, 2 column groups), a workgroup's partial [k_in][n_out] goes to its slab, slab_reduce sums the
// slabs in a fixed order.  8 waves: wave (wi = w & 3, wo = w >> 2) owns up to 5 input tiles x 5 output tiles of its
// column group (100 accumulator registers).
// A stage is 32 rows.  Both operands are fp32 in HBM and both are needed as three bf16 pieces, column-wise (the MFMA sums
// over ROWS), by several waves each — so the split is done ONCE per element, through LDS:
//   1. LDS-DMA of the stage's fp32 quads (X: 76 per row; this column group's window of dZ: 40 per row) into a raw buffer
//      (59 KB), issued at the top of the previous stage's MFMA phase; quad-major planes (the hybrid layout) are walked
//      row-fastest, row-major rows quad-fastest, so that every wave-instruction covers whole cache lines;
//   2. split phase: every thread reads back the quads it requested, splits them (split3_pair) and writes the three piece
//      images, row-major [32][304] (X) / [32][176] (window) bf16 — row strides of 8 (odd) dwords, so that the transposed
//      reads below are bank-conflict free;
//   3. MFMA phase: operands come out of the images with ds_read_b64_tr_b16 exactly as in dw16_kernel (k-group g takes rows
//      4 g .. 4 g + 3 and 16 + 4 g ..), six passes of 25 MFMAs per stage, the next pass's fragments in flight.
// Two barriers per stage.  The raw slots are wave-private (a wave splits the quads it requested), so the next stage's DMA is
// issued as soon as a wave holds its quads in registers and stays in flight through the split, the barriers and the MFMA phase.
// ------------------------------------------------------------------------------------------------
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using u16 = unsigned short;

constexpr int kDwThreads = 512;
constexpr int kDwXq = 76, kDwZq = 40;                    // fp32 quads per stage row: X (304 columns), dZ window (160)
constexpr int kDwUnits = 32 * (kDwXq + kDwZq);           // 3712 quads per stage
constexpr int kDwPer = (kDwUnits + kDwThreads - 1) / kDwThreads;   // 8 per thread (the last wave-instruction is partial)
constexpr int kDwXld = 304, kDwZld = 176;                // piece image row strides (bf16 elements)
constexpr int kDwXimg = 32 * kDwXld, kDwZimg = 32 * kDwZld;
constexpr int kDwPieceElems = 3 * (kDwXimg + kDwZimg);   // 46,080 bf16 = 92,160 B
constexpr int kDwRawFloats = kDwPer * kDwThreads * 4;    // 16,384 floats = 65,536 B
constexpr size_t kDwLdsBytes = (size_t)kDwPieceElems * 2 + (size_t)kDwRawFloats * 4 + 1024;   // + tail slack for clamped tiles

__device__ __forceinline__ f32x4 tr_operand(const u16 *lds_row0_col, int row_stride_elems) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds_row0_col));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(lds_row0_col + 16 * row_stride_elems));
  return __builtin_bit_cast(f32x4, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

__global__ __launch_bounds__(kDwThreads, 2) void dw3_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  u16 *pieces = reinterpret_cast<u16 *>(lds);
  float *raw = lds + kDwPieceElems / 2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int bq = l16 >> 2, bp = l16 & 3;           // transposed-read address role: block row bq, column quad bp
  const int wi = wave & 3, wo = wave >> 2;

  // tile ownership (as dw16_kernel)
  const int tin = (p.k_in + 15) >> 4, tout = (p.n_out + 15) >> 4;
  const int ni = tin / 4 + (wi < tin % 4 ? 1 : 0);
  const int i0 = wi * (tin / 4) + (wi < tin % 4 ? wi : tin % 4);
  const int gbase = tout / gridDim.y, grem = tout % gridDim.y;
  const int gt0 = blockIdx.y * gbase + ((int)blockIdx.y < grem ? blockIdx.y : grem);
  const int gtn = gbase + ((int)blockIdx.y < grem ? 1 : 0);
  const int no = gtn / 2 + (wo < gtn % 2 ? 1 : 0);
  const int o0 = gt0 + wo * (gtn / 2) + (wo < gtn % 2 ? wo : gtn % 2);
  const int gcol0 = gt0 * 16;                      // first dZ column of this group's window

  // rows of this workgroup in stages of 32
  const int units = (p.m + 31) >> 5;
  const int ubase = units / gridDim.x, urem = units % gridDim.x;
  const int u0 = blockIdx.x * ubase + ((int)blockIdx.x < urem ? blockIdx.x : urem);
  const int nu = ubase + ((int)blockIdx.x < urem ? 1 : 0);

  // ---- this thread's quads of a stage: e = tid + 512 j.  Blocks of the e range (each a multiple of 32 long):
  //   [X quad-major quads: row-fastest][X row-major quads: quad-fastest][window quad-major quads][window row-major quads]
  const int nq0 = p.xq_nvert > 0 ? p.xq_quads : 0, nq1 = kDwXq - nq0;
  const int zq_quads_win = p.z0q_nvert > 0 ? max(min(p.z0q_quads - gcol0 / 4, kDwZq), 0) : 0;   // window quads served by the quad-major dZa
  const int nq2 = zq_quads_win, nq3 = kDwZq - nq2;
  const int e1 = 32 * nq0, e2 = e1 + 32 * nq1, e3 = e2 + 32 * nq2;
  const int qn = p.xq_nvert > 0 ? p.xq_nvert : p.z0q_nvert;   // vertices per mesh (the same for X and dZa)
  const float *sp[kDwPer];     // source of the quad for the next stage
  int pdst[kDwPer];            // bf16 element offset of the quad in piece image 0 (X images, then the window images); -1: none
  // What a quad's pointer advances by per stage, as a 3-bit code per quad packed into one register (the advances themselves
  // are five wave-uniform values; kept per quad as 64-bit pointer increments they cost 32 registers and spilled into the
  // stage loop, where a scratch reload waits on the one vmcnt counter behind the LDS-DMA issued a moment earlier):
  // 0 zero quad, 1 quad-major X, 2 row-major X, 3 row-major dZa, 4 row-major G, 5 quad-major dZa
  unsigned kinds = 0;
  const size_t row_first = (size_t)u0 * 32;
  auto unit_row = [&](int e) {   // stage row of quad e (the block structure above)
    if (e < e1) return e & 31;
    if (e < e2) return (e - e1) / nq1;
    if (e < e3) return (e - e2) & 31;
    return (e - e3) / nq3;
  };
#pragma unroll
  for (int j = 0; j < kDwPer; ++j) {
    const int e = tid + kDwThreads * j;
    const int row = unit_row(e);
    int quad;
    bool isx = true, qm = false;
    if (e < e1) { quad = e >> 5; qm = true; }
    else if (e < e2) { quad = nq0 + (e - e1 - row * nq1); }
    else if (e < e3) { quad = (e - e2) >> 5; isx = false; qm = true; }
    else { quad = nq2 + (e - e3 - row * nq3); isx = false; }
    const bool on = e < kDwUnits;
    const size_t grow = row_first + row;
    const float *src = p.zeros;
    unsigned kd = 0;
    if (on && isx) {
      const int col = quad * 4;
      if (qm) {
        const int b = (int)(grow / qn), v = (int)(grow - (size_t)b * qn);
        src = p.xq + (((size_t)b * p.xq_quads + quad) * qn + v) * 4;
        kd = 1;
      } else if (col < p.k_in) {
        src = p.x + grow * p.ldx_src + col;
        kd = 2;
      }
    } else if (on) {
      const int col = gcol0 + quad * 4;   // dZ column
      if (qm) {
        const int b = (int)(grow / qn), v = (int)(grow - (size_t)b * qn);
        src = p.z0 + (((size_t)b * p.z0q_quads + (col >> 2)) * qn + v) * 4;
        kd = 5;
      } else if (col < p.zsplit) {
        src = p.z0 + grow * p.ldz0 + col;
        kd = 3;
      } else if (col < p.n_out) {
        src = p.z1 + grow * p.ldz1 + col;
        kd = 4;
      }
    }
    sp[j] = src;
    kinds |= kd << (3 * j);
    pdst[j] = !on ? -1 : (isx ? row * kDwXld + quad * 4 : 3 * kDwXimg + row * kDwZld + quad * 4);
  }
  // vertex index (inside its mesh) of this thread's row of the NEXT stage to be issued: quad-major quads all sit on stage
  // row tid & 31 (every block of the e range starts at a multiple of 32)
  int vrow = qn > 0 ? (int)((row_first + (tid & 31)) % (size_t)qn) : 0;
  const int wrap_x = p.xq_nvert > 0 ? (p.xq_quads - 1) * qn * 4 : 0, wrap_z = p.z0q_nvert > 0 ? (p.z0q_quads - 1) * qn * 4 : 0;
  const int adv_x = 32 * p.ldx_src, adv_z0 = 32 * p.ldz0, adv_z1 = 32 * p.ldz1;

  auto issue = [&](int unit) {   // DMA of stage `unit` (global index) into the raw buffer, then move every pointer 32 rows on
    const int rows_left = p.m - unit * 32;
    if (rows_left >= 32) {
#pragma unroll
      for (int j = 0; j < kDwPer; ++j) {
        if (j * kDwThreads + wave * 64 >= kDwUnits) continue;   // wave-uniform: nothing of this instruction exists
#ifndef A3VT_DBG_DW3_NODMA
        glds16(sp[j], raw + (j * kDwThreads + wave * 64) * 4);
#endif
      }
    } else {   // the ragged last stage of the whole problem: rows >= m contribute zeros
#pragma unroll
      for (int j = 0; j < kDwPer; ++j) {
        if (j * kDwThreads + wave * 64 >= kDwUnits) continue;
        const int e = tid + kDwThreads * j;
        glds16(e < kDwUnits && unit_row(e) < rows_left ? sp[j] : p.zeros, raw + (j * kDwThreads + wave * 64) * 4);
      }
    }
    const bool wrap = qn > 0 && vrow + 32 >= qn;   // per-lane: this row crosses into the next mesh
    if (qn > 0) vrow = wrap ? vrow + 32 - qn : vrow + 32;
#pragma unroll
    for (int j = 0; j < kDwPer; ++j) {
      const unsigned kd = (kinds >> (3 * j)) & 7u;
      int a = kd == 2 ? adv_x : kd == 3 ? adv_z0 : kd == 4 ? adv_z1 : (kd == 1 || kd == 5) ? 128 : 0;
      if (wrap) a += kd == 1 ? wrap_x : kd == 5 ? wrap_z : 0;
      sp[j] += a;
    }
  };

  f32x4 acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane element offsets of the transposed reads inside piece image 0: block row (4 g + bq), column quad 4 bp of the
  // tile; tiles a wave does not own are clamped in-bounds (read, multiplied, never stored)
  int xoff[5], zoff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) xoff[i] = (4 * g + bq) * kDwXld + min((i0 + i) * 16 + 4 * bp, kDwXld - 4);
#pragma unroll
  for (int j = 0; j < 5; ++j) zoff[j] = 3 * kDwXimg + (4 * g + bq) * kDwZld + min((o0 - gt0 + j) * 16 + 4 * bp, 160 - 4);

  // A wave reads back exactly the quads it requested (raw slot (j * 512 + tid)), so the raw buffer needs no workgroup
  // barrier: as soon as a wave has its stage in registers it requests the next one, which then has the split, the barriers
  // and the whole MFMA phase to land.
  if (nu > 0) issue(u0);
  for (int t = 0; t < nu; ++t) {
    wait_vm0();                     // this wave's quads of stage t have landed
    f32x4 rv[kDwPer];
#pragma unroll
    for (int j = 0; j < kDwPer; ++j) rv[j] = *reinterpret_cast<const f32x4 *>(raw + (j * kDwThreads + tid) * 4);
    wait_lgkm0();
    if (t + 1 < nu) issue(u0 + t + 1);
    // ---- split (registers), then the piece images once everyone is done with those of stage t - 1
    unsigned hh[kDwPer][2], mm[kDwPer][2], ll[kDwPer][2];
#pragma unroll
    for (int j = 0; j < kDwPer; ++j) {
      split3_pair(rv[j][0], rv[j][1], hh[j][0], mm[j][0], ll[j][0]);
      split3_pair(rv[j][2], rv[j][3], hh[j][1], mm[j][1], ll[j][1]);
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < kDwPer; ++j) {
#ifdef A3VT_DBG_DW3_NOSPLITPHASE
      continue;
#endif
      if (pdst[j] >= 0) {
        const bool isx = pdst[j] < 3 * kDwXimg;
        u16 *d = pieces + pdst[j];
        const int img = isx ? kDwXimg : kDwZimg;
        using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
        *reinterpret_cast<u32x2 *>(d) = u32x2{hh[j][0], hh[j][1]};
        *reinterpret_cast<u32x2 *>(d + img) = u32x2{mm[j][0], mm[j][1]};
        *reinterpret_cast<u32x2 *>(d + 2 * img) = u32x2{ll[j][0], ll[j][1]};
      }
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();   // piece images of stage t complete
    // ---- MFMA phase: passes (a piece, b piece) = (h,h) (h,m) (h,l) (m,h) (m,m) (l,h)
    const u16 *xp = pieces, *zp = pieces;   // (zoff already carries the window images' base)
    // Passes (a piece, b piece) in the order (h,l) (h,m) (h,h) (l,h) (m,h) (m,m): one fragment set changes per pass and is
    // loaded under the previous pass's MFMAs — three sets of five fragments live (60 registers), 35 fragments per stage.
    f32x4 fa[5], fb[5], fc[5];
    auto load_a = [&](f32x4 (&dst)[5], int piece) {
#pragma unroll
      for (int i = 0; i < 5; ++i) dst[i] = tr_operand(xp + piece * kDwXimg + xoff[i], kDwXld);
    };
    auto load_b = [&](f32x4 (&dst)[5], int piece) {
#pragma unroll
      for (int j = 0; j < 5; ++j) dst[j] = tr_operand(zp + piece * kDwZimg + zoff[j], kDwZld);
    };
    auto pass = [&](const f32x4 (&av)[5], const f32x4 (&bv)[5]) {
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = mfma(av[i], bv[j], acc[i][j]);
    };
    load_a(fa, 0);                  // a_h
    load_b(fb, 2);                  // b_l
    __builtin_amdgcn_sched_barrier(0);
    load_b(fc, 1);                  // b_m
    __builtin_amdgcn_sched_barrier(0);
    pass(fa, fb);                   // (h, l)
    __builtin_amdgcn_sched_barrier(0);
    load_b(fb, 0);                  // b_h
    __builtin_amdgcn_sched_barrier(0);
    pass(fa, fc);                   // (h, m)
    __builtin_amdgcn_sched_barrier(0);
    load_a(fc, 2);                  // a_l
    __builtin_amdgcn_sched_barrier(0);
    pass(fa, fb);                   // (h, h)
    __builtin_amdgcn_sched_barrier(0);
    load_a(fa, 1);                  // a_m
    __builtin_amdgcn_sched_barrier(0);
    pass(fc, fb);                   // (l, h)
    __builtin_amdgcn_sched_barrier(0);
    load_b(fc, 1);                  // b_m
    __builtin_amdgcn_sched_barrier(0);
    pass(fa, fb);                   // (m, h)
    __builtin_amdgcn_sched_barrier(0);
    pass(fa, fc);                   // (m, m)
  }

  // partial -> slab[blockIdx.x][k_in][n_out]; C/D layout: lane holds column (lane & 15) of rows 4 g .. 4 g + 3 of the tile
  float *slab = p.slab + (size_t)blockIdx.x * p.k_in * p.n_out;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (i >= ni) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kin = (i0 + i) * 16 + g * 4 + r;
      if (kin >= p.k_in) continue;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        if (j >= no) continue;
        const int col = (o0 + j) * 16 + l16;
        if (col < p.n_out) slab[(size_t)kin * p.n_out + col] = acc[i][j][r];
      }
    }
  }
}

}  // namespace

#ifdef A3VT_DBG_RG3_STAMPS
extern "C" int a3vt_dbg_rg3_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_rg3_stamps), sizeof(unsigned long long) * 2 * 2 * 256 * 4 * 8);
}
#endif

int launch_weight_images3(const WeightImages &w, hipStream_t s) {
  A3VT_LAUNCH(weight_images3_kernel, dim3(10, 10, w.count), dim3(32, 8), 0, s, w);
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_split3(const float *x, size_t n, unsigned short *hi, unsigned short *mid, unsigned short *lo, hipStream_t s) {
  if (n == 0) return 0;
  A3VT_LAUNCH(split3_kernel, dim3(cdiv((long long)n, 256)), dim3(256), 0, s, x, n, hi, mid, lo);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Shapes the split-operand kernel takes: the hidden layers (K and n in (288, 320] / (288, 304]) with enough rows to fill
// the chip; everything else of a mode-3 stack runs the exact fp32 kernels.
bool rowgemm3_dims_ok(long long m, int k, int n_store) {
  return k > 288 && k <= 320 && k % 4 == 0 && n_store > 288 && n_store <= 304 && n_store % 4 == 0 && m >= 96 * 128 &&
         m * 320 < (1ll << 32);   // (the kernel addresses A rows with 32-bit float offsets: 13.4 M rows)
}

// A whole mode-3 STACK takes the split-operand kernels for its hidden layers only if both of their products do: the dX
// epilogue parks 32 rows of ReLU-sign bytes (mld bytes each) in a wave's 4 KiB slot, so a cut that makes the sign rows
// longer than 128 bytes (hidden 300: cut_len > 212) sends the WHOLE stack to the exact mode-0 kernels — never a forward on
// one kernel family and a backward that cannot follow it.
bool rowgemm3_stack_ok(long long m, int hidden, int mld) {
  return rowgemm3_dims_ok(m, hidden, hidden) && 32 * mld <= 4 * kMSlot;
}

bool rowgemm3_ok(const RowGemmArgs &a, int epi) {
  if (a.bf16 != 3 || (epi != EPI_FWD_HIDDEN && epi != EPI_DX_MASK)) return false;
  if (!rowgemm3_dims_ok(a.m, a.k, a.n_store)) return false;
  if (a.lda0 % 4 != 0 || a.lda1 % 4 != 0 || a.ksplit % 4 != 0 || a.ldc % 4 != 0 || a.csplit > a.n_store) return false;
  // the kernel addresses A rows with 32-bit float offsets from a0 / a1
  const long long ldmax = a.lda0 > a.lda1 ? a.lda0 : a.lda1;
  if ((long long)a.m * (ldmax > 320 ? ldmax : 320) >= (1ll << 32)) return false;
  if (a.a0q_nvert > 0 && (a.m % a.a0q_nvert != 0 || a.ksplit != a.a0q_quads * 4)) return false;
  if (a.zq_nvert > 0 && (a.c2 == nullptr || a.m % a.zq_nvert != 0 || a.zq_quads * 4 != pad4(a.csplit) || a.zq_quads * 4 > 160)) return false;
  if (a.yq_quads > 0 && (a.zq_nvert <= 0 || a.yq == nullptr || epi != EPI_FWD_HIDDEN || a.yq_quads < a.zq_quads || a.yq_quads * 4 > 160)) return false;
  if (epi == EPI_FWD_HIDDEN && a.zq_nvert == 0 && a.csplit > 0 && (a.c2 == nullptr || a.ldc2 % 4 != 0 || pad4(a.csplit) > a.ldc2)) return false;
  if (epi == EPI_DX_MASK && (a.maskb == nullptr || 32 * a.mld > 4 * kMSlot)) return false;
  return true;
}

int launch_rowgemm3(const RowGemmArgs &a0, int epi, hipStream_t s) {
  RowGemmArgs a = a0;
  a.rem_row0 = a.rem_rows = 0;
  if (!rowgemm3_ok(a, epi)) {
    set_error("rowgemm3: unsupported call (m=%d k=%d n=%d epi=%d mode=%d)", a.m, a.k, a.n_store, epi, a.bf16);
    return -1;
  }
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)rowgemm3_kernel<EPI_FWD_HIDDEN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    (void)hipFuncSetAttribute((const void *)rowgemm3_kernel<EPI_DX_MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  });
  // Load balance as launch_rowgemm_epi (gcn_gemm.hip): the chip runs 2048 waves of this kernel, each owning tiles in pairs;
  // the tiles that fill whole rounds go to the round loop, a small remainder (8 tiles = 128 rows at bs 64) to the tail.
  int tiles = cdiv(a.m, 16);
  const int full = tiles / 2048 * 2048, rem = tiles - full;
  if (full > 0 && rem > 0 && rem * kNT <= 1024) {
    a.rem_row0 = full * 16;
    a.rem_rows = a.m - full * 16;
    a.m = full * 16;
    tiles = full;
  }
  const int grid = cdiv(tiles, kWaves) < 256 ? cdiv(tiles, kWaves) : 256;
  path_count(PATH_RG3);
  if (epi == EPI_FWD_HIDDEN)
    A3VT_LAUNCH((rowgemm3_kernel<EPI_FWD_HIDDEN>), dim3(grid), dim3(64 * kWaves), kLdsBytes, s, a);
  else
    A3VT_LAUNCH((rowgemm3_kernel<EPI_DX_MASK>), dim3(grid), dim3(64 * kWaves), kLdsBytes, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}


// dW shapes of mode 3: both dimensions those of a hidden layer, enough rows for 128 slabs of a few stages each.
bool dw3_ok(const DwArgs &a) {
  if (a.bf16 != 3) return false;
  if (a.k_in <= 288 || a.k_in > 304 || a.n_out <= 288 || a.n_out > 304 || a.m < 96 * 128) return false;
  if (a.ldz0 % 4 != 0 || a.ldz1 % 4 != 0 || a.zsplit % 4 != 0 || a.zsplit > 160 || a.n_out > a.ldz1) return false;
  const int ldx_src = a.ldx_src > 0 ? a.ldx_src : a.ldx;
  if (ldx_src % 4 != 0) return false;
  if (a.xq_nvert > 0 && (a.xq == nullptr || a.xq_quads * 4 > 160 || a.m % a.xq_nvert != 0 || a.xq_nvert < 32)) return false;
  if (a.z0q_nvert > 0 && (a.z0q_quads * 4 != a.zsplit || a.m % a.z0q_nvert != 0 || a.z0q_nvert < 32)) return false;
  if (a.xq_nvert > 0 && a.z0q_nvert > 0 && a.xq_nvert != a.z0q_nvert) return false;
  if (a.z0q_nvert == 0 && a.zsplit > 0 && (a.zsplit > a.ldz0 || a.ldz0 == 4)) return false;   // (a stage step of 128 floats marks a quad-major quad)
  return true;
}

int launch_dw3(const DwArgs &a0, hipStream_t s) {
  DwArgs a = a0;
  if (a.ldx_src == 0) a.ldx_src = a.ldx;
  if (!dw3_ok(a)) {
    set_error("dw3: unsupported call (m=%d k_in=%d n_out=%d zsplit=%d mode=%d)", a.m, a.k_in, a.n_out, a.zsplit, a.bf16);
    return -1;
  }
  static OncePerDevice once;
  once.run([] { (void)hipFuncSetAttribute((const void *)dw3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDwLdsBytes); });
  path_count(PATH_DW3);
  A3VT_LAUNCH(dw3_kernel, dim3(dw_num_slabs(a.n_out), 2), dim3(kDwThreads), kDwLdsBytes, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
