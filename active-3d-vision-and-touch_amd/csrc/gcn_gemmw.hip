// gcn_gemmw.hip — the exact fp32 products of the hidden layers with the WEIGHTS RESIDENT IN REGISTERS (gfx950, round 6).
//
// Replaces torch.matmul(features, self.weight) at reconstruction/vision/model.py:352 and autograd's dX = dZ W^T for the
// 300 x 300 layers of a stack on hybrid rows (the shapes rowgemm_kernel<19, ..., ADIRECT> served; every other shape stays
// there).  rowgemm_kernel is round structured: per 256-row round it re-stages all of Bt (19 pieces per chunk from L2 into
// LDS), drains the accumulators through LDS, and only then starts the next round — its own phase stamps price the work
// outside the K loop at 12 points of the launch and the re-staging at 4 (profiles/r03_rowgemm_phase_stamps.txt).  Here one
// persistent 4-wave workgroup per CU, ONE wave per SIMD with the whole 512-entry register file, keeps the weight image in
// registers for the whole launch and the rows stream past it.
//
// What shapes everything (tools/ubench/mfma_plus_valu.hip, mfma_gap_budget.hip): on gfx950 the fp32 MFMA and the vector ALU
// share the SIMD's fp32 pipe — two waves of a SIMD issuing v_mfma_f32_16x16x4_f32 and v_fma_f32 reach 96 TFLOP/s together,
// not 155 + 72 — and for a lone wave EVERY vector-ALU instruction between two MFMAs costs matrix time: 14.8 ns per MFMA
// with nothing between, 21.0 with two v_add_f32, 27.2 with six.  Scalar instructions (<= 4 per gap) and LDS reads cost
// nothing.  So the waves are specialised:
//   * waves 0-2 do NOTHING but multiply: 5 column tiles each (T = w + 3 t), 380 registers of B fragments, per chunk one
//     ds_read_b128 (the A fragment) and 20 MFMAs; at the end of a 16-row tile the 5 accumulator tiles go to an LDS staging
//     slot with 5 ds_write_b128.  No address arithmetic, no epilogue, no memory instruction.
//   * wave 3 owns 4 column tiles (15..18: 16 MFMAs per chunk where the others issue 20) and spends the difference on
//     everything else: the LDS-DMA of the rows two tiles ahead (one chunk per chunk of its K loop, the address a scalar
//     base + a per-tile lane offset: no vector instruction), and the epilogue of the PREVIOUS tile — column tile c behind
//     chunk c, read back from the staging slot: ReLU / sign bytes / quad-major or row-major stores (forward), sign-byte
//     mask (backward); column tiles that lie wholly inside one output region take a branch-free path chosen by scalar
//     compares, whose store address is again scalar base + per-tile lane offset.
//   * the MFMA runs TRANSPOSED (a = W^T fragment, b = X fragment): a lane holds four consecutive output columns of one row —
//     one float4 of a quad-major plane or 16 bytes of a row-major row — so nothing is transposed through LDS.
//   * one workgroup barrier per 16-row tile; the rows ring is three stages deep, wave 3's counted vmcnt wait in front of the
//     barrier leaves its own recent stores and loads in flight.
// Same products in the same order along K as rowgemm_kernel (chunk c, step s takes k = 16 c + 4 (lane >> 4) + s; an fp32
// MFMA is an exact fma chain over its four k): outputs bit-identical.  Leftover tiles of the even split: rowtile_unit.
#include <type_traits>

#include "gemm_tile.h"

namespace a3vt {

constexpr int kWChunks = 19;                 // K chunks of 16 (288 < K <= 304) = column tiles of 16 (288 < n <= 304)
constexpr int kWStage = kWChunks * 256;      // floats of one rows stage / one staging slot: [chunk or tile][lane] float4
constexpr int kWStages = 3;
constexpr int kWMaskSlot = 512;              // floats (2 KiB) of sign bytes of 16 rows (mld <= 128)
// rows ring, two staging slots, sign bytes leaving (2 slots + one nobody reads) and arriving (4 slots + one nobody reads)
constexpr int kWLdsFloats = (kWStages + 2) * kWStage + (3 + 5) * kWMaskSlot;

#ifdef A3VT_DBG_RGW_STAMPS   // diagnostic build (tools/build_variants.sh rgw): s_memtime per tile phase, [workgroup][wave][tile < 48][4]
__device__ unsigned long long g_rgw_stamps[256 * 4 * 48 * 4];
#define RGW_STAMP(tl, k)                                                                                       \
  do {                                                                                                         \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (tl) < 48)                                              \
      g_rgw_stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 48 + (tl)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// wall-clock (s_memrealtime, 100 MHz) of wave 0 at: kernel entry, first tile, behind the last tile, kernel end
__device__ unsigned long long g_rgw_rt[256 * 4];
#define RGW_RT(k)                                                                                              \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 256) g_rgw_rt[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define RGW_STAMP(tl, k) do { } while (0)
#define RGW_RT(k) do { } while (0)
#endif

template <int V>
using WIdx = std::integral_constant<int, V>;

// The shape is a COMPILE-TIME property of this kernel: the reference's hidden layers (300 x 300, cut 0.33: 25 aggregated
// quads; hybrid rows: 40 quad-major quads of the activations).  Every decision "which output region does column tile T lie
// in", "which part of A does chunk c come from" then folds, and wave 3's side work is straight-line code; with run-time
// bounds it was ~15 scalar branches per chunk, and a taken branch is ~30 idle cycles for a lone wave (20.6 k cycles per tile
// where the multiplying waves need 13.7 k).  Other shapes stay on rowgemm_kernel (rowgemmw_ok).
template <int EPI>
struct WShape {
  static constexpr int K = 300, N = 300, ZQ = 25;                    // K, n_store, aggregated quads (pad4(99) / 4)
  static constexpr int YQ = EPI == EPI_FWD_HIDDEN ? 40 : 25;         // quad-major quads of the output (forward: activations)
  static constexpr int KSPLIT = EPI == EPI_FWD_HIDDEN ? 160 : 100;   // quad-major columns of A (activations / dZa)
};

// The MFMA as written instructions, with the register FILE of every operand chosen here: accumulators in AGPRs, the B
// fragment (one element of the rows' ds_read_b128) in a VGPR, the weight element in a VGPR ("v") or an AGPR ("a").  From the
// builtin the compiler keeps the weights it cannot fit into 256 VGPRs in AGPRs and COPIES each one into a VGPR in front of
// its MFMA (104-124 v_accvgpr_read per tile: vector instructions, ~5 cycles of matrix time each) although the instruction
// takes AGPR sources.  What the compiler no longer knows is that these are MFMAs: the wait states between the last MFMA of a
// tile and the first read of its result are written out below (w_mfma_done).
// (first / a_in_agpr are constants after unrolling: one of the four statements survives)
__device__ __forceinline__ void w_mfma(bool first, bool a_in_agpr, f32x4 &acc, float a, float b) {
  if (first) {
    if (a_in_agpr) asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "a"(a), "v"(b));
    else asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
  } else {
    if (a_in_agpr) asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "a"(a), "v"(b));
    else asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  }
}
// a 16x16x4 fp32 MFMA is 8 passes: its result may be read by a non-MFMA instruction 18 wait states after issue
__device__ __forceinline__ void w_mfma_done() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }


// ---- waves 0-2: multiply -------------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void rowgemmw_compute(const RowGemmArgs &p, float *lds, int wave, int lane, int t0, int t1) {
  constexpr int NT = 5;
  const int l16 = lane & 15, q = lane >> 4;
  float *sA = lds, *sE = lds + kWStages * kWStage;
  // lane (l16, q) holds Bt[16 T + l16][16 c + 4 q + s] = the a-operand of step s of chunk c (scalars: as float4 tuples the
  // allocator reloaded whole groups in front of every MFMA that takes one element)
  constexpr int NV = 3;   // column tiles whose weights live in VGPRs (228 registers); the other two in AGPRs (152)
  float bv[kWChunks][4][NV], ba[kWChunks][4][NT - NV];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float *br = p.bt + (size_t)((wave + 3 * t) * 16 + l16) * p.ldb + q * 4;
#pragma unroll
    for (int c = 0; c < kWChunks; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(br + c * 16);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        if (t < NV) bv[c][sidx][t] = v[sidx];
        else ba[c][sidx][t - NV] = v[sidx];
      }
    }
  }
  wait_vmcnt<0>();
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();   // (prologue barrier: the first two tiles' rows are in the ring)
  RGW_RT(1);
  int st = 0;
  for (int tile = t0; tile < t1; ++tile) {
    RGW_STAMP(tile - t0, 0);
    const float *sa = sA + st * kWStage + lane * 4;
    float *se = sE + ((tile - t0) & 1) * kWStage + lane * 4;
    f32x4 acc[NT];
    f32x4 af = *reinterpret_cast<const f32x4 *>(sa);
#pragma unroll
    for (int c = 0; c < kWChunks; ++c) {
      f32x4 an = af;
      if (c + 1 < kWChunks) an = *reinterpret_cast<const f32x4 *>(sa + (c + 1) * 256);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t < NV) w_mfma(c == 0 && sidx == 0, false, acc[t], bv[c][sidx][t], af[sidx]);
          else w_mfma(c == 0 && sidx == 0, true, acc[t], ba[c][sidx][t - NV], af[sidx]);
        }
      }
      af = an;
    }
    w_mfma_done();
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(se + (wave + 3 * t) * 256) = acc[t];
    RGW_STAMP(tile - t0, 1);
    wait_lgkm0();
    RGW_STAMP(tile - t0, 2);
    __builtin_amdgcn_s_barrier();   // this tile's results staged; the next tile's rows are in the ring
    RGW_STAMP(tile - t0, 3);
    st = st == 2 ? 0 : st + 1;
  }
  __builtin_amdgcn_s_barrier();     // (drain barrier: wave 3 has finished the last tile's epilogue)
  RGW_RT(2);
}

// ---- wave 3: 4 column tiles, the rows' DMA, every epilogue ------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void rowgemmw_service(const RowGemmArgs &p, float *lds, int lane, int t0, int t1) {
  constexpr int NT = 4, T0 = 15;
  using S = WShape<EPI>;
  const int l16 = lane & 15, q = lane >> 4;
  const int nvert = p.zq_nvert;
  const unsigned nv16 = (unsigned)nvert * 16u;
  float *sA = lds, *sE = lds + kWStages * kWStage;
  uint8_t *msout = reinterpret_cast<uint8_t *>(lds + (kWStages + 2) * kWStage);                    // [2 + 1][2 KiB]
  uint8_t *msin = reinterpret_cast<uint8_t *>(lds + (kWStages + 2) * kWStage + 3 * kWMaskSlot);   // [4 + 1][2 KiB]
  constexpr int NV = 2;   // column tiles whose weights live in VGPRs (152 registers); the other two in AGPRs
  float bv[kWChunks][4][NV], ba[kWChunks][4][NT - NV];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float *br = p.bt + (size_t)((T0 + t) * 16 + l16) * p.ldb + q * 4;
#pragma unroll
    for (int c = 0; c < kWChunks; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(br + c * 16);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
        if (t < NV) bv[c][sidx][t] = v[sidx];
        else ba[c][sidx][t - NV] = v[sidx];
      }
    }
  }
  if (EPI == EPI_FWD_HIDDEN) {   // sign bytes nobody writes (aggregated channels, pad) leave as zeros
    for (int i = lane; i < 2 * kWMaskSlot; i += 64) reinterpret_cast<float *>(msout)[i] = 0.f;
  }

  // Per-lane row state, advanced by 16 rows per tile (no division in the loop): mesh and vertex of this lane's row of the tile
  // whose DMA is issued next (d_*) and of the tile whose epilogue runs next (e_*); from them 32-bit byte offsets (every array
  // of a stack call is far below 4 GB: the launcher checks) to which the per-chunk / per-column-tile parts are SCALAR.
  int d_bq, d_vq, e_bq, e_vq;
  auto locate = [&](int tile, int &bq, int &vq) {
    const int row = tile * 16 + l16;
    bq = row / nvert;
    vq = row - bq * nvert;
  };
  auto advance = [&](int &bq, int &vq) {
    vq += 16;
    const bool wrap = vq >= nvert;
    vq = wrap ? vq - nvert : vq;
    bq = wrap ? bq + 1 : bq;
  };
  unsigned d_off0, d_off1;           // rows' sources: element (row, k = 4 q) of the quad-major / row-major part
  auto refresh_d = [&](int tile, bool on) {
    d_off0 = on ? ((unsigned)(d_bq * (S::KSPLIT / 4) + q) * (unsigned)nvert + (unsigned)d_vq) * 16u : 0u;
    d_off1 = on ? ((unsigned)(tile * 16 + l16) * (unsigned)p.lda1 + (unsigned)q * 4u) * 4u : 0u;
  };
  unsigned e_off0, e_off1, e_off2, e_mrow;   // outputs: quad q of the planes, columns 4 q of the row, sign byte q of the row
  auto refresh_e = [&](int tile, int mslot) {
    e_off0 = ((unsigned)(e_bq * S::ZQ + q) * (unsigned)nvert + (unsigned)e_vq) * 16u;
    e_off1 = ((unsigned)(e_bq * S::YQ + q) * (unsigned)nvert + (unsigned)e_vq) * 16u;
    e_off2 = ((unsigned)(tile * 16 + l16) * (unsigned)p.ldc + (unsigned)q * 4u) * 4u;
    // (forward without a stash: the bytes go to the slot nobody flushes — no branch in the epilogue)
    e_mrow = (unsigned)((EPI == EPI_FWD_HIDDEN && p.maskb == nullptr ? 2 : mslot) * 2048 + l16 * p.mld + p.moff + q);
  };
  // chunk c of the tile behind (d_off0, d_off1) -> ring stage: lane (l16, q) fetches the 16 bytes lane (l16, q) will read.
  // Chunks that lie wholly in one part take a scalar base + the lane's 32-bit offset (no vector instruction).
  // (sz: a scalar zero the compiler cannot see through, renewed per tile — without it the 19 x 3 scalar bases below are hoisted
  // out of the loops and live in spilled scalar registers: 450 v_writelane / 900 v_readlane, vector instructions all)
  int sz = 0;
  auto a_issue = [&](auto cc, int stage, bool on) {
    constexpr int c = decltype(cc)::value;
    float *dst = sA + stage * kWStage + c * 256;
    // (`on` false — past this workgroup's last tile: the zero page, through the scalar base and a zeroed lane offset, no branch)
    const float *b0 = on ? p.a0 + (size_t)c * 4 * nvert * 4 + sz : p.zeros;
    const float *b1 = on ? p.a1 + c * 16 + sz : p.zeros;
    const char *src;
    if (c * 16 + 16 <= S::KSPLIT) {
      src = w_at(b0, d_off0);
    } else if (c * 16 >= S::KSPLIT && c * 16 + 16 <= S::K) {
      src = w_at(b1, d_off1);
    } else {
      const int k0 = c * 16 + q * 4;
      src = k0 >= S::K ? reinterpret_cast<const char *>(p.zeros) : (k0 < S::KSPLIT ? w_at(b0, d_off0) : w_at(b1, d_off1));
    }
    glds16(reinterpret_cast<const float *>(src), dst);
  };
  auto m_issue = [&](int tile, int slot, bool on) {   // EPI_DX_MASK: the tile's sign bytes (16 contiguous rows of mld bytes)
    const uint8_t *src0 = p.maskb + (size_t)tile * 16 * p.mld;
    const int nbytes = on ? 16 * p.mld : 0;
#pragma unroll
    for (int o = 0; o < 2048; o += 1024) {
      const int b = o + lane * 16;
      const void *src = b < nbytes ? (const void *)(src0 + b) : (const void *)p.zeros;
      glds16(reinterpret_cast<const float *>(src), reinterpret_cast<float *>(msin + (on ? slot : 4) * 2048 + o));
    }
  };
  auto m_flush = [&](int tile, int mslot) {   // forward: a finished tile's sign bytes leave as one contiguous block
    const int nbytes = 16 * p.mld;
    uint8_t *dstm = p.maskb + (size_t)tile * 16 * p.mld;
#pragma unroll
    for (int o = 0; o < 2048; o += 1024) {
      const int b = o + lane * 16;
      if (b < nbytes) *reinterpret_cast<f32x4 *>(dstm + b) = *reinterpret_cast<const f32x4 *>(msout + mslot * 2048 + b);
    }
  };
  // Epilogue of column tile T of the tile behind (e_off*): this lane's float4 = columns 4 c4 .. 4 c4 + 3, c4 = 4 T + q.
  // v = max(v, 0) and bit t = v[t] > 0, eleven instructions: after the max a value is +0 or positive, i.e. its bit pattern is 0
  // or a positive integer, and min(pattern, 1) is the sign bit asked for (v_max_f32 returns +0 for max(-0, +0) and the other
  // operand for a NaN: the same values as `v > 0 ? v : 0`).  Written as instructions: from `fmaxf` / `min` the compiler made
  // a canonicalising second v_max per element and a v_cmp_class + v_cndmask + v_mov chain per bit (30 instructions).
  auto relu_bits = [](f32x4 &v) {   // (inputs read in place from the LDS read's registers, outputs fresh: tied operands cost a copy each)
    float o0, o1, o2, o3;
    unsigned b0, b1, b2, b3;
    asm("v_max_f32 %0, 0, %8\n\tv_max_f32 %1, 0, %9\n\tv_max_f32 %2, 0, %10\n\tv_max_f32 %3, 0, %11\n\t"
        "v_min_u32 %4, 1, %0\n\tv_min_u32 %5, 1, %1\n\tv_min_u32 %6, 1, %2\n\tv_min_u32 %7, 1, %3\n\t"
        "v_lshl_or_b32 %4, %5, 1, %4\n\tv_lshl_or_b32 %4, %6, 2, %4\n\tv_lshl_or_b32 %4, %7, 3, %4"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    v = f32x4{o0, o1, o2, o3};
    return b0;
  };
  auto mask4 = [](f32x4 &v, unsigned byte) {   // v[t] = bit t of byte ? v[t] : +0, eight instructions (sign-extended bit AND value)
    float o0, o1, o2, o3;
    asm("v_bfe_i32 %0, %8, 0, 1\n\tv_bfe_i32 %1, %8, 1, 1\n\tv_bfe_i32 %2, %8, 2, 1\n\tv_bfe_i32 %3, %8, 3, 1\n\t"
        "v_and_b32 %0, %0, %4\n\tv_and_b32 %1, %1, %5\n\tv_and_b32 %2, %2, %6\n\tv_and_b32 %3, %3, %7"
        : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(byte));
    v = f32x4{o0, o1, o2, o3};
  };
  // (the staged results and, backward, the sign byte are read BEFORE the chunk's MFMAs and used behind them: with one wave per
  // SIMD an LDS round trip in the epilogue is ~130 idle cycles per column tile)
  auto epi_fetch = [&](auto tc, int eslot, int mslot_in, f32x4 &v, unsigned &byte) {
    constexpr int T = decltype(tc)::value;
    v = *reinterpret_cast<const f32x4 *>(sE + eslot * kWStage + T * 256 + lane * 4);
    byte = EPI == EPI_DX_MASK ? msin[mslot_in * 2048 + (e_mrow & 2047u) + 4 * T] : 0u;
  };
  auto epi = [&](auto tc, f32x4 v, unsigned byte) {
    constexpr int T = decltype(tc)::value;
    const int lo = 4 * T, hi = 4 * T + 3;   // this tile's quads
    if (hi < S::ZQ) {                                        // raw: the aggregation kernel's input, quad-major
      *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.c2 + (size_t)T * 4 * nvert * 4 + sz, e_off0))) = v;
    } else if (EPI == EPI_FWD_HIDDEN && lo >= S::ZQ && hi < S::YQ) {   // activation, quad-major part
      const unsigned bits = relu_bits(v);
      *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.yq + (size_t)T * 4 * nvert * 4 + sz, e_off1))) = v;
      msout[e_mrow + 4 * T] = (uint8_t)bits;
    } else if (lo >= S::YQ && hi * 4 + 3 < S::N) {   // row-major part
      if (EPI == EPI_FWD_HIDDEN) {
        const unsigned bits = relu_bits(v);
        msout[e_mrow + 4 * T] = (uint8_t)bits;
      } else {
        mask4(v, byte);
      }
      *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.c + T * 16 + sz, e_off2))) = v;
    } else {                                                      // a tile that straddles two regions or the last column
      const int c4 = lo + q;
      const bool valid = c4 * 4 < S::N, act = c4 >= S::ZQ;
      if (valid && !act) {
        *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.c2 + (size_t)T * 4 * nvert * 4 + sz, e_off0))) = v;
      } else if (valid) {
        if (EPI == EPI_FWD_HIDDEN) {
          const unsigned bits = relu_bits(v);
          msout[e_mrow + 4 * T] = (uint8_t)bits;
          if (c4 < S::YQ) *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.yq + (size_t)T * 4 * nvert * 4 + sz, e_off1))) = v;
          else *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.c + T * 16 + sz, e_off2))) = v;
        } else {
          mask4(v, byte);
          *reinterpret_cast<f32x4 *>(const_cast<char *>(w_at(p.c + T * 16 + sz, e_off2))) = v;
        }
      }
    }
  };

  // ---- prologue: the first two tiles' rows
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    locate(t0 + d, d_bq, d_vq);
    refresh_d(t0 + d, t0 + d < t1);
    if (EPI == EPI_DX_MASK) m_issue(t0 + d, (t0 + d) & 3, t0 + d < t1);
    [&]<int... C>(std::integer_sequence<int, C...>) { (a_issue(WIdx<C>{}, d, t0 + d < t1), ...); }(std::make_integer_sequence<int, kWChunks>{});
  }
  locate(t0 + 2, d_bq, d_vq);
  refresh_d(t0 + 2, t0 + 2 < t1);
  locate(t0, e_bq, e_vq);
  wait_vmcnt<0>();
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();

  int st = 0;
  // (the first tile has no predecessor whose epilogue to run: its own instantiation, no branch per chunk in the others)
  auto tile_body = [&](int tile, auto firstc) {
    constexpr bool have_prev = !decltype(firstc)::value;
    RGW_STAMP(tile - t0, 0);
    const int par = (tile - t0) & 1;
    const bool have_next = tile + 2 < t1;
    const int nst = st == 0 ? 2 : st - 1;             // (st + 2) % 3: the stage tile - 1 has just left
    const float *sa = sA + st * kWStage + lane * 4;
    asm volatile("s_mov_b32 %0, 0" : "=s"(sz));
    // this tile's bookkeeping, one clump: the epilogue offsets of tile - 1, whose staged results are read below
    if (have_prev) refresh_e(tile - 1, EPI == EPI_FWD_HIDDEN ? (par ^ 1) : 0);
    if (EPI == EPI_DX_MASK) m_issue(tile + 2, (tile + 2) & 3, have_next);
    if (EPI == EPI_FWD_HIDDEN && p.maskb && tile - 2 >= t0) m_flush(tile - 2, par);
    f32x4 acc[NT];
    f32x4 af = *reinterpret_cast<const f32x4 *>(sa);
    auto chunk = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      f32x4 an = af, ev = {0.f, 0.f, 0.f, 0.f};
      unsigned eb = 0;
      if (c + 1 < kWChunks) an = *reinterpret_cast<const f32x4 *>(sa + (c + 1) * 256);
#ifndef A3VT_DBG_RGW_NOEPI
      if (have_prev) epi_fetch(cc, par ^ 1, (tile - 1) & 3, ev, eb);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t < NV) w_mfma(c == 0 && sidx == 0, false, acc[t], bv[c][sidx][t], af[sidx]);
          else w_mfma(c == 0 && sidx == 0, true, acc[t], ba[c][sidx][t - NV], af[sidx]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_mov_b32 %0, 0" : "=s"(sz));   // (per chunk: see the declaration)
#ifndef A3VT_DBG_RGW_NODMA   // timing-only ablations (tools/build_variants.sh rgw): results are wrong by design
      a_issue(cc, nst, have_next);
#endif
#ifndef A3VT_DBG_RGW_NOEPI
      if (have_prev) epi(cc, ev, eb);
#endif
      __builtin_amdgcn_sched_barrier(0);
      af = an;
    };
    [&]<int... C>(std::integer_sequence<int, C...>) { (chunk(WIdx<C>{}), ...); }(std::make_integer_sequence<int, kWChunks>{});
    {
      float *se = sE + par * kWStage + lane * 4;
      w_mfma_done();   // (the last chunk's side work has passed, but nothing the compiler knows of orders it)
#pragma unroll
      for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4 *>(se + (T0 + t) * 256) = acc[t];
    }
    advance(d_bq, d_vq);
    refresh_d(tile + 3, tile + 3 < t1);
    if (have_prev) advance(e_bq, e_vq);
    RGW_STAMP(tile - t0, 1);
    // the rows of tile + 1 were requested a tile ago: everything but this tile's last 24 memory instructions has landed
    wait_vmcnt<24>();
    wait_lgkm0();
    RGW_STAMP(tile - t0, 2);
    __builtin_amdgcn_s_barrier();
    RGW_STAMP(tile - t0, 3);
    st = st == 2 ? 0 : st + 1;
  };
  if (t1 > t0) tile_body(t0, std::true_type{});
  for (int tile = t0 + 1; tile < t1; ++tile) tile_body(tile, std::false_type{});
  // ---- drain: the last tile's epilogue, the sign bytes still in LDS
  if (t1 > t0) {
    const int last = t1 - 1, lpar = (last - t0) & 1;
    asm volatile("s_mov_b32 %0, 0" : "=s"(sz));
    refresh_e(last, EPI == EPI_FWD_HIDDEN ? lpar : 0);
    [&]<int... C>(std::integer_sequence<int, C...>) {
      ((void)[&] {
        f32x4 ev;
        unsigned eb;
        epi_fetch(WIdx<C>{}, lpar, last & 3, ev, eb);
        epi(WIdx<C>{}, ev, eb);
      }(), ...);
    }(std::make_integer_sequence<int, kWChunks>{});
    wait_lgkm0();
    if (EPI == EPI_FWD_HIDDEN && p.maskb) {
      if (last - 1 >= t0) m_flush(last - 1, lpar ^ 1);
      m_flush(last, lpar);
    }
  }
  __builtin_amdgcn_s_barrier();
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rowgemmw_kernel(RowGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the main part: the same number of 16-row tiles for every workgroup; the rest is the launch's tail (below)
  const int per = (p.m >> 4) / (int)gridDim.x;
  const int t0 = blockIdx.x * per, t1 = t0 + per;
  RGW_RT(0);
  if (wave < 3) rowgemmw_compute<EPI>(p, lds, wave, lane, t0, t1);
  else rowgemmw_service<EPI>(p, lds, lane, t0, t1);
  // ---- leftover tiles of the even split: one 16 x 16 output tile per wave, dealt across the workgroups (rowtile_unit)
  if (p.rem_rows > 0) {
    const int ntl = (p.n_store + 15) >> 4;
    const int units = ((p.rem_rows + 15) >> 4) * ntl;
    for (int u = wave * gridDim.x + blockIdx.x; u < units; u += 4 * gridDim.x)
      rowtile_unit<EPI, 0>(p, p.rem_row0, p.rem_row0 + p.rem_rows, u / ntl, (u % ntl) * 16, lane);
  }
  RGW_RT(3);
}

// The shapes this kernel takes: exact fp32, a hidden layer of a stack on hybrid rows (quad-major a0 and side outputs),
// 19 column tiles x 19 K chunks, whole meshes.
bool rowgemmw_ok(const RowGemmArgs &a, int epi) {
#ifdef A3VT_DBG_RGW_OFF   // variant build (tools/build_variants.sh rgw): the round-5 kernel everywhere, for A/B timing
  return false;
#endif
  if (a.bf16 != 0 || (epi != EPI_FWD_HIDDEN && epi != EPI_DX_MASK)) return false;
  if (a.zq_nvert <= 0 || a.a0q_nvert != a.zq_nvert || a.c2 == nullptr || a.m % a.zq_nvert != 0 || a.m % 16 != 0) return false;
  if (a.k <= 288 || a.k > 304 || a.n_store <= 288 || a.n_store > 304 || a.n_store % 4 != 0 || a.k % 4 != 0) return false;
  if (a.ksplit != a.a0q_quads * 4 || a.ldc % 4 != 0 || a.lda1 % 4 != 0 || a.ldb < 304) return false;
  if (a.mld > 128 || a.mld % 4 != 0 || a.zq_quads * 4 > a.n_store) return false;
  if (epi == EPI_FWD_HIDDEN && (a.yq_quads < a.zq_quads || (a.yq_quads > a.zq_quads && a.yq == nullptr))) return false;
  if (epi == EPI_DX_MASK && a.maskb == nullptr) return false;
  if (a.trash == nullptr || a.zq_nvert < 16 || a.m > 3000000) return false;   // (32-bit byte offsets inside every array)
  // the compiled shape (WShape): the reference's hidden layers on hybrid rows
  if (a.k != 300 || a.n_store != 300 || a.zq_quads != 25 || a.no_relu) return false;
  if (epi == EPI_FWD_HIDDEN && (a.yq_quads != 40 || a.ksplit != 160)) return false;
  if (epi == EPI_DX_MASK && a.ksplit != 100) return false;
  return (a.m >> 4) >= 768;   // = the 12 288 rows from which a stack runs on hybrid rows at all (capi.hip use_csrq)
}

template <int EPI>
static int launch_rowgemmw_epi(const RowGemmArgs &a0, hipStream_t s) {
  RowGemmArgs a = a0;
  a.bt_rows = rowgemm_bt_rows(a.n_store);
  a.col0 = 0;
  const int ncu = (a.m >> 4) >= 4 * 256 ? 256 : (a.m >> 4) / 4;   // fewer, fuller workgroups for few tiles
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)rowgemmw_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(kWLdsFloats * sizeof(float)));
  });
  const int tiles = a.m >> 4, per = tiles / ncu;
  a.rem_row0 = per * ncu * 16;
  a.rem_rows = a.m - a.rem_row0;
  path_count(PATH_RGW);
  A3VT_LAUNCH((rowgemmw_kernel<EPI>), dim3(ncu), dim3(256), kWLdsFloats * sizeof(float), s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

#ifdef A3VT_DBG_RGW_STAMPS
}  // namespace a3vt
extern "C" int a3vt_dbg_rgw_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(a3vt::g_rgw_stamps), sizeof(unsigned long long) * 256 * 4 * 48 * 4);
}
extern "C" int a3vt_dbg_rgw_rt(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(a3vt::g_rgw_rt), sizeof(unsigned long long) * 256 * 4);
}
namespace a3vt {
#endif

int launch_rowgemmw(const RowGemmArgs &a, int epi, hipStream_t s) {
  if (!rowgemmw_ok(a, epi)) {
    set_error("rowgemmw: shape not taken (m=%d k=%d n=%d epi=%d)", a.m, a.k, a.n_store, epi);
    return -1;
  }
  return epi == EPI_FWD_HIDDEN ? launch_rowgemmw_epi<EPI_FWD_HIDDEN>(a, s) : launch_rowgemmw_epi<EPI_DX_MASK>(a, s);
}

}  // namespace a3vt
