// gcn_dww.hip — dW = X^T dZ of the hidden layers, exact fp32, on specialised waves (gfx950, round 6).
//
// Replaces autograd's weight gradient of torch.matmul(features, self.weight) (reconstruction/vision/model.py:352) for the
// 300 x 300 layers of a stack on hybrid rows — the shape dw_kernel<fast, hybrid> served at 0.64 of the fp32 matrix peak, the
// weakest of the three product kernels since round 3 and "unexplained by anything we can stamp" (rounds 4-5: not the
// tile imbalance, not X staged twice).  What explains it (tools/ubench/mfma_gap_budget.hip): on gfx950 the vector ALU shares
// the SIMD's fp32 pipe with the fp32 MFMA, so every vector instruction in the matrix stream — the per-stage address
// arithmetic of 16 waves — costs matrix time; scalar instructions and LDS reads do not.  The recipe of rowgemmw_kernel
// (gcn_gemmw.hip) applied to the reduction over rows:
//   * 256 workgroups of 4 waves, one wave per SIMD.  A workgroup owns a row group (1/128 of the rows) and ONE HALF of the
//     output columns (n-tiles 0..9 / 10..18: the same 128 slab images as dw_kernel's two column groups, summed by the same
//     slab_reduce); its accumulators — 19 k-tiles x 10 (9) n-tiles — never leave the registers until the end of the launch.
//   * waves 0-2 own 5 k-tiles each (X columns 0-79 / 80-159: quad-major planes; 160-239: row-major), wave 3 owns 4 and issues
//     ALL the LDS-DMA: per 16-row stage 10 + 9 instructions for X and 11 (9) for this half of dZ, every address a scalar base
//     + a lane offset that is constant (row-major pieces) or advanced once per stage (quad-major planes: mesh / vertex of the
//     lane's row).  Three stages in flight, one barrier per stage.
//   * per 4-row k-step a wave reads 5 + 10 operand fragments (ds_read_b32 at immediate offsets from four lane bases) and
//     issues 50 MFMAs, written as instructions (accumulators in AGPRs; see gcn_gemmw.hip).  Transposed product
//     (a = dZ fragment, b = X fragment): a lane ends with four consecutive n of one k — 16-byte slab stores.
// Sums run over a workgroup's rows in ascending order and over the 128 slab images in slab_reduce's order — the same terms
// as dw_kernel in another association (its row groups are cut elsewhere): equal to fp32 rounding, not bit for bit.
#include <type_traits>

#include "gemm_tile.h"

namespace a3vt {

// compiled shape: hidden 300, hybrid rows (X: 40 quad-major quads + 140 row-major columns; dZ: 25 quad-major quads of dZa +
// columns 100..299 of the row-major gradient block)
constexpr int kDK = 300, kDN = 300, kDXQ = 40, kDZQ = 25, kDXRW = 140, kDLdz = 300;
constexpr int kDStageRows = 16;
constexpr int kDwwGroupsA = 135, kDwwGroupsB = 121;   // = 256 workgroups; kDwwGroupsA slab images
// LDS stage (floats): XQ [40][16] float4 | XR [16][140] (+pad to 9 KiB) | Z: half A: ZQ [25][16] float4 (+pad to 7 KiB), ZR [16][60]
// (+pad to 4 KiB); half B: ZR [16][140] (+pad to 9 KiB)
constexpr int kDXQ_off = 0, kDXR_off = 2560, kDZ_off = 2560 + 2304;
constexpr int kDZRa_off = kDZ_off + 1792;
constexpr int kDStage = kDZ_off + 2816;   // 7680 floats = 30 KiB
constexpr int kDStages = 4;
constexpr int kDLdsFloats = kDStages * kDStage;

#ifdef A3VT_DBG_RGW_STAMPS   // diagnostic build (tools/build_variants.sh rgw): s_memtime per stage, [workgroup][wave][stage < 96][2]
__device__ unsigned long long g_dww_stamps[256 * 4 * 96 * 2];
#define DWW_STAMP(tl, k)                                                                                       \
  do {                                                                                                         \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (tl) < 96)                                              \
      g_dww_stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 96 + (tl)) * 2 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
// wall-clock (s_memrealtime, 100 MHz) of wave 0 at: kernel entry, first stage, behind the last stage, kernel end
__device__ unsigned long long g_dww_rt[256 * 4];
#define DWW_RT(k)                                                                                              \
  do {                                                                                                         \
    if (threadIdx.x == 0 && blockIdx.x < 256) g_dww_rt[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DWW_STAMP(tl, k) do { } while (0)
#define DWW_RT(k) do { } while (0)
#endif

// MFMA with the accumulator in an AGPR tuple, operands in VGPRs (see w_mfma in gcn_gemmw.hip)
// (volatile: the order written below — one fragment read of the NEXT k-step behind every third MFMA — is the schedule)
__device__ __forceinline__ void d_mfma(f32x4 &acc, float a, float b) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// One fragment element from the stage, as a written instruction: from a C++ load the compiler places "s_waitcnt lgkmcnt(0)" in
// front of the MFMAs of every second k-step — behind the reads it has just issued for the NEXT step, i.e. a full LDS round
// trip (with the 4-way bank conflicts of the quad-major planes) exposed every other step: 36-37 cycles per MFMA instead of
// 33.  Here the wait is written out and COUNTED (d_wait): LDS returns in order, the fragments of the current step are
// complete when at most the reads issued behind them are outstanding.
__device__ __forceinline__ void d_lds(float &dst, unsigned base, int imm) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(imm));
}
// s_waitcnt lgkmcnt(N) that the consumers of x[..] / z[..] cannot be moved in front of (they take its outputs)
template <int N, int NX, int NZ>
__device__ __forceinline__ void d_wait(float (&x)[NX], float (&z)[NZ]) {
  static_assert(NX <= 5 && NZ <= 10, "operand list");
  if constexpr (NX == 5 && NZ == 10)
    asm volatile("s_waitcnt lgkmcnt(%15)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]),
                 "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]), "+v"(z[9]) : "n"(N));
  else if constexpr (NX == 4 && NZ == 10)
    asm volatile("s_waitcnt lgkmcnt(%14)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]),
                 "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]), "+v"(z[9]) : "n"(N));
  else if constexpr (NX == 5 && NZ == 9)
    asm volatile("s_waitcnt lgkmcnt(%14)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]),
                 "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%13)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(z[0]), "+v"(z[1]), "+v"(z[2]),
                 "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(z[6]), "+v"(z[7]), "+v"(z[8]) : "n"(N));
}

// HALF: 0 = n-tiles 0..9 (dZ columns 0..159), 1 = n-tiles 10..18 (columns 160..303).  XROW: this wave's k-tiles lie in the
// row-major part of X.  NKT: k-tiles of this wave, from tile KT0.
template <int HALF, bool XROW, int NKT, bool SERVICE, int KT0>
__device__ __forceinline__ void dww_wave(const DwArgs &p, float *lds, int lane, int tile0, int tile1, int image) {
  constexpr int kt0 = KT0;
  constexpr int kt0x = XROW ? KT0 - 10 : KT0;   // tile index inside its part of X (the row-major part starts at k-tile 10)
  constexpr int NNT = HALF == 0 ? 10 : 9, NT0 = HALF == 0 ? 0 : 10;
  const int l16 = lane & 15, q = lane >> 4;
  const int nvert = p.xq_nvert;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;   // LDS byte address of the ring

  // ---- fragment addresses inside a stage (bytes): lane base + (tile, step) immediates
  // X (b operand: j = l16 <-> k column, kk = q <-> row of the step)
  const unsigned xq_lane = kDXQ_off * 4 + (unsigned)((l16 >> 2) * 256 + q * 16 + (l16 & 3) * 4);
  const unsigned xr_lane = kDXR_off * 4 + (unsigned)((q * kDXRW + l16) * 4);
  // dZ (a operand: i = l16 <-> n column)
  const unsigned zq_lane = kDZ_off * 4 + (unsigned)((l16 >> 2) * 256 + q * 16 + (l16 & 3) * 4);
  const unsigned zra_lane = kDZRa_off * 4 + (unsigned)((q * 60 + l16) * 4);
  const unsigned zrb_lane = kDZ_off * 4 + (unsigned)((q * kDXRW + l16) * 4);
  // n-tile 6 (columns 96..111) straddles the quad-major / row-major split of dZ at column 100
  const bool t6q = l16 < 4;
  const unsigned z6_lane = t6q ? kDZ_off * 4 + (unsigned)(24 * 256 + q * 16 + l16 * 4) : kDZRa_off * 4 + (unsigned)((q * 60 + l16 - 4) * 4);
  const unsigned z6_step = t6q ? 64u : 960u;

  // ---- service wave: DMA sources.  Row-major pieces: lane-constant offsets from a per-stage scalar base; quad-major planes:
  // (mesh, vertex) of the lane's row, advanced by 16 rows per stage.
  unsigned xr_off[9], zr_off[9];
  int d_bq = 0, d_vq = 0;
  unsigned xq_off = 0, zq_off = 0;
  auto locate = [&](int tile) {
    const int row = tile * kDStageRows + l16;
    d_bq = row / nvert;
    d_vq = row - d_bq * nvert;
  };
  auto refresh = [&]() {
    xq_off = ((unsigned)(d_bq * kDXQ + q) * (unsigned)nvert + (unsigned)d_vq) * 16u;
    zq_off = ((unsigned)(d_bq * kDZQ + q) * (unsigned)nvert + (unsigned)d_vq) * 16u;
  };
  auto advance = [&]() {
    d_vq += kDStageRows;
    const bool wrap = d_vq >= nvert;
    d_vq = wrap ? d_vq - nvert : d_vq;
    d_bq = wrap ? d_bq + 1 : d_bq;
  };
  if (SERVICE) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int i = 64 * j + lane;
      const int rx = i / 35, px = i - rx * 35;
      xr_off[j] = i < 16 * 35 ? (unsigned)((rx * p.ldx_src + px * 4) * 4) : 0u;
      if (HALF == 0) {
        const int rz = i / 15, pz = i - rz * 15;
        zr_off[j] = (j < 4 && i < 16 * 15) ? (unsigned)((rz * kDLdz + pz * 4) * 4) : 0u;
      } else {
        zr_off[j] = i < 16 * 35 ? (unsigned)((rx * kDLdz + px * 4) * 4) : 0u;
      }
    }
  }
  // DMA instruction `idx` of the stage holding 16-row tile `tile` -> ring stage `st`
  constexpr int NDMA = HALF == 0 ? 10 + 9 + 7 + 4 : 10 + 9 + 9;
  // (`tile` past this row group's end: the caller passes the group's last tile again — valid addresses, data nobody reads;
  // no pointer is ever selected against the zero page: the compiler turns such a select into a table in scratch memory)
  auto dma = [&](int idx, int tile, int st) {
    float *sbase = lds + st * kDStage;
    const size_t row0 = (size_t)tile * kDStageRows;
    if (idx < 10) {   // X quad-major planes 4 idx .. 4 idx + 3
      glds16(reinterpret_cast<const float *>(w_at(p.xq + (size_t)idx * 4 * nvert * 4, xq_off)), sbase + kDXQ_off + idx * 256);
    } else if (idx < 19) {   // X row-major columns 160..299
      const int j = idx - 10;
      glds16(reinterpret_cast<const float *>(w_at(p.x + row0 * p.ldx_src + 160, xr_off[j])), sbase + kDXR_off + j * 256);
    } else if (HALF == 0 && idx < 26) {   // dZa quad-major planes (25: the last instruction re-reads plane 24 into the pad)
      const int j = idx - 19;
      const unsigned off = j < 6 ? zq_off : zq_off - (unsigned)q * (unsigned)nvert * 16u;
      glds16(reinterpret_cast<const float *>(w_at(p.z0 + (size_t)(j < 6 ? j * 4 : 24) * nvert * 4, off)), sbase + kDZ_off + j * 256);
    } else if (HALF == 0) {   // gradient block columns 100..159
      const int j = idx - 26;
      glds16(reinterpret_cast<const float *>(w_at(p.z1 + row0 * kDLdz + 100, zr_off[j])), sbase + kDZRa_off + j * 256);
    } else {   // gradient block columns 160..299
      const int j = idx - 19;
      glds16(reinterpret_cast<const float *>(w_at(p.z1 + row0 * kDLdz + 160, zr_off[j])), sbase + kDZ_off + j * 256);
    }
  };

  if (SERVICE) {   // prologue: the first three tiles
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int tsrc = tile0 + d < tile1 ? tile0 + d : tile1 - 1;
      locate(tsrc);
      refresh();
#pragma unroll
      for (int idx = 0; idx < NDMA; ++idx) dma(idx, tsrc, d);
    }
    locate(tile0 + 2 < tile1 ? tile0 + 2 : tile1 - 1);
    refresh();   // (the first barrier of the loop is followed by the loads of tile0 + 2)
  }
  wait_vmcnt<0>();
  wait_lgkm0();
  __builtin_amdgcn_s_barrier();

  f32x4 acc[NKT][NNT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int n = 0; n < NNT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  int st = 0;
  // The fragments of a k-step are requested one step ahead, and the barrier of a stage sits between its steps 2 and 3, so that
  // step 3 already requests step 0 of the NEXT stage.  Four buffers: behind the barrier of tile t the service wave starts the
  // loads of tile t + 3 (into the buffer tile t - 1 has just left) and spreads them over step 3 of t and steps 0-2 of t + 1;
  // they must have landed at the barrier of t + 2 — a whole stage later.
  // fragment k of k-step s of the stage in buffer `stg`: k < NKT: X of k-tile kt0 + k; else dZ of n-tile NT0 + k - NKT
  auto frag = [&](int stg, auto sc, auto kc, float (&xf)[NKT], float (&zf)[NNT]) {
    constexpr int s = decltype(sc)::value, k = decltype(kc)::value;
    const unsigned sb = lds0 + (unsigned)(stg * kDStage * 4);
    if constexpr (k < NKT) {
      // quad-major planes (1 KiB per tile, 64 B per step) or row-major columns (64 B per tile, 2240 B per step)
      const unsigned xb = sb + (XROW ? xr_lane : xq_lane);
      if (XROW) d_lds(xf[k], xb, (kt0x + k) * 64 + s * 4 * kDXRW * 4);
      else d_lds(xf[k], xb, (kt0x + k) * 1024 + s * 64);
    } else if constexpr (k < NKT + NNT) {
      constexpr int n = k - NKT, T = NT0 + n;
      if (HALF == 0 && T < 6) d_lds(zf[n], sb + zq_lane, T * 1024 + s * 64);
      else if (HALF == 0 && T == 6) d_lds(zf[n], sb + z6_lane + (unsigned)s * z6_step, 0);
      else if (HALF == 0) d_lds(zf[n], sb + zra_lane, (T * 16 - 100) * 4 + s * 4 * 60 * 4);
      else d_lds(zf[n], sb + zrb_lane, (T * 16 - 160) * 4 + s * 4 * kDXRW * 4);
    }
  };
  auto dma_part = [&](int part, int dtile, int dst) {
    constexpr int PER = (NDMA + 3) / 4;
#pragma unroll
    for (int u = 0; u < PER; ++u)
      if (part * PER + u < NDMA) dma(part * PER + u, dtile < tile1 ? dtile : tile1 - 1, dst);
  };
  float xc[NKT], zc[NNT];
  [&]<int... K>(std::integer_sequence<int, K...>) {
    (frag(0, std::integral_constant<int, 0>{}, std::integral_constant<int, K>{}, xc, zc), ...);
  }(std::make_integer_sequence<int, NKT + NNT>{});
  d_wait<0>(xc, zc);
  DWW_RT(1);
  for (int tile = tile0; tile < tile1; ++tile) {
    DWW_STAMP(tile - tile0, 0);
    auto step = [&](auto sc) {
      constexpr int s = decltype(sc)::value;
      float xn[NKT], zn[NNT];
      constexpr int EVERY = (NKT * NNT) / (NKT + NNT);   // 3 (50 or 45 MFMAs, 15 or 14 fragments) or 2 (40 or 36, 14 or 13)
      // 50 MFMAs; behind every third one a fragment of the next k-step (step 3: step 0 of the next stage, behind the barrier):
      // issued in one burst the 15 reads — 4-way bank conflicts on the quad-major planes — held up the MFMAs behind them
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ((d_mfma(acc[I / NNT][I % NNT], zc[I % NNT], xc[I / NNT]),
          (I % EVERY == 0 ? frag(s < 3 ? st : ((st + 1) & 3), std::integral_constant<int, (s + 1) & 3>{}, std::integral_constant<int, I / EVERY>{}, xn, zn)
                          : (void)0)),
         ...);
      }(std::make_integer_sequence<int, NKT * NNT>{});
      d_wait<0>(xn, zn);   // (the last read was issued >= 5 MFMAs ago)
#pragma unroll
      for (int t = 0; t < NKT; ++t) xc[t] = xn[t];
#pragma unroll
      for (int n = 0; n < NNT; ++n) zc[n] = zn[n];
    };
    step(std::integral_constant<int, 0>{});
    if (SERVICE && tile > tile0) dma_part(1, tile + 2, (st + 2) & 3);
    step(std::integral_constant<int, 1>{});
    if (SERVICE && tile > tile0) dma_part(2, tile + 2, (st + 2) & 3);
    step(std::integral_constant<int, 2>{});
    if (SERVICE) {
      if (tile > tile0) dma_part(3, tile + 2, (st + 2) & 3);
      // tile + 1's loads were issued before the previous barrier; behind them only tile + 2's (loads return in order)
      wait_vmcnt<NDMA>();
    }
    DWW_STAMP(tile - tile0, 1);
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();   // tile + 1 is in the ring; everyone has left tile - 1's buffer
    step(std::integral_constant<int, 3>{});
    if (SERVICE) {
      if (tile + 3 < tile1) {   // rows of the next DMA tile (past the end the offsets stay on the group's last tile)
        advance();
        refresh();
      }
      dma_part(0, tile + 3, (st + 3) & 3);
    }
    st = (st + 1) & 3;
  }
  DWW_RT(2);
  // ---- this workgroup's partial sums -> its half of slab image `image`: dW[k = 16 Tk + l16][n = 16 Tn + 4 q + r]
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // (the last MFMAs' results: see w_mfma_done in gcn_gemmw.hip)
  float *img = p.slab + (size_t)image * kDK * kDN;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int k = (kt0 + t) * 16 + l16;
#pragma unroll
    for (int n = 0; n < NNT; ++n) {
      const int col = (NT0 + n) * 16 + q * 4;
      if (k < kDK && col < kDN) *reinterpret_cast<f32x4 *>(img + (size_t)k * kDN + col) = acc[t][n];
    }
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void dww_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Half A multiplies 10 n-tiles per k-tile, half B 9: row groups of equal size would leave half the CUs idle for a tenth of
  // the launch.  kDwwGroupsA : kDwwGroupsB = 135 : 121 ~ 10 : 9 workgroups share the rows in each half; slab image i holds
  // A group i's columns 0..159 and B group i's columns 160..299 (different rows: only the sum over the images means
  // anything), and the B halves of images 121..134 are zeros written by the first 14 B workgroups.
  const int half = blockIdx.x < kDwwGroupsA ? 0 : 1;
  const int group = half == 0 ? blockIdx.x : blockIdx.x - kDwwGroupsA, ngroups = half == 0 ? kDwwGroupsA : kDwwGroupsB;
  // 16-row tiles of this row group (the groups of a half differ by at most one tile)
  const int tiles = p.m / kDStageRows;
  const int t0 = (int)((long long)group * tiles / ngroups), t1 = (int)((long long)(group + 1) * tiles / ngroups);
  DWW_RT(0);
  if (half == 1 && group < kDwwGroupsA - kDwwGroupsB) {
    float *img = p.slab + (size_t)(kDwwGroupsB + group) * kDK * kDN;
    for (int i = threadIdx.x; i < kDK * 35; i += 256) {
      const int k = i / 35, c = i - k * 35;
      *reinterpret_cast<f32x4 *>(img + (size_t)k * kDN + 160 + c * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (half == 0) {
    if (wave == 0) dww_wave<0, false, 5, false, 0>(p, lds, lane, t0, t1, group);
    else if (wave == 1) dww_wave<0, false, 5, false, 5>(p, lds, lane, t0, t1, group);
    else if (wave == 2) dww_wave<0, true, 5, false, 10>(p, lds, lane, t0, t1, group);
    else dww_wave<0, true, 4, true, 15>(p, lds, lane, t0, t1, group);
  } else {
    if (wave == 0) dww_wave<1, false, 5, false, 0>(p, lds, lane, t0, t1, group);
    else if (wave == 1) dww_wave<1, false, 5, false, 5>(p, lds, lane, t0, t1, group);
    else if (wave == 2) dww_wave<1, true, 5, false, 10>(p, lds, lane, t0, t1, group);
    else dww_wave<1, true, 4, true, 15>(p, lds, lane, t0, t1, group);
  }
  DWW_RT(3);
}

// The shape this kernel takes: exact fp32, a hidden layer of a stack on hybrid rows (dw_kernel<fast, hybrid>'s), enough rows.
bool dww_ok(const DwArgs &a) {
#ifdef A3VT_DBG_DWW_OFF   // variant build (tools/build_variants.sh rgw): dw_kernel everywhere, for A/B timing
  return false;
#endif
  if (a.bf16 != 0 || a.xq_nvert <= 0 || a.z0q_nvert != a.xq_nvert || a.xq == nullptr) return false;
  if (a.k_in != kDK || a.n_out != kDN || a.xq_quads != kDXQ || a.z0q_quads != kDZQ || a.zsplit != 4 * kDZQ) return false;
  if (a.ldx_src != kDXRW || a.ldz1 != kDLdz || a.m % a.xq_nvert != 0 || a.m % kDStageRows != 0 || a.xq_nvert < 16) return false;
  if (a.m > 3000000) return false;                    // 32-bit byte offsets inside every array
  return a.m / kDStageRows >= kDwwGroupsA * 5;        // at least five stages per row group (12 288 rows: where hybrid rows start)
}

#ifdef A3VT_DBG_RGW_STAMPS
}  // namespace a3vt
extern "C" int a3vt_dbg_dww_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(a3vt::g_dww_stamps), sizeof(unsigned long long) * 256 * 4 * 96 * 2);
}
extern "C" int a3vt_dbg_dww_rt(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(a3vt::g_dww_rt), sizeof(unsigned long long) * 256 * 4);
}
namespace a3vt {
#endif

int dww_images() { return kDwwGroupsA; }

int launch_dww(const DwArgs &a, hipStream_t s) {
  if (!dww_ok(a)) {
    set_error("dww: shape not taken (m=%d k_in=%d n_out=%d)", a.m, a.k_in, a.n_out);
    return -1;
  }
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)dww_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kDLdsFloats * sizeof(float)));
  });
  path_count(PATH_DWW);
  A3VT_LAUNCH(dww_kernel, dim3(kDwwGroupsA + kDwwGroupsB), dim3(256), kDLdsFloats * sizeof(float), s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
