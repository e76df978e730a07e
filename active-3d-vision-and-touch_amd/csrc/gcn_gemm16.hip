// gcn_gemm16.hip — the per-vertex product of the bf16 STORAGE mode (gemm mode 2) with the weights held in REGISTERS.
//
//   C[M][n] = A[M][K] * Bt[n][K]^T,  A / Bt / C bf16, fp32 accumulation,  K <= 320, n <= 304
//   (torch.matmul(features, W) of GCN_layer.forward, vision/model.py:352, and autograd's dX = dZ W^T)
//
// Why a second kernel: in this mode rowgemm_kernel (gcn_gemm.hip) is bound by its operand stream, not by the matrix
// pipe — a 256-row round stages 78 KB of A and, again, all 185 KB of the weight image (verdict r02 #7: "the larger half
// of the DMA bytes"), 87 / 68 us per launch where the bytes of A and C take 32 us at the rate a copy reaches.
// Here a workgroup (8 waves, one per CU, persistent) loads the weight image ONCE: wave w keeps the B fragments of the
// column tiles w, w + 8, w + 16 for all ten k-steps in 120 registers.  Rows stream through a 3-stage LDS ring in blocks
// of 48 (LDS-DMA, 16-row x 64-byte pieces in the bank-swizzled order of rowgemm_kernel); every wave reads every A
// fragment (ds_read_b128 = the operand of v_mfma_f32_16x16x32_bf16) and multiplies it with its own column tiles.  The
// accumulators of a block cross an fp32 LDS tile so that whole rows leave with 16-byte stores; the fused epilogues
// (forward: raw aggregated columns + ReLU'd pass-through columns + sign bytes; backward: ReLU-sign multiply) are the
// ones of rowgemm_kernel, value for value.
//
// Waves 0-3 issue the LDS-DMA; all eight run the MFMAs and store 6 rows of every block each (lane = 8-column group: 38
// active lanes write one 608-byte row per instruction, straight-line code).  The DMA waves' counted vmcnt waits step over
// their own stores of the previous block (a fixed number per block; loads, LDS-DMA and stores retire in issue order on the
// one counter).  Two workgroup barriers per block.
#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u16 = unsigned short;

constexpr int kRows = 48;                  // rows per block (3 tiles of 16)
constexpr int kRT = kRows / 16;
constexpr int kKS = 10;                    // k-steps of 32 bf16 (K <= 320)
constexpr int kStages = 3;
constexpr int kStage = kRT * kKS * 256;    // floats per ring stage: 30 pieces of 1 KiB
constexpr int kEpLd = 308;                 // floats per staged output row (304 + 4: rows 4 apart on different banks)
constexpr int kEp = kRows * kEpLd;
constexpr int kMaskSlot = 1280;            // floats (5120 B) per sign-byte slot: 48 rows x mld <= 106 bytes
constexpr int kLdsFloats = kStages * kStage + kEp + 2 * kMaskSlot;   // 161,536 B
constexpr int kAInstr = kRT * kKS;         // LDS-DMA wave-instructions per block of A

__device__ __forceinline__ void glds16(const float *gsrc, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ f32x4 mfma(f32x4 a, f32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 pack8(f32x4 lo, f32x4 hi) {   // 8 floats -> 8 bf16 (RNE, v_cvt_pk_bf16_f32)
  const bf16x2 a = __builtin_convertvector((f32x2){lo[0], lo[1]}, bf16x2), b = __builtin_convertvector((f32x2){lo[2], lo[3]}, bf16x2);
  const bf16x2 c = __builtin_convertvector((f32x2){hi[0], hi[1]}, bf16x2), d = __builtin_convertvector((f32x2){hi[2], hi[3]}, bf16x2);
  return __builtin_bit_cast(f32x4, (u32x4){__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b),
                                           __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)});
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// all but the n youngest vector-memory operations of this wave are done (n <= 26: the A pieces of one block + one block's stores)
__device__ __forceinline__ void wait_younger(int n) {
  switch (n) {
#define A3VT_W(N) case N: wait_vmcnt<N>(); break;
    A3VT_W(7) A3VT_W(8) A3VT_W(13) A3VT_W(14) A3VT_W(19) A3VT_W(20) A3VT_W(25) A3VT_W(26)
#undef A3VT_W
    default: wait_vmcnt<0>(); break;
  }
}

#ifdef A3VT_DBG_R16_STAMPS   // diagnostic build (tools/build_variants.sh stamps16): s_memrealtime (100 MHz) at the phase boundaries
__device__ unsigned long long g_r16_stamps[2 * 256 * 16 * 8];   // [epilogue][workgroup][block (< 16)][8]; a3vt_dbg_r16_stamps
#define R16_STAMP(w, k)                                                                                      \
  do {                                                                                                       \
    if (wave == (w) && lane == 0 && blockIdx.x < 256 && b < 16)                                              \
      g_r16_stamps[(((EPI == EPI_DX_MASK ? 1 : 0) * 256 + blockIdx.x) * 16 + b) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define R16_STAMP(w, k) do { } while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(512, 2) void rowgemm16_kernel(RowGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *ep = lds + kStages * kStage;
  float *mask_lds = ep + kEp;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, q = lane >> 4;
  const int qs = q ^ ((l16 >> 1) & 2);                       // reader side of the bank swizzle (rowgemm_kernel)
  const int drow = lane >> 2;                                // DMA lane: row of the 16-row tile ...
  const int dk = ((lane & 3) ^ ((lane >> 3) & 2)) * 4;       // ... and float offset of its 16-byte piece in the 64-byte k-step
  const bool dma_wave = wave < 4;
  const int nt = (p.n_store + 15) >> 4;
  const bool third = wave + 16 < nt;                         // wave-uniform: this wave owns a third column tile

#ifdef A3VT_DBG_R16_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 256)   // kernel entry (block slot 15 of the stamp array)
    g_r16_stamps[(((EPI == EPI_DX_MASK ? 1 : 0) * 256 + blockIdx.x) * 16 + 15) * 8] = __builtin_amdgcn_s_memrealtime();
#endif
  // blocks of kRows rows, dealt evenly to the workgroups
  const int nblk_all = (p.m + kRows - 1) / kRows;
  const int bbase = nblk_all / gridDim.x, brem = nblk_all % gridDim.x;
  const int blk0 = blockIdx.x * bbase + ((int)blockIdx.x < brem ? blockIdx.x : brem);
  const int nblk = bbase + ((int)blockIdx.x < brem ? 1 : 0);

  // LDS-DMA of a block: A pieces 0 .. 29 (piece = row tile * 10 + k-step) dealt round-robin to waves 0-3 — a wave issues
  // the same number (8, 8, 7, 7) for every block, which is what its counted waits rely on.
  const int mine_a = dma_wave ? (kAInstr - wave + 3) >> 2 : 0;
  // store instructions a wave issues per (full) block — 6 rows: the activation row, the raw row, the sign bytes
  const bool has_z = EPI == EPI_FWD_HIDDEN && p.csplit > 0;
  const int store_instr = EPI == EPI_FWD_HIDDEN ? 6 * (1 + (has_z ? 1 : 0) + (p.maskb ? 1 : 0)) : 6;
  // t-th A piece of this wave (piece = wave + 4 t) of block b: row pointers from per-lane bases, everything else scalar
  auto issue_piece = [&](int b, int stage, int t) {
    const int pc = wave + 4 * t;
    if (pc >= kAInstr) return;                               // wave-uniform
    const int rt = pc / kKS, s = pc - rt * kKS;
    const int r = (blk0 + b) * kRows + rt * 16 + drow;
    const int kk = s * 16 + dk;                              // floats (pairs of bf16) along K
    const bool left = kk < p.ksplit;                         // (selects, no branches: this runs between MFMAs)
    const float *base = left ? p.a0 : p.a1;
    const int ld = left ? p.lda0 : p.lda1;
    const float *src = (r < p.m && kk < p.k) ? base + (size_t)(unsigned)r * (unsigned)ld + kk : p.zeros;
    glds16(src, lds + stage * kStage + pc * 256);
  };
  auto issue_a = [&](int b, int stage) {                     // b: block index inside this workgroup's range
#pragma unroll
    for (int t = 0; t < 8; ++t) issue_piece(b, stage, t);
  };
  // EPI_DX_MASK: the ReLU-sign bytes of block b (48 contiguous rows of mld bytes) -> slot b & 1, in 1 KiB pieces.  Always
  // issued BEFORE the A pieces of block b + 1, so "all but my A pieces of one block" covers them.
  const int nmask = EPI == EPI_DX_MASK ? (kRows * p.mld + 1023) >> 10 : 0;   // <= 5
  auto issue_mask = [&](int b) {
    const long long first = (long long)(blk0 + b) * kRows * p.mld, end = (long long)p.m * p.mld;
    for (int i = wave; i < nmask; i += 4) {
      const long long off = first + i * 1024 + lane * 16;
      const void *src = off < end ? (const void *)(p.maskb + off) : (const void *)p.zeros;
      glds16(reinterpret_cast<const float *>(src), mask_lds + (b & 1) * kMaskSlot + i * 256);
    }
  };

  // ---- prologue: the weight image into registers, the first blocks into the ring --------------------------------------
  if (dma_wave) {
    if (EPI == EPI_DX_MASK && nblk > 0) issue_mask(0);
    if (nblk > 0) issue_a(0, 0);
    if (nblk > 1) issue_a(1, 1);
  }
  // Column tiles of this wave: logical tiles wave, wave + 8, wave + 16, rotated by a per-workgroup offset — every
  // workgroup reads the whole weight image from L2 at launch, and with the same wave -> tile map everywhere all 256 CUs
  // ask for the same lines at the same moment (the prologue took 8.5 us, 3.6 without these loads).
  const int rot = blockIdx.x % nt;
  int tile_of[3];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    const int t = wave + 8 * jt + rot;
    tile_of[jt] = wave + 8 * jt < nt ? (t >= nt ? t - nt : t) : nt;   // (absent tile: the zero rows behind the image's last tile)
  }
  f32x4 bfrag[3][kKS];
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    int br = tile_of[jt] * 16 + l16;
    br = br < p.bt_rows ? br : p.bt_rows - 1;
    const float *brow = p.bt + (size_t)br * p.ldb + q * 4;
#pragma unroll
    for (int s = 0; s < kKS; ++s) {
#ifdef A3VT_DBG_R16_NOB   // timing-only: what the weight loads cost the prologue (results are wrong by design)
      bfrag[jt][s] = f32x4{1.f + s, 2.f + jt, 3.f + lane, 4.f};
#else
      bfrag[jt][s] = *reinterpret_cast<const f32x4 *>(brow + s * 16);
#endif
    }
  }
  // (the B loads of waves 0-3 are younger than their first two blocks' DMA: the first counted wait below covers them
  // only by waiting for everything — one vmcnt(0) at the top of block 0)
  wait_vmcnt<0>();

  u16 *c16 = reinterpret_cast<u16 *>(p.c);
  u16 *z16 = reinterpret_cast<u16 *>(p.c2);
  for (int b = 0; b < nblk; ++b) {
    const int stage = b % kStages;
    R16_STAMP(0, 0);
    if (dma_wave && b > 0) {
      // block b (its A pieces and its sign bytes) landed; the A pieces of block b + 1, issued after them, may be in flight
      // (... and so may this wave's stores of block b-1, issued after them: one counter, retired in issue order)
      if (b + 1 < nblk) wait_younger(mine_a + store_instr);
      else wait_vmcnt<0>();
    }
    R16_STAMP(0, 1);
    __builtin_amdgcn_s_barrier();   // block b visible; everyone is done with block b-1: its ring stage, the output tile, its sign slot
    // sign bytes of block b + 1 (slot (b + 1) & 1 was block b-1's) now; the A pieces of block b + 2 (into block b-1's stage)
    // one per k-step inside the MFMA stream below, where their issue cost hides under the matrix pipe
    if (dma_wave && EPI == EPI_DX_MASK && b + 1 < nblk) issue_mask(b + 1);
    const bool feed = dma_wave && b + 2 < nblk;

    R16_STAMP(0, 2);
    // ---- K phase: 3 row tiles x 10 k-steps x this wave's 2-3 column tiles ------------------------------------------
    const float *sA = lds + stage * kStage + l16 * 16 + qs * 4;
    f32x4 acc[kRT][3];
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) acc[rt][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[kRT];
#pragma unroll
    for (int rt = 0; rt < kRT; ++rt) a[rt] = *reinterpret_cast<const f32x4 *>(sA + (rt * kKS) * 256);
#pragma unroll
    for (int s = 0; s < kKS; ++s) {
      f32x4 an[kRT];                                         // next k-step's fragments: their LDS latency hides under this step's MFMAs
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt)
        an[rt] = s + 1 < kKS ? *reinterpret_cast<const f32x4 *>(sA + (rt * kKS + s + 1) * 256) : a[rt];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt) {
        acc[rt][0] = mfma(a[rt], bfrag[0][s], acc[rt][0]);
        acc[rt][1] = mfma(a[rt], bfrag[1][s], acc[rt][1]);
      }
      if (third) {
#pragma unroll
        for (int rt = 0; rt < kRT; ++rt) acc[rt][2] = mfma(a[rt], bfrag[2][s], acc[rt][2]);
      }
      if (s < 8) {
        __builtin_amdgcn_sched_barrier(0);
        if (feed) issue_piece(b + 2, (b + 2) % kStages, s);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt) a[rt] = an[rt];
    }
    R16_STAMP(0, 3);
    // ---- accumulators -> fp32 tile [48][308] (C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg) ---------------
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      if (wave + 8 * jt >= nt) continue;
      const int col = tile_of[jt] * 16 + l16;
#pragma unroll
      for (int rt = 0; rt < kRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(rt * 16 + q * 4 + r) * kEpLd + col] = acc[rt][jt][r];
    }
    wait_lgkm0();
    R16_STAMP(0, 4);
    __builtin_amdgcn_s_barrier();   // output tile complete
    R16_STAMP(0, 5);
    R16_STAMP(4, 6);

    // ---- store phase: every wave takes 6 rows; lane = 8-column group (38 of 64 lanes for 304 columns), so a lane's
    // column class (aggregated / straddling the cut / pass-through) is fixed and the loop body is straight-line code:
    // one 608-byte row per store instruction.  The instruction count per wave and block is fixed (store_instr below).
    {
      const int c8 = lane, col = lane * 8;
      const bool on = col < p.ldc;                           // all columns < ldc are written (pad columns hold exact zeros)
      const int row_first = (blk0 + b) * kRows + wave * 6;
      const float *e = ep + (wave * 6) * kEpLd + (on ? col : 0);
      if (EPI == EPI_FWD_HIDDEN) {
        const bool to_z = on && col < p.csplit;              // raw Z for the neighbour gather (the straddling group whole)
        const bool to_y = on && col + 7 >= p.csplit;         // ReLU(Z), no bias
        unsigned live = 0;                                   // columns of this group whose sign is recorded
#pragma unroll
        for (int t = 0; t < 8; ++t) live |= ((col + t >= p.csplit && col + t < p.n_store) ? 1u : 0u) << t;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int row = row_first + i;
          const bool rok = row < p.m;
          f32x4 v[2] = {*reinterpret_cast<const f32x4 *>(e + i * kEpLd), *reinterpret_cast<const f32x4 *>(e + i * kEpLd + 4)};
          const f32x4 raw = pack8(v[0], v[1]);
          unsigned bits = 0;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              bits |= (v[h][t] > 0.f ? 1u : 0u) << (4 * h + t);
              v[h][t] = (v[h][t] > 0.f || p.no_relu) ? v[h][t] : 0.f;
            }
          bits &= live;
          if (has_z) {
            if (rok && to_z) *reinterpret_cast<f32x4 *>(z16 + (size_t)row * p.ldc2 + col) = raw;
          }
          if (rok && to_y) *reinterpret_cast<f32x4 *>(c16 + (size_t)row * p.ldc + col) = pack8(v[0], v[1]);
          if (p.maskb) {   // two sign bytes (4 columns each); bytes of groups left of the cut are never read
            if (rok && on)
              *reinterpret_cast<u16 *>(p.maskb + (size_t)row * p.mld + p.moff + 2 * c8) = (u16)((bits & 15u) | ((bits >> 4) << 8));
          }
        }
      } else {  // EPI_DX_MASK: gradient through the ReLU of the producing layer (sign bytes of this block in LDS)
        const uint8_t *ms = reinterpret_cast<const uint8_t *>(mask_lds + (b & 1) * kMaskSlot) + (wave * 6) * p.mld + (on ? 2 * c8 : 0);
        unsigned left = 0;                                   // columns of this group that take the aggregated-channel bytes
#pragma unroll
        for (int t = 0; t < 8; ++t) left |= (col + t < p.csplit ? 1u : 0u) << t;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int row = row_first + i;
          f32x4 v[2] = {*reinterpret_cast<const f32x4 *>(e + i * kEpLd), *reinterpret_cast<const f32x4 *>(e + i * kEpLd + 4)};
          const unsigned a2 = *reinterpret_cast<const u16 *>(ms + i * p.mld);            // bytes of columns col..col+3, col+4..col+7
          const unsigned b2 = *reinterpret_cast<const u16 *>(ms + i * p.mld + p.moff);
          const unsigned ba = (a2 & 15u) | (((a2 >> 8) & 15u) << 4), bb = (b2 & 15u) | (((b2 >> 8) & 15u) << 4);
          const unsigned keep = (ba & left) | (bb & ~left);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t) v[h][t] = ((keep >> (4 * h + t)) & 1u) ? v[h][t] : 0.f;
          if (row < p.m && on) *reinterpret_cast<f32x4 *>(c16 + (size_t)row * p.ldc + col) = pack8(v[0], v[1]);
        }
      }
      wait_lgkm0();   // this wave's reads of the output tile and the sign slot are done before it reaches the next barrier
      R16_STAMP(4, 7);
    }
  }
}

}  // namespace

// The register-resident form takes the hidden layers of the bf16 storage mode: K in (288, 320] bf16, up to 19 column
// tiles, sign rows that fit the LDS slots, 16-byte aligned rows.  Everything else stays on rowgemm_kernel.
bool rowgemm16_ok(const RowGemmArgs &a, int epi) {
  if (a.bf16 != 2 || (epi != EPI_FWD_HIDDEN && epi != EPI_DX_MASK)) return false;
  if (a.k <= 144 || a.k > 160 || a.ldb < 160 || a.n_store > 304 || a.n_store < 32) return false;
  if (a.ldc % 8 != 0 || a.ldc > 304 || a.ldc < a.n_store) return false;
  if (a.lda0 % 4 != 0 || a.lda1 % 4 != 0 || a.ksplit % 4 != 0) return false;
  if (a.m < 48 * 128) return false;                           // small batches: the column-split launches of rowgemm_kernel
  if (epi == EPI_FWD_HIDDEN && (a.c2 == nullptr || a.ldc2 % 8 != 0 || a.csplit > a.ldc2)) return false;
  if (a.maskb != nullptr || epi == EPI_DX_MASK) {
    if (a.mld % 2 != 0 || kRows * a.mld > kMaskSlot * 4 || a.moff + a.ldc / 4 > a.mld) return false;
  }
  return true;
}

#ifdef A3VT_DBG_R16_STAMPS
extern "C" int a3vt_dbg_r16_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_r16_stamps), sizeof(unsigned long long) * 2 * 256 * 16 * 8);
}
#endif

int launch_rowgemm16(const RowGemmArgs &a0, int epi, hipStream_t s) {
  RowGemmArgs a = a0;
  a.bt_rows = rowgemm_bt_rows(a.n_store);
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)rowgemm16_kernel<EPI_FWD_HIDDEN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsFloats * 4);
    (void)hipFuncSetAttribute((const void *)rowgemm16_kernel<EPI_DX_MASK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLdsFloats * 4);
  });
  const int nblk = cdiv(a.m, kRows);
  const int grid = nblk < 256 ? nblk : 256;
  path_count(PATH_RG16);
  if (epi == EPI_FWD_HIDDEN)
    A3VT_LAUNCH((rowgemm16_kernel<EPI_FWD_HIDDEN>), dim3(grid), dim3(512), kLdsFloats * 4, s, a);
  else
    A3VT_LAUNCH((rowgemm16_kernel<EPI_DX_MASK>), dim3(grid), dim3(512), kLdsFloats * 4, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
