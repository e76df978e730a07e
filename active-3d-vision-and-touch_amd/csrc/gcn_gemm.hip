// gcn_gemm.hip — fp32 MFMA kernels for the per-vertex dense products of the GCN stack (gfx950).
//
// Replaces torch.matmul(features, self.weight) at reconstruction/vision/model.py:352 and the two
// products autograd derives from it (dX = dZ W^T, dW = X^T dZ).
//
//   rowgemm_kernel : C[M][N] = A[M][K] * Bt[N][K]^T      (forward Z = X W with Bt = W^T; backward dX with Bt = W)
//   dw_kernel      : partial dW[K][N] = X[rows]^T dZ[rows] per workgroup -> slabs, then slab_reduce
//
// Both use v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD — MI355X_MICROARCH.md
// "Matrix cores"), operands staged HBM/L2 -> LDS with LDS-DMA (global_load_lds_dwordx4), counted
// vmcnt waits and raw s_barrier so the next chunk's DMA stays in flight under the MFMAs.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "gemm_tile.h"

namespace a3vt {


#ifdef A3VT_DBG_RG_STAMPS   // diagnostic build (tools/build_variants.sh stamps): s_memrealtime (100 MHz) at the phase boundaries
__device__ unsigned long long g_rg_stamps[2 * 256 * 64];   // [real time | shader cycles][workgroup][round (<= 8)][8]; a3vt_dbg_rg_stamps
#define RG_STAMP(round, k)                                                                                   \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 256 && blockIdx.y == 0 && (round) < 8) {                            \
      g_rg_stamps[blockIdx.x * 64 + (round) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();                   \
      g_rg_stamps[256 * 64 + blockIdx.x * 64 + (round) * 8 + (k)] = __builtin_amdgcn_s_memtime();            \
    }                                                                                                        \
  } while (0)
#else
#define RG_STAMP(round, k) do { } while (0)
#endif


// ------------------------------------------------------------------------------------------------
// rowgemm: persistent workgroups; the M rows are cut into 16-row tiles that are dealt evenly to the
// workgroups, and inside a workgroup to its waves, two tiles per wave per round (a wave's round =
// 2 x NT accumulator tiles of 16 x 16, 152 VGPRs at NT = 19).  The last, partial round of a workgroup gives
// every wave one tile instead of half the waves two, so all four SIMDs stay equally loaded.
// K is walked in chunks of 16 through an NSTAGE-deep LDS ring filled by LDS-DMA: each wave DMAs its own A rows
// [32][16] plus its share of the Bt chunk [BROWS][16]; chunk t+NSTAGE-1 is issued right after the single
// barrier of iteration t (counted vmcnt, raw s_barrier).
// A fragment read is one ds_read_b128 per (m-tile, chunk): lane l holds A[row l&15][k0 + 4*(l>>4) + t],
// t = 0..3, and MFMA step t consumes element t of both fragments — the k order inside a chunk is permuted
// identically for A and B, which a sum over k does not see.
//
// Geometry (RowGemmCfg) is a template choice: WAVES waves per workgroup, NSTAGE ring stages.  One 8-wave
// workgroup per CU with a 3-stage ring is the fastest measured; the single-tile remainder launches use a
// deeper ring because a handful of MFMAs per chunk cannot cover a DMA round trip.  Known cost: the epilogue is
// bound by the CU's store path (~7-10 B/clk: 770 KB per CU per launch = 30-45 us) and is not overlapped.
// ------------------------------------------------------------------------------------------------
template <int V>
using RgIdx = std::integral_constant<int, V>;

// ADIRECT: the A operand goes from global memory straight into registers — lane (row l16, k-quad q) loads the 16 bytes that
// ARE its MFMA operand for the chunk (one global_load_dwordx4 in place of one LDS-DMA instruction and one ds_read_b128 per
// 16-row tile and chunk; same rows, same 64-byte runs per row, same position in the wave's vmcnt sequence), held in one
// register set per ring stage; only Bt still crosses the LDS.  The chunk loop is then unrolled by NSTAGE so that the sets
// are indexed statically.
template <int NT, int EPI, int NSTAGE, int WAVES, int MODE, bool ADIRECT = false>
__global__ __launch_bounds__(64 * WAVES, 2) void rowgemm_kernel(RowGemmArgs p) {
  constexpr bool BF16 = MODE == 1;   // bf16 operand mode (fp32 storage, operands rounded on the way into the matrix pipe)
  constexpr bool ST16 = MODE == 2;   // bf16 storage mode
  constexpr int MT = 2;
  constexpr int BROWS = ((NT * 16 + 16 * WAVES - 1) / (16 * WAVES)) * (16 * WAVES);  // Bt rows staged per chunk
  constexpr int A_INSTR = MT;                   // DMA wave-instructions per wave per chunk (16 rows each)
  constexpr int B_INSTR = BROWS / (16 * WAVES);
  constexpr int A_FLOATS = WAVES * MT * 16 * 16;
  constexpr int STAGE = A_FLOATS + BROWS * 16;  // floats per stage
  constexpr int PER = A_INSTR + B_INSTR;        // DMA instructions per wave per chunk
  constexpr int DIST = NSTAGE - 1;              // chunks issued ahead of the one being consumed
  constexpr int MSLOT = 1024;                   // floats (4 KiB) of ReLU-sign bytes per wave (EPI_DX_MASK)
  static_assert((DIST - 1) * PER <= 63, "vmcnt immediate");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR)
  const int nchunks = (p.k + 15) >> 4;
  const int l16 = lane & 15, q = lane >> 4;
  // LDS bank swizzle.  A staged row is 16 floats (64 B), so rows r and r+4 share banks, and the 16-lane groups a
  // ds_read_b128 is served in ({0-3,12-15,20-27}, ... MI355X_MICROARCH.md "LDS") pair exactly such rows: a 2-way
  // conflict on every fragment read (SQ_LDS_BANK_CONFLICT = 45 % of the LDS cycles).  Rows with bit 2 set keep
  // their four 16-B k-quads in the order 2,3,0,1: the DMA lane fetches quad (slot ^ 2), the reader of quad q
  // looks in slot (q ^ 2).  Conflict-free, and free: both are per-lane constants.
  const int kpiece = ((lane & 3) ^ ((lane >> 3) & 2)) * 4;
  const int qs = q ^ ((l16 >> 1) & 2);

  // 16-row tiles, dealt evenly to the workgroups
  const int tiles = (p.m + 15) >> 4;
  const int tbase = tiles / gridDim.x, trem = tiles % gridDim.x;
  const int t0 = blockIdx.x * tbase + ((int)blockIdx.x < trem ? blockIdx.x : trem);
  const int t1 = t0 + tbase + ((int)blockIdx.x < trem ? 1 : 0);

  const int col0 = p.col0 + blockIdx.y * (NT * 16);  // column block of this workgroup (remainder launches)
  // The NT 16-row pieces of a Bt chunk are dealt round-robin to the waves (piece j * WAVES + wave): every wave carries
  // B_INSTR or B_INSTR - 1 of them and none is fetched for rows past the last n-tile (a blocked deal of the padded
  // BROWS = 384 rows made 5 of the 24 pieces of every chunk pure waste at NT = 19, all of them on the last two waves).
  const int nbp = (B_INSTR - 1) * WAVES + wave < NT ? B_INSTR : B_INSTR - 1;   // wave-uniform
  const bool full_per = nbp == B_INSTR;
  const float *brow[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    int br = col0 + (j * WAVES + wave) * 16 + (lane >> 2);
    br = br < p.bt_rows ? br : p.bt_rows - 1;
    brow[j] = p.bt + (size_t)br * p.ldb;
  }

  int rnd_ = 0;
  for (int tb = t0; tb < t1; tb += MT * WAVES, ++rnd_) {
    RG_STAMP(rnd_, 0);
    // tiles of this round for this wave: two each when the round is full, an even split otherwise
    const int cnt = t1 - tb < MT * WAVES ? t1 - tb : MT * WAVES;
    const int base = cnt / WAVES, extra = cnt % WAVES;
    const int nm = base + (wave < extra ? 1 : 0);  // 0..2, wave-uniform
    const int first = tb + wave * base + (wave < extra ? wave : extra);
    const bool active = nm > 0;
    const int row0 = first * 16;

    // this wave's A source rows (ragged tail: duplicate the last row, never stored)
    const float *a0row[A_INSTR];
    const float *a1row[A_INSTR];
    const size_t a0mul = p.a0q_nvert > 0 ? (size_t)p.a0q_nvert : 1;   // quad-major a0: k -> k * N floats past the row's base
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      int r = row0 + j * 16 + (ADIRECT ? l16 : (lane >> 2));
      r = r < p.m ? r : p.m - 1;
      a0row[j] = p.a0 + (size_t)r * p.lda0;
      if (p.a0q_nvert > 0) {
        const int bq = r / p.a0q_nvert;
        a0row[j] = p.a0 + ((size_t)bq * p.a0q_quads * p.a0q_nvert + (size_t)(r - bq * p.a0q_nvert)) * 4;
      }
      a1row[j] = p.a1 + (size_t)r * p.lda1;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One DMA wave-instruction of a chunk: pieces [0, A_INSTR) are this wave's A tiles, the rest its share of Bt.
    f32x4 areg[ADIRECT ? NSTAGE : 1][MT];   // ADIRECT: the A fragments of the chunks in flight, one set per ring stage
    auto issue_piece = [&](int chunk, auto bufc, int piece) {
      const int buf = bufc;
      const int kk = chunk * 16 + ((ADIRECT && piece < A_INSTR) ? q * 4 : kpiece);
#ifdef A3VT_DBG_RG_NOA   // timing-only ablations (tools/build_variants.sh rowgemm): results are wrong by design
      if (piece < A_INSTR) return;
#endif
#ifdef A3VT_DBG_RG_NOB
      if (piece >= A_INSTR) return;
#endif
      if (piece < A_INSTR) {
        float *sA = lds + buf * STAGE + wave * (MT * 256);
        const float *src =
            (piece >= nm || kk >= p.k) ? p.zeros : (kk < p.ksplit ? a0row[piece] + (size_t)kk * a0mul : a1row[piece] + kk);
        if constexpr (ADIRECT) {
          (void)sA;
          areg[decltype(bufc)::value][piece] = *reinterpret_cast<const f32x4 *>(src);
        } else {
          glds16(src, sA + piece * 256);
        }
      } else {
        const int j = piece - A_INSTR;
        if (j >= nbp) return;
        float *sB = lds + buf * STAGE + A_FLOATS;
        glds16(brow[j] + kk, sB + (j * WAVES + wave) * 256);
      }
    };
    auto issue = [&](int chunk, auto bufc) {
#pragma unroll
      for (int pc = 0; pc < PER; ++pc) issue_piece(chunk, bufc, pc);
    };

    if (EPI == EPI_DX_MASK) {
      // This round's ReLU-sign bytes (16*nm rows x mld contiguous bytes) ride along with the first chunks: issued
      // before them, so the first counted wait of the K loop also covers them.
      float *ms = lds + NSTAGE * STAGE + wave * MSLOT;
      const uint8_t *src0 = p.maskb + (size_t)row0 * p.mld;
      const int nbytes = 16 * nm * p.mld;
      for (int o = 0; o < 32 * p.mld; o += 1024) {
        const int b = o + lane * 16;
        const void *src = b < nbytes ? (const void *)(src0 + b) : (const void *)p.zeros;
        glds16(reinterpret_cast<const float *>(src), ms + o / 4);
      }
    }
    if constexpr (ADIRECT) {
      if (0 < DIST && 0 < nchunks) issue(0, RgIdx<0>{});
      if (1 < DIST && 1 < nchunks) issue(1, RgIdx<1 < NSTAGE ? 1 : 0>{});
      if (2 < DIST && 2 < nchunks) issue(2, RgIdx<2 < NSTAGE ? 2 : 0>{});
      if (3 < DIST && 3 < nchunks) issue(3, RgIdx<3 < NSTAGE ? 3 : 0>{});
      static_assert(DIST <= 4, "prologue of the register-resident A path");
    } else {
#pragma unroll
      for (int d = 0; d < DIST; ++d)
        if (d < nchunks) issue(d, d);
    }
    auto step = [&](const int t, auto bufc) {
      const int buf = bufc;
      // chunk t landed; up to DIST-1 younger chunks may still be in flight
      const int younger = nchunks - 1 - t < DIST - 1 ? nchunks - 1 - t : DIST - 1;
      if (full_per) {   // this wave issues PER instructions per chunk ...
        if (younger >= 4) wait_vmcnt<(DIST > 4 ? 4 : 0) * PER>();
        else if (younger == 3) wait_vmcnt<(DIST > 3 ? 3 : 0) * PER>();
        else if (younger == 2) wait_vmcnt<(DIST > 2 ? 2 : 0) * PER>();
        else if (younger == 1) wait_vmcnt<(DIST > 1 ? 1 : 0) * PER>();
        else wait_vmcnt<0>();
      } else {          // ... or one fewer (the deal of the Bt pieces above)
        if (younger >= 4) wait_vmcnt<(DIST > 4 ? 4 : 0) * (PER - 1)>();
        else if (younger == 3) wait_vmcnt<(DIST > 3 ? 3 : 0) * (PER - 1)>();
        else if (younger == 2) wait_vmcnt<(DIST > 2 ? 2 : 0) * (PER - 1)>();
        else if (younger == 1) wait_vmcnt<(DIST > 1 ? 1 : 0) * (PER - 1)>();
        else wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();  // chunk t visible to all waves; everyone is done with chunk t-1's stage
      if (t == 0) RG_STAMP(rnd_, 1);
      if (t == 1) RG_STAMP(rnd_, 5);
      // Chunk t+DIST goes to stage (t-1) % NSTAGE, free since the barrier above.  With enough n-tile pairs its DMA
      // instructions are spread through the MFMA stream below so their issue cost hides under the matrix pipe.
      constexpr int NPAIR = (NT + 1) / 2;
#ifdef A3VT_DBG_RG_NOSPREAD   // timing experiment: the whole next chunk is issued right after the barrier
      constexpr bool SPREAD = false;
#else
      constexpr bool SPREAD = NPAIR >= 2 * PER;
#endif
      const bool prefetch = t + DIST < nchunks;
      auto nbuf = [&] {   // stage of chunk t + DIST = the one freed by the barrier above (a constant when buf is)
        if constexpr (ADIRECT) return RgIdx<(decltype(bufc)::value + NSTAGE - 1) % NSTAGE>{};
        else return buf >= 1 ? buf - 1 : NSTAGE - 1;
      }();
      if (prefetch && (!SPREAD || !active)) issue(t + DIST, nbuf);
      if (active) {
        const float *sA = lds + buf * STAGE + wave * (MT * 256);
        const float *sB = lds + buf * STAGE + A_FLOATS;
        f32x4 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if constexpr (ADIRECT) af[i] = areg[decltype(bufc)::value][i];
          else af[i] = *reinterpret_cast<const f32x4 *>(sA + (i * 16 + l16) * 16 + qs * 4);
        }
        s16x4 abf[MT];
        if (BF16) {
#pragma unroll
          for (int i = 0; i < MT; ++i) abf[i] = cvt_bf16x4(af[i]);
        }
        // B fragments two n-tiles at a time, register double-buffered: the ds_reads of pair jp+1 are issued one
        // MFMA group into pair jp, so the LDS latency hides under the matrix pipe.
        const float *sBl = sB + l16 * 16 + qs * 4;
        constexpr int NP = (NT + 1) / 2;
        f32x4 bc0 = *reinterpret_cast<const f32x4 *>(sBl);
        f32x4 bc1 = NT > 1 ? *reinterpret_cast<const f32x4 *>(sBl + 256) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
          f32x4 bn0 = bc0, bn1 = bc1;
          if (ST16) {
            acc[0][2 * jp] = mfma_bf16s(af[0], bc0, acc[0][2 * jp]);
            __builtin_amdgcn_sched_barrier(0);
            if (jp + 1 < NP) {
              bn0 = *reinterpret_cast<const f32x4 *>(sBl + (2 * jp + 2) * 256);
              if (2 * jp + 3 < NT) bn1 = *reinterpret_cast<const f32x4 *>(sBl + (2 * jp + 3) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (2 * jp + 1 < NT) acc[0][2 * jp + 1] = mfma_bf16s(af[0], bc1, acc[0][2 * jp + 1]);
            if (nm > 1) {
              acc[1][2 * jp] = mfma_bf16s(af[1], bc0, acc[1][2 * jp]);
              if (2 * jp + 1 < NT) acc[1][2 * jp + 1] = mfma_bf16s(af[1], bc1, acc[1][2 * jp + 1]);
            }
          } else if (BF16) {
            const s16x4 b0 = cvt_bf16x4(bc0), b1 = cvt_bf16x4(bc1);
            acc[0][2 * jp] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(abf[0], b0, acc[0][2 * jp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (jp + 1 < NP) {
              bn0 = *reinterpret_cast<const f32x4 *>(sBl + (2 * jp + 2) * 256);
              if (2 * jp + 3 < NT) bn1 = *reinterpret_cast<const f32x4 *>(sBl + (2 * jp + 3) * 256);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (2 * jp + 1 < NT)
              acc[0][2 * jp + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(abf[0], b1, acc[0][2 * jp + 1], 0, 0, 0);
            if (nm > 1) {
              acc[1][2 * jp] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(abf[1], b0, acc[1][2 * jp], 0, 0, 0);
              if (2 * jp + 1 < NT)
                acc[1][2 * jp + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(abf[1], b1, acc[1][2 * jp + 1], 0, 0, 0);
            }
          } else {
          // first m-tile: 8 MFMAs over two accumulators (dependent distance 2 = 64 cycles > the 40-cycle latency)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s == 1) {
              __builtin_amdgcn_sched_barrier(0);
              if (jp + 1 < NP) {
                bn0 = *reinterpret_cast<const f32x4 *>(sBl + (2 * jp + 2) * 256);
                if (2 * jp + 3 < NT) bn1 = *reinterpret_cast<const f32x4 *>(sBl + (2 * jp + 3) * 256);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            acc[0][2 * jp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][s], bc0[s], acc[0][2 * jp], 0, 0, 0);
            if (2 * jp + 1 < NT)
              acc[0][2 * jp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][s], bc1[s], acc[0][2 * jp + 1], 0, 0, 0);
          }
          if (nm > 1) {  // wave-uniform: the second m-tile is absent in a partial round
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc[1][2 * jp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][s], bc0[s], acc[1][2 * jp], 0, 0, 0);
              if (2 * jp + 1 < NT)
                acc[1][2 * jp + 1] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][s], bc1[s], acc[1][2 * jp + 1], 0, 0, 0);
            }
          }
          }  // !BF16
          __builtin_amdgcn_sched_barrier(0);
          if (SPREAD && (jp & 1) && (jp >> 1) < PER) {
            if (prefetch) issue_piece(t + DIST, nbuf, jp >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          bc0 = bn0;
          bc1 = bn1;
        }
      }
    };
    if constexpr (ADIRECT) {
      for (int tt = 0; tt < nchunks; tt += NSTAGE) {
        step(tt, RgIdx<0>{});
        if (NSTAGE > 1 && tt + 1 < nchunks) step(tt + 1, RgIdx<1 < NSTAGE ? 1 : 0>{});
        if (NSTAGE > 2 && tt + 2 < nchunks) step(tt + 2, RgIdx<2 < NSTAGE ? 2 : 0>{});
        if (NSTAGE > 3 && tt + 3 < nchunks) step(tt + 3, RgIdx<3 < NSTAGE ? 3 : 0>{});
        if (NSTAGE > 4 && tt + 4 < nchunks) step(tt + 4, RgIdx<4 < NSTAGE ? 4 : 0>{});
        static_assert(NSTAGE <= 5, "chunk loop of the register-resident A path");
      }
    } else {
      int buf = 0;
      for (int t = 0; t < nchunks; ++t) {
        step(t, buf);
        buf = buf == NSTAGE - 1 ? 0 : buf + 1;
      }
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();  // all waves finished reading the ring -> reuse it for the epilogue
    RG_STAMP(rnd_, 2);

    // Epilogue.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg — a lane holds 4 rows
    // of one column, so direct stores would be 64-byte fragments.  Each wave transposes its accumulators through
    // a private slice of the idle ring and moves whole rows with 16-byte accesses; the fused ReLU / raw-Z split
    // (forward) and the ReLU-sign multiply (backward, bytes already in LDS) happen on the way out.
#ifdef A3VT_DBG_RG_NOEPI
    if (active && acc[0][0][0] == 1.2345e-33f) {   // never true in practice: keeps the accumulators alive, skips the epilogue
#else
    if (active) {
#endif
      // The epilogue's per-lane address arithmetic depends on the lane alone, so the compiler hoists it in front of the round
      // loop and spills it across the K loop (the forward instantiation sat at 256 registers with 9 spilled): the reloads
      // then sit between the epilogue's global stores, and a scratch reload waits — on the one vmcnt counter — for every
      // store issued before it (round 4: that, not the instruction mix, made the forward launch slower than dX).  An opaque
      // copy of the lane index keeps the arithmetic here, where the operand registers are free again.  Same values.
      int le = lane;
#ifndef A3VT_DBG_RG_HOISTED_EPI   // variant build (tools/build_variants.sh epi): the round-3 code generation, for A/B timing
      asm volatile("" : "+v"(le));
#endif
      const int l16 = le & 15, q = le >> 4;
      float *ep = lds + wave * ((NSTAGE * STAGE) / WAVES);
      constexpr int G0 = (NT + 1) / 2;  // n-tiles in the first column group (second gets NT - G0)
      static_assert(16 * (G0 * 16 + 4) <= (NSTAGE * STAGE) / WAVES, "epilogue slice too small");
      const bool vec_ok = (p.ldc % 4 == 0) && (EPI != EPI_FWD_HIDDEN || p.ldc2 % 4 == 0);
      uint8_t *mslot = reinterpret_cast<uint8_t *>(lds + NSTAGE * STAGE + wave * MSLOT);
      // EPI_FWD_HIDDEN: the ReLU-sign bytes of this wave's rows are collected in its LDS slot and leave as whole rows
      // with 16-byte stores (the rows of a round are contiguous: 16 nm mld bytes from row0 * mld) instead of one
      // scattered byte store per 4 columns.  Only when this workgroup covers all columns (no column blocks) and the slot
      // holds the rows.
      const bool mask_rows = EPI == EPI_FWD_HIDDEN && p.maskb != nullptr && gridDim.y == 1 && 32 * p.mld <= 4 * MSLOT;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i >= nm) continue;
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
          const int j0 = grp == 0 ? 0 : G0;
          const int tiles_g = grp == 0 ? G0 : NT - G0;
          if (tiles_g == 0) continue;
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
          if (ST16 && EPI != EPI_PLAIN) {
            // bf16 rows out: a le packs 8 consecutive columns into one 16-byte store.  Columns up to the padded row
            // length ldc are written (the pad columns hold exact zeros: their Bt rows are zero).
            const int f8row = ncols / 8, nf8 = 16 * f8row;
            for (int f = le; f < nf8; f += 64) {
              const int rl = f / f8row, c8 = f - rl * f8row;
              const int row = row0 + i * 16 + rl;
              const int col = col0 + j0 * 16 + c8 * 8;
              if (row >= p.m || col >= p.ldc) continue;
              f32x4 v[2] = {*reinterpret_cast<const f32x4 *>(ep + rl * stride + c8 * 8),
                            *reinterpret_cast<const f32x4 *>(ep + rl * stride + c8 * 8 + 4)};
#ifdef A3VT_DBG_RG_NOSTORE   // timing-only: the LDS transposition runs, the global stores do not
              if (v[0][0] != 1.2345e-33f) continue;
#endif
              u16 *yo = reinterpret_cast<u16 *>(p.c) + (size_t)row * p.ldc + col;
              if (EPI == EPI_FWD_HIDDEN) {
                if (p.maskb) {
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    const int g4 = col + 4 * h;
                    if (g4 + 3 < p.csplit || g4 >= ((p.n_store + 3) & ~3)) continue;
                    unsigned bits = 0;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                      bits |= ((g4 + t >= p.csplit && g4 + t < p.n_store && v[h][t] > 0.f) ? 1u : 0u) << t;
                    if (mask_rows) mslot[(i * 16 + rl) * p.mld + p.moff + (g4 >> 2)] = (uint8_t)bits;
                    else p.maskb[(size_t)row * p.mld + p.moff + (g4 >> 2)] = (uint8_t)bits;
                  }
                }
                u16 *zo = reinterpret_cast<u16 *>(p.c2) + (size_t)row * p.ldc2 + col;
                if (col + 7 < p.csplit) {  // aggregated channels: raw Z for the neighbour gather
                  *reinterpret_cast<f32x4 *>(zo) = pack_bf16x8(v[0], v[1]);
                } else if (col >= p.csplit) {  // pass-through channels: ReLU(Z), no bias
#pragma unroll
                  for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[h][t] = (v[h][t] > 0.f || p.no_relu) ? v[h][t] : 0.f;
                  *reinterpret_cast<f32x4 *>(yo) = pack_bf16x8(v[0], v[1]);
                } else {
                  // The group that straddles the cut goes to BOTH outputs whole: the raw copy's columns >= csplit are
                  // never gathered (the aggregation stops at csplit), and the activation's columns < csplit are
                  // overwritten by the aggregation kernel that runs next.  (Element-wise stores here cost 16 divergent
                  // store instructions per 64 groups for 3 of them.)
                  *reinterpret_cast<f32x4 *>(zo) = pack_bf16x8(v[0], v[1]);
#pragma unroll
                  for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[h][t] = (v[h][t] > 0.f || p.no_relu) ? v[h][t] : 0.f;
                  *reinterpret_cast<f32x4 *>(yo) = pack_bf16x8(v[0], v[1]);
                }
              } else {  // EPI_DX_MASK
                const int ur = i * 16 + rl;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const int g4 = col + 4 * h;
                  const unsigned ba = mslot[ur * p.mld + (g4 >> 2)];
                  const unsigned bb = mslot[ur * p.mld + p.moff + (g4 >> 2)];
#pragma unroll
                  for (int t = 0; t < 4; ++t) {
                    const unsigned bit = ((g4 + t < p.csplit ? ba : bb) >> t) & 1u;
                    v[h][t] = bit ? v[h][t] : 0.f;
                  }
                }
                *reinterpret_cast<f32x4 *>(yo) = pack_bf16x8(v[0], v[1]);
              }
            }
            continue;
          }
          // Quad-major side output (RowGemmArgs::zq_nvert; one column block, so col0 == 0): the first zq_quads column
          // quads of this tile leave as 16 consecutive rows x 16 B per quad (le & 15 = row: 256 contiguous bytes of one
          // quad plane per 16 lanes) instead of as pieces of row-major rows.
          const bool zq_mode = EPI != EPI_PLAIN && p.zq_nvert > 0;
          int qlo = 0;
          if (zq_mode && grp == 0) {
            const int nqz = p.zq_quads;
            const int rl = le & 15;
            const int row = row0 + i * 16 + rl;
            const int bq = row / p.zq_nvert;
            float *qbase = p.c2 + ((size_t)bq * nqz * p.zq_nvert + (size_t)(row - bq * p.zq_nvert)) * 4;
            for (int c4 = le >> 4; c4 < nqz; c4 += 4) {
              const f32x4 v = *reinterpret_cast<const f32x4 *>(ep + rl * stride + c4 * 4);
#ifdef A3VT_DBG_RG_NOSTORE
              if (v[0] != 1.2345e-33f) continue;
#endif
              if (row < p.m) *reinterpret_cast<f32x4 *>(qbase + (size_t)c4 * p.zq_nvert * 4) = v;
            }
            qlo = nqz;   // the quad-major columns [0, 4 Q) belong to the aggregation kernels alone (RowGemmArgs::zq_nvert)
            if (EPI == EPI_FWD_HIDDEN && p.yq_quads > nqz) {
              // forward: the pass-through columns [4 Q, 4 yq_quads) — the rest of this column group — are part of the
              // quad-major region of the activations too (RowGemmArgs::yq): ReLU, sign bits, same row-fastest stores
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
#ifdef A3VT_DBG_RG_NOSTORE
                if (v[0] != 1.2345e-33f) continue;
#endif
                *reinterpret_cast<f32x4 *>(ybase + (size_t)c4 * p.zq_nvert * 4) = v;
              }
              qlo = p.yq_quads;
            }
          }
          const int wq = f4row - qlo, nf4 = 16 * wq;
          for (int f = le; f < nf4; f += 64) {
            const int rl = f / wq, c4 = qlo + (f - rl * wq);
            const int row = row0 + i * 16 + rl;
            const int col = col0 + j0 * 16 + c4 * 4;
            if (row >= p.m || col >= p.n_store) continue;
            f32x4 v = *reinterpret_cast<const f32x4 *>(ep + rl * stride + c4 * 4);
#ifdef A3VT_DBG_RG_NOSTORE
            if (v[0] != 1.2345e-33f) continue;
#endif
            const bool full = vec_ok && col + 3 < p.n_store;
            if (EPI == EPI_PLAIN) {
              float *dst = p.c + (size_t)row * p.ldc + col;
              if (p.plain_relu) {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = v[t] > 0.f ? v[t] : 0.f;
              }
              if (full) {
                *reinterpret_cast<f32x4 *>(dst) = v;
              } else {
                for (int t = 0; t < 4; ++t)
                  if (col + t < p.n_store) dst[t] = v[t];
              }
            } else if (EPI == EPI_FWD_HIDDEN) {
              if (p.maskb && col + 3 >= p.csplit) {  // ReLU sign of the pass-through channels, 1 byte per 4 columns
                unsigned bits = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                  bits |= ((col + t >= p.csplit && col + t < p.n_store && v[t] > 0.f) ? 1u : 0u) << t;
                if (mask_rows) mslot[(i * 16 + rl) * p.mld + p.moff + (col >> 2)] = (uint8_t)bits;
                else p.maskb[(size_t)row * p.mld + p.moff + (col >> 2)] = (uint8_t)bits;
              }
              if (!zq_mode && full && col + 3 < p.csplit) {  // aggregated channels: raw Z for the neighbour gather
                *reinterpret_cast<f32x4 *>(p.c2 + (size_t)row * p.ldc2 + col) = v;
              } else if (full && col >= p.csplit) {  // pass-through channels: ReLU(Z), no bias
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (v[t] > 0.f || p.no_relu) ? v[t] : 0.f;
                *reinterpret_cast<f32x4 *>(p.c + (size_t)row * p.ldc + col) = v;
              } else if (full && (zq_mode || col + 4 <= p.ldc2)) {
                // the group that straddles the cut goes to BOTH outputs whole: the raw copy's columns >= csplit are never
                // gathered, the activation's columns < csplit are overwritten by the aggregation kernel that runs next
                // (quad-major mode: the raw copy has left above)
                if (!zq_mode) *reinterpret_cast<f32x4 *>(p.c2 + (size_t)row * p.ldc2 + col) = v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = (v[t] > 0.f || p.no_relu) ? v[t] : 0.f;
                *reinterpret_cast<f32x4 *>(p.c + (size_t)row * p.ldc + col) = v;
              } else {
                for (int t = 0; t < 4; ++t) {
                  if (col + t >= p.n_store) break;
                  if (col + t < p.csplit) {
                    if (!zq_mode) p.c2[(size_t)row * p.ldc2 + col + t] = v[t];
                  } else {
                    p.c[(size_t)row * p.ldc + col + t] = (v[t] > 0.f || p.no_relu) ? v[t] : 0.f;
                  }
                }
              }
            } else {  // EPI_DX_MASK: gradient through the ReLU of the producing layer (sign bytes from LDS)
              const int ur = i * 16 + rl;  // row inside this wave's round
              const unsigned ba = mslot[ur * p.mld + (col >> 2)];
              const unsigned bb = mslot[ur * p.mld + p.moff + (col >> 2)];
              float *dst = p.c + (size_t)row * p.ldc + col;
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const unsigned bit = ((col + t < p.csplit ? ba : bb) >> t) & 1u;
                v[t] = bit ? v[t] : 0.f;
              }
              if (full) {
                *reinterpret_cast<f32x4 *>(dst) = v;
              } else {
                for (int t = 0; t < 4; ++t)
                  if (col + t < p.n_store) dst[t] = v[t];
              }
            }
          }
        }
      }
      if (mask_rows) {  // sign bytes of this wave's 16 nm rows: one contiguous block, 16 bytes per le
        wait_lgkm0();
        __builtin_amdgcn_wave_barrier();
        const int nbytes = 16 * nm * p.mld;
        uint8_t *dstm = p.maskb + (size_t)row0 * p.mld;
        for (int o = le * 16; o < nbytes; o += 1024)
          *reinterpret_cast<f32x4 *>(dstm + o) = *reinterpret_cast<const f32x4 *>(mslot + o);
      }
    }
    RG_STAMP(rnd_, 3);
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();  // epilogue slices are free again before the next round's DMA
    RG_STAMP(rnd_, 4);
  }
  // Leftover rows of the load-balanced split (a few 16-row tiles): one 16 x 16 output tile per wave, dealt across the
  // workgroups, operands straight from global memory (rowtile_unit) — a couple of microseconds at the end of this launch
  // instead of a launch of their own.
  if (p.rem_rows > 0) {
    const int ntl = min(NT, (p.n_store - col0 + 15) >> 4);  // n-tiles of this column block that exist
    const int units = ((p.rem_rows + 15) >> 4) * ntl;
    for (int u = wave * gridDim.x + blockIdx.x; u < units; u += WAVES * gridDim.x)
      rowtile_unit<EPI, MODE>(p, p.rem_row0, p.rem_row0 + p.rem_rows, u / ntl, col0 + (u % ntl) * 16, lane);
  }
}

template <int NT>
struct RowGemmCfg {
  // Measured at M = 163,968, 300 x 300 (r01): one 8-wave workgroup per CU with a 3-stage ring 277-290 us per launch;
  // two 4-wave workgroups per CU with 2-stage rings 297-321 us (twice the Bt staging, shallower prefetch).
  static constexpr int WAVES = 8;
  static constexpr int NSTAGE = NT == 1 ? 5 : 3;
  static constexpr int WG_PER_CU = 1;
  static constexpr int BROWS = ((NT * 16 + 16 * WAVES - 1) / (16 * WAVES)) * (16 * WAVES);
};

#ifdef A3VT_DBG_RG_NSTAGE4   // timing experiment (plain epilogue only: the sign-byte slots do not fit beside a 4-stage ring)
template <int NT, int EPI>
struct RowGemmCfgE : RowGemmCfg<NT> {
  static constexpr int NSTAGE = (EPI == EPI_PLAIN && NT == 19) ? 4 : RowGemmCfg<NT>::NSTAGE;
};
#else
template <int NT, int EPI>
struct RowGemmCfgE : RowGemmCfg<NT> {};
#endif

template <int NT, int EPI>
static int launch_rowgemm_nt(const RowGemmArgs &a, int grid_y, hipStream_t s) {
  using C = RowGemmCfgE<NT, EPI>;
  constexpr size_t shmem = (C::NSTAGE * (size_t)(C::WAVES * 2 * 256 + C::BROWS * 16) +
                            (EPI != EPI_PLAIN ? C::WAVES * 1024 : 0)) * sizeof(float);   // + a sign-byte slot per wave
  static_assert(shmem * C::WG_PER_CU <= 160 * 1024, "LDS budget");
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 0>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void *)rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    (void)hipFuncSetAttribute((const void *)rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 2>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (NT == 19)
      (void)hipFuncSetAttribute((const void *)rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 0, NT == 19>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  });
  const int tiles = cdiv(a.m, 16);
#ifdef A3VT_DBG_ENV   // variant builds only (tools/build_variants.sh env): the shipped library reads no environment variable
  static const int env_wg = getenv("A3VT_RG_MAXWG") ? atoi(getenv("A3VT_RG_MAXWG")) : 0;  // developer override (experiments)
#else
  constexpr int env_wg = 0;
#endif
  const int max_wg = env_wg > 0 ? env_wg : 256 * C::WG_PER_CU;
  // one tile per wave until every CU has a workgroup; beyond that the kernel deals tiles evenly (two per wave per round)
  const int grid = cdiv(tiles, C::WAVES) < max_wg ? cdiv(tiles, C::WAVES) : max_wg;
  if (a.bf16 == 2)
    A3VT_LAUNCH((rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 2>), dim3(grid, grid_y), dim3(64 * C::WAVES), shmem, s, a);
  else if (a.bf16)
    A3VT_LAUNCH((rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 1>), dim3(grid, grid_y), dim3(64 * C::WAVES), shmem, s, a);
  else {
    // the exact fp32 products of the full-width layers take their A operand straight into registers (ADIRECT)
#ifdef A3VT_DBG_RG_ADIRECT_OFF   // variant build (tools/build_variants.sh adirect): A through the LDS ring everywhere
    constexpr bool via_lds = true;
#else
    constexpr bool via_lds = false;
#endif
    if (NT == 19 && !via_lds) path_count(PATH_RG_ADIRECT);
    if (NT == 19 && !via_lds)
      A3VT_LAUNCH((rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 0, NT == 19>), dim3(grid, grid_y), dim3(64 * C::WAVES), shmem, s, a);
    else
      A3VT_LAUNCH((rowgemm_kernel<NT, EPI, C::NSTAGE, C::WAVES, 0>), dim3(grid, grid_y), dim3(64 * C::WAVES), shmem, s, a);
  }
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <int EPI>
static int launch_rowgemm_cols(const RowGemmArgs &a, hipStream_t s) {
  const int nt = cdiv(a.n_store, 16);
  // Few rows (forward-only scoring of a handful of meshes: M = 2-7 k): the row tiles alone occupy a fraction of the
  // CUs, so the output columns are cut into blocks (blockIdx.y) until ~all CUs have a workgroup.
  const int row_wgs = cdiv(cdiv(a.m, 16), 8);
  if (row_wgs < 96 && nt > 4) {
    if (row_wgs * cdiv(nt, 4) >= 96) return launch_rowgemm_nt<4, EPI>(a, cdiv(nt, 4), s);
    return launch_rowgemm_nt<1, EPI>(a, nt, s);
  }
  if (nt <= 1) return launch_rowgemm_nt<1, EPI>(a, 1, s);
  if (nt <= 4) return launch_rowgemm_nt<4, EPI>(a, 1, s);
  if (nt <= 8) return launch_rowgemm_nt<8, EPI>(a, 1, s);
  if (nt <= 13) return launch_rowgemm_nt<13, EPI>(a, 1, s);
  // wider outputs: column blocks of 19 tiles (blockIdx.y), A re-read once per block
  return launch_rowgemm_nt<19, EPI>(a, cdiv(nt, 19), s);
}

// ------------------------------------------------------------------------------------------------
// rowtile: the same product for a handful of rows (the remainder of the load-balanced split below: 128 rows at
// bs 64).  With so little work the ring kernel is pure latency (19 dependent DMA round trips, 12.7 us per launch,
// 111 launches per step), so here every wave owns ONE 16 x 16 output tile and pulls its operands straight from
// global memory into registers — all loads of up to 19 K-chunks in flight at once, no LDS, no barrier — then runs
// the MFMA chain: one round trip instead of nineteen.  Same arithmetic order along K as rowgemm_kernel.
// grid = (column tiles, row tiles / 4), 4 waves per workgroup.
// ------------------------------------------------------------------------------------------------
template <int EPI, int MODE>
__global__ __launch_bounds__(256) void rowtile_kernel(RowGemmArgs p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mt = blockIdx.y * 4 + wave;
  if (mt * 16 >= p.m) return;
  rowtile_unit<EPI, MODE>(p, 0, p.m, mt, p.col0 + blockIdx.x * 16, lane);
}

template <int EPI>
static int launch_rowtile(const RowGemmArgs &a, hipStream_t s) {
  const dim3 grid(cdiv(a.n_store, 16), cdiv(cdiv(a.m, 16), 4));
  if (a.bf16 == 2) A3VT_LAUNCH((rowtile_kernel<EPI, 2>), grid, dim3(256), 0, s, a);
  else if (a.bf16) A3VT_LAUNCH((rowtile_kernel<EPI, 1>), grid, dim3(256), 0, s, a);
  else A3VT_LAUNCH((rowtile_kernel<EPI, 0>), grid, dim3(256), 0, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Load balance.  The chip runs 2048 waves of this kernel at once (256 CUs x 4 SIMDs x 2), each owning 16-row
// tiles in pairs; M = 163,968 rows = 10,248 tiles would leave almost every SIMD idle while a few run a sixth
// tile pair (+20 % time).  When such a small remainder exists, the tiles that fill whole rounds go to the main
// launch and the leftover rows go to rowtile_kernel: one 16 x 16 tile per wave, operands from registers.
template <int EPI>
static int launch_rowgemm_epi(const RowGemmArgs &a0, hipStream_t s) {
  RowGemmArgs a = a0;
  a.bt_rows = rowgemm_bt_rows(a.n_store);
  a.col0 = 0;
  const int tiles = cdiv(a.m, 16), nt = cdiv(a.n_store, 16);
  const int full = tiles / 2048 * 2048, rem = tiles - full;
  if (full == 0 || rem == 0 || rem * nt > 1024 || nt < 2) return launch_rowgemm_cols<EPI>(a, s);
  RowGemmArgs m = a;
  m.m = full * 16;
#ifdef A3VT_DBG_ENV
  static const bool separate = getenv("A3VT_ROWTILE_SEPARATE") != nullptr;  // developer switch: remainder as its own launch
#else
  constexpr bool separate = false;
#endif
  if (!separate) {
    m.rem_row0 = full * 16;
    m.rem_rows = a.m - full * 16;
    return launch_rowgemm_cols<EPI>(m, s);
  }
  if (int rc = launch_rowgemm_cols<EPI>(m, s)) return rc;
  RowGemmArgs r = a;
  const size_t r0 = (size_t)full * 16;
  r.m = a.m - (int)r0;
  r.a0 = a.a0 + r0 * a.lda0;
  r.a1 = a.a1 + r0 * a.lda1;
  r.c = a.c + r0 * a.ldc;
  if (a.c2) r.c2 = a.c2 + r0 * a.ldc2;
  if (a.maskb) r.maskb = a.maskb + r0 * a.mld;
  return launch_rowtile<EPI>(r, s);
}

#ifdef A3VT_DBG_RG_STAMPS
extern "C" int a3vt_dbg_rg_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_rg_stamps), sizeof(unsigned long long) * 2 * 256 * 64);
}
#endif

// Rows of Bt the kernel stages for a given n (must exist, zero padded, in the Bt buffer).
int rowgemm_bt_rows(int n_store) {
  const int nt = cdiv(n_store, 16);
  if (nt > 19) return ((n_store + 127) / 128) * 128;
  const int tnt = nt <= 1 ? 1 : nt <= 4 ? 4 : nt <= 8 ? 8 : nt <= 13 ? 13 : 19;
  return ((tnt * 16 + 127) / 128) * 128;
}

bool rowgemm_quad_major_ok(int m, int n_store, int cpad) {
  const int nt = cdiv(n_store, 16);
  if (nt < 14 || nt > 19 || cpad <= 0 || cpad > 160 || cpad % 4 != 0) return false;   // NT = 19, first column group = 160 columns
  const int tiles = cdiv(m, 16), full = tiles / 2048 * 2048, rem = tiles - full;     // as launch_rowgemm_epi
  const int main_m = (full == 0 || rem == 0 || rem * nt > 1024) ? m : full * 16;
  return cdiv(cdiv(main_m, 16), 8) >= 96;                                             // as launch_rowgemm_cols: no column blocks
}

int launch_rowgemm(const RowGemmArgs &a, int epi, hipStream_t s) {
  if (a.bf16 == 3) return launch_rowgemm3(a, epi, s);   // split-operand mode: its own kernel and checks (gcn_gemm3.hip)
  if (a.a0q_nvert > 0 && (a.bf16 == 2 || a.m % a.a0q_nvert != 0 || a.ksplit != a.a0q_quads * 4)) {
    set_error("rowgemm: quad-major a0 needs ksplit = 4 * quads (ksplit=%d quads=%d) and whole meshes (m=%d)", a.ksplit, a.a0q_quads, a.m);
    return -1;
  }
  if (a.yq_quads > 0 && (a.zq_nvert <= 0 || a.yq == nullptr || epi != EPI_FWD_HIDDEN || a.yq_quads < a.zq_quads || a.yq_quads * 4 > 160 ||
                         a.yq_quads * 4 > a.n_store)) {
    set_error("rowgemm: quad-major activations need the quad-major forward epilogue and at most 160 columns (yq_quads=%d)", a.yq_quads);
    return -1;
  }
  if (a.zq_nvert > 0 && (epi == EPI_PLAIN || a.bf16 == 2 || a.c2 == nullptr || a.m % a.zq_nvert != 0 ||
                         a.zq_quads * 4 != pad4(a.csplit) || !rowgemm_quad_major_ok(a.m, a.n_store, a.zq_quads * 4))) {
    set_error("rowgemm: quad-major output unsupported for m=%d n=%d csplit=%d epi=%d mode=%d", a.m, a.n_store, a.csplit, epi, a.bf16);
    return -1;
  }
  if (epi == EPI_DX_MASK && (a.maskb == nullptr || 32 * a.mld > 4096)) {
    set_error("rowgemm: EPI_DX_MASK needs sign bytes with 32*mld <= 4096 (mld=%d)", a.mld);
    return -1;
  }
  if (a.k % 4 != 0 || a.ksplit % 4 != 0 || a.ldb < pad16(a.k)) {
    set_error("rowgemm: k=%d ksplit=%d ldb=%d violate alignment rules", a.k, a.ksplit, a.ldb);
    return -1;
  }
  if (rowgemmw_ok(a, epi)) return launch_rowgemmw(a, epi, s);
#ifndef A3VT_DBG_ROWGEMM16_OFF   // (variant build, tools/build_variants.sh adirect: the bf16 storage mode on rowgemm_kernel)
  if (rowgemm16_ok(a, epi)) return launch_rowgemm16(a, epi, s);
#endif
  switch (epi) {
    case EPI_PLAIN: return launch_rowgemm_epi<EPI_PLAIN>(a, s);
    case EPI_FWD_HIDDEN: return launch_rowgemm_epi<EPI_FWD_HIDDEN>(a, s);
    case EPI_DX_MASK: return launch_rowgemm_epi<EPI_DX_MASK>(a, s);
  }
  set_error("rowgemm: bad epilogue %d", epi);
  return -1;
}

// ------------------------------------------------------------------------------------------------
// Weight transpose + zero pad: W [k][n] -> Wt [rows][ld] with rows >= n, ld >= k.
// ------------------------------------------------------------------------------------------------
__global__ void transpose_pad_kernel(const float *__restrict__ w, int k, int n, float *__restrict__ wt,
                                     int rows, int ld) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over ld (k index), by over rows (n index)
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int kk = bx + i, nn = by + threadIdx.x;
    tile[i][threadIdx.x] = (kk < k && nn < n) ? w[(size_t)kk * n + nn] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int nn = by + i, kk = bx + threadIdx.x;
    if (nn < rows && kk < ld) wt[(size_t)nn * ld + kk] = tile[threadIdx.x][i];
  }
}

int launch_transpose_pad(const float *w, int k, int n, float *wt, int rows, int ld, hipStream_t s) {
  dim3 grid(cdiv(ld, 32), cdiv(rows, 32));
  A3VT_LAUNCH(transpose_pad_kernel, grid, dim3(32, 8), 0, s, w, k, n, wt, rows, ld);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// All weight images of one stack call in ONE launch (blockIdx.z = layer): transposed + padded (forward operand
// Bt = W^T) or copied + padded (backward operand Bt = W).  Saves 2 x (L-1) five-microsecond launches per stage.
__global__ void weight_images_kernel(WeightImages w) {
  const int l = blockIdx.z;
  const float *src = w.w[l];
  float *dst = w.dst + (size_t)l * w.dst_stride;
  const int k = w.k[l], n = w.n, rows = w.rows[l], ld = w.ld[l];
  if (w.transpose) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over ld (k index), by over rows (n index)
    if (bx >= ld || by >= rows) return;
    for (int i = threadIdx.y; i < 32; i += 8) {
      const int kk = bx + i, nn = by + threadIdx.x;
      tile[i][threadIdx.x] = (kk < k && nn < n) ? src[(size_t)kk * n + nn] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
      const int nn = by + i, kk = bx + threadIdx.x;
      if (nn < rows && kk < ld) dst[(size_t)nn * ld + kk] = tile[threadIdx.x][i];
    }
  } else {
    const int tid = threadIdx.y * 32 + threadIdx.x;
    const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
    for (int idx = blk * 256 + tid; idx < rows * ld; idx += nblk * 256) {
      const int r = idx / ld, c = idx % ld;
      dst[idx] = (r < k && c < n) ? src[(size_t)r * n + c] : 0.f;
    }
  }
}

int launch_weight_images(const WeightImages &w, int max_rows, int max_ld, hipStream_t s) {
  A3VT_LAUNCH(weight_images_kernel, dim3(cdiv(max_ld, 32), cdiv(max_rows, 32), w.count), dim3(32, 8), 0, s, w);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Copy + zero pad (no transpose): W [rows_in][cols_in] -> out [rows][ld].
__global__ void copy_pad_kernel(const float *__restrict__ w, int rows_in, int cols_in, float *__restrict__ out,
                                int rows, int ld) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * ld) return;
  const int r = idx / ld, c = idx % ld;
  out[idx] = (r < rows_in && c < cols_in) ? w[(size_t)r * cols_in + c] : 0.f;
}

int launch_copy_pad(const float *w, int rows_in, int cols_in, float *out, int rows, int ld, hipStream_t s) {
  A3VT_LAUNCH(copy_pad_kernel, dim3(cdiv((long long)rows * ld, 256)), dim3(256), 0, s, w, rows_in, cols_in, out,
                     rows, ld);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// dst[m][0..w) = src[m][c0..c0+w)   (w % 4 == 0, both 16-byte aligned): contiguous column block for dw_kernel
__global__ void copy_cols_kernel(const float *__restrict__ src, int ld_src, int c0, int w, float *__restrict__ dst,
                                 long long m) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w4 = w >> 2;
  if (i >= m * w4) return;
  const long long r = i / w4;
  const int c = (int)(i - r * w4) * 4;
  *reinterpret_cast<f32x4 *>(dst + r * w + c) = *reinterpret_cast<const f32x4 *>(src + r * ld_src + c0 + c);
}

int launch_copy_cols(const float *src, int ld_src, int c0, int w, float *dst, long long m, hipStream_t s) {
  A3VT_LAUNCH(copy_cols_kernel, dim3(cdiv(m * (w >> 2), 256)), dim3(256), 0, s, src, ld_src, c0, w, dst, m);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// dW: persistent 1024-thread workgroups, one per CU; grid = (row groups, output-column groups).
// 16 waves arranged 4 (input-channel groups, = SIMD) x 4 (output-channel groups), each owning up to
// 5 x 3 accumulator tiles of 16 x 16, so a workgroup covers up to 320 input x 192 output channels.
// The workgroup walks its row range 16 rows at a time.  Three row-contiguous images are DMA'd into LDS exactly as they
// lie in memory (no per-lane index arithmetic): X[16][ldx], dZa[16][ldz0] (aggregated channels) and
// G[16][ldz1] (full gradient rows; only columns >= zsplit are consumed).  MFMA operands are read with
// ds_read_b32 straight from those images:
//   A[i][k] = X[row 4*ks + (lane>>4)][in0 + (lane&15)],  B[k][j] = dZ[row 4*ks + (lane>>4)][out0 + (lane&15)].
// Each workgroup writes its partial [k_in][n_out] to a slab; slab_reduce sums slabs in a fixed order
// (deterministic, unlike float atomics — cdna_hip_programming.md Guideline 12).
// ------------------------------------------------------------------------------------------------
constexpr int DW_MAXI = 5;   // input-channel tiles per wave
constexpr int DW_MAXO = 3;   // output-channel tiles per wave

// One 16-row stage: 4 k-steps of up to NI x NO MFMAs.  NI/NO < 0 selects the guarded generic form.
template <int NI, int NO>
__device__ __forceinline__ void dw_stage(const float *__restrict__ sb, int ldx, int ldz0, int ldz1, int offG,
                                         int xoff, const int (&zoff)[DW_MAXO], int q, int ni, int no, bool z0q, int rot,
                                         f32x4 (&acc)[DW_MAXI][DW_MAXO]) {
#pragma unroll 1
  for (int ks = 0; ks < 4; ++ks) {
    const int r = q * 4 + ks;  // see dw_stage_fast
    const float *xr = sb + r * ldx + xoff;
    const int ra = z0q ? ((r + rot) & 15) * 4 : r * ldz0, rg = r * ldz1;   // z0q: blocked dZa window, see dw_kernel
    float a[DW_MAXI], b[DW_MAXO];
#ifdef A3VT_DBG_NOLDSREAD
#pragma unroll
    for (int i = 0; i < DW_MAXI; ++i) a[i] = (float)(r + i);
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) b[j] = (float)(ra + j);
    (void)xr; (void)rg;
#else
    // unconditional loads (tiles a wave does not own read a few floats past its window, inside the stage, and are
    // never multiplied): conditional ones become branches and early lgkmcnt waits, see dw_stage_fast
#pragma unroll
    for (int i = 0; i < DW_MAXI; ++i) a[i] = xr[i * 16];
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) b[j] = sb[zoff[j] + (zoff[j] < offG ? ra : rg)];
#endif
#pragma unroll
    for (int i = 0; i < DW_MAXI; ++i) {
      if (NI >= 0 ? i < NI : i < ni) {
#pragma unroll
        for (int j = 0; j < DW_MAXO; ++j)
          if (NO >= 0 ? j < NO : j < no)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
}

// Common shape (ni in {4,5}, no in {2,3} — e.g. 19 x 10 tiles over 4 x 4 waves): the 4 x 2 core block runs
// unconditionally, the 5th row / 3rd column / corner behind three wave-uniform branches, and the operand
// reads of k-step ks+1 are issued before the MFMAs of k-step ks (register double buffer).
// HYB (quad-major sources, see dw_kernel): the leading blocks of X and (z0q) the dZa window are staged as 16-column
// blocks [4 quads][16 slots][4 floats], row r of quad qd in slot (r + qd) & 15 (rot = this lane's qd); xo = offset of this
// lane's column in the block of its first input tile (< 0: this wave's tiles are columns of the row-major image), xoff =
// the same in the row-major image of the remaining columns (row stride ldx).
template <bool HYB, bool BLK>
__device__ __forceinline__ void dw_stage_fast(const float *__restrict__ sb, int ldx, int ldz0, int ldz1, int offG,
                                              int xoff, const int (&zoff)[DW_MAXO], int q, bool row5, bool col3,
                                              int xo, bool z0q, int rot, f32x4 (&acc)[DW_MAXI][DW_MAXO]) {
  float a[DW_MAXI], b[DW_MAXO], an[DW_MAXI], bn[DW_MAXO];
  auto load = [&](int ks, float (&av)[DW_MAXI], float (&bv)[DW_MAXO]) {
    // k-step ks takes rows {ks, 4+ks, 8+ks, 12+ks} of the stage (lane group q reads row 4q+ks) rather than four
    // adjacent rows: a ds_read_b32 is served 32 lanes at a time (two q groups), and rows 4 apart sit 4*ld floats
    // apart = 16 banks off for every ld = 4 mod 8 (300, 100, 52, 148) -> the two 16-lane runs never share a bank.
    // Adjacent rows (ld mod 32 = 12 or 4) overlapped: 2-way conflicts on every operand read.
    const int r = q * 4 + ks;
    const float *xr = sb + r * ldx + xoff;
    const int rq = ((r + rot) & 15) * 4;   // row term inside a 16-column block (dw_kernel: row r of quad qd at slot (r + qd) & 15)
    const int ra = HYB && z0q ? rq : r * ldz0, rg = r * ldz1;
#ifdef A3VT_DBG_NOLDSREAD
#pragma unroll
    for (int i = 0; i < DW_MAXI; ++i) av[i] = (float)(r + i);
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) bv[j] = (float)(ra + j);
    (void)xr; (void)rg;
#else
    // All five / three operand loads are unconditional: a wave without a 5th row tile or 3rd column tile reads a few
    // floats past its window (still inside the stage) and never uses them.  Selecting the address instead
    // ("i < 4 || row5 ? i : 3") made the compiler reuse the loaded a[3] through a v_mov, i.e. wait for the NEXT
    // k-step's ds_reads (s_waitcnt lgkmcnt(0)) before the current k-step's MFMAs — 19 % of the launch.
    (void)row5; (void)col3;
    if (HYB && BLK) {   // this wave's five input tiles are 1 KiB blocks: tile stride 256 floats, slot row rq
      const float *xb = sb + xo + rq;
#pragma unroll
      for (int i = 0; i < DW_MAXI; ++i) av[i] = xb[i * 256];
    } else {
#pragma unroll
      for (int i = 0; i < DW_MAXI; ++i) av[i] = xr[i * 16];
    }
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) bv[j] = sb[zoff[j] + (zoff[j] < offG ? ra : rg)];
#endif
  };
  load(0, a, b);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < 3) load(ks + 1, an, bn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    if (row5) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[4][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], b[j], acc[4][j], 0, 0, 0);
    }
    if (col3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[2], acc[i][2], 0, 0, 0);
      if (row5) acc[4][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], b[2], acc[4][2], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < DW_MAXI; ++i) a[i] = an[i];
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) b[j] = bn[j];
  }
}

// bf16 operand mode: the four rows a lane owns in a 16-row stage (4q .. 4q+3) are the four k values of the
// 16x16x16 bf16 instruction, so each (input tile, output tile) pair takes ONE MFMA per stage instead of four.
__device__ __forceinline__ void dw_stage_bf16(const float *__restrict__ sb, int ldx, int ldz0, int ldz1, int offG,
                                              int xoff, const int (&zoff)[DW_MAXO], int q, int ni, int no,
                                              f32x4 (&acc)[DW_MAXI][DW_MAXO]) {
  s16x4 a[DW_MAXI], b[DW_MAXO];
#pragma unroll
  for (int i = 0; i < DW_MAXI; ++i) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < ni) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) v[ks] = sb[(q * 4 + ks) * ldx + xoff + i * 16];
    }
    a[i] = cvt_bf16x4(v);
  }
#pragma unroll
  for (int j = 0; j < DW_MAXO; ++j) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (j < no) {
      const int ld = zoff[j] < offG ? ldz0 : ldz1;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) v[ks] = sb[zoff[j] + (q * 4 + ks) * ld];
    }
    b[j] = cvt_bf16x4(v);
  }
#pragma unroll
  for (int i = 0; i < DW_MAXI; ++i) {
    if (i < ni) {
#pragma unroll
      for (int j = 0; j < DW_MAXO; ++j)
        if (j < no) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
}

// Narrow inputs (k_in <= 64: the stack's first layer, 53 features): every wave row owns ONE input tile, so a stage is 1 x 2-3
// tiles per wave — straight-line, operands of k-step ks + 1 read before the MFMAs of k-step ks, three accumulator tiles
// instead of fifteen.  (The guarded, rolled dw_stage<-1,-1> took 135 us for the first layer's 234 MB; inside the general
// instantiation a straight-line 1 x 3 stage spilled 65 registers next to the fifteen tiles: DESIGN.md section 8, round 4.)
template <bool HYB>
__device__ __forceinline__ void dw_stage_narrow(const float *__restrict__ sb, int ldx, int ldz0, int ldz1, int offG, int xoff,
                                                const int (&zoff)[DW_MAXO], int q, bool col3, bool z0q, int rot,
                                                f32x4 (&acc)[1][DW_MAXO]) {
  float a, b[DW_MAXO], an = 0.f, bn[DW_MAXO] = {0.f, 0.f, 0.f};
  auto load = [&](int ks, float &av, float (&bv)[DW_MAXO]) {
    const int r = q * 4 + ks;   // rows {ks, 4 + ks, 8 + ks, 12 + ks}: see dw_stage_fast
    av = sb[r * ldx + xoff];
    const int ra = HYB && z0q ? ((r + rot) & 15) * 4 : r * ldz0, rg = r * ldz1;
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) bv[j] = sb[zoff[j] + (zoff[j] < offG ? ra : rg)];   // unconditional: see dw_stage
  };
  load(0, a, b);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (ks < 3) load(ks + 1, an, bn);
    __builtin_amdgcn_sched_barrier(0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[0], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[1], acc[0][1], 0, 0, 0);
    if (col3) acc[0][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[2], acc[0][2], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) b[j] = bn[j];
  }
}

// Input tiles of X staged as blocks when its first 4 * quads columns are quad-major: up to the end of the wave (wi) that
// holds the last of them — waves own tin / 4 (+1) consecutive tiles each, as in dw_kernel.
__host__ __device__ static inline int dw_blocked_tiles(int k_in, int quads) {
  const int tin = (k_in + 15) >> 4, need = (quads * 4 + 15) >> 4;
  int end = 0;
  for (int wi = 0; wi < 4 && end < need; ++wi) end += tin / 4 + (wi < tin % 4 ? 1 : 0);
  return end < tin ? end : tin;
}

template <bool FAST, bool BF16, bool HYB, bool NARROW = false>
__global__ __launch_bounds__(1024, 1) void dw_kernel(DwArgs p) {
  constexpr int MAXI = NARROW ? 1 : DW_MAXI;   // input tiles a wave can own
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR)
  const int l16 = lane & 15, q = lane >> 4;
  const int wi = wave & 3, wo = wave >> 2;  // waves of one SIMD share wo-spread, differ in nothing else

  // tile ownership
  // balanced split: input tiles over wi; this column group's output tiles over wo
  const int tin = (p.k_in + 15) >> 4, tout = (p.n_out + 15) >> 4;
  const int ni = tin / 4 + (wi < tin % 4 ? 1 : 0);
  const int i0 = wi * (tin / 4) + (wi < tin % 4 ? wi : tin % 4);
  const int gbase = tout / gridDim.y, grem = tout % gridDim.y;
  const int gt0 = blockIdx.y * gbase + ((int)blockIdx.y < grem ? blockIdx.y : grem);  // first tile of the group
  const int gtn = gbase + ((int)blockIdx.y < grem ? 1 : 0);
  const int no = gtn / 4 + (wo < gtn % 4 ? 1 : 0);
  const int o0 = gt0 + wo * (gtn / 4) + (wo < gtn % 4 ? wo : gtn % 4);

  // Column windows.  A column group only consumes ITS output columns of dZ, so it stages just those: the part of them
  // that lies in the aggregated block (z0 = dZa, columns < zsplit) and the part in the pass-through block (z1 = G).
  // (Staging whole dZa / G rows in both groups made the kernel read dZ twice: 682 MB per launch against 460 MB.)
  const int gcol0 = gt0 * 16, gcol1 = min(p.n_out, (gt0 + gtn) * 16);
  const int a0 = min(gcol0, p.zsplit) & ~3, a1 = (min(gcol1, p.zsplit) + 3) & ~3;
  const int g0 = max(gcol0, p.zsplit) & ~3, g1 = min((max(gcol1, p.zsplit) + 3) & ~3, p.ldz1);
  const int wa = max(a1 - a0, 0), wg = max(g1 - g0, 0);

  // 16-row images: floats, DMA wave-instructions (1 KiB each), LDS offsets.
  // Quad-major sources (HYB; DwArgs::xq / z0q_nvert: the outputs of the channel-sliced aggregation kernels) are staged as
  // 16-COLUMN BLOCKS of 1 KiB, one DMA instruction each: lane l fetches 16 bytes of column quad l >> 4 of the block, the
  // 16 lanes of a quad walking its 16 rows — 256 contiguous bytes of that quad's plane, so the instruction touches eight
  // cache lines like a contiguous one (pieces taken in compact row-major order are 64 different lines per instruction:
  // +14-24 us per launch, measured).  In LDS the block is [4 quads][16 slots][4 floats] with row r of quad qd in slot
  // (r + qd) & 15: the ds_read_b32 of a k-step (rows 4 q + ks, 16 columns) then hits 64 different banks.
  // X: blocks for its quad-major columns [0, 4 xq_quads) (a multiple of 16, and whole waves' tiles: launch_dw), a compact
  // row-major image for the other wrm columns.  dZa window: blocks when z0q.  G window: compact row-major image.
  const bool z0q = HYB && p.z0q_nvert > 0;
  const int nbq = HYB && p.xq_nvert > 0 ? p.xq_quads >> 2 : 0;
  const int wrm = p.ldx - nbq * 16;
  const int afl = 16 * wa, gfl = 16 * wg;
  const int xin = nbq + ((16 * wrm + 255) >> 8), ain = z0q ? (wa + 15) >> 4 : (afl + 255) >> 8, gin = (gfl + 255) >> 8;
  const int offA = xin * 256, offG = offA + ain * 256, offD = offG + gin * 256;
  const int stage = offD + 256;  // + one dummy 1 KiB slot for idle DMA slots

  // row range of this workgroup in units of 16 rows
  const int units = (p.m + 15) >> 4;
  const int ubase = units / gridDim.x, urem = units % gridDim.x;
  const int u0 = blockIdx.x * ubase + (blockIdx.x < urem ? blockIdx.x : urem);
  const int nu = ubase + (blockIdx.x < urem ? 1 : 0);

  // DMA slot s = wave*3 + j (wave-uniform): [0,xin) -> X, [xin,xin+ain) -> dZa window, [..,+gin) -> G window, else
  // dummy.  A lane's piece is float4 number f4 of the compact [16][w] image: row f4 / (w/4), column c0 + 4 (f4 % (w/4))
  // of the source; the source advances by 16 rows per unit, so each lane just bumps a pointer.
  // Quad-major sources (DwArgs::xq / z0q_nvert: the outputs of the channel-sliced aggregation kernels): the LDS images
  // stay the same compact row-major [16][w] — only where a lane's 16-byte piece comes from changes.  Piece (row, column
  // quad) of mesh b lies at ((b Q + quad) N + v) * 4; 16 rows further it is 64 floats on, plus (Q - 1) N * 4 when the row
  // has crossed into the next mesh.
  const float *sp[3];   // this lane's source for the next unit (zeros for idle lanes)
  int sstep[3];         // floats to advance per unit (0 for idle lanes)
  int sdst[3];          // LDS float offset of the slot inside a stage (wave-uniform)
  int srow[3];          // image row of this lane's piece (ragged-tail test); huge for idle lanes
  const int qn = p.xq_nvert > 0 ? p.xq_nvert : p.z0q_nvert;   // vertices per mesh (the same for X and dZa)
  int sjq[3];           // that further step (wave-uniform: a slot is X or dZa for the whole wave)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s = wave * 3 + j;
    const float *img = nullptr, *imgq = nullptr;   // row-major source; quad-major source of the columns below 4 nquad
    int ld = 0, w = 4, c0 = 0, li = 0, nquad = 0, cend = 0;
    bool blocked = false;
    sdst[j] = offD;
    if (s < nbq) {                       // X, blocked part
      img = p.x; ld = p.ldx_src; c0 = 0; cend = p.ldx; li = s; sdst[j] = li * 256;
      imgq = p.xq; nquad = p.xq_quads; blocked = true;
    } else if (s < xin) {                // X, row-major image of the columns [16 nbq, ldx)
      img = p.x; ld = p.ldx_src; w = wrm; c0 = nbq * 16; li = s - nbq; sdst[j] = s * 256;
    } else if (s < xin + ain) {          // dZa window
      li = s - xin; sdst[j] = offA + li * 256;
      if (z0q) { imgq = p.z0; nquad = p.z0q_quads; c0 = a0; cend = a0 + wa; blocked = true; }
      else { img = p.z0; ld = p.ldz0; w = wa; c0 = a0; }
    } else if (s < xin + ain + gin) {    // G window
      img = p.z1; ld = p.ldz1; w = wg; c0 = g0; li = s - xin - ain; sdst[j] = offG + li * 256;
    }
    int row, col;
    bool valid;
    if (blocked) {
      const int qd = lane >> 4;
      row = ((lane & 15) - qd) & 15;
      col = c0 + li * 16 + qd * 4;
      valid = col < cend && (img != nullptr || col < nquad * 4);
    } else {
      const int wq = w >> 2, f4 = li * 64 + lane;
      row = f4 / wq;
      col = c0 + (f4 - row * wq) * 4;
      valid = img != nullptr && row < 16;
    }
    const size_t grow = (size_t)u0 * 16 + row;
    sp[j] = valid && img != nullptr ? img + grow * ld + col : p.zeros;
    sstep[j] = valid ? 16 * ld : 0;
    srow[j] = valid ? row : 0x7fffffff;
    sjq[j] = (nquad - 1) * qn * 4;
    if (valid && imgq != nullptr && col < nquad * 4) {
      const int bq = (int)(grow / qn), vq = (int)(grow - (size_t)bq * qn);
      sp[j] = imgq + (((size_t)bq * nquad + (col >> 2)) * qn + vq) * 4;
      sstep[j] = 64;   // marks a quad-major lane (launch_dw: no row-major source has a 4-float row)
    }
  }

  // After the pieces of `unit` are on their way every pointer moves 16 rows on.  A quad-major lane whose row thereby
  // crosses into the next mesh steps (Q - 1) N * 4 floats further; `next_mesh` (scalar) is the first row of the next mesh,
  // so the per-lane test runs only in the one or two units per mesh (of ~160) that touch a boundary.
  int next_mesh = qn > 0 ? (int)((((size_t)u0 * 16) / qn + 1) * qn) : 0x7fffffff;
  auto advance = [&](int unit) {
#pragma unroll
    for (int j = 0; j < 3; ++j) sp[j] += sstep[j];
#ifndef A3VT_DBG_DW_NOWRAP   // timing-only ablation: no mesh-boundary bookkeeping (wrong addresses past the first mesh)
    if (HYB && unit * 16 + 31 >= next_mesh) {   // wave-uniform
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int g = unit * 16 + (srow[j] & 15);
        if (sstep[j] == 64 && g < next_mesh && g + 16 >= next_mesh) sp[j] += sjq[j];
      }
      if (unit * 16 + 16 >= next_mesh) next_mesh += qn;
    }
#endif
  };
  auto issue = [&](int unit, int buf) {
    float *base = lds + buf * stage;
    const int rows_left = p.m - unit * 16;
    if (rows_left >= 16) {
#pragma unroll
      for (int j = 0; j < 3; ++j) glds16(sp[j], base + sdst[j]);
    } else {  // ragged global tail: rows >= m contribute zeros
#pragma unroll
      for (int j = 0; j < 3; ++j) glds16(srow[j] < rows_left ? sp[j] : p.zeros, base + sdst[j]);
    }
    advance(unit);
  };

  f32x4 acc[MAXI][DW_MAXO];
#pragma unroll
  for (int i = 0; i < MAXI; ++i)
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B-operand column of each owned output tile and which image it lives in
  int zoff[DW_MAXO];
#pragma unroll
  for (int j = 0; j < DW_MAXO; ++j) {
    const int col = (o0 + j) * 16 + l16;
    const int ca = col - a0;   // blocked dZa window: block ca / 16, column ca % 16 of its 16-float slot rows
    zoff[j] = col < p.zsplit ? offA + (z0q ? (ca >> 4) * 256 + ((ca & 15) >> 2) * 64 + (ca & 3) : ca) : offG + (col - g0);
  }
  const int xo = i0 < nbq ? i0 * 256 + (l16 >> 2) * 64 + (l16 & 3) : -1;   // HYB: this lane's column in its first input tile's block; < 0: row-major image
  const int rot = l16 >> 2;   // this lane's quad inside a block (a0 and the tiles are multiples of 16 columns)

  // NST-stage ring, one barrier per 16-row stage; unit t+NST-1 is issued right after the barrier of iteration t (into
  // the stage everyone finished reading before that barrier).  NST comes from the launcher: as many stages as the
  // windowed images leave room for in LDS (4 at 300 x 300), at least 3.
  const int xoff = HYB ? nbq * 256 + (i0 - nbq) * 16 + l16 : i0 * 16 + l16;   // in the row-major image (behind the blocks)
  const int nst = p.nstage;
  for (int d = 0; d < nst - 1; ++d)
    if (d < nu) issue(u0 + d, d);
  int buf = 0;
  for (int t = 0; t < nu; ++t) {
    const int younger = min(nu - 1 - t, nst - 2);  // stages issued after unit t that may still be in flight
    if (younger >= 3) wait_vmcnt<9>();
    else if (younger == 2) wait_vmcnt<6>();
    else if (younger == 1) wait_vmcnt<3>();
    else wait_vmcnt<0>();
#ifndef A3VT_DBG_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
#ifndef A3VT_DBG_NODMA
    if (t + nst - 1 < nu) issue(u0 + t + nst - 1, buf >= 1 ? buf - 1 : nst - 1);
#endif
    const float *sb = lds + buf * stage;
    if constexpr (NARROW) {
      dw_stage_narrow<HYB>(sb, p.ldx, wa, wg, offG, xoff, zoff, q, no == 3, z0q, rot, acc);
    } else {
    if (BF16) dw_stage_bf16(sb, p.ldx, wa, wg, offG, xoff, zoff, q, ni, no, acc);
    // every wave runs five row tiles: the one SIMD whose waves own four (19 = 5+5+5+4) would otherwise idle for that
    // fifth of the time anyway, its extra tile reads finite neighbouring data and is dropped at the slab write, and the
    // k-step loop loses two of its three wave-uniform branches
    else if (FAST && HYB) {
      // a wave's five input tiles are all blocks or all columns of the row-major image (launch_dw): two copies of the stage
      // so that every operand read keeps an immediate tile offset
#if defined(A3VT_DBG_DW_ARM)      // timing-only bisection: every wave reads its A operand as from the row-major image
      dw_stage_fast<true, false>(sb, wrm, wa, wg, offG, xoff & 0xfff, zoff, q, true, no == 3, xo, z0q, rot, acc);
#elif defined(A3VT_DBG_DW_BRM)    // timing-only bisection: B operand row term of the compact image
      if (xo >= 0) dw_stage_fast<true, true>(sb, wrm, wa, wg, offG, xoff, zoff, q, true, no == 3, xo, false, rot, acc);
      else dw_stage_fast<true, false>(sb, wrm, wa, wg, offG, xoff, zoff, q, true, no == 3, xo, false, rot, acc);
#else
      if (xo >= 0) dw_stage_fast<true, true>(sb, wrm, wa, wg, offG, xoff, zoff, q, true, no == 3, xo, z0q, rot, acc);
      else dw_stage_fast<true, false>(sb, wrm, wa, wg, offG, xoff, zoff, q, true, no == 3, xo, z0q, rot, acc);
#endif
    } else if (FAST) dw_stage_fast<false, false>(sb, p.ldx, wa, wg, offG, xoff, zoff, q, true, no == 3, xo, z0q, rot, acc);
    else dw_stage<-1, -1>(sb, p.ldx, wa, wg, offG, xoff, zoff, q, ni, no, z0q, rot, acc);
    }
    buf = buf == nst - 1 ? 0 : buf + 1;
  }
  wait_lgkm0();

  // Partial -> slab[blockIdx][k_in][n_out]  (row = input channel, col = output channel)
  float *slab = p.slab + (size_t)blockIdx.x * p.k_in * p.n_out;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (i >= ni) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kin = (i0 + i) * 16 + q * 4 + r;
      if (kin >= p.k_in) continue;
#pragma unroll
      for (int j = 0; j < DW_MAXO; ++j) {
        if (j >= no) continue;
        const int col = (o0 + j) * 16 + l16;
        if (col < p.n_out) slab[(size_t)kin * p.n_out + col] = acc[i][j][r];
      }
    }
  }
}

bool dw_quad_major_ok(int k_in, int quads) { return quads % 4 == 0 && dw_blocked_tiles(k_in, quads) == quads / 4; }

static int dw_col_groups(int n_out) { return cdiv(cdiv(n_out, 16), 4 * DW_MAXO); }
int dw_num_slabs(int n_out) {
  const int g = dw_col_groups(n_out);
  return 256 / g > 0 ? 256 / g : 1;
}

int dw_images(const DwArgs &a) { return a.bf16 != 3 && dww_ok(a) ? dww_images() : dw_num_slabs(a.n_out); }
int dw_slab_capacity(int n_out) { return dw_num_slabs(n_out) > dww_images() ? dw_num_slabs(n_out) : dww_images(); }

int launch_dw(const DwArgs &a0, hipStream_t s) {
  if (a0.bf16 == 3) return launch_dw3(a0, s);   // split-operand mode (gcn_gemm3.hip)
  DwArgs a = a0;
#ifdef A3VT_DBG_DW_NOHYB   // timing-only: the plain kernel on the same buffers (wrong results)
  a.xq = nullptr; a.xq_nvert = a.xq_quads = a.z0q_nvert = a.z0q_quads = 0; a.ldx_src = 0;
  if (a.ldz0 < a.zsplit) a.ldz0 = a.zsplit;
#endif
  if (a.ldx_src == 0) a.ldx_src = a.ldx;
  if ((a.xq_nvert > 0 && (a.xq == nullptr || a.xq_quads * 4 > a.k_in || a.m % a.xq_nvert != 0)) ||
      (a.z0q_nvert > 0 && (a.z0q_quads * 4 != a.zsplit || a.m % a.z0q_nvert != 0)) ||
      (a.xq_nvert > 0 && a.z0q_nvert > 0 && a.xq_nvert != a.z0q_nvert) || a.bf16 == 2 ||
      (a.xq_nvert > 0 && a.xq_nvert < 32) || (a.z0q_nvert > 0 && a.z0q_nvert < 32) ||
      ((a.xq_nvert > 0 || a.z0q_nvert > 0) && (a.ldx_src == 4 || a.ldz1 == 4 || a.ldz0 == 4))) {
    set_error("dw: quad-major operands unsupported (xq %d x %d, z0q %d x %d, m=%d)", a.xq_nvert, a.xq_quads, a.z0q_nvert,
              a.z0q_quads, a.m);
    return -1;
  }
  if (a.ldx % 4 || a.ldz0 % 4 || a.ldz1 % 4 || a.k_in > DW_MAXI * 64 || a.k_in > a.ldx ||
      a.n_out > a.ldz1 || a.zsplit > a.ldz0) {
    set_error("dw: unsupported dims k_in=%d n_out=%d ldx=%d ldz0=%d ldz1=%d", a.k_in, a.n_out, a.ldx, a.ldz0, a.ldz1);
    return -1;
  }
  if (dww_ok(a)) return launch_dww(a, s);
  const bool hyb = a.xq_nvert > 0 || a.z0q_nvert > 0;
  if (hyb) path_count(PATH_DW_HYBRID);
  const int nbq = a.xq_nvert > 0 ? a.xq_quads / 4 : 0, wrm = a.ldx - nbq * 16;   // as dw_kernel
  if (a.xq_nvert > 0 && (a.xq_quads % 4 != 0 || dw_blocked_tiles(a.k_in, a.xq_quads) != nbq)) {
    set_error("dw: quad-major X columns (%d) must end where a wave's input tiles end (k_in=%d)", a.xq_quads * 4, a.k_in);
    return -1;
  }
  const int xin = nbq + (16 * wrm + 255) / 256, ain = (16 * a.ldz0 + 255) / 256, gin = (16 * a.ldz1 + 255) / 256;
  if (xin + ain + gin > 48) {
    set_error("dw: rows too wide (%d + %d + %d floats)", a.ldx, a.ldz0, a.ldz1);
    return -1;
  }
  // Stage size of the largest column group (the kernel stages only each group's own dZ columns), ring as deep as fits.
  const int tout_all = cdiv(a.n_out, 16), ngrp = dw_col_groups(a.n_out);
  int worst = 0;
  for (int gy = 0; gy < ngrp; ++gy) {
    const int gbase = tout_all / ngrp, grem = tout_all % ngrp;
    const int gt0 = gy * gbase + (gy < grem ? gy : grem), gtn = gbase + (gy < grem ? 1 : 0);
    const int c0 = gt0 * 16, c1 = (gt0 + gtn) * 16 < a.n_out ? (gt0 + gtn) * 16 : a.n_out;
    const int a0 = (c0 < a.zsplit ? c0 : a.zsplit) & ~3, a1 = ((c1 < a.zsplit ? c1 : a.zsplit) + 3) & ~3;
    const int g0 = (c0 > a.zsplit ? c0 : a.zsplit) & ~3;
    int g1 = ((c1 > a.zsplit ? c1 : a.zsplit) + 3) & ~3;
    g1 = g1 < a.ldz1 ? g1 : a.ldz1;
    const int wa = a1 > a0 ? a1 - a0 : 0, wg = g1 > g0 ? g1 - g0 : 0;
    const int units = xin + (a.z0q_nvert > 0 ? (wa + 15) / 16 : (16 * wa + 255) / 256) + (16 * wg + 255) / 256 + 1;
    worst = units > worst ? units : worst;
  }
  DwArgs args = a;
  args.nstage = (int)((160 * 1024) / ((size_t)worst * 1024));
  args.nstage = args.nstage > 5 ? 5 : args.nstage;
  if (args.nstage < 3) {
    set_error("dw: rows too wide for a 3-stage ring (%d KB per stage)", worst);
    return -1;
  }
  const size_t shmem = (size_t)args.nstage * worst * 1024;
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)dw_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_kernel<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_kernel<false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)dw_kernel<false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  // fast path: every wave owns 4-5 input tiles and 2-3 output tiles (true for 300 x 300)
  const int tin = cdiv(a.k_in, 16), tout = cdiv(a.n_out, 16), groups = dw_col_groups(a.n_out);
  const bool fast = tin / 4 >= 4 && tin <= 20 && (tout / groups) / 4 >= 2 && cdiv(tout, groups) <= 12;
  if (hyb && (a.bf16 || (a.xq_nvert > 0 && !fast))) {
    set_error("dw: quad-major operands need the fp32 kernels (and quad-major X the 300-wide shape): k_in=%d n_out=%d mode=%d",
              a.k_in, a.n_out, a.bf16);
    return -1;
  }
  const dim3 grid(dw_num_slabs(a.n_out), groups);
  // narrow inputs: at most one input tile per wave row, 2-3 output tiles per wave column; row-major X (the first layer's)
  const bool narrow = !a.bf16 && tin <= 4 && a.xq_nvert == 0 && (tout / groups) / 4 >= 2 && cdiv(tout, groups) <= 12;
  if (narrow && hyb) A3VT_LAUNCH((dw_kernel<false, false, true, true>), grid, dim3(1024), shmem, s, args);
  else if (narrow) A3VT_LAUNCH((dw_kernel<false, false, false, true>), grid, dim3(1024), shmem, s, args);
  else if (a.bf16) A3VT_LAUNCH((dw_kernel<false, true, false>), grid, dim3(1024), shmem, s, args);
  else if (fast && hyb) A3VT_LAUNCH((dw_kernel<true, false, true>), grid, dim3(1024), shmem, s, args);
  else if (fast) A3VT_LAUNCH((dw_kernel<true, false, false>), grid, dim3(1024), shmem, s, args);
  else if (hyb) A3VT_LAUNCH((dw_kernel<false, false, true>), grid, dim3(1024), shmem, s, args);
  else A3VT_LAUNCH((dw_kernel<false, false, false>), grid, dim3(1024), shmem, s, args);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// out[i] = sum_s slab[s * stride + i]   (fixed summation order -> bitwise reproducible).
// Workgroup = 64 consecutive outputs x 16 slab groups (1024 threads): each thread sums its 1/16 of the slabs with
// every load in flight at once (the reduce is latency-bound: 128 slabs x 360 KB per layer), then the sixteen
// partials are combined through LDS in a fixed order.
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float *__restrict__ slab, int nslab, size_t stride,
                                                           size_t n, size_t n_out, float *__restrict__ out, int accumulate) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + lane;
  const int per = (nslab + 15) / 16;
  const int s0 = grp * per, s1 = min(s0 + per, nslab);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int s = s0;
    for (; s + 7 < s1; s += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(s + u) * stride + i];
    }
    for (; s < s1; ++s) a[0] += slab[(size_t)s * stride + i];
  }
  part[grp][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && i < n_out) {  // entries [n, n_out) are written as zeros (dead bias channels)
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][lane];
    t = i < n ? t : 0.f;
    out[i] = accumulate ? out[i] + t : t;   // (accumulate: a gradient written where it lives, second use of shared weights)
  }
}

// Many slabs, few outputs (the bias gradient: 2048 partial rows of 100 floats): the kernel above would run two workgroups
// whose threads each walk 128 slabs.  Here a workgroup owns 16 consecutive outputs and cuts the slabs 64 ways (thread =
// output o, slab group g: slabs g, g + 64, ...), so a thread sums nslab / 64 values with eight loads in flight; the 64
// partials per output are combined through LDS in a fixed order.  9.5 -> 4 us per call, 57 calls per step.
__global__ __launch_bounds__(1024) void slab_reduce_tall_kernel(const float *__restrict__ slab, int nslab, size_t stride,
                                                                size_t n, size_t n_out, float *__restrict__ out, int accumulate) {
  __shared__ float part[64][17];
  const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + o;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int sl = grp;
    for (; sl + 7 * 64 < nslab; sl += 8 * 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(sl + u * 64) * stride + i];
    }
    for (; sl < nslab; sl += 64) a[0] += slab[(size_t)sl * stride + i];
  }
  part[grp][o] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && i < n_out) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 64; ++g) t += part[g][o];
    t = i < n ? t : 0.f;
    out[i] = accumulate ? out[i] + t : t;
  }
}

// one launch for the (few, short) partial rows of many layers: blockIdx.y = layer; same summation order as slab_reduce_kernel
__global__ __launch_bounds__(1024) void slab_reduce_batch_kernel(SlabReduceBatch b) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t i = (size_t)blockIdx.x * 64 + lane;
  const float *__restrict__ slab = b.slab + (size_t)blockIdx.y * b.layer_stride;
  const int per = (b.nslab + 15) / 16;
  const int s0 = grp * per, s1 = min(s0 + per, b.nslab);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < b.n) {
    int s = s0;
    for (; s + 7 < s1; s += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(s + u) * b.stride + i];
    }
    for (; s < s1; ++s) a[0] += slab[(size_t)s * b.stride + i];
  }
  part[grp][lane] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && i < b.n_out) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][lane];
    t = i < b.n ? t : 0.f;
    float *out = b.out[blockIdx.y];
    out[i] = b.accumulate ? out[i] + t : t;
  }
}
// the same for many partial rows per layer (the row-walk backward leaves 2048 of them): slab_reduce_tall_kernel's order
__global__ __launch_bounds__(1024) void slab_reduce_tall_batch_kernel(SlabReduceBatch b) {
  __shared__ float part[64][17];
  const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + o;
  const float *__restrict__ slab = b.slab + (size_t)blockIdx.y * b.layer_stride;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < b.n) {
    int sl = grp;
    for (; sl + 7 * 64 < b.nslab; sl += 8 * 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += slab[(size_t)(sl + u * 64) * b.stride + i];
    }
    for (; sl < b.nslab; sl += 64) a[0] += slab[(size_t)sl * b.stride + i];
  }
  part[grp][o] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (grp == 0 && i < b.n_out) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 64; ++g) t += part[g][o];
    t = i < b.n ? t : 0.f;
    float *out = b.out[blockIdx.y];
    out[i] = b.accumulate ? out[i] + t : t;
  }
}
int launch_slab_reduce_batch(const SlabReduceBatch &b, hipStream_t s) {
  if (b.count <= 0) return 0;
  if (b.nslab >= 512 && b.n_out <= 1024)   // as launch_slab_reduce_za chooses
    A3VT_LAUNCH(slab_reduce_tall_batch_kernel, dim3(cdiv((long long)b.n_out, 16), b.count), dim3(1024), 0, s, b);
  else
    A3VT_LAUNCH(slab_reduce_batch_kernel, dim3(cdiv((long long)b.n_out, 64), b.count), dim3(1024), 0, s, b);
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_slab_reduce_za(const float *slab, int nslab, size_t stride, size_t n, size_t n_out, float *out, int accumulate,
                          hipStream_t s) {
  if (nslab >= 512 && n_out <= 1024)
    A3VT_LAUNCH(slab_reduce_tall_kernel, dim3(cdiv((long long)n_out, 16)), dim3(1024), 0, s, slab, nslab, stride, n, n_out, out,
                accumulate);
  else
    A3VT_LAUNCH(slab_reduce_kernel, dim3(cdiv((long long)n_out, 64)), dim3(1024), 0, s, slab, nslab, stride, n, n_out, out,
                accumulate);
  A3VT_CHECK_LAUNCH();
  return 0;
}
int launch_slab_reduce_z(const float *slab, int nslab, size_t stride, size_t n, size_t n_out, float *out, hipStream_t s) {
  return launch_slab_reduce_za(slab, nslab, stride, n, n_out, out, 0, s);
}
int launch_slab_reduce(const float *slab, int nslab, size_t stride, size_t n, float *out, hipStream_t s) {
  return launch_slab_reduce_za(slab, nslab, stride, n, n, out, 0, s);
}

}  // namespace a3vt
