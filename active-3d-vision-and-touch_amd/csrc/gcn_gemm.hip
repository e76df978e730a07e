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
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// 16-byte LDS-DMA: each active lane copies 16 B from its own global address to lds_base + lane*16.
__device__ __forceinline__ void glds16(const float *gsrc, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// rowgemm: workgroup = 4 waves; wave w owns rows [w*MT*16, (w+1)*MT*16) of a BM = 64*MT row tile and
// all NT 16-column tiles.  K is walked in chunks of 16: per chunk the A tile [BM][16] and the Bt
// tile [BROWS][16] are DMA'd into LDS (two stages).  A fragment read is one ds_read_b128 per
// (m-tile, chunk): lane l holds A[row l&15][k0 + 4*(l>>4) + t], t = 0..3, and MFMA step t consumes
// element t of both fragments — the k order inside a chunk is permuted identically for A and B,
// which a sum over k does not see.
// ------------------------------------------------------------------------------------------------
template <int MT, int NT, int EPI>
__global__ __launch_bounds__(256, 2) void rowgemm_kernel(RowGemmArgs p) {
  constexpr int BM = 64 * MT;
  constexpr int BROWS = ((NT * 16 + 63) / 64) * 64;
  constexpr int A_INSTR = MT;          // LDS-DMA wave-instructions per wave per chunk for A (16 rows each)
  constexpr int B_INSTR = BROWS / 64;  // same for Bt
  constexpr int STAGE = (BM + BROWS) * 16;  // floats per stage
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR)
  const int row0 = blockIdx.x * BM;
  const int nchunks = (p.k + 15) >> 4;
  const int l16 = lane & 15, q = lane >> 4;

  // Loop-invariant per-lane source rows for the DMA (row within tile = 16*instr + lane/4, piece = lane%4).
  const float *a0row[A_INSTR];
  const float *a1row[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    int r = row0 + (wave * A_INSTR + j) * 16 + (lane >> 2);
    r = r < p.m ? r : p.m - 1;  // ragged last tile: duplicate the last row (never stored)
    a0row[j] = p.a0 + (size_t)r * p.lda0;
    a1row[j] = p.a1 + (size_t)r * p.lda1;
  }
  const float *brow[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j)
    brow[j] = p.bt + (size_t)((wave * B_INSTR + j) * 16 + (lane >> 2)) * p.ldb;
  const int kpiece = (lane & 3) * 4;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int chunk, int buf) {
    float *sA = lds + buf * STAGE;
    float *sB = sA + BM * 16;
    const int kk = chunk * 16 + kpiece;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
      const float *src = kk >= p.k ? p.zeros : (kk < p.ksplit ? a0row[j] + kk : a1row[j] + kk);
      glds16(src, sA + (wave * A_INSTR + j) * 256);
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) glds16(brow[j] + kk, sB + (wave * B_INSTR + j) * 256);
  };

  issue(0, 0);
  for (int t = 0; t < nchunks; ++t) {
    if (t + 1 < nchunks) {
      issue(t + 1, (t + 1) & 1);
      wait_vmcnt<A_INSTR + B_INSTR>();  // chunk t landed; chunk t+1 stays in flight
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const float *sA = lds + (t & 1) * STAGE;
    const float *sB = sA + BM * 16;
    f32x4 af[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      af[i] = *reinterpret_cast<const f32x4 *>(sA + ((wave * MT + i) * 16 + l16) * 16 + q * 4);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const f32x4 bf = *reinterpret_cast<const f32x4 *>(sB + (j * 16 + l16) * 16 + q * 4);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[s], acc[i][j], 0, 0, 0);
      }
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();  // everyone finished reading stage t&1 before it is refilled
  }

  // Epilogue. C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + (wave * MT + i) * 16 + q * 4 + r;
      if (row >= p.m) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 16 + l16;
        if (col >= p.n_store) continue;
        const float v = acc[i][j][r];
        if (EPI == EPI_PLAIN) {
          p.c[(size_t)row * p.ldc + col] = v;
        } else if (EPI == EPI_FWD_HIDDEN) {
          if (col < p.csplit)
            p.c2[(size_t)row * p.ldc2 + col] = v;  // raw Z for the neighbour aggregation
          else
            p.c[(size_t)row * p.ldc + col] = v > 0.f ? v : 0.f;  // un-aggregated channels: ReLU(Z), no bias
        } else {  // EPI_DX_MASK: gradient through the ReLU of the producing layer
          const float y = p.mask[(size_t)row * p.ldmask + col];
          p.c[(size_t)row * p.ldc + col] = y > 0.f ? v : 0.f;
        }
      }
    }
  }
}

template <int NT, int EPI>
static int launch_rowgemm_nt(const RowGemmArgs &a, hipStream_t s) {
  constexpr int MT = 2;
  const int grid = cdiv(a.m, 64 * MT);
  A3VT_LAUNCH((rowgemm_kernel<MT, NT, EPI>), dim3(grid), dim3(256), 0, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <int EPI>
static int launch_rowgemm_epi(const RowGemmArgs &a, hipStream_t s) {
  const int nt = cdiv(a.n_store, 16);
  if (nt <= 1) return launch_rowgemm_nt<1, EPI>(a, s);
  if (nt <= 4) return launch_rowgemm_nt<4, EPI>(a, s);
  if (nt <= 7) return launch_rowgemm_nt<7, EPI>(a, s);
  if (nt <= 13) return launch_rowgemm_nt<13, EPI>(a, s);
  if (nt <= 19) return launch_rowgemm_nt<19, EPI>(a, s);
  set_error("rowgemm: n_out=%d > 304 not supported yet", a.n_store);
  return -1;
}

// Rows of Bt the kernel stages for a given n (must exist, zero padded, in the Bt buffer).
int rowgemm_bt_rows(int n_store) {
  const int nt = cdiv(n_store, 16);
  const int tnt = nt <= 1 ? 1 : nt <= 4 ? 4 : nt <= 7 ? 7 : nt <= 13 ? 13 : 19;
  return ((tnt * 16 + 63) / 64) * 64;
}

int launch_rowgemm(const RowGemmArgs &a, int epi, hipStream_t s) {
  if (a.k % 4 != 0 || a.ksplit % 4 != 0 || a.ldb < pad16(a.k)) {
    set_error("rowgemm: k=%d ksplit=%d ldb=%d violate alignment rules", a.k, a.ksplit, a.ldb);
    return -1;
  }
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

__global__ __launch_bounds__(1024, 1) void dw_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform (SGPR)
  const int l16 = lane & 15, q = lane >> 4;
  const int wi = wave & 3, wo = wave >> 2;  // waves of one SIMD share wo-spread, differ in nothing else

  // 16-row images: floats, DMA wave-instructions (1 KiB each), LDS offsets
  const int xfl = 16 * p.ldx, afl = 16 * p.ldz0, gfl = 16 * p.ldz1;
  const int xin = (xfl + 255) >> 8, ain = (afl + 255) >> 8, gin = (gfl + 255) >> 8;
  const int offA = xin * 256, offG = offA + ain * 256, offD = offG + gin * 256;
  const int stage = offD + 256;  // + one dummy 1 KiB slot for idle DMA slots

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

  // row range of this workgroup in units of 16 rows
  const int units = (p.m + 15) >> 4;
  const int ubase = units / gridDim.x, urem = units % gridDim.x;
  const int u0 = blockIdx.x * ubase + (blockIdx.x < urem ? blockIdx.x : urem);
  const int nu = ubase + (blockIdx.x < urem ? 1 : 0);

  // DMA slot s = wave*3 + j (wave-uniform): [0,xin) -> X, [xin,xin+ain) -> dZa, [..,+gin) -> G, else dummy.
  // Every image is one contiguous run of 16*ld floats per unit, so each lane just bumps a pointer.
  const float *sp[3];   // this lane's source for the next unit (zeros for idle lanes)
  int sstep[3];         // floats to advance per unit (0 for idle lanes)
  int sdst[3];          // LDS float offset of the slot inside a stage (wave-uniform)
  int sf[3], sld[3];    // flat float index of this lane's piece inside the image, and the image's row stride
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s = wave * 3 + j;
    const float *img = nullptr;
    int ld = 0, fl = 0, li = 0;
    sdst[j] = offD;
    if (s < xin) {
      img = p.x; ld = p.ldx; fl = xfl; li = s; sdst[j] = li * 256;
    } else if (s < xin + ain) {
      img = p.z0; ld = p.ldz0; fl = afl; li = s - xin; sdst[j] = offA + li * 256;
    } else if (s < xin + ain + gin) {
      img = p.z1; ld = p.ldz1; fl = gfl; li = s - xin - ain; sdst[j] = offG + li * 256;
    }
    const int f = (li * 64 + lane) * 4;
    const bool valid = img != nullptr && f < fl;
    sp[j] = valid ? img + (size_t)u0 * 16 * ld + f : p.zeros;
    sstep[j] = valid ? 16 * ld : 0;
    sf[j] = valid ? f : 0x7fffffff;
    sld[j] = ld;
  }

  auto issue = [&](int unit, int buf) {
    float *base = lds + buf * stage;
    const int rows_left = p.m - unit * 16;
    if (rows_left >= 16) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        glds16(sp[j], base + sdst[j]);
        sp[j] += sstep[j];
      }
    } else {  // ragged global tail: rows >= m contribute zeros
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        glds16(sf[j] < rows_left * sld[j] ? sp[j] : p.zeros, base + sdst[j]);
        sp[j] += sstep[j];
      }
    }
  };

  f32x4 acc[DW_MAXI][DW_MAXO];
#pragma unroll
  for (int i = 0; i < DW_MAXI; ++i)
#pragma unroll
    for (int j = 0; j < DW_MAXO; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B-operand column of each owned output tile and which image it lives in
  int zoff[DW_MAXO];
#pragma unroll
  for (int j = 0; j < DW_MAXO; ++j) {
    const int col = (o0 + j) * 16 + l16;
    zoff[j] = col < p.zsplit ? offA + col : offG + col;
  }

  if (nu > 0) issue(u0, 0);
  for (int t = 0; t < nu; ++t) {
    if (t + 1 < nu) {
      issue(u0 + t + 1, (t + 1) & 1);
      wait_vmcnt<3>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    const float *sb = lds + (t & 1) * stage;
#pragma unroll 1
    for (int ks = 0; ks < 4; ++ks) {
      const int r = ks * 4 + q;
      const float *xr = sb + r * p.ldx + i0 * 16 + l16;
      const int ra = r * p.ldz0, rg = r * p.ldz1;
      float a[DW_MAXI], b[DW_MAXO];
#pragma unroll
      for (int i = 0; i < DW_MAXI; ++i) a[i] = i < ni ? xr[i * 16] : 0.f;
#pragma unroll
      for (int j = 0; j < DW_MAXO; ++j) b[j] = j < no ? sb[zoff[j] + (zoff[j] < offG ? ra : rg)] : 0.f;
#pragma unroll
      for (int i = 0; i < DW_MAXI; ++i) {
        if (i < ni) {
#pragma unroll
          for (int j = 0; j < DW_MAXO; ++j)
            if (j < no) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();
  }

  // Partial -> slab[blockIdx][k_in][n_out]  (row = input channel, col = output channel)
  float *slab = p.slab + (size_t)blockIdx.x * p.k_in * p.n_out;
#pragma unroll
  for (int i = 0; i < DW_MAXI; ++i) {
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

static int dw_col_groups(int n_out) { return cdiv(cdiv(n_out, 16), 4 * DW_MAXO); }
int dw_num_slabs(int n_out) {
  const int g = dw_col_groups(n_out);
  return 256 / g > 0 ? 256 / g : 1;
}

int launch_dw(const DwArgs &a, hipStream_t s) {
  if (a.ldx % 4 || a.ldz0 % 4 || a.ldz1 % 4 || a.k_in > DW_MAXI * 64 || a.k_in > a.ldx ||
      a.n_out > a.ldz1 || a.zsplit > a.ldz0) {
    set_error("dw: unsupported dims k_in=%d n_out=%d ldx=%d ldz0=%d ldz1=%d", a.k_in, a.n_out, a.ldx, a.ldz0, a.ldz1);
    return -1;
  }
  const int xin = (16 * a.ldx + 255) / 256, ain = (16 * a.ldz0 + 255) / 256, gin = (16 * a.ldz1 + 255) / 256;
  if (xin + ain + gin > 48) {
    set_error("dw: rows too wide (%d + %d + %d floats)", a.ldx, a.ldz0, a.ldz1);
    return -1;
  }
  const size_t shmem = 2 * (size_t)((xin + ain + gin) * 256 + 256) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void *)dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  A3VT_LAUNCH(dw_kernel, dim3(dw_num_slabs(a.n_out), dw_col_groups(a.n_out)), dim3(1024), shmem, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// out[i] = sum_s slab[s][i]   (fixed order -> bitwise reproducible)
__global__ void slab_reduce_kernel(const float *__restrict__ slab, int nslab, size_t stride, size_t n,
                                   float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int s = 0;
  for (; s + 3 < nslab; s += 4) {
    s0 += slab[(size_t)s * stride + i];
    s1 += slab[(size_t)(s + 1) * stride + i];
    s2 += slab[(size_t)(s + 2) * stride + i];
    s3 += slab[(size_t)(s + 3) * stride + i];
  }
  for (; s < nslab; ++s) s0 += slab[(size_t)s * stride + i];
  out[i] = (s0 + s1) + (s2 + s3);
}

int launch_slab_reduce(const float *slab, int nslab, size_t stride, size_t n, float *out, hipStream_t s) {
  A3VT_LAUNCH(slab_reduce_kernel, dim3(cdiv((long long)n, 256)), dim3(256), 0, s, slab, nslab, stride, n, out);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
