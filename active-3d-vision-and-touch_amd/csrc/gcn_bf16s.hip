// gcn_bf16s.hip — the GCN stack's kernels for bf16 STORAGE (gemm mode 2, BASELINE configs[3]/[4]: "bf16 + MFMA feature
// MLP"): activations, their gradients, the raw aggregated channels and the weight images are bf16 in HBM; every sum
// (MFMA accumulators, neighbour aggregation, bias / weight gradients) is fp32; master weights, optimizer state, the
// stack's input features and its 3-channel output stay fp32.  In fp32 the hidden-layer products are MFMA-bound
// (AI = 75 flop/B); with bf16 operands the matrix pipe is 16x faster and the same products become HBM-bound
//  — so what matters here is bytes: a launch moves 100 + 100 MB instead of 197 + 197 MB.
//
//   rowgemm (Z = X W, dX = dZ W^T)  : gcn_gemm.hip, MODE 2 of rowgemm_kernel (v_mfma_f32_16x16x32_bf16 on raw LDS rows)
//   dw16_kernel (dW = X^T dZ)        : here — reduction over rows; the [row][channel] LDS images are consumed COLUMN-wise
//                                     through ds_read_b64_tr_b16 (hardware transpose read), no register shuffles
//   csr16_* / thin16_*              : here — the CSR gathers and the 3-channel output layer on bf16 rows (8 channels per
//                                     16-byte lane access, 16-lane groups per vertex)
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16 = unsigned short;

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {  // RNE, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ u16 bf16_of(float v) { return (u16)(pack2_bf16(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float f32_of(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// 8 bf16 (one 16-byte access) <-> 8 floats
struct F8 {
  f32x4 lo, hi;
};
__device__ __forceinline__ F8 unpack8(u32x4 r) {
  F8 o;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    o.lo[2 * t] = __builtin_bit_cast(float, r[t] << 16);
    o.lo[2 * t + 1] = __builtin_bit_cast(float, r[t] & 0xffff0000u);
    o.hi[2 * t] = __builtin_bit_cast(float, r[2 + t] << 16);
    o.hi[2 * t + 1] = __builtin_bit_cast(float, r[2 + t] & 0xffff0000u);
  }
  return o;
}
__device__ __forceinline__ u32x4 pack8(const F8 &v) {
  return u32x4{pack2_bf16(v.lo[0], v.lo[1]), pack2_bf16(v.lo[2], v.lo[3]), pack2_bf16(v.hi[0], v.hi[1]),
               pack2_bf16(v.hi[2], v.hi[3])};
}
__device__ __forceinline__ float f8_get(const F8 &v, int t) { return t < 4 ? v.lo[t] : v.hi[t - 4]; }

__device__ __forceinline__ void glds16b(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------
// fp32 rows -> bf16 rows, zero padded to ld_out (stack input features; ld_out % 8 == 0)
// ------------------------------------------------------------------------------------------------
__global__ void cvt_rows_kernel(const float *__restrict__ in, int ld_in, int n, u16 *__restrict__ out, int ld_out,
                                long long m) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-column group per thread
  const int g8 = ld_out >> 3;
  if (i >= m * g8) return;
  const long long r = i / g8;
  const int c = (int)(i - r * g8) * 8;
  F8 v;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = c + t < n ? in[r * ld_in + c + t] : 0.f;
    if (t < 4) v.lo[t] = x;
    else v.hi[t - 4] = x;
  }
  *reinterpret_cast<u32x4 *>(out + r * ld_out + c) = pack8(v);
}
int launch_cvt_rows(const float *in, int ld_in, int n, void *out, int ld_out, long long m, hipStream_t s) {
  const long long total = m * (ld_out >> 3);
  A3VT_LAUNCH(cvt_rows_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, in, ld_in, n, static_cast<u16 *>(out),
              ld_out, m);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// bf16 weight images of up to kMaxImages layers in one launch (blockIdx.z = layer): dst_l [rows_l][ld_l] bf16 =
// W_l^T (transpose = 1, W_l is [k_l][n]) or W_l (transpose = 0), zero padded.  dst_l = dst + l * dst_stride (floats).
__global__ void weight_images16_kernel(WeightImages w) {
  const int l = blockIdx.z;
  const float *src = w.w[l];
  u16 *dst = reinterpret_cast<u16 *>(w.dst + (size_t)l * w.dst_stride);
  const int k = w.k[l], n = w.n, rows = w.rows[l], ld = w.ld[l];
  const int tid = threadIdx.y * 32 + threadIdx.x;
  const int nblk = gridDim.x * gridDim.y, blk = blockIdx.y * gridDim.x + blockIdx.x;
  for (int idx = blk * 256 + tid; idx < rows * ld; idx += nblk * 256) {
    const int r = idx / ld, c = idx % ld;
    float v = 0.f;
    if (w.transpose) {
      if (c < k && r < n) v = src[(size_t)c * n + r];
    } else {
      if (r < k && c < n) v = src[(size_t)r * n + c];
    }
    dst[idx] = bf16_of(v);
  }
}
int launch_weight_images16(const WeightImages &w, int max_rows, int max_ld, hipStream_t s) {
  A3VT_LAUNCH(weight_images16_kernel, dim3(cdiv(max_ld, 32), cdiv(max_rows, 32), w.count), dim3(32, 8), 0, s, w);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// dW = X^T dZ on bf16 rows.  Same decomposition as dw_kernel (gcn_gemm.hip): persistent 1024-thread workgroups, grid =
// (row slabs, column groups); wave (wi, wo) owns input tiles i0.. (ni <= 5) x output tiles o0.. (no <= 3) of its column
// group; a workgroup's partial [k_in][n_out] goes to its slab, slab_reduce sums the slabs in a fixed order.
// A stage is 32 rows: the X window and this column group's dZa / G windows are DMA'd into compact [32][w] images.  The
// MFMA sums over ROWS, so both operands are needed column-wise: lane 16 g + i of v_mfma_f32_16x16x32_bf16 holds
// A[m = i][k = 8 g + j] = X[row 8 g + j][channel i] — eight rows of one channel.  ds_read_b64_tr_b16 delivers exactly
// that from the row-major image: per 16-lane group a 4-row x 16-column block comes back column-major, so two reads fill
// an operand.  Which eight rows of the stage a k-group takes is free as long as both operands agree (a sum over rows):
// group g takes rows 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3, so that the two groups of a 32-lane half read ADJACENT
// 4-row blocks — with 608-byte rows (24 banks mod 64) rows r and r + 8 would land on the same banks, rows r and r + 4 do not.
// ------------------------------------------------------------------------------------------------
constexpr int DW16_MAXI = 5, DW16_MAXO = 3;

__device__ __forceinline__ bf16x8 tr_operand(const u16 *lds_row0_col, int row_stride_elems) {
  // address of (the lane's row of its first block, the lane's 4-column quad); the second block sits 16 rows further
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds_row0_col));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(lds_row0_col + 16 * row_stride_elems));
  return __builtin_bit_cast(bf16x8, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

__global__ __launch_bounds__(1024, 1) void dw16_kernel(Dw16Args p) {
  extern __shared__ __attribute__((aligned(16))) u16 lds16[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l16 = lane & 15, g = lane >> 4;      // MFMA k-group g: rows 4 g .. 4 g + 3 and 16 + 4 g .. of the stage
  const int bq = l16 >> 2, bp = l16 & 3;         // transposed-read address role: block row bq, column quad bp
  const int wi = wave & 3, wo = wave >> 2;

  const int tin = (p.k_in + 15) >> 4, tout = (p.n_out + 15) >> 4;
  const int ni = tin / 4 + (wi < tin % 4 ? 1 : 0);
  const int i0 = wi * (tin / 4) + (wi < tin % 4 ? wi : tin % 4);
  const int gbase = tout / gridDim.y, grem = tout % gridDim.y;
  const int gt0 = blockIdx.y * gbase + ((int)blockIdx.y < grem ? blockIdx.y : grem);
  const int gtn = gbase + ((int)blockIdx.y < grem ? 1 : 0);
  const int no = gtn / 4 + (wo < gtn % 4 ? 1 : 0);
  const int o0 = gt0 + wo * (gtn / 4) + (wo < gtn % 4 ? wo : gtn % 4);

  // column windows of this group in dZa (columns < zsplit) and G (columns >= zsplit), multiples of 8 elements
  const int gcol0 = gt0 * 16, gcol1 = min((gt0 + gtn) * 16, p.ldz1);
  const int a0 = min(gcol0, p.zsplit) & ~7, a1 = (min(gcol1, p.zsplit) + 7) & ~7;
  const int g0 = max(gcol0, p.zsplit) & ~7, g1 = min((max(gcol1, p.zsplit) + 7) & ~7, p.ldz1);
  const int wa = max(a1 - a0, 0), wg = max(g1 - g0, 0), wx = p.xw;

  // 32-row images: elements, DMA wave-instructions (1 KiB = 512 elements each), LDS element offsets
  const int xin = (32 * wx + 511) >> 9, ain = (32 * wa + 511) >> 9, gin = (32 * wg + 511) >> 9;
  const int offA = xin * 512, offG = offA + ain * 512, offD = offG + gin * 512;
  const int stage = offD + 512;  // + one dummy 1 KiB slot for idle DMA slots

  const int units = (p.m + 31) >> 5;
  const int ubase = units / gridDim.x, urem = units % gridDim.x;
  const int u0 = blockIdx.x * ubase + ((int)blockIdx.x < urem ? blockIdx.x : urem);
  const int nu = ubase + ((int)blockIdx.x < urem ? 1 : 0);

  // DMA slots: s = wave * 2 + j
  const u16 *sp[2];
  long long sstep[2];
  int sdst[2], srow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int s = wave * 2 + j;
    const u16 *img = nullptr;
    int ld = 0, w = 8, c0 = 0, li = 0;
    sdst[j] = offD;
    if (s < xin) {
      img = p.x; ld = p.ldx; w = wx; c0 = p.xc0; li = s; sdst[j] = li * 512;
    } else if (s < xin + ain) {
      img = p.z0; ld = p.ldz0; w = wa; c0 = a0; li = s - xin; sdst[j] = offA + li * 512;
    } else if (s < xin + ain + gin) {
      img = p.z1; ld = p.ldz1; w = wg; c0 = g0; li = s - xin - ain; sdst[j] = offG + li * 512;
    }
    const int w8 = w >> 3, f8 = li * 64 + lane;
    const int row = f8 / w8, c8 = f8 - row * w8;
    const bool valid = img != nullptr && row < 32;
    sp[j] = valid ? img + ((size_t)u0 * 32 + row) * ld + c0 + c8 * 8 : reinterpret_cast<const u16 *>(p.zeros);
    sstep[j] = valid ? 32ll * ld : 0;
    srow[j] = valid ? row : 0x7fffffff;
  }
  auto issue = [&](int unit, int buf) {
    u16 *base = lds16 + buf * stage;
    const int rows_left = p.m - unit * 32;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      glds16b(srow[j] < rows_left || rows_left >= 32 ? (const void *)sp[j] : (const void *)p.zeros, base + sdst[j]);
      sp[j] += sstep[j];
    }
  };

  f32x4 acc[DW16_MAXI][DW16_MAXO];
#pragma unroll
  for (int i = 0; i < DW16_MAXI; ++i)
#pragma unroll
    for (int j = 0; j < DW16_MAXO; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane element offsets (inside a stage) of the transposed reads: block row (4 g + bq), column quad 4 bp of the tile
  int xoff[DW16_MAXI], zoff[DW16_MAXO], zld[DW16_MAXO];
#pragma unroll
  for (int i = 0; i < DW16_MAXI; ++i) {
    const int ch = min((i0 + i) * 16 + 4 * bp, wx - 4);  // tiles past ni are clamped (read, never used)
    xoff[i] = (4 * g + bq) * wx + ch;
  }
#pragma unroll
  for (int j = 0; j < DW16_MAXO; ++j) {
    const int col = (o0 + j) * 16 + 4 * bp;
    const bool in_a = col < p.zsplit;
    const int w = in_a ? wa : wg;
    const int c = in_a ? min(col - a0, max(wa - 4, 0)) : min(max(col - g0, 0), max(wg - 4, 0));
    zld[j] = w;
    zoff[j] = (in_a ? offA : offG) + (4 * g + bq) * w + c;
  }

  const int nst = p.nstage;
  for (int d = 0; d < nst - 1; ++d)
    if (d < nu) issue(u0 + d, d);
  int buf = 0;
  for (int t = 0; t < nu; ++t) {
    const int younger = min(nu - 1 - t, nst - 2);
    if (younger >= 3) wait_vm<6>();
    else if (younger == 2) wait_vm<4>();
    else if (younger == 1) wait_vm<2>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (t + nst - 1 < nu) issue(u0 + t + nst - 1, buf >= 1 ? buf - 1 : nst - 1);
    const u16 *sb = lds16 + buf * stage;
    bf16x8 a[DW16_MAXI], b[DW16_MAXO];
    // every lane issues every read (the transposed read needs EXEC all ones); unused tiles are clamped in-bounds
#pragma unroll
    for (int i = 0; i < DW16_MAXI; ++i) a[i] = tr_operand(sb + xoff[i], wx);
#pragma unroll
    for (int j = 0; j < DW16_MAXO; ++j) b[j] = tr_operand(sb + zoff[j], zld[j]);
#pragma unroll
    for (int i = 0; i < DW16_MAXI; ++i) {
      if (i >= ni) continue;  // wave-uniform
#pragma unroll
      for (int j = 0; j < DW16_MAXO; ++j)
        if (j < no) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    buf = buf == nst - 1 ? 0 : buf + 1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // C/D layout: lane holds column (lane & 15) of rows 4 g .. 4 g + 3 of the tile (row = input channel)
  float *slab = p.slab + (size_t)blockIdx.x * p.k_in * p.n_out;
#pragma unroll
  for (int i = 0; i < DW16_MAXI; ++i) {
    if (i >= ni) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kin = (i0 + i) * 16 + g * 4 + r;
      if (kin >= p.k_in) continue;
#pragma unroll
      for (int j = 0; j < DW16_MAXO; ++j) {
        if (j >= no) continue;
        const int col = (o0 + j) * 16 + l16;
        if (col < p.n_out) slab[(size_t)kin * p.n_out + col] = acc[i][j][r];
      }
    }
  }
}

static int dw16_col_groups(int n_out) { return cdiv(cdiv(n_out, 16), 4 * DW16_MAXO); }
int dw16_num_slabs(int n_out) {
  const int g = dw16_col_groups(n_out);
  return 256 / g > 0 ? 256 / g : 1;
}

int launch_dw16(const Dw16Args &a, hipStream_t s) {
  if (a.ldx % 8 || a.ldz0 % 8 || a.ldz1 % 8 || a.xw % 8 || a.xc0 % 8 || a.zsplit % 8 || a.xw > DW16_MAXI * 64 ||
      a.k_in > a.xw || a.xc0 + a.xw > a.ldx || a.n_out > a.ldz1 || a.zsplit > a.ldz0) {
    set_error("dw16: unsupported dims k_in=%d xw=%d n_out=%d ldx=%d ldz0=%d ldz1=%d zsplit=%d", a.k_in, a.xw, a.n_out,
              a.ldx, a.ldz0, a.ldz1, a.zsplit);
    return -1;
  }
  const int tout_all = cdiv(a.n_out, 16), ngrp = dw16_col_groups(a.n_out);
  const int xin = cdiv(32 * a.xw, 512);
  int worst = 0;
  for (int gy = 0; gy < ngrp; ++gy) {
    const int gbase = tout_all / ngrp, grem = tout_all % ngrp;
    const int gt0 = gy * gbase + (gy < grem ? gy : grem), gtn = gbase + (gy < grem ? 1 : 0);
    const int c0 = gt0 * 16, c1 = (gt0 + gtn) * 16 < a.ldz1 ? (gt0 + gtn) * 16 : a.ldz1;
    const int a0 = (c0 < a.zsplit ? c0 : a.zsplit) & ~7, a1 = ((c1 < a.zsplit ? c1 : a.zsplit) + 7) & ~7;
    const int g0 = (c0 > a.zsplit ? c0 : a.zsplit) & ~7;
    int g1 = ((c1 > a.zsplit ? c1 : a.zsplit) + 7) & ~7;
    g1 = g1 < a.ldz1 ? g1 : a.ldz1;
    const int wa = a1 > a0 ? a1 - a0 : 0, wg = g1 > g0 ? g1 - g0 : 0;
    const int slots = xin + cdiv(32 * wa, 512) + cdiv(32 * wg, 512);
    if (slots > 32) {
      set_error("dw16: rows too wide (%d DMA slots per stage, 32 available)", slots);
      return -1;
    }
    worst = slots + 1 > worst ? slots + 1 : worst;
  }
  Dw16Args args = a;
  args.nstage = (int)((160 * 1024) / ((size_t)worst * 1024));
  args.nstage = args.nstage > 5 ? 5 : args.nstage;
  if (args.nstage < 3) {
    set_error("dw16: rows too wide for a 3-stage ring (%d KB per stage)", worst);
    return -1;
  }
  const size_t shmem = (size_t)args.nstage * worst * 1024;
  static OncePerDevice once;
  once.run([] { (void)hipFuncSetAttribute((const void *)dw16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  A3VT_LAUNCH(dw16_kernel, dim3(dw16_num_slabs(a.n_out), ngrp), dim3(1024), shmem, s, args);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// CSR neighbour aggregation on bf16 rows.  A 16-lane group owns one vertex; lane l handles channels 8 l .. 8 l + 7 (one
// 16-byte load of the neighbour row), so up to 128 aggregated channels per pass; sums in fp32.  Work mapping as the
// fp32 kernels (gcn_csr.hip): mesh b is processed by the workgroups of XCD group b % 8.
// ------------------------------------------------------------------------------------------------
struct XcdWalk16 {
  int xcd, nloc, stride, gps, nmesh;
  long long ngroups;
  __device__ XcdWalk16(int batch, int n_vert, int skip = 0) {   // skip: workgroups in front that do other work (% 8 == 0)
    const int bid = blockIdx.x - skip, grid = gridDim.x - skip;
    xcd = bid & 7;
    nloc = bid >> 3;
    stride = (grid + 7 - xcd) >> 3;
    gps = (n_vert + 15) >> 4;  // 16 vertices per workgroup pass
    nmesh = batch > xcd ? (batch - xcd + 7) >> 3 : 0;
    ngroups = (long long)nmesh * gps;
  }
  __device__ bool locate(long long g, int sub, int n_vert, long long &b, int &v) const {
    b = xcd + 8 * (g / gps);
    v = (int)(g % gps) * 16 + sub;
    return v < n_vert;
  }
};

// weighted sum of neighbour rows for one vertex by a 16-lane group; (col, val) fetched 16 at a time and handed around
__device__ __forceinline__ F8 gather_row16(const u16 *__restrict__ base, long long ld, int ch, bool lane_on, int e0,
                                           int e1, int hl, const int32_t *__restrict__ colidx,
                                           const float *__restrict__ val) {
  F8 acc;
  acc.lo = acc.hi = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int eb = e0; eb < e1; eb += 16) {
    const int n = min(16, e1 - eb);
    const int myc = hl < n ? colidx[eb + hl] : 0;
    const float myw = hl < n ? val[eb + hl] : 0.f;
    int j = 0;
    for (; j + 3 < n; j += 4) {
      int c[4];
      float w[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        c[t] = __shfl(myc, j + t, 16);
        w[t] = __shfl(myw, j + t, 16);
      }
      if (lane_on) {
        u32x4 r[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) r[t] = *reinterpret_cast<const u32x4 *>(base + c[t] * ld + ch);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const F8 v = unpack8(r[t]);
          acc.lo += w[t] * v.lo;
          acc.hi += w[t] * v.hi;
        }
      }
    }
    if (j < n) {  // one to three rows left: requested together as well (gcn_csr.hip gather_row); same order of accumulation
      const bool p[3] = {true, j + 1 < n, j + 2 < n};
      int c[3];
      float w[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        c[t] = __shfl(myc, j + t, 16);
        w[t] = __shfl(myw, j + t, 16);
      }
      if (lane_on) {
        u32x4 r[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) r[t] = p[t] ? *reinterpret_cast<const u32x4 *>(base + c[t] * ld + ch) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if (p[t]) {
            const F8 v = unpack8(r[t]);
            acc.lo += w[t] * v.lo;
            acc.hi += w[t] * v.hi;
          }
        }
      }
    }
  }
  return acc;
}

// forward epilogue of one vertex's 8-channel group: bias (this lane's 8 values, zero past c), ReLU, sign bytes, bf16
// store of the channels < c
__device__ __forceinline__ void csr16_fwd_store(const F8 &acc, int ch, int c, const F8 &bs, int relu,
                                                u16 *__restrict__ yo, uint8_t *__restrict__ mrow) {
  F8 o;
  unsigned bits = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float pre = ch + t < c ? f8_get(acc, t) + f8_get(bs, t) : 0.f;
    const float out = (pre > 0.f || !relu) ? pre : 0.f;
    bits |= (pre > 0.f ? 1u : 0u) << t;
    if (t < 4) o.lo[t] = out;
    else o.hi[t - 4] = out;
  }
  if (ch + 7 < c) {
    *reinterpret_cast<u32x4 *>(yo) = pack8(o);
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (ch + t < c) yo[t] = bf16_of(f8_get(o, t));
  }
  // two sign bytes (4 channels each) in one 2-byte store: ch is a multiple of 8 and the mask rows have even length
  if (mrow) *reinterpret_cast<u16 *>(mrow + (ch >> 2)) = (u16)((bits & 15u) | ((bits >> 4) << 8));
}
__device__ __forceinline__ F8 csr16_load_bias(const float *__restrict__ bias, int ch, int c) {
  F8 b;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float v = ch + t < c ? bias[ch + t] : 0.f;
    if (t < 4) b.lo[t] = v;
    else b.hi[t - 4] = v;
  }
  return b;
}

// Hub rows (see gcn_csr.hip): one workgroup per (mesh, row); its 16 lane groups take a sixteenth of the edge list each.
// Runs in the FIRST kCsr16HubWgs workgroups of the aggregation launch itself (they start first and finish under the
// row walk of the others): as a launch of its own the hub pass cost 16-18 us per layer and direction on the
// vision + touch topology (profiles/r03_config3_bf16s_steady_kernels.txt) for a few thousand short items.
constexpr int kCsr16HubWgs = 256;   // multiple of 8: the row walk's blockIdx -> XCD mapping is unchanged behind them

template <int MODE>  // 0 forward epilogue, 1 backward (A^T gather, channels [c, cpad) pass the own gradient through)
__device__ __forceinline__ void csr16_hub_rows(F8 (*red)[16], const u16 *__restrict__ src, int ld_src,
                                               const float *__restrict__ bias, int c, int cpad,
                                               const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                               const float *__restrict__ val, int n_vert, int batch,
                                               const int32_t *__restrict__ heavy, u16 *__restrict__ dst, int ld_dst,
                                               uint8_t *__restrict__ maskb, int mld, int relu) {
  const int hl = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int count = heavy[0];
  const int width = MODE == 0 ? c : cpad;
  for (long long item = blockIdx.x; item < (long long)count * batch; item += kCsr16HubWgs) {
    const int v = heavy[64 + (int)(item % count)];
    const long long b = item / count, row = b * n_vert + v;
    const u16 *sb = src + b * n_vert * (long long)ld_src;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    const int per = ((e1 - e0 + 15) / 16 + 3) & ~3;
    const int s0 = min(e1, e0 + sub * per), s1 = min(e1, s0 + per);
    for (int ch0 = 0; ch0 < width; ch0 += 128) {
      const int ch = ch0 + hl * 8;
      const bool on = ch < width;
      red[sub][hl] = gather_row16(sb, ld_src, ch, on, s0, s1, hl, colidx, val);
      __syncthreads();
      if (sub == 0 && on) {
        F8 acc = red[0][hl];
#pragma unroll 1   // (fully unrolled, the 15 staged rows held 120 registers and set the whole kernel's allocation)
        for (int r = 1; r < 16; ++r) {
          acc.lo += red[r][hl].lo;
          acc.hi += red[r][hl].hi;
        }
        u16 *o = dst + row * ld_dst + ch;
        if (MODE == 0) {
          csr16_fwd_store(acc, ch, c, csr16_load_bias(bias, ch, c), relu, o, maskb ? maskb + row * mld : nullptr);
        } else {
          const F8 own = unpack8(*reinterpret_cast<const u32x4 *>(sb + (long long)v * ld_src + ch));
          F8 out;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float x = ch + t < c ? f8_get(acc, t) : f8_get(own, t);
            if (t < 4) out.lo[t] = x;
            else out.hi[t - 4] = x;
          }
          *reinterpret_cast<u32x4 *>(o) = pack8(out);
        }
      }
      __syncthreads();
    }
  }
}

template <bool HUB>   // HUB: the first kCsr16HubWgs workgroups take the hub rows (a second instantiation: the hub body's
                      // registers would otherwise cost the plain graphs a quarter of their resident waves)
__global__ __launch_bounds__(256) void csr16_fwd_kernel(const u16 *__restrict__ za, int ldza,
                                                        const float *__restrict__ bias, int c,
                                                        const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ colidx,
                                                        const float *__restrict__ val, int n_vert, int batch,
                                                        u16 *__restrict__ y, int ldy, uint8_t *__restrict__ maskb,
                                                        int mld, int relu, int heavy_thresh,
                                                        const int32_t *__restrict__ heavy) {
  constexpr int hub = HUB ? kCsr16HubWgs : 0;
  if (HUB && (int)blockIdx.x < hub) {
    __shared__ F8 red[16][16];
    csr16_hub_rows<0>(red, za, ldza, bias, c, (c + 7) & ~7, rowptr, colidx, val, n_vert, batch, heavy, y, ldy, maskb, mld,
                      relu);
    return;
  }
  const int hl = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const XcdWalk16 w(batch, n_vert, hub);
  // this lane's bias values stay in registers (c <= 128: one channel pass): loading them per vertex put 8 loads and a
  // full vmcnt wait — which also waits for the previous vertex's stores — behind every gather
  const F8 bs0 = csr16_load_bias(bias, hl * 8, c);
  for (long long g = w.nloc; g < w.ngroups; g += w.stride) {
    long long b;
    int v;
    if (!w.locate(g, sub, n_vert, b, v)) continue;  // uniform per 16-lane group
    const long long row = b * n_vert + v;
    const u16 *zb = za + b * n_vert * (long long)ldza;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    if (e1 - e0 > heavy_thresh) continue;
    for (int ch0 = 0; ch0 < c; ch0 += 128) {
      const int ch = ch0 + hl * 8;
      const bool on = ch < c;
      const F8 acc = gather_row16(zb, ldza, ch, on, e0, e1, hl, colidx, val);
#ifdef A3VT_DBG_CSR_NOSTORE   // timing-only ablation (tools/build_variants.sh csr): every gathered value stays live, nothing is stored
      if (on && acc.lo[0] + acc.lo[1] + acc.lo[2] + acc.lo[3] + acc.hi[0] + acc.hi[1] + acc.hi[2] + acc.hi[3] == 1.2345e-33f)
#else
      if (on)
#endif
        csr16_fwd_store(acc, ch, c, c <= 128 ? bs0 : csr16_load_bias(bias, ch, c), relu, y + row * ldy + ch,
                        maskb ? maskb + row * mld : nullptr);
    }
  }
}

int launch_csr16_fwd(const void *za, int ldza, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                     const float *val, const int32_t *heavy, int n_vert, int batch, void *y, int ldy, uint8_t *maskb,
                     int mld, int relu, hipStream_t s) {
  if (ldza % 8 != 0 || ldza < ((c + 7) & ~7)) {
    set_error("csr16_fwd: ldza=%d must be a multiple of 8 and >= pad8(c=%d)", ldza, c);
    return -1;
  }
  const long long m = (long long)batch * n_vert;
  const int grid = (int)(cdiv(m, 16) < 4096 ? (cdiv(m, 16) + 7) / 8 * 8 : 4096);
  if (heavy)
    A3VT_LAUNCH(csr16_fwd_kernel<true>, dim3(grid + kCsr16HubWgs), dim3(256), 0, s, static_cast<const u16 *>(za), ldza, bias, c,
                rowptr, col, val, n_vert, batch, static_cast<u16 *>(y), ldy, maskb, mld, relu, csr_heavy_degree(), heavy);
  else
    A3VT_LAUNCH(csr16_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, static_cast<const u16 *>(za), ldza, bias, c, rowptr, col,
                val, n_vert, batch, static_cast<u16 *>(y), ldy, maskb, mld, relu, 0x7fffffff, heavy);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Backward: dZa[m][ch] = sum_e valT[e] G[b][colT[e]][ch] (ch < c), = G[m][ch] (c <= ch < cpad); bias-gradient partials.
template <bool HUB>
__global__ __launch_bounds__(256) void csr16_bwd_kernel(const u16 *__restrict__ g, int ldg, int c, int cpad,
                                                        const int32_t *__restrict__ rowptr,
                                                        const int32_t *__restrict__ colidx,
                                                        const float *__restrict__ val, int n_vert, int batch,
                                                        u16 *__restrict__ dza, int lddza,
                                                        float *__restrict__ db_slab, int heavy_thresh,
                                                        const int32_t *__restrict__ heavy) {
  __shared__ float red[16][128];
  constexpr int hub = HUB ? kCsr16HubWgs : 0;
  if (HUB && (int)blockIdx.x < hub) {
    static_assert(sizeof(F8) * 16 * 16 <= sizeof(red), "hub rows reuse the bias-gradient staging");
    csr16_hub_rows<1>(reinterpret_cast<F8(*)[16]>(&red[0][0]), g, ldg, nullptr, c, cpad, rowptr, colidx, val, n_vert, batch,
                      heavy, dza, lddza, nullptr, 0, 0);
    return;
  }
  const int hl = threadIdx.x & 15, sub = threadIdx.x >> 4;
  for (int ch0 = 0; ch0 < cpad; ch0 += 128) {
    const int ch = ch0 + hl * 8;
    const bool on = ch < cpad;
    F8 bsum;
    bsum.lo = bsum.hi = f32x4{0.f, 0.f, 0.f, 0.f};
    const XcdWalk16 w(batch, n_vert, hub);
    for (long long gi = w.nloc; gi < w.ngroups; gi += w.stride) {
      long long b;
      int v;
      if (!w.locate(gi, sub, n_vert, b, v)) continue;
      const long long row = b * n_vert + v;
      const u16 *gb = g + b * n_vert * (long long)ldg;
      F8 own;
      own.lo = own.hi = f32x4{0.f, 0.f, 0.f, 0.f};
      if (on) own = unpack8(*reinterpret_cast<const u32x4 *>(gb + (long long)v * ldg + ch));
      bsum.lo += own.lo;
      bsum.hi += own.hi;
      const int e0 = rowptr[v], e1 = rowptr[v + 1];
      if (e1 - e0 > heavy_thresh) continue;
      const F8 acc = gather_row16(gb, ldg, ch, on, e0, e1, hl, colidx, val);
      if (on) {
        F8 out;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float x = ch + t < c ? f8_get(acc, t) : f8_get(own, t);
          if (t < 4) out.lo[t] = x;
          else out.hi[t - 4] = x;
        }
        *reinterpret_cast<u32x4 *>(dza + row * lddza + ch) = pack8(out);
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) red[sub][hl * 8 + t] = f8_get(bsum, t);
    __syncthreads();
    if (threadIdx.x < 128 && ch0 + threadIdx.x < cpad) {
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sm += red[r][threadIdx.x];
      db_slab[(size_t)(blockIdx.x - hub) * cpad + ch0 + threadIdx.x] = sm;
    }
    __syncthreads();
  }
}

int launch_csr16_bwd(const void *g, int ldg, int c, int cpad, const int32_t *rowptrT, const int32_t *colT,
                     const float *valT, const int32_t *heavyT, int n_vert, int batch, void *dza, int lddza,
                     float *db_slab, hipStream_t s) {
  if (ldg % 8 != 0 || lddza % 8 != 0 || cpad % 8 != 0 || lddza < cpad || ldg < cpad) {
    set_error("csr16_bwd: ldg=%d lddza=%d cpad=%d violate alignment rules", ldg, lddza, cpad);
    return -1;
  }
  const int slabs = csr_bwd_num_slabs(batch, n_vert);
  if (heavyT)
    A3VT_LAUNCH(csr16_bwd_kernel<true>, dim3(slabs + kCsr16HubWgs), dim3(256), 0, s, static_cast<const u16 *>(g), ldg, c, cpad,
                rowptrT, colT, valT, n_vert, batch, static_cast<u16 *>(dza), lddza, db_slab, csr_heavy_degree(), heavyT);
  else
    A3VT_LAUNCH(csr16_bwd_kernel<false>, dim3(slabs), dim3(256), 0, s, static_cast<const u16 *>(g), ldg, c, cpad, rowptrT, colT,
                valT, n_vert, batch, static_cast<u16 *>(dza), lddza, db_slab, 0x7fffffff, heavyT);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Tiled aggregation on bf16 rows (round 6): the same sums as csr16_fwd / csr16_bwd for graphs of bounded degree, with the
// neighbour rows read from LDS instead of L2.  The row walk gathers every neighbour's 208-byte row from L2 — seven times
// the array per launch (231 MB at 164 k rows), which is what bounds it (35 us).  The templates number their vertices along a
// surface-filling curve, so a tile of 96 consecutive vertices needs its own rows plus a HALO of 50-95 others: a workgroup
// brings tile + halo into LDS once (LDS-DMA, 1.6x the array instead of 7x, mostly L2 hits on the neighbouring tiles'
// rows), and every gather is a ds_read_b128.  The row-major layout of both arrays is unchanged: rowgemm16 / dw16 / thin16
// read the results in place.
//
// A per-adjacency PLAN, built on the device at the start of a stack call (csr16t_build_kernel), holds per tile the halo's
// vertex list and per vertex eight 16-bit LDS positions (row x 13 pieces) + eight weights (unused slots: the zero row, weight 0).  A tile
// whose halo or degrees exceed the plan's bounds is marked and walks its rows from L2 as before (same kernel, same sums).
// Sums are formed in CSR order with the row walk's own expression: results are bit-identical to csr16_fwd / csr16_bwd.
// ------------------------------------------------------------------------------------------------
constexpr int kT16Tile = 64, kT16Halo = 79, kT16Slots = 8, kT16Pieces = 13;
constexpr int kT16Rows = kT16Tile + kT16Halo + 1;      // LDS rows; the last one is zeros
constexpr int kT16Zero = kT16Rows - 1;
constexpr int kT16Hdr = 1 + kT16Halo;                  // ints per tile: halo count (-1: row walk), halo vertices
constexpr int kT16MaxVerts = 16384;                    // bit mask of the builder
#ifndef A3VT_T16_BUFS
#define A3VT_T16_BUFS 1
#endif
constexpr int kT16Bufs = A3VT_T16_BUFS;                // staging buffers per workgroup: 2 = the next unit's rows arrive during the gathers
constexpr int kT16MaxUnits = kT16Bufs == 2 ? 32 : 12;  // units per persistent workgroup (their plan headers sit in LDS)
constexpr int kT16Wgs = kT16Bufs == 2 ? 512 : 1024;    // persistent workgroups of a launch: two (75 KB of LDS each) or four (37 KB) per CU

__host__ __device__ static inline int t16_tiles(int n_vert) { return (n_vert + kT16Tile - 1) / kT16Tile; }
size_t csr16t_plan_ints(int n_vert) { return (size_t)t16_tiles(n_vert) * kT16Hdr + (size_t)n_vert * (4 + 8) + 16; }
bool csr16t_ok(int n_vert, int c, int max_degree, long long m) {
  return n_vert >= kT16Tile && n_vert <= kT16MaxVerts && ((c + 7) >> 3) == kT16Pieces && max_degree > 0 &&
         max_degree <= kT16Slots && m >= 6144 && (m / n_vert + 7) / 8 * 8 * (long long)t16_tiles(n_vert) <= (long long)kT16MaxUnits * kT16Wgs;
}
struct T16Plan {
  const int32_t *hdr;     // [tiles][kT16Hdr]
  const u16 *slot;        // [n_vert][8]
  const float *wgt;       // [n_vert][8]
};
static inline T16Plan t16_plan(const int32_t *plan, int n_vert) {
  T16Plan p;
  p.hdr = plan;
  p.slot = reinterpret_cast<const u16 *>(plan + (size_t)t16_tiles(n_vert) * kT16Hdr);
  p.wgt = reinterpret_cast<const float *>(plan + (size_t)t16_tiles(n_vert) * kT16Hdr + (size_t)n_vert * 4);
  return p;
}

// one workgroup per tile: halo = the set of out-of-tile neighbours in increasing vertex order (a bit mask in LDS, ranks by
// popcount prefix), slots = tile-relative row or kT16Tile + rank, in the CSR order of the row's entries
__global__ __launch_bounds__(256) void csr16t_build_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const float *__restrict__ val, int n_vert, int32_t *plan) {
  __shared__ unsigned words[kT16MaxVerts / 32];
  __shared__ int prefix[kT16MaxVerts / 32 + 1];
  __shared__ int bad;
  const int t = threadIdx.x, k = blockIdx.x;
  const int t0 = k * kT16Tile, t1 = min(n_vert, t0 + kT16Tile);
  const int nwords = (n_vert + 31) >> 5;
  int32_t *hdr = plan + (size_t)k * kT16Hdr;
  u16 *slot = reinterpret_cast<u16 *>(plan + (size_t)t16_tiles(n_vert) * kT16Hdr);
  float *wgt = reinterpret_cast<float *>(plan + (size_t)t16_tiles(n_vert) * kT16Hdr + (size_t)n_vert * 4);
  for (int i = t; i < nwords; i += 256) words[i] = 0u;
  if (t == 0) bad = 0;
  __syncthreads();
  const int e0 = rowptr[t0], e1 = rowptr[t1];
  for (int e = e0 + t; e < e1; e += 256) {
    const int c = col[e];
    if (c < t0 || c >= t1) atomicOr(&words[c >> 5], 1u << (c & 31));
  }
  for (int r = t0 + t; r < t1; r += 256)
    if (rowptr[r + 1] - rowptr[r] > kT16Slots) bad = 1;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < nwords; ++i) {
      prefix[i] = run;
      run += __popc(words[i]);
    }
    prefix[nwords] = run;
    if (run > kT16Halo) bad = 1;
  }
  __syncthreads();
  const bool walk = bad != 0;
  if (t == 0) hdr[0] = walk ? -1 : prefix[nwords];
  if (!walk)
    for (int i = t; i < nwords; i += 256) {
      unsigned w = words[i];
      int rank = prefix[i];
      while (w) {
        const int b = __ffs(w) - 1;
        w &= w - 1;
        hdr[1 + rank++] = i * 32 + b;
      }
    }
  for (int i = t; i < (t1 - t0) * kT16Slots; i += 256) {
    const int r = t0 + i / kT16Slots, sidx = i % kT16Slots;
    const int e = rowptr[r] + sidx;
    int sl = kT16Zero;
    float w = 0.f;
    if (!walk && e < rowptr[r + 1]) {
      const int c = col[e];
      sl = (c >= t0 && c < t1) ? c - t0 : kT16Tile + prefix[c >> 5] + __popc(words[c >> 5] & ((1u << (c & 31)) - 1u));
      w = val[e];
    }
    slot[(size_t)r * kT16Slots + sidx] = (u16)(sl * kT16Pieces);   // in 16-byte pieces from the start of the staged rows
    wgt[(size_t)r * kT16Slots + sidx] = w;
  }
}

int launch_csr16t_build(const int32_t *rowptr, const int32_t *col, const float *val, int n_vert, int32_t *plan, hipStream_t s) {
  A3VT_LAUNCH(csr16t_build_kernel, dim3(t16_tiles(n_vert)), dim3(256), 0, s, rowptr, col, val, n_vert, plan);
  A3VT_CHECK_LAUNCH();
  return 0;
}

#ifdef A3VT_DBG_T16_STAMPS   // diagnostic build (tools/build_variants.sh t16): wall clock (s_memrealtime, 100 MHz) of thread 0 at the
// phase boundaries of csr16t_kernel, per workgroup: [0] entry, [1] first unit requested, [3] all units gathered and stored
__device__ unsigned long long g_t16_stamps[4096 * 4];
#define T16_STAMP(k)                                                                                             \
  do {                                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_t16_stamps[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define T16_STAMP(k) do { } while (0)
#endif

struct T16Args {
  const u16 *src;        // [M][ld_src] bf16: raw aggregated columns (forward) / output gradient rows (backward)
  int ld_src;
  u16 *dst;              // forward: layer output rows; backward: dZa [M][ld_dst]
  int ld_dst;
  const float *bias;     // forward
  uint8_t *maskb;        // forward, optional
  int mld, relu;
  float *db_slab;        // backward: [gridDim.x][cpad]
  int c, cpad, n_vert, batch, n_tiles;
  T16Plan plan;
  const int32_t *rowptr, *col;   // the CSR itself, for the tiles that walk their rows
  const float *val;
};

// LDS, per staging buffer: rows [kT16Rows][13] x 16 B, slot records [tile] x 16 B, weights [tile] x 32 B, the halo list.
// Two buffers per workgroup: the rows of its NEXT tile arrive while it gathers the current one.
constexpr int kT16LdsRows = kT16Rows * kT16Pieces * 16;
constexpr int kT16LdsSlots = kT16Tile * 16, kT16LdsWgt = kT16Tile * 32;
constexpr int kT16LdsBuf = kT16LdsRows + kT16LdsSlots + kT16LdsWgt;
constexpr int kT16LdsHdrs = kT16MaxUnits * kT16Hdr * 4;
constexpr int kT16Lds = kT16Bufs * kT16LdsBuf + kT16LdsHdrs;
static_assert(16 * 128 * 4 <= kT16LdsRows, "bias-gradient staging reuses a rows region");

// blockIdx -> (mesh, tile): the tiles of a mesh on neighbouring workgroups of ONE die (they share halo rows in its L2)
__device__ __forceinline__ bool t16_unit(long long u, int batch, int n_tiles, long long &b, int &k) {
  const int xcd = (int)(u & 7);
  const long long loc = u >> 3;
  b = xcd + 8 * (loc / n_tiles);
  k = (int)(loc % n_tiles);
  return b < batch;
}

// LDS reads as written instructions.  The compiler orders every LDS read it knows of behind ALL outstanding vector-memory
// operations once an LDS-DMA is in flight (it cannot tell which buffer the DMA fills): as plain loads the gathers below sat
// behind an s_waitcnt vmcnt(0) per row — the previous row's global stores and the NEXT unit's DMA — 2 200 cycles per row of 16-lane
// groups, 32 / 27 us per launch, no better than the row walk.  Written out, only the waits that are needed remain.
__device__ __forceinline__ u32x4 t16_lds128(unsigned addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ int t16_lds32(unsigned addr) {
  int r;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ void t16_wait3(u32x4 &a, u32x4 &b, u32x4 &c) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ void t16_wait9(u32x4 (&v)[kT16Slots], u32x4 &o) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(o));
}

// The tile's own rows and its records -> LDS (LDS-DMA: wave-contiguous 16-byte pieces); they need nothing from the plan's header.
__device__ __forceinline__ void t16_stage_tile(const T16Args &a, char *buf, long long b, int k) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int t0 = k * kT16Tile, rows_here = min(a.n_vert - t0, kT16Tile);
  const u16 *sb = a.src + b * a.n_vert * (long long)a.ld_src;
  constexpr int kTilePieces = kT16Tile * kT16Pieces;       // 832 = 13 wave instructions
  for (int i0 = wave * 64; i0 < kTilePieces; i0 += 256) {
    const int i = i0 + lane;
    const int r = (i * 5042) >> 16, p = i - r * kT16Pieces;       // i / 13 (exact below 2^12)
    glds16b(sb + (long long)(t0 + (r < rows_here ? r : 0)) * a.ld_src + p * 8, buf + i0 * 16);
  }
  if (t < rows_here) glds16b(a.plan.slot + (size_t)(t0 + t) * kT16Slots, buf + kT16LdsRows + wave * 64 * 16);
  if (t < rows_here * 2) glds16b(a.plan.wgt + (size_t)t0 * kT16Slots + t * 4, buf + kT16LdsRows + kT16LdsSlots + wave * 64 * 16);
  if (t < kT16Pieces) *reinterpret_cast<u32x4 *>(buf + (kT16Zero * kT16Pieces + t) * 16) = u32x4{0u, 0u, 0u, 0u};
}
// The halo rows, from the unit's header in LDS (list = LDS byte address of its first entry).
__device__ __forceinline__ void t16_stage_halo(const T16Args &a, char *buf, long long b, int halo, unsigned list) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u16 *sb = a.src + b * a.n_vert * (long long)a.ld_src;
  const int total = halo * kT16Pieces;
  for (int i0 = wave * 64; i0 < total; i0 += 256) {
    const int i = i0 + lane;
    if (i < total) {
      const int r = (i * 5042) >> 16, p = i - r * kT16Pieces;
      const int v = t16_lds32(list + r * 4);       // (a written read: see t16_lds128)
      glds16b(sb + (long long)v * a.ld_src + p * 8, buf + (kT16Tile * kT16Pieces + i0) * 16);
    }
  }
}

// weighted sum of row r's neighbour rows from LDS (8 slots; unused ones add 0 x the zero row), CSR order; own = row r itself
__device__ __forceinline__ F8 t16_gather(unsigned buf, int r, int hl, bool on, u32x4 &own) {
  u32x4 sl = t16_lds128(buf + kT16LdsRows + r * 16);
  u32x4 wa = t16_lds128(buf + kT16LdsRows + kT16LdsSlots + r * 32);
  u32x4 wb = t16_lds128(buf + kT16LdsRows + kT16LdsSlots + r * 32 + 16);
  t16_wait3(sl, wa, wb);
  const f32x4 w0 = __builtin_bit_cast(f32x4, wa), w1 = __builtin_bit_cast(f32x4, wb);
  u32x4 v[kT16Slots];
  const unsigned lane_base = buf + hl * 16;
#pragma unroll
  for (int q = 0; q < kT16Slots; ++q) {
    const unsigned idx = (q & 1) ? sl[q >> 1] >> 16 : sl[q >> 1] & 0xffffu;     // the neighbour row's first piece
    v[q] = t16_lds128(lane_base + (idx << 4));       // (lanes 13..15 of a group read past the row: unused)
  }
  own = t16_lds128(buf + (r * kT16Pieces + hl) * 16);
  t16_wait9(v, own);
  F8 acc;
  acc.lo = acc.hi = f32x4{0.f, 0.f, 0.f, 0.f};
  if (on) {
#pragma unroll
    for (int q = 0; q < kT16Slots; ++q) {
      const F8 x = unpack8(v[q]);
      const float w = q < 4 ? w0[q] : w1[q - 4];
      acc.lo += w * x.lo;
      acc.hi += w * x.hi;
    }
  }
  return acc;
}

// The forward epilogue of one vertex's 8-channel group, written for instruction count (the kernel is bound by it: 290
// instructions per four rows with csr16_fwd_store, a third of them here): ReLU and sign bits as rowgemmw's relu_bits
// (v_max_f32 / v_min_u32 / v_lshl_or_b32: bit = ReLU output != 0, the same bit as pre > 0), the last, partial group of a row
// stored as dword + short instead of element by element.  Same values and bytes as csr16_fwd_store with relu = 1.
__device__ __forceinline__ unsigned t16_relu_bits(f32x4 &v) {
  float o0, o1, o2, o3;
  unsigned b0, b1, b2, b3;
  asm("v_max_f32 %0, 0, %8\n\tv_max_f32 %1, 0, %9\n\tv_max_f32 %2, 0, %10\n\tv_max_f32 %3, 0, %11\n\t"
      "v_min_u32 %4, 1, %0\n\tv_min_u32 %5, 1, %1\n\tv_min_u32 %6, 1, %2\n\tv_min_u32 %7, 1, %3\n\t"
      "v_lshl_or_b32 %4, %5, 1, %4\n\tv_lshl_or_b32 %4, %6, 2, %4\n\tv_lshl_or_b32 %4, %7, 3, %4"
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  v = f32x4{o0, o1, o2, o3};
  return b0;
}
__device__ __forceinline__ void t16_fwd_store(const F8 &acc, int ch, int c, const F8 &bs, unsigned valid_bits,
                                              u16 *__restrict__ yo, uint8_t *__restrict__ mrow) {
  F8 o;
  o.lo = acc.lo + bs.lo;
  o.hi = acc.hi + bs.hi;
  const unsigned bits = (t16_relu_bits(o.lo) | (t16_relu_bits(o.hi) << 8)) & valid_bits;   // [3:0] channels ch..ch+3, [11:8] ch+4..ch+7
  const u32x4 pk = pack8(o);
  if (ch + 7 < c) {
#ifndef A3VT_T16_NO_NT   // streaming stores: the results are not read again by this launch, and the L2 write-back at its end shrinks (forward 30.9 -> 28.2 us)
    __builtin_nontemporal_store(pk, reinterpret_cast<u32x4 *>(yo));
#else
    *reinterpret_cast<u32x4 *>(yo) = pk;
#endif
  } else {   // the row's last group: channels ch .. c - 1
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (ch + 2 * d + 1 < c) reinterpret_cast<unsigned *>(yo)[d] = pk[d];
      else if (ch + 2 * d < c) yo[2 * d] = (u16)(pk[d] & 0xffffu);
    }
  }
  if (mrow) *reinterpret_cast<u16 *>(mrow + (ch >> 2)) = (u16)bits;
}

// One persistent kernel for both directions.  BWD: the A^T plan; channels [c, cpad) pass the own gradient through; bias-gradient
// partial sums per workgroup (rows of db_slab beyond the persistent workgroups are written as zeros).
// Pipeline per workgroup, units u0, u1, ... (stride = the persistent workgroups): while unit j is gathered out of buffer j & 1,
// unit j + 1's rows arrive in the other buffer and unit j + 2's plan header is on its way into registers.
template <bool BWD>
__global__ __launch_bounds__(256) void csr16t_kernel(T16Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, hl = t & 15, sub = t >> 4;
  const int ch = hl * 8;
  const bool on = ch < (BWD ? a.cpad : a.c);
  F8 bsum;
  bsum.lo = bsum.hi = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nwg = min((int)gridDim.x, kT16Wgs);
  if (BWD && (int)blockIdx.x >= nwg) {      // not a worker: its row of partial sums is zero
    if (t < a.cpad) a.db_slab[(size_t)blockIdx.x * a.cpad + t] = 0.f;
    return;
  }
  const long long units = 8ll * ((a.batch + 7) / 8) * a.n_tiles;
  F8 bs0;
  unsigned valid_bits = 0;      // sign-bit positions of this lane's channels that exist (ch + t < c)
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (ch + q < a.c) valid_bits |= 1u << (q < 4 ? q : q + 4);
  if (!BWD) {
    bs0 = csr16_load_bias(a.bias, ch, a.c);
    // used here once, so that the compiler waits for these loads HERE and not at their first use inside the unit loop — where
    // its s_waitcnt vmcnt(0) would also wait for the next unit's rows, requested a moment before
    asm volatile("" : "+v"(bs0.lo), "+v"(bs0.hi));
  }
  // The plan headers (halo count + list) of ALL this workgroup's units go to LDS once, up front: fetched per unit they were
  // vector loads in flight across the gather loop, and the compiler then put an s_waitcnt vmcnt(0) in front of that loop
  // ("flush before a loop that stores and uses loaded registers") — behind the next unit's DMA.  In the steady state the
  // kernel issues no global load but its DMA.
  char *hdrs = lds + kT16Bufs * kT16LdsBuf;
  const unsigned hdrs_a = (unsigned)(size_t)(__attribute__((address_space(3))) char *)hdrs;
  auto valid = [&](long long u) {
    long long b;
    int k;
    return u < units && t16_unit(u, a.batch, a.n_tiles, b, k);
  };
  {
    int j = 0;
    for (long long u = blockIdx.x; u < units && j < kT16MaxUnits; u += nwg, ++j) {
      long long b;
      int k;
      if (!t16_unit(u, a.batch, a.n_tiles, b, k)) continue;
      if (t < kT16Hdr) reinterpret_cast<int *>(hdrs)[j * kT16Hdr + t] = a.plan.hdr[(size_t)k * kT16Hdr + t];
    }
  }
  __syncthreads();
  auto count_of = [&](int j) { return __builtin_amdgcn_readfirstlane(t16_lds32(hdrs_a + j * kT16Hdr * 4)); };
  auto issue = [&](long long u, int j, int halo, char *buf) {     // all DMA of unit u (valid; its header is slot j) into buf
    long long b;
    int k;
    t16_unit(u, a.batch, a.n_tiles, b, k);
    t16_stage_tile(a, buf, b, k);
    if (halo > 0) t16_stage_halo(a, buf, b, halo, hdrs_a + (j * kT16Hdr + 1) * 4);
  };
  long long u = blockIdx.x;
  T16_STAMP(0);
  int halo_c = valid(u) ? count_of(0) : -2;
  if (halo_c > -2) issue(u, 0, halo_c, lds);
  T16_STAMP(1);
  int par = 0, j = 0;
  for (; u < units; u += nwg, par ^= (kT16Bufs - 1), ++j) {
    char *buf = lds + par * kT16LdsBuf;
    // (the builtin, not an asm: the compiler's own bookkeeping must see that nothing is outstanding here)
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();         // unit u's rows are in buf; everyone has left the other buffer (unit u - nwg)
    const int halo = halo_c;
    halo_c = valid(u + nwg) ? count_of(j + 1) : -2;
    if (kT16Bufs == 2 && halo_c > -2) issue(u + nwg, j + 1, halo_c, lds + (par ^ 1) * kT16LdsBuf);
    if (kT16Bufs == 2 && halo == -2) continue;     // (a padding unit: batch not a multiple of 8)
    if (halo > -2) {
    long long b;
    int k;
    t16_unit(u, a.batch, a.n_tiles, b, k);
    const int t0 = k * kT16Tile, rows_here = min(a.n_vert - t0, kT16Tile);
    const u16 *sb = a.src + b * a.n_vert * (long long)a.ld_src;
    const unsigned bufa = (unsigned)(size_t)(__attribute__((address_space(3))) char *)buf;
    // the epilogue of one row: forward bias / ReLU / sign bytes / bf16 store; backward pass-through columns, bias-gradient sums
    auto finish = [&](int r, const F8 &acc, const u32x4 &own_raw) {
      const long long row = b * a.n_vert + t0 + r;
      if (!BWD) {
        if (on) {
          t16_fwd_store(acc, ch, a.c, bs0, valid_bits, a.dst + row * a.ld_dst + ch, a.maskb ? a.maskb + row * a.mld : nullptr);
        }
      } else {
        F8 own;
        own.lo = own.hi = f32x4{0.f, 0.f, 0.f, 0.f};
        if (on) own = unpack8(own_raw);
        bsum.lo += own.lo;
        bsum.hi += own.hi;
        if (on) {
          F8 out;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float x = ch + q < a.c ? f8_get(acc, q) : f8_get(own, q);
            if (q < 4) out.lo[q] = x;
            else out.hi[q - 4] = x;
          }
#ifndef A3VT_T16_NO_NT   // streaming stores: the results are not read again by this launch, and the L2 write-back at its end shrinks (forward 30.9 -> 28.2 us)
          __builtin_nontemporal_store(pack8(out), reinterpret_cast<u32x4 *>(a.dst + row * a.ld_dst + ch));
#else
          *reinterpret_cast<u32x4 *>(a.dst + row * a.ld_dst + ch) = pack8(out);
#endif
        }
      }
    };
    // (two loops: with the row walk's global loads inside it, the compiler made every LDS read of the tiled loop wait for
    // vector memory — a load of a previous trip could still own the register)
    if (halo >= 0) {
      // (no loop: in front of a loop that stores and reads registers it believes loaded, the compiler "flushes" with an
      // s_waitcnt vmcnt(0) — here that is a wait for the next unit's DMA; kT16Tile / 16 = 4 rows per 16-lane group)
#pragma unroll
      for (int i = 0; i < kT16Tile / 16; ++i) {
        const int r = sub + 16 * i;
        if (r < rows_here) {
          u32x4 own_raw;
          const F8 acc = t16_gather(bufa, r, hl, on, own_raw);
          finish(r, acc, own_raw);
        }
      }
    } else {
      for (int r = sub; r < rows_here; r += 16) {
        const int v = t0 + r;
        u32x4 own_raw = u32x4{0u, 0u, 0u, 0u};
        if (BWD && on) own_raw = *reinterpret_cast<const u32x4 *>(sb + (long long)v * a.ld_src + ch);
        const F8 acc = gather_row16(sb, a.ld_src, ch, on, a.rowptr[v], a.rowptr[v + 1], hl, a.col, a.val);
        finish(r, acc, own_raw);
      }
      // (nothing of this path's vector memory stays outstanding — in the compiler's books, too: registers its loads "may still
      // own" made every LDS read of the tiled path above wait for vector memory)
      __builtin_amdgcn_s_waitcnt(0x0070);
    }
    }   // (unit u valid)
    if (kT16Bufs == 1 && halo_c > -2) {   // one buffer: the next unit's rows are requested when everyone has left this one's
      __builtin_amdgcn_s_barrier();
      issue(u + nwg, j + 1, halo_c, buf);
    }
  }
  T16_STAMP(3);
  if (BWD) {
    wait_vm<0>();
    __syncthreads();          // (no DMA in flight, nobody reads a buffer any more)
    float(*red)[128] = reinterpret_cast<float(*)[128]>(lds);
#pragma unroll
    for (int q = 0; q < 8; ++q) red[sub][hl * 8 + q] = f8_get(bsum, q);
    __syncthreads();
    if (t < 128 && t < a.cpad) {
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sm += red[r][t];
      a.db_slab[(size_t)blockIdx.x * a.cpad + t] = sm;
    }
  }
}

#ifdef A3VT_DBG_T16_STAMPS
}  // namespace a3vt
extern "C" int a3vt_dbg_t16_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(a3vt::g_t16_stamps), sizeof(unsigned long long) * 4096 * 4);
}
namespace a3vt {
#endif

int launch_csr16t_fwd(const void *za, int ldza, const float *bias, int c, const int32_t *plan, const int32_t *rowptr,
                      const int32_t *col, const float *val, int n_vert, int batch, void *y, int ldy, uint8_t *maskb, int mld,
                      int relu, hipStream_t s) {
  if (ldza % 8 != 0 || ldza < ((c + 7) & ~7) || ((c + 7) >> 3) != kT16Pieces || !relu) {   // (the epilogue is the ReLU one)
    set_error("csr16t_fwd: ldza=%d c=%d relu=%d not taken", ldza, c, relu);
    return -1;
  }
  T16Args a{};
  a.src = static_cast<const u16 *>(za);
  a.ld_src = ldza;
  a.dst = static_cast<u16 *>(y);
  a.ld_dst = ldy;
  a.bias = bias;
  a.maskb = maskb;
  a.mld = mld;
  a.relu = relu;
  a.c = c;
  a.cpad = (c + 7) & ~7;
  a.n_vert = n_vert;
  a.batch = batch;
  a.n_tiles = t16_tiles(n_vert);
  a.plan = t16_plan(plan, n_vert);
  a.rowptr = rowptr;
  a.col = col;
  a.val = val;
  static OncePerDevice once;
  once.run([] { (void)hipFuncSetAttribute((const void *)csr16t_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kT16Lds); });
  const long long units = 8ll * ((batch + 7) / 8) * a.n_tiles;
  const int grid = (int)(units < kT16Wgs ? units : kT16Wgs);
  path_count(PATH_CSR16T);
  A3VT_LAUNCH(csr16t_kernel<false>, dim3(grid), dim3(256), kT16Lds, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_csr16t_bwd(const void *g, int ldg, int c, int cpad, const int32_t *planT, const int32_t *rowptrT,
                      const int32_t *colT, const float *valT, int n_vert, int batch, void *dza, int lddza, float *db_slab,
                      hipStream_t s) {
  if (ldg % 8 != 0 || lddza % 8 != 0 || cpad != ((c + 7) & ~7) || (cpad >> 3) != kT16Pieces || lddza < cpad || ldg < cpad) {
    set_error("csr16t_bwd: ldg=%d lddza=%d c=%d cpad=%d not taken", ldg, lddza, c, cpad);
    return -1;
  }
  T16Args a{};
  a.src = static_cast<const u16 *>(g);
  a.ld_src = ldg;
  a.dst = static_cast<u16 *>(dza);
  a.ld_dst = lddza;
  a.db_slab = db_slab;
  a.c = c;
  a.cpad = cpad;
  a.n_vert = n_vert;
  a.batch = batch;
  a.n_tiles = t16_tiles(n_vert);
  a.plan = t16_plan(planT, n_vert);
  a.rowptr = rowptrT;
  a.col = colT;
  a.val = valT;
  static OncePerDevice once;
  once.run([] { (void)hipFuncSetAttribute((const void *)csr16t_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kT16Lds); });
  // exactly the rows of partial bias sums the caller reduces (csr_bwd_num_slabs): the first kT16Wgs workgroups work, the
  // others write a zero row
  const int grid = csr_bwd_num_slabs(batch, n_vert);
  A3VT_LAUNCH(csr16t_kernel<true>, dim3(grid), dim3(256), kT16Lds, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Output layer (3 channels) on bf16 rows: z3 = X W (16 lanes per row, 3 x 8-channel pieces per lane: k <= 384), then
// the fp32 3-channel aggregation of gcn_csr.hip; backward fuses G_prev (bf16, ReLU-masked by X > 0), dW and db partials.
// ------------------------------------------------------------------------------------------------
constexpr int kThin16Pieces = 3;
constexpr int kThin16FwdRows = 4, kThin16FwdBlocks = 768;

__global__ __launch_bounds__(256) void thin16_fwd_kernel(const u16 *__restrict__ x, int ldx, int k,
                                                         const float *__restrict__ w, long long m,
                                                         float *__restrict__ z3) {
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float wr[kThin16Pieces][8][3];
#pragma unroll
  for (int p = 0; p < kThin16Pieces; ++p)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int kk = (p * 16 + l16) * 8 + t;
#pragma unroll
      for (int j = 0; j < 3; ++j) wr[p][t][j] = kk < k ? w[kk * 3 + j] : 0.f;
    }
  // kThin16FwdRows rows per trip: that many times three 16-byte loads in flight per lane (gcn_csr.hip thin_fwd_kernel)
  const long long step = (long long)gridDim.x * 16;
  for (long long row0 = (long long)blockIdx.x * 16 + grp; row0 < m; row0 += kThin16FwdRows * step) {
    u32x4 xr[kThin16FwdRows][kThin16Pieces];
#pragma unroll
    for (int u = 0; u < kThin16FwdRows; ++u) {
      const long long row = row0 + u * step;
#pragma unroll
      for (int p = 0; p < kThin16Pieces; ++p) {
        const int kk = (p * 16 + l16) * 8;
        xr[u][p] = u32x4{0u, 0u, 0u, 0u};
        if (kk < k && row < m) xr[u][p] = *reinterpret_cast<const u32x4 *>(x + row * ldx + kk);
      }
    }
#pragma unroll
    for (int u = 0; u < kThin16FwdRows; ++u) {
      const long long row = row0 + u * step;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int p = 0; p < kThin16Pieces; ++p) {
        const int kk = (p * 16 + l16) * 8;
        if (kk < k) {
          const F8 xv = unpack8(xr[u][p]);
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float xe = f8_get(xv, t);
            s0 += xe * wr[p][t][0];
            s1 += xe * wr[p][t][1];
            s2 += xe * wr[p][t][2];
          }
        }
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off, 16);
        s1 += __shfl_xor(s1, off, 16);
        s2 += __shfl_xor(s2, off, 16);
      }
      if (l16 == 0 && row < m) *reinterpret_cast<f32x4 *>(z3 + row * 4) = f32x4{s0, s1, s2, 0.f};
    }
  }
}

int launch_thin16_fwd_product(const void *x, int ldx, int k, const float *w, long long m, float *z3, hipStream_t s) {
  if (k > kThin16Pieces * 128 || ldx % 8 != 0) {
    set_error("thin16_fwd: k=%d (max %d) ldx=%d unsupported", k, kThin16Pieces * 128, ldx);
    return -1;
  }
  const int grid = (int)(cdiv(m, 16) < kThin16FwdBlocks ? cdiv(m, 16) : kThin16FwdBlocks);
  A3VT_LAUNCH(thin16_fwd_kernel, dim3(grid), dim3(256), 0, s, static_cast<const u16 *>(x), ldx, k, w, m, z3);
  A3VT_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void thin16_bwd_kernel(const u16 *__restrict__ x, int ldx, int k,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ dz3,
                                                         const float *__restrict__ du, long long m, int apply_mask,
                                                         u16 *__restrict__ gprev, int ldg,
                                                         float *__restrict__ dw_slab, float *__restrict__ db_slab) {
  __shared__ float red[16][kThin16Pieces * 24 + 1];
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  float wr[kThin16Pieces][8][3], dwp[kThin16Pieces][8][3];
#pragma unroll
  for (int p = 0; p < kThin16Pieces; ++p)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int kk = (p * 16 + l16) * 8 + t;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wr[p][t][j] = kk < k ? w[kk * 3 + j] : 0.f;
        dwp[p][t][j] = 0.f;
      }
    }
  float db0 = 0.f, db1 = 0.f, db2 = 0.f;
  // two rows per trip (six 16-byte loads of X in flight per lane), same order of accumulation
  const long long step = (long long)gridDim.x * 16;
  for (long long row0 = (long long)blockIdx.x * 16 + grp; row0 < m; row0 += 2 * step) {
    u32x4 xr[2][kThin16Pieces];
    f32x4 d[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long row = row0 + u * step;
      d[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < m) {
        d[u] = *reinterpret_cast<const f32x4 *>(dz3 + row * 4);
        if (l16 == 0) {
          db0 += du[row * 3 + 0];
          db1 += du[row * 3 + 1];
          db2 += du[row * 3 + 2];
        }
      }
#pragma unroll
      for (int p = 0; p < kThin16Pieces; ++p) {
        const int kk = (p * 16 + l16) * 8;
        xr[u][p] = u32x4{0u, 0u, 0u, 0u};
        if (kk < ldg && row < m) xr[u][p] = *reinterpret_cast<const u32x4 *>(x + row * ldx + kk);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long row = row0 + u * step;
      if (row >= m) continue;   // (uniform per 16-lane group)
      u16 *gr = gprev + row * ldg;
#pragma unroll
      for (int p = 0; p < kThin16Pieces; ++p) {
        const int kk = (p * 16 + l16) * 8;
        if (kk < ldg) {  // pad columns [k, ldg) are written as zeros (wr = 0 there)
          const F8 xv = unpack8(xr[u][p]);
          F8 o;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float xe = f8_get(xv, t);
            const float gk = d[u][0] * wr[p][t][0] + d[u][1] * wr[p][t][1] + d[u][2] * wr[p][t][2];
            const float ov = (!apply_mask || xe > 0.f) ? gk : 0.f;
            if (t < 4) o.lo[t] = ov;
            else o.hi[t - 4] = ov;
            dwp[p][t][0] += xe * d[u][0];
            dwp[p][t][1] += xe * d[u][1];
            dwp[p][t][2] += xe * d[u][2];
          }
          *reinterpret_cast<u32x4 *>(gr + kk) = pack8(o);
        }
      }
    }
  }
  float *slab = dw_slab + (size_t)blockIdx.x * k * 3;
  for (int col = 0; col < 16; ++col) {
    if (l16 == col) {
#pragma unroll
      for (int p = 0; p < kThin16Pieces; ++p)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int j = 0; j < 3; ++j) red[grp][(p * 8 + t) * 3 + j] = dwp[p][t][j];
    }
    __syncthreads();
    if (threadIdx.x < kThin16Pieces * 24) {
      float sm = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sm += red[r][threadIdx.x];
      const int p = threadIdx.x / 24, t = (threadIdx.x % 24) / 3, j = threadIdx.x % 3;
      const int kk = (p * 16 + col) * 8 + t;
      if (kk < k) slab[kk * 3 + j] = sm;
    }
    __syncthreads();
  }
  if (l16 == 0) {
    red[grp][0] = db0;
    red[grp][1] = db1;
    red[grp][2] = db2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float sm = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sm += red[r][threadIdx.x];
    db_slab[(size_t)blockIdx.x * 3 + threadIdx.x] = sm;
  }
}

int launch_thin16_bwd_main(const void *x, int ldx, int k, const float *w, const float *dz3, const float *du, long long m,
                           int apply_mask, void *gprev, int ldg, float *dw_slab, float *db_slab, hipStream_t s) {
  if (k > kThin16Pieces * 128 || ldx % 8 != 0 || ldg % 8 != 0 || ldg > kThin16Pieces * 128 || ldg > ldx) {
    set_error("thin16_bwd: k=%d ldx=%d ldg=%d unsupported", k, ldx, ldg);
    return -1;
  }
  A3VT_LAUNCH(thin16_bwd_kernel, dim3(thin_num_slabs()), dim3(256), 0, s, static_cast<const u16 *>(x), ldx, k, w, dz3, du,
              m, apply_mask, static_cast<u16 *>(gprev), ldg, dw_slab, db_slab);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
