// gemm_tile.h — helpers shared by the fp32 / bf16 product kernels (gcn_gemm.hip, gcn_gemmw.hip): LDS-DMA, bf16 packing, counted
// waits, and rowtile_unit — one 16 x 16 output tile with its operands pulled straight from global memory (the leftover rows
// of a load-balanced split).  Internal; moved here unchanged from gcn_gemm.hip in round 6.
#pragma once
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// 16-byte LDS-DMA: each active lane copies 16 B from its own global address to lds_base + lane*16.
__device__ __forceinline__ void glds16(const float *gsrc, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// bf16 operand mode ("bf16 + MFMA feature MLP", BASELINE configs[3]/[4]): the fp32 values staged in LDS are rounded to
// bf16 (RNE, v_cvt_pk_bf16_f32) as they are read into fragments and multiplied with v_mfma_f32_16x16x16_bf16 into the
// same fp32 accumulators.  A lane's ds_read_b128 already holds k = 4q..4q+3 of its row — exactly the operand layout of
// the 16x16x16 instruction — so ONE bf16 MFMA replaces the four fp32 16x16x4 steps (1/8 of the matrix-pipe time).
using f32x2 = __attribute__((ext_vector_type(2))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ s16x4 cvt_bf16x4(f32x4 v) {
  const bf16x2 lo = __builtin_convertvector((f32x2){v[0], v[1]}, bf16x2);
  const bf16x2 hi = __builtin_convertvector((f32x2){v[2], v[3]}, bf16x2);
  const u32x2 r = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, r);
}

// bf16 STORAGE mode (gemm mode 2, "bf16s"): activations, gradients and the weight images are stored as bf16; a staged
// 64-byte chunk row then holds 32 k values instead of 16 and a lane's ds_read_b128 (8 consecutive bf16 of its row) IS
// the A / B operand of v_mfma_f32_16x16x32_bf16 — no conversion, one MFMA per (m-tile, n-tile) and chunk, half the chunks.
// All pointers / strides of RowGemmArgs stay in 4-byte units on the operand side (a row of 304 bf16 = 152 "floats"); the
// outputs of the hidden-layer epilogues are bf16 with their leading dimension in elements.  fp32 accumulation throughout.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16 = unsigned short;
__device__ __forceinline__ f32x4 mfma_bf16s(f32x4 a, f32x4 b, f32x4 c) {
#ifdef A3VT_DBG_RG_NOMFMA
  c[0] += a[0] * b[0];   // one VALU op instead of the matrix instruction (operands stay live)
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
__device__ __forceinline__ u16 to_bf16(float v) {  // round to nearest even (v_cvt_pk_bf16_f32)
  const bf16x2 r = __builtin_convertvector((f32x2){v, 0.f}, bf16x2);
  return (u16)(__builtin_bit_cast(unsigned, r) & 0xffffu);
}
__device__ __forceinline__ f32x4 pack_bf16x8(f32x4 lo, f32x4 hi) {  // 8 floats -> 8 bf16 in one 16-byte register group
  const s16x4 a = cvt_bf16x4(lo), b = cvt_bf16x4(hi);
  const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  return __builtin_bit_cast(f32x4, (u32x4){ua[0], ua[1], ub[0], ub[1]});
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One 16 x 16 output tile — rows row_base + 16 mt .., columns n0 .. — with its operands pulled straight from global
// memory into registers: all loads of up to 19 K-chunks in flight at once, no LDS, no barrier, then the MFMA chain (one
// round trip instead of nineteen).  Same arithmetic order along K as rowgemm_kernel.  Used for the handful of rows the
// load-balanced split leaves over (launch_rowgemm_epi): by rowtile_kernel, and by the tail of rowgemm_kernel itself.
template <int EPI, int MODE>
__device__ __forceinline__ void rowtile_unit(const RowGemmArgs &p, int row_base, int row_end, int mt, int n0, int lane) {
  constexpr int KB = 19;  // K-chunks (of 16) per register block: all of K = 300 in one round trip
  const int l16 = lane & 15, q = lane >> 4;
  const int ar = min(row_base + mt * 16 + l16, row_end - 1);  // A row of this lane (ragged tail: duplicate, never stored)
  const int br = min(n0 + l16, p.bt_rows - 1);                 // Bt row (= output column) of this lane
  // a0 may be quad-major (RowGemmArgs::a0q_nvert): element (row, k) at ((b Q + k / 4) N + v) * 4 + k % 4 -> k * N past the
  // row's base for the 4-aligned k a lane reads
  const float *a1r = p.a1 + (size_t)ar * p.lda1;
  const float *a0r = p.a0 + (size_t)ar * p.lda0;
  size_t a0mul = 1;
  if (p.a0q_nvert > 0) {
    const int bq = ar / p.a0q_nvert;
    a0r = p.a0 + ((size_t)bq * p.a0q_quads * p.a0q_nvert + (size_t)(ar - bq * p.a0q_nvert)) * 4;
    a0mul = (size_t)p.a0q_nvert;
  }
  const float *btr = p.bt + (size_t)br * p.ldb;
  const int nch = (p.k + 15) >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nch; c0 += KB) {
    f32x4 af[KB], bf[KB];
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      const int kk = (c0 + c) * 16 + q * 4;
      const bool on = c0 + c < nch && kk < p.k;
      af[c] = on ? *reinterpret_cast<const f32x4 *>(kk < p.ksplit ? a0r + (size_t)kk * a0mul : a1r + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
      bf[c] = on ? *reinterpret_cast<const f32x4 *>(btr + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < KB; ++c) {
      if (MODE == 2) {
        acc = mfma_bf16s(af[c], bf[c], acc);
      } else if (MODE == 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(cvt_bf16x4(af[c]), cvt_bf16x4(bf[c]), acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][t], bf[c][t], acc, 0, 0, 0);
      }
    }
  }
  // C/D layout: this lane holds column n0 + l16 of rows 4q .. 4q+3 of the tile
  const int col = n0 + l16;
  const bool col_ok = col < p.n_store;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row_base + mt * 16 + q * 4 + r;
    const bool row_ok = row < row_end;
    float v = acc[r];
    if (EPI == EPI_PLAIN) {
      if (p.plain_relu) v = v > 0.f ? v : 0.f;
      if (row_ok && col_ok) p.c[(size_t)row * p.ldc + col] = v;
    } else if (EPI == EPI_FWD_HIDDEN) {
      // ReLU-sign byte of 4 consecutive pass-through columns: gathered from the 4 lanes that hold them
      const unsigned mybit = (col_ok && col >= p.csplit && v > 0.f) ? 1u << (l16 & 3) : 0u;
      unsigned bits = mybit;
      bits |= __shfl_xor(bits, 1, 64);
      bits |= __shfl_xor(bits, 2, 64);
      if (row_ok && p.maskb && (l16 & 3) == 0 && (col | 3) >= p.csplit && col < ((p.n_store + 3) & ~3))
        p.maskb[(size_t)row * p.mld + p.moff + (col >> 2)] = (uint8_t)bits;
      if (MODE == 2) {  // bf16 rows; the pad columns [n_store, ldc) are written too (exact zeros: Bt rows there are zero)
        if (row_ok && col < p.ldc) {
          if (col < p.csplit) reinterpret_cast<u16 *>(p.c2)[(size_t)row * p.ldc2 + col] = to_bf16(v);
          else reinterpret_cast<u16 *>(p.c)[(size_t)row * p.ldc + col] = to_bf16((v > 0.f || p.no_relu) ? v : 0.f);
        }
      } else if (row_ok && col_ok) {
        if (p.zq_nvert > 0 && col < p.zq_quads * 4) {   // quad-major raw columns (RowGemmArgs::zq_nvert)
          const int bq = row / p.zq_nvert;
          p.c2[(((size_t)bq * p.zq_quads + (col >> 2)) * p.zq_nvert + (row - bq * p.zq_nvert)) * 4 + (col & 3)] = v;
        } else if (p.zq_nvert > 0 && col < p.yq_quads * 4) {   // quad-major pass-through columns of the activations
          const int bq = row / p.zq_nvert;
          p.yq[(((size_t)bq * p.yq_quads + (col >> 2)) * p.zq_nvert + (row - bq * p.zq_nvert)) * 4 + (col & 3)] =
              (v > 0.f || p.no_relu) ? v : 0.f;
        } else if (p.zq_nvert == 0 && col < p.csplit) {
          p.c2[(size_t)row * p.ldc2 + col] = v;
        } else {
          p.c[(size_t)row * p.ldc + col] = (v > 0.f || p.no_relu) ? v : 0.f;
        }
      }
    } else {  // EPI_DX_MASK
      if (MODE == 2) {
        if (row_ok && col < p.ldc) {
          const unsigned byte = col_ok ? p.maskb[(size_t)row * p.mld + (col < p.csplit ? 0 : p.moff) + (col >> 2)] : 0u;
          reinterpret_cast<u16 *>(p.c)[(size_t)row * p.ldc + col] = to_bf16(((byte >> (col & 3)) & 1u) ? v : 0.f);
        }
      } else if (row_ok && col_ok) {
        if (p.zq_nvert > 0 && col < p.zq_quads * 4) {   // quad-major gradient columns: unmasked (RowGemmArgs::zq_nvert)
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

// base + a 32-bit byte offset: the form the compiler turns into a scalar base + a vector offset (no 64-bit vector arithmetic)
__device__ __forceinline__ const char *w_at(const void *base, unsigned off) { return reinterpret_cast<const char *>(base) + (size_t)off; }

}  // namespace a3vt
