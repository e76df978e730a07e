// bnrelu.hip — training-mode BatchNorm2d + ReLU of the image pyramid as ONE operator on channels-last bf16 maps:
//   y = relu(gamma * (x - mean_c) / sqrt(var_c + eps) + beta),   mean / var over (n, h, w) per channel, biased var for y,
//   running_mean / running_var (unbiased) / num_batches_tracked updated as nn.BatchNorm2d does           (x, y: [rows = N*H*W][C] bf16)
// and its backward (d gamma, d beta in fp32, dx in bf16).  Reference: `CNN_layer` = BatchNorm2d -> ReLU -> Conv2d,
// pterotactyl/reconstruction/vision/model.py:15-23, applied 13 times per encoder and step (`Image_Encoder.forward`, :147-164).
//
// On torch ops the pair is 3 + 1 launches forward (MIOpen's mean-variance, final mean-variance, normalise; ReLU in place),
// 3 + 1 backward, plus the counter increment: 234 launches and 2.2 ms of a configs[3] step (profiles/r06_config3_bf16s_steady_kernels.txt).
// Here: three launches each way (the counter increment is in the second).  HBM-bound: forward reads x twice and writes y once, backward reads dy and x twice and writes dx.
//
// A thread walks 16-byte pieces gtid, gtid + S, ... with 8 * S a multiple of C (bias_grad.hip's walk), so each of its eight
// register slots stays on ONE channel: per-channel coefficients are loaded once, sums need no index arithmetic.  Sums are
// folded per workgroup in a fixed order, written as partial rows, and a one-workgroup launch adds the rows in a fixed order in
// float64: the results do not depend on scheduling.  (The rows were first added by the workgroup that arrives last at a counter,
// two launches each way: the device-scope fences that needs — an L2 write-back per workgroup on this multi-die part — cost
// 30 us per launch on the 32 MB maps, against 5 us for the extra launch.)
// The ReLU mask of the backward pass is recomputed from x with the forward's own fma and coefficients (bit-identical to y > 0).
#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

constexpr int kBnThreads = 256;
constexpr int kBnMaxWgs = 1024;

__device__ __forceinline__ float bn_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bn_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t bn_pack(float a, float b) {   // two floats -> two bf16, round to nearest even
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a, b}, b2));
}

__device__ __forceinline__ float bn_relu(float r) { return (r > 0.f || r != r) ? r : 0.f; }   // NaN stays NaN (torch.relu)

// piece `p` (8 consecutive elements) of a bf16 array of n_elem elements; elements past the end read as 0
__device__ __forceinline__ void bn_load(const uint16_t *g, long long p, long long n_elem, float v[8]) {
  const long long e0 = p * 8;
  if (e0 >= n_elem) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
  } else if (e0 + 8 <= n_elem) {
    const uint4 u = *reinterpret_cast<const uint4 *>(g + e0);
    v[0] = bn_lo(u.x); v[1] = bn_hi(u.x); v[2] = bn_lo(u.y); v[3] = bn_hi(u.y);
    v[4] = bn_lo(u.z); v[5] = bn_hi(u.z); v[6] = bn_lo(u.w); v[7] = bn_hi(u.w);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = e0 + j < n_elem ? bn_lo(g[e0 + j]) : 0.f;
  }
}
__device__ __forceinline__ void bn_store(uint16_t *g, long long p, long long n_elem, const float v[8]) {
  const long long e0 = p * 8;
  if (e0 + 8 <= n_elem) {
    *reinterpret_cast<uint4 *>(g + e0) = make_uint4(bn_pack(v[0], v[1]), bn_pack(v[2], v[3]), bn_pack(v[4], v[5]), bn_pack(v[6], v[7]));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (e0 + j < n_elem) g[e0 + j] = (uint16_t)(bn_pack(v[j], 0.f) & 0xffffu);   // (nothing for pieces past the end)
  }
}

// Fold NACC accumulator sets of 8 slots per thread into per-channel sums of this workgroup and write them as row
// blockIdx.x of `partial` ([gridDim.x][NACC][c]).  Slot j of thread u holds channel (base + 8 u + j) mod c.  Three forms, all
// with a fixed order of additions:
//  * c a multiple of 8 with a power-of-two period P = c / 8 <= 64 pieces (the pyramid's 16 ... 256): lanes l, l + P, ... of a
//    wave hold the same eight channels — butterfly over the lane bits above log2 P, then the four waves through LDS;
//  * c == 3 (the full-resolution map): a thread's eight slots are channels (phase + j) mod 3 — three sums per thread, a
//    butterfly over all 64 lanes, then the waves;
//  * anything else: every channel's owner thread walks the workgroup's 256 x 8 slots in LDS.
template <int NACC>
__device__ __forceinline__ void bn_fold(float *lds, const float acc[NACC][8], int c, float *partial) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int base = (int)(((long long)blockIdx.x * kBnThreads * 8) % c);
  const int period = c >> 3;
  if ((c & 7) == 0 && period <= 64 && (period & (period - 1)) == 0) {
    float v[NACC][8];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[a][j] = acc[a][j];
    for (int x = 32; x >= period; x >>= 1)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[a][j] += __shfl_xor(v[a][j], x, 64);
    // lane l < P of every wave: channels ((base / 8 + l) mod P) * 8 + j  (base is a multiple of 8 here)
    if (lane < period) {
      const int grp = ((base >> 3) + lane) & (period - 1);
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) lds[(wave * NACC + a) * c + grp * 8 + j] = v[a][j];
    }
    __syncthreads();
    for (int i = t; i < NACC * c; i += kBnThreads) {
      const float s = ((lds[i] + lds[NACC * c + i]) + lds[2 * NACC * c + i]) + lds[3 * NACC * c + i];
      partial[(size_t)blockIdx.x * NACC * c + i] = s;
    }
    return;
  }
  if (c == 3) {
    const int phase = (base + 8 * t) % 3;   // channel of slot 0
    float v[NACC][3];
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      // slots by residue r = j mod 3: {0, 3, 6}, {1, 4, 7}, {2, 5}; residue r is channel (phase + r) mod 3
      const float r0 = (acc[a][0] + acc[a][3]) + acc[a][6], r1 = (acc[a][1] + acc[a][4]) + acc[a][7], r2 = acc[a][2] + acc[a][5];
      v[a][0] = phase == 0 ? r0 : (phase == 1 ? r2 : r1);
      v[a][1] = phase == 0 ? r1 : (phase == 1 ? r0 : r2);
      v[a][2] = phase == 0 ? r2 : (phase == 1 ? r1 : r0);
    }
    for (int x = 32; x >= 1; x >>= 1)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) v[a][k] += __shfl_xor(v[a][k], x, 64);
    if (lane == 0)
#pragma unroll
      for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) lds[(wave * NACC + a) * 3 + k] = v[a][k];
    __syncthreads();
    if (t < NACC * 3) partial[(size_t)blockIdx.x * NACC * 3 + t] = ((lds[t] + lds[NACC * 3 + t]) + lds[2 * NACC * 3 + t]) + lds[3 * NACC * 3 + t];
    return;
  }
  float(*part)[NACC * 8 + 1] = reinterpret_cast<float(*)[NACC * 8 + 1]>(lds);
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int j = 0; j < 8; ++j) part[t][a * 8 + j] = acc[a][j];
  __syncthreads();
  for (int ch = t; ch < c; ch += kBnThreads) {
    int x = ch - base;
    x = x < 0 ? x + c : x;
    float s[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) s[a] = 0.f;
    for (int u = 0; u < kBnThreads; ++u) {
      for (int j = x; j < 8; j += c)
#pragma unroll
        for (int a = 0; a < NACC; ++a) s[a] += part[u][a * 8 + j];
      x -= 8 % c;
      x = x < 0 ? x + c : x;
    }
#pragma unroll
    for (int a = 0; a < NACC; ++a) partial[((size_t)blockIdx.x * NACC + a) * c + ch] = s[a];
  }
}

// s += rows sub, sub + nsub, ... of column 0 of partial[.][0][.], ss += the same rows of partial[.][1][.], eight rows' loads in
// flight at a time (one workgroup adds up to 64 values per thread here: issued one by one this was the longest part of a launch)
__device__ __forceinline__ void bn_sum_rows(const float *col, int c, int sub, int nsub, int nwg, double &s, double &ss) {
  int g = sub;
  for (; g + 7 * nsub < nwg; g += 8 * nsub) {
    float a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a[k] = __builtin_nontemporal_load(col + ((size_t)(g + k * nsub) * 2 + 0) * c);
      b[k] = __builtin_nontemporal_load(col + ((size_t)(g + k * nsub) * 2 + 1) * c);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += (double)a[k];
      ss += (double)b[k];
    }
  }
  for (; g < nwg; g += nsub) {
    s += (double)col[((size_t)g * 2 + 0) * c];
    ss += (double)col[((size_t)g * 2 + 1) * c];
  }
}

struct BnFwd {
  const uint16_t *x;
  uint16_t *y;
  long long n_elem, rows;
  int c;
  const float *gamma, *beta;
  const float *pre_bias;   // optional [c]: a per-channel constant the producer did NOT add to x (a Conv2d bias): the statistics of x + pre_bias
                           // differ from x's by the mean alone, so y is unchanged and only running_mean takes it
  float eps, momentum;
  float *running_mean, *running_var;
  long long *num_batches;
  float *save;       // [4][c]: mean, invstd, scale = gamma * invstd, shift = beta - mean * scale
  float *partial;    // [wgs][2][c]
};

__global__ __launch_bounds__(kBnThreads) void bnrelu_stats_kernel(BnFwd p) {
  __shared__ float part[kBnThreads * 17];
  const int t = threadIdx.x;
  const long long stride = (long long)gridDim.x * kBnThreads;
  const long long n_piece = (p.n_elem + 7) / 8;
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  long long q = (long long)blockIdx.x * kBnThreads + t;
  for (; q < n_piece; q += 8 * stride) {   // eight pieces in flight; pieces past the end read as zeros
    float v[8][8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bn_load(p.x, q + u * stride, p.n_elem, v[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[0][j] += v[u][j];
        acc[1][j] = fmaf(v[u][j], v[u][j], acc[1][j]);
      }
  }
  bn_fold<2>(part, acc, p.c, p.partial);
}

// one workgroup: per-channel mean / variance from the nwg partial rows (float64, fixed order), coefficients, running statistics
__global__ __launch_bounds__(kBnThreads) void bnrelu_stats_final_kernel(BnFwd p, int nwg) {
  __shared__ double red[kBnThreads][2];
  const int t = threadIdx.x;
  const int c = p.c;
  for (int c0 = 0; c0 < c; c0 += kBnThreads) {
    const int cw = c - c0 < kBnThreads ? c - c0 : kBnThreads;   // channels of this round
    const int nsub = kBnThreads / cw;
    const int ch = t % cw, sub = t / cw;
    double s = 0.0, ss = 0.0;
    if (sub < nsub) bn_sum_rows(p.partial + c0 + ch, c, sub, nsub, nwg, s, ss);
    __syncthreads();
    red[t][0] = s;
    red[t][1] = ss;
    __syncthreads();
    if (sub == 0) {
      for (int k = 1; k < nsub; ++k) {
        s += red[k * cw + ch][0];
        ss += red[k * cw + ch][1];
      }
      const double m = (double)p.rows;
      const double mean = s / m;
      double var = ss / m - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const float meanf = (float)mean, varf = (float)var;
      const float invstd = 1.0f / sqrtf(varf + p.eps);
      const float scale = p.gamma[c0 + ch] * invstd;
      const float shift = fmaf(-meanf, scale, p.beta[c0 + ch]);
      p.save[0 * c + c0 + ch] = meanf;
      p.save[1 * c + c0 + ch] = invstd;
      p.save[2 * c + c0 + ch] = scale;
      p.save[3 * c + c0 + ch] = shift;
      if (p.running_mean) {
        const float unb = (float)(var * m / (m - 1.0));
        const float mean_in = p.pre_bias ? meanf + p.pre_bias[c0 + ch] : meanf;
        p.running_mean[c0 + ch] = fmaf(p.momentum, mean_in - p.running_mean[c0 + ch], p.running_mean[c0 + ch]);
        p.running_var[c0 + ch] = fmaf(p.momentum, unb - p.running_var[c0 + ch], p.running_var[c0 + ch]);
      }
    }
  }
  if (t == 0 && p.num_batches) *p.num_batches += 1;
}

// y = relu(x * scale_c + shift_c)
__global__ __launch_bounds__(kBnThreads) void bnrelu_apply_kernel(BnFwd p) {
  const long long stride = (long long)gridDim.x * kBnThreads;
  const long long n_piece = (p.n_elem + 7) / 8;
  long long q = (long long)blockIdx.x * kBnThreads + threadIdx.x;
  float sc[8], sh[8];
  {
    int ch = (int)((q * 8) % p.c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = p.save[2 * p.c + ch];
      sh[j] = p.save[3 * p.c + ch];
      ch = ch + 1 == p.c ? 0 : ch + 1;
    }
  }
  for (; q < n_piece; q += 4 * stride) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) bn_load(p.x, q + u * stride, p.n_elem, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[u][j] = bn_relu(fmaf(v[u][j], sc[j], sh[j]));
      bn_store(p.y, q + u * stride, p.n_elem, v[u]);
    }
  }
}

struct BnBwd {
  const uint16_t *dy, *x;
  uint16_t *dx;
  long long n_elem, rows;
  int c;
  const float *save;   // [4][c] of the forward
  float *dgamma, *dbeta;
  float *coef;         // [2][c]: dx = scale * g + a * x + b
  float *partial;      // [wgs][2][c]; the dx launch reuses it for its column sums ([wgs][c])
  float *dx_colsum;    // optional [c]: sum over the rows of dx AS STORED (bf16) = the bias gradient of the convolution in front
};

// d beta = sum g, d gamma = sum g * xhat with g = dy where the forward's output was positive
__global__ __launch_bounds__(kBnThreads) void bnrelu_bwd_reduce_kernel(BnBwd p) {
  __shared__ float part[kBnThreads * 17];
  const int t = threadIdx.x;
  const long long stride = (long long)gridDim.x * kBnThreads;
  const long long n_piece = (p.n_elem + 7) / 8;
  long long q = (long long)blockIdx.x * kBnThreads + t;
  float sc[8], sh[8], mu[8], is[8];
  {
    int ch = (int)((q * 8) % p.c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      mu[j] = p.save[0 * p.c + ch];
      is[j] = p.save[1 * p.c + ch];
      sc[j] = p.save[2 * p.c + ch];
      sh[j] = p.save[3 * p.c + ch];
      ch = ch + 1 == p.c ? 0 : ch + 1;
    }
  }
  float acc[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = acc[1][j] = 0.f;
  for (; q < n_piece; q += 4 * stride) {
    float v[4][8], g[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bn_load(p.x, q + u * stride, p.n_elem, v[u]);
      bn_load(p.dy, q + u * stride, p.n_elem, g[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = fmaf(v[u][j], sc[j], sh[j]) > 0.f ? g[u][j] : 0.f;
        acc[0][j] += gg;
        acc[1][j] = fmaf(gg, (v[u][j] - mu[j]) * is[j], acc[1][j]);
      }
  }
  bn_fold<2>(part, acc, p.c, p.partial);
}

// one workgroup: d gamma, d beta from the nwg partial rows, and the coefficients of dx
__global__ __launch_bounds__(kBnThreads) void bnrelu_bwd_final_kernel(BnBwd p, int nwg) {
  __shared__ double red[kBnThreads][2];
  const int t = threadIdx.x;
  const int c = p.c;
  for (int c0 = 0; c0 < c; c0 += kBnThreads) {
    const int cw = c - c0 < kBnThreads ? c - c0 : kBnThreads;
    const int nsub = kBnThreads / cw;
    const int ch = t % cw, sub = t / cw;
    double s = 0.0, ss = 0.0;
    if (sub < nsub) bn_sum_rows(p.partial + c0 + ch, c, sub, nsub, nwg, s, ss);
    __syncthreads();
    red[t][0] = s;
    red[t][1] = ss;
    __syncthreads();
    if (sub == 0) {
      for (int k = 1; k < nsub; ++k) {
        s += red[k * cw + ch][0];
        ss += red[k * cw + ch][1];
      }
      const float db = (float)s, dg = (float)ss;
      p.dbeta[c0 + ch] = db;
      p.dgamma[c0 + ch] = dg;
      // dx = scale * (g - db / M - xhat * dg / M),  xhat = (x - mean) * invstd
      const float mean = p.save[0 * c + c0 + ch], invstd = p.save[1 * c + c0 + ch], scale = p.save[2 * c + c0 + ch];
      const float inv_m = (float)(1.0 / (double)p.rows);
      const float a = -scale * dg * invstd * inv_m;
      p.coef[0 * c + c0 + ch] = a;
      p.coef[1 * c + c0 + ch] = fmaf(-a, mean, -scale * db * inv_m);
    }
  }
}

// dx = scale * g + a * x + b.  COLSUM: also the per-channel sums of the bf16 values written — the producer of x is a Conv2d,
// and its bias gradient is exactly this sum (bias_grad.hip would read dx again for it: 22 us + a reduce launch per layer).
template <bool COLSUM>
__global__ __launch_bounds__(kBnThreads) void bnrelu_bwd_dx_kernel(BnBwd p) {
  __shared__ float part[COLSUM ? kBnThreads * 9 : 1];
  const long long stride = (long long)gridDim.x * kBnThreads;
  const long long n_piece = (p.n_elem + 7) / 8;
  long long q = (long long)blockIdx.x * kBnThreads + threadIdx.x;
  float sc[8], sh[8], ca[8], cb[8];
  {
    int ch = (int)((q * 8) % p.c);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[j] = p.save[2 * p.c + ch];
      sh[j] = p.save[3 * p.c + ch];
      ca[j] = p.coef[0 * p.c + ch];
      cb[j] = p.coef[1 * p.c + ch];
      ch = ch + 1 == p.c ? 0 : ch + 1;
    }
  }
  float acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[0][j] = 0.f;
  for (; q < n_piece; q += 4 * stride) {
    float v[4][8], g[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bn_load(p.x, q + u * stride, p.n_elem, v[u]);
      bn_load(p.dy, q + u * stride, p.n_elem, g[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = fmaf(v[u][j], sc[j], sh[j]) > 0.f ? g[u][j] : 0.f;
        v[u][j] = fmaf(sc[j], gg, fmaf(ca[j], v[u][j], cb[j]));
      }
      bn_store(p.dx, q + u * stride, p.n_elem, v[u]);
      if (COLSUM) {
        const long long e0 = (q + u * stride) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (e0 + j < p.n_elem) acc[0][j] += bn_lo(bn_pack(v[u][j], 0.f));   // the value as stored
      }
    }
  }
  if (COLSUM) bn_fold<1>(part, acc, p.c, p.partial);
}

// one workgroup: out[ch] = sum over the nwg partial rows ([nwg][c]) in float64, fixed order
__global__ __launch_bounds__(kBnThreads) void bnrelu_colsum_final_kernel(const float *partial, int nwg, int c, float *out) {
  __shared__ double red[kBnThreads];
  const int t = threadIdx.x;
  for (int c0 = 0; c0 < c; c0 += kBnThreads) {
    const int cw = c - c0 < kBnThreads ? c - c0 : kBnThreads;
    const int nsub = kBnThreads / cw;
    const int ch = t % cw, sub = t / cw;
    double s = 0.0;
    if (sub < nsub) {
      int g = sub;
      for (; g + 7 * nsub < nwg; g += 8 * nsub) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = __builtin_nontemporal_load(partial + (size_t)(g + k * nsub) * c + c0 + ch);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += (double)a[k];
      }
      for (; g < nwg; g += nsub) s += (double)partial[(size_t)g * c + c0 + ch];
    }
    __syncthreads();
    red[t] = s;
    __syncthreads();
    if (sub == 0) {
      for (int k = 1; k < nsub; ++k) s += red[k * cw + ch];
      out[c0 + ch] = (float)s;
    }
  }
}

// ---- every convolution weight and bias of the pyramid to bf16 in ONE launch -------------------------------------------------
// tensor k: src fp32 [outer][inner][hw] (a Conv2d weight, O x I x (KH KW); a bias: inner = hw = 1) -> dst bf16 [outer][hw][inner]
// (the channels-last layout MIOpen's NHWC kernels take).  One thread per destination element.
__global__ __launch_bounds__(256) void cast_weights_kernel(CastBatch b) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= b.start[b.n]) return;
  int lo = 0, hi = b.n - 1;          // the tensor this element belongs to: start[k] <= e < start[k + 1]
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (b.start[mid] <= e) lo = mid;
    else hi = mid - 1;
  }
  const long long r = e - b.start[lo];
  const int inner = b.inner[lo], hw = b.hw[lo];
  const long long o = r / ((long long)inner * hw);
  const int rem = (int)(r - o * inner * hw);
  const int p = rem / inner, i = rem - p * inner;          // destination order: [o][p][i]
  const float v = b.src[lo][(o * inner + i) * hw + p];
  b.dst[lo][r] = (uint16_t)(bn_pack(v, 0.f) & 0xffffu);
}

}  // namespace

// Workgroups of a launch over n_elem elements of c channels: a multiple of c / gcd(c, 2048) (a thread's stride is then whole
// periods of the channel pattern), enough for `per_thread` pieces per thread, at most `cap`.  0: c not supported.
int bnrelu_wgs(long long n_elem, int c, int per_thread, int cap) {
  long long a = c, b = 2048;
  while (b) { const long long r = a % b; a = b; b = r; }
  const int m = (int)(c / a);
  if (m > kBnMaxWgs) return 0;
  cap = cap > kBnMaxWgs ? kBnMaxWgs : cap;
  const long long want = (n_elem / 8 + (long long)kBnThreads * per_thread - 1) / ((long long)kBnThreads * per_thread);
  long long n = want < cap ? want : cap;
  n = n / m * m;
  return (int)(n < m ? m : n);
}
// the reducing launches: their last workgroup adds one partial row per workgroup and channel — at most 64 values per thread
static int bnrelu_reduce_wgs(long long n_elem, int c) { return bnrelu_wgs(n_elem, c, 4, 8192 / c < 16 ? 16 : 8192 / c); }

// scratch: 64 bytes unused, then the backward's coefficients [2][c], then the partial rows [kBnMaxWgs][2][c]
size_t bnrelu_scratch_bytes(int c) { return 64 + ((size_t)2 * c + (size_t)kBnMaxWgs * 2 * c) * sizeof(float); }

int launch_bnrelu_fwd(const void *x, long long rows, int c, const float *gamma, const float *beta, const float *pre_bias, float eps,
                      float momentum, float *running_mean, float *running_var, long long *num_batches, void *y, float *save,
                      void *scratch, hipStream_t s) {
  BnFwd p;
  p.x = static_cast<const uint16_t *>(x);
  p.y = static_cast<uint16_t *>(y);
  p.rows = rows;
  p.n_elem = rows * c;
  p.c = c;
  p.gamma = gamma;
  p.beta = beta;
  p.pre_bias = pre_bias;
  p.eps = eps;
  p.momentum = momentum;
  p.running_mean = running_mean;
  p.running_var = running_var;
  p.num_batches = num_batches;
  p.save = save;
  p.partial = reinterpret_cast<float *>(static_cast<char *>(scratch) + 64) + 2 * c;
  const int g1 = bnrelu_reduce_wgs(p.n_elem, c), g2 = bnrelu_wgs(p.n_elem, c, 4, kBnMaxWgs);
  A3VT_CHECK_ARG(g1 > 0 && g2 > 0);
  A3VT_LAUNCH(bnrelu_stats_kernel, dim3(g1), dim3(kBnThreads), 0, s, p);
  A3VT_CHECK_LAUNCH();
  A3VT_LAUNCH(bnrelu_stats_final_kernel, dim3(1), dim3(kBnThreads), 0, s, p, g1);
  A3VT_CHECK_LAUNCH();
  A3VT_LAUNCH(bnrelu_apply_kernel, dim3(g2), dim3(kBnThreads), 0, s, p);
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_bnrelu_bwd(const void *dy, const void *x, long long rows, int c, const float *save, void *dx, float *dgamma,
                      float *dbeta, float *dx_colsum, void *scratch, hipStream_t s) {
  BnBwd p;
  p.dy = static_cast<const uint16_t *>(dy);
  p.x = static_cast<const uint16_t *>(x);
  p.dx = static_cast<uint16_t *>(dx);
  p.rows = rows;
  p.n_elem = rows * c;
  p.c = c;
  p.save = save;
  p.dgamma = dgamma;
  p.dbeta = dbeta;
  p.dx_colsum = dx_colsum;
  p.coef = reinterpret_cast<float *>(static_cast<char *>(scratch) + 64);
  p.partial = p.coef + 2 * c;
  const int g1 = bnrelu_reduce_wgs(p.n_elem, c), g2 = bnrelu_wgs(p.n_elem, c, 4, kBnMaxWgs);
  A3VT_CHECK_ARG(g1 > 0 && g2 > 0);
  A3VT_LAUNCH(bnrelu_bwd_reduce_kernel, dim3(g1), dim3(kBnThreads), 0, s, p);
  A3VT_CHECK_LAUNCH();
  A3VT_LAUNCH(bnrelu_bwd_final_kernel, dim3(1), dim3(kBnThreads), 0, s, p, g1);
  A3VT_CHECK_LAUNCH();
  if (dx_colsum == nullptr) {
    A3VT_LAUNCH(bnrelu_bwd_dx_kernel<false>, dim3(g2), dim3(kBnThreads), 0, s, p);
    A3VT_CHECK_LAUNCH();
    return 0;
  }
  // (fewer, longer workgroups where the channel count is large: the one-workgroup sum behind adds a row per workgroup)
  const int g3 = bnrelu_wgs(p.n_elem, c, 4, 16384 / c < 16 ? 16 : 16384 / c);
  A3VT_CHECK_ARG(g3 > 0);
  A3VT_LAUNCH(bnrelu_bwd_dx_kernel<true>, dim3(g3), dim3(kBnThreads), 0, s, p);
  A3VT_CHECK_LAUNCH();
  A3VT_LAUNCH(bnrelu_colsum_final_kernel, dim3(1), dim3(kBnThreads), 0, s, (const float *)p.partial, g3, c, dx_colsum);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt

namespace a3vt {
int launch_cast_weights(const CastBatch &b, hipStream_t s) {
  const long long total = b.start[b.n];
  if (total == 0) return 0;
  A3VT_LAUNCH(cast_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, b);
  A3VT_CHECK_LAUNCH();
  return 0;
}
}  // namespace a3vt
