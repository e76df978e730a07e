// Developer micro-benchmark (round 3): does the fp32 MFMA SHAPE change what an LDS-fed GEMM inner loop sustains?
//
// rowgemm / dw run v_mfma_f32_16x16x4_f32 with one ds_read_b32 of the B operand per MFMA (16 rows per wave, 19 column
// tiles).  v_mfma_f32_32x32x2_f32 does the same FLOP per cycle on paper with half the LDS operand bytes per FLOP (32 rows
// per wave, 10 column tiles of 32), and the chip lowers its clock under load by an amount that depends on the energy per
// MFMA (MI355X_MICROARCH.md, DVFS give-back 4 and 7) — so the question is wall time on random operands, not cycles.
// Both variants: one 512-thread workgroup per CU (2 waves per SIMD), operands re-read from a 64 KB LDS image of random
// floats every k-step exactly as the product kernels do (A fragment once per k-step, B fragment per MFMA), accumulators of
// a full 304-column row block live in registers.  Reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 mfma_shape.hip -o mfma_shape && ./mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kLdsFloats = 16384;   // 64 KB

template <int SHAPE>   // 0: 16x16x4 (19 tiles), 1: 32x32x2 (10 tiles)
__global__ __launch_bounds__(512) void k(const float *__restrict__ src, float *__restrict__ out, int ksteps,
                                         unsigned long long *__restrict__ stamps) {
  __shared__ float lds[kLdsFloats];
  for (int i = threadIdx.x; i < kLdsFloats; i += 512) lds[i] = src[(size_t)blockIdx.x * kLdsFloats + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 0) {
    f32x4 acc[19];
#pragma unroll
    for (int j = 0; j < 19; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a, b[19];   // fragments of the NEXT k-step are read while this one's MFMAs issue (as the product kernels do)
    auto rd = [&](int ks, float &a_, float (&b_)[19]) {
      const float *p = lds + (ks & 7) * 2048 + lane;   // one address register, immediate offsets (max 1279 + 640)
      a_ = p[wave * 80];
#pragma unroll
      for (int j = 0; j < 19; ++j) b_[j] = p[64 + j * 64];
    };
    rd(0, a, b);
    for (int ks = 0; ks < ksteps; ++ks) {
      float an, bn[19];
      rd(ks + 1, an, bn);
#pragma unroll
      for (int j = 0; j < 19; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], acc[j], 0, 0, 0);
      a = an;
#pragma unroll
      for (int j = 0; j < 19; ++j) b[j] = bn[j];
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int j = 1; j < 19; ++j) s += acc[j];
    if (s[0] + s[1] + s[2] + s[3] == 1.2345e-33f) out[threadIdx.x] = s[0];
  } else {
    f32x16 acc[10];
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[j][t] = 0.f;
    float a, b[10];
    auto rd = [&](int ks, float &a_, float (&b_)[10]) {
      const float *p = lds + (ks & 7) * 2048 + lane;
      a_ = p[wave * 80];
#pragma unroll
      for (int j = 0; j < 10; ++j) b_[j] = p[64 + j * 64];
    };
    rd(0, a, b);
    for (int ks = 0; ks < ksteps; ++ks) {
      float an, bn[10];
      rd(ks + 1, an, bn);
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
      a = an;
#pragma unroll
      for (int j = 0; j < 10; ++j) b[j] = bn[j];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j)
#pragma unroll
      for (int t = 0; t < 16; ++t) s += acc[j][t];
    if (s == 1.2345e-33f) out[threadIdx.x] = s;
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = c1 - c0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE>
void run(const char *name, const float *src, float *out, unsigned long long *stamps, int ksteps, double flop_per_wave_step) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, src, out, ksteps, stamps);   // ~2 s warm
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, src, out, ksteps, stamps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), stamps, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < 256; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);   // GHz (realtime = 100 MHz)
  std::sort(clk.begin(), clk.end());
  const double flop = flop_per_wave_step * ksteps * 8 * 256;
  printf("%-28s %8.3f ms/launch  %7.1f TFLOP/s   in-kernel clock median %.3f GHz (p10 %.3f p90 %.3f)\n", name, ms / reps,
         flop / (ms / reps * 1e-3) * 1e-12, clk[128], clk[25], clk[230]);
}

int main() {
  float *src, *out;
  unsigned long long *stamps;
  const size_t n = (size_t)256 * kLdsFloats;
  std::vector<float> h(n);
  srand(1);
  for (auto &x : h) x = (float)rand() / RAND_MAX - 0.5f;
  hipMalloc(&src, n * 4);
  hipMalloc(&out, 4096);
  hipMalloc(&stamps, 512 * 8);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("16x16x4 f32, 19 tiles/wave", src, out, stamps, 120000, 19 * 2048.0);
    run<1>("32x32x2 f32, 10 tiles/wave", src, out, stamps, 114000, 10 * 4096.0);
  }
  return 0;
}
