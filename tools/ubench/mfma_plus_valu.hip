// Developer micro-benchmark (round 6): do fp32 MFMAs and fp32 VALU FMAs of two waves on the SAME SIMD add up?  gfx950's fp32
// matrix peak equals its fp32 vector peak (157.3 TFLOP/s, MI355X_MICROARCH.md) — if the two are separate execution resources a
// product kernel could run part of its rows on the vector pipe beside the matrix pipe; if they share the multipliers it
// cannot.  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x4_f32 back to back on 8
// independent accumulators, waves 4-7 issue v_fma_f32 back to back on 32 independent accumulators.
// hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize mfma_plus_valu.hip -o mfma_plus_valu && ./mfma_plus_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(512) void both(float *out, const float *in, int iters, int mode) {
  const int role = threadIdx.x >> 8;   // 0: matrix waves, 1: vector waves
  float s = 0.f;
  if (role == 0) {
    if (mode == 2) return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x * 8 + i], b[i] = in[4096 + threadIdx.x * 8 + i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    if (mode == 1) return;
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = in[(threadIdx.x + i * 7) & 4095];
    const float m = in[5000 + (threadIdx.x & 63)], c = in[6000 + (threadIdx.x & 63)];
    // the same number of loop iterations; per iteration 32 MFMAs x 32 cycles = 1024 cycles of matrix pipe on the sibling wave;
    // here VPI v_fma_f32 per iteration (wave64 fp32 fma = 2 cycles at full rate -> 512 per iteration would fill the same time)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = __builtin_fmaf(acc[i], m, c);
    }
    for (int i = 0; i < 32; ++i) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&in, 8192 * sizeof(float));
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = 0.5f + (i % 97) * 1e-3f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const char *names[3] = {"both", "matrix waves only", "vector waves only"};
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(both, dim3(256), dim3(512), 0, 0, out, in, 100, mode);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(both, dim3(256), dim3(512), 0, 0, out, in, iters, mode);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double mf = mode == 2 ? 0 : 256.0 * 4 * iters * 32 * 2048, vf = mode == 1 ? 0 : 256.0 * 4 * iters * 512 * 128;
      printf("%-18s %8.2f ms   matrix %6.1f TFLOP/s   vector %6.1f TFLOP/s   sum %6.1f\n", names[mode], ms, mf / ms / 1e9, vf / ms / 1e9,
             (mf + vf) / ms / 1e9);
    }
  return 0;
}
