// Developer micro-benchmark: raw fp32 MFMA issue rate on gfx950 (no memory traffic).
// hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(512) void k16(float *out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void k32(float *out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Same MFMA order as rowgemm's inner loop: 2 m-tiles x NT n-tiles, pairs of n-tiles, distinct operand registers.
template <int NT>
__global__ __launch_bounds__(512, 2) void kgemm(float *out, const float *in, int iters) {
  f32x4 acc[2][NT];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  f32x4 af[2], bf[4];
  for (int i = 0; i < 2; ++i) af[i] = *(const f32x4 *)(in + threadIdx.x * 4 + i * 4096);
  for (int i = 0; i < 4; ++i) bf[i] = *(const f32x4 *)(in + threadIdx.x * 4 + 8192 + i * 4096);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int jp = 0; jp < (NT + 1) / 2; ++jp) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          acc[i][2 * jp] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[(jp & 1) * 2][s], acc[i][2 * jp], 0, 0, 0);
          if (2 * jp + 1 < NT)
            acc[i][2 * jp + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[(jp & 1) * 2 + 1][s], acc[i][2 * jp + 1], 0, 0, 0);
        }
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
void run(const char *name, F launch, double flop_per_thread_iter_wave, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  double waves = 256.0 * threads / 64;
  printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, waves * iters * flop_per_thread_iter_wave / ms / 1e9);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  // flop per wave per iteration: 4 * NACC MFMAs * 2048 (16x16x4) or 4096 (32x32x2)
  run("16x16x4 nacc=4 1w/simd", [&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 4 * 2048.0, 256, iters);
  run("16x16x4 nacc=4 2w/simd", [&] { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(512), 0, 0, out, iters, 1.f, 2.f); }, 4 * 4 * 2048.0, 512, iters);
  run("16x16x4 nacc=1 1w/simd", [&] { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 1 * 2048.0, 256, iters);
  run("16x16x4 nacc=1 2w/simd", [&] { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(512), 0, 0, out, iters, 1.f, 2.f); }, 4 * 1 * 2048.0, 512, iters);
  run("16x16x4 nacc=2 1w/simd", [&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 2 * 2048.0, 256, iters);
  run("32x32x2 nacc=2 1w/simd", [&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 2 * 4096.0, 256, iters);
  run("32x32x2 nacc=2 2w/simd", [&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(512), 0, 0, out, iters, 1.f, 2.f); }, 4 * 2 * 4096.0, 512, iters);
  run("32x32x2 nacc=1 1w/simd", [&] { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4 * 1 * 4096.0, 256, iters);
  float *in;
  hipMalloc(&in, 65536 * sizeof(float));
  hipMemset(in, 0, 65536 * sizeof(float));
  run("kgemm NT=19 2w/simd", [&] { hipLaunchKernelGGL(kgemm<19>, dim3(256), dim3(512), 0, 0, out, in, 500); }, 2 * 19 * 4 * 2048.0, 512, 500);
  run("kgemm NT=19 1w/simd", [&] { hipLaunchKernelGGL(kgemm<19>, dim3(256), dim3(256), 0, 0, out, in, 500); }, 2 * 19 * 4 * 2048.0, 256, 500);
  run("kgemm NT=4 2w/simd", [&] { hipLaunchKernelGGL(kgemm<4>, dim3(256), dim3(512), 0, 0, out, in, 2000); }, 2 * 4 * 4 * 2048.0, 512, 2000);
  return 0;
}
