// Developer micro-benchmark: SUSTAINED fp32 MFMA rate on gfx950 with random operand data (power draw depends on the bits
// that toggle), for the two fp32 instruction shapes, over a few seconds each — is the chip clock-limited by power, and does
// the 32x32x2 shape (half the operand reads per flop) sustain more than 16x16x4?
// hipcc -O3 --offload-arch=gfx950 mfma_sustained.hip -o mfma_sustained && ./mfma_sustained [seconds per case]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(512) void k16(float *out, const float *in, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x * 8 + i], b[i] = in[4096 + threadIdx.x * 8 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float *out, const float *in, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x * 8 + i], b[i] = in[4096 + threadIdx.x * 8 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[(r + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
void run(const char *name, F launch, double flop_per_launch, double seconds) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  // first second vs the rest: the clock sags as the chip heats / the power controller settles
  for (int phase = 0; phase < 2; ++phase) {
    const double budget = phase == 0 ? 0.5 : seconds;
    int n = 0;
    float ms = 0, total = 0;
    while (total < budget * 1e3) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      total += ms;
      n += 10;
    }
    printf("%-26s %s %6.2f s  %7.1f TFLOP/s\n", name, phase == 0 ? "first" : "then ", total / 1e3, n * flop_per_launch / total / 1e9);
  }
}

int main(int argc, char **argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  float *out, *in;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&in, 8192 * sizeof(float));
  float h[8192];
  srand(1);
  for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 4000;
  const double waves = 256.0 * 512 / 64;
  run("16x16x4 8 acc tiles 2w/simd", [&] { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, out, in, iters); }, waves * iters * 8 * 8 * 2048.0, secs);
  run("32x32x2 2 acc tiles 2w/simd", [&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(512), 0, 0, out, in, iters); }, waves * iters * 8 * 2 * 4096.0, secs);
  run("16x16x4 8 acc tiles 2w/simd", [&] { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, out, in, iters); }, waves * iters * 8 * 8 * 2048.0, secs);
  run("32x32x2 2 acc tiles 2w/simd", [&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(512), 0, 0, out, in, iters); }, waves * iters * 8 * 2 * 4096.0, secs);
  return 0;
}
