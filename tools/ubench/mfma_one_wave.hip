// Developer micro-benchmark (round 6): fp32 MFMA issue rate of ONE wave per SIMD (256-thread workgroups, one per CU) against two,
// with 5 or 8 independent accumulators, and with the a-operand taken from many different registers (a register-resident
// weight image) or from a few.  rowgemmw_kernel (gcn_gemmw.hip) measured 39.6 cycles per MFMA in its K loop: is that the wave?
// hipcc -O3 --offload-arch=gfx950 mfma_one_wave.hip -o mfma_one_wave && ./mfma_one_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int THREADS, int NACC, int NREG>
__global__ __launch_bounds__(THREADS) void k(float *out, const float *in, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[NREG], b[4];
  for (int i = 0; i < NREG; ++i) a[i] = in[(threadIdx.x * 7 + i * 13) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = in[4096 + ((threadIdx.x + i) & 1023)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NREG / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r * NACC + i], b[r & 3], acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int THREADS, int NACC, int NREG>
void run(const char *name, float *out, const float *in) {
  const int iters = 400000 / (NREG / NACC * NACC) * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<THREADS, NACC, NREG>), dim3(256), dim3(THREADS), 0, 0, out, in, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<THREADS, NACC, NREG>), dim3(256), dim3(THREADS), 0, 0, out, in, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * (NREG / NACC * NACC);   // per wave
  const double flop = mfmas * 2048 * (THREADS / 64) * 256;
  printf("%-44s %7.2f ms  %6.1f TFLOP/s   %5.1f ns per MFMA per SIMD\n", name, ms, flop / ms / 1e9, ms * 1e6 / (mfmas * (THREADS / 256)));
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&in, 8192 * sizeof(float));
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = 0.5f + (i % 97) * 1e-3f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<512, 8, 8>("2 waves/SIMD, 8 acc, 8 a-registers", out, in);
  run<256, 8, 8>("1 wave/SIMD,  8 acc, 8 a-registers", out, in);
  run<256, 5, 5>("1 wave/SIMD,  5 acc, 5 a-registers", out, in);
  run<256, 5, 300>("1 wave/SIMD,  5 acc, 300 a-registers", out, in);
  run<256, 10, 300>("1 wave/SIMD, 10 acc, 300 a-registers", out, in);
  run<256, 20, 300>("1 wave/SIMD, 20 acc, 300 a-registers", out, in);
  return 0;
}
