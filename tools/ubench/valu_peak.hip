// Developer micro-benchmark: fp32 VALU issue rates on gfx950 (packed vs scalar fma/add/min), 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int MODE>
__global__ void k(float *out, int iters, float a, float b) {
  f32x2 x[8];
  for (int i = 0; i < 8; ++i) x[i] = f32x2{a + i + threadIdx.x, b - i};
  f32x2 m = {a, b}, c = {b, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) x[i] = __builtin_elementwise_fma(x[i], m, c);                       // v_pk_fma_f32
      if (MODE == 1) { x[i][0] = fmaf(x[i][0], m[0], c[0]); x[i][1] = fmaf(x[i][1], m[1], c[0]); }  // 2 x v_fma_f32
      if (MODE == 2) x[i] = x[i] + m;                                                     // v_pk_add_f32
      if (MODE == 3) x[i][0] = fminf(fminf(x[i][0], x[i][1]), m[0] + it);                 // v_min3
      if (MODE == 4) x[i] = x[i] * m;                                                     // v_pk_mul_f32
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int wps, float *out, double ops_per_instr_group) {
  const int iters = 20000, threads = 256;
  dim3 grid(256 * wps);  // wps workgroups of 4 waves per CU -> wps waves per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, grid, dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = wps, groups = (double)iters * 8;   // instruction groups per wave
  const double cyc = ms * 1e-3 * 2.4e9 / (groups * waves_per_simd);
  printf("%-22s wps=%d  %7.3f ms  %.2f cycles per group per SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
}

int main() {
  float *out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int wps : {1, 2, 4, 8}) {
    run<0>("v_pk_fma_f32", wps, out, 1);
    run<1>("2 x v_fma_f32", wps, out, 1);
    run<2>("v_pk_add_f32", wps, out, 1);
    run<4>("v_pk_mul_f32", wps, out, 1);
    run<3>("min+min(+add)", wps, out, 1);
  }
  return 0;
}
