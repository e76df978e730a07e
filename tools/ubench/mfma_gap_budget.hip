// Developer micro-benchmark (round 6): with ONE wave per SIMD, how many other instructions fit between two back-to-back fp32
// MFMAs (32 cycles of matrix pipe each) before the pipe idles?  Per kind: k v_add_f32 / k s_add_u32 / k v_mul_lo_u32 (quarter
// rate) / one ds_read_b128 / one global_store_dwordx4 between consecutive MFMAs on 5 accumulators, 300 operand registers.
// hipcc -O3 --offload-arch=gfx950 mfma_gap_budget.hip -o mfma_gap_budget && ./mfma_gap_budget
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int KIND, int K>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
  constexpr int NACC = 5, NREG = 100;
  __shared__ f32x4 lds[256];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[NREG], b[4];
  for (int i = 0; i < NREG; ++i) a[i] = in[(threadIdx.x * 7 + i * 13) & 4095];
  for (int i = 0; i < 4; ++i) b[i] = in[4096 + ((threadIdx.x + i) & 1023)];
  lds[threadIdx.x] = f32x4{b[0], b[1], b[2], b[3]};
  float v[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) v[i] = in[i], u[i] = threadIdx.x + i;
  unsigned sacc = iters;
  f32x4 l = {0, 0, 0, 0};
  float *dst = out + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
      acc[r % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b[r & 3], acc[r % NACC], 0, 0, 0);
      if (KIND == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(b[0]));
      } else if (KIND == 1) {
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
      } else if (KIND == 2) {
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j]) : "v"(u[7]));
      } else if (KIND == 3) {
        if (K > 0 && r % K == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((unsigned)(threadIdx.x * 16)) : "memory");
      } else if (KIND == 4) {
        if (K > 0 && r % K == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(acc[(r + 2) % NACC]) : "memory");
      } else if (KIND == 5) {
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(b[1]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s = l[0] + sacc;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += v[i] + u[i];
  if (s == 1.2345e-33f) dst[0] = s;
}

template <int KIND, int K>
void run(const char *name, float *out, const float *in) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, in, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, K>), dim3(256), dim3(256), 0, 0, out, in, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s k=%d  %6.1f ns per MFMA (13.3 = 32 cycles at 2.4 GHz)\n", name, K, ms * 1e6 / (iters * 100.0));
}

int main() {
  float *out, *in;
  (void)hipMalloc(&out, 256 * 256 * 4 * sizeof(float));
  (void)hipMalloc(&in, 8192 * sizeof(float));
  float h[8192];
  for (int i = 0; i < 8192; ++i) h[i] = 0.5f + (i % 97) * 1e-3f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<0, 0>("nothing between", out, in);
  run<0, 2>("v_add_f32", out, in); run<0, 4>("v_add_f32", out, in); run<0, 5>("v_add_f32", out, in); run<0, 6>("v_add_f32", out, in); run<0, 7>("v_add_f32", out, in); run<0, 8>("v_add_f32", out, in);
  run<1, 2>("s_add_u32", out, in); run<1, 4>("s_add_u32", out, in); run<1, 6>("s_add_u32", out, in); run<1, 8>("s_add_u32", out, in);
  run<2, 1>("v_mul_lo_u32", out, in); run<2, 2>("v_mul_lo_u32", out, in); run<2, 3>("v_mul_lo_u32", out, in);
  run<5, 4>("v_cndmask_b32 (vcc)", out, in); run<5, 6>("v_cndmask_b32 (vcc)", out, in);
  run<3, 1>("ds_read_b128 every k-th MFMA", out, in); run<3, 2>("ds_read_b128 every k-th MFMA", out, in); run<3, 5>("ds_read_b128 every k-th MFMA", out, in);
  run<4, 1>("global_store_dwordx4 every k-th MFMA", out, in); run<4, 2>("global_store_dwordx4 every k-th MFMA", out, in); run<4, 5>("global_store_dwordx4 every k-th MFMA", out, in);
  return 0;
}
