// Developer micro-benchmark: what does one vector-memory instruction cost a wave that is otherwise issuing fp32 MFMAs
// back to back?  Skeleton of rowgemm_kernel's K loop (8 waves per workgroup, one workgroup per CU, one s_barrier and one
// counted vmcnt wait per "chunk" of 152 MFMAs, P loads spread through the chunk) with the kind of load as the variable:
//   kind 0  none
//   kind 1  global_load_lds_dwordx4 (LDS-DMA, 1 KiB per wave-instruction), source streams from HBM (the A pieces)
//   kind 2  the same, source L2-resident (the Bt pieces)
//   kind 3  global_load_dwordx4 into registers, MFMA fragment mapping (lane & 15 = row, lane >> 4 = k-quad), HBM stream
//   kind 4  global_load_dwordx4 into registers, DMA mapping (lane >> 2 = row, lane & 3 = k-quad), HBM stream
// hipcc -O3 --offload-arch=gfx950 mfma_with_loads.hip -o mfma_with_loads && ./mfma_with_loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ void glds16(const float *gsrc, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int LD = 304;   // floats per source row (like an activation row)

template <int KIND, int P>
__global__ __launch_bounds__(512, 2) void k(float *out, const float *in, const float *src, size_t src_rows, int chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) a[i] = in[threadIdx.x * 8 + i], b[i] = in[4096 + threadIdx.x * 8 + i];
  // this wave's 32 source rows: a private stripe of the big buffer (HBM kinds) or of the first 320 rows (L2 kind)
  const size_t stripe = KIND == 2 ? (size_t)(wave * 32) : ((size_t)(blockIdx.x * 8 + wave) * 32) % 65536;
  const int row_dma = lane >> 2, quad_dma = lane & 3;
  const int row_frag = lane & 15, quad_frag = lane >> 4;
  const bool frag = KIND == 3;
  const float *rowp0 = src + (stripe + (frag ? row_frag : row_dma)) * LD + (frag ? quad_frag : quad_dma) * 4;
  const float *rowp1 = rowp0 + 16 * LD;
  float *slot = lds + wave * (3 * P * 256);
  f32x4 ring[3][P > 0 ? P : 1];
  for (int s = 0; s < 3; ++s)
    for (int i = 0; i < (P > 0 ? P : 1); ++i) ring[s][i] = f32x4{0, 0, 0, 0};

  auto piece = [&](int chunk, int st, int pc) {
    const int kk = (chunk % 19) * 16;
    const size_t round_off = KIND == 2 ? 0 : (size_t)((chunk / 19) % 2) * 65536 * LD;   // a new stripe per "round": an HBM stream
    const float *g = ((pc & 1) ? rowp1 : rowp0) + round_off + kk;
    if (KIND == 1 || KIND == 2) glds16(g, slot + (st * P + pc) * 256);
    if (KIND == 3 || KIND == 4) ring[st][pc] = *reinterpret_cast<const f32x4 *>(g);
  };
  if (KIND != 0) {
#pragma unroll
    for (int pc = 0; pc < P; ++pc) piece(0, 0, pc);
#pragma unroll
    for (int pc = 0; pc < P; ++pc) piece(1, 1, pc);
  }
  for (int t0 = 0; t0 < chunks; t0 += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int t = t0 + u;
      if (KIND != 0) wait_vmcnt<P>();   // chunk t landed, chunk t+1 may be in flight
      __builtin_amdgcn_s_barrier();
      float fold = 0.f;
      if (KIND == 3 || KIND == 4) {
#pragma unroll
        for (int pc = 0; pc < P; ++pc) fold += ring[u][pc][0] + ring[u][pc][3];
        fold *= 1e-30f;
      }
#pragma unroll
      for (int g = 0; g < 10; ++g) {
#pragma unroll
        for (int r = 0; r < (g < 9 ? 2 : 1); ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(2 * g + r) & 7] + fold, b[(r + i) & 7], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (KIND != 0 && (g & 1) && (g >> 1) < P) {
          piece(t + 2, (u + 2) % 3, g >> 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  wait_vmcnt<0>();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (KIND == 1 || KIND == 2) s += slot[lane];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double run(F launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3;
}

template <int KIND, int P>
void bench(const char *name, float *out, const float *in, const float *src, size_t rows) {
  const int chunks = 48;   // ~2.5 rounds x 19 chunks, like one rowgemm launch
  const size_t shmem = 8 * 3 * (P > 0 ? P : 1) * 256 * sizeof(float);
  (void)hipFuncSetAttribute((const void *)k<KIND, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  const double us = run([&] { hipLaunchKernelGGL((k<KIND, P>), dim3(256), dim3(512), shmem, 0, out, in, src, rows, chunks); });
  const double mfma = 256.0 * 8 * chunks * 152;
  printf("%-44s P=%d  %8.1f us  %6.1f TFLOP/s  (%.0f cycles per chunk per SIMD at 2.4 GHz)\n", name, P, us, mfma * 2048 / us / 1e6,
         us * 2400 / chunks);
}

int main() {
  float *out, *in, *src;
  const size_t rows = 163968;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  (void)hipMalloc(&in, 8192 * sizeof(float));
  (void)hipMalloc(&src, rows * LD * sizeof(float));
  (void)hipMemset(src, 0, rows * LD * sizeof(float));
  float h[8192];
  srand(1);
  for (int i = 0; i < 8192; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    bench<0, 0>("no loads", out, in, src, rows);
    bench<1, 2>("LDS-DMA from HBM stream", out, in, src, rows);
    bench<1, 5>("LDS-DMA from HBM stream", out, in, src, rows);
    bench<2, 3>("LDS-DMA from L2-resident rows", out, in, src, rows);
    bench<2, 5>("LDS-DMA from L2-resident rows", out, in, src, rows);
    bench<3, 2>("dwordx4 -> VGPR, fragment mapping", out, in, src, rows);
    bench<3, 5>("dwordx4 -> VGPR, fragment mapping", out, in, src, rows);
    bench<4, 2>("dwordx4 -> VGPR, row-contiguous mapping", out, in, src, rows);
    bench<4, 5>("dwordx4 -> VGPR, row-contiguous mapping", out, in, src, rows);
  }
  return 0;
}
