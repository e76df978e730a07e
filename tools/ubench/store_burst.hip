// Developer micro-benchmark (round 3): what does a synchronised store burst cost, as a function of its size?
//
// rowgemm's epilogue writes 307 KB per CU per round, all 256 CUs at the same moment, and that burst is not overlapped
// with anything.  Model under test: an XCD's 4 MiB L2 absorbs a burst up to some size at full speed and the excess
// drains at the fabric / HBM write rate.  Here every workgroup (one per CU, 512 threads) alternates
//   compute : NMFMA fp32 MFMAs per wave (2 waves per SIMD), optionally beside a streaming read of RD bytes per CU
//   burst   : S bytes per CU of 16-byte stores to rows of 1200 B (its own region), then s_waitcnt vmcnt(0)
// and lane 0 of wave 0 stamps s_memrealtime (100 MHz) at the phase boundaries.  SHIFT = 1 delays the odd workgroups of
// every XCD (blockIdx >> 3 odd) by half a compute phase, so the two halves of an XCD burst at different times.
// hipcc -O3 --offload-arch=gfx950 store_burst.hip -o store_burst && ./store_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(512, 2) void k(float *out, const float *rd, size_t region_floats, int s_bytes, int rd_bytes,
                                            int nmfma, int rounds, int shift, unsigned long long *stamps) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float a = threadIdx.x * 1e-3f, b = 1.0f + lane * 1e-3f;
  float *mine = out + (size_t)blockIdx.x * region_floats;
  const float *myrd = rd + (size_t)blockIdx.x * region_floats;
  f32x4 sink = {0, 0, 0, 0};
  const bool late = shift && ((blockIdx.x >> 3) & 1);
  auto compute = [&](int n) {
    // optional read stream, issued up front (in flight under the MFMAs)
    f32x4 r[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const int nrd = rd_bytes / (512 * 16);
    int ri = 0;
    for (int i = 0; i < n; i += 4) {
      if (ri < nrd) {
        r[ri & 3] = *reinterpret_cast<const f32x4 *>(myrd + ((size_t)ri * 512 + threadIdx.x) * 4);
        ++ri;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    sink += r[0] + r[1] + r[2] + r[3];
  };
  if (late) compute(nmfma / 2);
  for (int rnd = 0; rnd < rounds; ++rnd) {
    compute(nmfma);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    // burst: S bytes as 16-byte pieces, 75 per 1200-B row, this round's rows after the previous round's
    const int pieces = s_bytes / 16;
    const f32x4 v = acc[0] + sink;
    float *dst = mine + (size_t)rnd * (s_bytes / 4);
    for (int p = threadIdx.x; p < pieces; p += 512) *reinterpret_cast<f32x4 *>(dst + (size_t)p * 4) = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      stamps[((size_t)blockIdx.x * rounds + rnd) * 2] = t0;
      stamps[((size_t)blockIdx.x * rounds + rnd) * 2 + 1] = t1;
    }
  }
  if (acc[1][0] == 1.2345e-33f) out[0] = acc[1][1] + acc[2][0] + acc[3][0];
}

int main(int argc, char **argv) {
  const int rounds = 6;
  const size_t region = (size_t)rounds * 640 * 1024 / 4;   // floats per workgroup region (>= rounds * max S)
  float *out, *rd;
  unsigned long long *stamps;
  hipMalloc(&out, 256 * region * 4);
  hipMalloc(&rd, 256 * region * 4);
  hipMemset(out, 0, 256 * region * 4);
  hipMemset(rd, 0, 256 * region * 4);
  hipMalloc(&stamps, 256 * rounds * 2 * 8);
  std::vector<unsigned long long> h(256 * rounds * 2);
  const int nmfma = argc > 1 ? atoi(argv[1]) : 1500;   // per wave: 2 waves x 1500 x 32 cycles = 96 k cycles = 40 us
  printf("grid shift rd_KB S_KB : kernel_us  burst_us(median over WGs, rounds 1..)  -> burst GB/s per CU, TB/s chip\n");
  for (int grid : {256, 192, 128})
    for (int shift : {0, 1})
      for (int rdkb : {0, 300})
        for (int skb : {32, 64, 96, 128, 160, 192, 256, 307, 384, 512}) {
          if (grid != 256 && (shift || rdkb)) continue;
          hipEvent_t e0, e1;
          hipEventCreate(&e0);
          hipEventCreate(&e1);
          for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, rd, region, skb * 1024, rdkb * 1024, nmfma, rounds, shift,
                               stamps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
          }
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          hipMemcpy(h.data(), stamps, grid * rounds * 2 * 8, hipMemcpyDeviceToHost);
          std::vector<double> d;
          for (int w = 0; w < grid; ++w)
            for (int r = 1; r < rounds; ++r) d.push_back((h[(w * rounds + r) * 2 + 1] - h[(w * rounds + r) * 2]) * 0.01);
          std::sort(d.begin(), d.end());
          const double med = d[d.size() / 2], p90 = d[d.size() * 9 / 10];
          printf("%3d %d %3d %3d : %7.1f  %6.2f (p90 %6.2f)  -> %.1f GB/s per CU, %.2f TB/s\n", grid, shift, rdkb, skb,
                 ms * 1e3, med, p90, skb * 1024.0 / med * 1e-3, skb * 1024.0 * grid / med * 1e-6);
          fflush(stdout);
        }
  return 0;
}
