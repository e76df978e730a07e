// adam.hip — the trainer's optimizer step over ALL parameter tensors in one launch.
//
// Reference: `optim.Adam(params, lr=args.lr, weight_decay=0)` + `self.optimizer.step()`,
// pterotactyl/reconstruction/vision/train.py:64,148 (torch's Adam: amsgrad off, maximize off).  Per element, in torch's own
// order of operations (`_single_tensor_adam`):
//   g' = g + weight_decay * p                           (weight_decay != 0 only)
//   m  = m + (g' - m) * (1 - beta1)                     (Tensor.lerp_, weight < 0.5)
//   v  = v * beta2 + ((1 - beta2) * g') * g'            (mul_ + addcmul_)
//   p  = p - step_size * (m / (sqrt(v) / sqrt(1 - beta2^t) + eps)),   step_size = lr / (1 - beta1^t)
// step_size and sqrt(1 - beta2^t) come from the host (computed in double, rounded once).
//
// HBM-bound: 16 bytes read and 12 written per parameter.  torch's fused multi-tensor kernel takes 7 launches of 104 us for the 47 M
// parameters of the image model (1.8 TB/s); here the tensors are cut into chunks of 4096 elements once, when the optimizer is
// built (a table of {tensor, offset} in device memory), and one grid of workgroups walks the chunk list with 16-byte accesses.
#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kAdamChunk = 4096;      // elements per chunk: 256 threads x 4 pieces of 16 bytes
constexpr int kAdamMaxWgs = 2048;

struct AdamArgs {
  float *const *param;         // [tensors]
  const float *const *grad;    // [tensors]
  float *const *exp_avg;       // [tensors]
  float *const *exp_avg_sq;    // [tensors]
  const long long *numel;      // [tensors]
  const int *chunk_tensor;     // [chunks]
  const long long *chunk_off;  // [chunks]: first element of the chunk in its tensor (a multiple of 4096)
  int n_chunks;
  float one_minus_beta1, beta2, one_minus_beta2, step_size, bc2_sqrt, eps, weight_decay;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamArgs &a) {
  if (a.weight_decay != 0.f) g = g + a.weight_decay * p;
  m = m + (g - m) * a.one_minus_beta1;
  v = v * a.beta2 + (a.one_minus_beta2 * g) * g;
  const float denom = __fsqrt_rn(v) / a.bc2_sqrt + a.eps;
  p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  for (int c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
    const int t = a.chunk_tensor[c];
    const long long off = a.chunk_off[c], n = a.numel[t];
    float *__restrict__ p = a.param[t] + off;
    const float *__restrict__ g = a.grad[t] + off;
    float *__restrict__ m = a.exp_avg[t] + off;
    float *__restrict__ v = a.exp_avg_sq[t] + off;
    const long long left = n - off;      // > 0
    const bool aligned = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
    if (aligned && left >= kAdamChunk) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = (k * 256 + threadIdx.x) * 4;
        f32x4 pv = *reinterpret_cast<const f32x4 *>(p + i), mv = *reinterpret_cast<const f32x4 *>(m + i);
        f32x4 vv = *reinterpret_cast<const f32x4 *>(v + i);
        const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(g + i));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = pv[e], me = mv[e], ve = vv[e];
          adam_one(pe, gv[e], me, ve, a);
          pv[e] = pe;
          mv[e] = me;
          vv[e] = ve;
        }
        *reinterpret_cast<f32x4 *>(p + i) = pv;
        *reinterpret_cast<f32x4 *>(m + i) = mv;
        *reinterpret_cast<f32x4 *>(v + i) = vv;
      }
    } else {      // a tensor's last chunk, or a tensor that does not start on a 16-byte boundary
      const int cnt = left < kAdamChunk ? (int)left : kAdamChunk;
      for (int i = threadIdx.x; i < cnt; i += 256) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_one(pv, g[i], mv, vv, a);
        p[i] = pv;
        m[i] = mv;
        v[i] = vv;
      }
    }
  }
}

}  // namespace

int adam_chunk_elems() { return kAdamChunk; }

int launch_adam(void *const *param, const void *const *grad, void *const *exp_avg, void *const *exp_avg_sq,
                const long long *numel, const int *chunk_tensor, const long long *chunk_off, int n_chunks, double lr, double beta1,
                double beta2, double eps, double weight_decay, long long step, hipStream_t s) {
  if (n_chunks == 0) return 0;
  AdamArgs a{};
  a.param = reinterpret_cast<float *const *>(param);
  a.grad = reinterpret_cast<const float *const *>(grad);
  a.exp_avg = reinterpret_cast<float *const *>(exp_avg);
  a.exp_avg_sq = reinterpret_cast<float *const *>(exp_avg_sq);
  a.numel = numel;
  a.chunk_tensor = chunk_tensor;
  a.chunk_off = chunk_off;
  a.n_chunks = n_chunks;
  // torch forms these in double (Python floats) and rounds each once to the kernel's fp32
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.one_minus_beta1 = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.one_minus_beta2 = (float)(1.0 - beta2);
  a.step_size = (float)(lr / bc1);
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  a.weight_decay = (float)weight_decay;
  const int grid = n_chunks < kAdamMaxWgs ? n_chunks : kAdamMaxWgs;
  A3VT_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
