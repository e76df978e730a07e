// sample.hip — area-weighted surface sampling of a batch of fixed-topology meshes (gfx950).
//
// Replaces batch_sample, utility/utils.py:152-187:
//   areas (PyTorch3D mesh_face_areas_normals, :164) -> NaN scrub / normalise (:165-168)
//   -> multinomial(num, replacement=True) (:170) -> _rand_barycentric_coords (:179) -> gather + blend (:175-185)
// The reference draws with torch's CPU/CUDA generators, which cannot be reproduced bit-for-bit on
// another device; parity mode therefore takes the (face, u, v) triples as input, and the production
// mode draws them from Philox4x32-10 through an inverse-CDF search (same distribution, tested
// statistically).  Backward = scatter-add of w_k * grad into the three corner vertices (autograd of
// the gather at :175,182-184).
#include "common.h"
#include "kernels.h"

namespace a3vt {

__device__ __forceinline__ float tri_area(const float *__restrict__ vb, const int32_t *__restrict__ f) {
  const float *a = vb + 3 * (long long)f[0], *b = vb + 3 * (long long)f[1], *c = vb + 3 * (long long)f[2];
  const float e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
  const float e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
  const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
  return 0.5f * sqrtf(nx * nx + ny * ny + nz * nz);
}

// One workgroup per mesh: total area, then inclusive scan of p_f = |a_f / total| (NaN -> 1).
__global__ __launch_bounds__(256) void face_cdf_kernel(const float *__restrict__ verts,
                                                       const int32_t *__restrict__ faces, int n_vert, int n_faces,
                                                       float *__restrict__ cdf) {
  __shared__ float part[256];
  __shared__ float total_s;
  const int b = blockIdx.x, t = threadIdx.x;
  const float *vb = verts + (long long)b * n_vert * 3;
  const int per = (n_faces + 255) / 256;
  const int f0 = t * per, f1 = min(f0 + per, n_faces);
  float s = 0.f;
  for (int f = f0; f < f1; ++f) {
    float a = tri_area(vb, faces + 3 * f);
    if (a != a) a = 0.f;  // utils.py:166
    s += a;
  }
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    float tot = 0.f;
    for (int i = 0; i < 256; ++i) tot += part[i];
    total_s = tot;
  }
  __syncthreads();
  const float total = total_s;
  s = 0.f;
  for (int f = f0; f < f1; ++f) {
    float a = tri_area(vb, faces + 3 * f);
    if (a != a) a = 0.f;
    float p = fabsf(a / total);  // utils.py:167
    if (p != p) p = 1.f;         // utils.py:168 (all-zero-area mesh -> uniform)
    s += p;
  }
  __syncthreads();
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    float run = 0.f;
    for (int i = 0; i < 256; ++i) {
      const float v = part[i];
      part[i] = run;
      run += v;
    }
  }
  __syncthreads();
  float run = part[t];
  float *cb = cdf + (long long)b * n_faces;
  for (int f = f0; f < f1; ++f) {
    float a = tri_area(vb, faces + 3 * f);
    if (a != a) a = 0.f;
    float p = fabsf(a / total);
    if (p != p) p = 1.f;
    run += p;
    cb[f] = run;
  }
}

int launch_face_cdf(const float *verts, const int32_t *faces, int batch, int n_vert, int n_faces, float *cdf,
                    hipStream_t s) {
  A3VT_LAUNCH(face_cdf_kernel, dim3(batch), dim3(256), 0, s, verts, faces, n_vert, n_faces, cdf);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// Philox4x32-10 (Salmon et al., SC'11): counter (4 x u32), key (2 x u32).
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)

__global__ __launch_bounds__(256) void sample_fwd_kernel(
    const float *__restrict__ verts, const int32_t *__restrict__ faces, const float *__restrict__ cdf, int batch,
    int n_vert, int n_faces, long long total, int num, const int32_t *__restrict__ fi_in,
    const float *__restrict__ u_in, const float *__restrict__ v_in, uint64_t seed, uint64_t offset,
    float *__restrict__ points, int32_t *__restrict__ fi_out, float *__restrict__ u_out, float *__restrict__ v_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // flat (draw, b, s)
  if (i >= total) return;
  const int b = (int)((i / num) % batch);
  int f;
  float u, v;
  if (fi_in) {
    f = fi_in[i];
    u = u_in[i];
    v = v_in[i];
  } else {
    const uint64_t ctr = offset + (uint64_t)i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float *cb = cdf + (long long)b * n_faces;
    const float tot = cb[n_faces - 1];
    float tgt = u01(c[0]) * tot;
    tgt = fminf(tgt, tot * (1.0f - 5.9604645e-8f));  // keep strictly below the total
    int lo = 0, hi = n_faces - 1;                    // first f with cdf[f] > tgt
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cb[mid] > tgt) hi = mid; else lo = mid + 1;
    }
    f = lo;
    u = u01(c[1]);
    v = u01(c[2]);
  }
  const float su = sqrtf(u);
  const float w0 = 1.0f - su, w1 = su * (1.0f - v), w2 = su * v;
  const float *vb = verts + (long long)b * n_vert * 3;
  const int32_t *fc = faces + 3 * (long long)f;
  const float *A = vb + 3 * (long long)fc[0], *B = vb + 3 * (long long)fc[1], *C = vb + 3 * (long long)fc[2];
#pragma unroll
  for (int d = 0; d < 3; ++d) points[i * 3 + d] = w0 * A[d] + w1 * B[d] + w2 * C[d];
  if (fi_out) fi_out[i] = f;
  if (u_out) u_out[i] = u;
  if (v_out) v_out[i] = v;
}

int launch_sample_fwd(const float *verts, const int32_t *faces, const float *cdf, int batch, int n_vert, int n_faces,
                      int draws, int num, const int32_t *fi_in, const float *u_in, const float *v_in, uint64_t seed,
                      uint64_t offset, float *points, int32_t *fi_out, float *u_out, float *v_out, hipStream_t s) {
  if (fi_in && (!u_in || !v_in)) {
    set_error("sample_fwd: face_idx_in given without u_in/v_in");
    return -1;
  }
  if (!fi_in && !cdf) {
    set_error("sample_fwd: cdf required when samples are not injected");
    return -1;
  }
  const long long total = (long long)draws * batch * num;
  A3VT_LAUNCH(sample_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, verts, faces, cdf, batch, n_vert,
                     n_faces, total, num, fi_in, u_in, v_in, seed, offset, points, fi_out, u_out, v_out);
  A3VT_CHECK_LAUNCH();
  return 0;
}

__global__ __launch_bounds__(256) void sample_bwd_kernel(const int32_t *__restrict__ faces, int batch, int n_vert,
                                                         long long total, int num, const int32_t *__restrict__ fi,
                                                         const float *__restrict__ u, const float *__restrict__ v,
                                                         const float *__restrict__ gp, float *__restrict__ gverts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)((i / num) % batch);
  const float su = sqrtf(u[i]);
  const float w[3] = {1.0f - su, su * (1.0f - v[i]), su * v[i]};
  const int32_t *fc = faces + 3 * (long long)fi[i];
  const float g0 = gp[i * 3], g1 = gp[i * 3 + 1], g2 = gp[i * 3 + 2];
  float *gb = gverts + (long long)b * n_vert * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float *dst = gb + 3 * (long long)fc[k];
    atomicAdd(dst + 0, w[k] * g0);
    atomicAdd(dst + 1, w[k] * g1);
    atomicAdd(dst + 2, w[k] * g2);
  }
}

// Same scatter with the vertex gradient of ONE mesh accumulated in LDS (n_vert * 12 bytes; 30 KB for 2562 vertices):
// the ~9 float atomics per sample hit LDS (ds_add_f32) instead of L2, and each workgroup then adds its slice of the
// mesh's samples to global memory with one coalesced pass.  grid = (batch, kSampleSplits).
constexpr int kSampleSplits = 4;
__global__ __launch_bounds__(1024) void sample_bwd_lds_kernel(const int32_t *__restrict__ faces, int batch, int n_vert,
                                                              int draws, int num, const int32_t *__restrict__ fi,
                                                              const float *__restrict__ u, const float *__restrict__ v,
                                                              const float *__restrict__ gp, float *__restrict__ gverts) {
  extern __shared__ float acc[];
  const int b = blockIdx.x;
  const int nf = n_vert * 3;
  for (int i = threadIdx.x; i < nf; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int per = (num + gridDim.y - 1) / gridDim.y;
  const int j0 = blockIdx.y * per, j1 = min(num, j0 + per);
  for (int d = 0; d < draws; ++d) {
    const long long base = ((long long)d * batch + b) * num;
    for (int j = j0 + threadIdx.x; j < j1; j += blockDim.x) {
      const long long i = base + j;
      const float su = sqrtf(u[i]);
      const float w[3] = {1.0f - su, su * (1.0f - v[i]), su * v[i]};
      const int32_t *fc = faces + 3 * (long long)fi[i];
      const float g0 = gp[i * 3], g1 = gp[i * 3 + 1], g2 = gp[i * 3 + 2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float *dst = acc + 3 * fc[k];
        atomicAdd(dst + 0, w[k] * g0);
        atomicAdd(dst + 1, w[k] * g1);
        atomicAdd(dst + 2, w[k] * g2);
      }
    }
  }
  __syncthreads();
  float *gb = gverts + (long long)b * nf;
  for (int i = threadIdx.x; i < nf; i += blockDim.x) {
    const float a = acc[i];
    if (a != 0.f) atomicAdd(gb + i, a);
  }
}

int launch_sample_bwd(const int32_t *faces, int batch, int n_vert, int n_faces, int draws, int num, const int32_t *fi,
                      const float *u, const float *v, const float *gpoints, float *gverts, hipStream_t s) {
  (void)n_faces;
  if (int rc = launch_fill_zero(gverts, (size_t)batch * n_vert * 3, s)) return rc;
  const size_t shmem = (size_t)n_vert * 3 * sizeof(float);
  if (shmem <= 144 * 1024) {  // up to 12 288 vertices per mesh (icosphere-5: 10 242)
    static size_t attr = 64 * 1024;
    if (shmem > attr) {
      (void)hipFuncSetAttribute((const void *)sample_bwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
      attr = 144 * 1024;
    }
    A3VT_LAUNCH(sample_bwd_lds_kernel, dim3(batch, kSampleSplits), dim3(1024), shmem, s, faces, batch, n_vert, draws, num,
                fi, u, v, gpoints, gverts);
    A3VT_CHECK_LAUNCH();
    return 0;
  }
  const long long total = (long long)draws * batch * num;
  A3VT_LAUNCH(sample_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, faces, batch, n_vert, total, num, fi, u,
                     v, gpoints, gverts);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
