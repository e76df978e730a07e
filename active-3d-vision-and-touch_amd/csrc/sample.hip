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

// Backward of the barycentric blend: grad_verts[b][f_k] += w_k * grad_points[s] for the 3 corners of every sample's face
// (the reference's autograd does this with index_put atomics, utils.py:174-187).  One workgroup owns a tile of <= 6144
// vertices of ONE mesh, scans the mesh's draws * num samples and accumulates the contributions that land in its tile in
// 64-bit fixed point in LDS (common.h): integer sums do not depend on the arrival order, so the vertex gradient is
// reproducible bit for bit — the float atomics this replaces were the reason a training step was not.  Every gradient
// element is written exactly once (nothing to zero beforehand).  grid = (batch, tiles).
constexpr int kSampleBwdTile = 6144;  // 144 KiB of accumulators
__global__ __launch_bounds__(1024) void sample_bwd_kernel(const int32_t *__restrict__ faces, int batch, int n_vert,
                                                          int draws, int num, const int32_t *__restrict__ fi,
                                                          const float *__restrict__ u, const float *__restrict__ v,
                                                          const float *__restrict__ gp, float *__restrict__ gverts) {
  extern __shared__ long long facc[];  // [tile][3]
  __shared__ float red[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int tile = (n_vert + gridDim.y - 1) / gridDim.y;
  const int v0 = blockIdx.y * tile, v1 = min(n_vert, v0 + tile);
  // bound on |w_k * g| (0 <= w_k <= 1): the largest |grad_points| component of this mesh's samples
  float mx = 0.f;
  for (int d = 0; d < draws; ++d) {
    const float *g = gp + ((long long)d * batch + b) * num * 3;
    for (int i = tid; i < num * 3; i += 1024) mx = fmaxf(mx, fabsf(g[i]));
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  for (int i = tid; i < (v1 - v0) * 3; i += 1024) facc[i] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w) mx = fmaxf(mx, red[w]);
  const bool finite = mx < 3.0e38f;
  const FixScale fs = fix_scale(finite ? mx : 0.f, (long long)draws * num);
  for (int d = 0; d < draws; ++d) {
    const long long base = ((long long)d * batch + b) * num;
    for (int j = tid; j < num; j += 1024) {
      const long long i = base + j;
      const int32_t *fc = faces + 3 * (long long)fi[i];
      const int c0 = fc[0], c1 = fc[1], c2 = fc[2];
      const bool in0 = c0 >= v0 && c0 < v1, in1 = c1 >= v0 && c1 < v1, in2 = c2 >= v0 && c2 < v1;
      if (!(in0 || in1 || in2)) continue;
      const float su = sqrtf(u[i]);
      const float w[3] = {1.0f - su, su * (1.0f - v[i]), su * v[i]};
      const float g[3] = {gp[i * 3], gp[i * 3 + 1], gp[i * 3 + 2]};
      const int c[3] = {c0, c1, c2};
      const bool in[3] = {in0, in1, in2};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (!in[k]) continue;
#pragma unroll
        for (int e = 0; e < 3; ++e) fix_add(facc + (c[k] - v0) * 3 + e, fix_from(w[k] * g[e], fs));
      }
    }
  }
  __syncthreads();
  float *gb = gverts + ((long long)b * n_vert + v0) * 3;
  for (int i = tid; i < (v1 - v0) * 3; i += 1024) gb[i] = finite ? fix_to(facc[i], fs) : __builtin_nanf("");
}

int launch_sample_bwd(const int32_t *faces, int batch, int n_vert, int n_faces, int draws, int num, const int32_t *fi,
                      const float *u, const float *v, const float *gpoints, float *gverts, hipStream_t s) {
  (void)n_faces;
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)sample_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kSampleBwdTile * 3 * sizeof(long long));
  });
  const int tiles = cdiv(n_vert, kSampleBwdTile), tile = cdiv(n_vert, tiles);
  A3VT_LAUNCH(sample_bwd_kernel, dim3(batch, tiles), dim3(1024), (size_t)tile * 3 * sizeof(long long), s, faces, batch,
              n_vert, draws, num, fi, u, v, gpoints, gverts);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
