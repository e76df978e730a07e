// pooling.hip — per-vertex image-feature pooling (gfx950): projection + bilinear gather from every feature map +
// concatenation in one pass.
//
// Replaces Image_Encoder.pooling, reconstruction/vision/model.py:70-103: vertices are projected with the fixed camera
// matrix K.RT (:50-67), `z == 0 -> 0.1`, xs = P1 / P2 / 256, ys = P0 / P2 / 256, `inf -> 0.5`, grid = 2 (ys, xs) - 1,
// torch.nn.functional.grid_sample(map, grid, bilinear, zeros padding, align_corners=True) per map, torch.cat over
// the maps, permute to (B, N, C) — 3 grid_sample launches + cats per refinement stage in the reference.
//
// Layout: feature maps are read channels-last (B, H, W, C) — torch's channels_last memory format of a (B, C, H, W)
// tensor — so the C values of one pixel are contiguous and a vertex reads four contiguous C-vectors per map instead of
// C strided scalars.  One wave per vertex; a lane owns float4 channel groups.  HBM/L2-bound: per vertex 4 x 448 x 4 B
// read (maps of one sample: 170 KB, L2 resident) and 1792 B written.
// Backward: map gradients are accumulated per (sample, map, channel block) in LDS over all vertices of the sample
// (ds_add_f32); a sample's vertices are split over a few workgroups whose images are then added to global memory
// (one atomic per pixel-channel per workgroup); the position gradient is a wave reduction per vertex.
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct Bilinear {
  int x0, y0;             // north-west corner
  float wnw, wne, wsw, wse;
  float ix, iy;
  bool in_nw, in_ne, in_sw, in_se;
};

// grid_sample(align_corners=True, bilinear, zeros): unnormalise exactly as ATen does, ((g + 1) / 2) * (size - 1).
__device__ __forceinline__ Bilinear bilinear_setup(float gx, float gy, int H, int W) {
  Bilinear b;
  b.ix = ((gx + 1.f) / 2.f) * (float)(W - 1);
  b.iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
  const float fx = floorf(b.ix), fy = floorf(b.iy);
  b.x0 = (int)fx;
  b.y0 = (int)fy;
  const float x1 = fx + 1.f, y1 = fy + 1.f;
  b.wnw = (x1 - b.ix) * (y1 - b.iy);
  b.wne = (b.ix - fx) * (y1 - b.iy);
  b.wsw = (x1 - b.ix) * (b.iy - fy);
  b.wse = (b.ix - fx) * (b.iy - fy);
  const bool xin0 = b.x0 >= 0 && b.x0 < W, xin1 = b.x0 + 1 >= 0 && b.x0 + 1 < W;
  const bool yin0 = b.y0 >= 0 && b.y0 < H, yin1 = b.y0 + 1 >= 0 && b.y0 + 1 < H;
  b.in_nw = xin0 && yin0;
  b.in_ne = xin1 && yin0;
  b.in_sw = xin0 && yin1;
  b.in_se = xin1 && yin1;
  return b;
}

struct Projected {
  float gx, gy;        // grid coordinates: gx = 2 ys - 1 (width axis), gy = 2 xs - 1 (height axis)
  float p0, p1, p2;    // homogeneous image coordinates (p2 after the z == 0 -> 0.1 patch)
  bool z_patched, xs_patched, ys_patched;
};

__device__ __forceinline__ Projected project(const float *__restrict__ v, const float *__restrict__ m) {
  Projected r;
  const float x = v[0], y = v[1], z = v[2];
  r.p0 = ((x * m[0] + y * m[1]) + z * m[2]) + m[3];
  r.p1 = ((x * m[4] + y * m[5]) + z * m[6]) + m[7];
  float p2 = ((x * m[8] + y * m[9]) + z * m[10]) + m[11];
  r.z_patched = p2 == 0.f;
  r.p2 = r.z_patched ? 0.1f : p2;
  float xs = r.p1 / r.p2 / 256.0f, ys = r.p0 / r.p2 / 256.0f;
  r.xs_patched = isinf(xs);
  r.ys_patched = isinf(ys);
  xs = r.xs_patched ? 0.5f : xs;
  ys = r.ys_patched ? 0.5f : ys;
  r.gx = ys * 2.f - 1.f;
  r.gy = xs * 2.f - 1.f;
  return r;
}

// ---- forward: one wave per vertex
__global__ __launch_bounds__(256) void pool_fwd_kernel(PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)a.batch * a.n_vert;
  for (long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v < m; v += (long long)gridDim.x * 4) {
    const int b = (int)(v / a.n_vert);
    const Projected p = project(a.verts + v * 3, a.proj);
    float *out = a.feats + v * a.ld;
    for (int k = 0; k < a.n_maps; ++k) {
      const int C = a.C[k], H = a.H[k], W = a.W[k];
      const Bilinear bl = bilinear_setup(p.gx, p.gy, H, W);
      const float *mp = a.maps[k] + (long long)b * H * W * C;
      const float *nw = mp + ((long long)bl.y0 * W + bl.x0) * C;
      for (int c = lane * 4; c < C; c += 256) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (bl.in_nw) acc += bl.wnw * *reinterpret_cast<const f32x4 *>(nw + c);
        if (bl.in_ne) acc += bl.wne * *reinterpret_cast<const f32x4 *>(nw + C + c);
        if (bl.in_sw) acc += bl.wsw * *reinterpret_cast<const f32x4 *>(nw + (long long)W * C + c);
        if (bl.in_se) acc += bl.wse * *reinterpret_cast<const f32x4 *>(nw + (long long)(W + 1) * C + c);
        if (a.base) acc = *reinterpret_cast<const f32x4 *>(a.base + v * a.ld + a.off[k] + c) + acc;
        *reinterpret_cast<f32x4 *>(out + a.off[k] + c) = acc;
      }
    }
  }
}

// ---- backward, position: d feats / d (ix, iy) summed over channels, chained through the un-normalisation, the
// perspective division and the camera matrix (the in-place patches of the reference cut the gradient where they fire).
__global__ __launch_bounds__(256) void pool_bwd_verts_kernel(PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)a.batch * a.n_vert;
  for (long long v = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v < m; v += (long long)gridDim.x * 4) {
    const int b = (int)(v / a.n_vert);
    const Projected p = project(a.verts + v * 3, a.proj);
    const float *g = a.gfeats + v * a.ld;
    float ggx = 0.f, ggy = 0.f;  // gradient w.r.t. the grid coordinates
    for (int k = 0; k < a.n_maps; ++k) {
      const int C = a.C[k], H = a.H[k], W = a.W[k];
      const Bilinear bl = bilinear_setup(p.gx, p.gy, H, W);
      const float *mp = a.maps[k] + (long long)b * H * W * C;
      const float *nw = mp + ((long long)bl.y0 * W + bl.x0) * C;
      const float x1 = (float)(bl.x0 + 1), y1 = (float)(bl.y0 + 1), x0 = (float)bl.x0, y0 = (float)bl.y0;
      float gix = 0.f, giy = 0.f;
      for (int c = lane * 4; c < C; c += 256) {
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + a.off[k] + c);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 vnw = bl.in_nw ? *reinterpret_cast<const f32x4 *>(nw + c) : z4;
        const f32x4 vne = bl.in_ne ? *reinterpret_cast<const f32x4 *>(nw + C + c) : z4;
        const f32x4 vsw = bl.in_sw ? *reinterpret_cast<const f32x4 *>(nw + (long long)W * C + c) : z4;
        const f32x4 vse = bl.in_se ? *reinterpret_cast<const f32x4 *>(nw + (long long)(W + 1) * C + c) : z4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          gix += gv[t] * (-vnw[t] * (y1 - bl.iy) + vne[t] * (y1 - bl.iy) - vsw[t] * (bl.iy - y0) + vse[t] * (bl.iy - y0));
          giy += gv[t] * (-vnw[t] * (x1 - bl.ix) - vne[t] * (bl.ix - x0) + vsw[t] * (x1 - bl.ix) + vse[t] * (bl.ix - x0));
        }
      }
      ggx += gix * ((float)(W - 1) / 2.f);
      ggy += giy * ((float)(H - 1) / 2.f);
    }
    ggx = wave_sum(ggx);
    ggy = wave_sum(ggy);
    if (lane == 0) {
      // gx = 2 ys - 1, ys = p0 / p2 / 256 ; gy = 2 xs - 1, xs = p1 / p2 / 256
      const float gys = p.ys_patched ? 0.f : 2.f * ggx, gxs = p.xs_patched ? 0.f : 2.f * ggy;
      const float inv = 1.f / (p.p2 * 256.f);
      const float gp0 = gys * inv, gp1 = gxs * inv;
      const float gp2 = p.z_patched ? 0.f : -(gys * p.p0 + gxs * p.p1) * inv / p.p2;
      float *o = a.gverts + v * 3;
#pragma unroll
      for (int d = 0; d < 3; ++d) o[d] = gp0 * a.proj[d] + gp1 * a.proj[4 + d] + gp2 * a.proj[8 + d];
    }
  }
}

// ---- round 6: the same two kernels with every lane busy.  One wave per vertex with a lane per float4 channel group of ONE map
// leaves 48 of 64 lanes idle on the 64-channel map and 32 on the 128-channel one, and a wave has four loads in flight per pass:
// 120 / 164 us for 220 MB of features at bs 64 (1.8 / 1.3 TB/s).  Here a lane owns up to two of the vertex's (<= 128) channel
// groups ACROSS the maps — its map, channel offset, size and bilinear set-up are its own — and a wave handles two vertices per
// trip: sixteen 16-byte loads in flight per lane.  Same arithmetic per element as above (forward: bit-identical).
struct LaneGroup {
  int k, c;        // map and first channel of this lane's group (k < 0: none)
  int C, H, W, off;
};
__device__ __forceinline__ LaneGroup lane_group(const PoolArgs &a, int g) {
  LaneGroup r;
  r.k = -1;
  r.c = r.C = r.H = r.W = r.off = 0;
  for (int k = 0; k < a.n_maps; ++k) {
    const int g0 = a.off[k] >> 2, g1 = (a.off[k] + a.C[k]) >> 2;
    if (g >= g0 && g < g1) {
      r.k = k;
      r.c = (g - g0) * 4;
      r.C = a.C[k];
      r.H = a.H[k];
      r.W = a.W[k];
      r.off = a.off[k];
    }
  }
  return r;
}
__device__ __forceinline__ const float *lane_map(const PoolArgs &a, const LaneGroup &lg, int b) {
  const float *mp = a.maps[0];
#pragma unroll
  for (int k = 1; k < kMaxMaps; ++k)
    if (lg.k == k) mp = a.maps[k];
  return mp + (long long)b * lg.H * lg.W * lg.C;
}

__global__ __launch_bounds__(256) void pool_fwd2_kernel(PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)a.batch * a.n_vert;
  const LaneGroup lg[2] = {lane_group(a, lane), lane_group(a, lane + 64)};
  const long long stride = (long long)gridDim.x * 4;
  for (long long v0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v0 < m; v0 += 2 * stride) {
    f32x4 acc[2][2];
    f32x4 c4[2][2][4];
    float w4[2][2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long v = v0 + u * stride;
      const bool von = v < m;
      const long long vv = von ? v : v0;
      const int b = (int)(vv / a.n_vert);
      const Projected p = project(a.verts + vv * 3, a.proj);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        c4[u][j][0] = c4[u][j][1] = c4[u][j][2] = c4[u][j][3] = z4;
        w4[u][j][0] = w4[u][j][1] = w4[u][j][2] = w4[u][j][3] = 0.f;
        if (lg[j].k >= 0 && von) {
          const Bilinear bl = bilinear_setup(p.gx, p.gy, lg[j].H, lg[j].W);
          const float *nw = lane_map(a, lg[j], b) + ((long long)bl.y0 * lg[j].W + bl.x0) * lg[j].C + lg[j].c;
          w4[u][j][0] = bl.wnw; w4[u][j][1] = bl.wne; w4[u][j][2] = bl.wsw; w4[u][j][3] = bl.wse;
          if (bl.in_nw) c4[u][j][0] = *reinterpret_cast<const f32x4 *>(nw);
          if (bl.in_ne) c4[u][j][1] = *reinterpret_cast<const f32x4 *>(nw + lg[j].C);
          if (bl.in_sw) c4[u][j][2] = *reinterpret_cast<const f32x4 *>(nw + (long long)lg[j].W * lg[j].C);
          if (bl.in_se) c4[u][j][3] = *reinterpret_cast<const f32x4 *>(nw + (long long)(lg[j].W + 1) * lg[j].C);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long v = v0 + u * stride;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (a corner outside the map contributes w * 0: the same sum as skipping it)
        acc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][j] += w4[u][j][q] * c4[u][j][q];
        if (lg[j].k >= 0 && v < m) {
          if (a.base) acc[u][j] = *reinterpret_cast<const f32x4 *>(a.base + v * a.ld + lg[j].off + lg[j].c) + acc[u][j];
          *reinterpret_cast<f32x4 *>(a.feats + v * a.ld + lg[j].off + lg[j].c) = acc[u][j];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void pool_bwd_verts2_kernel(PoolArgs a) {
  const int lane = threadIdx.x & 63;
  const long long m = (long long)a.batch * a.n_vert;
  const LaneGroup lg[2] = {lane_group(a, lane), lane_group(a, lane + 64)};
  const long long stride = (long long)gridDim.x * 4;
  for (long long v0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); v0 < m; v0 += 2 * stride) {
    float ggx[2] = {0.f, 0.f}, ggy[2] = {0.f, 0.f};
    Projected pr[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long v = v0 + u * stride;
      const bool von = v < m;
      const long long vv = von ? v : v0;
      const int b = (int)(vv / a.n_vert);
      pr[u] = project(a.verts + vv * 3, a.proj);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (lg[j].k >= 0 && von) {
          const Bilinear bl = bilinear_setup(pr[u].gx, pr[u].gy, lg[j].H, lg[j].W);
          const float *nw = lane_map(a, lg[j], b) + ((long long)bl.y0 * lg[j].W + bl.x0) * lg[j].C + lg[j].c;
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 gv = *reinterpret_cast<const f32x4 *>(a.gfeats + vv * a.ld + lg[j].off + lg[j].c);
          const f32x4 vnw = bl.in_nw ? *reinterpret_cast<const f32x4 *>(nw) : z4;
          const f32x4 vne = bl.in_ne ? *reinterpret_cast<const f32x4 *>(nw + lg[j].C) : z4;
          const f32x4 vsw = bl.in_sw ? *reinterpret_cast<const f32x4 *>(nw + (long long)lg[j].W * lg[j].C) : z4;
          const f32x4 vse = bl.in_se ? *reinterpret_cast<const f32x4 *>(nw + (long long)(lg[j].W + 1) * lg[j].C) : z4;
          const float x1 = (float)(bl.x0 + 1), y1 = (float)(bl.y0 + 1), x0 = (float)bl.x0, y0 = (float)bl.y0;
          float gix = 0.f, giy = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            gix += gv[t] * (-vnw[t] * (y1 - bl.iy) + vne[t] * (y1 - bl.iy) - vsw[t] * (bl.iy - y0) + vse[t] * (bl.iy - y0));
            giy += gv[t] * (-vnw[t] * (x1 - bl.ix) - vne[t] * (bl.ix - x0) + vsw[t] * (x1 - bl.ix) + vse[t] * (bl.ix - x0));
          }
          ggx[u] += gix * ((float)(lg[j].W - 1) / 2.f);
          ggy[u] += giy * ((float)(lg[j].H - 1) / 2.f);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long v = v0 + u * stride;
      const float sx = wave_sum(ggx[u]), sy = wave_sum(ggy[u]);
      if (lane == 0 && v < m) {
        const Projected &p = pr[u];
        const float gys = p.ys_patched ? 0.f : 2.f * sx, gxs = p.xs_patched ? 0.f : 2.f * sy;
        const float inv = 1.f / (p.p2 * 256.f);
        const float gp0 = gys * inv, gp1 = gxs * inv;
        const float gp2 = p.z_patched ? 0.f : -(gys * p.p0 + gxs * p.p1) * inv / p.p2;
        float *o = a.gverts + v * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) o[d] = gp0 * a.proj[d] + gp1 * a.proj[4 + d] + gp2 * a.proj[8 + d];
      }
    }
  }
}

// ---- backward, maps: workgroup = (sample, map, block of CB channels); the block's gradient image [H*W][CB] lives in
// LDS, every vertex of the sample adds its four weighted corners, then the image is written once.  The accumulators
// are 64-bit fixed point (common.h): integer sums do not depend on the order in which the vertices arrive, so the map
// gradients — and with them the CNN weight gradients — are reproducible bit for bit (float LDS atomics plus a
// float-atomic combine of vertex slices, as before, were not).  Scale: the bilinear weights are <= 1, so the largest
// |grad_feats| of the block's channels over the sample's vertices bounds every term (a first pass over the same values),
// and a pixel receives at most one term per vertex and corner.
constexpr int kPoolLdsCells = 8192;  // 64 KB of accumulators: 23*23 pixels x 8 channels, 7*7 x 128, 3*3 x 256 all fit
__device__ __forceinline__ void pool_bwd_maps_body(const PoolArgs &a, int k, int CB, int block, long long *img, float *red) {
  const int C = a.C[k], H = a.H[k], W = a.W[k];
  const int nblk = (C + CB - 1) / CB;
  const int b = block / nblk, c0 = (block % nblk) * CB;
  const int cb = min(CB, C - c0);
  const int lanes_c = CB >> 2;               // float4 groups per vertex
  const int vpar = 256 / lanes_c;            // vertices in flight
  const int cl = (threadIdx.x % lanes_c) * 4, vs = threadIdx.x / lanes_c;
  const bool lane_on = cl < cb && vs < vpar;
  float mx = 0.f;
  for (int v = vs; v < a.n_vert && lane_on; v += vpar) {
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(a.gfeats + ((long long)b * a.n_vert + v) * a.ld + a.off[k] + c0 + cl);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(gv[0]), fabsf(gv[1]))), fmaxf(fabsf(gv[2]), fabsf(gv[3])));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  for (int i = threadIdx.x; i < H * W * CB; i += 256) img[i] = 0;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const bool finite = mx < 3.0e38f;
  const FixScale fs = fix_scale(finite ? mx : 0.f, a.n_vert);
  for (int v = vs; v < a.n_vert && lane_on; v += vpar) {
    const long long row = (long long)b * a.n_vert + v;
    const Projected p = project(a.verts + row * 3, a.proj);
    const Bilinear bl = bilinear_setup(p.gx, p.gy, H, W);
    const f32x4 gv = *reinterpret_cast<const f32x4 *>(a.gfeats + row * a.ld + a.off[k] + c0 + cl);
    long long *nw = img + (bl.y0 * W + bl.x0) * CB + cl;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (bl.in_nw) fix_add(nw + t, fix_from(bl.wnw * gv[t], fs));
      if (bl.in_ne) fix_add(nw + CB + t, fix_from(bl.wne * gv[t], fs));
      if (bl.in_sw) fix_add(nw + W * CB + t, fix_from(bl.wsw * gv[t], fs));
      if (bl.in_se) fix_add(nw + (W + 1) * CB + t, fix_from(bl.wse * gv[t], fs));
    }
  }
  __syncthreads();
  float *gm = a.gmaps[k] + (long long)b * H * W * C;
  for (int i = threadIdx.x; i < H * W * cb; i += 256) {
    const int px = i / cb, c = i - px * cb;
    gm[(long long)px * C + c0 + c] = finite ? fix_to(img[px * CB + c], fs) : __builtin_nanf("");
  }
}

// all maps in ONE launch (round 6): the three launches of a pyramid are independent, each fills the chip twice over at most and
// is bound by the latency of its serial vertex loop — 3 x 75 us back to back.  first[k] = first workgroup of map k.
struct PoolMapsPlan {
  int first[kMaxMaps + 1], cb[kMaxMaps];
};
__global__ __launch_bounds__(256) void pool_bwd_maps_all_kernel(PoolArgs a, PoolMapsPlan pl) {
  extern __shared__ long long img[];
  __shared__ float red[4];
  int k = 0;
#pragma unroll
  for (int j = 1; j < kMaxMaps; ++j)
    if (j < a.n_maps && (int)blockIdx.x >= pl.first[j]) k = j;
  pool_bwd_maps_body(a, k, pl.cb[k], blockIdx.x - pl.first[k], img, red);
}

static int check_pool(const PoolArgs &a) {
  if (a.n_maps < 1 || a.n_maps > kMaxMaps) { set_error("image_pool: n_maps=%d (1..%d)", a.n_maps, kMaxMaps); return -1; }
  int off = 0;
  for (int k = 0; k < a.n_maps; ++k) {
    if (a.C[k] % 4 != 0 || a.C[k] < 4 || a.H[k] < 1 || a.W[k] < 1) {
      set_error("image_pool: map %d is %d x %d x %d (channels must be a multiple of 4)", k, a.H[k], a.W[k], a.C[k]);
      return -1;
    }
    off += a.C[k];
  }
  if (a.ld % 4 != 0 || a.ld < off) { set_error("image_pool: ld=%d must be a multiple of 4 and >= %d", a.ld, off); return -1; }
  return 0;
}

// the lane-per-group kernels take a vertex whose maps are contiguous in the output row and hold <= 128 float4 groups in all
static bool pool_lanes_cover(const PoolArgs &a) {
  int off = 0;
  for (int k = 0; k < a.n_maps; ++k) {
    if (a.off[k] != off) return false;
    off += a.C[k];
  }
  return off <= 512;
}

int launch_pool_fwd(PoolArgs a, hipStream_t s) {
  if (int rc = check_pool(a)) return rc;
  const long long m = (long long)a.batch * a.n_vert;
  const int grid = (int)(cdiv(m, 4) < 8192 ? cdiv(m, 4) : 8192);
  if (pool_lanes_cover(a)) A3VT_LAUNCH(pool_fwd2_kernel, dim3(grid), dim3(256), 0, s, a);
  else A3VT_LAUNCH(pool_fwd_kernel, dim3(grid), dim3(256), 0, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_pool_bwd(PoolArgs a, hipStream_t s) {
  if (int rc = check_pool(a)) return rc;
  const long long m = (long long)a.batch * a.n_vert;
  const int grid = (int)(cdiv(m, 4) < 8192 ? cdiv(m, 4) : 8192);
  if (pool_lanes_cover(a)) A3VT_LAUNCH(pool_bwd_verts2_kernel, dim3(grid), dim3(256), 0, s, a);
  else A3VT_LAUNCH(pool_bwd_verts_kernel, dim3(grid), dim3(256), 0, s, a);
  A3VT_CHECK_LAUNCH();
  PoolMapsPlan pl{};
  size_t lds = 0;
  for (int k = 0; k < a.n_maps; ++k) {
    const int px = a.H[k] * a.W[k];
    int CB = a.C[k];                                   // channel block: largest power-of-two split that fits LDS ...
    while (CB > 4 && (px * CB > kPoolLdsCells || CB > 256)) CB >>= 1;
    if (px * CB > kPoolLdsCells) { set_error("image_pool: map %d x %d too large for the LDS image", a.H[k], a.W[k]); return -1; }
    while (CB > 8 && a.batch * cdiv(a.C[k], CB) < 512) CB >>= 1;   // ... and leaves every CU a couple of workgroups
    CB = (CB + 3) & ~3;
    pl.cb[k] = CB;
    pl.first[k + 1] = pl.first[k] + a.batch * cdiv(a.C[k], CB);
    const size_t need = (size_t)px * CB * sizeof(long long);
    lds = need > lds ? need : lds;
  }
  A3VT_LAUNCH(pool_bwd_maps_all_kernel, dim3(pl.first[a.n_maps]), dim3(256), lds, s, a, pl);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
