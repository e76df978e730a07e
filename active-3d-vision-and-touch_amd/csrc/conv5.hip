// conv5.hip — the 5 x 5, stride-1, 16 -> 16 channel convolutions of the image pyramid on channels-last bf16 maps (round 6):
// layers 2 and 3 of `Image_Encoder` (reference: `CNN_layer` = BatchNorm2d -> ReLU -> Conv2d(k = 5, padding = 1),
// pterotactyl/reconstruction/vision/model.py:15-47; 16 -> 16 on the 126^2 and 124^2 maps) and — the same arithmetic with the
// weights transposed and flipped and padding 3 — the gradients of those layers with respect to their inputs.
//
// MIOpen runs these shapes on its generic NHWC implicit-GEMM kernel: 138 us per launch forward at bs 64 where the bytes (32 MB in,
// 31 MB out) are 13 us at 5 TB/s — K = 25 x 16 = 400, N = 16 is nothing for a 256 x 32 GEMM tile.  Here a workgroup owns a 16 x 16
// tile of output pixels: the 20 x 20 input patch goes to LDS by LDS-DMA (12.8 KB, pixels outside the map from a zero page), the
// 25 taps x 16 input channels are the k index of v_mfma_f32_16x16x32_bf16 (two taps per instruction, 13 per 16 pixels; the 26th
// tap has zero weights), the weights sit in 52 registers per lane as A fragments, a lane's B fragment is one ds_read_b128 (eight
// channels of one tap of its pixel), and the result tile D[channel][pixel] leaves a lane with four consecutive channels of its
// pixel: 8-byte stores, 512 contiguous bytes per wave instruction.  fp32 accumulation, bias (optional) in fp32, one rounding to bf16.
#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16 = unsigned short;

constexpr int kC = 16, kTaps = 25, kSteps = 13;   // channels in and out; taps; k-steps of 32 (two taps each)
constexpr int kTile = 16, kPatch = kTile + 4;     // output tile edge; input patch edge
constexpr int kPatchBytes = kPatch * kPatch * kC * 2;

__device__ __attribute__((aligned(16))) unsigned g_conv5_zero[4];   // 16 bytes of zeros: the source of out-of-map patch pieces

__device__ __forceinline__ void c5_glds16(const void *gsrc, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned c5_pack(float a, float b) {   // two floats -> two bf16, round to nearest even
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){a, b}, b2));
}

struct Conv5Args {
  const u16 *x;       // [B][H][W][16] bf16
  u16 *y;             // [B][Ho][Wo][16] bf16,  Ho = H + 2 pad - 4
  const u16 *wfrag;   // [13][64][8] bf16: the A fragments (conv5_weight_image_kernel)
  const float *bias;  // optional [16]
  int B, H, W, Ho, Wo, pad, tiles_x, tiles_y;
};

// A fragments of the weights.  Forward: Wm[co][tap = 5 ky + kx][ci] = w[co][ci][ky][kx] (w fp32, OIHW); gradient with respect to
// the input (flip != 0): Wm[ci][tap][co] = w[co][ci][4 - ky][4 - kx].  Fragment of k-step s for lane (m = lane & 15, q = lane >> 4):
// the eight values Wm[m][2 s + (q >> 1)][8 (q & 1) .. + 8]; tap 25 is zeros.
__global__ void conv5_weight_image_kernel(const float *__restrict__ w, int flip, u16 *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kSteps * 64 * 8) return;
  const int e = i & 7, lane = (i >> 3) & 63, s = i >> 9;
  const int m = lane & 15, q = lane >> 4;
  const int tap = 2 * s + (q >> 1), c = 8 * (q & 1) + e;
  float v = 0.f;
  if (tap < kTaps) {
    const int ky = tap / 5, kx = tap % 5;
    v = flip ? w[((c * kC + m) * 5 + (4 - ky)) * 5 + (4 - kx)] : w[((m * kC + c) * 5 + ky) * 5 + kx];
  }
  out[i] = (u16)(c5_pack(v, 0.f) & 0xffffu);
}

__global__ __launch_bounds__(256) void conv5x16_kernel(Conv5Args a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l16 = lane & 15, q = lane >> 4;
  const int tx = blockIdx.x % a.tiles_x, ty = (blockIdx.x / a.tiles_x) % a.tiles_y, b = blockIdx.x / (a.tiles_x * a.tiles_y);
  const int ox0 = tx * kTile, oy0 = ty * kTile;
  // ---- the input patch: pixel (py, px) of the patch is map pixel (oy0 + py - pad, ox0 + px - pad); two 16-byte pieces per pixel
  {
    const u16 *xb = a.x + (size_t)b * a.H * a.W * kC;
    constexpr int kPieces = kPatch * kPatch * 2;   // 800
    for (int i0 = wave * 64; i0 < kPieces; i0 += 256) {
      const int i = i0 + lane;
      if (i < kPieces) {
        const int p = i >> 1, half = i & 1;
        const int py = p / kPatch, px = p - py * kPatch;
        const int iy = oy0 + py - a.pad, ix = ox0 + px - a.pad;
        const void *src = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                              ? static_cast<const void *>(xb + ((size_t)iy * a.W + ix) * kC + half * 8)
                              : static_cast<const void *>(g_conv5_zero);
        c5_glds16(src, lds + i0 * 16);
      }
    }
  }
  // ---- the weights' A fragments (13 x 16 bytes per lane) and this lane's LDS offsets per k-step
  u32x4 wf[kSteps];
  int off[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    wf[s] = *reinterpret_cast<const u32x4 *>(a.wfrag + ((size_t)s * 64 + lane) * 8);
    int tap = 2 * s + (q >> 1);
    tap = tap < kTaps ? tap : kTaps - 1;          // (the 26th tap: zero weights; any finite pixel will do)
    const int ky = tap / 5, kx = tap - ky * 5;
    off[s] = ((ky * kPatch + kx) * kC + (q & 1) * 8) * 2;
  }
  f32x4 bs = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bs = *reinterpret_cast<const f32x4 *>(a.bias + 4 * q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- four output rows of 16 pixels per wave; D[m = channel][n = pixel]: this lane holds channels 4 q .. 4 q + 3 of pixel l16
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr;
    const char *base = lds + ((r * kPatch + l16) * kC) * 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const u32x4 pix = *reinterpret_cast<const u32x4 *>(base + off[s]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, pix), acc, 0, 0, 0);
    }
    const int oy = oy0 + r, ox = ox0 + l16;
    if (oy < a.Ho && ox < a.Wo) {
      acc += bs;
      const u32x2 o = {c5_pack(acc[0], acc[1]), c5_pack(acc[2], acc[3])};
      *reinterpret_cast<u32x2 *>(a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * kC + 4 * q) = o;
    }
  }
}

}  // namespace

size_t conv5_weight_image_bytes() { return (size_t)kSteps * 64 * 8 * sizeof(u16); }

int launch_conv5_weight_image(const float *w, int flip, void *image, hipStream_t s) {
  const int n = kSteps * 64 * 8;
  A3VT_LAUNCH(conv5_weight_image_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, flip, static_cast<u16 *>(image));
  A3VT_CHECK_LAUNCH();
  return 0;
}

int launch_conv5x16(const void *x, int batch, int h, int w, int pad, const void *image, const float *bias, void *y, hipStream_t s) {
  Conv5Args a{};
  a.x = static_cast<const u16 *>(x);
  a.y = static_cast<u16 *>(y);
  a.wfrag = static_cast<const u16 *>(image);
  a.bias = bias;
  a.B = batch;
  a.H = h;
  a.W = w;
  a.pad = pad;
  a.Ho = h + 2 * pad - 4;
  a.Wo = w + 2 * pad - 4;
  a.tiles_x = (a.Wo + kTile - 1) / kTile;
  a.tiles_y = (a.Ho + kTile - 1) / kTile;
  A3VT_LAUNCH(conv5x16_kernel, dim3((unsigned)(batch * a.tiles_x * a.tiles_y)), dim3(256), kPatchBytes, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace a3vt
