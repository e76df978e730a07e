// conv5.hip — the 5 x 5 convolutions of the image pyramid's first seven layers on channels-last bf16 maps (round 6): `Image_Encoder`
// (reference: `CNN_layer` = BatchNorm2d -> ReLU -> Conv2d(k = 5, padding = 1), pterotactyl/reconstruction/vision/model.py:15-47) runs
// 3 -> 3, 3 -> 16 stride 2, 16 -> 16 twice, 16 -> 32 stride 2, 32 -> 32 twice on 256^2 ... 58^2 maps before its channel counts become
// GEMM-sized.  This file holds, for those shapes: the forward (conv5_kernel / conv5c3_kernel), the gradients with respect to the
// inputs (the same arithmetic with the weights transposed and flipped and padding 3; layer 1's through the zero-upsampled form) and
// the gradients with respect to the weights (conv5_wrw_kernel / conv5c3_wrw_kernel + their fixed-order reduces, further down).
//
// MIOpen runs these shapes on its generic NHWC implicit-GEMM kernel: 138 us per launch forward at bs 64 where the bytes (32 MB in,
// 31 MB out) are 13 us at 5 TB/s — K = 25 x 16 = 400, N = 16 is nothing for a 256 x 32 GEMM tile.  Here a workgroup owns a 16 x 16
// tile of output pixels: the 20 x 20 input patch goes to LDS by LDS-DMA (12.8 KB, pixels outside the map from a zero page), the
// 25 taps x 16 input channels are the k index of v_mfma_f32_16x16x32_bf16 (two taps per instruction, 13 per 16 pixels; the 26th
// tap has zero weights), the weights sit in 52 registers per lane as A fragments, a lane's B fragment is one ds_read_b128 (eight
// channels of one tap of its pixel), and the result tile D[channel][pixel] leaves a lane with four consecutive channels of its
// pixel: 8-byte stores, 512 contiguous bytes per wave instruction.  fp32 accumulation, bias (optional) in fp32, one rounding to bf16.
#include <type_traits>
#include <utility>

#include "common.h"
#include "kernels.h"

namespace a3vt {

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u16 = unsigned short;

constexpr int kTaps = 25;
constexpr int kTile = 16;                         // output tile edge (pixels)

// Shape of an instantiation: CIN, COUT in {16, 32}, STRIDE in {1, 2}.  k index of the MFMA = (tap, input channel): two taps per
// k-step of 32 with 16 input channels (13 steps, the 26th tap has zero weights), one tap per step with 32 (25 steps).
template <int CIN, int COUT, int STRIDE>
struct C5 {
  static constexpr int kSteps = CIN == 16 ? 13 : 25;
  static constexpr int kMB = COUT / 16;                              // 16-channel blocks of the output
  static constexpr int kPatch = (kTile - 1) * STRIDE + 5;            // input patch edge: 20 (stride 1) or 35 (stride 2)
  static constexpr int kPix = CIN * 2;                               // bytes per input pixel
  static constexpr int kPatchBytes = kPatch * kPatch * kPix;
  static constexpr int kImageElems = kMB * kSteps * 64 * 8;          // A fragments: [m-block][step][lane][8]
};

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
  const u16 *x;       // [B][H][W][CIN] bf16
  u16 *y;             // [B][Ho][Wo][COUT] bf16,  Ho = (H + 2 pad - 5) / stride + 1
  const u16 *wfrag;   // the A fragments (conv5_weight_image_kernel)
  const float *bias;  // optional [COUT]
  int B, H, W, Ho, Wo, pad, tiles_x, tiles_y;
};

// A fragments of the weights.  Forward: Wm[co][tap = 5 ky + kx][ci] = w[co][ci][ky][kx] (w fp32, OIHW [cout][cin][5][5]); gradient
// with respect to the input (flip != 0; then the roles of the channel counts swap): Wm[ci][tap][co] = w[co][ci][4 - ky][4 - kx].
// Fragment of m-block mb, k-step s for lane (m = lane & 15, q = lane >> 4): eight consecutive k of row 16 mb + m, where
// k = 8 q + e is (tap 2 s + (q >> 1), channel 8 (q & 1) + e) with 16 inner channels and (tap s, channel 8 q + e) with 32.
__global__ void conv5_weight_image_kernel(const float *__restrict__ w, int flip, int rows, int inner, int real_rows, u16 *__restrict__ out) {
  const int steps = inner == 16 ? 13 : 25;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (rows / 16) * steps * 64 * 8) return;
  const int e = i & 7, lane = (i >> 3) & 63, s = (i >> 9) % steps, mb = (i >> 9) / steps;
  const int m = mb * 16 + (lane & 15), q = lane >> 4;
  const int tap = inner == 16 ? 2 * s + (q >> 1) : s, c = inner == 16 ? 8 * (q & 1) + e : 8 * q + e;
  float v = 0.f;
  if (tap < kTaps && m < real_rows) {     // (real_rows < rows: the flipped image of the 3-channel layer 1, rows 3 .. 15 zero)
    const int ky = tap / 5, kx = tap % 5;
    // forward: rows = cout, inner = cin, w[m][c]; flipped: rows = cin, inner = cout, w[co = c][ci = m]
    v = flip ? w[((c * real_rows + m) * 5 + (4 - ky)) * 5 + (4 - kx)] : w[((m * inner + c) * 5 + ky) * 5 + kx];
  }
  out[i] = (u16)(c5_pack(v, 0.f) & 0xffffu);
}

// UP3 (the input gradient of layer 1, 3 <- 16 at stride 2, as a stride-1 convolution): the input is read as if upsampled by two with
// zeros in between — patch pixel (iy, ix) exists when both are even and is map pixel (iy / 2, ix / 2); a.H, a.W are the UPSAMPLED
// sizes — and only channels 0 .. 2 of the result are stored, 6 bytes per pixel.  Three quarters of the products are with zeros:
// 13 MFMAs per 16 pixels do not care, and the kernel and its staging stay the ones above.
template <int CIN, int COUT, int STRIDE, bool UP3 = false>
__global__ __launch_bounds__(256) void conv5_kernel(Conv5Args a) {
  using S = C5<CIN, COUT, STRIDE>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l16 = lane & 15, q = lane >> 4;
  const int tx = blockIdx.x % a.tiles_x, ty = (blockIdx.x / a.tiles_x) % a.tiles_y, b = blockIdx.x / (a.tiles_x * a.tiles_y);
  const int ox0 = tx * kTile, oy0 = ty * kTile;
  // ---- the input patch: patch pixel (py, px) is map pixel (oy0 STRIDE + py - pad, ox0 STRIDE + px - pad); 16-byte pieces
  {
    const int mh = UP3 ? a.H >> 1 : a.H, mw = UP3 ? a.W >> 1 : a.W;    // the map as stored
    const u16 *xb = a.x + (size_t)b * mh * mw * CIN;
    constexpr int kPP = S::kPix / 16;                             // pieces per pixel: 2 or 4
    constexpr int kPieces = S::kPatch * S::kPatch * kPP;
    for (int i0 = wave * 64; i0 < kPieces; i0 += 256) {
      const int i = i0 + lane;
      if (i < kPieces) {
        const int p = i / kPP, part = i - p * kPP;
        const int py = p / S::kPatch, px = p - py * S::kPatch;
        const int iy = oy0 * STRIDE + py - a.pad, ix = ox0 * STRIDE + px - a.pad;
        const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && (!UP3 || ((iy | ix) & 1) == 0);
        const int sy = UP3 ? iy >> 1 : iy, sx = UP3 ? ix >> 1 : ix;
        const void *src = in ? static_cast<const void *>(xb + ((size_t)sy * mw + sx) * CIN + part * 8)
                             : static_cast<const void *>(g_conv5_zero);
        c5_glds16(src, lds + i0 * 16);
      }
    }
  }
  // ---- the weights' A fragments and this lane's LDS offsets per k-step
  u32x4 wf[S::kMB][S::kSteps];
  int off[S::kSteps];
#pragma unroll
  for (int s = 0; s < S::kSteps; ++s) {
#pragma unroll
    for (int mb = 0; mb < S::kMB; ++mb)
      wf[mb][s] = *reinterpret_cast<const u32x4 *>(a.wfrag + (((size_t)mb * S::kSteps + s) * 64 + lane) * 8);
    int tap = CIN == 16 ? 2 * s + (q >> 1) : s;
    tap = tap < kTaps ? tap : kTaps - 1;          // (the 26th tap: zero weights; any finite pixel will do)
    const int ky = tap / 5, kx = tap - ky * 5;
    off[s] = (ky * S::kPatch + kx) * S::kPix + (CIN == 16 ? (q & 1) : q) * 16;
  }
  f32x4 bs[S::kMB];
#pragma unroll
  for (int mb = 0; mb < S::kMB; ++mb) {
    bs[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bs[mb] = *reinterpret_cast<const f32x4 *>(a.bias + mb * 16 + 4 * q);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ---- four output rows of 16 pixels per wave; D[m = channel][n = pixel]: this lane holds channels 4 q .. 4 q + 3 of pixel l16
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr;
    const char *base = lds + (r * STRIDE * S::kPatch + l16 * STRIDE) * S::kPix;
    f32x4 acc[S::kMB];
#pragma unroll
    for (int mb = 0; mb < S::kMB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S::kSteps; ++s) {
      const u32x4 pix = *reinterpret_cast<const u32x4 *>(base + off[s]);
#pragma unroll
      for (int mb = 0; mb < S::kMB; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[mb][s]), __builtin_bit_cast(bf16x8, pix), acc[mb], 0, 0, 0);
    }
    const int oy = oy0 + r, ox = ox0 + l16;
    if (oy < a.Ho && ox < a.Wo) {
      if (UP3) {
        if (q == 0) {
          u16 *o = a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * 3;
          const unsigned lo = c5_pack(acc[0][0], acc[0][1]), hi = c5_pack(acc[0][2], 0.f);
          o[0] = (u16)(lo & 0xffffu);
          o[1] = (u16)(lo >> 16);
          o[2] = (u16)(hi & 0xffffu);
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < S::kMB; ++mb) {
          const f32x4 v = acc[mb] + bs[mb];
          const u32x2 o = {c5_pack(v[0], v[1]), c5_pack(v[2], v[3])};
          *reinterpret_cast<u32x2 *>(a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * COUT + mb * 16 + 4 * q) = o;
        }
      }
    }
  }
}

// ---- the two 3-channel layers of the pyramid: layer 0 (3 -> 3, stride 1) and layer 1 (3 -> 16, stride 2) ---------------------------
// A pixel is 6 bytes in the map and 8 in LDS (a zero fourth channel): the patch is staged through registers (three 2-byte loads and
// one ds_write_b64 per pixel).  k index = (tap, channel of 4): eight taps per k-step, FOUR k-steps for the 25 taps — a lane's B
// fragment is two 8-byte LDS reads (taps 8 s + 2 q and 8 s + 2 q + 1 of its pixel).  Output rows beyond COUT (3 -> 3: thirteen of
// the sixteen) have zero weights and are not stored.
template <int COUT, int STRIDE>
__global__ __launch_bounds__(256) void conv5c3_kernel(Conv5Args a) {
  constexpr int kPatch = (kTile - 1) * STRIDE + 5, kSteps3 = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l16 = lane & 15, q = lane >> 4;
  const int tx = blockIdx.x % a.tiles_x, ty = (blockIdx.x / a.tiles_x) % a.tiles_y, b = blockIdx.x / (a.tiles_x * a.tiles_y);
  const int ox0 = tx * kTile, oy0 = ty * kTile;
  {
    const u16 *xb = a.x + (size_t)b * a.H * a.W * 3;
    for (int p = t; p < kPatch * kPatch; p += 256) {
      const int py = p / kPatch, px = p - py * kPatch;
      const int iy = oy0 * STRIDE + py - a.pad, ix = ox0 * STRIDE + px - a.pad;
      u32x2 v = {0u, 0u};
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
        const u16 *src = xb + ((size_t)iy * a.W + ix) * 3;
        v[0] = (unsigned)src[0] | ((unsigned)src[1] << 16);
        v[1] = (unsigned)src[2];
      }
      *reinterpret_cast<u32x2 *>(lds + p * 8) = v;
    }
  }
  u32x4 wf[kSteps3];
  int off0[kSteps3], off1[kSteps3];
#pragma unroll
  for (int s = 0; s < kSteps3; ++s) {
    wf[s] = *reinterpret_cast<const u32x4 *>(a.wfrag + ((size_t)s * 64 + lane) * 8);
    int t0 = 8 * s + 2 * q, t1 = t0 + 1;
    t0 = t0 < kTaps ? t0 : kTaps - 1;             // (taps 25 .. 31: zero weights)
    t1 = t1 < kTaps ? t1 : kTaps - 1;
    off0[s] = ((t0 / 5) * kPatch + t0 % 5) * 8;
    off1[s] = ((t1 / 5) * kPatch + t1 % 5) * 8;
  }
  f32x4 bs = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
    if (COUT == 16) bs = *reinterpret_cast<const f32x4 *>(a.bias + 4 * q);
    else if (q == 0) bs = f32x4{a.bias[0], a.bias[1], a.bias[2], 0.f};
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = wave * 4 + rr;
    const char *base = lds + (r * STRIDE * kPatch + l16 * STRIDE) * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kSteps3; ++s) {
      const u32x2 p0 = *reinterpret_cast<const u32x2 *>(base + off0[s]);
      const u32x2 p1 = *reinterpret_cast<const u32x2 *>(base + off1[s]);
      const u32x4 pix = {p0[0], p0[1], p1[0], p1[1]};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, pix), acc, 0, 0, 0);
    }
    const int oy = oy0 + r, ox = ox0 + l16;
    if (oy < a.Ho && ox < a.Wo) {
      const f32x4 v = acc + bs;
      if (COUT == 16) {
        const u32x2 o = {c5_pack(v[0], v[1]), c5_pack(v[2], v[3])};
        *reinterpret_cast<u32x2 *>(a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * 16 + 4 * q) = o;
      } else if (q == 0) {
        u16 *o = a.y + (((size_t)b * a.Ho + oy) * a.Wo + ox) * 3;
        const unsigned lo = c5_pack(v[0], v[1]), hi = c5_pack(v[2], 0.f);
        o[0] = (u16)(lo & 0xffffu);
        o[1] = (u16)(lo >> 16);
        o[2] = (u16)(hi & 0xffffu);
      }
    }
  }
}

// A fragments for the 3-channel layers: Wm[co][tap][ci of 4] = w[co][ci][ky][kx] (ci 3, taps >= 25 and rows >= cout: zeros); lane
// (m, q) of k-step s holds taps 8 s + 2 q and 8 s + 2 q + 1, four channels each
__global__ void conv5c3_weight_image_kernel(const float *__restrict__ w, int cout, u16 *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * 64 * 8) return;
  const int e = i & 7, lane = (i >> 3) & 63, s = i >> 9;
  const int m = lane & 15, q = lane >> 4;
  const int tap = 8 * s + 2 * q + (e >> 2), c = e & 3;
  float v = 0.f;
  if (tap < kTaps && c < 3 && m < cout) v = w[((m * 3 + c) * 5 + tap / 5) * 5 + tap % 5];
  out[i] = (u16)(c5_pack(v, 0.f) & 0xffffu);
}

template <int COUT, int STRIDE>
int conv5c3_launch(const Conv5Args &a, hipStream_t s) {
  constexpr int kPatch = (kTile - 1) * STRIDE + 5;
  A3VT_LAUNCH((conv5c3_kernel<COUT, STRIDE>), dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), kPatch * kPatch * 8, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <int CIN, int COUT, int STRIDE, bool UP3 = false>
int conv5_launch(const Conv5Args &a, hipStream_t s) {
  using S = C5<CIN, COUT, STRIDE>;
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)conv5_kernel<CIN, COUT, STRIDE, UP3>, hipFuncAttributeMaxDynamicSharedMemorySize, S::kPatchBytes);
  });
  A3VT_LAUNCH((conv5_kernel<CIN, COUT, STRIDE, UP3>), dim3((unsigned)(a.B * a.tiles_x * a.tiles_y)), dim3(256), S::kPatchBytes, s, a);
  A3VT_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// ---- weight gradients ------------------------------------------------------------------------------------------------------------
// gw[co][ci][ky][kx] = sum over (b, oy, ox) of gy[b][oy][ox][co] * x[b][oy S + ky - 1][ox S + kx - 1][ci]: per tap a [COUT x CIN] product
// summed over PIXELS — both MFMA operands are needed pixel-wise from channels-last data, which is what ds_read_b64_tr_b16 delivers
// (dw16_kernel's tr_operand, gcn_bf16s.hip).  A k-step is 32 consecutive output pixels of one row: A = gy^T (rows of 16 output
// channels), B = the same 32 pixels of x shifted by the tap.  A workgroup loops over tiles of 8 rows x 32 pixels (gy tile and x
// patch by LDS-DMA into two stages, out-of-map pieces from the zero page); wave ky of five owns the taps (ky, 0 .. 4) and keeps
// their COUT x CIN sums in registers across ALL its tiles; at the end every workgroup writes one partial image [25][COUT][CIN] and a second launch adds the
// images in a fixed order, straight into the fp32 [cout][cin][5][5] gradient: no atomics, no fp32 workspace to zero and cast
// (MIOpen: a fill, the kernel, a cast, then torch's cast to fp32 — four launches per layer).
constexpr int kWR = 8, kWC = 32;                  // tile: output rows x output pixels per row
constexpr int kWrwWgs = 512;                     // most partial images a launch writes

struct Conv5WrwArgs {
  const u16 *x, *gy;    // [B][H][W][CIN], [B][Ho][Wo][COUT] bf16
  float *partial;       // [workgroups][25][COUT][CIN]
  int B, H, W, Ho, Wo, tiles_x, tiles_y;
};

// Transposed LDS reads as written instructions: with an LDS-DMA in flight the compiler puts s_waitcnt vmcnt(0) in front of every
// LDS read it issues itself (csr16t_kernel, gcn_bf16s.hip) — here that is the NEXT tile's prefetch.  The waits are written too and
// tied to the registers they release.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <int OFF>
__device__ __forceinline__ u32x2 c5_tr64(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
struct C5Op { u32x2 lo, hi; };      // one MFMA operand: pixels 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 of the k-step
template <int OFF, int BLOCK2>
__device__ __forceinline__ C5Op c5_op(unsigned addr) { return C5Op{c5_tr64<OFF>(addr), c5_tr64<OFF + BLOCK2>(addr)}; }
__device__ __forceinline__ bf16x8 c5_bf(const C5Op &o) { return __builtin_bit_cast(bf16x8, (u32x4){o.lo[0], o.lo[1], o.hi[0], o.hi[1]}); }
template <int I, int N, class F>
__device__ __forceinline__ void c5_static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c5_static_for<I + 1, N>(f);
  }
}
template <int N>
__device__ __forceinline__ void c5_wait_tied(C5Op *a, int na, C5Op *b) {      // lgkmcnt(N), releasing a[0 .. na) and b[0 .. 5)
  if (na == 2)
    asm volatile("s_waitcnt lgkmcnt(%14)" : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(a[1].lo), "+v"(a[1].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(b[1].lo),
                 "+v"(b[1].hi), "+v"(b[2].lo), "+v"(b[2].hi), "+v"(b[3].lo), "+v"(b[3].hi), "+v"(b[4].lo), "+v"(b[4].hi) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a[0].lo), "+v"(a[0].hi), "+v"(b[0].lo), "+v"(b[0].hi), "+v"(b[1].lo),
                 "+v"(b[1].hi), "+v"(b[2].lo), "+v"(b[2].hi), "+v"(b[3].lo), "+v"(b[3].hi), "+v"(b[4].lo), "+v"(b[4].hi) : "n"(N));
}

// Five waves: wave ky owns the taps (ky, 0 .. 4).  A "group" is (tile row r, input-channel block nb): its reads are the gy operands
// of row r (with nb == 0) and the five x operands of block nb; the reads of group i + 1 are issued in front of the MFMAs of group i.
template <int CIN, int COUT, int STRIDE>
__global__ __launch_bounds__(320) void conv5_wrw_kernel(Conv5WrwArgs a) {
  constexpr int kMB = COUT / 16, kNB = CIN / 16;
  constexpr int kPH = (kWR - 1) * STRIDE + 5, kPW = (kWC - 1) * STRIDE + 5;      // x patch
  constexpr int kGyBytes = kWR * kWC * COUT * 2, kXBytes = kPH * kPW * CIN * 2, kStage = kGyBytes + kXBytes;
  constexpr int kThreads = 320;
  extern __shared__ __attribute__((aligned(16))) char lds[];      // two stages: the next tile arrives while this one is multiplied
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g = lane >> 4;
  const int bq = l16 >> 2, bp = l16 & 3;          // transposed-read address role (dw16_kernel): block row bq, column quad bp
  f32x4 acc[5][kMB][kNB];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb) acc[i][mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tiles = a.B * a.tiles_x * a.tiles_y;
  auto stage = [&](int tile, char *st) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int ox0 = tx * kWC, oy0 = ty * kWR;
    {   // gy tile: pixel (r, c) of the tile = (oy0 + r, ox0 + c); pieces of 16 bytes
      const u16 *gb = a.gy + (size_t)b * a.Ho * a.Wo * COUT;
      constexpr int kPP = COUT / 8, kPieces = kWR * kWC * kPP;
      for (int i0 = wave * 64; i0 < kPieces; i0 += kThreads) {
        const int i = i0 + lane;
        if (i < kPieces) {
          const int p = i / kPP, part = i - p * kPP;
          const int r = p / kWC, c = p - r * kWC;
          const int oy = oy0 + r, ox = ox0 + c;
          const void *src = (oy < a.Ho && ox < a.Wo) ? static_cast<const void *>(gb + ((size_t)oy * a.Wo + ox) * COUT + part * 8)
                                                      : static_cast<const void *>(g_conv5_zero);
          c5_glds16(src, st + i0 * 16);
        }
      }
    }
    {   // x patch: pixel (py, px) = map pixel (oy0 S + py - 1, ox0 S + px - 1)
      const u16 *xb = a.x + (size_t)b * a.H * a.W * CIN;
      constexpr int kPP = CIN / 8, kPieces = kPH * kPW * kPP;
      for (int i0 = wave * 64; i0 < kPieces; i0 += kThreads) {
        const int i = i0 + lane;
        if (i < kPieces) {
          const int p = i / kPP, part = i - p * kPP;
          const int py = p / kPW, px = p - py * kPW;
          const int iy = oy0 * STRIDE + py - 1, ix = ox0 * STRIDE + px - 1;
          const void *src = (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? static_cast<const void *>(xb + ((size_t)iy * a.W + ix) * CIN + part * 8)
                                                                         : static_cast<const void *>(g_conv5_zero);
          c5_glds16(src, st + kGyBytes + i0 * 16);
        }
      }
    }
  };
  // LDS byte addresses of this lane's transposed reads in stage 0: gy pixel (4 g + bq) of row 0, channel quad bp; x pixel
  // ((4 g + bq) S, row ky) of the patch
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
  const unsigned a_addr0 = lds0 + ((4 * g + bq) * COUT + 4 * bp) * 2;
  const unsigned b_addr0 = lds0 + kGyBytes + ((wave * kPW + (4 * g + bq) * STRIDE) * CIN + 4 * bp) * 2;
  int buf = 0;
  if ((int)blockIdx.x < tiles) stage(blockIdx.x, lds);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the tile have landed ...
    __builtin_amdgcn_s_barrier();                        // ... everyone's have, and everyone has left the other stage
    if (tile + (int)gridDim.x < tiles) stage(tile + gridDim.x, lds + (buf ^ 1) * kStage);
    const unsigned a_addr = a_addr0 + buf * kStage, b_addr = b_addr0 + buf * kStage;
    C5Op af[2][kMB], bf[2][5];
    auto reads = [&](auto R, auto NB, auto SLOT) {       // the reads of group (R, NB) into operand set SLOT
      constexpr int r = decltype(R)::value, nb = decltype(NB)::value, slot = decltype(SLOT)::value;
      if constexpr (nb == 0) {
        af[r & 1][0] = c5_op<r * kWC * COUT * 2, 16 * COUT * 2>(a_addr);
        if constexpr (kMB == 2) af[r & 1][1] = c5_op<r * kWC * COUT * 2 + 32, 16 * COUT * 2>(a_addr);
      }
      bf[slot][0] = c5_op<((r * STRIDE) * kPW + 0) * CIN * 2 + nb * 32, 16 * STRIDE * CIN * 2>(b_addr);
      bf[slot][1] = c5_op<((r * STRIDE) * kPW + 1) * CIN * 2 + nb * 32, 16 * STRIDE * CIN * 2>(b_addr);
      bf[slot][2] = c5_op<((r * STRIDE) * kPW + 2) * CIN * 2 + nb * 32, 16 * STRIDE * CIN * 2>(b_addr);
      bf[slot][3] = c5_op<((r * STRIDE) * kPW + 3) * CIN * 2 + nb * 32, 16 * STRIDE * CIN * 2>(b_addr);
      bf[slot][4] = c5_op<((r * STRIDE) * kPW + 4) * CIN * 2 + nb * 32, 16 * STRIDE * CIN * 2>(b_addr);
    };
    auto group = [&](auto GI) {
      constexpr int gi = decltype(GI)::value, r = gi / kNB, nb = gi % kNB, slot = gi & 1;
      constexpr int gn = gi + 1, rn = gn / kNB, nbn = gn % kNB;
      constexpr int next_reads = gn < kWR * kNB ? 10 + (nbn == 0 ? 2 * kMB : 0) : 0;
      if constexpr (gn < kWR * kNB) reads(std::integral_constant<int, rn>{}, std::integral_constant<int, nbn>{}, std::integral_constant<int, (slot ^ 1)>{});
      c5_wait_tied<next_reads>(af[r & 1], kMB, bf[slot]);
#pragma unroll
      for (int kx = 0; kx < 5; ++kx)
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb)
          acc[kx][mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c5_bf(af[r & 1][mb]), c5_bf(bf[slot][kx]), acc[kx][mb][nb], 0, 0, 0);
    };
    reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    c5_static_for<0, kWR * kNB>(group);
    buf ^= 1;
  }
  // ---- this workgroup's image: D[m = co][n = ci]: the lane holds co = 16 mb + 4 g + e, ci = 16 nb + l16
  float *img = a.partial + (size_t)blockIdx.x * kTaps * COUT * CIN;
#pragma unroll
  for (int kx = 0; kx < 5; ++kx)
#pragma unroll
    for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
      for (int nb = 0; nb < kNB; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) img[((size_t)(wave * 5 + kx) * COUT + mb * 16 + 4 * g + e) * CIN + nb * 16 + l16] = acc[kx][mb][nb][e];
}

// gw[co][ci][ky][kx] (fp32, OIHW) = sum over the workgroups' images [wg][tap][co][ci] in a fixed order: a workgroup owns 64
// consecutive entries (16 lanes x float4) and splits the images over 32 slices of lanes; slice s adds images s, s + 32, ... in
// that order, then 64 lanes add the 32 slice sums of their entry in slice order.
constexpr int kRedSlices = 32;
__global__ __launch_bounds__(16 * kRedSlices) void conv5_wrw_reduce_kernel(const float *__restrict__ partial, int nwg, int cout, int cin, float *__restrict__ gw) {
  __shared__ f32x4 part[kRedSlices][16];
  const int n = kTaps * cout * cin;                  // a multiple of 64
  const int q = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const size_t i4 = (size_t)blockIdx.x * 64 + q * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int w = sl; w < nwg; w += kRedSlices) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(partial + (size_t)w * n + i4));
    s += v;
  }
  part[sl][q] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    const float *p = reinterpret_cast<const float *>(&part[0][0]) + threadIdx.x;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < kRedSlices; ++k) r += p[k * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int tap = i / (cout * cin), co = (i / cin) % cout, ci = i % cin;
    gw[((size_t)co * cin + ci) * kTaps + tap] = r;
  }
}

// ---- weight gradients of the two 3-channel layers (3 -> 3 stride 1, 3 -> 16 stride 2) ------------------------------------------------
// The same product with the roles the 3-channel forward gives them: the x patch sits in LDS with 8-byte pixels (a zero fourth
// channel), so the four channels of a pixel are what one lane hands to a transposed read; column n of the B operand is (tap, channel)
// = (n >> 2, n & 3): 25 taps = 100 columns = SEVEN 16-column blocks (the lane that addresses for column quad bp of block nb points
// at tap 4 nb + bp of its pixel; taps 25 .. 27 are clamped and their columns dropped).  A = gy^T as above (3 -> 3: gy also with
// 8-byte pixels, the lanes of channel quads 1 .. 3 read a zero region).  Wave w owns rows 2 w, 2 w + 1 of a tile with all seven
// accumulators; the four waves' sums are added through LDS and the workgroup writes ONE image in fragment order, which
// conv5c3_wrw_reduce_kernel adds over the workgroups in a fixed order and scatters into [cout][3][5][5].
constexpr int kC3Blocks = 7, kC3Image = kC3Blocks * 64 * 4;      // floats per partial image
constexpr int kC3Wgs = 1024;

__device__ __forceinline__ bf16x8 c5_tr_pair(const char *p, int block2_bytes) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p + block2_bytes));
  return __builtin_bit_cast(bf16x8, (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

// a [rows][cols] window of 6-byte pixels of a map -> 8-byte pixels in LDS (zero outside the map)
__device__ __forceinline__ void c5_stage_px3(const u16 *map, int H, int W, int y0, int x0, int rows, int cols, char *dst) {
  for (int p = threadIdx.x; p < rows * cols; p += 256) {
    const int py = p / cols, px = p - py * cols;
    const int iy = y0 + py, ix = x0 + px;
    u32x2 v = {0u, 0u};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
      const u16 *src = map + ((size_t)iy * W + ix) * 3;
      v[0] = (unsigned)src[0] | ((unsigned)src[1] << 16);
      v[1] = (unsigned)src[2];
    }
    *reinterpret_cast<u32x2 *>(dst + p * 8) = v;
  }
}

template <int COUT, int STRIDE>
__global__ __launch_bounds__(256) void conv5c3_wrw_kernel(Conv5WrwArgs a) {
  constexpr int kPH = (kWR - 1) * STRIDE + 5, kPW = (kWC - 1) * STRIDE + 5;
  constexpr int kXBytes = (kPH * kPW * 8 + 15) / 16 * 16, kGyPix = COUT == 16 ? 32 : 8, kGyBytes = kWR * kWC * kGyPix;
  extern __shared__ __attribute__((aligned(16))) char lds[];      // x patch | gy tile | 256 zero bytes; at the end the waves' sums
  char *x_l = lds, *gy_l = lds + kXBytes, *zero_l = lds + kXBytes + kGyBytes;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l16 = lane & 15, g = lane >> 4;
  const int bq = l16 >> 2, bp = l16 & 3;
  if (t < 16) reinterpret_cast<u32x4 *>(zero_l)[t] = u32x4{0u, 0u, 0u, 0u};
  // this lane's read addresses in row 0 of the tile
  const char *a_ptr;
  int a_row;
  if (COUT == 16) {
    a_ptr = gy_l + ((4 * g + bq) * 16 + 4 * bp) * 2;
    a_row = kWC * 32;
  } else {
    a_ptr = bp == 0 ? gy_l + (4 * g + bq) * 8 : zero_l;
    a_row = bp == 0 ? kWC * 8 : 0;
  }
  int boff[kC3Blocks];
#pragma unroll
  for (int nb = 0; nb < kC3Blocks; ++nb) {
    int tap = 4 * nb + bp;
    tap = tap < kTaps ? tap : kTaps - 1;
    boff[nb] = (((tap / 5) * kPW + tap % 5) + (4 * g + bq) * STRIDE) * 8;
  }
  f32x4 acc[kC3Blocks];
#pragma unroll
  for (int nb = 0; nb < kC3Blocks; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tiles = a.B * a.tiles_x * a.tiles_y;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
    const int ox0 = tx * kWC, oy0 = ty * kWR;
    __syncthreads();        // everyone has left the previous tile
    c5_stage_px3(a.x + (size_t)b * a.H * a.W * 3, a.H, a.W, oy0 * STRIDE - 1, ox0 * STRIDE - 1, kPH, kPW, x_l);
    if (COUT == 16) {
      const u16 *gb = a.gy + (size_t)b * a.Ho * a.Wo * 16;
#pragma unroll
      for (int k = 0; k < 2; ++k) {      // 512 pieces of 16 bytes
        const int i0 = (wave + 4 * k) * 64, i = i0 + lane;
        const int p = i >> 1, part = i & 1;
        const int oy = oy0 + p / kWC, ox = ox0 + p % kWC;
        const void *src = (oy < a.Ho && ox < a.Wo) ? static_cast<const void *>(gb + ((size_t)oy * a.Wo + ox) * 16 + part * 8)
                                                    : static_cast<const void *>(g_conv5_zero);
        c5_glds16(src, gy_l + i0 * 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      c5_stage_px3(a.gy + (size_t)b * a.Ho * a.Wo * 3, a.Ho, a.Wo, oy0, ox0, kWR, kWC, gy_l);
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = wave * 2 + rr;
      const bf16x8 af = c5_tr_pair(a_ptr + r * a_row, COUT == 16 ? 16 * 32 : 16 * 8);
      const char *xr = x_l + r * STRIDE * kPW * 8;
#pragma unroll
      for (int nb = 0; nb < kC3Blocks; ++nb) {
        const bf16x8 bf = c5_tr_pair(xr + boff[nb], 16 * STRIDE * 8);
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, acc[nb], 0, 0, 0);
      }
    }
  }
  // ---- the four waves' sums, added in wave order; image entry (nb, lane, e) = D[co = 4 g + e][n = 16 nb + l16]
  __syncthreads();
  f32x4 *sum_l = reinterpret_cast<f32x4 *>(lds);
#pragma unroll
  for (int nb = 0; nb < kC3Blocks; ++nb) sum_l[(wave * kC3Blocks + nb) * 64 + lane] = acc[nb];
  __syncthreads();
  float *img = a.partial + (size_t)blockIdx.x * kC3Image;
  const float *sf = reinterpret_cast<const float *>(lds);
  for (int i = t; i < kC3Image; i += 256) img[i] = ((sf[i] + sf[kC3Image + i]) + sf[2 * kC3Image + i]) + sf[3 * kC3Image + i];
}

// gw[co][ci][ky][kx] (fp32, [cout][3][5][5]) from the workgroups' fragment-order images.  An image is only 1 792 entries: a workgroup
// owns 16 of them (4 lanes x float4) and splits the images over 128 slices (slice s adds images s, s + 128, ... in that order); then
// 64 lanes add 32 slice sums each in slice order and 16 lanes the four results — a fixed order, 112 workgroups.
__global__ __launch_bounds__(512) void conv5c3_wrw_reduce_kernel(const float *__restrict__ partial, int nwg, int cout, float *__restrict__ gw) {
  __shared__ f32x4 part[128][4];
  __shared__ float part2[4][16];
  const int q = threadIdx.x & 3, sl = threadIdx.x >> 2;
  const size_t i4 = (size_t)blockIdx.x * 16 + q * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int w = sl; w < nwg; w += 128) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(partial + (size_t)w * kC3Image + i4));
  part[sl][q] = s;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int en = threadIdx.x & 15, sub = threadIdx.x >> 4;
    const float *p = reinterpret_cast<const float *>(&part[sub * 32][0]) + en;
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) r += p[k * 16];
    part2[sub][en] = r;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const float r = ((part2[0][threadIdx.x] + part2[1][threadIdx.x]) + part2[2][threadIdx.x]) + part2[3][threadIdx.x];
    const int i = blockIdx.x * 16 + threadIdx.x;                 // (nb, lane, e)
    const int e = i & 3, lane = (i >> 2) & 63, nb = i >> 8;
    const int co = 4 * (lane >> 4) + e, tap = 4 * nb + ((lane & 15) >> 2), ci = lane & 3;
    if (co < cout && tap < kTaps && ci < 3) gw[((size_t)co * 3 + ci) * kTaps + tap] = r;
  }
}

template <int COUT, int STRIDE>
int conv5c3_wrw_launch(const Conv5WrwArgs &a, float *gw, hipStream_t s) {
  constexpr int kPH = (kWR - 1) * STRIDE + 5, kPW = (kWC - 1) * STRIDE + 5;
  constexpr int kStage = (kPH * kPW * 8 + 15) / 16 * 16 + kWR * kWC * (COUT == 16 ? 32 : 8) + 256;
  constexpr int kLds = kStage > 4 * kC3Image * 4 ? kStage : 4 * kC3Image * 4;
  const int tiles = a.B * a.tiles_x * a.tiles_y;
  const int grid = tiles < kC3Wgs ? tiles : kC3Wgs;
  A3VT_LAUNCH((conv5c3_wrw_kernel<COUT, STRIDE>), dim3(grid), dim3(256), kLds, s, a);
  A3VT_CHECK_LAUNCH();
  A3VT_LAUNCH(conv5c3_wrw_reduce_kernel, dim3(kC3Image / 16), dim3(512), 0, s, (const float *)a.partial, grid, COUT, gw);
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <int CIN, int COUT, int STRIDE>
int conv5_wrw_launch(const Conv5WrwArgs &a, float *gw, hipStream_t s) {
  constexpr int kPH = (kWR - 1) * STRIDE + 5, kPW = (kWC - 1) * STRIDE + 5;
  constexpr int kLds = 2 * (kWR * kWC * COUT * 2 + kPH * kPW * CIN * 2);
  static OncePerDevice once;
  once.run([] { (void)hipFuncSetAttribute((const void *)conv5_wrw_kernel<CIN, COUT, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds); });
  const int tiles = a.B * a.tiles_x * a.tiles_y;
  // 16 -> 16 (44 KB of LDS: three workgroups per CU) takes two workgroups per CU, the other two shapes (87 / 114 KB) one: measured
  // 23.5 / 23.2 / 18.4 us at the configs[3] maps, against 33.4 / 25.8 / 20.9 with the other choice
  constexpr int cap = CIN == 16 && COUT == 16 ? kWrwWgs : kWrwWgs / 2;
  const int grid = tiles < cap ? tiles : cap;
  A3VT_LAUNCH((conv5_wrw_kernel<CIN, COUT, STRIDE>), dim3(grid), dim3(320), kLds, s, a);
  A3VT_CHECK_LAUNCH();
  const int n = kTaps * COUT * CIN;
  A3VT_LAUNCH(conv5_wrw_reduce_kernel, dim3(n / 64), dim3(16 * kRedSlices), 0, s, (const float *)a.partial, grid, COUT, CIN, gw);
  A3VT_CHECK_LAUNCH();
  return 0;
}

// the shapes taken: (cin, cout, stride) in {(16,16,1), (32,32,1), (16,32,2)} — layers 2-3, 5-6 and 4 of the pyramid; with stride 1 the
// input gradient is the same kernel (weights flipped, padding 3) — and the 3-channel layers 0 and 1, (3,3,1) and (3,16,2), forward only
bool conv5_shape_ok(int cin, int cout, int stride) {
  return (cin == 16 && cout == 16 && stride == 1) || (cin == 32 && cout == 32 && stride == 1) || (cin == 16 && cout == 32 && stride == 2) ||
         (cin == 3 && cout == 3 && stride == 1) || (cin == 3 && cout == 16 && stride == 2);
}

size_t conv5_weight_image_bytes(int rows, int inner) {
  if (inner == 3) return (size_t)4 * 64 * 8 * sizeof(u16);
  rows = rows < 16 ? 16 : rows;
  return (size_t)(rows / 16) * (inner == 16 ? 13 : 25) * 64 * 8 * sizeof(u16);
}

int launch_conv5_weight_image(const float *w, int flip, int cout, int cin, void *image, hipStream_t s) {
  if (cin == 3 && !flip) {
    A3VT_LAUNCH(conv5c3_weight_image_kernel, dim3(8), dim3(256), 0, s, w, cout, static_cast<u16 *>(image));
    A3VT_CHECK_LAUNCH();
    return 0;
  }
  // (flipped image of layer 1, cin = 3, cout = 16: sixteen rows of which three are real)
  const int real_rows = flip ? cin : cout, rows = real_rows < 16 ? 16 : real_rows, inner = flip ? cout : cin;
  const int n = (rows / 16) * (inner == 16 ? 13 : 25) * 64 * 8;
  A3VT_LAUNCH(conv5_weight_image_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, flip, rows, inner, real_rows, static_cast<u16 *>(image));
  A3VT_CHECK_LAUNCH();
  return 0;
}

// the input gradient of layer 1 (3 <- 16, stride 2): gy [batch][ho][wo][16] -> gx [batch][2 ho + 2][2 wo + 2][3] (image: flip = 1 of the
// (16, 3, 5, 5) weight)
int launch_conv5_up3(const void *gy, int batch, int ho, int wo, const void *image, void *gx, hipStream_t s) {
  Conv5Args a{};
  a.x = static_cast<const u16 *>(gy);
  a.y = static_cast<u16 *>(gx);
  a.wfrag = static_cast<const u16 *>(image);
  a.B = batch;
  a.H = 2 * ho;
  a.W = 2 * wo;
  a.pad = 3;
  a.Ho = a.H + 2;
  a.Wo = a.W + 2;
  a.tiles_x = (a.Wo + kTile - 1) / kTile;
  a.tiles_y = (a.Ho + kTile - 1) / kTile;
  return conv5_launch<16, 16, 1, true>(a, s);
}

int launch_conv5(const void *x, int batch, int h, int w, int cin, int cout, int stride, int pad, const void *image, const float *bias,
                 void *y, hipStream_t s) {
  Conv5Args a{};
  a.x = static_cast<const u16 *>(x);
  a.y = static_cast<u16 *>(y);
  a.wfrag = static_cast<const u16 *>(image);
  a.bias = bias;
  a.B = batch;
  a.H = h;
  a.W = w;
  a.pad = pad;
  a.Ho = (h + 2 * pad - 5) / stride + 1;
  a.Wo = (w + 2 * pad - 5) / stride + 1;
  a.tiles_x = (a.Wo + kTile - 1) / kTile;
  a.tiles_y = (a.Ho + kTile - 1) / kTile;
  if (cin == 16 && cout == 16 && stride == 1) return conv5_launch<16, 16, 1>(a, s);
  if (cin == 32 && cout == 32 && stride == 1) return conv5_launch<32, 32, 1>(a, s);
  if (cin == 16 && cout == 32 && stride == 2) return conv5_launch<16, 32, 2>(a, s);
  if (cin == 3 && cout == 3 && stride == 1) return conv5c3_launch<3, 1>(a, s);
  if (cin == 3 && cout == 16 && stride == 2) return conv5c3_launch<16, 2>(a, s);
  set_error("conv5: shape %d -> %d stride %d not taken", cin, cout, stride);
  return -1;
}

size_t conv5_wrw_scratch_bytes(int cin, int cout) {
  return cin == 3 ? (size_t)kC3Wgs * kC3Image * sizeof(float) : (size_t)kWrwWgs * kTaps * cout * cin * sizeof(float);
}

// weight gradient of the (3,3,1), (3,16,2), (16,16,1), (32,32,1), (16,32,2) layers: x [B][h][w][cin], gy [B][ho][wo][cout] bf16 -> gw fp32 [cout][cin][5][5]
int launch_conv5_wrw(const void *x, const void *gy, int batch, int h, int w, int cin, int cout, int stride, float *gw, void *scratch,
                     hipStream_t s) {
  Conv5WrwArgs a{};
  a.x = static_cast<const u16 *>(x);
  a.gy = static_cast<const u16 *>(gy);
  a.partial = static_cast<float *>(scratch);
  a.B = batch;
  a.H = h;
  a.W = w;
  a.Ho = (h + 2 - 5) / stride + 1;
  a.Wo = (w + 2 - 5) / stride + 1;
  a.tiles_x = (a.Wo + kWC - 1) / kWC;
  a.tiles_y = (a.Ho + kWR - 1) / kWR;
  if (cin == 3 && cout == 3 && stride == 1) return conv5c3_wrw_launch<3, 1>(a, gw, s);
  if (cin == 3 && cout == 16 && stride == 2) return conv5c3_wrw_launch<16, 2>(a, gw, s);
  if (cin == 16 && cout == 16 && stride == 1) return conv5_wrw_launch<16, 16, 1>(a, gw, s);
  if (cin == 32 && cout == 32 && stride == 1) return conv5_wrw_launch<32, 32, 1>(a, gw, s);
  if (cin == 16 && cout == 32 && stride == 2) return conv5_wrw_launch<16, 32, 2>(a, gw, s);
  set_error("conv5_wrw: shape %d -> %d stride %d not taken", cin, cout, stride);
  return -1;
}

}  // namespace a3vt
