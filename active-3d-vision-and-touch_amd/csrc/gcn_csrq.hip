// gcn_csrq.hip — channel-sliced neighbour aggregation of the GCN stack, whole mesh resident in LDS (gfx950, round 3).
//
// Replaces the dense (N,N) adjacency products of reconstruction/vision/model.py:356,360 (and autograd's A^T products)
// for the hidden layers of a stack, fused with the partial bias add / ReLU of model.py:357-358,363.  The half-wave-per-
// vertex kernels of gcn_csr.hip remain for the stand-alone layer, for few-row calls and for meshes that do not fit.
// Compiled with -fno-slp-vectorize (lib.py): the sums below are written per component on purpose — as <4 x float> the
// compiler forms v_pk_fma_f32 and keeps every edge weight as a (w, w) register pair, which doubles the registers of the
// index entries a thread holds and spills them; packed fp32 has no throughput advantage on gfx950.
#include "common.h"
#include "kernels.h"

namespace a3vt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------------------------------------
// Channel-sliced aggregation with the whole mesh resident in LDS (round 3; the stack's default when it fits).
//
// The half-wave-per-vertex kernels above re-gather ~7 neighbour rows per output row through the vector-memory path
// (585 MB per launch for 130 MB of algorithmic traffic) and sit at 27-31 % of the HBM rate.  Here the work is cut along
// CHANNELS instead of vertices: a workgroup owns (mesh b, channel quad q) — 4 of the aggregated channels for ALL n_vert
// vertices of one mesh — streams that slice into LDS once (n_vert x 16 B = 41 KB on the 2562-vertex icosphere, three
// workgroups per CU so one streams while another gathers), and every vertex then takes its neighbours from LDS with one
// ds_read_b128 each: no halo, no redundant global read, each input byte fetched once.  The producers write the slice-major
// ("quad-major") arrays directly: src[((b * Q + q) * n_vert + v)] as float4, Q = cpad / 4 —
//   forward : raw Z of the aggregated channels, from the EPI_FWD_HIDDEN epilogue of rowgemm (RowGemmArgs::zq_nvert)
//   backward: the UNMASKED gradient columns [0, cpad) from the EPI_DX_MASK epilogue / thin_bwd; the ReLU signs of the
//             aggregated channels never leave this pair of kernels: csrq<0> writes them quad-major (one byte per vertex
//             and quad, coalesced), csrq<1> applies them while it fills LDS.
// Outputs are quad-major too — the activations' / dZa's columns [0, 4 Q) live in that layout for the whole stack and the
// consumers (rowgemm's A operand, dw's X / dZa images, the output layer) fetch their 16-byte pieces from it: written
// row-major, as 16-byte pieces of 1200- / 400-byte rows, the stores alone cost 45 us per launch (measured; 35 us with
// quad-major stores, profiles/r03_csrq_ablation.txt).
// Arithmetic and summation order per output element are those of csr_fwd / csr_bwd (acc += w * r in CSR order).
// ------------------------------------------------------------------------------------------------
constexpr int kCsrqThreads = 512;
constexpr int kHeavyDegQ = 64;   // = csr_heavy_degree(): rows listed by csr_heavy_list_kernel
constexpr int kEllW = 8;      // edge slots per vertex in the slot-major index image (icosphere rows have 6-7 edges)

// Slot-major ("ELL") image of the first kEllW edges of every row: ell[j * n_vert + v] = (col, weight bits) of edge j of
// row v; past the row's end (n_vert, +0.0f) — column n_vert is a row of zeros the kernel keeps behind each slice, so an empty
// slot adds +0 * 0 and needs no predicate (48 loop-invariant lane masks spilled the scalar registers); degs[v] = the row's
// edge count.  With one lane per vertex the CSR arrays themselves
// are read at a 28-byte lane stride, i.e. every 128-byte line of col / val is touched once per edge slot (8.6 line
// accesses per line); here consecutive lanes read consecutive 8-byte entries.  Built once per stack call
// (n_vert x 17 words); edges beyond kEllW stay in CSR.
using i32x2 = __attribute__((ext_vector_type(2))) int;
// by-value helpers: __builtin_bit_cast applied DIRECTLY to an element of an ext_vector reads element 0 with this hipcc
// ("a compiler trap worth recording") — the first version of this kernel multiplied by the column index
__device__ __forceinline__ float bits_to_f32(int x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ int f32_to_bits(float x) { return __builtin_bit_cast(int, x); }
__global__ void csr_ell_build_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                     const float *__restrict__ val, int n_vert, i32x2 *__restrict__ ell,
                                     int32_t *__restrict__ degs) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vert) return;
  const int e0 = rowptr[v], n = rowptr[v + 1] - e0;
  degs[v] = n;
#pragma unroll
  for (int j = 0; j < kEllW; ++j)
    ell[(size_t)j * n_vert + v] = j < n ? i32x2{colidx[e0 + j], f32_to_bits(val[e0 + j])} : i32x2{n_vert, 0};
}
size_t csrq_ell_ints(int n_vert) { return (size_t)n_vert * (2 * kEllW + 1) + 64; }
int launch_csrq_ell(const int32_t *rowptr, const int32_t *col, const float *val, int n_vert, int32_t *ell, hipStream_t s) {
  // layout: [kEllW][n_vert] pairs (8-byte aligned: the caller's region is 256-B aligned), then degs[n_vert]
  A3VT_LAUNCH(csr_ell_build_kernel, dim3(cdiv(n_vert, 256)), dim3(256), 0, s, rowptr, col, val, n_vert,
              reinterpret_cast<i32x2 *>(ell), ell + (size_t)n_vert * 2 * kEllW);
  A3VT_CHECK_LAUNCH();
  return 0;
}

#ifdef A3VT_DBG_CSRQ_STAMPS   // diagnostic build (tools/build_variants.sh stampsq): s_memrealtime (100 MHz) + s_memtime per quad
__device__ unsigned long long g_csrq_stamps[2 * 2 * 256 * 8 * 4];   // [real time | shader cycles][MODE][workgroup][quad (< 8)][4]
#define CSRQ_STAMP(qi, k)                                                                                     \
  do {                                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 256 && (qi) < 8) {                                                   \
      const int i_ = ((MODE * 256 + blockIdx.x) * 8 + (qi)) * 4 + (k);                                        \
      g_csrq_stamps[i_] = __builtin_amdgcn_s_memrealtime();                                                   \
      g_csrq_stamps[2 * 256 * 8 * 4 + i_] = __builtin_amdgcn_s_memtime();                                     \
    }                                                                                                         \
  } while (0)
#else
#define CSRQ_STAMP(qi, k) do { } while (0)
#endif

// One persistent workgroup per (mesh, part of the mesh's quads): a thread owns VPT vertices and keeps THEIR index entries
// (column, weight) in registers for the whole launch — the index image is read once per workgroup, not once per quad (read
// per quad it was twice the bytes of the data itself: 174 KB against 41 + 41 KB per slice, 46 us per launch even with
// perfectly coalesced stores) — and walks its quads with the slice double-buffered in LDS: the next quad's slice is
// requested before the current one is gathered and written to the other buffer behind the gather; one barrier per quad.
template <int MODE, int VPT>
__global__ __launch_bounds__(kCsrqThreads) void csrq_kernel(const float *__restrict__ srcq, int nq, int parts,
                                                            const float *__restrict__ bias, int c,
                                                            const int32_t *__restrict__ rowptr,
                                                            const int32_t *__restrict__ colidx,
                                                            const float *__restrict__ val, int n_vert, int batch,
                                                            float *__restrict__ dst, int nq_dst,
                                                            uint8_t *__restrict__ signq, int relu, int heavy_thresh,
                                                            float *__restrict__ db_slab,
                                                            const int32_t *__restrict__ ellbuf) {
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  __shared__ float red[2][kCsrqThreads / 64][4];
  const i32x2 *ell = reinterpret_cast<const i32x2 *>(ellbuf);
  const int32_t *degs = ellbuf + (size_t)n_vert * 2 * kEllW;
  f32x4 *tile0 = reinterpret_cast<f32x4 *>(lds_raw);
  // the parts of a mesh share an XCD (blockIdx % 8): their 16-byte output pieces of a cache line meet in one L2
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int jm = loc / parts, part = loc - jm * parts;
  const int b = xcd + 8 * jm;
  if (b >= batch) return;   // uniform per workgroup
  const int q_lo = part * nq / parts, q_hi = (part + 1) * nq / parts;
  if (q_lo >= q_hi) return;
  CSRQ_STAMP(7, 0);   // kernel entry (slot 7 is never a quad: a workgroup walks at most seven)

  // this thread's vertices: index entries and edge counts, held for all quads.  Threads past the mesh's end work on a
  // copy of the last vertex and skip only the stores: branch-free loads (per-element `v < n_vert ?` guards became 280
  // basic blocks and spilled the index entries).
  int vv[VPT], dg[VPT];
  bool on[VPT];
  int cj[VPT][kEllW];
  float wj[VPT][kEllW];
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int v = threadIdx.x + k * kCsrqThreads;
    on[k] = v < n_vert;
    vv[k] = on[k] ? v : n_vert - 1;
#ifdef A3VT_DBG_CSRQ_NOINDEX   // timing-only: no index image; every lane gathers rows at fixed offsets from its own, so the
    dg[k] = 6;                 // 16 lanes of a ds_read_b128 group hit 16 different 16-byte slots: conflict-free gathers
#pragma unroll
    for (int j = 0; j < kEllW; ++j) cj[k][j] = (vv[k] + j * 37) % n_vert, wj[k][j] = 0.125f;
#else
    dg[k] = degs[vv[k]];
#pragma unroll
    for (int j = 0; j < kEllW; ++j) {
      const i32x2 e = ell[(size_t)j * n_vert + vv[k]];
      const int e0 = e[0], e1 = e[1];
      cj[k][j] = e0;
      wj[k][j] = bits_to_f32(e1);
    }
#endif
  }

  // a slice element on its way to LDS: the gradient passes the ReLU of the aggregated channels here (MODE 1)
  auto fetch = [&](int q, int k, f32x4 &v4, unsigned &bits) {
    const size_t plane = ((size_t)b * nq + q) * n_vert;
    v4 = reinterpret_cast<const f32x4 *>(srcq)[plane + vv[k]];
    bits = MODE == 1 ? signq[plane + vv[k]] : 0u;
  };
  auto park = [&](f32x4 *tile, int q, int k, f32x4 v4, unsigned bits) {
    if (MODE == 1) {   // all four channels of the quad: the forward kept a sign bit for the pass-through ones too
#pragma unroll
      for (int t = 0; t < 4; ++t) v4[t] = ((bits >> t) & 1u) ? v4[t] : 0.f;
    }
    (void)q;
    tile[vv[k]] = v4;   // (the copies of the last vertex all write the same value)
  };

  if (threadIdx.x < 2) tile0[(size_t)threadIdx.x * (n_vert + 1) + n_vert] = f32x4{0.f, 0.f, 0.f, 0.f};   // the rows of zeros
  {
    f32x4 v4[VPT];
    unsigned bits[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) fetch(q_lo, k, v4[k], bits[k]);
#pragma unroll
    for (int k = 0; k < VPT; ++k) park(tile0, q_lo, k, v4[k], bits[k]);
  }
  __syncthreads();

  for (int q = q_lo; q < q_hi; ++q) {
    CSRQ_STAMP(q - q_lo, 0);
    const int par = (q - q_lo) & 1;
    const f32x4 *tile = tile0 + (size_t)par * (n_vert + 1);
    f32x4 *tnext = tile0 + (size_t)(par ^ 1) * (n_vert + 1);
    const int ch = q * 4;
    const size_t plane = ((size_t)b * nq + q) * n_vert;
    const int qn = q + 1 < q_hi ? q + 1 : q;   // last quad: re-fetch itself (parked nowhere)
    // next slice: requested now, parked behind the gather
    f32x4 nv4[VPT];
    unsigned nbits[VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) fetch(qn, k, nv4[k], nbits[k]);
    f32x4 bs = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (ch + t < c) bs[t] = bias[ch + t];
    }
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      // one vertex at a time (the scheduler would otherwise hoist the LDS gathers of all the thread's vertices: 48 x 4 registers)
      __builtin_amdgcn_sched_barrier(0);
      const int v = vv[k];
      const f32x4 own = tile[v];
      if (MODE == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bsum[t] += on[k] ? own[t] : 0.f;
      }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < kEllW; ++j) {
#ifdef A3VT_DBG_CSRQ_NOGATHER   // timing-only: no LDS gather (the index entries stay live)
        acc[j & 3] += wj[k][j] * (float)cj[k][j];
#else
        {   // empty slots: + 0 * (row of zeros)
          const f32x4 r = tile[cj[k][j]];
          const float w = wj[k][j];
          acc[0] = __builtin_fmaf(w, r[0], acc[0]);
          acc[1] = __builtin_fmaf(w, r[1], acc[1]);
          acc[2] = __builtin_fmaf(w, r[2], acc[2]);
          acc[3] = __builtin_fmaf(w, r[3], acc[3]);
        }
#endif
      }
      if (dg[k] > kEllW && dg[k] <= heavy_thresh) {   // the rest of a long row, from the CSR arrays (same order)
        const int ea = rowptr[v];
        for (int e = ea + kEllW; e < ea + dg[k]; ++e) {
          const f32x4 r = tile[colidx[e]];
          const float w = val[e];
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_fmaf(w, r[t], acc[t]);
        }
      }
      if (!on[k] || dg[k] > heavy_thresh) continue;   // hub rows: csrq_heavy_kernel (their bias share is counted above)
      f32x4 o;
      if (MODE == 0) {
        unsigned bits = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // aggregated channels: neighbour sum + bias; the pass-through channels of the last quad (model.py:358: no
          // bias): the raw value itself
          const float pre = ch + t < c ? acc[t] + bs[t] : own[t];
          o[t] = (pre > 0.f || !relu) ? pre : 0.f;
          bits |= (pre > 0.f ? 1u : 0u) << t;
        }
        if (signq) signq[plane + v] = (uint8_t)bits;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) o[t] = ch + t < c ? acc[t] : own[t];
      }
      // quad-major like the input (1 KiB per wave-instruction); the output array may hold more planes per mesh (nq_dst)
      // (streaming stores — which take 2.7 us off the bf16 tiled aggregation — were measured here and change nothing: 35.1 / 37.0 -> 34.5 / 37.9 us)
      *reinterpret_cast<f32x4 *>(dst + ((((size_t)b * nq_dst + q) * n_vert) + v) * 4) = o;
    }
    __builtin_amdgcn_sched_barrier(0);
    CSRQ_STAMP(q - q_lo, 1);
    if (MODE == 1) {   // bias-gradient partial of this (mesh, quad), fixed order; red[par] is read behind the barrier below
#pragma unroll
      for (int t = 0; t < 4; ++t) bsum[t] = wave_sum(bsum[t]);
      if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) red[par][threadIdx.x >> 6][t] = bsum[t];
      }
    }
    if (q + 1 < q_hi) {
#pragma unroll
      for (int k = 0; k < VPT; ++k) park(tnext, q + 1, k, nv4[k], nbits[k]);
    }
    CSRQ_STAMP(q - q_lo, 2);
    __syncthreads();   // next slice visible; everyone is done with this one (and with red[par ^ 1] of the previous quad)
    if (MODE == 1 && threadIdx.x < 4) {
      float sacc = 0.f;
#pragma unroll
      for (int w = 0; w < kCsrqThreads / 64; ++w) sacc += red[par][w][threadIdx.x];
      db_slab[(size_t)b * (nq * 4) + ch + threadIdx.x] = sacc;
    }
    CSRQ_STAMP(q - q_lo, 3);
  }
}

// Hub rows on the quad-major arrays: a workgroup per (mesh, hub row), its 8 half-waves take an eighth of the edge list
// each, lane hl owns quad hl (up to 32 quads = 128 aggregated channels).
template <int MODE>
__global__ __launch_bounds__(256) void csrq_heavy_kernel(const float *__restrict__ srcq, int nq,
                                                         const float *__restrict__ bias, int c,
                                                         const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ colidx,
                                                         const float *__restrict__ val, int n_vert, int batch,
                                                         const int32_t *__restrict__ heavy, float *__restrict__ dst,
                                                         int nq_dst, uint8_t *__restrict__ signq, int relu) {
  __shared__ f32x4 red[8][32];
  const int hl = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int count = heavy[0];
  const bool on = hl < nq;
  const int ch = hl * 4;
  for (long long item = blockIdx.x; item < (long long)count * batch; item += gridDim.x) {
    const int v = heavy[64 + (int)(item % count)];
    const long long b = item / count;
    const size_t plane = ((size_t)b * nq + (on ? hl : 0)) * n_vert;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(srcq) + plane;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    const int per = ((e1 - e0 + 7) / 8 + 3) & ~3;
    const int s0 = min(e1, e0 + sub * per), s1 = min(e1, s0 + per);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (on) {
      for (int e = s0; e < s1; ++e) {   // same order inside a segment as gather_row
        const int cc = colidx[e];
        f32x4 r = src[cc];
        if (MODE == 1) {
          const unsigned bits = signq[plane + cc];
#pragma unroll
          for (int t = 0; t < 4; ++t) r[t] = ((bits >> t) & 1u) ? r[t] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_fmaf(val[e], r[t], acc[t]);
      }
    }
    red[sub][hl] = acc;
    __syncthreads();
    if (sub == 0 && on) {
      f32x4 a = red[0][hl];
#pragma unroll
      for (int r = 1; r < 8; ++r) a += red[r][hl];
      f32x4 own = src[v];   // columns >= c pass the row's own value through
      f32x4 o;
      if (MODE == 0) {
        unsigned bits = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float pre = ch + t < c ? a[t] + bias[ch + t] : own[t];
          o[t] = (pre > 0.f || !relu) ? pre : 0.f;
          bits |= (pre > 0.f ? 1u : 0u) << t;
        }
        if (signq) signq[plane + v] = (uint8_t)bits;
      } else {
        const unsigned bits = signq[plane + v];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          own[t] = ((bits >> t) & 1u) ? own[t] : 0.f;
          o[t] = ch + t < c ? a[t] : own[t];
        }
      }
      *reinterpret_cast<f32x4 *>(dst + ((((size_t)b * nq_dst + hl) * n_vert) + v) * 4) = o;
    }
    __syncthreads();
  }
}

int csrq_max_degree() { return kEllW; }
#ifdef A3VT_DBG_CSRQ_STAMPS
}  // namespace a3vt
extern "C" int a3vt_dbg_csrq_stamps(unsigned long long *host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(a3vt::g_csrq_stamps), sizeof(unsigned long long) * 2 * 2 * 256 * 8 * 4);
}
namespace a3vt {
#endif
bool csrq_fits(int n_vert, int cut_len) {
  // two slices of one mesh in LDS; a thread owns at most 6 vertices (their index entries stay in registers); hub lanes
  // cover up to 32 quads
  return cut_len > 0 && pad4(cut_len) <= 128 && n_vert <= 6 * kCsrqThreads && ((size_t)n_vert + 1) * 32 <= 158 * 1024;
}

template <int MODE, int VPT>
static int launch_csrq_vpt(const float *srcq, int nq, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                           const float *val, const int32_t *heavy, const int32_t *ell, int n_vert, int batch, float *dst,
                           int nq_dst, uint8_t *signq, int relu, float *db_slab, hipStream_t s) {
  const size_t shmem = ((size_t)n_vert + 1) * 32;
  static OncePerDevice once;
  once.run([] {
    (void)hipFuncSetAttribute((const void *)csrq_kernel<MODE, VPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  });
  // one workgroup per CU when the batch allows it: parts of a mesh = 256 / batch (at least 1, at most one per quad)
  int parts = 256 / (((batch + 7) / 8) * 8);
  parts = parts < 1 ? 1 : parts > nq ? nq : parts;
  const int grid = 8 * ((batch + 7) / 8) * parts;   // (XCD group, mesh of the group, part): the parts of a mesh share an XCD
  A3VT_LAUNCH((csrq_kernel<MODE, VPT>), dim3(grid), dim3(kCsrqThreads), shmem, s, srcq, nq, parts, bias, c, rowptr, col, val,
              n_vert, batch, dst, nq_dst, signq, relu, heavy ? kHeavyDegQ : 0x7fffffff, db_slab, ell);
  A3VT_CHECK_LAUNCH();
  return 0;
}

template <int MODE>
static int launch_csrq(const float *srcq, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                       const float *val, const int32_t *heavy, const int32_t *ell, int n_vert, int batch, float *dst,
                       int nq_dst, uint8_t *signq, int relu, float *db_slab, hipStream_t s) {
  const int nq = pad4(c) / 4;
  if (!csrq_fits(n_vert, c) || nq_dst < nq || (MODE == 1 && (!signq || !db_slab))) {
    set_error("csrq: n_vert=%d c=%d output planes=%d unsupported", n_vert, c, nq_dst);
    return -1;
  }
  int rc;
  if (n_vert <= 4 * kCsrqThreads)
    rc = launch_csrq_vpt<MODE, 4>(srcq, nq, bias, c, rowptr, col, val, heavy, ell, n_vert, batch, dst, nq_dst, signq, relu, db_slab, s);
  else
    rc = launch_csrq_vpt<MODE, 6>(srcq, nq, bias, c, rowptr, col, val, heavy, ell, n_vert, batch, dst, nq_dst, signq, relu, db_slab, s);
  if (rc) return rc;
  if (heavy) {
    A3VT_LAUNCH(csrq_heavy_kernel<MODE>, dim3(2048), dim3(256), 0, s, srcq, nq, bias, c, rowptr, col, val, n_vert, batch,
                heavy, dst, nq_dst, signq, relu);
    A3VT_CHECK_LAUNCH();
  }
  return 0;
}

int launch_csrq_fwd(const float *zq, const float *bias, int c, const int32_t *rowptr, const int32_t *col,
                    const float *val, const int32_t *heavy, const int32_t *ell, int n_vert, int batch, float *yq,
                    int yq_quads, uint8_t *signq, int relu, hipStream_t s) {
  return launch_csrq<0>(zq, bias, c, rowptr, col, val, heavy, ell, n_vert, batch, yq, yq_quads, signq, relu, nullptr, s);
}
int launch_csrq_bwd(const float *gq, int c, const int32_t *rowptrT, const int32_t *colT, const float *valT,
                    const int32_t *heavyT, const int32_t *ellT, int n_vert, int batch, float *dzaq,
                    const uint8_t *signq, float *db_slab, hipStream_t s) {
  return launch_csrq<1>(gq, nullptr, c, rowptrT, colT, valT, heavyT, ellT, n_vert, batch, dzaq, pad4(c) / 4,
                        const_cast<uint8_t *>(signq), 0, db_slab, s);
}

}  // namespace a3vt
