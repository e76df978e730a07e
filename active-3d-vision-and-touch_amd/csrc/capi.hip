// capi.hip — extern "C" entry points declared in include/a3vt.h, plus the GCN stack orchestration
// (the layer loop of reconstruction/vision/model.py:316-331 lives here so that Python makes one call
// per refinement stage and nothing syncs the host).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/a3vt.h"
#include <atomic>
#include <mutex>
#include <unordered_map>

#include "common.h"
#include "kernels.h"

namespace a3vt {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// 256 bytes of zeros in the code object: source for out-of-range LDS-DMA lanes.
__device__ float g_zero_page[64];
static const float *zero_page() {  // the symbol has one address per device
  static const float *ptr[64] = {};
  static std::mutex mu;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  const float *&slot = ptr[dev & 63];
  if (!slot) {
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_page)) != hipSuccess) return nullptr;
    slot = static_cast<const float *>(p);
  }
  return slot;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when on, every MFMA launch of the GCN stack is bracketed by an event pair.
enum { PROF_GEMM_FWD = 0, PROF_GEMM_DX = 1, PROF_DW = 2, PROF_AGG = 3, PROF_OUT = 4, PROF_SEARCH = 5, PROF_LOSS = 6, PROF_ENC = 7,
       PROF_OPT = 8, PROF_CLASSES = 9 };   // a3vt_profile_read: the first three; a3vt_profile_read_classes: all
struct Prof {
  bool on = false;
  static constexpr int kMax = 8192;
  hipEvent_t ev[kMax][2];
  int cls[kMax];
  int created = 0, used = 0;
};
static Prof g_prof;

struct ProfScope {
  int slot = -1;
  hipStream_t s;
  ProfScope(int cls, hipStream_t stream) : s(stream) {
    if (!g_prof.on || g_prof.used >= Prof::kMax) return;
    slot = g_prof.used++;
    if (slot >= g_prof.created) {
      (void)hipEventCreate(&g_prof.ev[slot][0]);
      (void)hipEventCreate(&g_prof.ev[slot][1]);
      g_prof.created = slot + 1;
    }
    g_prof.cls[slot] = cls;
    (void)hipEventRecord(g_prof.ev[slot][0], s);
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[slot][1], s);
  }
};

// Scratch layout of a GCN stack call (float offsets, every region 256-B aligned).
struct StackLayout {
  size_t wt, wt_stride;  // per hidden layer: transposed (fwd) or padded (bwd) weight image
  size_t za;             // [M][cpad]
  size_t z3;             // [2][M][4]
  size_t ping[2];        // [M][hidden] each (fwd without acts: layer outputs; bwd: gradients)
  size_t panel;          // bwd, inputs wider than 304 columns only: [M][300] contiguous column block of X_0 for dW_0
  size_t dw_slab, db_slab, thin_dw_slab, thin_db_slab;
  size_t heavy;          // int32 list of hub rows (csr_heavy_scratch_ints)
  size_t gq;             // bwd: quad-major gradient columns [0, cpad) for the channel-sliced aggregation, [M][cpad]
  size_t ell;            // slot-major index image of the adjacency (csrq_ell_ints)
  size_t total;
};

// Channel-sliced aggregation (gcn_csr.hip "csrq") for the hidden layers of a stack: the mesh slice must fit LDS and the
// product kernel must run the shape as one column block so that its epilogues can write quad-major.  The forward decides
// and records the decision with the stash (stash_layout_*); the backward follows the record.  a3vt_dbg_csr_algo() is the
// test hook that forces either path (both give the same outputs; the shipped library reads no environment variable).
constexpr int kQuadCols = 160;   // quad-major columns of a hybrid layer output: the first column group of rowgemm's epilogue
static std::atomic<long long> g_path_counts[PATH_COUNT];
void path_count(int which) { g_path_counts[which].fetch_add(1, std::memory_order_relaxed); }
static int g_csr_algo = 0;       // 0 = by shape, 1 = half-wave ("rows"), 2 = channel-sliced where it fits, long rows included
// A split the channel-sliced kernels of gcn_csrqs.hip take (a3vt_adj_split; P short enough for the slots a thread keeps)
static bool split_usable(const a3vt_adj_split *sp, int n_vert, int cut_len) {
  return sp && sp->rowptr && sp->col && sp->scale && sp->cls && sp->max_degree > 0 && sp->max_degree <= csrqs_max_degree() &&
         csrqs_fits(n_vert, cut_len);
}
static SplitRef split_ref(const a3vt_adj_split *sp) {
  return sp ? SplitRef{sp->rowptr, sp->col, sp->scale, sp->cls} : SplitRef{nullptr, nullptr, nullptr, nullptr};
}
static bool use_csrq(int batch, int n_vert, int hidden, int cut_len, int gemm_bf16, int max_degree) {
  if (gemm_bf16 == 1 || gemm_bf16 == 2) return false;   // the bf16 operand / storage modes keep the half-wave kernels (mode 3 stores fp32: as mode 0)
  // Rows longer than the eight index slots a thread keeps in registers fall back to per-lane CSR walks: on the fused
  // vision + touch graphs (mean degree 12-26, hub rows of ~1150) that made the step 108 ms where the half-wave kernels
  // take 64 — those graphs stay on the half-wave kernels unless the test hook forces them (they are correct, just slow).
  if (g_csr_algo != 2 && (max_degree <= 0 || max_degree > csrq_max_degree())) return false;
  if (g_csr_algo == 1 || cut_len <= 0) return false;
  return hidden % 4 == 0 && hidden >= kQuadCols + 16 && csrq_fits(n_vert, cut_len) && dw_quad_major_ok(hidden, kQuadCols / 4) &&
         rowgemm_quad_major_ok(batch * n_vert, hidden, pad4(cut_len));
}
// Which layout a forward call left in a stash (activations + sign bytes), keyed by the sign-byte pointer: the backward
// call that receives the same stash adopts it instead of re-deriving it from its own max_degree hint (a C-ABI caller
// that passes csr_max_degree to the forward and csrT_max_degree to the backward would otherwise read hybrid rows as
// row-major ones, silently).  Host-side, a few bytes per live stash; cleared when it grows past 4096 entries.
static std::mutex g_stash_mu;
static std::unordered_map<const void *, int> g_stash_layout;
static void stash_layout_record(const void *masks, bool quad) {
  if (!masks) return;
  std::lock_guard<std::mutex> lock(g_stash_mu);
  if (g_stash_layout.size() > 4096) g_stash_layout.clear();
  g_stash_layout[masks] = quad ? 1 : 0;
}
static int stash_layout_lookup(const void *masks) {   // 1 / 0, or -1 when this library did not write that stash
  std::lock_guard<std::mutex> lock(g_stash_mu);
  auto it = g_stash_layout.find(masks);
  return it == g_stash_layout.end() ? -1 : it->second;
}
// sign bytes of the aggregated channels, quad-major [batch][Q][n_vert] per hidden layer, kept behind the row-major
// sign bytes in the caller's `masks` buffer
static inline size_t signq_stride(size_t m, int cut_len) { return align_up(m * (size_t)(pad4(cut_len) / 4), 256); }

// bias gradients of the hidden layers [0, last) from per-layer blocks of partial rows (layer i at slab + i * layer_stride)
static int reduce_bias_partials(const float *slab, size_t layer_stride, int nslab, int cpad, int cut_len, int hidden, int last,
                                float *const *grad_biases, int acc, hipStream_t s) {
  for (int i0 = 0; i0 < last; i0 += kMaxImages) {
    SlabReduceBatch b{};
    b.count = last - i0 < kMaxImages ? last - i0 : kMaxImages;
    b.slab = slab + (size_t)i0 * layer_stride;
    for (int j = 0; j < b.count; ++j) b.out[j] = grad_biases[i0 + j];
    b.layer_stride = layer_stride;
    b.stride = cpad;
    b.n = cut_len;
    b.n_out = hidden;
    b.nslab = nslab;
    b.accumulate = acc;
    if (int rc = launch_slab_reduce_batch(b, s)) return rc;
  }
  return 0;
}

static inline int mask_ld(int hidden, int cut_len) { return pad4(cut_len) / 4 + (hidden + 3) / 4; }

static StackLayout stack_layout(int batch, int n_vert, int in_features, int hidden, int num_layers, int cut_len,
                                int need_backward, int gemm_mode = 0) {
  StackLayout L{};
  const size_t m = (size_t)batch * n_vert;
  const int cpad = pad4(cut_len);
  size_t off = 0;
  auto take = [&](size_t nfloats) {
    const size_t o = off;
    off = align_up(off + nfloats, 64);
    return o;
  };
  const int kmax = hidden > pad4(in_features) ? hidden : pad4(in_features);
  L.wt_stride = align_up((size_t)rowgemm_bt_rows(kmax) * pad16(kmax), 64);
  if (gemm_mode == 3 && L.wt_stride < 3 * (size_t)kX3ImageFloats) L.wt_stride = 3 * (size_t)kX3ImageFloats;   // hi / mid / lo images (gcn_gemm3.hip)
  L.wt = take(L.wt_stride * (num_layers > 1 ? num_layers - 1 : 1));
  L.za = take(m * (cpad > 4 ? cpad : 4));
  L.z3 = take(2 * m * 4);
  L.ping[0] = take(m * hidden);
  L.ping[1] = take(m * hidden);
  L.heavy = take(csr_heavy_scratch_ints(n_vert));
  L.ell = take(csrq_ell_ints(n_vert));
  if (need_backward) {
    const size_t kin = kmax;
    // its own region: with hidden < 300 a ping buffer ([M][hidden]) is smaller than a 300-column block of X_0
    L.panel = take(pad4(in_features) > 304 ? m * 300 : 0);
    L.dw_slab = take((size_t)dw_slab_capacity(hidden) * kin * hidden);
    const size_t nslab = (size_t)csr_bwd_num_slabs(batch, n_vert) > (size_t)batch ? csr_bwd_num_slabs(batch, n_vert) : batch;
    // one block of partial rows PER LAYER ([batch][cpad] on the channel-sliced path, [row-walk workgroups][cpad] otherwise),
    // reduced by one launch at the end of the backward
    L.db_slab = take(align_up(nslab * cpad, 64) * (num_layers > 1 ? num_layers - 1 : 1));
    L.gq = take(m * (cpad > 4 ? cpad : 4));
    L.thin_dw_slab = take((size_t)thin_num_slabs() * kin * 3);
    L.thin_db_slab = take((size_t)thin_num_slabs() * 3);
  }
  L.total = off;
  return L;
}

static int check_stack_dims(int ld_feats, int in_features, int num_layers, int hidden, int cut_len) {
  if (num_layers < 1) { set_error("gcn_stack: num_layers=%d", num_layers); return -1; }
  // the scratch layout (weight-image stride, dW slabs) is sized from in_features alone: the row stride must be the
  // 4-float granule right above it, nothing wider
  if (ld_feats != pad4(in_features)) {
    set_error("gcn_stack: ld_feats=%d must be in_features=%d rounded up to a multiple of 4", ld_feats, in_features);
    return -1;
  }
  if (num_layers > 1 && (hidden % 4 != 0 || hidden > 304 || cut_len < 0 || cut_len > hidden)) {
    set_error("gcn_stack: hidden=%d (must be a multiple of 4, <= 304) cut_len=%d unsupported", hidden, cut_len);
    return -1;
  }
  if (ld_feats > 600) { set_error("gcn_stack: in_features=%d > 600 unsupported", in_features); return -1; }
  if (num_layers == 1 && ld_feats > 320) { set_error("gcn_stack: single-layer stack with %d inputs unsupported", in_features); return -1; }
  return 0;
}


// ---- bf16 STORAGE mode (gemm_bf16 == 2): scratch layout and the stack loops on the bf16 kernels (gcn_bf16s.hip) ----------
static inline int pad8(int n) { return (n + 7) & ~7; }
// ReLU-sign bytes per row in this mode: the aggregated-channel bytes cover pad8(cut_len) columns, and the row length is
// even so that the 2-byte groups of an 8-column epilogue store never straddle rows
static inline int mask_ld16(int hidden, int cut_len) { return (pad8(cut_len) / 4 + (hidden + 3) / 4 + 1) & ~1; }

constexpr int kStashInputLd = 608;   // pad8 of the widest stack input (check_stack16_dims: in_features <= 600)
struct Stack16Layout {
  size_t wt, wt_stride;   // bf16 weight images (offsets / stride in floats)
  size_t feats16;         // [M][ld0] bf16: the stack's fp32 input features, converted
  size_t za;              // [M][cpad] bf16
  size_t z3;              // [2][M][4] fp32
  size_t ping[2];         // [M][ldh] bf16 each
  size_t dw_slab, db_slab, thin_dw_slab, thin_db_slab, heavy, total;
  size_t tplan;           // the tiled aggregation's plan (csr16t_plan_ints)
  int ld0, ldh, cpad;
};

static Stack16Layout stack16_layout(int batch, int n_vert, int in_features, int hidden, int num_layers, int cut_len,
                                    int need_backward) {
  Stack16Layout L{};
  const size_t m = (size_t)batch * n_vert;
  L.ld0 = pad8(in_features);
  L.ldh = pad8(hidden);
  L.cpad = pad8(cut_len);
  size_t off = 0;
  auto take = [&](size_t nfloats) {
    const size_t o = off;
    off = align_up(off + nfloats, 64);
    return o;
  };
  auto take16 = [&](size_t nelems) { return take((nelems + 1) / 2 + 64); };  // bf16 region (+ slack for 16-byte tails)
  // images: forward W_i^T [bt_rows(hidden)][2 pad16(k_i / 2)], backward W_i [bt_rows(n_store_i)][2 pad16(ldh / 2)]
  const int kmax = L.ld0 > L.ldh ? L.ld0 : L.ldh;
  const int rows_f = rowgemm_bt_rows(hidden), rows_b = rowgemm_bt_rows(kmax);
  const size_t img_f = (size_t)rows_f * pad16(kmax / 2), img_b = (size_t)rows_b * pad16(L.ldh / 2);
  L.wt_stride = align_up(img_f > img_b ? img_f : img_b, 64);
  L.wt = take(L.wt_stride * (num_layers > 1 ? num_layers - 1 : 1));
  L.feats16 = take16(m * L.ld0);
  L.za = take16(m * (L.cpad > 8 ? L.cpad : 8));
  L.z3 = take(2 * m * 4);
  L.ping[0] = take16(m * L.ldh);
  L.ping[1] = take16(m * L.ldh);
  L.heavy = take(csr_heavy_scratch_ints(n_vert));
  L.tplan = take(csr16t_plan_ints(n_vert));
  if (need_backward) {
    const size_t kin = in_features > hidden ? in_features : hidden;
    L.dw_slab = take((size_t)dw16_num_slabs(hidden) * (kin > 304 ? 304 : kin) * hidden);
    L.db_slab = take(align_up((size_t)csr_bwd_num_slabs(batch, n_vert) * (L.cpad > 8 ? L.cpad : 8), 64) * (num_layers > 1 ? num_layers - 1 : 1));
    L.thin_dw_slab = take((size_t)thin_num_slabs() * hidden * 3);
    L.thin_db_slab = take((size_t)thin_num_slabs() * 3);
  }
  L.total = off;
  return L;
}

static int check_stack16_dims(int num_layers, int hidden, int in_features) {
  if (num_layers < 2) { set_error("gcn_stack: the bf16 storage mode needs at least one hidden layer"); return -1; }
  if (hidden > 304 || hidden % 4 != 0 || hidden < 16) { set_error("gcn_stack: bf16 storage mode: hidden=%d unsupported", hidden); return -1; }
  if (in_features > 600) { set_error("gcn_stack: in_features=%d > 600 unsupported", in_features); return -1; }
  if (in_features > 304 && in_features % 8 != 0) {
    set_error("gcn_stack: bf16 storage mode: in_features=%d > 304 must be a multiple of 8", in_features);
    return -1;
  }
  return 0;
}

static int stack_fwd16(const float *feats, int ld_feats, int in_features, const float *const *weights,
                       const float *const *biases, int num_layers, int hidden, int cut_len, const int32_t *rowptr,
                       const int32_t *col, const float *val, int max_degree, const a3vt_adj_split *split, int n_vert,
                       int batch, void *acts, uint8_t *masks, float *scratch, float *update, hipStream_t s) {
  if (int rc = check_stack16_dims(num_layers, hidden, in_features)) return rc;
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  const Stack16Layout L = stack16_layout(batch, n_vert, in_features, hidden, num_layers, cut_len, 0);
  const size_t m = (size_t)batch * n_vert;
  const int mld = mask_ld16(hidden, cut_len);
  const size_t mpad = (m + 31) / 32 * 32;
  using u16 = unsigned short;
  int32_t *heavy = nullptr;
  if (max_degree <= 0 || max_degree > csr_heavy_degree()) {
    heavy = reinterpret_cast<int32_t *>(scratch + L.heavy);
    if (int rc = launch_csr_heavy_list(rowptr, n_vert, heavy, s)) return rc;
  }
  if (num_layers - 1 > kMaxImages) { set_error("gcn_stack: bf16 storage mode supports up to %d hidden layers", kMaxImages); return -1; }
  {
    WeightImages wi{};
    int max_ld = 0;
    for (int i = 0; i + 1 < num_layers; ++i) {
      wi.w[i] = weights[i];
      wi.k[i] = i == 0 ? in_features : hidden;
      wi.rows[i] = rowgemm_bt_rows(hidden);
      wi.ld[i] = 2 * pad16((i == 0 ? L.ld0 : L.ldh) / 2);
      max_ld = wi.ld[i] > max_ld ? wi.ld[i] : max_ld;
    }
    wi.dst = scratch + L.wt;
    wi.dst_stride = L.wt_stride;
    wi.n = hidden;
    wi.count = num_layers - 1;
    wi.transpose = 1;
    if (int rc = launch_weight_images16(wi, rowgemm_bt_rows(hidden), max_ld, s)) return rc;
  }
  // the input rows in bf16: behind the hidden layers' rows in the stash when there is one (the backward reads them there)
  u16 *f16 = acts ? static_cast<u16 *>(acts) + (size_t)(num_layers - 1) * m * L.ldh : reinterpret_cast<u16 *>(scratch + L.feats16);
  if (int rc = launch_cvt_rows(feats, ld_feats, in_features, f16, L.ld0, (long long)m, s)) return rc;
  // bounded-degree graphs: the aggregation reads its neighbour rows from LDS tiles (gcn_bf16s.hip, csr16t); plan per call
  const bool tiled = heavy == nullptr && cut_len > 0 && g_csr_algo != 1 && csr16t_ok(n_vert, cut_len, max_degree, (long long)m);
  int32_t *tplan = reinterpret_cast<int32_t *>(scratch + L.tplan);
  if (tiled)
    if (int rc = launch_csr16t_build(rowptr, col, val, n_vert, tplan, s)) return rc;

  const u16 *x = f16;
  int ldx = L.ld0;
  u16 *za = reinterpret_cast<u16 *>(scratch + L.za);
  for (int i = 0; i + 1 < num_layers; ++i) {
    u16 *y = acts ? static_cast<u16 *>(acts) + (size_t)i * m * L.ldh : reinterpret_cast<u16 *>(scratch + L.ping[i & 1]);
    RowGemmArgs g{};
    g.a0 = g.a1 = reinterpret_cast<const float *>(x);
    g.lda0 = g.lda1 = ldx / 2;
    g.ksplit = g.k = ldx / 2;
    g.bt = scratch + L.wt + L.wt_stride * i;
    g.ldb = pad16(ldx / 2);
    g.zeros = zeros;
    g.m = (int)m;
    g.n_store = hidden;
    g.c = reinterpret_cast<float *>(y);
    g.ldc = L.ldh;
    g.c2 = reinterpret_cast<float *>(za);
    g.ldc2 = L.cpad > 8 ? L.cpad : 8;
    g.csplit = cut_len;
    uint8_t *mk = masks ? masks + (size_t)i * mpad * mld : nullptr;
    g.maskb = mk;
    g.mld = mld;
    g.moff = L.cpad / 4;
    g.bf16 = 2;
    {
      ProfScope ps(PROF_GEMM_FWD, s);
      if (int rc = launch_rowgemm(g, EPI_FWD_HIDDEN, s)) return rc;
    }
    if (cut_len > 0) {
      ProfScope psa(PROF_AGG, s);
      if (tiled) {
        if (int rc = launch_csr16t_fwd(za, g.ldc2, biases[i], cut_len, tplan, rowptr, col, val, n_vert, batch, y, L.ldh, mk, mld, 1, s))
          return rc;
      } else if (int rc = launch_csr16_fwd(za, g.ldc2, biases[i], cut_len, rowptr, col, val, heavy, n_vert, batch, y, L.ldh, mk, mld, 1, s)) {
        return rc;
      }
    }
    x = y;
    ldx = L.ldh;
  }
  const int last = num_layers - 1;
  ProfScope pso(PROF_OUT, s);
  if (int rc = launch_thin16_fwd_product(x, ldx, hidden, weights[last], (long long)m, scratch + L.z3, s)) return rc;
  const SplitRef sref = split_ref(split);
  return launch_csr3(scratch + L.z3, biases[last], rowptr, col, val, heavy, n_vert, batch, update, 3, s,
                     split && split->rowptr ? &sref : nullptr, false);
}

static int stack_bwd16(const float *feats, int ld_feats, int in_features, const float *const *weights, int num_layers,
                       int hidden, int cut_len, const int32_t *rowptrT, const int32_t *colT, const float *valT,
                       int max_degreeT, const a3vt_adj_split *split, int n_vert, int batch, const void *acts,
                       const uint8_t *masks, const float *grad_update, float *const *grad_weights,
                       float *const *grad_biases, float *grad_feats, float *scratch, int acc, hipStream_t s) {
  if (int rc = check_stack16_dims(num_layers, hidden, in_features)) return rc;
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  const Stack16Layout L = stack16_layout(batch, n_vert, in_features, hidden, num_layers, cut_len, 1);
  const size_t m = (size_t)batch * n_vert;
  const int mld = mask_ld16(hidden, cut_len);
  const size_t mpad = (m + 31) / 32 * 32;
  const int last = num_layers - 1;
  using u16 = unsigned short;
  const u16 *acts16 = static_cast<const u16 *>(acts);
  int32_t *heavyT = nullptr;
  if (max_degreeT <= 0 || max_degreeT > csr_heavy_degree()) {
    heavyT = reinterpret_cast<int32_t *>(scratch + L.heavy);
    if (int rc = launch_csr_heavy_list(rowptrT, n_vert, heavyT, s)) return rc;
  }
  // the stack's input in bf16: the forward left it behind the hidden layers' rows in the stash (a3vt_gcn_stack_stash_bytes)
  const u16 *f16 = acts16 + (size_t)last * m * L.ldh;
  (void)feats;
  (void)ld_feats;
  // the tiled aggregation's plan for A^T (see stack_fwd16)
  const bool tiled = heavyT == nullptr && cut_len > 0 && g_csr_algo != 1 && csr16t_ok(n_vert, cut_len, max_degreeT, (long long)m);
  int32_t *tplan = reinterpret_cast<int32_t *>(scratch + L.tplan);
  if (tiled)
    if (int rc = launch_csr16t_build(rowptrT, colT, valT, n_vert, tplan, s)) return rc;

  // ---- output layer: dz3 = A^T dU, then one pass over X_{L-1}: G (bf16, ReLU-masked), dW, db partials
  u16 *ping[2] = {reinterpret_cast<u16 *>(scratch + L.ping[0]), reinterpret_cast<u16 *>(scratch + L.ping[1])};
  {
    float *du4 = scratch + L.z3, *res = scratch + L.z3 + m * 4;
    ProfScope pso(PROF_OUT, s);
    if (int rc = launch_pad3to4(grad_update, (long long)m, du4, s)) return rc;
    const SplitRef sref = split_ref(split);
    if (int rc = launch_csr3(du4, nullptr, rowptrT, colT, valT, heavyT, n_vert, batch, res, 4, s,
                             split && split->rowptr ? &sref : nullptr, true))
      return rc;
    const u16 *x = acts16 + (size_t)(last - 1) * m * L.ldh;
    if (int rc = launch_thin16_bwd_main(x, L.ldh, hidden, weights[last], res, grad_update, (long long)m, 1, ping[0], L.ldh,
                                        scratch + L.thin_dw_slab, scratch + L.thin_db_slab, s))
      return rc;
    if (int rc = launch_slab_reduce_za(scratch + L.thin_dw_slab, thin_num_slabs(), (size_t)hidden * 3, (size_t)hidden * 3,
                                       (size_t)hidden * 3, grad_weights[last], acc, s))
      return rc;
    if (int rc = launch_slab_reduce_za(scratch + L.thin_db_slab, thin_num_slabs(), 3, 3, 3, grad_biases[last], acc, s)) return rc;
  }
  // bf16 images Bt = W_i (zero padded) for dX_i = dZ W_i^T
  {
    WeightImages wi{};
    int max_rows = 0;
    for (int i = 0; i < last; ++i) {
      wi.w[i] = weights[i];
      wi.k[i] = i == 0 ? in_features : hidden;
      wi.rows[i] = rowgemm_bt_rows(i == 0 ? ld_feats : L.ldh);
      wi.ld[i] = 2 * pad16(L.ldh / 2);
      max_rows = wi.rows[i] > max_rows ? wi.rows[i] : max_rows;
    }
    wi.dst = scratch + L.wt;
    wi.dst_stride = L.wt_stride;
    wi.n = hidden;
    wi.count = last;
    wi.transpose = 0;
    if (int rc = launch_weight_images16(wi, max_rows, 2 * pad16(L.ldh / 2), s)) return rc;
  }
  u16 *dza = reinterpret_cast<u16 *>(scratch + L.za);
  const int cpad = L.cpad, ldza = cpad > 8 ? cpad : 8;
  const size_t db_layer_stride = align_up((size_t)csr_bwd_num_slabs(batch, n_vert) * (cpad > 8 ? cpad : 8), 64);
  int cur = 0;
  for (int i = last - 1; i >= 0; --i) {
    u16 *g = ping[cur];
    const u16 *x = i == 0 ? f16 : acts16 + (size_t)(i - 1) * m * L.ldh;
    const int ldx = i == 0 ? L.ld0 : L.ldh;
    const int kin = i == 0 ? in_features : hidden;
    if (cut_len > 0) {
      ProfScope psa(PROF_AGG, s);
      if (tiled) {
        if (int rc = launch_csr16t_bwd(g, L.ldh, cut_len, cpad, tplan, rowptrT, colT, valT, n_vert, batch, dza, ldza,
                                       scratch + L.db_slab + (size_t)i * db_layer_stride, s))
          return rc;
      } else if (int rc = launch_csr16_bwd(g, L.ldh, cut_len, cpad, rowptrT, colT, valT, heavyT, n_vert, batch, dza, ldza,
                                           scratch + L.db_slab + (size_t)i * db_layer_stride, s)) {
        return rc;
      }
    } else if (!acc) {
      if (int rc = launch_fill_zero(grad_biases[i], hidden, s)) return rc;
    }
    // dW_i = X_i^T dZ in panels of <= 304 input channels (windows of the rows: no copies)
    for (int c0 = 0; c0 < kin; c0 += 304) {
      const int w = kin - c0 < 304 ? kin - c0 : 304;
      Dw16Args d{};
      d.x = x;
      d.ldx = ldx;
      d.xc0 = c0;
      d.xw = pad8(w);
      d.z0 = dza;
      d.ldz0 = ldza;
      d.z1 = g;
      d.ldz1 = L.ldh;
      d.zsplit = cut_len > 0 ? cpad : 0;
      d.zeros = zeros;
      d.slab = scratch + L.dw_slab;
      d.m = (int)m;
      d.k_in = w;
      d.n_out = hidden;
      {
        ProfScope ps(PROF_DW, s);
        if (int rc = launch_dw16(d, s)) return rc;
      }
      if (int rc = launch_slab_reduce_za(scratch + L.dw_slab, dw16_num_slabs(hidden), (size_t)w * hidden, (size_t)w * hidden,
                                         (size_t)w * hidden, grad_weights[i] + (size_t)c0 * hidden, acc, s))
        return rc;
    }
    // dX_i = dZ W_i^T
    RowGemmArgs r{};
    r.a0 = reinterpret_cast<const float *>(dza);
    r.lda0 = ldza / 2;
    r.a1 = reinterpret_cast<const float *>(g);
    r.lda1 = L.ldh / 2;
    r.ksplit = cut_len > 0 ? cpad / 2 : 0;
    r.k = L.ldh / 2;
    r.bt = scratch + L.wt + L.wt_stride * i;
    r.ldb = pad16(L.ldh / 2);
    r.zeros = zeros;
    r.m = (int)m;
    r.bf16 = 2;
    if (i == 0) {
      r.n_store = ld_feats;
      r.c = grad_feats;
      r.ldc = ld_feats;
      if (int rc = launch_rowgemm(r, EPI_PLAIN, s)) return rc;
    } else {
      r.n_store = hidden;
      r.c = reinterpret_cast<float *>(ping[cur ^ 1]);
      r.ldc = L.ldh;
      r.maskb = const_cast<uint8_t *>(masks) + (size_t)(i - 1) * mpad * mld;
      r.mld = mld;
      r.moff = cpad / 4;
      r.csplit = cut_len;
      {
        ProfScope ps(PROF_GEMM_DX, s);
        if (int rc = launch_rowgemm(r, EPI_DX_MASK, s)) return rc;
      }
      cur ^= 1;
    }
  }
  if (cut_len > 0)   // bias gradients of all hidden layers: one launch (as the fp32 stack)
    if (int rc = reduce_bias_partials(scratch + L.db_slab, db_layer_stride, csr_bwd_num_slabs(batch, n_vert), cpad, cut_len, hidden,
                                      last, grad_biases, acc, s))
      return rc;
  return 0;
}

}  // namespace a3vt

using namespace a3vt;

extern "C" {

int a3vt_version(void) { return A3VT_VERSION; }
const char *a3vt_last_error(void) { return g_err; }

int a3vt_csr_validate(const int32_t *rowptr, const int32_t *col, int n_vert, int nnz) {
  A3VT_CHECK_ARG(rowptr && col && n_vert > 0 && nnz >= 0);
  if (rowptr[0] != 0 || rowptr[n_vert] != nnz) {
    set_error("csr: rowptr[0]=%d rowptr[n]=%d nnz=%d", rowptr[0], rowptr[n_vert], nnz);
    return -1;
  }
  for (int i = 0; i < n_vert; ++i)
    if (rowptr[i + 1] < rowptr[i]) { set_error("csr: rowptr not monotone at %d", i); return -1; }
  for (int e = 0; e < nnz; ++e)
    if (col[e] < 0 || col[e] >= n_vert) { set_error("csr: col[%d]=%d out of range", e, col[e]); return -1; }
  return 0;
}

int a3vt_adj_split_validate(const int32_t *rowptr, const int32_t *col, const float *val, int n_vert,
                            const int32_t *p_rowptr, const int32_t *p_col, const float *scale, const uint8_t *cls) {
  A3VT_CHECK_ARG(rowptr && col && val && p_rowptr && p_col && scale && cls && n_vert > 0);
  if (p_rowptr[0] != 0) { set_error("adj_split: p_rowptr[0]=%d", p_rowptr[0]); return -1; }
  // the two classes, ascending
  int ns = 0, nc = 0;
  for (int i = 0; i < n_vert; ++i) {
    if (cls[i] > 2) { set_error("adj_split: cls[%d]=%d", i, (int)cls[i]); return -1; }
    ns += cls[i] == 1;
    nc += cls[i] == 2;
  }
  if ((ns == 0) != (nc == 0)) { set_error("adj_split: %d seam vertices but %d centres", ns, nc); return -1; }
  int32_t *members = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)(ns + nc + 1)));
  A3VT_CHECK_ARG(members != nullptr);
  int32_t *seam = members, *centre = members + ns;
  for (int i = 0, a = 0, b = 0; i < n_vert; ++i) {
    if (cls[i] == 1) seam[a++] = i;
    if (cls[i] == 2) centre[b++] = i;
  }
  int rc = 0;
  for (int i = 0; i < n_vert && rc == 0; ++i) {
    const int e0 = rowptr[i], e1 = rowptr[i + 1], p0 = p_rowptr[i], p1 = p_rowptr[i + 1];
    const int32_t *other = cls[i] == 1 ? centre : cls[i] == 2 ? seam : nullptr;
    const int no = cls[i] == 1 ? nc : cls[i] == 2 ? ns : 0;
    if (p1 < p0 || e1 - e0 != (p1 - p0) + no) {
      set_error("adj_split: row %d has %d entries, the split gives %d + %d", i, e1 - e0, p1 - p0, no);
      rc = -1;
      break;
    }
    // merge of the row of P (ascending, in range, no duplicates) with the other class (ascending): must reproduce the row
    int a = p0, b = 0;
    for (int e = e0; e < e1; ++e) {
      const bool has_a = a < p1, has_b = b < no;
      int next;
      if (has_a && has_b && p_col[a] == other[b]) {
        set_error("adj_split: row %d: column %d is in P and in the bipartite block", i, p_col[a]);
        rc = -1;
        break;
      }
      if (has_a && (!has_b || p_col[a] < other[b])) {
        next = p_col[a++];
        if (next < 0 || next >= n_vert || (a - 1 > p0 && p_col[a - 2] >= next)) {
          set_error("adj_split: row %d of P is not ascending / in range at column %d", i, next);
          rc = -1;
          break;
        }
        // symmetry of P: (next, i) must be an entry of row `next` (binary search)
        int lo = p_rowptr[next], hi = p_rowptr[next + 1] - 1;
        bool found = false;
        while (lo <= hi) {
          const int mid = (lo + hi) / 2;
          if (p_col[mid] == i) { found = true; break; }
          if (p_col[mid] < i) lo = mid + 1;
          else hi = mid - 1;
        }
        if (!found) { set_error("adj_split: P has (%d, %d) but not (%d, %d)", i, next, next, i); rc = -1; break; }
      } else {
        next = other[b++];
      }
      if (col[e] != next) {
        set_error("adj_split: row %d entry %d: column %d, the split gives %d", i, e - e0, col[e], next);
        rc = -1;
        break;
      }
      if (memcmp(&val[e], &scale[i], sizeof(float)) != 0) {
        set_error("adj_split: row %d entry %d: value %.9g != scale %.9g", i, e - e0, (double)val[e], (double)scale[i]);
        rc = -1;
        break;
      }
    }
  }
  free(members);
  return rc;
}

// ReLU-sign bytes saved by the forward pass for the backward pass: [num_layers-1][pad32(M)][mld],
// mld = pad4(cut_len)/4 + ceil(hidden/4).  (mask_ld: defined with stack_x3 above)
size_t a3vt_gcn_stack_mask_bytes(int batch, int n_vert, int hidden, int num_layers, int cut_len) {
  if (num_layers < 2 || batch <= 0 || n_vert <= 0 || hidden <= 0 || cut_len < 0) return 0;
  const size_t mpad = ((size_t)batch * n_vert + 31) / 32 * 32;
  // row-major sign bytes, then the quad-major signs of the aggregated channels (channel-sliced aggregation)
  return (size_t)(num_layers - 1) * (mpad * mask_ld(hidden, cut_len) + signq_stride((size_t)batch * n_vert, cut_len));
}

size_t a3vt_gcn_stack_scratch_bytes(int batch, int n_vert, int in_features, int hidden, int num_layers, int cut_len,
                                    int need_backward) {
  // enough for EVERY gemm mode (mode 3 keeps three bf16 images per hidden layer: a larger weight slot than mode 0's;
  // mode 2 has a layout of its own): a caller that sizes its scratch here may pass any gemm_bf16
  size_t most = 0;
  for (int mode = 0; mode <= 3; ++mode) {
    if (mode == 2 && num_layers < 2) continue;
    const size_t b = a3vt_gcn_stack_scratch_bytes_mode(batch, n_vert, in_features, hidden, num_layers, cut_len, need_backward, mode);
    most = b > most ? b : most;
  }
  return most;
}

size_t a3vt_gcn_stack_scratch_bytes_mode(int batch, int n_vert, int in_features, int hidden, int num_layers,
                                         int cut_len, int need_backward, int gemm_bf16) {
  // (a size query never fails: sizes no stack call accepts give 0 — the layouts below divide by some of them)
  if (batch <= 0 || n_vert <= 0 || in_features <= 0 || hidden <= 0 || num_layers <= 0 || cut_len < 0 || gemm_bf16 < 0 || gemm_bf16 > 3)
    return 0;
  if (gemm_bf16 == 2)
    return stack16_layout(batch, n_vert, in_features, hidden, num_layers, cut_len, need_backward).total * sizeof(float);
  return stack_layout(batch, n_vert, in_features, hidden, num_layers, cut_len, need_backward, gemm_bf16).total * sizeof(float);
}

int a3vt_gcn_stack_stash_bytes(int batch, int n_vert, int hidden, int num_layers, int cut_len, int gemm_bf16,
                               size_t *acts_bytes, size_t *mask_bytes) {
  A3VT_CHECK_ARG(acts_bytes && mask_bytes && batch > 0 && n_vert > 0 && hidden > 0 && cut_len >= 0);
  *acts_bytes = *mask_bytes = 0;
  if (num_layers < 2) return 0;
  const size_t m = (size_t)batch * n_vert, mpad = (m + 31) / 32 * 32;
  if (gemm_bf16 == 2) {
    // the hidden layers' bf16 rows, then the stack's INPUT rows in bf16 (the backward needs them for dW_0 and would otherwise
    // convert the fp32 features a second time; sized for the widest input a stack accepts: in_features <= 600), + slack for
    // 16-byte reads of the last row's tail
    *acts_bytes = (size_t)(num_layers - 1) * m * pad8(hidden) * 2 + m * (size_t)kStashInputLd * 2 + 256;
    *mask_bytes = (size_t)(num_layers - 1) * mpad * mask_ld16(hidden, cut_len);
  } else {
    *acts_bytes = (size_t)(num_layers - 1) * m * hidden * sizeof(float);
    *mask_bytes = a3vt_gcn_stack_mask_bytes(batch, n_vert, hidden, num_layers, cut_len);
  }
  return 0;
}

int a3vt_gcn_stack_fwd(const float *feats, int ld_feats, int in_features, const float *const *weights,
                       const float *const *biases, int num_layers, int hidden, int cut_len,
                       const int32_t *rowptr, const int32_t *col, const float *val, int max_degree, int n_vert,
                       int batch, int gemm_bf16, void *acts_v, uint8_t *masks, float *scratch, float *update,
                       void *stream) {
  return a3vt_gcn_stack_fwd_adj(feats, ld_feats, in_features, weights, biases, num_layers, hidden, cut_len, rowptr, col, val,
                                max_degree, nullptr, n_vert, batch, gemm_bf16, acts_v, masks, scratch, update, stream);
}

int a3vt_gcn_stack_fwd_adj(const float *feats, int ld_feats, int in_features, const float *const *weights,
                           const float *const *biases, int num_layers, int hidden, int cut_len,
                           const int32_t *rowptr, const int32_t *col, const float *val, int max_degree,
                           const a3vt_adj_split *split, int n_vert,
                           int batch, int gemm_bf16, void *acts_v, uint8_t *masks, float *scratch, float *update,
                           void *stream) {
  float *acts = static_cast<float *>(acts_v);
  A3VT_CHECK_ARG(feats && weights && biases && rowptr && col && val && scratch && update);
  A3VT_CHECK_ARG((acts == nullptr) == (masks == nullptr) || num_layers < 2);
  A3VT_CHECK_ARG(n_vert > 0 && batch > 0);
  A3VT_CHECK_ARG(gemm_bf16 >= 0 && gemm_bf16 <= 3);
  if (int rc = check_stack_dims(ld_feats, in_features, num_layers, hidden, cut_len)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (gemm_bf16 == 2)  // bf16 storage: `acts` holds bf16 rows (a3vt_gcn_stack_stash_bytes)
    return stack_fwd16(feats, ld_feats, in_features, weights, biases, num_layers, hidden, cut_len, rowptr, col, val,
                       max_degree, split, n_vert, batch, acts, masks, scratch, update, s);
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  const StackLayout L = stack_layout(batch, n_vert, in_features, hidden, num_layers, cut_len, 0, gemm_bf16);
  const size_t m = (size_t)batch * n_vert;
  const int cpad = pad4(cut_len);
  const int mld = mask_ld(hidden, cut_len);
  const size_t mpad = (m + 31) / 32 * 32;

  // the 3-channel aggregation of the output layer through the split (any P: its rows are walked from global memory)
  const SplitRef sref = split_ref(split);
  const SplitRef *sp3 = split && split->rowptr && split->col && split->scale && split->cls ? &sref : nullptr;
  // hub rows (if any): listed once, then every layer's aggregation hands them to whole workgroups
  int32_t *heavy = nullptr;
  if (max_degree <= 0 || max_degree > csr_heavy_degree()) {
    heavy = reinterpret_cast<int32_t *>(scratch + L.heavy);
    if (int rc = launch_csr_heavy_list(rowptr, n_vert, heavy, s)) return rc;
  }
  // Gemm mode 3 ("fp32x3", gcn_gemm3.hip): the hidden-layer products whose shape the split-operand kernels take run there;
  // everything else of the call (the first layer, narrow or short stacks) on the exact fp32 kernels.
  const bool x3 = gemm_bf16 == 3 && rowgemm3_stack_ok((long long)m, hidden, mask_ld(hidden, cut_len));
  const int omode = gemm_bf16 == 1 ? 1 : 0;   // operand mode of the kernels shared with modes 0 / 1
  // transposed, zero-padded weight images of all hidden layers (one launch)
  const bool batched_images = num_layers - 1 <= kMaxImages;
  if (x3) {   // layers 1 .. L-2 as three bf16 images each; layer 0 (K = in_features) keeps its fp32 image
    for (int i0 = 1; i0 + 1 < num_layers; i0 += kMaxImages) {
      WeightImages w3{};
      const int cnt = num_layers - 1 - i0 < kMaxImages ? num_layers - 1 - i0 : kMaxImages;
      for (int j = 0; j < cnt; ++j) {
        w3.w[j] = weights[i0 + j];
        w3.k[j] = hidden;
      }
      w3.dst = scratch + L.wt + L.wt_stride * i0;
      w3.dst_stride = L.wt_stride;
      w3.n = hidden;
      w3.count = cnt;
      w3.transpose = 1;
      if (int rc = launch_weight_images3(w3, s)) return rc;
    }
  }
  if (batched_images && num_layers > 1) {
    WeightImages wi{};
    for (int i = 0; i + 1 < (x3 ? 2 : num_layers); ++i) {
      wi.w[i] = weights[i];
      wi.k[i] = i == 0 ? in_features : hidden;
      wi.rows[i] = rowgemm_bt_rows(hidden);
      wi.ld[i] = pad16(i == 0 ? ld_feats : hidden);
    }
    wi.dst = scratch + L.wt;
    wi.dst_stride = L.wt_stride;
    wi.n = hidden;
    wi.count = x3 ? 1 : num_layers - 1;
    wi.transpose = 1;
    if (int rc = launch_weight_images(wi, rowgemm_bt_rows(hidden), pad16(ld_feats > hidden ? ld_feats : hidden), s)) return rc;
  }

  // a usable split stands for "every row short": the hub and seam rows are gone from P
  const bool split_ok = split_usable(split, n_vert, cut_len);
  const bool quad = use_csrq(batch, n_vert, hidden, cut_len, gemm_bf16, split_ok ? csrq_max_degree() : max_degree) && num_layers > 1;
  const bool qsplit = quad && split_ok;
  if (num_layers > 1) path_count(quad ? PATH_STACK_QUAD : PATH_STACK_ROWS);
  if (qsplit) path_count(PATH_STACK_SPLIT);
  if (num_layers > 1) stash_layout_record(masks, quad);
  int32_t *ell = reinterpret_cast<int32_t *>(scratch + L.ell);
  if (qsplit) {
    if (int rc = launch_csrqs_image(split->rowptr, split->col, split->scale, split->cls, n_vert, ell, s)) return rc;
  } else if (quad) {
    if (int rc = launch_csrq_ell(rowptr, col, val, n_vert, ell, s)) return rc;
  }
  // Layer outputs.  Row-major [M][hidden] on the half-wave path; on the channel-sliced path ("hybrid" rows) the block of
  // M * hidden floats holds columns [0, 160) quad-major [batch][40][n_vert] float4 — the aggregated channels written by
  // the aggregation kernel, the rest of the product kernel's first column group by its epilogue — followed by columns
  // [160, hidden) row-major [M][hidden - 160].  160 = ten 16-column tiles = the input tiles of two waves of dw_kernel
  // and ten whole K chunks of rowgemm_kernel: no tile or chunk straddles the two parts.  Every consumer reads either.
  const int qcols = quad ? kQuadCols : 0;                    // columns of a layer's output kept quad-major
  const int rm_ld = hidden - qcols;                          // row stride of the row-major part
  const size_t rm_off = m * qcols;                           // its offset inside a layer's block
  const float *x = feats, *xq = nullptr;                     // x: pre-offset so that x + row * ldx + col is column col
  int ldx = ld_feats;
  for (int i = 0; i + 1 < num_layers; ++i) {
    const int k = i == 0 ? ld_feats : hidden;       // K walked by the kernel (pad columns of feats are zero)
    const int kin = i == 0 ? in_features : hidden;  // rows of W_i
    float *wt = scratch + L.wt + L.wt_stride * i;
    const bool l3 = x3 && i > 0;   // this layer's product on the split-operand kernel
    if (!batched_images && !l3)
      if (int rc = launch_transpose_pad(weights[i], kin, hidden, wt, rowgemm_bt_rows(hidden), pad16(k), s)) return rc;
    float *y = acts ? acts + (size_t)i * m * hidden : scratch + L.ping[i & 1];
    RowGemmArgs g{};
    g.a0 = g.a1 = x;
    g.lda0 = g.lda1 = ldx;
    g.ksplit = k;
    if (xq) {   // hybrid input rows: columns [0, 160) from the quad-major part
      g.a0 = xq;
      g.ksplit = qcols;
      g.a0q_nvert = n_vert;
      g.a0q_quads = qcols / 4;
    }
    g.bt = wt;
    g.ldb = pad16(k);
    g.zeros = zeros;
    g.m = (int)m;
    g.k = k;
    g.n_store = hidden;
    g.c = y + rm_off - qcols;
    g.ldc = rm_ld;
    g.c2 = scratch + L.za;
    g.ldc2 = cpad;
    g.csplit = cut_len;
    uint8_t *mk = masks ? masks + (size_t)i * mpad * mld : nullptr;
    g.maskb = mk;
    g.mld = mld;
    g.moff = cpad / 4;
    g.bf16 = l3 ? 3 : omode;
    if (quad) {   // raw columns [0, cpad) leave the product quad-major, for the channel-sliced aggregation;
      g.zq_nvert = n_vert;   // activated columns [cpad, 160) quad-major into the layer's output
      g.zq_quads = cpad / 4;
      g.yq = y;
      g.yq_quads = qcols / 4;
      g.trash = scratch + L.z3;   // (the output layer's scratch: idle while the hidden layers run)
    }
    {
      ProfScope ps(PROF_GEMM_FWD, s);
      if (int rc = launch_rowgemm(g, EPI_FWD_HIDDEN, s)) return rc;
    }
    if (quad) {
      uint8_t *sq = masks ? masks + (size_t)(num_layers - 1) * mpad * mld + (size_t)i * signq_stride(m, cut_len) : nullptr;
      ProfScope psa(PROF_AGG, s);
      if (qsplit) {
        if (int rc = launch_csrqs_fwd(scratch + L.za, biases[i], cut_len, ell, n_vert, batch, y, qcols / 4, sq, 1, s)) return rc;
      } else if (int rc = launch_csrq_fwd(scratch + L.za, biases[i], cut_len, rowptr, col, val, heavy, ell, n_vert, batch, y, qcols / 4, sq, 1, s)) {
        return rc;
      }
      xq = y;
    } else if (cut_len > 0) {
      ProfScope psa(PROF_AGG, s);
      if (int rc = launch_csr_fwd(scratch + L.za, cpad, biases[i], cut_len, rowptr, col, val, heavy, n_vert, batch, y, hidden, mk, mld, 1, s))
        return rc;
    }
    x = y + rm_off - qcols;
    ldx = rm_ld;
  }
  const int klast = num_layers == 1 ? in_features : hidden;
  ProfScope pso(PROF_OUT, s);
  return launch_thin_fwd(x, ldx, klast, weights[num_layers - 1], biases[num_layers - 1], rowptr, col, val, heavy, n_vert,
                         batch, scratch + L.z3, update, xq, qcols / 4, s, sp3);
}

int a3vt_gcn_stack_bwd(const float *feats, int ld_feats, int in_features, const float *const *weights,
                       const float *const *biases, int num_layers, int hidden, int cut_len,
                       const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *rowptrT,
                       const int32_t *colT, const float *valT, int max_degreeT, int n_vert, int batch, int gemm_bf16,
                       const void *acts_v,
                       const uint8_t *masks, const float *grad_update, float *const *grad_weights, float *const *grad_biases,
                       float *grad_feats, float *scratch, void *stream) {
  return a3vt_gcn_stack_bwd_adj(feats, ld_feats, in_features, weights, biases, num_layers, hidden, cut_len, rowptr, col, val,
                                rowptrT, colT, valT, max_degreeT, nullptr, n_vert, batch, gemm_bf16, acts_v, masks, grad_update,
                                grad_weights, grad_biases, grad_feats, scratch, 0, stream);
}

int a3vt_gcn_stack_bwd_acc(const float *feats, int ld_feats, int in_features, const float *const *weights,
                           const float *const *biases, int num_layers, int hidden, int cut_len,
                           const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *rowptrT,
                           const int32_t *colT, const float *valT, int max_degreeT, int n_vert, int batch, int gemm_bf16,
                           const void *acts_v,
                           const uint8_t *masks, const float *grad_update, float *const *grad_weights,
                           float *const *grad_biases, float *grad_feats, float *scratch, int accumulate, void *stream) {
  return a3vt_gcn_stack_bwd_adj(feats, ld_feats, in_features, weights, biases, num_layers, hidden, cut_len, rowptr, col, val,
                                rowptrT, colT, valT, max_degreeT, nullptr, n_vert, batch, gemm_bf16, acts_v, masks, grad_update,
                                grad_weights, grad_biases, grad_feats, scratch, accumulate, stream);
}

int a3vt_gcn_stack_bwd_adj(const float *feats, int ld_feats, int in_features, const float *const *weights,
                           const float *const *biases, int num_layers, int hidden, int cut_len,
                           const int32_t *rowptr, const int32_t *col, const float *val, const int32_t *rowptrT,
                           const int32_t *colT, const float *valT, int max_degreeT, const a3vt_adj_split *split,
                           int n_vert, int batch, int gemm_bf16, const void *acts_v,
                           const uint8_t *masks, const float *grad_update, float *const *grad_weights,
                           float *const *grad_biases, float *grad_feats, float *scratch, int accumulate, void *stream) {
  const int acc = accumulate ? 1 : 0;
  const float *acts = static_cast<const float *>(acts_v);
  (void)biases; (void)rowptr; (void)col; (void)val;
  A3VT_CHECK_ARG(feats && weights && rowptrT && colT && valT && grad_update && grad_weights && grad_biases);
  A3VT_CHECK_ARG(grad_feats && scratch && n_vert > 0 && batch > 0);
  A3VT_CHECK_ARG(gemm_bf16 >= 0 && gemm_bf16 <= 3);
  A3VT_CHECK_ARG(num_layers == 1 || acts != nullptr);
  A3VT_CHECK_ARG(num_layers <= 2 || masks != nullptr);
  if (int rc = check_stack_dims(ld_feats, in_features, num_layers, hidden, cut_len)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (gemm_bf16 == 2)
    return stack_bwd16(feats, ld_feats, in_features, weights, num_layers, hidden, cut_len, rowptrT, colT, valT,
                       max_degreeT, split, n_vert, batch, acts, masks, grad_update, grad_weights, grad_biases, grad_feats,
                       scratch, acc, s);
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  const StackLayout L = stack_layout(batch, n_vert, in_features, hidden, num_layers, cut_len, 1, gemm_bf16);
  const size_t m = (size_t)batch * n_vert;
  const int cpad = pad4(cut_len);
  const int mld = mask_ld(hidden, cut_len);
  const size_t mpad = (m + 31) / 32 * 32;
  const int last = num_layers - 1;

  const SplitRef sref = split_ref(split);
  const SplitRef *sp3 = split && split->rowptr && split->col && split->scale && split->cls ? &sref : nullptr;
  // hub rows of A^T (if any): listed once for every aggregation of this call
  int32_t *heavyT = nullptr;
  if (max_degreeT <= 0 || max_degreeT > csr_heavy_degree()) {
    heavyT = reinterpret_cast<int32_t *>(scratch + L.heavy);
    if (int rc = launch_csr_heavy_list(rowptrT, n_vert, heavyT, s)) return rc;
  }

  // the layout the forward left in this stash (recorded by a3vt_gcn_stack_fwd); a stash this library did not write: by shape
  const int rec = stash_layout_lookup(masks);
  const bool quad = masks != nullptr && num_layers > 1 &&
                    (rec >= 0 ? rec == 1 : use_csrq(batch, n_vert, hidden, cut_len, gemm_bf16,
                                                    split_usable(split, n_vert, cut_len) ? csrq_max_degree() : max_degreeT));
  const bool qsplit = quad && split_usable(split, n_vert, cut_len);   // P and J are symmetric: the same image serves A^T
  int32_t *ellT = reinterpret_cast<int32_t *>(scratch + L.ell);
  if (qsplit) {
    if (int rc = launch_csrqs_image(split->rowptr, split->col, split->scale, split->cls, n_vert, ellT, s)) return rc;
  } else if (quad) {
    if (int rc = launch_csrq_ell(rowptrT, colT, valT, n_vert, ellT, s)) return rc;
  }
  const int qcols = quad ? kQuadCols : 0;            // hybrid activation rows (a3vt_gcn_stack_fwd)
  const int rm_ld = hidden - qcols;
  const size_t rm_off = m * qcols;
  // ---- output layer
  {
    const float *xl = num_layers == 1 ? feats : acts + (size_t)(last - 1) * m * hidden;
    const float *x = xl + rm_off - qcols;                 // hybrid rows: see a3vt_gcn_stack_fwd (qcols = 0: plain rows)
    const int ldx = num_layers == 1 ? ld_feats : rm_ld;
    const int k = num_layers == 1 ? in_features : hidden;
    float *gprev = num_layers == 1 ? grad_feats : scratch + L.ping[0];
    const int ldg = num_layers == 1 ? ld_feats : hidden;
    ProfScope pso(PROF_OUT, s);
    if (int rc = launch_thin_bwd(x, ldx, k, weights[last], rowptrT, colT, valT, heavyT, n_vert, batch, grad_update,
                                 scratch + L.z3, num_layers > 1, gprev, ldg, ldg, scratch + L.thin_dw_slab,
                                 scratch + L.thin_db_slab, quad ? scratch + L.gq : nullptr, cpad / 4, quad ? xl : nullptr,
                                 qcols / 4, s, sp3))
      return rc;
    if (int rc = launch_slab_reduce_za(scratch + L.thin_dw_slab, thin_num_slabs(), (size_t)k * 3, (size_t)k * 3, (size_t)k * 3,
                                       grad_weights[last], acc, s))
      return rc;
    if (int rc = launch_slab_reduce_za(scratch + L.thin_db_slab, thin_num_slabs(), 3, 3, 3, grad_biases[last], acc, s)) return rc;
  }

  const bool x3 = gemm_bf16 == 3 && rowgemm3_stack_ok((long long)m, hidden, mask_ld(hidden, cut_len));   // as a3vt_gcn_stack_fwd
  const int omode = gemm_bf16 == 1 ? 1 : 0;
  // zero-padded weight images (Bt = W_i for dX) of all hidden layers, one launch
  const bool batched_images = num_layers - 1 <= kMaxImages;
  if (x3) {   // layers 1 .. L-2: three bf16 images each (layer 0's dX has the plain epilogue: exact kernels, fp32 image)
    for (int i0 = 1; i0 < last; i0 += kMaxImages) {
      WeightImages w3{};
      const int cnt = last - i0 < kMaxImages ? last - i0 : kMaxImages;
      for (int j = 0; j < cnt; ++j) {
        w3.w[j] = weights[i0 + j];
        w3.k[j] = hidden;
      }
      w3.dst = scratch + L.wt + L.wt_stride * i0;
      w3.dst_stride = L.wt_stride;
      w3.n = hidden;
      w3.count = cnt;
      w3.transpose = 0;
      if (int rc = launch_weight_images3(w3, s)) return rc;
    }
  }
  if (batched_images && num_layers > 1) {
    WeightImages wi{};
    int max_rows = 0;
    for (int i = 0; i < (x3 ? 1 : last); ++i) {
      wi.w[i] = weights[i];
      wi.k[i] = i == 0 ? in_features : hidden;
      wi.rows[i] = rowgemm_bt_rows(i == 0 ? ld_feats : hidden);
      wi.ld[i] = pad16(hidden);
      max_rows = wi.rows[i] > max_rows ? wi.rows[i] : max_rows;
    }
    wi.dst = scratch + L.wt;
    wi.dst_stride = L.wt_stride;
    wi.n = hidden;
    wi.count = x3 ? 1 : last;
    wi.transpose = 0;
    if (int rc = launch_weight_images(wi, max_rows, pad16(hidden), s)) return rc;
  }

  // ---- hidden layers, last to first.  g = dL/dY_i already multiplied by the ReLU mask of layer i.
  const int db_rows = quad ? batch : csr_bwd_num_slabs(batch, n_vert);
  const size_t db_layer_stride = align_up((size_t)(csr_bwd_num_slabs(batch, n_vert) > batch ? csr_bwd_num_slabs(batch, n_vert) : batch) * cpad, 64);
  int cur = 0;
  for (int i = last - 1; i >= 0; --i) {
    float *g = scratch + L.ping[cur];
    float *dza = scratch + L.za;
    const float *xl = i == 0 ? feats : acts + (size_t)(i - 1) * m * hidden;
    const bool xhyb = quad && i > 0;                  // X_i = output of layer i-1: hybrid rows on the channel-sliced path
    const float *x = xhyb ? xl + rm_off - qcols : xl;
    const int ldx = i == 0 ? ld_feats : (xhyb ? rm_ld : hidden);
    const int kin = i == 0 ? in_features : hidden;

    // bias gradient + A^T gather on the aggregated channels
    if (cut_len > 0 && quad) {
      const uint8_t *sq = masks + (size_t)(num_layers - 1) * mpad * mld + (size_t)i * signq_stride(m, cut_len);
      ProfScope psa(PROF_AGG, s);
      // the (mesh, channel) partials of this layer: summed over the meshes by ONE launch for all layers behind the loop
      if (qsplit) {
        if (int rc = launch_csrqs_bwd(scratch + L.gq, cut_len, ellT, n_vert, batch, dza, sq,
                                      scratch + L.db_slab + (size_t)i * db_layer_stride, s))
          return rc;
      } else if (int rc = launch_csrq_bwd(scratch + L.gq, cut_len, rowptrT, colT, valT, heavyT, ellT, n_vert, batch, dza, sq,
                                          scratch + L.db_slab + (size_t)i * db_layer_stride, s)) {
        return rc;
      }
    } else if (cut_len > 0) {
      ProfScope psa(PROF_AGG, s);
      if (int rc = launch_csr_bwd(g, hidden, cut_len, rowptrT, colT, valT, heavyT, n_vert, batch, dza, cpad,
                                  scratch + L.db_slab + (size_t)i * db_layer_stride, s))
        return rc;
      // (channels >= cut_len are dead bias parameters (model.py:358): written as exact zeros by the reduce behind the loop)
    } else if (!acc) {
      if (int rc = launch_fill_zero(grad_biases[i], hidden, s)) return rc;
    }

    // dW_i = X_i^T dZ.  The kernel covers up to 304 input channels per pass; wider inputs (the 448-wide image
    // model's first layer) go in column blocks of <= 300, each first copied into a contiguous panel (StackLayout::panel).
    for (int c0 = 0; c0 < kin; c0 += 300) {
      const int w = kin - c0 < 300 ? kin - c0 : 300;
      const float *xs = x;
      int ldxs = ldx;
      if (kin > 304) {
        float *panel = scratch + L.panel;
        const int wp = pad4(w);
        if (wp != w) { set_error("gcn_stack: in_features=%d needs a multiple of 4 past 300", kin); return -1; }
        if (int rc = launch_copy_cols(x, ldx, c0, w, panel, (long long)m, s)) return rc;
        xs = panel;
        ldxs = w;
      }
      DwArgs d{};
      d.x = xs;
      d.ldx = ldxs;
      if (xhyb) {   // (kin = hidden <= 304 here: no panels) staged image [16][hidden], sources: quad-major + row-major parts
        d.ldx = hidden;
        d.ldx_src = rm_ld;
        d.xq = xl;
        d.xq_nvert = n_vert;
        d.xq_quads = qcols / 4;
      }
      if (quad) {
        d.z0q_nvert = n_vert;
        d.z0q_quads = cpad / 4;
      }
      d.z0 = dza;
      d.ldz0 = cpad > 0 ? cpad : 4;
      d.z1 = g;
      d.ldz1 = hidden;
      d.zsplit = cpad;
      d.zeros = zeros;
      d.slab = scratch + L.dw_slab;
      d.m = (int)m;
      d.k_in = w;
      d.n_out = hidden;
      d.bf16 = omode;
      if (x3 && i > 0) {   // hidden layer of a mode-3 stack: the split-operand kernel when it takes the shape
        d.bf16 = 3;
        if (!dw3_ok(d)) d.bf16 = omode;
      }
      {
        ProfScope ps(PROF_DW, s);
        if (int rc = launch_dw(d, s)) return rc;
      }
      if (int rc = launch_slab_reduce_za(scratch + L.dw_slab, dw_images(d), (size_t)w * hidden, (size_t)w * hidden,
                                         (size_t)w * hidden, grad_weights[i] + (size_t)c0 * hidden, acc, s))
        return rc;
    }

    // dX_i = dZ W_i^T  (masked by the ReLU of layer i-1, whose output is X_i)
    const int n_store = i == 0 ? ld_feats : hidden;
    float *wp = scratch + L.wt + L.wt_stride * i;
    const bool l3 = x3 && i > 0;
    if (!batched_images && !l3)
      if (int rc = launch_copy_pad(weights[i], kin, hidden, wp, rowgemm_bt_rows(n_store), pad16(hidden), s)) return rc;
    RowGemmArgs r{};
    r.a0 = dza;
    r.lda0 = cpad > 0 ? cpad : 4;
    if (quad) {   // dZa is quad-major (csrq_kernel<1>)
      r.a0q_nvert = n_vert;
      r.a0q_quads = cpad / 4;
    }
    r.a1 = g;
    r.lda1 = hidden;
    r.ksplit = cpad;
    r.bt = wp;
    r.ldb = pad16(hidden);
    r.zeros = zeros;
    r.m = (int)m;
    r.k = hidden;
    r.n_store = n_store;
    r.bf16 = l3 ? 3 : omode;
    if (i == 0) {
      r.c = grad_feats;
      r.ldc = ld_feats;
      if (int rc = launch_rowgemm(r, EPI_PLAIN, s)) return rc;
    } else {
      r.c = scratch + L.ping[cur ^ 1];
      r.ldc = hidden;
      // X_i is the output of layer i-1: its ReLU signs were saved by that layer's forward launches
      r.maskb = const_cast<uint8_t *>(masks) + (size_t)(i - 1) * mpad * mld;
      r.mld = mld;
      r.moff = cpad / 4;
      r.csplit = cut_len;
      if (quad) {   // gradient columns [0, cpad) go quad-major (unmasked below cut_len) to the next aggregation
        r.c2 = scratch + L.gq;
        r.zq_nvert = n_vert;
        r.zq_quads = cpad / 4;
        r.trash = scratch + L.z3;
      }
      {
        ProfScope ps(PROF_GEMM_DX, s);
        if (int rc = launch_rowgemm(r, EPI_DX_MASK, s)) return rc;
      }
      cur ^= 1;
    }
  }
  // bias gradients of all hidden layers from their partial rows (one per mesh on the channel-sliced path, one per row-walk
  // workgroup otherwise): one launch instead of one per layer (19 x 3.8-5.5 us per stack call)
  if (cut_len > 0)
    if (int rc = reduce_bias_partials(scratch + L.db_slab, db_layer_stride, db_rows, cpad, cut_len, hidden, last, grad_biases, acc, s))
      return rc;
  return 0;
}

// ---- one layer on its own (GCN_layer.forward for the auto-encoder / DDQN layer loops)
namespace {
struct LayerLayout {
  size_t wt, za, ga, dz, panel, dw_slab, db_slab, heavy, total;
};
LayerLayout layer_layout(int batch, int n_vert, int ld_x, int n_out, int cut_len, int need_backward) {
  LayerLayout L{};
  const size_t m = (size_t)batch * n_vert;
  const int cpad = pad4(cut_len), npad = pad4(n_out);
  size_t off = 0;
  auto take = [&](size_t nfloats) {
    const size_t o = off;
    off = align_up(off + nfloats, 64);
    return o;
  };
  const size_t wt_fwd = (size_t)rowgemm_bt_rows(n_out) * pad16(ld_x);
  const size_t wt_bwd = (size_t)rowgemm_bt_rows(ld_x) * pad16(npad);
  L.wt = take(wt_fwd > wt_bwd ? wt_fwd : wt_bwd);
  L.za = take(m * (cpad > 4 ? cpad : 4));  // forward: raw Z of the aggregated channels; backward: 4-float dummy rows
  L.heavy = take(csr_heavy_scratch_ints(n_vert));
  if (need_backward) {
    L.ga = take(m * (cpad > 4 ? cpad : 4));
    L.dz = take(m * npad);
    L.panel = take(ld_x > 304 ? m * 300 : 0);
    const int kin = ld_x > 304 ? 300 : ld_x;
    L.dw_slab = take((size_t)dw_num_slabs(n_out) * kin * n_out);
    L.db_slab = take((size_t)csr_bwd_num_slabs(batch, n_vert) * (cpad > 4 ? cpad : 4));
  }
  L.total = off;
  return L;
}
int check_layer_dims(int ld_x, int in_features, int n_out, int cut_len) {
  if (ld_x % 4 != 0 || ld_x < in_features || ld_x > 600 || in_features < 1) {
    set_error("gcn_layer: ld_x=%d must be a multiple of 4, >= in_features=%d and <= 600", ld_x, in_features);
    return -1;
  }
  if (n_out < 1 || n_out > 304 || cut_len < 0 || cut_len > n_out) {
    set_error("gcn_layer: out_features=%d (1..304) cut_len=%d unsupported", n_out, cut_len);
    return -1;
  }
  if (in_features > 304 && in_features % 4 != 0) {
    set_error("gcn_layer: in_features=%d > 304 must be a multiple of 4", in_features);
    return -1;
  }
  return 0;
}
}  // namespace

size_t a3vt_gcn_layer_scratch_bytes(int batch, int n_vert, int ld_x, int out_features, int cut_len, int need_backward) {
  return layer_layout(batch, n_vert, ld_x, out_features, cut_len, need_backward).total * sizeof(float);
}

int a3vt_gcn_layer_fwd(const float *x, int ld_x, int in_features, const float *weight, const float *bias,
                       int out_features, int cut_len, int relu, const int32_t *rowptr, const int32_t *col,
                       const float *val, int max_degree, int n_vert, int batch, int gemm_bf16, float *y, int ld_y,
                       float *scratch, void *stream) {
  A3VT_CHECK_ARG(x && weight && bias && rowptr && col && val && y && scratch && n_vert > 0 && batch > 0);
  if (int rc = check_layer_dims(ld_x, in_features, out_features, cut_len)) return rc;
  A3VT_CHECK_ARG(ld_y % 4 == 0 && ld_y >= out_features);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  const LayerLayout L = layer_layout(batch, n_vert, ld_x, out_features, cut_len, 0);
  const size_t m = (size_t)batch * n_vert;
  const int cpad = pad4(cut_len);
  float *wt = scratch + L.wt;
  if (int rc = launch_transpose_pad(weight, in_features, out_features, wt, rowgemm_bt_rows(out_features), pad16(ld_x), s))
    return rc;
  RowGemmArgs g{};
  g.a0 = g.a1 = x;
  g.lda0 = g.lda1 = ld_x;
  g.ksplit = ld_x;
  g.bt = wt;
  g.ldb = pad16(ld_x);
  g.zeros = zeros;
  g.m = (int)m;
  g.k = ld_x;
  g.n_store = out_features;
  g.c = y;
  g.ldc = ld_y;
  g.c2 = scratch + L.za;
  g.ldc2 = cpad > 4 ? cpad : 4;
  g.csplit = cut_len;
  g.no_relu = relu ? 0 : 1;
  g.bf16 = gemm_bf16 == 1 ? 1 : 0;   // (mode 3 is a stack mode: a lone layer runs exact)
  if (int rc = launch_rowgemm(g, EPI_FWD_HIDDEN, s)) return rc;
  if (cut_len > 0) {
    int32_t *heavy = nullptr;
    if (max_degree <= 0 || max_degree > csr_heavy_degree()) {
      heavy = reinterpret_cast<int32_t *>(scratch + L.heavy);
      if (int rc = launch_csr_heavy_list(rowptr, n_vert, heavy, s)) return rc;
    }
    if (int rc = launch_csr_fwd(scratch + L.za, g.ldc2, bias, cut_len, rowptr, col, val, heavy, n_vert, batch, y, ld_y,
                                nullptr, 0, relu ? 1 : 0, s))
      return rc;
  }
  return 0;
}

int a3vt_gcn_layer_bwd(const float *x, int ld_x, int in_features, const float *weight, int out_features, int cut_len,
                       int relu, const int32_t *rowptrT, const int32_t *colT, const float *valT, int max_degreeT,
                       int n_vert, int batch, int gemm_bf16, const float *y, int ld_y, const float *grad_y, int ld_gy,
                       float *grad_weight, float *grad_bias, float *grad_x, float *scratch, void *stream) {
  A3VT_CHECK_ARG(x && weight && rowptrT && colT && valT && grad_y && grad_weight && grad_bias && grad_x && scratch);
  A3VT_CHECK_ARG(n_vert > 0 && batch > 0 && ld_gy >= out_features && (!relu || (y && ld_y >= out_features)));
  if (int rc = check_layer_dims(ld_x, in_features, out_features, cut_len)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  const LayerLayout L = layer_layout(batch, n_vert, ld_x, out_features, cut_len, 1);
  const size_t m = (size_t)batch * n_vert;
  const int cpad = pad4(cut_len), npad = pad4(out_features);
  float *ga = scratch + L.ga, *dz = scratch + L.dz;

  // gradient through the activation; aggregated columns to `ga`, the rest straight into the merged dZ rows
  if (int rc = launch_relu_split(grad_y, ld_gy, y, ld_y, relu ? 1 : 0, out_features, cpad, npad, (long long)m, ga, dz, s))
    return rc;
  if (int rc = launch_fill_zero(grad_bias, out_features, s)) return rc;
  if (cut_len > 0) {
    // dZ[:, :c] = A^T G[:, :c] (columns c..cpad pass through), bias gradient = column sums of G[:, :c]
    int32_t *heavyT = nullptr;
    if (max_degreeT <= 0 || max_degreeT > csr_heavy_degree()) {
      heavyT = reinterpret_cast<int32_t *>(scratch + L.heavy);
      if (int rc = launch_csr_heavy_list(rowptrT, n_vert, heavyT, s)) return rc;
    }
    if (int rc = launch_csr_bwd(ga, cpad, cut_len, rowptrT, colT, valT, heavyT, n_vert, batch, dz, npad,
                                scratch + L.db_slab, s))
      return rc;
    if (int rc = launch_slab_reduce(scratch + L.db_slab, csr_bwd_num_slabs(batch, n_vert), cpad, cut_len, grad_bias, s))
      return rc;
  }
  // dW = X^T dZ, in column panels of <= 300 input channels when the input is wider than the kernel covers
  for (int c0 = 0; c0 < in_features; c0 += 300) {
    const int w = in_features > 304 ? (in_features - c0 < 300 ? in_features - c0 : 300) : in_features;
    const float *xs = x;
    int ldxs = ld_x;
    if (in_features > 304) {
      float *panel = scratch + L.panel;
      if (int rc = launch_copy_cols(x, ld_x, c0, w, panel, (long long)m, s)) return rc;
      xs = panel;
      ldxs = w;
    }
    DwArgs d{};
    d.x = xs;
    d.ldx = ldxs;
    d.z0 = scratch + L.za;  // zsplit = 0: never consumed, only staged
    d.ldz0 = 4;
    d.z1 = dz;
    d.ldz1 = npad;
    d.zsplit = 0;
    d.zeros = zeros;
    d.slab = scratch + L.dw_slab;
    d.m = (int)m;
    d.k_in = w;
    d.n_out = out_features;
    d.bf16 = gemm_bf16 == 1 ? 1 : 0;
    if (int rc = launch_dw(d, s)) return rc;
    if (int rc = launch_slab_reduce(scratch + L.dw_slab, dw_num_slabs(out_features), (size_t)w * out_features,
                                    (size_t)w * out_features, grad_weight + (size_t)c0 * out_features, s))
      return rc;
    if (in_features <= 304) break;
  }
  // dX = dZ W^T
  float *wp = scratch + L.wt;
  if (int rc = launch_copy_pad(weight, in_features, out_features, wp, rowgemm_bt_rows(ld_x), pad16(npad), s)) return rc;
  RowGemmArgs r{};
  r.a0 = scratch + L.za;
  r.lda0 = 4;
  r.ksplit = 0;
  r.a1 = dz;
  r.lda1 = npad;
  r.bt = wp;
  r.ldb = pad16(npad);
  r.zeros = zeros;
  r.m = (int)m;
  r.k = npad;
  r.n_store = ld_x;
  r.bf16 = gemm_bf16 == 1 ? 1 : 0;
  r.c = grad_x;
  r.ldc = ld_x;
  return launch_rowgemm(r, EPI_PLAIN, s);
}

int a3vt_wt_rows(int n_out) { return rowgemm_bt_rows(n_out); }
int a3vt_wt_ld(int k) { return pad16(k); }

int a3vt_transpose_weight(const float *w, int k, int n_out, float *wt, void *stream) {
  A3VT_CHECK_ARG(w && wt && k > 0 && n_out > 0 && n_out <= 304);
  return launch_transpose_pad(w, k, n_out, wt, rowgemm_bt_rows(n_out), pad16(k), static_cast<hipStream_t>(stream));
}

int a3vt_rowgemm(const float *a, int lda, int m, int k, const float *wt, int n_out, int gemm_bf16, float *c, int ldc,
                 void *stream) {
  A3VT_CHECK_ARG(a && wt && c && m > 0 && k > 0 && k % 4 == 0 && lda >= k && ldc >= n_out && n_out <= 304);
  RowGemmArgs g{};
  g.a0 = g.a1 = a;
  g.lda0 = g.lda1 = lda;
  g.ksplit = k;
  g.bt = wt;
  g.ldb = pad16(k);
  g.zeros = zero_page();
  A3VT_CHECK_ARG(g.zeros != nullptr);
  g.m = m;
  g.k = k;
  g.n_store = n_out;
  g.c = c;
  g.ldc = ldc;
  g.bf16 = gemm_bf16 == 1 ? 1 : 0;   // (mode 3 is a stack mode: a lone layer runs exact)
  return launch_rowgemm(g, EPI_PLAIN, static_cast<hipStream_t>(stream));
}

size_t a3vt_posenc_param_count(int input_size) { return posenc_param_count(input_size); }
size_t a3vt_posenc_scratch_bytes(int m, int input_size) {
  return (size_t)posenc_num_slabs(m) * posenc_param_count(input_size) * sizeof(float);
}
int a3vt_posenc_mask_fwd(const float *verts, const float *mask, int m, int input_size, const float *pe_params,
                         float *feats, int ld_feats, void *stream) {
  A3VT_CHECK_ARG(verts && mask && pe_params && feats && m > 0 && ld_feats >= input_size);
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_posenc_fwd(verts, mask, m, input_size, pe_params, feats, ld_feats, static_cast<hipStream_t>(stream));
}
int a3vt_posenc_mask_bwd(const float *verts, const float *mask, int m, int input_size, const float *pe_params,
                         const float *grad_feats, int ld_feats, float *grad_verts, float *grad_params, float *scratch,
                         void *stream) {
  A3VT_CHECK_ARG(verts && mask && pe_params && grad_feats && grad_verts && grad_params && scratch && m > 0);
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_posenc_bwd(verts, mask, m, input_size, pe_params, grad_feats, ld_feats, grad_verts, grad_params,
                           scratch, static_cast<hipStream_t>(stream));
}

// ---- wide inputs (I = 448 of the image models): posenc_wide.hip
int a3vt_posenc_wide_supported(int input_size) { return posenc_wide_supported(input_size) ? 1 : 0; }
size_t a3vt_posenc_wide_acts_bytes(int m, int input_size) {
  return posenc_wide_supported(input_size) && m > 0 ? posenc_wide_acts_floats(m, input_size) * sizeof(float) : 0;
}
size_t a3vt_posenc_wide_scratch_bytes(int m, int input_size, int need_backward) {
  return posenc_wide_supported(input_size) && m > 0 ? posenc_wide_scratch_floats(m, input_size, need_backward) * sizeof(float) : 0;
}
int a3vt_posenc_wide_fwd(const float *verts, const float *mask, int m, int input_size, const float *pe_params, float *feats,
                         int ld_feats, float *acts, float *scratch, int gemm_bf16, void *stream) {
  A3VT_CHECK_ARG(verts && mask && pe_params && feats && acts && scratch && m > 0);
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  A3VT_CHECK_ARG(gemm_bf16 == 0 || gemm_bf16 == 1);
  return launch_posenc_wide_fwd(verts, mask, m, input_size, pe_params, feats, ld_feats, acts, scratch, zeros, gemm_bf16,
                                static_cast<hipStream_t>(stream));
}
int a3vt_posenc_wide_bwd(const float *verts, const float *mask, int m, int input_size, const float *pe_params,
                         const float *grad_feats, int ld_feats, const float *acts, float *grad_verts, float *grad_params,
                         float *scratch, int gemm_bf16, void *stream) {
  A3VT_CHECK_ARG(verts && mask && pe_params && grad_feats && acts && grad_verts && grad_params && scratch && m > 0);
  const float *zeros = zero_page();
  A3VT_CHECK_ARG(zeros != nullptr);
  A3VT_CHECK_ARG(gemm_bf16 == 0 || gemm_bf16 == 1);
  return launch_posenc_wide_bwd(verts, mask, m, input_size, pe_params, grad_feats, ld_feats, acts, grad_verts, grad_params,
                                scratch, zeros, gemm_bf16, static_cast<hipStream_t>(stream));
}

int a3vt_vertex_update(const float *verts_in, const float *update, int batch, int n_vert, int n_vision,
                       float *verts_out, void *stream) {
  A3VT_CHECK_ARG(verts_in && update && verts_out && batch > 0 && n_vert > 0 && n_vision >= 0 && n_vision <= n_vert);
  return launch_vertex_update(verts_in, update, batch, n_vert, n_vision, verts_out, static_cast<hipStream_t>(stream));
}

int a3vt_face_cdf(const float *verts, const int32_t *faces, int batch, int n_vert, int n_faces, float *cdf,
                  void *stream) {
  A3VT_CHECK_ARG(verts && faces && cdf && batch > 0 && n_vert > 0 && n_faces > 0);
  return launch_face_cdf(verts, faces, batch, n_vert, n_faces, cdf, static_cast<hipStream_t>(stream));
}

int a3vt_sample_points_fwd(const float *verts, const int32_t *faces, const float *cdf, int batch, int n_vert,
                           int n_faces, int draws, int num, const int32_t *face_idx_in, const float *u_in,
                           const float *v_in, uint64_t seed, uint64_t offset, float *points, int32_t *face_idx_out,
                           float *u_out, float *v_out, void *stream) {
  A3VT_CHECK_ARG(verts && faces && points && batch > 0 && n_vert > 0 && n_faces > 0 && draws > 0 && num > 0);
  ProfScope psc(PROF_LOSS, static_cast<hipStream_t>(stream));
  return launch_sample_fwd(verts, faces, cdf, batch, n_vert, n_faces, draws, num, face_idx_in, u_in, v_in, seed,
                           offset, points, face_idx_out, u_out, v_out, static_cast<hipStream_t>(stream));
}

int a3vt_sample_points_bwd(const int32_t *faces, int batch, int n_vert, int n_faces, int draws, int num,
                           const int32_t *face_idx, const float *u, const float *v, const float *grad_points,
                           float *grad_verts, void *stream) {
  A3VT_CHECK_ARG(faces && face_idx && u && v && grad_points && grad_verts && batch > 0 && draws > 0 && num > 0);
  ProfScope psc(PROF_LOSS, static_cast<hipStream_t>(stream));
  return launch_sample_bwd(faces, batch, n_vert, n_faces, draws, num, face_idx, u, v, grad_points, grad_verts,
                           static_cast<hipStream_t>(stream));
}

size_t a3vt_chamfer_scratch_bytes(int draws, int batch, int p, int q) {
  (void)p;
  return draws > 0 && batch > 0 && q > 0 ? chamfer_scratch_bytes(draws, batch, q) : 0;
}

int a3vt_chamfer_fwd(const float *x, const float *y, int draws, int batch, int p, int q, float *dist_xy,
                     int32_t *idx_xy, float *dist_yx, int32_t *idx_yx, float *cd, void *scratch, void *stream) {
  A3VT_CHECK_ARG(x && y && dist_xy && idx_xy && dist_yx && idx_yx && cd);
  // brute force only: the scratch of this entry point is a3vt_chamfer_scratch_bytes() (the column minima of the sweep)
  ProfScope psc(PROF_SEARCH, static_cast<hipStream_t>(stream));
  return launch_chamfer_fwd(x, y, draws, batch, p, q, dist_xy, idx_xy, dist_yx, idx_yx, cd, scratch,
                            scratch ? a3vt_chamfer_scratch_bytes(draws, batch, p, q) : 0,
                            scratch ? NN_BRUTE_SWEEP : NN_BRUTE_TWO_PASS, static_cast<hipStream_t>(stream));
}

size_t a3vt_chamfer_workspace_bytes(int draws, int batch, int p, int q) {
  return draws > 0 && batch > 0 && p > 0 && q > 0 ? chamfer_workspace_bytes(draws, batch, p, q) : 0;
}

int a3vt_chamfer_fwd_ws(const float *x, const float *y, int draws, int batch, int p, int q, float *dist_xy,
                        int32_t *idx_xy, float *dist_yx, int32_t *idx_yx, float *cd, void *workspace,
                        size_t workspace_bytes, int algo, void *stream) {
  A3VT_CHECK_ARG(x && y && dist_xy && idx_xy && dist_yx && idx_yx && cd);
  ProfScope psc(PROF_SEARCH, static_cast<hipStream_t>(stream));
  return launch_chamfer_fwd(x, y, draws, batch, p, q, dist_xy, idx_xy, dist_yx, idx_yx, cd, workspace,
                            workspace ? workspace_bytes : 0, algo, static_cast<hipStream_t>(stream));
}

int a3vt_chamfer_fwd_shared(const float *x, const float *y, int draws, int batch, int y_batch, int p, int q, float *dist_xy,
                            int32_t *idx_xy, float *dist_yx, int32_t *idx_yx, float *cd, void *workspace,
                            size_t workspace_bytes, int algo, void *stream) {
  A3VT_CHECK_ARG(x && y && dist_xy && idx_xy && dist_yx && idx_yx && cd && y_batch > 0);
  ProfScope psc(PROF_SEARCH, static_cast<hipStream_t>(stream));
  return launch_chamfer_fwd(x, y, draws, batch, p, q, dist_xy, idx_xy, dist_yx, idx_yx, cd, workspace,
                            workspace ? workspace_bytes : 0, algo, static_cast<hipStream_t>(stream), y_batch);
}

int a3vt_chamfer_bwd(const float *x, const float *y, int draws, int batch, int p, int q, const int32_t *idx_xy,
                     const int32_t *idx_yx, const float *grad_cd, float *grad_x, float *grad_y, void *stream) {
  A3VT_CHECK_ARG(x && y && idx_xy && idx_yx && grad_cd && grad_x && draws > 0 && batch > 0 && p > 0 && q > 0);
  ProfScope psc(PROF_LOSS, static_cast<hipStream_t>(stream));
  return launch_chamfer_bwd(x, y, draws, batch, p, q, idx_xy, idx_yx, grad_cd, grad_x, grad_y,
                            static_cast<hipStream_t>(stream));
}

static int fill_pool_args(PoolArgs &a, const float *verts, int batch, int n_vert, const float *proj, int n_maps,
                          const float *const *maps, const int *chans, const int *heights, const int *widths, int ld) {
  A3VT_CHECK_ARG(verts && proj && maps && chans && heights && widths && batch > 0 && n_vert > 0);
  A3VT_CHECK_ARG(n_maps >= 1 && n_maps <= kMaxMaps);
  a.verts = verts;
  for (int i = 0; i < 12; ++i) a.proj[i] = proj[i];
  a.batch = batch;
  a.n_vert = n_vert;
  a.n_maps = n_maps;
  int off = 0;
  for (int k = 0; k < n_maps; ++k) {
    A3VT_CHECK_ARG(maps[k] != nullptr);
    a.maps[k] = maps[k];
    a.C[k] = chans[k];
    a.H[k] = heights[k];
    a.W[k] = widths[k];
    a.off[k] = off;
    off += chans[k];
  }
  a.ld = ld;
  return 0;
}

int a3vt_image_pool_fwd(const float *verts, int batch, int n_vert, const float *proj, int n_maps,
                        const float *const *maps, const int *chans, const int *heights, const int *widths,
                        float *feats, int ld_feats, void *stream) {
  PoolArgs a{};
  if (int rc = fill_pool_args(a, verts, batch, n_vert, proj, n_maps, maps, chans, heights, widths, ld_feats)) return rc;
  A3VT_CHECK_ARG(feats != nullptr);
  a.feats = feats;
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_pool_fwd(a, static_cast<hipStream_t>(stream));
}

int a3vt_image_pool_fwd_add(const float *verts, int batch, int n_vert, const float *proj, int n_maps,
                            const float *const *maps, const int *chans, const int *heights, const int *widths,
                            const float *base, float *feats, int ld_feats, void *stream) {
  PoolArgs a{};
  if (int rc = fill_pool_args(a, verts, batch, n_vert, proj, n_maps, maps, chans, heights, widths, ld_feats)) return rc;
  A3VT_CHECK_ARG(feats != nullptr && base != nullptr);
  int width = 0;
  for (int k = 0; k < n_maps; ++k) width += chans[k];
  A3VT_CHECK_ARG(width == ld_feats);     // every column of a row is written: feats = base + pooled
  a.feats = feats;
  a.base = base;
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_pool_fwd(a, static_cast<hipStream_t>(stream));
}

int a3vt_image_pool_bwd(const float *verts, int batch, int n_vert, const float *proj, int n_maps,
                        const float *const *maps, const int *chans, const int *heights, const int *widths,
                        const float *grad_feats, int ld_feats, float *const *grad_maps, float *grad_verts,
                        void *stream) {
  PoolArgs a{};
  if (int rc = fill_pool_args(a, verts, batch, n_vert, proj, n_maps, maps, chans, heights, widths, ld_feats)) return rc;
  A3VT_CHECK_ARG(grad_feats && grad_maps && grad_verts);
  for (int k = 0; k < n_maps; ++k) {
    A3VT_CHECK_ARG(grad_maps[k] != nullptr);
    a.gmaps[k] = grad_maps[k];
  }
  a.gfeats = grad_feats;
  a.gverts = grad_verts;
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_pool_bwd(a, static_cast<hipStream_t>(stream));
}

size_t a3vt_bias_grad_scratch_bytes(long long rows, int channels) {
  if (rows <= 0 || channels <= 0) return 0;
  return (size_t)bias_grad_wgs(rows * channels, channels) * channels * sizeof(float);
}

int a3vt_bias_grad_nhwc(const void *grad, int bf16, long long rows, int channels, float *out, void *scratch,
                        size_t scratch_bytes, void *stream) {
  A3VT_CHECK_ARG(grad && out && scratch);
  A3VT_CHECK_ARG(rows > 0 && channels > 0 && (bf16 == 0 || bf16 == 1));
  A3VT_CHECK_ARG((reinterpret_cast<uintptr_t>(grad) & 15) == 0);
  const size_t need = a3vt_bias_grad_scratch_bytes(rows, channels);
  A3VT_CHECK_ARG(need > 0 && scratch_bytes >= need);
  return launch_bias_grad(grad, bf16, rows, channels, out, static_cast<float *>(scratch), static_cast<hipStream_t>(stream));
}

size_t a3vt_bnrelu_scratch_bytes(int channels) { return channels > 0 ? bnrelu_scratch_bytes(channels) : 0; }

int a3vt_bnrelu_fwd(const void *x, long long rows, int channels, const float *gamma, const float *beta, const float *pre_bias,
                    float eps, float momentum, float *running_mean, float *running_var, long long *num_batches_tracked, void *y,
                    float *save, void *scratch, size_t scratch_bytes, void *stream) {
  A3VT_CHECK_ARG(x && y && gamma && beta && save && scratch);
  A3VT_CHECK_ARG(rows >= 2 && channels > 0 && rows <= (1ll << 40) / channels);
  A3VT_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr));
  A3VT_CHECK_ARG(eps >= 0.f && momentum >= 0.f && momentum <= 1.f);
  A3VT_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(scratch)) & 15) == 0);
  A3VT_CHECK_ARG(bnrelu_wgs(rows * channels, channels, 4, 1024) > 0 && scratch_bytes >= bnrelu_scratch_bytes(channels));
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_bnrelu_fwd(x, rows, channels, gamma, beta, pre_bias, eps, momentum, running_mean, running_var, num_batches_tracked, y,
                           save, scratch, static_cast<hipStream_t>(stream));
}

int a3vt_bnrelu_bwd(const void *dy, const void *x, long long rows, int channels, const float *save, void *dx, float *dgamma,
                    float *dbeta, float *dx_colsum, void *scratch, size_t scratch_bytes, void *stream) {
  A3VT_CHECK_ARG(dy && x && save && dx && dgamma && dbeta && scratch);
  A3VT_CHECK_ARG(rows >= 2 && channels > 0 && rows <= (1ll << 40) / channels);
  A3VT_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) |
                   reinterpret_cast<uintptr_t>(scratch)) & 15) == 0);
  A3VT_CHECK_ARG(bnrelu_wgs(rows * channels, channels, 4, 1024) > 0 && scratch_bytes >= bnrelu_scratch_bytes(channels));
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_bnrelu_bwd(dy, x, rows, channels, save, dx, dgamma, dbeta, dx_colsum, scratch, static_cast<hipStream_t>(stream));
}

int a3vt_cast_weights_bf16(int n, const float *const *src, void *const *dst, const long long *outer, const int *inner,
                           const int *hw, void *stream) {
  A3VT_CHECK_ARG(n >= 0 && n <= kCastBatchMax);
  A3VT_CHECK_ARG(n == 0 || (src && dst && outer && inner && hw));
  CastBatch b;
  b.n = n;
  b.start[0] = 0;
  for (int k = 0; k < n; ++k) {
    A3VT_CHECK_ARG(src[k] && dst[k] && outer[k] > 0 && inner[k] > 0 && hw[k] > 0);
    A3VT_CHECK_ARG(outer[k] <= (1ll << 31) / inner[k] / hw[k]);
    b.src[k] = src[k];
    b.dst[k] = static_cast<uint16_t *>(dst[k]);
    b.inner[k] = inner[k];
    b.hw[k] = hw[k];
    b.start[k + 1] = b.start[k] + outer[k] * inner[k] * hw[k];
  }
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_cast_weights(b, static_cast<hipStream_t>(stream));
}

int a3vt_conv5_supported(int cin, int cout, int stride) { return conv5_shape_ok(cin, cout, stride) ? 1 : 0; }

size_t a3vt_conv5_image_bytes(int cout, int cin, int flip) {
  if (cin == 3 && flip) return cout == 16 ? conv5_weight_image_bytes(16, 16) : 0;     // (the input gradient of layer 1)
  if (cin == 3) return cout == 3 || cout == 16 ? conv5_weight_image_bytes(cout, 3) : 0;
  if ((cin != 16 && cin != 32) || (cout != 16 && cout != 32)) return 0;
  return flip ? conv5_weight_image_bytes(cin, cout) : conv5_weight_image_bytes(cout, cin);
}

int a3vt_conv5_weight_image(const float *weight, int cout, int cin, int flip, void *image, void *stream) {
  A3VT_CHECK_ARG(weight && image && (flip == 0 || flip == 1));
  A3VT_CHECK_ARG(((cin == 16 || cin == 32) && (cout == 16 || cout == 32)) || (cin == 3 && flip == 0 && (cout == 3 || cout == 16)) ||
                 (cin == 3 && flip == 1 && cout == 16));
  A3VT_CHECK_ARG((reinterpret_cast<uintptr_t>(image) & 15) == 0);
  return launch_conv5_weight_image(weight, flip, cout, cin, image, static_cast<hipStream_t>(stream));
}

int a3vt_conv5_nhwc(const void *x, int batch, int height, int width, int cin, int cout, int stride, int pad, const void *image,
                    const float *bias, void *y, void *stream) {
  A3VT_CHECK_ARG(x && image && y);
  A3VT_CHECK_ARG(conv5_shape_ok(cin, cout, stride));
  A3VT_CHECK_ARG(batch > 0 && height > 0 && width > 0 && pad >= 0 && pad <= 4);
  A3VT_CHECK_ARG(height + 2 * pad >= 5 && width + 2 * pad >= 5);
  A3VT_CHECK_ARG((long long)batch * height * width <= (1ll << 31) / 32);
  A3VT_CHECK_ARG(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & (cin == 3 ? 1 : 15)) == 0);
  A3VT_CHECK_ARG((reinterpret_cast<uintptr_t>(image) & 15) == 0 && (cout == 3 || (reinterpret_cast<uintptr_t>(y) & 15) == 0));
  A3VT_CHECK_ARG(bias == nullptr || (reinterpret_cast<uintptr_t>(bias) & (cout == 3 ? 3 : 15)) == 0);
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_conv5(x, batch, height, width, cin, cout, stride, pad, image, bias, y, static_cast<hipStream_t>(stream));
}

int a3vt_conv5_input_grad_3x16s2(const void *grad_out, int batch, int out_height, int out_width, const void *image, void *grad_in,
                                 void *stream) {
  A3VT_CHECK_ARG(grad_out && image && grad_in && batch > 0 && out_height > 0 && out_width > 0);
  A3VT_CHECK_ARG((long long)batch * (2 * out_height + 2) * (2 * out_width + 2) <= (1ll << 31) / 16);
  A3VT_CHECK_ARG(((reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(image)) & 15) == 0);
  A3VT_CHECK_ARG((reinterpret_cast<uintptr_t>(grad_in) & 1) == 0);
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_conv5_up3(grad_out, batch, out_height, out_width, image, grad_in, static_cast<hipStream_t>(stream));
}

size_t a3vt_conv5_wrw_scratch_bytes(int cin, int cout) {
  return ((cin == 16 || cin == 32) && (cout == 16 || cout == 32)) || (cin == 3 && (cout == 3 || cout == 16)) ? conv5_wrw_scratch_bytes(cin, cout) : 0;
}

int a3vt_conv5_weight_grad(const void *x, const void *grad_out, int batch, int height, int width, int cin, int cout, int stride,
                           float *grad_weight, void *scratch, size_t scratch_bytes, void *stream) {
  A3VT_CHECK_ARG(x && grad_out && grad_weight && scratch);
  A3VT_CHECK_ARG(conv5_shape_ok(cin, cout, stride));
  A3VT_CHECK_ARG(batch > 0 && height >= 3 && width >= 3 && (long long)batch * height * width <= (1ll << 31) / 32);
  A3VT_CHECK_ARG((reinterpret_cast<uintptr_t>(scratch) & 15) == 0);
  A3VT_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & (cin == 3 ? 1 : 15)) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & (cout == 3 ? 1 : 15)) == 0);
  A3VT_CHECK_ARG(scratch_bytes >= conv5_wrw_scratch_bytes(cin, cout));
  ProfScope psc(PROF_ENC, static_cast<hipStream_t>(stream));
  return launch_conv5_wrw(x, grad_out, batch, height, width, cin, cout, stride, grad_weight, scratch, static_cast<hipStream_t>(stream));
}

int a3vt_adam_chunk_elems(void) { return adam_chunk_elems(); }

int a3vt_adam_step(void *const *param, const void *const *grad, void *const *exp_avg, void *const *exp_avg_sq,
                   const long long *numel, const int *chunk_tensor, const long long *chunk_off, int n_chunks, double lr, double beta1,
                   double beta2, double eps, double weight_decay, long long step, void *stream) {
  A3VT_CHECK_ARG(n_chunks >= 0 && step >= 1);
  A3VT_CHECK_ARG(n_chunks == 0 || (param && grad && exp_avg && exp_avg_sq && numel && chunk_tensor && chunk_off));
  A3VT_CHECK_ARG(lr >= 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0);
  ProfScope psc(PROF_OPT, static_cast<hipStream_t>(stream));
  return launch_adam(param, grad, exp_avg, exp_avg_sq, numel, chunk_tensor, chunk_off, n_chunks, lr, beta1, beta2, eps, weight_decay, step,
                     static_cast<hipStream_t>(stream));
}

int a3vt_profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  return 0;
}

int a3vt_profile_read_classes(double *total_ms, int *count, int n) {
  A3VT_CHECK_ARG(total_ms && count && n >= 0);
  for (int c = 0; c < n; ++c) { total_ms[c] = 0.0; count[c] = 0; }
  for (int i = 0; i < g_prof.used; ++i) {
    if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess) { set_error("profile_read: event sync failed"); return -2; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]) != hipSuccess) continue;
    if (g_prof.cls[i] >= n) continue;
    total_ms[g_prof.cls[i]] += ms;
    count[g_prof.cls[i]] += 1;
  }
  g_prof.used = 0;
  return PROF_CLASSES;
}

int a3vt_profile_read(double *total_ms, int *count) {
  const int rc = a3vt_profile_read_classes(total_ms, count, 3);
  return rc < 0 ? rc : 0;
}

int a3vt_dbg_path_counts(long long *counts, int n, int reset) {
  A3VT_CHECK_ARG(n >= 0 && (counts != nullptr || n == 0));
  for (int i = 0; i < n; ++i) counts[i] = i < PATH_COUNT ? g_path_counts[i].load(std::memory_order_relaxed) : 0;
  if (reset)
    for (auto &c : g_path_counts) c.store(0, std::memory_order_relaxed);
  return PATH_COUNT;
}

int a3vt_dbg_csr_algo(int algo) {
  A3VT_CHECK_ARG(algo >= 0 && algo <= 2);
  g_csr_algo = algo;
  return 0;
}

int a3vt_split3_bf16(const float *x, size_t n, uint16_t *hi, uint16_t *mid, uint16_t *lo, void *stream) {
  A3VT_CHECK_ARG(n == 0 || (x && hi && mid && lo));
  return launch_split3(x, n, hi, mid, lo, static_cast<hipStream_t>(stream));
}

int a3vt_check_finite(const float *data, size_t n, int32_t *flag, void *stream) {
  A3VT_CHECK_ARG(data && flag);
  return launch_check_finite(data, n, flag, static_cast<hipStream_t>(stream));
}

}  // extern "C"
