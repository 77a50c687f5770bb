// One Block of PT-v3 (point_transformer_v3m1_base.py:318-338, pre-norm form with LayerNorm / GELU) as TWO native calls
// for training: ptv3_block_train_fwd runs the forward statements and leaves the activations the backward needs,
// ptv3_block_train_bwd runs the whole backward.  Arithmetic and kernel sequence are exactly those of the per-op
// composition in ptv3_hip/autograd.py (BlockFn, kept as the fallback and as the checker of this file's tests); what
// changes is WHO issues the ~12 + ~30 launches: a training step of the fork model is ~1800 launches, and issued one by
// one through Python wrappers (argument checks, output allocation, ctypes marshalling: ~12 us each) the host, not the
// GPU, set the step time (25.1 ms wall for 20.6 ms of kernels).  Here the host cost per launch is the launch itself.
//
//   c  = LN0(lin(conv(conv_feat)));  f1 = feat + c
//   f2 = f1 + mask1 * proj(attn(qkv(LN1(f1))))
//   out = f2 + mask2 * fc2(GELU(fc1(LN2(f2))))
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

// deferred slab reductions (backward.hip): the reductions of the calls between begin and flush run as one launch
void slab_defer_begin(void* storage);
int slab_defer_flush(hipStream_t s);
size_t slab_defer_storage_bytes();
// deferred weight gradients (backward.hip): the ptv3_gemm_tn calls between begin and flush run as one grouped launch; their operands must stay alive until the flush
void tn_defer_begin(void* storage);
int tn_defer_flush(hipStream_t s);
size_t tn_defer_storage_bytes();

// out[i][:] = (skip ? skip[i][:] : 0) + f[i] * x[i][:], f[i] = u[i] < keep ? 1 / keep : 0: timm's DropPath on an (N, C)
// matrix (a Bernoulli(keep) factor per point, scaled by 1 / keep) from a uniform draw u; fp32 fma, one rounding
template <typename T>
__global__ void __launch_bounds__(256) rowscale_add_kernel(const T* __restrict__ x, const float* __restrict__ u,
                                                           float keep, float inv_keep, const T* __restrict__ skip,
                                                           T* __restrict__ out, int64_t total4, int c4) {
  typedef typename Vec4<T>::type V4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float f = u[i / c4] < keep ? inv_keep : 0.f;
  float xv[4], sv[4] = {0.f, 0.f, 0.f, 0.f};
  unpack4<T>(reinterpret_cast<const V4*>(x)[i], xv);
  if (skip) unpack4<T>(reinterpret_cast<const V4*>(skip)[i], sv);
  reinterpret_cast<V4*>(out)[i] = pack4<T>(fmaf(f, xv[0], sv[0]), fmaf(f, xv[1], sv[1]), fmaf(f, xv[2], sv[2]),
                                           fmaf(f, xv[3], sv[3]));
}

static int rowscale_add(const void* x, const float* u, float keep, const void* skip, void* out, int64_t m, int c,
                        int dtype, hipStream_t s) {
  const float inv_keep = keep > 0.f ? 1.0f / keep : 0.f;
  const int64_t total4 = m * c / 4;
  if (total4 == 0) return PTV3_OK;
  dim3 grid((unsigned)cdiv(total4, 256));
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL(rowscale_add_kernel<float>, grid, dim3(256), 0, s, (const float*)x, u, keep, inv_keep,
                       (const float*)skip, (float*)out, total4, c / 4);
  else
    hipLaunchKernelGGL(rowscale_add_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, u, keep, inv_keep,
                       (const __bf16*)skip, (__bf16*)out, total4, c / 4);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

struct SlabSegsStorage { alignas(16) unsigned char bytes[1024]; };   // >= slab_defer_storage_bytes(), checked at use
struct TnQueueStorage { alignas(16) unsigned char bytes[2048]; };    // >= tn_defer_storage_bytes(), checked at use

struct Bump {   // 256-byte aligned carving of the caller's workspace
  char* base; size_t size, at;
  void* take(size_t bytes) {
    const size_t a = (at + 255) & ~(size_t)255;
    at = a + bytes;
    return at <= size ? base + a : nullptr;
  }
};

static size_t esize(int dtype) { return dtype == PTV3_F32 ? 4 : 2; }

// scratch shared by every op of a call (each op has finished with it when the next one starts: one stream)
static size_t op_scratch_bytes(const ptv3_block_train* b, bool backward) {
  const int64_t n = b->n;
  const int c = b->c, h = b->hidden, k = b->kvol;
  size_t s = ptv3_gemm_workspace_bytes(n, c, c, k, b->dtype);
  auto up = [&](size_t v) { if (v > s) s = v; };
  up(ptv3_gemm_workspace_bytes(n, c, c, 1, b->dtype));
  up(ptv3_gemm_workspace_bytes(n, c, 3 * c, 1, b->dtype));
  up(ptv3_gemm_workspace_bytes(n, c, h, 1, b->dtype));
  up(ptv3_gemm_workspace_bytes(n, h, c, 1, b->dtype));
  if (backward) {
    up(ptv3_gemm_workspace_bytes(n, 3 * c, c, 1, b->dtype));
    up(ptv3_gemm_tn_workspace_bytes(n, c, c, k));
    up(ptv3_gemm_tn_workspace_bytes(n, c, c, 1));
    up(ptv3_gemm_tn_workspace_bytes(n, 3 * c, c, 1));
    up(ptv3_gemm_tn_workspace_bytes(n, h, c, 1));
    up(ptv3_gemm_tn_workspace_bytes(n, c, h, 1));
    up(ptv3_col_reduce_workspace_bytes(n, c));
    up(ptv3_window_attn_bwd_workspace_bytes(n, b->n_pad, c, b->heads, b->dtype));
  }
  return s;
}

static int check_block(const ptv3_block_train* b, const char* what) {
  PTV3_REQUIRE(b != nullptr, "%s: NULL descriptor", what);
  PTV3_REQUIRE(b->dtype == PTV3_F32 || b->dtype == PTV3_BF16, "%s: dtype %d", what, b->dtype);
  const int gran = b->dtype == PTV3_F32 ? 4 : 8;
  PTV3_REQUIRE(b->n >= 0 && b->c > 0 && b->hidden > 0 && b->c % gran == 0 && b->hidden % gran == 0,
               "%s: n=%lld c=%d hidden=%d (channels in multiples of %d)", what, (long long)b->n, b->c, b->hidden, gran);
  PTV3_REQUIRE(b->heads > 0 && b->c % b->heads == 0, "%s: %d heads for %d channels", what, b->heads, b->c);
  PTV3_REQUIRE(b->kvol > 1 && b->nbr != nullptr, "%s: the xCPE convolution needs its neighbour table", what);
  PTV3_REQUIRE(b->win_order && b->win_inverse, "%s: window maps are required", what);
  return PTV3_OK;
}

}  // namespace ptv3

using namespace ptv3;

#define TRY(call)                    \
  do {                               \
    const int rc__ = (call);         \
    if (rc__ != PTV3_OK) return rc__; \
  } while (0)

extern "C" size_t ptv3_block_train_workspace_bytes(const ptv3_block_train* b, int backward) {
  if (!b) return 0;
  const size_t e = esize(b->dtype);
  const size_t nc = (size_t)b->n * b->c * e + 256, nh = (size_t)b->n * b->hidden * e + 256;
  size_t s = op_scratch_bytes(b, backward != 0) + 256;
  if (!backward) return s + nc;                 // p / m before the DropPath factor
  // dm, dh (reused as dh0), dt5, df2, dp, da, dqkv, dt3, df1, dc2, dc1; then one slab region per deferred reduction
  s += nh + 9 * nc + 3 * nc;
  const int c = b->c, h = b->hidden;
  s += ptv3_gemm_tn_workspace_bytes(b->n, c, h, 1) + ptv3_gemm_tn_workspace_bytes(b->n, h, c, 1) +
       2 * ptv3_gemm_tn_workspace_bytes(b->n, c, c, 1) + ptv3_gemm_tn_workspace_bytes(b->n, 3 * c, c, 1) +
       ptv3_gemm_tn_workspace_bytes(b->n, c, c, b->kvol) + 3 * ptv3_col_reduce_workspace_bytes(b->n, c) + 9 * 256;
  return s;
}

extern "C" int ptv3_block_train_fwd(const ptv3_block_train* b, void* stream) {
  TRY(check_block(b, "block_train_fwd"));
  if (b->n == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = b->n;
  const int c = b->c, hd = b->hidden, dt = b->dtype;
  Bump ws{(char*)b->workspace, b->workspace_bytes, 0};
  const size_t sb = op_scratch_bytes(b, false);
  void* scratch = ws.take(sb);
  void* tmp = ws.take((size_t)n * c * esize(dt));
  PTV3_REQUIRE(scratch && tmp, "block_train_fwd: workspace too small (%zu bytes)", b->workspace_bytes);
  const void* xin = b->conv_feat ? b->conv_feat : b->feat;
  TRY(ptv3_gemm(xin, b->w_conv, b->c1, n, c, c, b->kvol, b->nbr, b->row_order, b->b_conv, nullptr, nullptr,
                PTV3_ACT_NONE, nullptr, nullptr, nullptr, dt, scratch, sb, s));
  TRY(ptv3_gemm(b->c1, b->w_lin, b->c2, n, c, c, 1, nullptr, nullptr, b->b_lin, nullptr, nullptr, PTV3_ACT_NONE, nullptr,
                nullptr, nullptr, dt, scratch, sb, s));
  // f1 = feat + LN0(c2) and t3 = LN1(f1) in one pass over the rows
  TRY(ptv3_layernorm(b->c2, b->g0, b->b0, b->feat, b->f1, b->g1, b->b1, b->t3, n, c, b->eps, dt, s));
  TRY(ptv3_gemm(b->t3, b->w_qkv, b->qkv, n, c, 3 * c, 1, nullptr, nullptr, b->b_qkv, nullptr, nullptr, PTV3_ACT_NONE,
                nullptr, nullptr, nullptr, dt, scratch, sb, s));
  if (b->attn_lse)     // the backward takes the log-sum-exp rows from here instead of recomputing them
    TRY(ptv3_window_attn_train_fwd(b->qkv, b->win_order, b->win_inverse, b->cu_seqlens, b->num_windows, b->a, b->attn_lse,
                                   n, b->n_pad, c, b->heads, b->patch, b->scale, b->sum_len_sq, dt, s));
  else if (b->cu_seqlens)
    TRY(ptv3_window_attn_varlen_fwd(b->qkv, b->win_order, b->win_inverse, b->cu_seqlens, b->num_windows, b->a, n,
                                    b->n_pad, c, b->heads, b->patch, b->scale, b->sum_len_sq, dt, s));
  else
    TRY(ptv3_window_attn_fwd(b->qkv, b->win_order, b->win_inverse, b->a, n, b->n_pad, c, b->heads, b->patch, b->scale,
                             nullptr, dt, s));
  if (b->mask1) {
    TRY(ptv3_gemm(b->a, b->w_proj, tmp, n, c, c, 1, nullptr, nullptr, b->b_proj, nullptr, nullptr, PTV3_ACT_NONE,
                  nullptr, nullptr, nullptr, dt, scratch, sb, s));
    TRY(rowscale_add(tmp, b->mask1, b->keep1, b->f1, b->f2, n, c, dt, s));
  } else {
    TRY(ptv3_gemm(b->a, b->w_proj, b->f2, n, c, c, 1, nullptr, nullptr, b->b_proj, nullptr, nullptr, PTV3_ACT_NONE,
                  b->f1, nullptr, nullptr, dt, scratch, sb, s));
  }
  TRY(ptv3_layernorm(b->f2, b->g2, b->b2, nullptr, b->t5, nullptr, nullptr, nullptr, n, c, b->eps, dt, s));
  TRY(ptv3_gemm(b->t5, b->w_fc1, b->h0, n, c, hd, 1, nullptr, nullptr, b->b_fc1, nullptr, nullptr, PTV3_ACT_NONE,
                nullptr, nullptr, nullptr, dt, scratch, sb, s));
  TRY(ptv3_affine_act(b->h0, nullptr, nullptr, PTV3_ACT_GELU, b->h, n, hd, dt, s));
  if (b->mask2) {
    TRY(ptv3_gemm(b->h, b->w_fc2, tmp, n, hd, c, 1, nullptr, nullptr, b->b_fc2, nullptr, nullptr, PTV3_ACT_NONE,
                  nullptr, nullptr, nullptr, dt, scratch, sb, s));
    TRY(rowscale_add(tmp, b->mask2, b->keep2, b->f2, b->out, n, c, dt, s));
  } else {
    TRY(ptv3_gemm(b->h, b->w_fc2, b->out, n, hd, c, 1, nullptr, nullptr, b->b_fc2, nullptr, nullptr, PTV3_ACT_NONE,
                  b->f2, nullptr, nullptr, dt, scratch, sb, s));
  }
  return PTV3_OK;
}

extern "C" int ptv3_block_train_bwd(const ptv3_block_train* b, void* stream) {
  TRY(check_block(b, "block_train_bwd"));
  PTV3_REQUIRE(b->dout && b->dfeat, "block_train_bwd: dout / dfeat are required");
  PTV3_REQUIRE((b->conv_feat != nullptr) == (b->dconv_feat != nullptr),
               "block_train_bwd: dconv_feat goes with conv_feat");
  PTV3_REQUIRE(b->wt_conv && b->wt_lin && b->wt_qkv && b->wt_proj && b->wt_fc1 && b->wt_fc2,
               "block_train_bwd: transposed weights are required");
  if (b->n == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = b->n;
  const int c = b->c, hd = b->hidden, dt = b->dtype;
  const size_t e = esize(dt), nc = (size_t)n * c * e, nh = (size_t)n * hd * e;
  Bump ws{(char*)b->workspace, b->workspace_bytes, 0};
  const size_t sb = op_scratch_bytes(b, true);
  void* scratch = ws.take(sb);
  void* dh = ws.take(nh);
  void* dm = ws.take(nc);
  void* dt5 = ws.take(nc);
  void* df2 = ws.take(nc);
  void* dp = ws.take(nc);
  void* da = ws.take(nc);
  void* dqkv = ws.take(3 * nc);
  void* dt3 = ws.take(nc);
  void* df1 = ws.take(nc);
  void* dc2 = ws.take(nc);
  void* dc1 = ws.take(nc);
  PTV3_REQUIRE(scratch && dc1, "block_train_bwd: workspace too small (%zu bytes)", b->workspace_bytes);
  PTV3_REQUIRE(slab_defer_storage_bytes() <= sizeof(SlabSegsStorage), "block_train_bwd: slab queue storage");
  const bool same = b->conv_feat == nullptr;
  const void* xin = same ? b->feat : b->conv_feat;
#define GEMM(x, w, out, cin, cout) \
  TRY(ptv3_gemm(x, w, out, n, cin, cout, 1, nullptr, nullptr, nullptr, nullptr, nullptr, PTV3_ACT_NONE, nullptr, nullptr, \
                nullptr, dt, scratch, sb, s))
  // the nine reductions over row chunks (six weight gradients, three LayerNorm affine gradients) are collected and run
  // as one launch at the end: each producer gets slab memory of its own
  SlabSegsStorage segs;
  slab_defer_begin(&segs);
  // the six weight gradients wait for the end of the block as well (dmp, dh, dpp, dqkv, dc2, dc1 and their inputs stay
  // where they are until then) and run as one grouped launch: PTV3_TN_GROUP=0 launches them one by one
  static const bool tn_group = [] { const char* e = getenv("PTV3_TN_GROUP"); return !(e && atoi(e) == 0); }();
  TnQueueStorage tnq;
  PTV3_REQUIRE(tn_defer_storage_bytes() <= sizeof(TnQueueStorage), "block_train_bwd: weight-gradient queue storage");
  if (tn_group) tn_defer_begin(&tnq);
  struct EndDefer { ~EndDefer() { slab_defer_begin(nullptr); tn_defer_begin(nullptr); } } end_defer;
  auto own = [&](size_t bytes, size_t* got) { *got = bytes; return ws.take(bytes); };
#define GEMM_TN(dy, x, dw, db, cout, cin)                                                            \
  do {                                                                                               \
    size_t wb__;                                                                                     \
    void* w__ = own(ptv3_gemm_tn_workspace_bytes(n, cout, cin, 1), &wb__);                           \
    PTV3_REQUIRE(w__ != nullptr, "block_train_bwd: workspace too small");                            \
    TRY(ptv3_gemm_tn(dy, x, nullptr, dw, db, n, cout, cin, 1, dt, w__, wb__, s));                    \
  } while (0)
#define LN_BWD(x, dy, add, g, dx, dgb)                                                               \
  do {                                                                                               \
    size_t wb__;                                                                                     \
    void* w__ = own(ptv3_col_reduce_workspace_bytes(n, c), &wb__);                                   \
    PTV3_REQUIRE(w__ != nullptr, "block_train_bwd: workspace too small");                            \
    TRY(ptv3_layernorm_bwd(x, dy, add, g, b->eps, dx, dgb, n, c, dt, w__, wb__, s));                 \
  } while (0)
  // ---- MLP branch
  const void* dmp = b->dout;
  if (b->mask2) { TRY(rowscale_add(b->dout, b->mask2, b->keep2, nullptr, dm, n, c, dt, s)); dmp = dm; }
  GEMM(dmp, b->wt_fc2, dh, c, hd);
  GEMM_TN(dmp, b->h, b->dw_fc2, b->db_fc2, c, hd);
  TRY(ptv3_act_bwd(dh, b->h0, nullptr, nullptr, PTV3_ACT_GELU, dh, n, hd, dt, s));            // dh0 in place
  GEMM(dh, b->wt_fc1, dt5, hd, c);
  GEMM_TN(dh, b->t5, b->dw_fc1, b->db_fc1, hd, c);
  LN_BWD(b->f2, dt5, b->dout, b->g2, df2, b->dln2);
  // ---- attention branch
  const void* dpp = df2;
  if (b->mask1) { TRY(rowscale_add(df2, b->mask1, b->keep1, nullptr, dp, n, c, dt, s)); dpp = dp; }
  GEMM(dpp, b->wt_proj, da, c, c);
  GEMM_TN(dpp, b->a, b->dw_proj, b->db_proj, c, c);
  if (b->attn_lse)
    TRY(ptv3_window_attn_train_bwd(b->qkv, b->a, da, b->attn_lse, b->win_order, b->win_inverse, b->cu_seqlens,
                                   b->num_windows, dqkv, n, b->n_pad, c, b->heads, b->patch, b->scale, dt, scratch, sb, s));
  else if (b->cu_seqlens)
    TRY(ptv3_window_attn_varlen_bwd(b->qkv, b->a, da, b->win_order, b->win_inverse, b->cu_seqlens, b->num_windows, dqkv,
                                    n, b->n_pad, c, b->heads, b->patch, b->scale, dt, scratch, sb, s));
  else
    TRY(ptv3_window_attn_bwd(b->qkv, b->a, da, b->win_order, b->win_inverse, dqkv, n, b->n_pad, c, b->heads, b->patch,
                             b->scale, dt, scratch, sb, s));
  GEMM(dqkv, b->wt_qkv, dt3, 3 * c, c);
  GEMM_TN(dqkv, b->t3, b->dw_qkv, b->db_qkv, 3 * c, c);
  // df1: the block's gradient with respect to feat when the conv reads another tensor, an intermediate otherwise
  void* df1_out = same ? df1 : b->dfeat;
  LN_BWD(b->f1, dt3, df2, b->g1, df1_out, b->dln1);
  // ---- xCPE branch
  LN_BWD(b->c2, df1_out, nullptr, b->g0, dc2, b->dln0);
  GEMM(dc2, b->wt_lin, dc1, c, c);
  GEMM_TN(dc2, b->c1, b->dw_lin, b->db_lin, c, c);
  // conv input gradient = the forward kernel on mirrored, transposed taps; lands on df1 when the conv read feat
  TRY(ptv3_gemm(dc1, b->wt_conv, same ? b->dfeat : b->dconv_feat, n, c, c, b->kvol, b->nbr, b->row_order, nullptr,
                nullptr, nullptr, PTV3_ACT_NONE, same ? df1 : nullptr, nullptr, nullptr, dt, scratch, sb, s));
  {
    size_t wb;
    void* w = own(ptv3_gemm_tn_workspace_bytes(n, c, c, b->kvol), &wb);
    PTV3_REQUIRE(w != nullptr, "block_train_bwd: workspace too small");
    TRY(ptv3_gemm_tn(dc1, xin, b->nbr, b->dw_conv, b->db_conv, n, c, c, b->kvol, dt, w, wb, s));
  }
  TRY(tn_defer_flush(s));
  TRY(slab_defer_flush(s));
#undef GEMM
#undef GEMM_TN
#undef LN_BWD
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
