// Whole-model forward executor: PointTransformerV3.forward (+ optional dense head) as ONE C-ABI call.
// The per-op entry points of include/ptv3_hip.h are launched back to back from native code (no
// interpreter between launches); device memory comes from one caller-provided arena; the only
// host<->device round trips are the pooled scene offsets after each SerializedPooling (4 per forward),
// which the reference also needs (torch.unique) to size the next stage.
// Mirrors point_transformer_v3m1_base.py:699-714 -> Embedding :485-515, Block :318-338,
// SerializedPooling :371-444, SerializedUnpooling :471-482 in eval mode.
#include <vector>
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

struct Arena {
  char* base; size_t cap; size_t off; size_t peak; bool dry; bool failed;
  void* alloc(size_t bytes) {
    size_t a = (off + 255) / 256 * 256;
    off = a + bytes;
    if (off > peak) peak = off;
    if (dry) return (void*)(uintptr_t)256;  // never dereferenced
    if (off > cap) { failed = true; return nullptr; }
    return base + a;
  }
};

__global__ void make_indices_kernel(const int64_t* __restrict__ batch, const void* __restrict__ grid,
                                    int is_i64, int64_t n, int32_t* __restrict__ idx,
                                    int64_t* __restrict__ grid64, int32_t* __restrict__ row_order,
                                    const int64_t* __restrict__ order0) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t x, y, z;
  if (is_i64) {
    const int64_t* g = (const int64_t*)grid;
    x = g[3 * i]; y = g[3 * i + 1]; z = g[3 * i + 2];
  } else {
    const int32_t* g = (const int32_t*)grid;
    x = g[3 * i]; y = g[3 * i + 1]; z = g[3 * i + 2];
  }
  idx[4 * i] = (int32_t)batch[i];
  idx[4 * i + 1] = (int32_t)x; idx[4 * i + 2] = (int32_t)y; idx[4 * i + 3] = (int32_t)z;
  if (grid64) { grid64[3 * i] = x; grid64[3 * i + 1] = y; grid64[3 * i + 2] = z; }
  if (row_order) row_order[i] = (int32_t)order0[i];
}

__global__ void i64_to_i32_kernel(const int64_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (int32_t)src[i];
}

struct Level {
  int64_t n = 0; int depth = 0;
  std::vector<int64_t> off_host;
  const int64_t* offset = nullptr;
  const int64_t* grid = nullptr; const int64_t* batch = nullptr;
  int64_t *code = nullptr, *order = nullptr, *inverse = nullptr;
  int32_t* indices = nullptr; void* table = nullptr; int64_t slots = 0;
  int32_t* nbr3 = nullptr; int32_t* row_order = nullptr;
  int K = -1; int64_t n_pad = 0; int64_t *pad = nullptr, *unpad = nullptr; int32_t* cu = nullptr;
  int32_t* wo[8] = {nullptr}; int32_t* wi[8] = {nullptr};
  void* feat = nullptr; void* conv_feat = nullptr; int channels = 0;
  int64_t* cluster = nullptr;  // pooling_inverse of the PARENT's points into this level
};

struct Run {
  const ptv3_model_desc* d; const void* const* params; int pi = 0; Arena* A; hipStream_t s; int es; bool dry;
  int rc = 0;
  const void* next() { return params ? params[pi++] : (pi++, nullptr); }
  void* alloc(size_t b) { return A->alloc(b); }
  bool ok() const { return rc == 0 && !A->failed; }
#define RUN(call) do { if (!dry && ok()) { int r__ = (call); if (r__) rc = r__; } } while (0)

  void gemm(const void* x, const void* w, void* out, int64_t m, int cin, int cout, int kvol, const int32_t* nbr,
            const int32_t* row_order, const float* bias, const float* bns, const float* bnt, int act,
            const void* res, const int32_t* res_index, void* out2) {
    size_t wsb = ptv3_gemm_workspace_bytes(m, cin, cout, kvol, d->dtype);
    size_t mark = A->off;
    void* ws = wsb ? alloc(wsb) : nullptr;
    RUN(ptv3_gemm(x, w, out, m, cin, cout, kvol, nbr, row_order, bias, bns, bnt, act, res, res_index, out2,
                  d->dtype, ws, wsb, s));
    A->off = mark;  // stream order makes immediate reuse safe
  }

  void prepare_sites(Level& L, const void* grid_in, int is_i64, bool need_grid64) {
    L.indices = (int32_t*)alloc((size_t)L.n * 16);
    L.row_order = (int32_t*)alloc((size_t)L.n * 4);
    int64_t* g64 = nullptr;
    if (need_grid64) { g64 = (int64_t*)alloc((size_t)L.n * 24); L.grid = g64; }
    if (!dry && ok())
      hipLaunchKernelGGL(make_indices_kernel, dim3((unsigned)cdiv(L.n, 256)), dim3(256), 0, s, L.batch, grid_in, is_i64,
                         L.n, L.indices, g64, L.row_order, L.order);
    L.slots = ptv3_subm_table_slots(L.n);
    L.table = alloc((size_t)L.slots * 12);
    RUN(ptv3_subm_build_table(L.indices, L.n, L.table, L.slots, s));
  }

  void attention_plan(Level& L, int patch, int oi) {
    int K = patch;
    if (!d->enable_flash) {
      int64_t mn = L.off_host[0];
      for (size_t i = 1; i < L.off_host.size(); ++i) mn = std::min(mn, L.off_host[i] - L.off_host[i - 1]);
      K = (int)std::min<int64_t>(mn, patch);
    }
    if (K != L.K) {
      int64_t n_pad = 0, prev = 0;
      for (int64_t o : L.off_host) {
        int64_t cnt = o - prev;
        n_pad += cnt > K ? (cnt + K - 1) / K * K : cnt;
        prev = o;
      }
      L.K = K; L.n_pad = n_pad;
      // one launch: pad plan + both window maps of every order (:114-170, :184-185)
      const int k = d->num_orders;
      int32_t* wo = (int32_t*)alloc((size_t)k * n_pad * 4);
      int32_t* wi = (int32_t*)alloc((size_t)k * L.n * 4);
      RUN(ptv3_window_plan(L.order, L.inverse, L.offset, (int)L.off_host.size(), k, L.n, n_pad, K, wo, wi, s));
      for (int i = 0; i < 8; ++i) {
        L.wo[i] = i < k ? wo + (int64_t)i * n_pad : nullptr;
        L.wi[i] = i < k ? wi + (int64_t)i * L.n : nullptr;
      }
    }
    (void)oi;
  }

  // Block.forward (:318-338), eval, pre_norm; writes the new features into `L.feat` (same buffer)
  void block(Level& L, int C, int H, int patch, int oi) {
    const void* conv_w = next(); const float* conv_b = (const float*)next();  // cpe Linear folded in
    const float* ln0_g = (const float*)next(); const float* ln0_b = (const float*)next();
    const float* n1_g = (const float*)next(); const float* n1_b = (const float*)next();
    const void* qkv_w = next(); const float* qkv_b = (const float*)next();
    const void* proj_w = next(); const float* proj_b = (const float*)next();
    const float* n2_g = (const float*)next(); const float* n2_b = (const float*)next();
    const void* fc1_w = next(); const float* fc1_b = (const float*)next();
    const void* fc2_w = next(); const float* fc2_b = (const float*)next();
    const int hidden = (int)(C * d->mlp_ratio);
    if (!L.nbr3) {
      L.nbr3 = (int32_t*)alloc((size_t)L.n * 27 * 4);
      RUN(ptv3_subm_neighbors(L.indices, L.n, L.table, L.slots, 3, L.nbr3, s));
    }
    attention_plan(L, patch, oi);
    const size_t mark = A->off;
    const size_t row = (size_t)L.n * es;
    void* t2 = alloc(row * C); void* f1 = alloc(row * C); void* t3 = alloc(row * C);
    void* qkv = alloc(row * 3 * C); void* t4 = alloc(row * C); void* f2 = alloc(row * C); void* t5 = alloc(row * C);
    void* t6 = alloc(row * hidden);
    gemm(L.conv_feat, conv_w, t2, L.n, C, C, 27, L.nbr3, L.row_order, conv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    RUN(ptv3_layernorm(t2, ln0_g, ln0_b, L.feat, f1, n1_g, n1_b, t3, L.n, C, d->ln_eps, d->dtype, s));
    gemm(t3, qkv_w, qkv, L.n, C, 3 * C, 1, nullptr, nullptr, qkv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
    const float scale = d->qk_scale > 0.f ? d->qk_scale : 1.0f / sqrtf((float)(C / H));
    RUN(ptv3_window_attn_fwd(qkv, L.wo[oi], L.wi[oi], t4, L.n, L.n_pad, C, H, L.K, scale, nullptr, d->dtype, s));
    gemm(t4, proj_w, f2, L.n, C, C, 1, nullptr, nullptr, proj_b, nullptr, nullptr, 0, f1, nullptr, nullptr);
    RUN(ptv3_layernorm(f2, n2_g, n2_b, nullptr, t5, nullptr, nullptr, nullptr, L.n, C, d->ln_eps, d->dtype, s));
    gemm(t5, fc1_w, t6, L.n, C, hidden, 1, nullptr, nullptr, fc1_b, nullptr, nullptr, PTV3_ACT_GELU, nullptr, nullptr, nullptr);
    gemm(t6, fc2_w, L.feat, L.n, hidden, C, 1, nullptr, nullptr, fc2_b, nullptr, nullptr, 0, f2, nullptr, nullptr);
    L.conv_feat = L.feat;
    A->off = mark;
  }
};

static int run_forward(const ptv3_model_desc* d, const void* const* params, const ptv3_forward_io* io, Arena& A,
                       hipStream_t s, bool dry, size_t* peak, int* param_count = nullptr) {
  Run R; R.d = d; R.params = params; R.A = &A; R.s = s; R.dry = dry; R.es = d->dtype == PTV3_F32 ? 4 : 2;
  const int S = d->num_stages, k = d->num_orders, es = R.es;
  std::vector<Level> lv(S);
  // ---- level 0: serialization (structure.py:52-109)
  Level& L0 = lv[0];
  L0.n = io->n; L0.depth = io->depth; L0.batch = io->batch; L0.offset = io->offset;
  L0.off_host.assign(io->offset_host, io->offset_host + io->b);
  L0.code = io->code; L0.order = io->order; L0.inverse = io->inverse;
  const int nbits = io->b > 1 ? 32 - __builtin_clz((unsigned)(io->b - 1)) : 0;
  {
    if (!dry) { int r = ptv3_sfc_encode(io->grid_coord, io->coord_is_i64, io->batch, io->n, io->depth,
                                        io->order_ids_host, k, L0.code, s); if (r) return r; }
    size_t wsb = ptv3_argsort_workspace_bytes(k, io->n);
    size_t mark = A.off;
    void* ws = A.alloc(wsb);
    if (!dry && !A.failed) { int r = ptv3_argsort_i64(L0.code, k, io->n, std::max(1, 3 * io->depth + nbits), L0.order,
                                                      L0.inverse, ws, wsb, s); if (r) return r; }
    A.off = mark;
  }
  R.prepare_sites(L0, io->grid_coord, io->coord_is_i64, true);
  // ---- embedding (:485-515): SubMConv3d k=5 -> folded BN -> GELU
  {
    const void* w = R.next(); const float* bns = (const float*)R.next(); const float* bnt = (const float*)R.next();
    const int C0 = d->enc_channels[0];
    L0.feat = R.alloc((size_t)L0.n * C0 * es); L0.channels = C0;
    const size_t mark = A.off;
    int32_t* nbr5 = (int32_t*)R.alloc((size_t)L0.n * 125 * 4);
    if (!dry && R.ok()) { int r = ptv3_subm_neighbors(L0.indices, L0.n, L0.table, L0.slots, 5, nbr5, s); if (r) R.rc = r; }
    R.gemm(io->feat, w, L0.feat, L0.n, d->in_channels, C0, 125, nbr5, L0.row_order, nullptr, bns, bnt, PTV3_ACT_GELU,
           nullptr, nullptr, nullptr);
    A.off = mark;
    L0.conv_feat = L0.feat;
  }
  // ---- encoder
  for (int st = 0; st < S; ++st) {
    Level& L = lv[st];
    if (st > 0) {
      Level& P = lv[st - 1];
      const void* w = R.next(); const float* b = (const float*)R.next();
      const float* bns = (const float*)R.next(); const float* bnt = (const float*)R.next();
      const int C = d->enc_channels[st], Cp = d->enc_channels[st - 1];
      int pd = 0; { int v = d->stride[st - 1] - 1; while (v > 0) { ++pd; v >>= 1; } }
      if (pd > P.depth) pd = 0;
      P.cluster = (int64_t*)R.alloc((size_t)P.n * 8);
      int64_t* poff = (int64_t*)R.alloc((size_t)io->b * 8);
      void* proj = R.alloc((size_t)P.n * C * es);
      R.gemm(P.feat, w, proj, P.n, Cp, C, 1, nullptr, nullptr, b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
      int32_t* seg = (int32_t*)R.alloc((size_t)(P.n + 1) * 4);
      int32_t* nout_dev = (int32_t*)R.alloc(256);
      size_t pwsb = ptv3_pool_workspace_bytes(P.n);
      void* pws = R.alloc(pwsb);
      if (!dry && R.ok()) {
        int r = ptv3_pool_segments(P.code, P.order, P.n, 3 * pd, P.batch, P.cluster, seg, nout_dev, poff, pws, pwsb, s);
        if (r) R.rc = r;
      }
      L.off_host.resize(io->b);
      if (dry) {
        L.off_host = P.off_host;  // worst case: nothing merges
      } else if (R.ok()) {
        // the one host round trip of the stage (the reference's torch.unique has the same)
        if (hipMemcpyAsync(L.off_host.data(), poff, (size_t)io->b * 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { set_error("forward: offset read-back failed"); return PTV3_ERR_LAUNCH; }
      }
      if (!R.ok()) break;
      L.n = L.off_host[io->b - 1]; L.depth = P.depth - pd; L.offset = poff; L.channels = C;
      if (io->stage_points_host) io->stage_points_host[st] = L.n;
      const int* perm = io->pool_perm_host + (size_t)(st - 1) * k;  // order shuffle of :408-412
      // pooled level tensors (sized by the n_out just read back); the stage's temporaries simply stay
      // allocated below them for the rest of the forward (a few MB)
      L.feat = R.alloc((size_t)L.n * C * es);
      int64_t* g = (int64_t*)R.alloc((size_t)L.n * 24);
      int64_t* bt = (int64_t*)R.alloc((size_t)L.n * 8);
      L.code = (int64_t*)R.alloc((size_t)k * L.n * 8);
      L.order = (int64_t*)R.alloc((size_t)k * L.n * 8);
      L.inverse = (int64_t*)R.alloc((size_t)k * L.n * 8);
      if (!dry && R.ok()) {
        int r = ptv3_pool_reduce(proj, nullptr, P.grid, P.batch, P.code, k, P.order, seg, P.n, L.n, C, pd, bns, bnt,
                                 PTV3_ACT_GELU, perm, L.feat, nullptr, g, bt, L.code, d->dtype, s);
        if (r) R.rc = r;
      }
      L.grid = g; L.batch = bt; L.conv_feat = L.feat;
      {
        size_t wsb = ptv3_argsort_workspace_bytes(k, L.n);
        size_t m2 = A.off;
        void* ws = R.alloc(wsb);
        if (!dry && R.ok()) {
          int r = ptv3_argsort_i64(L.code, k, L.n, std::max(1, 3 * L.depth + nbits), L.order, L.inverse, ws, wsb, s);
          if (r) R.rc = r;
        }
        A.off = m2;
      }
      R.prepare_sites(L, L.grid, 1, false);
    } else if (io->stage_points_host) {
      io->stage_points_host[0] = L.n;
    }
    for (int i = 0; i < d->enc_depths[st]; ++i)
      R.block(L, d->enc_channels[st], d->enc_heads[st], d->enc_patch[st], i % k);
    if (!R.ok()) break;
  }
  // ---- decoder (:651-697)
  if (!d->enc_mode && R.ok()) {
    for (int st = S - 2; st >= 0; --st) {
      Level& P = lv[st];
      Level& Ch = lv[st + 1];
      const int Cd = d->dec_channels[st];
      const int Cin = Ch.channels, Cskip = d->enc_channels[st];
      const void* w = R.next(); const float* b = (const float*)R.next();
      const float* bns = (const float*)R.next(); const float* bnt = (const float*)R.next();
      const void* ws_ = R.next(); const float* bs = (const float*)R.next();
      const float* bnss = (const float*)R.next(); const float* bnst = (const float*)R.next();
      void* skip = R.alloc((size_t)P.n * Cd * es);
      void* fused = (st == 0 && io->out_feat) ? io->out_feat : R.alloc((size_t)P.n * Cd * es);
      int32_t* c32 = (int32_t*)R.alloc((size_t)P.n * 4);
      const size_t mark = A.off;
      void* up = R.alloc((size_t)Ch.n * Cd * es);
      if (!dry && R.ok())
        hipLaunchKernelGGL(i64_to_i32_kernel, dim3((unsigned)cdiv(P.n, 256)), dim3(256), 0, s, P.cluster, c32, P.n);
      R.gemm(Ch.feat, w, up, Ch.n, Cin, Cd, 1, nullptr, nullptr, b, bns, bnt, PTV3_ACT_GELU, nullptr, nullptr, nullptr);
      // skip branch; its epilogue gathers the up-branch rows by cluster id.  `skip` alone stays the sparse
      // tensor's features for the next block's conv (the reference never refreshes it after the add, :478)
      R.gemm(P.feat, ws_, skip, P.n, Cskip, Cd, 1, nullptr, nullptr, bs, bnss, bnst, PTV3_ACT_GELU, up, c32, fused);
      A.off = mark;
      P.feat = fused; P.conv_feat = skip; P.channels = Cd;
      for (int i = 0; i < d->dec_depths[st]; ++i)
        R.block(P, Cd, d->dec_heads[st], d->dec_patch[st], i % k);
      if (!R.ok()) break;
    }
  }
  // ---- dense head: Linear -> folded BN -> ReLU -> Linear (offset_keypoint_ptv3.py:26-31)
  if (R.ok() && d->head_out > 0 && !d->enc_mode) {
    Level& L = lv[0];
    const void* w0 = R.next(); const float* b0 = (const float*)R.next();
    const float* bns = (const float*)R.next(); const float* bnt = (const float*)R.next();
    const void* w1 = R.next(); const float* b1 = (const float*)R.next();
    const size_t mark = A.off;
    void* hid = R.alloc((size_t)L.n * d->head_hidden * es);
    R.gemm(L.feat, w0, hid, L.n, L.channels, d->head_hidden, 1, nullptr, nullptr, b0, bns, bnt, PTV3_ACT_RELU, nullptr,
           nullptr, nullptr);
    if (d->dtype == PTV3_F32) {
      R.gemm(hid, w1, io->out_head, L.n, d->head_hidden, d->head_out, 1, nullptr, nullptr, b1, nullptr, nullptr, 0,
             nullptr, nullptr, nullptr);
    } else {
      void* o = R.alloc((size_t)L.n * d->head_out * es);
      R.gemm(hid, w1, o, L.n, d->head_hidden, d->head_out, 1, nullptr, nullptr, b1, nullptr, nullptr, 0, nullptr,
             nullptr, nullptr);
      if (!dry && R.ok()) { int r = ptv3_cast(o, PTV3_BF16, io->out_head, PTV3_F32, L.n * d->head_out, s); if (r) R.rc = r; }
    }
    A.off = mark;
  }
  if (!dry && R.ok() && !d->enc_mode && io->out_feat && lv[0].feat != io->out_feat)
    (void)hipMemcpyAsync(io->out_feat, lv[0].feat, (size_t)lv[0].n * lv[0].channels * es, hipMemcpyDeviceToDevice, s);
  if (peak) *peak = A.peak;
  if (param_count) *param_count = R.pi;
  if (A.failed) { set_error("forward: workspace arena too small (need > %zu bytes)", A.cap); return PTV3_ERR_ARG; }
  if (R.rc) return R.rc;
  if (!dry) PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

static int check_desc(const ptv3_model_desc* d) {
  PTV3_REQUIRE(d->num_stages >= 1 && d->num_stages <= 8, "forward: num_stages %d", d->num_stages);
  PTV3_REQUIRE(d->num_orders >= 1 && d->num_orders <= 8, "forward: num_orders %d", d->num_orders);
  PTV3_REQUIRE(d->dtype == PTV3_F32 || d->dtype == PTV3_BF16, "forward: dtype");
  PTV3_REQUIRE(d->mlp_ratio > 0.f, "forward: mlp_ratio");
  return PTV3_OK;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" size_t ptv3_forward_workspace_bytes(const ptv3_model_desc* desc, int64_t n, int b) {
  if (check_desc(desc)) return 0;
  std::vector<int64_t> off(b);
  for (int i = 0; i < b; ++i) off[i] = n * (i + 1) / b;  // shape only matters through n and b
  std::vector<int> ids(desc->num_orders, 0), perm((size_t)desc->num_stages * desc->num_orders, 0);
  for (int st = 0; st < desc->num_stages; ++st)
    for (int r = 0; r < desc->num_orders; ++r) perm[(size_t)st * desc->num_orders + r] = (r + 1) % desc->num_orders;
  ptv3_forward_io io{};
  io.n = n; io.b = b; io.depth = 16; io.offset_host = off.data(); io.order_ids_host = ids.data();
  io.pool_perm_host = perm.data(); io.coord_is_i64 = 1;
  Arena A{nullptr, 0, 0, 0, true, false};
  size_t peak = 0;
  run_forward(desc, nullptr, &io, A, nullptr, true, &peak);
  // pad plan slack: every scene may gain up to one window of borrowed rows per attention plan
  return peak + (size_t)b * 16384 * 64 + (1 << 20);
}

extern "C" int ptv3_forward(const ptv3_model_desc* desc, const void* const* params, int num_params,
                            const ptv3_forward_io* io, void* workspace, size_t workspace_bytes, void* stream) {
  if (int r = check_desc(desc)) return r;
  PTV3_REQUIRE(io->n >= 1 && io->b >= 1, "forward: empty batch");
  PTV3_REQUIRE(io->depth >= 1 && io->depth <= 16, "forward: depth %d outside [1,16]", io->depth);
  PTV3_REQUIRE(io->code && io->order && io->inverse, "forward: level-0 serialization outputs are required");
  for (int i = 0; i < io->b; ++i)
    PTV3_REQUIRE(io->offset_host[i] > (i ? io->offset_host[i - 1] : 0), "forward: empty scene %d", i);
  PTV3_REQUIRE(io->offset_host[io->b - 1] == io->n, "forward: offset does not end at n");
  {  // the flat parameter table must match the walk of run_forward exactly
    Arena D{nullptr, 0, 0, 0, true, false};
    int count = 0;
    run_forward(desc, nullptr, io, D, nullptr, true, nullptr, &count);
    PTV3_REQUIRE(count == num_params, "forward: %d parameter pointers given, the model description needs %d",
                 num_params, count);
  }
  Arena A{(char*)workspace, workspace_bytes, 0, 0, false, false};
  return run_forward(desc, params, io, A, (hipStream_t)stream, false, nullptr);
}
