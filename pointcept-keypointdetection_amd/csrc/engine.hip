// Whole-model forward executor: PointTransformerV3.forward (+ optional dense head) as ONE C-ABI call.
//
// Two HIP streams.  Everything that depends only on the coordinates (serialization, radix sorts, pooling
// clusters for every level, site hashes, neighbour tables, window plans) is the GEOMETRY pipeline and runs
// on an internal stream; everything that touches features runs on the caller's stream.  The only host
// round trips (pooled scene offsets after each SerializedPooling: the reference's torch.unique needs the
// same) wait on the geometry stream alone, so they never drain the feature stream: while the level-0
// blocks execute, the host has already sized and queued the deeper levels.  Cross-stream order is by
// events (one per level).  Device memory comes from one caller-provided arena, split into a geometry
// part (never reused inside a call: the two streams run concurrently) and a feature part (stack).
//
// Mirrors point_transformer_v3m1_base.py:699-714 -> structure.py:52-146, Embedding :485-515,
// Block :318-338, SerializedPooling :371-444, SerializedUnpooling :471-482 in eval mode.
#include <algorithm>
#include <chrono>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
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

// max over all 3n grid coordinates -> out[0] (for depth = bit_length(max + 1), structure.py:73)
__global__ void coord_max_kernel(const void* __restrict__ grid, int is_i64, int64_t n3, int* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int v = 0;
  for (; i < n3; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t c = is_i64 ? ((const int64_t*)grid)[i] : (int64_t)((const int32_t*)grid)[i];
    v = max(v, (int)c);
  }
  for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_down(v, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, v);
}

// batch[i] = scene of point i, from the cumulative offsets (offset2batch, models/utils/misc.py:25-30)
__global__ void batch_from_offset_kernel(const int64_t* __restrict__ offset, int b, int64_t n,
                                         int64_t* __restrict__ batch) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = b;  // first scene whose end is > i
  while (lo < hi) { int mid = (lo + hi) >> 1; if (offset[mid] > i) hi = mid; else lo = mid + 1; }
  batch[i] = lo;
}

// window plan of one (level, patch): cu != nullptr when some scene is shorter than the patch (enable_flash=True
// keeps K fixed, so such a scene is ONE short window, v3m1_base.py:131-133) -> ragged-window attention
struct Plan {
  int K = -1; int64_t n_pad = 0; int32_t* wo[8] = {nullptr}; int32_t* wi[8] = {nullptr};
  int32_t* cu = nullptr; int nwin = 0; double sum_len_sq = 0.0;
};

struct Level {
  int64_t n = 0; int depth = 0;
  std::vector<int64_t> off_host;
  const int64_t* offset = nullptr;
  const int64_t* grid = nullptr; const int64_t* batch = nullptr;
  int64_t *code = nullptr, *order = nullptr, *inverse = nullptr;
  int32_t* indices = nullptr; void* table = nullptr; int64_t slots = 0;
  int32_t* nbr3 = nullptr; int32_t* nbr5 = nullptr; int32_t* row_order = nullptr;
  Plan plan[2];                 // [0] encoder patch, [1] decoder patch (shared when K is equal)
  int32_t* seg = nullptr;       // runs of THIS level's points in order 0 that form the next level's rows
  int64_t* cluster = nullptr;   // pooling_inverse: this level's point -> next level's row
  int32_t* cluster32 = nullptr;
  hipEvent_t ready = nullptr;   // geometry of this level complete (recorded on the geometry stream)
  void* feat = nullptr; void* conv_feat = nullptr; int channels = 0;
};

static inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Executor state: the streams, events, call parity and overlap handshake of ONE executor.  An executor belongs to
// one device; a model that wants its forwards ordered independently of every other model in the process creates its
// own (ptv3_executor_create, ptv3_forward_io.executor), otherwise the device's default executor is used.  Calls on
// one executor take turns on the host (mutex) and overlap on the GPU as consecutive calls of one thread do.
// The executor streams of a device (one geometry stream, two feature streams) exist ONCE per device and are shared by
// every executor on it.  HIP spreads streams over a handful of hardware queues in creation order: a second set of
// streams (a second model, or a model built after another one was deleted) shares queues among itself, and its geometry
// and feature pipelines serialise - the 120k-point LiDAR forward ran 18.4 ms instead of 11.9 ms when its model was
// built after the headline model, and whichever of two models was created second was the slow one.  Sharing is safe:
// a stream is an ordering domain, two executors on one stream merely add order between their calls (each still waits
// on its own events, arenas stay per executor).
struct DeviceStreams { hipStream_t geo = nullptr; hipStream_t feat[2] = {nullptr, nullptr}; };
static std::mutex g_pool_mutex;
static std::vector<DeviceStreams> g_device_streams;   // by device; never destroyed
static hipStream_t device_stream(int device, int which /* 0 geo, 1 feat[0], 2 feat[1] */) {
  std::lock_guard<std::mutex> guard(g_pool_mutex);
  if ((int)g_device_streams.size() <= device) g_device_streams.resize(device + 1);
  DeviceStreams& d = g_device_streams[device];
  hipStream_t& s = which == 0 ? d.geo : d.feat[which - 1];
  if (!s) {
    if (which == 0) {
      // highest priority: the geometry chain is a string of tiny dependent kernels whose read-backs gate the host
      int least = 0, greatest = 0;
      (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
      if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) != hipSuccess) s = nullptr;
    } else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      s = nullptr;
    }
  }
  return s;
}

struct Exec {
  int device = 0;
  hipStream_t geo = nullptr;
  hipStream_t feat[2] = {nullptr, nullptr};  // overlap_calls: feature pipelines of consecutive calls
  std::vector<hipEvent_t> events;
  unsigned call = 0;          // calls alternate between the two geometry arenas (see ptv3_forward)
  bool last_overlap = false;  // mode of the previous call (the overlap handshake reruns after a stream-ordered call)
  double sync_us = 0.0;       // host time blocked in geometry-stream read-backs (PTV3_ENGINE_TIMING=1)
  std::mutex mu;
  hipError_t timed_sync(hipStream_t s) {
    double t = now_us();
    hipError_t e = hipStreamSynchronize(s);
    sync_us += now_us() - t;
    return e;
  }
  hipEvent_t event_at(size_t i) {
    while (events.size() <= i) {
      hipEvent_t e;
      (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
      events.push_back(e);
    }
    return events[i];
  }
  void release() {
    // the streams belong to the device (DeviceStreams): drain what this executor queued, leave them in place
    if (geo) { (void)hipStreamSynchronize(geo); geo = nullptr; }
    for (auto& f : feat) if (f) { (void)hipStreamSynchronize(f); f = nullptr; }
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    events.clear();
  }
};
static std::mutex g_exec_mutex;
static std::vector<Exec*> g_default_exec;  // one per device, created on first use
static Exec* default_exec() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> guard(g_exec_mutex);
  if ((int)g_default_exec.size() <= dev) g_default_exec.resize(dev + 1, nullptr);
  if (!g_default_exec[dev]) { g_default_exec[dev] = new Exec(); g_default_exec[dev]->device = dev; }
  return g_default_exec[dev];
}

// (n, cin) rows in their original dtype -> (n, cpad) rows of the compute dtype, zero-padded
template <typename S, typename D>
__global__ void pad_cast_kernel(const S* __restrict__ x, int cin, D* __restrict__ y, int cpad, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * cpad) return;
  const int c = (int)(i % cpad);
  y[i] = from_f32<D>(c < cin ? to_f32<S>(x[(i / cpad) * cin + c]) : 0.f);
}
static void pad_cast(const void* x, int sdt, int cin, void* y, int ddt, int cpad, int64_t n, hipStream_t s) {
  dim3 grid((unsigned)cdiv(n * cpad, 256)), block(256);
  if (sdt == PTV3_F32 && ddt == PTV3_F32)
    hipLaunchKernelGGL((pad_cast_kernel<float, float>), grid, block, 0, s, (const float*)x, cin, (float*)y, cpad, n);
  else if (sdt == PTV3_F32)
    hipLaunchKernelGGL((pad_cast_kernel<float, __bf16>), grid, block, 0, s, (const float*)x, cin, (__bf16*)y, cpad, n);
  else if (ddt == PTV3_F32)
    hipLaunchKernelGGL((pad_cast_kernel<__bf16, float>), grid, block, 0, s, (const __bf16*)x, cin, (float*)y, cpad, n);
  else
    hipLaunchKernelGGL((pad_cast_kernel<__bf16, __bf16>), grid, block, 0, s, (const __bf16*)x, cin, (__bf16*)y, cpad, n);
}

struct Run {
  Exec* X;
  const ptv3_model_desc* d; const void* const* params; int pi = 0;
  Arena* G; Arena* F;            // geometry / feature arenas
  hipStream_t sg, sf;            // geometry / feature streams
  int es; bool dry; int rc = 0;
  const void* next() { return params ? params[pi++] : (pi++, nullptr); }
  bool ok() const { return rc == 0 && !G->failed && !F->failed; }
#define RUN(call) do { if (!dry && ok()) { int r__ = (call); if (r__) rc = r__; } } while (0)

  void gemm(const void* x, const void* w, void* out, int64_t m, int cin, int cout, int kvol, const int32_t* nbr,
            const int32_t* row_order, const float* bias, const float* bns, const float* bnt, int act,
            const void* res, const int32_t* res_index, void* out2) {
    size_t wsb = ptv3_gemm_workspace_bytes(m, cin, cout, kvol, d->dtype);
    size_t mark = F->off;
    void* ws = wsb ? F->alloc(wsb) : nullptr;
    RUN(ptv3_gemm(x, w, out, m, cin, cout, kvol, nbr, row_order, bias, bns, bnt, act, res, res_index, out2,
                  d->dtype, ws, wsb, sf));
    F->off = mark;  // stream order makes immediate reuse safe
  }

  // whole-row linears (ptv3_rows_linear) instead of LayerNorm + tiled GEMM launches: measured ahead at 245 .. 1388 rows
  // (column groups) and at 27 743+ rows alike (tools/bench_block_wide.py); PTV3_ROWS_MIN raises the row count it starts at
  bool rows_path(int64_t n, int C, int hidden) const {
    static const int64_t min_rows = [] { const char* e = getenv("PTV3_ROWS_MIN"); return e ? atoll(e) : (int64_t)0; }();
    return n >= min_rows && ptv3_rows_linear_capable(C, 3 * C, d->dtype, n) && ptv3_rows_linear_capable(C, C, d->dtype, n) &&
           ptv3_rows_linear_capable(C, hidden, d->dtype, n);
  }

  int patch_K(const Level& L, int patch) const {
    if (d->enable_flash) return patch;
    int64_t mn = L.off_host[0];
    for (size_t i = 1; i < L.off_host.size(); ++i) mn = std::min(mn, L.off_host[i] - L.off_host[i - 1]);
    return (int)std::min<int64_t>(mn, patch);
  }

  void window_plan(Level& L, Plan& P, int K) {
    int64_t n_pad = 0, prev = 0;
    for (int64_t o : L.off_host) {
      int64_t cnt = o - prev;
      n_pad += cnt > K ? (cnt + K - 1) / K * K : cnt;
      prev = o;
    }
    P.K = K; P.n_pad = n_pad;
    bool ragged = false;
    P.nwin = 0; P.sum_len_sq = 0.0; prev = 0;
    for (int64_t o : L.off_host) {
      const int64_t cnt = o - prev;
      prev = o;
      P.nwin += (int)((cnt + K - 1) / K);
      if (cnt < K) { ragged = true; P.sum_len_sq += (double)cnt * cnt; }
      else P.sum_len_sq += (double)((cnt + K - 1) / K) * K * K;
    }
    const int k = d->num_orders;
    int32_t* wo = (int32_t*)G->alloc((size_t)k * n_pad * 4);
    int32_t* wi = (int32_t*)G->alloc((size_t)k * L.n * 4);
    // dry run: reserve the table whenever the flag allows short windows (the sizes are not known yet)
    P.cu = (ragged || (dry && d->enable_flash)) ? (int32_t*)G->alloc(((size_t)P.nwin + L.off_host.size() + 1) * 4) : nullptr;
    if (!ragged && !dry) P.cu = nullptr;
    RUN(ptv3_window_plan(L.order, L.inverse, L.offset, (int)L.off_host.size(), k, L.n, n_pad, K, wo, wi, P.cu, sg));
    for (int i = 0; i < k; ++i) { P.wo[i] = wo + (int64_t)i * n_pad; P.wi[i] = wi + (int64_t)i * L.n; }
  }

  // sites, hash, neighbour tables and window plans of one level (geometry stream)
  hipEvent_t stem_ready = nullptr;
  void level_geometry(Level& L, int st, const void* grid_in, int is_i64, bool need_grid64) {
    L.indices = (int32_t*)G->alloc((size_t)L.n * 16);
    L.row_order = (int32_t*)G->alloc((size_t)L.n * 4);
    int64_t* g64 = nullptr;
    if (need_grid64) { g64 = (int64_t*)G->alloc((size_t)L.n * 24); L.grid = g64; }
    if (!dry && ok())
      hipLaunchKernelGGL(make_indices_kernel, dim3((unsigned)cdiv(L.n, 256)), dim3(256), 0, sg, L.batch, grid_in, is_i64,
                         L.n, L.indices, g64, L.row_order, L.order);
    L.slots = ptv3_subm_table_slots(L.n);
    L.table = G->alloc((size_t)L.slots * 12);
    RUN(ptv3_subm_build_table(L.indices, L.n, L.table, L.slots, sg));
    if (st == 0) {
      L.nbr5 = (int32_t*)G->alloc((size_t)L.n * 125 * 4);
      RUN(ptv3_subm_neighbors(L.indices, L.n, L.table, L.slots, 5, L.nbr5, sg));
      // the stem conv needs the 5^3 table and the row order only: it starts here, under the 3^3 table and the window plans
      if (!dry && ok() && stem_ready) (void)hipEventRecord(stem_ready, sg);
    }
    L.nbr3 = (int32_t*)G->alloc((size_t)L.n * 27 * 4);
    RUN(ptv3_subm_neighbors(L.indices, L.n, L.table, L.slots, 3, L.nbr3, sg));
    const int Ke = patch_K(L, d->enc_patch[st]);
    window_plan(L, L.plan[0], Ke);
    if (!d->enc_mode && st < d->num_stages - 1) {
      const int Kd = patch_K(L, d->dec_patch[st]);
      if (Kd == Ke) L.plan[1] = L.plan[0]; else window_plan(L, L.plan[1], Kd);
    }
  }

  // SerializedAttention core (:184-216): uniform K-slot windows, or ragged ones (a scene shorter than the fixed patch
  // of enable_flash=True is one short window, :131-133 + the varlen call :207-215)
  void attention(const Level& L, const Plan& P, const void* qkv, void* out, int C, int H, int oi, float scale) {
    if (P.cu)
      RUN(ptv3_window_attn_varlen_fwd(qkv, P.wo[oi], P.wi[oi], P.cu, P.nwin, out, L.n, P.n_pad, C, H, P.K, scale,
                                      P.sum_len_sq, d->dtype, sf));
    else
      RUN(ptv3_window_attn_fwd(qkv, P.wo[oi], P.wi[oi], out, L.n, P.n_pad, C, H, P.K, scale, nullptr, d->dtype, sf));
  }

  // Block.forward (:318-338), eval, pre_norm; cpe Linear folded into the conv taps
  void block(Level& L, const Plan& P, int C, int H, int oi) {
    const void* conv_w = next(); const float* conv_b = (const float*)next();
    const float* ln0_g = (const float*)next(); const float* ln0_b = (const float*)next();
    const float* n1_g = (const float*)next(); const float* n1_b = (const float*)next();
    const void* qkv_w = next(); const float* qkv_b = (const float*)next();
    const void* proj_w = next(); const float* proj_b = (const float*)next();
    const float* n2_g = (const float*)next(); const float* n2_b = (const float*)next();
    const void* fc1_w = next(); const float* fc1_b = (const float*)next();
    const void* fc2_w = next(); const float* fc2_b = (const float*)next();
    const int hidden = (int)(C * d->mlp_ratio);
    const size_t mark = F->off;
    const size_t row = (size_t)L.n * es;
    const float scale = d->qk_scale > 0.f ? d->qk_scale : 1.0f / sqrtf((float)(C / H));
    if (ptv3_block_fusable(C, hidden, d->dtype, L.n)) {
      // conv -> [LN + shortcut + LN + qkv] -> window attention -> [proj + shortcut + LN + fc1 + GELU + fc2 + shortcut]
      void* f1 = F->alloc(row * C); void* qkv = F->alloc(row * 3 * C); void* t4 = F->alloc(row * C);
      const int splits = ptv3_gemm_splits(L.n, C, C, 27, d->dtype);
      if (splits > 1) {
        const size_t wsb = ptv3_gemm_workspace_bytes(L.n, C, C, 27, d->dtype);
        float* slab = (float*)F->alloc(wsb);
        RUN(ptv3_gemm(L.conv_feat, conv_w, nullptr, L.n, C, C, 27, L.nbr3, L.row_order, nullptr, nullptr, nullptr, 0,
                      nullptr, nullptr, nullptr, d->dtype, slab, wsb, sf));
        RUN(ptv3_block_head(nullptr, slab, splits, conv_b, L.feat, ln0_g, ln0_b, n1_g, n1_b, qkv_w, qkv_b, f1, qkv, L.n,
                            C, d->ln_eps, d->dtype, sf));
      } else {
        void* t2 = F->alloc(row * C);
        gemm(L.conv_feat, conv_w, t2, L.n, C, C, 27, L.nbr3, L.row_order, conv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
        RUN(ptv3_block_head(t2, nullptr, 0, nullptr, L.feat, ln0_g, ln0_b, n1_g, n1_b, qkv_w, qkv_b, f1, qkv, L.n, C,
                            d->ln_eps, d->dtype, sf));
      }
      attention(L, P, qkv, t4, C, H, oi, scale);
      RUN(ptv3_block_tail(t4, f1, proj_w, proj_b, n2_g, n2_b, fc1_w, fc1_b, fc2_w, fc2_b, L.feat, L.n, C, hidden,
                          d->ln_eps, d->dtype, sf));
    } else {
      void* t2 = F->alloc(row * C); void* f1 = F->alloc(row * C); void* t3 = F->alloc(row * C);
      void* qkv = F->alloc(row * 3 * C); void* t4 = F->alloc(row * C); void* f2 = F->alloc(row * C);
      void* t5 = F->alloc(row * C); void* t6 = F->alloc(row * hidden);
      const int splits = ptv3_gemm_splits(L.n, C, C, 27, d->dtype);
      // whole-row linears with their LayerNorms folded in (ptv3_rows_linear): every row read once, no LayerNorm launch
      const bool rows = rows_path(L.n, C, hidden);
      if (splits > 1) {
        // the conv leaves its split-K slabs; the LayerNorm kernel sums them (no separate reduce launch)
        const size_t wsb = ptv3_gemm_workspace_bytes(L.n, C, C, 27, d->dtype);
        float* slab = (float*)F->alloc(wsb);
        RUN(ptv3_gemm(L.conv_feat, conv_w, nullptr, L.n, C, C, 27, L.nbr3, L.row_order, nullptr, nullptr, nullptr, 0,
                      nullptr, nullptr, nullptr, d->dtype, slab, wsb, sf));
        RUN(ptv3_layernorm_slabs(slab, splits, conv_b, ln0_g, ln0_b, L.feat, f1, n1_g, n1_b, t3, L.n, C, d->ln_eps,
                                 d->dtype, sf));
        gemm(t3, qkv_w, qkv, L.n, C, 3 * C, 1, nullptr, nullptr, qkv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
      } else if (rows) {
        gemm(L.conv_feat, conv_w, t2, L.n, C, C, 27, L.nbr3, L.row_order, conv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
        RUN(ptv3_rows_linear(t2, L.feat, ln0_g, ln0_b, n1_g, n1_b, qkv_w, qkv_b, 0, nullptr, f1, qkv, L.n, C, 3 * C,
                             d->ln_eps, d->dtype, sf));
      } else {
        gemm(L.conv_feat, conv_w, t2, L.n, C, C, 27, L.nbr3, L.row_order, conv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
        RUN(ptv3_layernorm(t2, ln0_g, ln0_b, L.feat, f1, n1_g, n1_b, t3, L.n, C, d->ln_eps, d->dtype, sf));
        gemm(t3, qkv_w, qkv, L.n, C, 3 * C, 1, nullptr, nullptr, qkv_b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
      }
      attention(L, P, qkv, t4, C, H, oi, scale);
      if (rows) {
        RUN(ptv3_rows_linear(t4, nullptr, nullptr, nullptr, nullptr, nullptr, proj_w, proj_b, 0, f1, nullptr, f2, L.n, C, C,
                             d->ln_eps, d->dtype, sf));
        RUN(ptv3_rows_linear(f2, nullptr, nullptr, nullptr, n2_g, n2_b, fc1_w, fc1_b, PTV3_ACT_GELU, nullptr, nullptr, t6,
                             L.n, C, hidden, d->ln_eps, d->dtype, sf));
      } else {
        gemm(t4, proj_w, f2, L.n, C, C, 1, nullptr, nullptr, proj_b, nullptr, nullptr, 0, f1, nullptr, nullptr);
        RUN(ptv3_layernorm(f2, n2_g, n2_b, nullptr, t5, nullptr, nullptr, nullptr, L.n, C, d->ln_eps, d->dtype, sf));
        gemm(t5, fc1_w, t6, L.n, C, hidden, 1, nullptr, nullptr, fc1_b, nullptr, nullptr, PTV3_ACT_GELU, nullptr, nullptr, nullptr);
      }
      gemm(t6, fc2_w, L.feat, L.n, hidden, C, 1, nullptr, nullptr, fc2_b, nullptr, nullptr, 0, f2, nullptr, nullptr);
    }
    L.conv_feat = L.feat;
    F->off = mark;
  }
};

#define RUNR(call) do { if (!dry && R.ok()) { int r__ = (call); if (r__) R.rc = r__; } } while (0)

static int run_forward(Exec* X, const ptv3_model_desc* d, const void* const* params, const ptv3_forward_io* io, Arena& G,
                       Arena& F, hipStream_t sf, bool dry, int* param_count = nullptr) {
  Run R; R.X = X; R.d = d; R.params = params; R.G = &G; R.F = &F; R.sf = sf; R.dry = dry;
  R.es = d->dtype == PTV3_F32 ? 4 : 2;
  const int S = d->num_stages, k = d->num_orders, es = R.es;
  if (!dry) {
    if (!X->geo) {
      // highest priority: the geometry chain is a string of tiny dependent kernels whose read-backs gate the
      // host; its workgroups must not queue behind the long feature kernels of the other stream
      X->geo = device_stream(X->device, 0);
      if (!X->geo) {
        set_error("forward: cannot create the geometry stream");
        return PTV3_ERR_LAUNCH;
      }
    }
    if (!io->inputs_resident) {
      // coordinates / offsets may have been produced on the caller's stream just before this call
      hipEvent_t e = X->event_at(0);
      (void)hipEventRecord(e, sf);
      (void)hipStreamWaitEvent(X->geo, e, 0);
    } else {
      // inputs already materialised: geometry of this call may overlap the feature tail of the previous one;
      // it only has to wait for the call that last used THIS geometry arena (two calls ago)
      (void)hipStreamWaitEvent(X->geo, X->event_at(20 + (X->call & 1)), 0);
    }
  }
  R.sg = dry ? nullptr : X->geo;
  hipStream_t sg = R.sg;
  std::vector<Level> lv(S);
  const int nbits = io->b > 1 ? 32 - __builtin_clz((unsigned)(io->b - 1)) : 0;

  // ---------------- level 0 geometry: depth, scene offsets, serialization (structure.py:52-109)
  Level& L0 = lv[0];
  L0.n = io->n; L0.batch = io->batch; L0.offset = io->offset;
  L0.code = io->code; L0.order = io->order; L0.inverse = io->inverse;
  L0.off_host.resize(io->b);
  int depth = io->depth;
  if (dry) {
    for (int i = 0; i < io->b; ++i) L0.off_host[i] = io->offset_host ? io->offset_host[i] : io->n * (i + 1) / io->b;
    if (depth <= 0) { depth = 16; G.alloc(256); }
  } else {
    int* dmax = nullptr;
    if (depth <= 0) {
      dmax = (int*)G.alloc(256);
      if (G.failed) { set_error("forward: workspace arena too small"); return PTV3_ERR_ARG; }
      (void)hipMemsetAsync(dmax, 0, 4, sg);
      hipLaunchKernelGGL(coord_max_kernel, dim3(256), dim3(256), 0, sg, io->grid_coord, io->coord_is_i64, 3 * io->n, dmax);
    }
    int hmax = 0;
    if (dmax) (void)hipMemcpyAsync(&hmax, dmax, 4, hipMemcpyDeviceToHost, sg);
    if (io->offset_host) std::copy(io->offset_host, io->offset_host + io->b, L0.off_host.begin());
    else (void)hipMemcpyAsync(L0.off_host.data(), io->offset, (size_t)io->b * 8, hipMemcpyDeviceToHost, sg);
    if (dmax || !io->offset_host) {
      if (X->timed_sync(sg) != hipSuccess) { set_error("forward: input read-back failed"); return PTV3_ERR_LAUNCH; }
    }
    if (dmax) { depth = 0; for (unsigned v = (unsigned)hmax + 1; v; v >>= 1) ++depth; }
    PTV3_REQUIRE(depth >= 1 && depth <= 16, "forward: serialization depth %d outside [1,16] (structure.py:81)", depth);
    for (int i = 0; i < io->b; ++i)
      PTV3_REQUIRE(L0.off_host[i] > (i ? L0.off_host[i - 1] : 0), "forward: empty scene %d", i);
    PTV3_REQUIRE(L0.off_host[io->b - 1] == io->n, "forward: offset does not end at n");
  }
  L0.depth = depth;
  if (io->depth_out) *io->depth_out = depth;
  if (io->batch == nullptr) {  // derive the batch ids on the geometry stream (no torch ops on the caller's stream)
    int64_t* bt = io->batch_out ? io->batch_out : (int64_t*)G.alloc((size_t)io->n * 8);
    if (!dry && R.ok())
      hipLaunchKernelGGL(batch_from_offset_kernel, dim3((unsigned)cdiv(io->n, 256)), dim3(256), 0, sg, io->offset, io->b,
                         io->n, bt);
    L0.batch = bt;
  }
  {
    RUNR(ptv3_sfc_encode(io->grid_coord, io->coord_is_i64, L0.batch, io->n, depth, io->order_ids_host, k, L0.code, sg));
    size_t wsb = ptv3_argsort_workspace_bytes(k, io->n);
    void* ws = G.alloc(wsb);
    RUNR(ptv3_argsort_i64(L0.code, k, io->n, std::max(1, 3 * depth + nbits), L0.order, L0.inverse, ws, wsb, sg));
  }
  if (!dry) R.stem_ready = X->event_at(19);
  R.level_geometry(L0, 0, io->grid_coord, io->coord_is_i64, true);
  if (io->stage_points_host) io->stage_points_host[0] = L0.n;
  if (!dry && R.ok()) { L0.ready = X->event_at(1); (void)hipEventRecord(L0.ready, sg); }

  // ---------------- features of level 0 start as soon as its geometry is queued
  auto wait_level = [&](Level& L) { if (!dry && R.ok()) (void)hipStreamWaitEvent(sf, L.ready, 0); };
  struct PoolParams { const void* w; const float *b, *bns, *bnt; };
  // parameter walk order (header): stem, enc stage 0 blocks, [down, blocks] ..., decoder, head.
  const void* stem_w = R.next(); const float* stem_s = (const float*)R.next(); const float* stem_t = (const float*)R.next();
  {
    if (!dry && R.ok()) (void)hipStreamWaitEvent(sf, R.stem_ready, 0);
    const int C0 = d->enc_channels[0];
    L0.feat = F.alloc((size_t)L0.n * C0 * es); L0.channels = C0;
    const void* feat_in = io->feat;
    if (io->raw_feat) {
      // overlap mode: pad + cast the caller's features here, on the executor's stream (no producer on `stream`)
      void* padded = F.alloc((size_t)L0.n * d->in_channels * es);
      if (!dry && R.ok())
        pad_cast(io->raw_feat, io->raw_feat_dtype, io->raw_feat_channels, padded, d->dtype, d->in_channels, L0.n, sf);
      feat_in = padded;
    }
    R.gemm(feat_in, stem_w, L0.feat, L0.n, d->in_channels, C0, 125, L0.nbr5, L0.row_order, nullptr, stem_s, stem_t,
           PTV3_ACT_GELU, nullptr, nullptr, nullptr);
    L0.conv_feat = L0.feat;
    wait_level(L0);    // the blocks need the 3^3 table and the window plans
  }
  for (int st = 0; st < S; ++st) {
    Level& L = lv[st];
    if (st > 0) {
      Level& P = lv[st - 1];
      PoolParams pp;
      pp.w = R.next(); pp.b = (const float*)R.next(); pp.bns = (const float*)R.next(); pp.bnt = (const float*)R.next();
      const int C = d->enc_channels[st], Cp = d->enc_channels[st - 1];
      int pd = 0; { int v = d->stride[st - 1] - 1; while (v > 0) { ++pd; v >>= 1; } }
      if (pd > P.depth) pd = 0;
      // ---- geometry of the pooled level (geometry stream; its read-back never drains the feature stream)
      P.cluster = (int64_t*)G.alloc((size_t)P.n * 8);
      P.seg = (int32_t*)G.alloc((size_t)(P.n + 1) * 4);
      int64_t* poff = (int64_t*)G.alloc((size_t)io->b * 8);
      int32_t* nout_dev = (int32_t*)G.alloc(256);
      size_t pwsb = ptv3_pool_workspace_bytes(P.n);
      void* pws = G.alloc(pwsb);
      RUNR(ptv3_pool_segments(P.code, P.order, P.n, 3 * pd, P.batch, P.cluster, P.seg, nout_dev, poff, pws, pwsb, sg));
      L.off_host.resize(io->b);
      if (dry) {
        L.off_host = P.off_host;  // worst case: nothing merges
      } else if (R.ok()) {
        if (hipMemcpyAsync(L.off_host.data(), poff, (size_t)io->b * 8, hipMemcpyDeviceToHost, sg) != hipSuccess ||
            X->timed_sync(sg) != hipSuccess) { set_error("forward: offset read-back failed"); return PTV3_ERR_LAUNCH; }
      }
      if (!R.ok()) break;
      L.n = L.off_host[io->b - 1]; L.depth = P.depth - pd; L.offset = poff; L.channels = C;
      if (io->stage_points_host) io->stage_points_host[st] = L.n;
      const int* perm = io->pool_perm_host + (size_t)(st - 1) * k;  // order shuffle of :408-412
      int64_t* g = (int64_t*)G.alloc((size_t)L.n * 24);
      int64_t* bt = (int64_t*)G.alloc((size_t)L.n * 8);
      L.code = (int64_t*)G.alloc((size_t)k * L.n * 8);
      L.order = (int64_t*)G.alloc((size_t)k * L.n * 8);
      L.inverse = (int64_t*)G.alloc((size_t)k * L.n * 8);
      RUNR(ptv3_pool_reduce(nullptr, nullptr, P.grid, P.batch, P.code, k, P.order, P.seg, P.n, L.n, C, pd, nullptr,
                           nullptr, 0, perm, nullptr, nullptr, g, bt, L.code, d->dtype, sg));
      L.grid = g; L.batch = bt;
      {
        size_t wsb = ptv3_argsort_workspace_bytes(k, L.n);
        void* ws = G.alloc(wsb);
        RUNR(ptv3_argsort_i64(L.code, k, L.n, std::max(1, 3 * L.depth + nbits), L.order, L.inverse, ws, wsb, sg));
      }
      R.level_geometry(L, st, L.grid, 1, false);
      if (!d->enc_mode) {
        P.cluster32 = (int32_t*)G.alloc((size_t)P.n * 4);
        if (!dry && R.ok())
          hipLaunchKernelGGL(i64_to_i32_kernel, dim3((unsigned)cdiv(P.n, 256)), dim3(256), 0, sg, P.cluster, P.cluster32, P.n);
      }
      if (!dry && R.ok()) { L.ready = X->event_at(1 + st); (void)hipEventRecord(L.ready, sg); }
      // ---- features of the pooled level: proj -> segmented max + folded BN + GELU (:416-418, 439-442)
      wait_level(L);
      L.feat = F.alloc((size_t)L.n * C * es);
      const size_t mark = F.off;
      void* proj = F.alloc((size_t)P.n * C * es);
      R.gemm(P.feat, pp.w, proj, P.n, Cp, C, 1, nullptr, nullptr, pp.b, nullptr, nullptr, 0, nullptr, nullptr, nullptr);
      RUNR(ptv3_pool_reduce(proj, nullptr, nullptr, nullptr, nullptr, k, P.order, P.seg, P.n, L.n, C, pd, pp.bns, pp.bnt,
                           PTV3_ACT_GELU, nullptr, L.feat, nullptr, nullptr, nullptr, nullptr, d->dtype, sf));
      F.off = mark;
      L.conv_feat = L.feat;
    }
    for (int i = 0; i < d->enc_depths[st]; ++i)
      R.block(L, L.plan[0], d->enc_channels[st], d->enc_heads[st], i % k);
    if (!R.ok()) break;
  }
  // ---------------- decoder (:651-697)
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
      void* skip = F.alloc((size_t)P.n * Cd * es);
      void* fused = (st == 0 && io->out_feat) ? io->out_feat : F.alloc((size_t)P.n * Cd * es);
      const size_t mark = F.off;
      void* up = F.alloc((size_t)Ch.n * Cd * es);
      R.gemm(Ch.feat, w, up, Ch.n, Cin, Cd, 1, nullptr, nullptr, b, bns, bnt, PTV3_ACT_GELU, nullptr, nullptr, nullptr);
      // skip branch; its epilogue gathers the up-branch rows by cluster id.  `skip` alone stays the sparse
      // tensor's features for the next block's conv (the reference never refreshes it after the add, :478)
      R.gemm(P.feat, ws_, skip, P.n, Cskip, Cd, 1, nullptr, nullptr, bs, bnss, bnst, PTV3_ACT_GELU, up, P.cluster32, fused);
      F.off = mark;
      P.feat = fused; P.conv_feat = skip; P.channels = Cd;
      for (int i = 0; i < d->dec_depths[st]; ++i)
        R.block(P, P.plan[1], Cd, d->dec_heads[st], i % k);
      if (!R.ok()) break;
    }
  }
  // ---------------- dense head: Linear -> folded BN -> ReLU -> Linear (offset_keypoint_ptv3.py:26-31)
  if (R.ok() && d->head_out > 0 && !d->enc_mode) {
    Level& L = lv[0];
    const void* w0 = R.next(); const float* b0 = (const float*)R.next();
    const float* bns = (const float*)R.next(); const float* bnt = (const float*)R.next();
    const void* w1 = R.next(); const float* b1 = (const float*)R.next();
    const size_t mark = F.off;
    if (ptv3_mlp2_fusable(L.channels, d->head_hidden, d->head_out, d->dtype)) {
      // w1 arrives chain-permuted and row-padded (ptv3_hip/engine.py packs it by the same predicate)
      if (!dry && R.ok()) {
        int r = ptv3_mlp2(L.feat, w0, b0, bns, bnt, PTV3_ACT_RELU, w1, b1, io->out_head, 1, L.n, L.channels,
                          d->head_hidden, d->head_out, d->dtype, sf);
        if (r) R.rc = r;
      }
    } else {
    void* hid = F.alloc((size_t)L.n * d->head_hidden * es);
    R.gemm(L.feat, w0, hid, L.n, L.channels, d->head_hidden, 1, nullptr, nullptr, b0, bns, bnt, PTV3_ACT_RELU, nullptr,
           nullptr, nullptr);
    if (d->dtype == PTV3_F32) {
      R.gemm(hid, w1, io->out_head, L.n, d->head_hidden, d->head_out, 1, nullptr, nullptr, b1, nullptr, nullptr, 0,
             nullptr, nullptr, nullptr);
    } else {
      void* o = F.alloc((size_t)L.n * d->head_out * es);
      R.gemm(hid, w1, o, L.n, d->head_hidden, d->head_out, 1, nullptr, nullptr, b1, nullptr, nullptr, 0, nullptr,
             nullptr, nullptr);
      if (!dry && R.ok()) { int r = ptv3_cast(o, PTV3_BF16, io->out_head, PTV3_F32, L.n * d->head_out, sf); if (r) R.rc = r; }
    }
    }
    F.off = mark;
  }
  if (!dry && R.ok() && !d->enc_mode && io->out_feat && lv[0].feat != io->out_feat)
    (void)hipMemcpyAsync(io->out_feat, lv[0].feat, (size_t)lv[0].n * lv[0].channels * es, hipMemcpyDeviceToDevice, sf);
  if (!dry) (void)hipEventRecord(X->event_at(20 + (X->call & 1)), sf);  // this call's readers of its geometry arena
  if (param_count) *param_count = R.pi;
  if (G.failed || F.failed) { set_error("forward: workspace arena too small"); return PTV3_ERR_ARG; }
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

// worst-case (nothing merges at any pooling) sizes of the two arena parts
static void plan_bytes(const ptv3_model_desc* desc, int64_t n, int b, size_t* geo, size_t* feat, int* count) {
  std::vector<int> ids(desc->num_orders, 0), perm((size_t)desc->num_stages * desc->num_orders, 0);
  ptv3_forward_io io{};
  io.n = n; io.b = b; io.depth = 0; io.order_ids_host = ids.data(); io.pool_perm_host = perm.data();
  io.coord_is_i64 = 1;
  Arena G{nullptr, 0, 0, 0, true, false}, F{nullptr, 0, 0, 0, true, false};
  run_forward(nullptr, desc, nullptr, &io, G, F, nullptr, true, count);
  // slack: every scene may gain up to one window of borrowed rows per window plan
  *geo = (G.peak + (size_t)b * 16384 * 64 + (1 << 20) + 255) / 256 * 256;
  *feat = (F.peak + (1 << 20) + 255) / 256 * 256;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" size_t ptv3_forward_workspace_bytes(const ptv3_model_desc* desc, int64_t n, int b) {
  if (check_desc(desc)) return 0;
  size_t g, f;
  plan_bytes(desc, n, b, &g, &f, nullptr);
  return 2 * g + 2 * f + 512;  // two geometry arenas (consecutive calls alternate), two feature arenas (overlap_calls)
}

extern "C" void* ptv3_executor_create(void) {
  Exec* X = new Exec();
  (void)hipGetDevice(&X->device);
  return X;
}

extern "C" int ptv3_executor_destroy(void* executor) {
  Exec* X = (Exec*)executor;
  if (!X) return PTV3_OK;
  {
    std::lock_guard<std::mutex> guard(X->mu);  // no call of this executor is on the host any more
    X->release();                                // drains its streams first
  }
  delete X;
  return PTV3_OK;
}

extern "C" int ptv3_forward(const ptv3_model_desc* desc, const void* const* params, int num_params,
                            const ptv3_forward_io* io, void* workspace, size_t workspace_bytes, void* stream) {
  // streams, events, call parity and arena halves belong to an executor (io->executor, or the current device's
  // default one): calls on ONE executor from several host threads take turns (they still overlap on the GPU exactly
  // as consecutive calls of one thread do); different executors are independent
  Exec* X = io->executor ? (Exec*)io->executor : default_exec();
  std::lock_guard<std::mutex> guard(X->mu);
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  PTV3_REQUIRE(cur_dev == X->device, "forward: executor belongs to device %d, the current device is %d", X->device, cur_dev);
  if (int r = check_desc(desc)) return r;
  PTV3_REQUIRE(io->n >= 1 && io->b >= 1, "forward: empty batch");
  PTV3_REQUIRE(io->depth <= 16, "forward: depth %d > 16", io->depth);
  PTV3_REQUIRE(io->code && io->order && io->inverse, "forward: level-0 serialization outputs are required");
  size_t g, f;
  int count = 0;
  plan_bytes(desc, io->n, io->b, &g, &f, &count);
  PTV3_REQUIRE(count == num_params, "forward: %d parameter pointers given, the model description needs %d",
               num_params, count);
  const bool overlap = io->overlap_calls != 0;
  PTV3_REQUIRE(!overlap || io->inputs_resident, "forward: overlap_calls needs inputs_resident");
  PTV3_REQUIRE(io->raw_feat == nullptr || (io->raw_feat_channels >= 1 && io->raw_feat_channels <= desc->in_channels &&
                                           (io->raw_feat_dtype == PTV3_F32 || io->raw_feat_dtype == PTV3_BF16)),
               "forward: bad raw_feat description");
  if (io->arena_n > 0) {
    // fixed layout: the parts are placed for the capacity the workspace was sized with, not for this call's n
    PTV3_REQUIRE(io->arena_n >= io->n && io->arena_b >= io->b, "forward: arena capacity (%lld, %d) below this call (%lld, %d)",
                 (long long)io->arena_n, io->arena_b, (long long)io->n, io->b);
    size_t gc, fc;
    plan_bytes(desc, io->arena_n, io->arena_b, &gc, &fc, nullptr);
    PTV3_REQUIRE(gc >= g && fc >= f, "forward: arena capacity plan smaller than this call's plan");
    g = gc; f = fc;
  } else {
    PTV3_REQUIRE(!io->inputs_resident || !overlap, "forward: overlap_calls needs a fixed arena layout (arena_n / arena_b)");
  }
  const size_t need = 2 * g + (overlap ? 2 : 1) * f + 256;
  PTV3_REQUIRE(workspace_bytes >= need, "forward: workspace %zu bytes < %zu (ptv3_forward_workspace_bytes)",
               workspace_bytes, need);
  char* base = (char*)workspace;
  size_t goff = (256 - ((uintptr_t)base & 255)) & 255;
  ++X->call;
  static const bool timing = getenv("PTV3_ENGINE_TIMING") != nullptr;
  const double t_begin = timing ? now_us() : 0.0;
  X->sync_us = 0.0;
  X->event_at(21);  // make sure the two arena events exist (an unrecorded event never blocks a wait)
  const unsigned par = X->call & 1;
  hipStream_t caller = (hipStream_t)stream, sf = caller;
  if (overlap) {
    if (!X->feat[par]) X->feat[par] = device_stream(X->device, 1 + par);
    if (!X->feat[par]) {
      set_error("forward: cannot create the feature stream");
      return PTV3_ERR_LAUNCH;
    }
    sf = X->feat[par];
    X->event_at(34);
    if (!X->last_overlap) {
      // first overlapped call after a stream-ordered one (or the first call at all): that call ran its feature
      // pipeline on the caller's stream out of the first feature arena - order the executor's streams behind it.
      // Reruns after EVERY stream-ordered call (last_overlap is the mode of the previous call, set below).
      (void)hipEventRecord(X->event_at(34), caller);
      for (int q = 0; q < 2; ++q)
        if (X->feat[q]) (void)hipStreamWaitEvent(X->feat[q], X->event_at(34), 0);
    }
    // the output buffers this call overwrites were last read by work the caller enqueued before the PREVIOUS call
    // started (contract in the header): wait for that marker, never for the previous call itself
    (void)hipStreamWaitEvent(sf, X->event_at(30 + (par ^ 1)), 0);
    (void)hipEventRecord(X->event_at(30 + par), caller);
  }
  X->last_overlap = overlap;
  Arena G{base + goff + par * g, g, 0, 0, false, false};
  Arena F{base + goff + 2 * g + (overlap ? par * f : 0), f, 0, 0, false, false};
  int rc = run_forward(X, desc, params, io, G, F, sf, false, nullptr);
  if (overlap) {
    (void)hipEventRecord(X->event_at(32 + par), sf);
    (void)hipStreamWaitEvent(caller, X->event_at(32 + par), 0);  // the caller's later work sees this call's outputs
  }
  if (timing)
    fprintf(stderr, "[ptv3_forward] host %.0f us total, %.0f us blocked in read-backs\n", now_us() - t_begin, X->sync_us);
  return rc;
}
