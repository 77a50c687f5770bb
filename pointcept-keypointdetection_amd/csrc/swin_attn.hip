// Swin3D window partition keys and cRSE window attention for gfx950.
//
// Reference: pointcept/models/swin3d/swin3d_layers.py:715-795 (window mapping through MinkowskiEngine pooling and
// kernel maps), :826-840 (shifted windows), :482-577 (WindowAttention.forward, which hands q, k, v, three
// concatenated cRSE tables and the index tuple to the external SelfAttnAIOFunction).  Neither MinkowskiEngine nor
// microsoft/Swin3D is in the reference tree: the arithmetic follows the Swin3D paper's contextual relative signal
// encoding as restated in oracle/swin3d.py (parity unpinned, see its header).
//
// Layout: one workgroup per (window, head).  A window holds at most window_size^3 occupied voxels (125 or 343 in
// the S3DIS config), far below a matrix-core tile's worth of rows for most windows, and every (query, key) pair
// needs 3 table rows per signal axis chosen by a data-dependent index, so the kernel is a gather machine rather
// than a GEMM: K, V and the signal vectors of the window sit in LDS, one wave takes one query at a time with its
// lanes across the keys, table rows come from global memory (the per-head slice of all tables is tens of KB and
// stays in L2 / the vector L1).  fp32 arithmetic throughout; bf16 only as the storage type of q, k, v, out.
#include "common.h"
#include "profile.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int SWIN_MAX_AXES = 9;
constexpr int SWIN_MAX_TOKENS = 512;   // w_w_id has 9 bits in the window key
constexpr int SWIN_WAVES = 4;

int swin_attn_mfma(int dtype, int D, int S, const void* q, const void* k, const void* v, const float* qt, const float* kt,
                   const float* vt, const long long* start, const int* rows, const long long* n2n, const int* w_start,
                   int nwin, const float* crse, void* out, int heads, int max_tokens, hipStream_t s);

struct SwinTables {
  long long start[SWIN_MAX_AXES];   // element offset of axis c's slab inside each concatenated table
  int rows[SWIN_MAX_AXES];          // 2 L_c
};

// key = (batch, window x, y, z) packed above the 9-bit position inside the window; sorting the keys gives the
// reference's sort by in_map = window * ws^3 + w_w_id (windows numbered lexicographically).
__global__ __launch_bounds__(256) void swin_window_keys_kernel(const int* __restrict__ coords, long long n, int stride,
                                                                int ws, int shift, long long* __restrict__ key,
                                                                int* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = coords[i * 4];
  long long w[3];
  int loc[3];
  bool ok = b >= 0 && b < 4096;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int c = coords[i * 4 + 1 + a];
    int v = c / stride;
    if (c % stride != 0 && c < 0) --v;   // floor
    v += shift;
    int q = v / ws;
    if (v % ws != 0 && v < 0) --q;
    loc[a] = v - q * ws;
    w[a] = (long long)q + 4096;
    ok = ok && w[a] >= 0 && w[a] < 8192;
  }
  if (!ok) atomicOr(bad, 1);
  const long long win = (((long long)b * 8192 + w[0]) * 8192 + w[1]) * 8192 + w[2];
  key[i] = (win << 9) | (long long)((loc[0] * ws + loc[1]) * ws + loc[2]);
}

template <int D>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float* r) {
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + d);
    r[d] = t[0]; r[d + 1] = t[1]; r[d + 2] = t[2]; r[d + 3] = t[3];
  }
}
template <int D>
__device__ __forceinline__ float dot(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) s = fmaf(a[d], b[d], s);
  return s;
}
// reductions over the G lanes (a power of two, 16..64) that share one query
__device__ __forceinline__ float group_max(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float group_sum(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One workgroup per (window, head).  Inside a window a query is served by G = max(16, D, pow2ceil(m)) lanes (<= 64),
// so a wave carries 64 / G queries: windows of 20 occupied voxels keep 40 of 64 lanes busy instead of 20.
// (Measured and not kept: the head's table slice staged into LDS with a run of windows per workgroup - the 27 random
// 64-byte row gathers per pair then queue on the LDS pipe, with 4-way bank conflicts between rows, and the per-window
// barriers serialise the one workgroup a CU can hold: 27.5 ms against 19.1 ms for 300k voxels in 7^3 windows.)
template <typename T, int D, int S>
__global__ __launch_bounds__(SWIN_WAVES * 64) void swin_attn_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ qt,
    const float* __restrict__ kt, const float* __restrict__ vt, SwinTables tab, const long long* __restrict__ n2n,
    const int* __restrict__ w_start, const float* __restrict__ crse, T* __restrict__ out, int heads, int max_tokens) {
  constexpr int RS = D + 4;                       // LDS row stride in floats (16-byte aligned rows)
  constexpr int NT = SWIN_WAVES * 64;
  extern __shared__ __align__(16) unsigned char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + (size_t)max_tokens * RS;
  float* sC = sV + (size_t)max_tokens * RS;       // [max_tokens][S]
  const int lcap = max_tokens > 64 ? max_tokens : 64;
  float* sL = sC + (size_t)max_tokens * S;        // [SWIN_WAVES][lcap] logits of the wave's current queries
  int* sRow = reinterpret_cast<int*>(sL + (size_t)SWIN_WAVES * lcap);

  const int w = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t hoff = (size_t)h * D;
  const size_t rstride = (size_t)heads * D;
  const int s0 = w_start[w];
  int m = w_start[w + 1] - s0;
  if (m > max_tokens) m = max_tokens;             // host contract; never true for a valid partition
  for (int t = tid; t < m; t += NT) sRow[t] = (int)n2n[s0 + t];
  for (int e = tid; e < m * S; e += NT) sC[e] = crse[(size_t)s0 * S + e];
  __syncthreads();
  for (int e = tid; e < m * (D / 4); e += NT) {
    const int t = e / (D / 4), d = (e % (D / 4)) * 4;
    const size_t g = ((size_t)sRow[t] * heads + h) * D + d;
    float kk[4], vv[4];
    unpack4<T>(*reinterpret_cast<const typename Vec4<T>::type*>(k + g), kk);
    unpack4<T>(*reinterpret_cast<const typename Vec4<T>::type*>(v + g), vv);
    *reinterpret_cast<f32x4*>(sK + t * RS + d) = f32x4{kk[0], kk[1], kk[2], kk[3]};
    *reinterpret_cast<f32x4*>(sV + t * RS + d) = f32x4{vv[0], vv[1], vv[2], vv[3]};
  }
  __syncthreads();

  float* myL = sL + (size_t)wave * lcap;
  int G = D > 16 ? D : 16;
  while (G < m && G < 64) G <<= 1;
  const int qpw = 64 / G;                         // queries per wave
  const int gl = lane & (G - 1), qs = lane / G;
  for (int base = wave * qpw; base < m; base += SWIN_WAVES * qpw) {
    const int i = base + qs;
    const bool live = i < m;
    const int ii = live ? i : m - 1;              // idle groups shadow the last query (no divergent shuffles)
    float qi[D], ci[S];
    {
      const size_t g = ((size_t)sRow[ii] * heads + h) * D;
#pragma unroll
      for (int d = 0; d < D; d += 4) unpack4<T>(*reinterpret_cast<const typename Vec4<T>::type*>(q + g + d), qi + d);
#pragma unroll
      for (int c = 0; c < S; ++c) ci[c] = sC[ii * S + c];
    }
    float* L = myL + qs * m;
    // pass 1: logits of every key of the window
    float mx = -INFINITY;
    for (int j = gl; j < m; j += G) {
      float kj[D];
      load_row<D>(sK + j * RS, kj);
      float e = dot<D>(qi, kj);
#pragma unroll
      for (int c = 0; c < S; ++c) {
        int idx = (int)floorf((ci[c] - sC[j * S + c]) + (float)(tab.rows[c] >> 1));
        idx = min(max(idx, 0), tab.rows[c] - 1);
        const size_t r = (size_t)tab.start[c] + (size_t)idx * rstride + hoff;
        float tk[D], tq[D];
        load_row<D>(kt + r, tk);
        load_row<D>(qt + r, tq);
        e += dot<D>(qi, tk) + dot<D>(kj, tq);
      }
      L[j] = e;
      mx = fmaxf(mx, e);
    }
    mx = group_max(mx, G);
    // pass 2: weights, value + value-table rows
    float acc[D], den = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    for (int j = gl; j < m; j += G) {
      const float p = __expf(L[j] - mx);
      den += p;
      float vj[D];
      load_row<D>(sV + j * RS, vj);
#pragma unroll
      for (int c = 0; c < S; ++c) {
        int idx = (int)floorf((ci[c] - sC[j * S + c]) + (float)(tab.rows[c] >> 1));
        idx = min(max(idx, 0), tab.rows[c] - 1);
        float tv[D];
        load_row<D>(vt + (size_t)tab.start[c] + (size_t)idx * rstride + hoff, tv);
#pragma unroll
        for (int d = 0; d < D; ++d) vj[d] += tv[d];
      }
#pragma unroll
      for (int d = 0; d < D; ++d) acc[d] = fmaf(p, vj[d], acc[d]);
    }
    den = group_sum(den, G);
    const float inv = 1.f / den;
    float mine = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float t = group_sum(acc[d], G);
      if (gl == d) mine = t * inv;
    }
    if (live && gl < D) out[((size_t)sRow[i] * heads + h) * D + gl] = from_f32<T>(mine);
  }
}

template <typename T, int D, int S>
static int launch_swin(const void* q, const void* k, const void* v, const float* qt, const float* kt, const float* vt,
                       const SwinTables& tab, const long long* n2n, const int* w_start, int nwin, const float* crse,
                       void* out, int heads, int max_tokens, hipStream_t s) {
  const int lcap = max_tokens > 64 ? max_tokens : 64;
  const size_t lds = ((size_t)max_tokens * (2 * (D + 4) + S) + (size_t)SWIN_WAVES * lcap) * 4 + (size_t)max_tokens * 4;
  if (lds > 160 * 1024) {
    set_error("swin_attn: %d tokens x head_dim %d needs %zu bytes of LDS", max_tokens, D, lds);
    return PTV3_ERR_UNSUPPORTED;
  }
  ensure_dynamic_lds(reinterpret_cast<const void*>(&swin_attn_kernel<T, D, S>), 160 * 1024);
  hipLaunchKernelGGL((swin_attn_kernel<T, D, S>), dim3((unsigned)nwin, (unsigned)heads), dim3(SWIN_WAVES * 64), lds, s,
                     (const T*)q, (const T*)k, (const T*)v, qt, kt, vt, tab, n2n, w_start, crse, (T*)out, heads,
                     max_tokens);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

template <typename T, int D>
static int dispatch_axes(int S, const void* q, const void* k, const void* v, const float* qt, const float* kt,
                         const float* vt, const SwinTables& tab, const long long* n2n, const int* w_start, int nwin,
                         const float* crse, void* out, int heads, int max_tokens, hipStream_t s) {
  switch (S) {
    case 3: return launch_swin<T, D, 3>(q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
    case 6: return launch_swin<T, D, 6>(q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
    case 9: return launch_swin<T, D, 9>(q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
  }
  set_error("swin_attn: %d signal axes (3, 6 or 9: XYZ, XYZ_RGB, XYZ_RGB_NORM)", S);
  return PTV3_ERR_UNSUPPORTED;
}

template <typename T>
static int dispatch_dim(int D, int S, const void* q, const void* k, const void* v, const float* qt, const float* kt,
                        const float* vt, const SwinTables& tab, const long long* n2n, const int* w_start, int nwin,
                        const float* crse, void* out, int heads, int max_tokens, hipStream_t s) {
  switch (D) {
    case 8: return dispatch_axes<T, 8>(S, q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
    case 16: return dispatch_axes<T, 16>(S, q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
    case 32: return dispatch_axes<T, 32>(S, q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
  }
  set_error("swin_attn: head_dim %d (8, 16 or 32)", D);
  return PTV3_ERR_UNSUPPORTED;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_swin_window_keys(const int32_t* coords, int64_t n, int stride, int window_size, int shift,
                                     int64_t* key, int32_t* bad, void* stream) {
  PTV3_REQUIRE(stride > 0, "swin_window_keys: tensor stride %d", stride);
  PTV3_REQUIRE(window_size > 0 && window_size * window_size * window_size <= SWIN_MAX_TOKENS,
               "swin_window_keys: window_size %d (window_size^3 <= %d)", window_size, SWIN_MAX_TOKENS);
  PTV3_REQUIRE(bad != nullptr, "swin_window_keys: the range flag is required");
  hipStream_t s = (hipStream_t)stream;
  (void)hipMemsetAsync(bad, 0, sizeof(int32_t), s);
  if (n > 0)
    hipLaunchKernelGGL(swin_window_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, coords, (long long)n,
                       stride, window_size, shift, (long long*)key, bad);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_swin_attn_fwd(const void* q, const void* k, const void* v, const float* q_table,
                                  const float* k_table, const float* v_table, const int32_t* table_offsets_host,
                                  int num_axes, const int64_t* n2n, const int32_t* w_start, int num_windows,
                                  const float* n_crse, void* out, int64_t n, int heads, int head_dim, int max_tokens,
                                  int dtype, void* stream) {
  PTV3_REQUIRE(num_axes > 0 && num_axes <= SWIN_MAX_AXES, "swin_attn: %d signal axes", num_axes);
  PTV3_REQUIRE(heads > 0 && head_dim > 0, "swin_attn: heads %d head_dim %d", heads, head_dim);
  PTV3_REQUIRE(max_tokens > 0 && max_tokens <= SWIN_MAX_TOKENS, "swin_attn: max_tokens %d (1..%d)", max_tokens,
               SWIN_MAX_TOKENS);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "swin_attn: dtype %d", dtype);
  PTV3_REQUIRE(n < (1ll << 31), "swin_attn: %lld voxels", (long long)n);
  SwinTables tab;
  long long at = 0;
  for (int c = 0; c < num_axes; ++c) {
    const int per = heads * head_dim;
    PTV3_REQUIRE(table_offsets_host[c] > 0 && table_offsets_host[c] % (2 * per) == 0,
                 "swin_attn: table_offsets[%d] = %d is not an even number of (heads x head_dim) rows", c,
                 table_offsets_host[c]);
    tab.start[c] = at;
    tab.rows[c] = table_offsets_host[c] / per;
    at += table_offsets_host[c];
  }
  for (int c = num_axes; c < SWIN_MAX_AXES; ++c) { tab.start[c] = 0; tab.rows[c] = 2; }
  if (num_windows <= 0 || n <= 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic work: (1 + 3 S) 2 D flops per (query, key) pair and head - the pair count lives in w_start on the
  // device; a caller that knows it says so through ptv3_profile_hint_flops - and q, k, v, out once each
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_WINDOW_ATTN, 0.0, 4.0 * (double)n * heads * head_dim * esz + 12.0 * n, nullptr, 0, 0.0);
  prof_kernel(prof, PK_SWIN_ATTN);
  // table products on the matrix core (swin_attn_mfma.hip) unless switched off or the window does not fit its LDS plan
  const char* mfma_env = getenv("PTV3_SWIN_ATTN_MFMA");      // read per call: tests run both kernels on the same inputs
  const bool mfma_on = !(mfma_env && atoi(mfma_env) == 0);
  int rc = -1;
  if (mfma_on && at < (1ll << 31))
    rc = swin_attn_mfma(dtype, head_dim, num_axes, q, k, v, q_table, k_table, v_table, tab.start, tab.rows,
                        (const long long*)n2n, w_start, num_windows, n_crse, out, heads, max_tokens, s);
  if (rc != -1) {
    prof_kernel(prof, PK_SWIN_ATTN_MFMA);
  } else if (dtype == PTV3_F32)
    rc = dispatch_dim<float>(head_dim, num_axes, q, k, v, q_table, k_table, v_table, tab, (const long long*)n2n,
                             w_start, num_windows, n_crse, out, heads, max_tokens, s);
  else
    rc = dispatch_dim<__bf16>(head_dim, num_axes, q, k, v, q_table, k_table, v_table, tab, (const long long*)n2n,
                              w_start, num_windows, n_crse, out, heads, max_tokens, s);
  prof_end(prof, s);
  return rc;
}
