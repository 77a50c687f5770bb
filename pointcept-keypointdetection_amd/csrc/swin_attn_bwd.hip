// Backward of the Swin3D cRSE window attention (ptv3_swin_attn_fwd): gradients of q, k, v and of the three tables.
//
// The reference trains this op through Swin3D.sparse_dl's SelfAttnAIOFunction.backward (microsoft/Swin3D, not in the
// reference tree; call site pointcept/models/swin3d/swin3d_layers.py:556-569).  What is differentiated here is the
// forward this package computes (oracle/swin3d.py: PARITY UNPINNED); the checker is torch autograd over a torch
// restatement of that forward (tests/test_hip_swin3d.py).
//   e_ij = q_i.k_j + sum_c (q_i.TK_c[r] + k_j.TQ_c[r]),  p = softmax_j(e),  o_i = sum_j p_ij (v_j + sum_c TV_c[r]),
//   r = r_c(i, j) the table row the pair reads on signal axis c
//   dP_ij = dO_i.(v_j + sum_c TV_c[r]);  delta_i = sum_j p_ij dP_ij;  dS_ij = p_ij (dP_ij - delta_i)
//   dq_i = sum_j dS_ij (k_j + sum_c TK_c[r])      dk_j = sum_i dS_ij (q_i + sum_c TQ_c[r])      dv_j = sum_i p_ij dO_i
//   dTK_c[r] += dS_ij q_i      dTQ_c[r] += dS_ij k_j      dTV_c[r] += p_ij dO_i        (over the pairs that read r)
// Same mapping as the forward kernel: workgroup = (window, head), a query is served by G lanes across the keys, three
// sweeps over the keys per query (logits, dP and delta, gradients).  dq_i is written by the one group that owns the
// query; dk / dv of the window's tokens accumulate in LDS (atomics across the groups) and are written once; the table
// gradients are fp32 atomics on global memory - 3 x D per pair and signal axis, the price of a scatter whose target
// is decided by data (the order of those additions, and with it the last bits of the table gradients, varies from run
// to run).
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int SWB_MAX_AXES = 9;
constexpr int SWB_WAVES = 4;

struct SwinBwdTables {
  long long start[SWB_MAX_AXES];
  int rows[SWB_MAX_AXES];
};

template <int D>
__device__ __forceinline__ void ldrow(const float* __restrict__ p, float* r) {
#pragma unroll
  for (int d = 0; d < D; d += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + d);
    r[d] = t[0]; r[d + 1] = t[1]; r[d + 2] = t[2]; r[d + 3] = t[3];
  }
}
template <int D>
__device__ __forceinline__ float dotd(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) s = fmaf(a[d], b[d], s);
  return s;
}
__device__ __forceinline__ float gmax(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float gsum(float v, int G) {
  for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T, int D, int S>
__global__ __launch_bounds__(SWB_WAVES * 64) void swin_attn_bwd_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const T* __restrict__ dout,
    const float* __restrict__ qt, const float* __restrict__ kt, const float* __restrict__ vt, SwinBwdTables tab,
    const long long* __restrict__ n2n, const int* __restrict__ w_start, const float* __restrict__ crse,
    T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv, float* __restrict__ dqt, float* __restrict__ dkt,
    float* __restrict__ dvt, int heads, int max_tokens) {
  constexpr int RS = D + 4;
  constexpr int NT = SWB_WAVES * 64;
  extern __shared__ __align__(16) unsigned char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + (size_t)max_tokens * RS;
  float* sdK = sV + (size_t)max_tokens * RS;      // [max_tokens][D] accumulated over the queries of the window
  float* sdV = sdK + (size_t)max_tokens * D;
  float* sC = sdV + (size_t)max_tokens * D;       // [max_tokens][S]
  const int lcap = max_tokens > 64 ? max_tokens : 64;
  float* sP = sC + (size_t)max_tokens * S;        // [SWB_WAVES][lcap] logits, then weights, of the wave's queries
  float* sDP = sP + (size_t)SWB_WAVES * lcap;     // [SWB_WAVES][lcap] dP
  int* sRow = reinterpret_cast<int*>(sDP + (size_t)SWB_WAVES * lcap);

  const int w = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t hoff = (size_t)h * D, rstride = (size_t)heads * D;
  const int s0 = w_start[w];
  int m = w_start[w + 1] - s0;
  if (m > max_tokens) m = max_tokens;
  for (int t = tid; t < m; t += NT) sRow[t] = (int)n2n[s0 + t];
  for (int e = tid; e < m * S; e += NT) sC[e] = crse[(size_t)s0 * S + e];
  for (int e = tid; e < m * D; e += NT) { sdK[e] = 0.f; sdV[e] = 0.f; }
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

  int G = D > 16 ? D : 16;
  while (G < m && G < 64) G <<= 1;
  const int qpw = 64 / G;
  const int gl = lane & (G - 1), qs = lane / G;
  float* P = sP + (size_t)wave * lcap + qs * m;
  float* DP = sDP + (size_t)wave * lcap + qs * m;
  // table row of pair (this group's query, key j) on axis c, as an element offset into each table
  auto trow = [&](const float* ci, int j, int c) -> size_t {
    int idx = (int)floorf((ci[c] - sC[j * S + c]) + (float)(tab.rows[c] >> 1));
    idx = min(max(idx, 0), tab.rows[c] - 1);
    return (size_t)tab.start[c] + (size_t)idx * rstride + hoff;
  };
  for (int base = wave * qpw; base < m; base += SWB_WAVES * qpw) {
    const int i = base + qs;
    const bool live = i < m;
    const int ii = live ? i : m - 1;              // idle groups shadow the last query and write nothing
    float qi[D], doi[D], ci[S];
    {
      const size_t g = ((size_t)sRow[ii] * heads + h) * D;
#pragma unroll
      for (int d = 0; d < D; d += 4) {
        unpack4<T>(*reinterpret_cast<const typename Vec4<T>::type*>(q + g + d), qi + d);
        unpack4<T>(*reinterpret_cast<const typename Vec4<T>::type*>(dout + g + d), doi + d);
      }
#pragma unroll
      for (int c = 0; c < S; ++c) ci[c] = sC[ii * S + c];
    }
    // sweep 1: logits
    float mx = -INFINITY;
    for (int j = gl; j < m; j += G) {
      float kj[D];
      ldrow<D>(sK + j * RS, kj);
      float e = dotd<D>(qi, kj);
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const size_t r = trow(ci, j, c);
        float tk[D], tq[D];
        ldrow<D>(kt + r, tk);
        ldrow<D>(qt + r, tq);
        e += dotd<D>(qi, tk) + dotd<D>(kj, tq);
      }
      P[j] = e;
      mx = fmaxf(mx, e);
    }
    mx = gmax(mx, G);
    float den = 0.f;
    for (int j = gl; j < m; j += G) den += __expf(P[j] - mx);
    den = gsum(den, G);
    const float inv = 1.f / den;
    // sweep 2: weights, dP, delta
    float delta = 0.f;
    for (int j = gl; j < m; j += G) {
      const float p = __expf(P[j] - mx) * inv;
      float wj[D];
      ldrow<D>(sV + j * RS, wj);
#pragma unroll
      for (int c = 0; c < S; ++c) {
        float tv[D];
        ldrow<D>(vt + trow(ci, j, c), tv);
#pragma unroll
        for (int d = 0; d < D; ++d) wj[d] += tv[d];
      }
      const float dp = dotd<D>(doi, wj);
      P[j] = p;
      DP[j] = dp;
      delta = fmaf(p, dp, delta);
    }
    delta = gsum(delta, G);
    // sweep 3: gradients.  The table gradients are scatter-adds of D-vectors to rows chosen by data.  Issued by the
    // pair's own lane (D atomics to one row, every lane a different row) each wave instruction touched 64 cache lines
    // for 256 useful bytes; instead the lanes of a query group take the pairs of the current sweep step in turn, 16
    // lanes per pair, lane ch adding channel ch (and ch + 16): one 64-byte piece of one row per 16 lanes.
    float dqa[D];
#pragma unroll
    for (int d = 0; d < D; ++d) dqa[d] = 0.f;
    const int ch = gl & 15, sub = gl >> 4, nsub = G >> 4;
    float qmine[(D + 15) / 16], domine[(D + 15) / 16];
#pragma unroll
    for (int u = 0; u < (D + 15) / 16; ++u) {
      qmine[u] = 0.f; domine[u] = 0.f;
#pragma unroll
      for (int d = 0; d < D; ++d)
        if (d == ch + 16 * u) { qmine[u] = qi[d]; domine[u] = doi[d]; }
    }
    const int lane0 = lane & ~(G - 1);             // first lane of this query group
    for (int jb = 0; jb < m; jb += G) {            // group-uniform sweep step: lane gl holds pair (i, jb + gl)
      const int j = jb + gl;
      const bool mine_ok = live && j < m;
      const int jj = j < m ? j : m - 1;
      const float p = mine_ok ? P[jj] : 0.f;
      const float ds = mine_ok ? p * (DP[jj] - delta) : 0.f;
      float kj[D], dkj[D];
      ldrow<D>(sK + jj * RS, kj);
#pragma unroll
      for (int d = 0; d < D; ++d) { dqa[d] = fmaf(ds, kj[d], dqa[d]); dkj[d] = ds * qi[d]; }
#pragma unroll
      for (int c = 0; c < S; ++c) {
        const size_t r = trow(ci, jj, c);
        float tk[D], tq[D];
        ldrow<D>(kt + r, tk);
        ldrow<D>(qt + r, tq);
#pragma unroll
        for (int d = 0; d < D; ++d) { dqa[d] = fmaf(ds, tk[d], dqa[d]); dkj[d] = fmaf(ds, tq[d], dkj[d]); }
        // cooperative scatter of this step's pairs: sub-group `sub` serves source lanes sub, sub + nsub, ...
        const unsigned roff = (unsigned)r;
        for (int s0 = 0; s0 < G; s0 += nsub) {
          const int src = s0 + sub;
          const unsigned rs = __shfl(roff, lane0 + src);
          const float dss = __shfl(ds, lane0 + src);
          const float ps = __shfl(p, lane0 + src);
          const int js = jb + src;
          if (js < m && (dss != 0.f || ps != 0.f)) {
#pragma unroll
            for (int u = 0; u < (D + 15) / 16; ++u) {
              const int d = ch + 16 * u;
              if (d < D) {
                atomicAdd(dkt + rs + d, dss * qmine[u]);
                atomicAdd(dqt + rs + d, dss * sK[js * RS + d]);
                atomicAdd(dvt + rs + d, ps * domine[u]);
              }
            }
          }
        }
      }
      if (mine_ok) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          atomicAdd(sdK + j * D + d, dkj[d]);
          atomicAdd(sdV + j * D + d, p * doi[d]);
        }
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const float t = gsum(dqa[d], G);
      if (gl == d) mine = t;
    }
    if (live && gl < D) dq[((size_t)sRow[i] * heads + h) * D + gl] = from_f32<T>(mine);
  }
  __syncthreads();
  for (int e = tid; e < m * D; e += NT) {
    const int t = e / D, d = e % D;
    const size_t g = ((size_t)sRow[t] * heads + h) * D + d;
    dk[g] = from_f32<T>(sdK[e]);
    dv[g] = from_f32<T>(sdV[e]);
  }
}

template <typename T, int D, int S>
static int launch_bwd(const void* q, const void* k, const void* v, const void* dout, const float* qt, const float* kt,
                      const float* vt, const SwinBwdTables& tab, const long long* n2n, const int* w_start, int nwin,
                      const float* crse, void* dq, void* dk, void* dv, float* dqt, float* dkt, float* dvt, int heads,
                      int max_tokens, hipStream_t s) {
  const int lcap = max_tokens > 64 ? max_tokens : 64;
  const size_t lds = ((size_t)max_tokens * (2 * (D + 4) + 2 * D + S) + (size_t)2 * SWB_WAVES * lcap) * 4 +
                     (size_t)max_tokens * 4;
  if (lds > 160 * 1024) {
    set_error("swin_attn_bwd: %d tokens x head_dim %d needs %zu bytes of LDS", max_tokens, D, lds);
    return PTV3_ERR_UNSUPPORTED;
  }
  ensure_dynamic_lds(reinterpret_cast<const void*>(&swin_attn_bwd_kernel<T, D, S>), 160 * 1024);
  hipLaunchKernelGGL((swin_attn_bwd_kernel<T, D, S>), dim3((unsigned)nwin, (unsigned)heads), dim3(SWB_WAVES * 64), lds,
                     s, (const T*)q, (const T*)k, (const T*)v, (const T*)dout, qt, kt, vt, tab, n2n, w_start, crse,
                     (T*)dq, (T*)dk, (T*)dv, dqt, dkt, dvt, heads, max_tokens);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

template <typename T, int D, typename... A>
static int bwd_axes(int S, A... a) {
  switch (S) {
    case 3: return launch_bwd<T, D, 3>(a...);
    case 6: return launch_bwd<T, D, 6>(a...);
    case 9: return launch_bwd<T, D, 9>(a...);
  }
  set_error("swin_attn_bwd: %d signal axes (3, 6 or 9)", S);
  return PTV3_ERR_UNSUPPORTED;
}

template <typename T, typename... A>
static int bwd_dim(int D, int S, A... a) {
  switch (D) {
    case 8: return bwd_axes<T, 8>(S, a...);
    case 16: return bwd_axes<T, 16>(S, a...);
    case 32: return bwd_axes<T, 32>(S, a...);
  }
  set_error("swin_attn_bwd: head_dim %d (8, 16 or 32)", D);
  return PTV3_ERR_UNSUPPORTED;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_swin_attn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* q_table,
                                  const float* k_table, const float* v_table, const int32_t* table_offsets_host,
                                  int num_axes, const int64_t* n2n, const int32_t* w_start, int num_windows,
                                  const float* n_crse, void* dq, void* dk, void* dv, float* dq_table, float* dk_table,
                                  float* dv_table, int64_t n, int heads, int head_dim, int max_tokens, int dtype,
                                  void* stream) {
  PTV3_REQUIRE(num_axes > 0 && num_axes <= SWB_MAX_AXES, "swin_attn_bwd: %d signal axes", num_axes);
  PTV3_REQUIRE(heads > 0 && head_dim > 0, "swin_attn_bwd: heads %d head_dim %d", heads, head_dim);
  PTV3_REQUIRE(max_tokens > 0 && max_tokens <= 512, "swin_attn_bwd: max_tokens %d (1..512)", max_tokens);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "swin_attn_bwd: dtype %d", dtype);
  PTV3_REQUIRE(n < (1ll << 31), "swin_attn_bwd: %lld voxels", (long long)n);
  SwinBwdTables tab;
  long long at = 0;
  const int per = heads * head_dim;
  for (int c = 0; c < num_axes; ++c) {
    PTV3_REQUIRE(table_offsets_host[c] > 0 && table_offsets_host[c] % (2 * per) == 0,
                 "swin_attn_bwd: table_offsets[%d] = %d is not an even number of (heads x head_dim) rows", c,
                 table_offsets_host[c]);
    tab.start[c] = at;
    tab.rows[c] = table_offsets_host[c] / per;
    at += table_offsets_host[c];
  }
  for (int c = num_axes; c < SWB_MAX_AXES; ++c) { tab.start[c] = 0; tab.rows[c] = 2; }
  hipStream_t s = (hipStream_t)stream;
  // table gradients accumulate by atomics: start from zero
  for (float* t : {dq_table, dk_table, dv_table})
    if (hipMemsetAsync(t, 0, (size_t)at * sizeof(float), s) != hipSuccess) {
      set_error("swin_attn_bwd: memset failed");
      return PTV3_ERR_LAUNCH;
    }
  if (num_windows <= 0 || n <= 0) return PTV3_OK;
  const long long* nn = (const long long*)n2n;
  if (dtype == PTV3_F32)
    return bwd_dim<float>(head_dim, num_axes, q, k, v, dout, q_table, k_table, v_table, tab, nn, w_start, num_windows,
                          n_crse, dq, dk, dv, dq_table, dk_table, dv_table, heads, max_tokens, s);
  return bwd_dim<__bf16>(head_dim, num_axes, q, k, v, dout, q_table, k_table, v_table, tab, nn, w_start, num_windows,
                         n_crse, dq, dk, dv, dq_table, dk_table, dv_table, heads, max_tokens, s);
}
