// Backward of the serialized window attention (SerializedAttention.forward, point_transformer_v3m1_base.py:
// 172-220, vanilla branch) for training.  No reference backward exists in the tree (autograd
// differentiates :196-204); this is the standard recompute form:
//   P = softmax(scale * Q K^T),  delta_q = sum_d dO[q][d] * O[q][d]
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - delta),  dQ = scale * dS K,  dK = scale * dS^T Q
// Two kernels, no atomics:
//   pass A (query-stationary): recomputes the row statistics (log-sum-exp, log2 domain), then dQ;
//   pass B (key-stationary):   dK and dV, reading the statistics pass A stored.
// Same transposed-product trick as the forward kernel: the 16x16 score tile lands in the accumulator with the
// stationary index on the lane, which is exactly the B-operand layout of the next product (dS / P never
// touch LDS).  The gather through win_order is fused into the loads; gradients are produced per padded slot
// and folded back to points afterwards (a borrowed point sits in two slots: kept slot + one duplicate).
#include <algorithm>
#include "common.h"
#include "profile.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

struct AttnBwdArgs {
  const void* qkv; const void* out; const void* dout;
  const int32_t* win_order; const int32_t* win_inverse;
  void* dqkv_pad; float* lse; float* delta;
  int c, heads, patch, nwin;
  float scale, scale_log2e;
  const int32_t* cu;  // ragged windows (ptv3_window_attn_varlen_bwd): slots [cu[w], cu[w+1]) of window w, else NULL
  // relative-position bias (RPE, point_transformer_v3m1_base.py:29-48): score += sum_axis table[axis][clamp(dgrid)][head].
  // grid (n, 3) int32 voxel coordinates in point order, table (3 * rpe_num, heads) fp32; dtab_slab receives one partial
  // table-column gradient [3 * rpe_num] per workgroup of pass A (summed per head in block order afterwards)
  const int32_t* grid; const float* table; int pos_bnd; float* dtab_slab;
  // attention dropout (common.h drop_keep): 0 = none; else the forward's threshold, seed and 1 / (1 - p)
  unsigned drop_thr, drop_seed; float drop_scale;
  const float* lse_in;   // the forward's log2-domain log-sum-exp per (slot, head), or NULL: pass A recomputes it
};

constexpr int AB_MAX_TAB = 1024;  // 3 * (2 * pos_bnd + 1) entries of one head's table column held in LDS

// bias of the pair (query coordinates qg, key coordinates kg) from one head's table column (already in the caller's
// unit: natural for the gradient bins, x log2(e) for the softmax recompute)
__device__ __forceinline__ float rpe_pair_bias(const float* __restrict__ tab, const int* qg, const int* kg, int bnd,
                                               int rpe_num) {
  float b = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d) b += tab[d * rpe_num + min(max(qg[d] - kg[d], -bnd), bnd) + bnd];
  return b;
}

template <int ND> struct AbCfg {
  static constexpr int D = 16 * ND;
  static constexpr int CT = 2048 / D;  // rows of the streamed side per LDS chunk (128 / 64 / 32): pass B fits 64 KB in fp32
  static constexpr int TS = CT + 4;    // row stride of the transposed copies (bank spread)
};

// -------------------------------------------------------------------------------------------------
// pass A: lane = query.  Streams K, V (and K^T) of the window through LDS.
// -------------------------------------------------------------------------------------------------
template <typename T, int ND, bool RPE>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(AttnBwdArgs a) {
  typedef typename Vec4<T>::type V4;
  constexpr int D = AbCfg<ND>::D, CT = AbCfg<ND>::CT, TS = AbCfg<ND>::TS;
  __shared__ __attribute__((aligned(16))) T sK[CT * D];
  __shared__ __attribute__((aligned(16))) T sV[CT * D];
  __shared__ __attribute__((aligned(16))) T sKt[D * TS];
  __shared__ int sG[RPE ? CT * 3 : 1];              // voxel coordinates of the streamed keys
  __shared__ float sTab[RPE ? AB_MAX_TAB : 1];      // this head's table column x log2(e)
  __shared__ float sHist[RPE ? 4 * AB_MAX_TAB : 1]; // per-wave gradient bins of the column
  const int rpe_num = 2 * a.pos_bnd + 1;
  const T* __restrict__ qkv = reinterpret_cast<const T*>(a.qkv);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int qblocks = (a.patch + 63) / 64;
  const int qb = blockIdx.x % qblocks;
  const int h = (blockIdx.x / qblocks) % a.heads;
  const int w = blockIdx.x / (qblocks * a.heads);
  const int c3 = 3 * a.c;
  const int64_t slot0 = a.cu ? (int64_t)a.cu[w] : (int64_t)w * a.patch;
  const int P = a.cu ? a.cu[w + 1] - a.cu[w] : a.patch;  // slots of this window
  if (qb * 64 >= P) return;                               // workgroup-uniform
  const int jq = qb * 64 + wave * 16 + i;
  const bool qv = jq < P;
  const int64_t pq = slot0 + (qv ? jq : P - 1);
  const int64_t src = a.win_order[pq];
  const bool kept = qv && a.win_inverse[src] == pq;
  // stationary fragments: Q^T and dO^T as B operands (lane (i, g): dims 16nd+4g..+3 of query i)
  V4 qf[ND], dof[ND];
  float dpart = 0.f;
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) {
    qf[nd] = *reinterpret_cast<const V4*>(qkv + src * c3 + h * D + 16 * nd + 4 * g);
    dof[nd] = zero4<T>();
    if (kept) {
      dof[nd] = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.dout) + src * a.c + h * D + 16 * nd + 4 * g);
      float dv[4], ov[4];
      unpack4<T>(dof[nd], dv);
      unpack4<T>(*reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.out) + src * a.c + h * D + 16 * nd + 4 * g), ov);
      dpart += dv[0] * ov[0] + dv[1] * ov[1] + dv[2] * ov[2] + dv[3] * ov[3];
    }
  }
  dpart += __shfl_xor(dpart, 16, 64);
  const float delta = dpart + __shfl_xor(dpart, 32, 64);
  int qg[3] = {0, 0, 0};
  if constexpr (RPE) {
#pragma unroll
    for (int d = 0; d < 3; ++d) qg[d] = a.grid[src * 3 + d];
    for (int j = threadIdx.x; j < 3 * rpe_num; j += 256) sTab[j] = a.table[(int64_t)j * a.heads + h] * 1.44269504088896340736f;
    for (int j = threadIdx.x; j < 4 * 3 * rpe_num; j += 256) sHist[j] = 0.f;
  }

  const int nchunks = (P + CT - 1) / CT;
  auto load_chunk = [&](int kc0, bool with_v) {
    __syncthreads();
    if constexpr (RPE) {
      for (int u = threadIdx.x; u < CT; u += 256) {
        const bool kv_ = kc0 + u < P;
        const int64_t ks = kv_ ? a.win_order[slot0 + kc0 + u] : 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) sG[3 * u + d] = kv_ ? a.grid[ks * 3 + d] : 0;
      }
    }
    for (int u = threadIdx.x; u < CT * (D / 4); u += 256) {
      const int key = u / (D / 4), dv = u % (D / 4);
      V4 kv = zero4<T>(), vv = zero4<T>();
      if (kc0 + key < P) {
        const int64_t ks = a.win_order[slot0 + kc0 + key];
        kv = *reinterpret_cast<const V4*>(qkv + ks * c3 + a.c + h * D + 4 * dv);
        if (with_v) vv = *reinterpret_cast<const V4*>(qkv + ks * c3 + 2 * a.c + h * D + 4 * dv);
      }
      *reinterpret_cast<V4*>(sK + key * D + 4 * dv) = kv;
      if (with_v) {
        *reinterpret_cast<V4*>(sV + key * D + 4 * dv) = vv;
        const T* ke = reinterpret_cast<const T*>(&kv);
#pragma unroll
        for (int e = 0; e < 4; ++e) sKt[(4 * dv + e) * TS + key] = ke[e];
      }
    }
    __syncthreads();
  };
  auto scores = [&](int kt) {
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
      s = mma16<T>(*reinterpret_cast<const V4*>(sK + (16 * kt + i) * D + 16 * nd + 4 * g), qf[nd], s);
    return s;  // s[r] = q_i . k_(16kt+4g+r)
  };

  // ---- sweep 1: log-sum-exp of the scaled scores (log2 domain) - unless the training forward left it
  float m2 = -INFINITY, l = 0.f;
  const bool have_lse = !RPE && a.lse_in != nullptr;
  for (int ch = 0; ch < (have_lse ? 0 : nchunks); ++ch) {
    const int kc0 = ch * CT;
    load_chunk(kc0, false);
    const int ntile = (min(CT, P - kc0) + 15) / 16;
    for (int kt = 0; kt < ntile; ++kt) {
      const f32x4 s = scores(kt);
      float t[4], mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t[r] = kc0 + 16 * kt + 4 * g + r < P ? s[r] * a.scale_log2e : -INFINITY;
        if constexpr (RPE) t[r] += rpe_pair_bias(sTab, qg, sG + 3 * (16 * kt + 4 * g + r), a.pos_bnd, rpe_num);
        mx = fmaxf(mx, t[r]);
      }
      if (mx == -INFINITY) continue;
      const float mn = fmaxf(m2, mx);
      l = l * __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
      for (int r = 0; r < 4; ++r) l += __builtin_amdgcn_exp2f(t[r] - mn);
      m2 = mn;
    }
  }
#pragma unroll
  for (int d = 16; d < 64; d <<= 1) {
    const float mo = __shfl_xor(m2, d, 64), lo = __shfl_xor(l, d, 64);
    const float mn = fmaxf(m2, mo);
    if (mn != -INFINITY) l = l * __builtin_amdgcn_exp2f(m2 - mn) + lo * __builtin_amdgcn_exp2f(mo - mn);
    m2 = mn;
  }
  const float lse2 = have_lse ? a.lse_in[pq * a.heads + h] : m2 + __log2f(l);

  // ---- sweep 2: dQ^T[d][q] += K^T[d][key] * dS^T[key][q]
  f32x4 acc[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) acc[nd] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < nchunks; ++ch) {
    const int kc0 = ch * CT;
    load_chunk(kc0, true);
    const int ntile = (min(CT, P - kc0) + 15) / 16;
    for (int kt = 0; kt < ntile; ++kt) {
      const f32x4 s = scores(kt);
      f32x4 dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
        dp = mma16<T>(*reinterpret_cast<const V4*>(sV + (16 * kt + i) * D + 16 * nd + 4 * g), dof[nd], dp);
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool kin = kc0 + 16 * kt + 4 * g + r < P;
        float sc = s[r] * a.scale_log2e;
        const int* kg = sG + (RPE ? 3 * (16 * kt + 4 * g + r) : 0);
        if constexpr (RPE) sc += rpe_pair_bias(sTab, qg, kg, a.pos_bnd, rpe_num);
        const float p = kin ? __builtin_amdgcn_exp2f(sc - lse2) : 0.f;
        float dpe = dp[r];
        if (a.drop_thr)   // d out / d P of a dropped pair is zero, of a kept one v / (1 - p); delta = dO . O already has it
          dpe = drop_keep(((unsigned long long)(pq * a.heads + h) << 14) | (unsigned)(kc0 + 16 * kt + 4 * g + r), a.drop_seed,
                          a.drop_thr) ? dpe * a.drop_scale : 0.f;
        ds[r] = p * (dpe - delta);
        if constexpr (RPE) {
          // d loss / d bias(q, k) = dS: one bin per axis of this wave's copy of the column gradient
          if (kin && qv) {
#pragma unroll
            for (int d = 0; d < 3; ++d)
              atomicAdd(&sHist[wave * 3 * rpe_num + d * rpe_num + min(max(qg[d] - kg[d], -a.pos_bnd), a.pos_bnd) + a.pos_bnd],
                        ds[r]);
          }
        }
      }
      const V4 dsv = pack4<T>(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
        acc[nd] = mma16<T>(*reinterpret_cast<const V4*>(sKt + (16 * nd + i) * TS + 16 * kt + 4 * g), dsv, acc[nd]);
    }
  }
  if (qv) {
    const int64_t p = slot0 + jq;
    T* dq = reinterpret_cast<T*>(a.dqkv_pad) + p * c3 + h * D;
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
      *reinterpret_cast<V4*>(dq + 16 * nd + 4 * g) =
          pack4<T>(acc[nd][0] * a.scale, acc[nd][1] * a.scale, acc[nd][2] * a.scale, acc[nd][3] * a.scale);
    if (g == 0) {
      a.lse[p * a.heads + h] = lse2;
      a.delta[p * a.heads + h] = delta;
    }
  }
  if constexpr (RPE) {
    __syncthreads();
    for (int j = threadIdx.x; j < 3 * rpe_num; j += 256)   // the four wave copies in wave order
      a.dtab_slab[(int64_t)blockIdx.x * 3 * rpe_num + j] =
          ((sHist[j] + sHist[3 * rpe_num + j]) + sHist[2 * 3 * rpe_num + j]) + sHist[3 * 3 * rpe_num + j];
  }
}

// -------------------------------------------------------------------------------------------------
// pass B: lane = key.  Streams Q, dO (and their transposes), lse, delta of the window through LDS.
// -------------------------------------------------------------------------------------------------
template <typename T, int ND, bool RPE>
__global__ void __launch_bounds__(256) attn_bwd_dkv_kernel(AttnBwdArgs a) {
  typedef typename Vec4<T>::type V4;
  constexpr int D = AbCfg<ND>::D, CT = AbCfg<ND>::CT, TS = AbCfg<ND>::TS;
  __shared__ int sG[RPE ? CT * 3 : 1];          // voxel coordinates of the streamed queries
  __shared__ float sTab[RPE ? AB_MAX_TAB : 1];  // this head's table column x log2(e)
  const int rpe_num = 2 * a.pos_bnd + 1;
  __shared__ __attribute__((aligned(16))) T sQ[CT * D];
  __shared__ __attribute__((aligned(16))) T sO[CT * D];
  __shared__ __attribute__((aligned(16))) T sQt[D * TS];
  __shared__ __attribute__((aligned(16))) T sOt[D * TS];
  __shared__ float sL[CT], sDl[CT];
  const T* __restrict__ qkv = reinterpret_cast<const T*>(a.qkv);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int kblocks = (a.patch + 63) / 64;
  const int kb = blockIdx.x % kblocks;
  const int h = (blockIdx.x / kblocks) % a.heads;
  const int w = blockIdx.x / (kblocks * a.heads);
  const int c3 = 3 * a.c;
  const int64_t slot0 = a.cu ? (int64_t)a.cu[w] : (int64_t)w * a.patch;
  const int P = a.cu ? a.cu[w + 1] - a.cu[w] : a.patch;  // slots of this window
  if (kb * 64 >= P) return;                               // workgroup-uniform
  const int jk = kb * 64 + wave * 16 + i;
  const bool kvalid = jk < P;
  const int64_t pk = slot0 + (kvalid ? jk : P - 1);
  const int64_t src = a.win_order[pk];
  V4 kf[ND], vf[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) {
    kf[nd] = *reinterpret_cast<const V4*>(qkv + src * c3 + a.c + h * D + 16 * nd + 4 * g);
    vf[nd] = *reinterpret_cast<const V4*>(qkv + src * c3 + 2 * a.c + h * D + 16 * nd + 4 * g);
  }
  f32x4 acck[ND], accv[ND];
#pragma unroll
  for (int nd = 0; nd < ND; ++nd) acck[nd] = accv[nd] = f32x4{0.f, 0.f, 0.f, 0.f};
  int kg[3] = {0, 0, 0};
  if constexpr (RPE) {
#pragma unroll
    for (int d = 0; d < 3; ++d) kg[d] = a.grid[src * 3 + d];
    for (int j = threadIdx.x; j < 3 * rpe_num; j += 256) sTab[j] = a.table[(int64_t)j * a.heads + h] * 1.44269504088896340736f;
  }

  const int nchunks = (P + CT - 1) / CT;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int qc0 = ch * CT;
    __syncthreads();
    for (int u = threadIdx.x; u < CT * (D / 4); u += 256) {
      const int q = u / (D / 4), dv = u % (D / 4);
      V4 qv4 = zero4<T>(), ov4 = zero4<T>();
      if (qc0 + q < P) {
        const int64_t p = slot0 + qc0 + q;
        const int64_t qs = a.win_order[p];
        qv4 = *reinterpret_cast<const V4*>(qkv + qs * c3 + h * D + 4 * dv);
        if (a.win_inverse[qs] == p)
          ov4 = *reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.dout) + qs * a.c + h * D + 4 * dv);
      }
      *reinterpret_cast<V4*>(sQ + q * D + 4 * dv) = qv4;
      *reinterpret_cast<V4*>(sO + q * D + 4 * dv) = ov4;
      const T* qe = reinterpret_cast<const T*>(&qv4);
      const T* oe = reinterpret_cast<const T*>(&ov4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sQt[(4 * dv + e) * TS + q] = qe[e];
        sOt[(4 * dv + e) * TS + q] = oe[e];
      }
    }
    for (int q = threadIdx.x; q < CT; q += 256) {
      const bool v = qc0 + q < P;
      sL[q] = v ? a.lse[(slot0 + qc0 + q) * a.heads + h] : INFINITY;  // exp2(-inf) = 0 for the tail rows
      sDl[q] = v ? a.delta[(slot0 + qc0 + q) * a.heads + h] : 0.f;
      if constexpr (RPE) {
        const int64_t qs = v ? a.win_order[slot0 + qc0 + q] : 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) sG[3 * q + d] = v ? a.grid[qs * 3 + d] : 0;
      }
    }
    __syncthreads();
    const int ntile = (min(CT, P - qc0) + 15) / 16;
    for (int qt = 0; qt < ntile; ++qt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        s = mma16<T>(*reinterpret_cast<const V4*>(sQ + (16 * qt + i) * D + 16 * nd + 4 * g), kf[nd], s);
        dp = mma16<T>(*reinterpret_cast<const V4*>(sO + (16 * qt + i) * D + 16 * nd + 4 * g), vf[nd], dp);
      }
      float p[4], ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 16 * qt + 4 * g + r;
        float sc = s[r] * a.scale_log2e;
        if constexpr (RPE) sc += rpe_pair_bias(sTab, sG + 3 * q, kg, a.pos_bnd, rpe_num);
        p[r] = __builtin_amdgcn_exp2f(sc - sL[q]);
        float dpe = dp[r];
        float pd = p[r];
        if (a.drop_thr) {
          const bool kp = drop_keep(((unsigned long long)((slot0 + qc0 + q) * a.heads + h) << 14) | (unsigned)jk, a.drop_seed,
                                    a.drop_thr);
          dpe = kp ? dpe * a.drop_scale : 0.f;
          pd = kp ? pd * a.drop_scale : 0.f;
        }
        ds[r] = p[r] * (dpe - sDl[q]);
        p[r] = pd;      // dV sums the dropped, rescaled weights
      }
      const V4 pv = pack4<T>(p[0], p[1], p[2], p[3]);
      const V4 dsv = pack4<T>(ds[0], ds[1], ds[2], ds[3]);
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        accv[nd] = mma16<T>(*reinterpret_cast<const V4*>(sOt + (16 * nd + i) * TS + 16 * qt + 4 * g), pv, accv[nd]);
        acck[nd] = mma16<T>(*reinterpret_cast<const V4*>(sQt + (16 * nd + i) * TS + 16 * qt + 4 * g), dsv, acck[nd]);
      }
    }
  }
  if (kvalid) {
    T* dk = reinterpret_cast<T*>(a.dqkv_pad) + (slot0 + jk) * c3 + a.c + h * D;
    T* dv = dk + a.c;
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) {
      *reinterpret_cast<V4*>(dk + 16 * nd + 4 * g) =
          pack4<T>(acck[nd][0] * a.scale, acck[nd][1] * a.scale, acck[nd][2] * a.scale, acck[nd][3] * a.scale);
      *reinterpret_cast<V4*>(dv + 16 * nd + 4 * g) = pack4<T>(accv[nd][0], accv[nd][1], accv[nd][2], accv[nd][3]);
    }
  }
}

// second slot of a borrowed point (pad plan: a point appears in at most two slots), -1 otherwise
__global__ void __launch_bounds__(256) dup_slot_kernel(const int32_t* __restrict__ win_order,
                                                        const int32_t* __restrict__ win_inverse, int64_t n_pad,
                                                        int32_t* __restrict__ dup) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pad) return;
  const int32_t src = win_order[p];
  if (win_inverse[src] != p) dup[src] = (int32_t)p;
}

// dqkv[point] = dpad[kept slot] + dpad[duplicate slot]
template <typename T>
__global__ void __launch_bounds__(256) unpad_add_kernel(const T* __restrict__ dpad, const int32_t* __restrict__ win_inverse,
                                                         const int32_t* __restrict__ dup, int64_t n, int c3,
                                                         T* __restrict__ dqkv) {
  typedef typename Vec4<T>::type V4;
  const int v4 = c3 / 4;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * v4) return;
  const int64_t r = t / v4;
  const int e4 = (int)(t % v4);
  float x[4];
  unpack4<T>(*reinterpret_cast<const V4*>(dpad + (int64_t)win_inverse[r] * c3 + 4 * e4), x);
  const int32_t d = dup[r];
  if (d >= 0) {
    float y[4];
    unpack4<T>(*reinterpret_cast<const V4*>(dpad + (int64_t)d * c3 + 4 * e4), y);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] += y[e];
  }
  *reinterpret_cast<V4*>(dqkv + r * c3 + 4 * e4) = pack4<T>(x[0], x[1], x[2], x[3]);
}

template <typename T, int ND>
static void launch_attn_bwd(const AttnBwdArgs& a, hipStream_t s) {
  const unsigned blocks = (unsigned)a.nwin * a.heads * ((a.patch + 63) / 64);
  // recompute form: pass A = S, P, dP, dS and dQ (4 products of len^2 d per head), pass B = S, P, dP, dS, dK, dV (5):
  // 2 len^2 c flops per product and window (uniform windows assumed for the count)
  const double prod = 2.0 * (double)a.nwin * a.patch * (double)a.patch * a.c;
  const double bytes = (double)a.nwin * a.patch * a.c * sizeof(T) * 5.0;
  int prof = prof_begin(s, PROF_BACKWARD, 4.0 * prod, bytes, nullptr, 0, 0.0);
  prof_kernel(prof, PK_ATTN_BWD_DQ);
  if (a.table) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, ND, true>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<T, ND, false>), dim3(blocks), dim3(256), 0, s, a);
  prof_end(prof, s);
  prof = prof_begin(s, PROF_BACKWARD, 5.0 * prod, bytes, nullptr, 0, 0.0);
  prof_kernel(prof, PK_ATTN_BWD_DKV);
  if (a.table) hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, ND, true>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, ND, false>), dim3(blocks), dim3(256), 0, s, a);
  prof_end(prof, s);
}

// dtable[j][h] = sum over the pass-A workgroups of head h (block id = (window * heads + h) * qblocks + qb) of their
// partial columns, in block order: deterministic
__global__ void __launch_bounds__(256) rpe_dtable_reduce_kernel(const float* __restrict__ slab, int nwin, int heads,
                                                                 int qblocks, int ncol, float* __restrict__ dtable) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ncol * heads) return;
  const int j = t / heads, h = t % heads;
  float acc = 0.f;
  for (int w = 0; w < nwin; ++w)
    for (int qb = 0; qb < qblocks; ++qb) acc += slab[((int64_t)(w * heads + h) * qblocks + qb) * ncol + j];
  dtable[t] = acc;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace ptv3

using namespace ptv3;

extern "C" size_t ptv3_window_attn_bwd_workspace_bytes(int64_t n, int64_t n_pad, int c, int heads, int dtype) {
  const size_t esz = dtype == PTV3_F32 ? 4 : 2;
  return align256((size_t)n_pad * 3 * c * esz) + 2 * align256((size_t)n_pad * heads * sizeof(float)) +
         align256((size_t)n * sizeof(int32_t));
}

struct RpeBwd { const int32_t* grid; const float* table; int pos_bnd; float* dtable; };

static int window_attn_bwd_impl(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                                const int32_t* win_inverse, const int32_t* cu, int nwin, void* dqkv, int64_t n,
                                int64_t n_pad, int c, int heads, int patch, float scale, int dtype, void* workspace,
                                hipStream_t s, RpeBwd rpe = RpeBwd{nullptr, nullptr, 0, nullptr}, float p_drop = 0.f,
                                unsigned seed = 0, const float* lse_in = nullptr) {
  const int d = c / heads;
  if (d != 16 && d != 32 && d != 64) {
    set_error("window_attn_bwd: head_dim %d unsupported (16, 32, 64)", d);
    return PTV3_ERR_UNSUPPORTED;
  }
  if (n == 0) return PTV3_OK;
  const size_t esz = dtype == PTV3_F32 ? 4 : 2;
  char* ws = (char*)workspace;
  AttnBwdArgs a;
  a.qkv = qkv; a.out = out; a.dout = dout; a.win_order = win_order; a.win_inverse = win_inverse;
  a.dqkv_pad = ws; ws += align256((size_t)n_pad * 3 * c * esz);
  a.lse = (float*)ws; ws += align256((size_t)n_pad * heads * sizeof(float));
  a.delta = (float*)ws; ws += align256((size_t)n_pad * heads * sizeof(float));
  int32_t* dup = (int32_t*)ws; ws += align256((size_t)n * sizeof(int32_t));
  a.c = c; a.heads = heads; a.patch = patch; a.nwin = nwin; a.cu = cu;
  a.grid = rpe.grid; a.table = rpe.table; a.pos_bnd = rpe.pos_bnd; a.dtab_slab = (float*)ws;
  a.scale = scale; a.scale_log2e = scale * 1.44269504088896340736f;
  a.drop_thr = p_drop > 0.f ? (unsigned)std::min(4294967295.0, (double)p_drop * 4294967296.0) : 0u;
  a.drop_seed = seed; a.drop_scale = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  a.lse_in = lse_in;
  if (hipMemsetAsync(dup, 0xFF, (size_t)n * sizeof(int32_t), s) != hipSuccess) return PTV3_ERR_LAUNCH;
  hipLaunchKernelGGL(dup_slot_kernel, dim3((unsigned)cdiv(n_pad, 256)), dim3(256), 0, s, win_order, win_inverse, n_pad, dup);
#define AB_CASE(T)                                     \
  switch (d) {                                         \
    case 16: launch_attn_bwd<T, 1>(a, s); break;       \
    case 32: launch_attn_bwd<T, 2>(a, s); break;       \
    default: launch_attn_bwd<T, 4>(a, s); break;       \
  }
  if (dtype == PTV3_F32) { AB_CASE(float) } else { AB_CASE(__bf16) }
#undef AB_CASE
  const int64_t tot = n * (3 * c / 4);
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL(unpad_add_kernel<float>, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, s, (const float*)a.dqkv_pad,
                       win_inverse, dup, n, 3 * c, (float*)dqkv);
  else
    hipLaunchKernelGGL(unpad_add_kernel<__bf16>, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, s,
                       (const __bf16*)a.dqkv_pad, win_inverse, dup, n, 3 * c, (__bf16*)dqkv);
  if (rpe.table) {
    const int ncol = 3 * (2 * rpe.pos_bnd + 1);
    hipLaunchKernelGGL(rpe_dtable_reduce_kernel, dim3((unsigned)cdiv((int64_t)ncol * heads, 256)), dim3(256), 0, s,
                       a.dtab_slab, nwin, heads, (patch + 63) / 64, ncol, rpe.dtable);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" size_t ptv3_window_attn_rpe_bwd_workspace_bytes(int64_t n, int64_t n_pad, int c, int heads, int patch,
                                                           int pos_bnd, int dtype) {
  const int64_t blocks = (patch > 0 ? n_pad / patch : 0) * heads * ((patch + 63) / 64);
  return ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, heads, dtype) +
         align256((size_t)blocks * 3 * (2 * pos_bnd + 1) * sizeof(float));
}

extern "C" int ptv3_window_attn_rpe_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                                        const int32_t* win_inverse, const int32_t* grid_coord, const float* rpe_table,
                                        int pos_bnd, void* dqkv, float* dtable, int64_t n, int64_t n_pad, int c,
                                        int heads, int patch, float scale, int dtype, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_rpe_bwd: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_rpe_bwd: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(n_pad % patch == 0, "window_attn_rpe_bwd: n_pad=%lld is not a multiple of patch=%d", (long long)n_pad, patch);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_rpe_bwd: bad dtype %d", dtype);
  PTV3_REQUIRE(grid_coord && rpe_table && dtable && pos_bnd >= 0, "window_attn_rpe_bwd: grid_coord, rpe_table, dtable required");
  if (3 * (2 * pos_bnd + 1) > AB_MAX_TAB) {
    set_error("window_attn_rpe_bwd: table column of %d entries exceeds %d", 3 * (2 * pos_bnd + 1), AB_MAX_TAB);
    return PTV3_ERR_UNSUPPORTED;
  }
  PTV3_REQUIRE(workspace_bytes >= ptv3_window_attn_rpe_bwd_workspace_bytes(n, n_pad, c, heads, patch, pos_bnd, dtype),
               "window_attn_rpe_bwd: workspace too small");
  if (n == 0) {
    if (hipMemsetAsync(dtable, 0, (size_t)3 * (2 * pos_bnd + 1) * heads * sizeof(float), (hipStream_t)stream) != hipSuccess)
      return PTV3_ERR_LAUNCH;
    return PTV3_OK;
  }
  return window_attn_bwd_impl(qkv, out, dout, win_order, win_inverse, nullptr, (int)(n_pad / patch), dqkv, n, n_pad, c,
                              heads, patch, scale, dtype, workspace, (hipStream_t)stream,
                              RpeBwd{grid_coord, rpe_table, pos_bnd, dtable});
}

extern "C" int ptv3_window_attn_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                                    const int32_t* win_inverse, void* dqkv, int64_t n, int64_t n_pad, int c, int heads,
                                    int patch, float scale, int dtype, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_bwd: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_bwd: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(n_pad % patch == 0, "window_attn_bwd: n_pad=%lld is not a multiple of patch=%d", (long long)n_pad, patch);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_bwd: bad dtype %d", dtype);
  PTV3_REQUIRE(workspace_bytes >= ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, heads, dtype),
               "window_attn_bwd: workspace too small");
  return window_attn_bwd_impl(qkv, out, dout, win_order, win_inverse, nullptr, (int)(n_pad / patch), dqkv, n, n_pad, c,
                              heads, patch, scale, dtype, workspace, (hipStream_t)stream);
}

extern "C" int ptv3_window_attn_varlen_bwd(const void* qkv, const void* out, const void* dout,
                                           const int32_t* win_order, const int32_t* win_inverse,
                                           const int32_t* cu_seqlens, int num_windows, void* dqkv, int64_t n,
                                           int64_t n_pad, int c, int heads, int max_seqlen, float scale, int dtype,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_varlen_bwd: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(max_seqlen >= 1 && max_seqlen <= 16384, "window_attn_varlen_bwd: max_seqlen %d outside [1,16384]", max_seqlen);
  PTV3_REQUIRE(cu_seqlens != nullptr && num_windows >= 1, "window_attn_varlen_bwd: cu_seqlens with >= 1 window required");
  PTV3_REQUIRE(n_pad <= (int64_t)num_windows * max_seqlen && n_pad >= num_windows,
               "window_attn_varlen_bwd: n_pad=%lld cannot be split into %d windows of 1..%d slots", (long long)n_pad,
               num_windows, max_seqlen);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_varlen_bwd: bad dtype %d", dtype);
  PTV3_REQUIRE(workspace_bytes >= ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, heads, dtype),
               "window_attn_varlen_bwd: workspace too small");
  return window_attn_bwd_impl(qkv, out, dout, win_order, win_inverse, cu_seqlens, num_windows, dqkv, n, n_pad, c, heads,
                              max_seqlen, scale, dtype, workspace, (hipStream_t)stream);
}

extern "C" int ptv3_window_attn_drop_bwd(const void* qkv, const void* out, const void* dout, const int32_t* win_order,
                                         const int32_t* win_inverse, const int32_t* cu_seqlens, int num_windows, void* dqkv,
                                         int64_t n, int64_t n_pad, int c, int heads, int patch, float scale, float p_drop,
                                         uint32_t seed, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_drop_bwd: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_drop_bwd: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(cu_seqlens != nullptr || n_pad % patch == 0, "window_attn_drop_bwd: n_pad=%lld is not a multiple of patch=%d",
               (long long)n_pad, patch);
  PTV3_REQUIRE(p_drop > 0.f && p_drop < 1.f, "window_attn_drop_bwd: p_drop %g outside (0,1)", (double)p_drop);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_drop_bwd: bad dtype %d", dtype);
  PTV3_REQUIRE(workspace_bytes >= ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, heads, dtype),
               "window_attn_drop_bwd: workspace too small");
  const int nwin = cu_seqlens ? num_windows : (int)(n_pad / patch);
  return window_attn_bwd_impl(qkv, out, dout, win_order, win_inverse, cu_seqlens, nwin, dqkv, n, n_pad, c, heads, patch,
                              scale, dtype, workspace, (hipStream_t)stream, RpeBwd{nullptr, nullptr, 0, nullptr}, p_drop, seed);
}

extern "C" int ptv3_window_attn_train_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                                          const int32_t* win_order, const int32_t* win_inverse, const int32_t* cu_seqlens,
                                          int num_windows, void* dqkv, int64_t n, int64_t n_pad, int c, int heads, int patch,
                                          float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_train_bwd: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_train_bwd: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(cu_seqlens != nullptr || n_pad % patch == 0, "window_attn_train_bwd: n_pad=%lld is not a multiple of patch=%d",
               (long long)n_pad, patch);
  PTV3_REQUIRE(lse != nullptr, "window_attn_train_bwd: the forward's lse is required");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_train_bwd: bad dtype %d", dtype);
  PTV3_REQUIRE(workspace_bytes >= ptv3_window_attn_bwd_workspace_bytes(n, n_pad, c, heads, dtype),
               "window_attn_train_bwd: workspace too small");
  const int nwin = cu_seqlens ? num_windows : (int)(n_pad / patch);
  return window_attn_bwd_impl(qkv, out, dout, win_order, win_inverse, cu_seqlens, nwin, dqkv, n, n_pad, c, heads, patch,
                              scale, dtype, workspace, (hipStream_t)stream, RpeBwd{nullptr, nullptr, 0, nullptr}, 0.f, 0,
                              lse);
}
