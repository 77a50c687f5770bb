// Per-window QK^T / softmax / PV with the serialized-order gather and the inverse scatter fused in.
// One workgroup = (window, head, block of queries); K and V of the window stream through LDS in
// 64-key tiles (K row-major, V transposed), scores never leave registers (online softmax).
// Matrix-core layout (mma16, common.h): S^T tile = K_tile(16 keys x 16 d) * Q^T  -> query on the
// lane, keys in the 4 acc registers x 4 lane groups; that accumulator is directly the B operand
// of O^T += V^T * P^T, so P never touches LDS.
// Reference semantics: SerializedAttention.forward, point_transformer_v3m1_base.py:184-216.
#include <type_traits>
#include <algorithm>
#include "common.h"
#include <stdlib.h>
#include "profile.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int WA_THREADS = 256;
constexpr int WA_WAVES = 4;
constexpr int WA_KT = 64;  // keys per LDS tile

template <typename T, int ND> struct WaCfg;
template <typename T> struct WaCfg<T, 1> { static constexpr int QT = 4; };
template <typename T> struct WaCfg<T, 2> { static constexpr int QT = 2; };
template <typename T> struct WaCfg<T, 4> { static constexpr int QT = 1; };

template <typename T, int ND>
__global__ void __launch_bounds__(WA_THREADS)
window_attn_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ win_order,
                   const int32_t* __restrict__ win_inverse, T* __restrict__ out, int C, int H, int Kmax,
                   int nwin, int qsplit, float scale_log2e, const float* __restrict__ rpe,
                   const int32_t* __restrict__ cu, unsigned drop_thr = 0, unsigned drop_seed = 0, float drop_scale = 1.f,
                   float* __restrict__ lse_out = nullptr) {
  typedef typename Vec4<T>::type V4;
  constexpr int D = 16 * ND;
  constexpr int QT = WaCfg<T, ND>::QT;
  constexpr int QB = WA_WAVES * QT * 16;  // queries per workgroup
  constexpr int KS = D + 4;               // K tile row stride (elements)
  constexpr int VS = WA_KT + 4;           // V^T tile row stride (elements)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);                      // [WA_KT][KS]
  T* sV = sK + WA_KT * KS;                                 // [D][VS]
  int32_t* sOrd = reinterpret_cast<int32_t*>(sV + D * VS);  // [K]

  const unsigned nwg = (unsigned)nwin * H * qsplit;
  const unsigned logical = xcd_remap(blockIdx.x, nwg);
  const int w = logical / (H * qsplit);
  const int rem = logical % (H * qsplit);
  const int h = rem / qsplit;
  const int qs = rem % qsplit;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  // varlen (cu != nullptr): window w holds the padded slots [cu[w], cu[w+1]), at most Kmax of them
  const int64_t wbase = cu ? (int64_t)cu[w] : (int64_t)w * Kmax;
  const int K = cu ? cu[w + 1] - cu[w] : Kmax;
  const int C3 = 3 * C;
  if (qs * QB >= K) return;  // workgroup-uniform: a short window needs fewer query blocks

  for (int i = tid; i < K; i += WA_THREADS) sOrd[i] = win_order[wbase + i];
  __syncthreads();

  // ---- Q fragments (B operand of the score product), pre-scaled by scale*log2(e)
  V4 qf[QT][ND];
  int qidx[QT];
  bool keep[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qidx[t] = qs * QB + (wave * QT + t) * 16 + li;
    const bool qv = qidx[t] < K;
    const int row = qv ? sOrd[qidx[t]] : 0;
    keep[t] = qv && win_inverse[row] == (int32_t)(wbase + qidx[t]);
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      V4 raw = zero4<T>();
      if (qv) raw = *reinterpret_cast<const V4*>(qkv + (int64_t)row * C3 + h * D + 16 * c + 4 * g);
      float f[4];
      unpack4<T>(raw, f);
      qf[t][c] = pack4<T>(f[0] * scale_log2e, f[1] * scale_log2e, f[2] * scale_log2e, f[3] * scale_log2e);
    }
  }

  f32x4 o[QT][ND];
  float m[QT], l[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    m[t] = -INFINITY;
    l[t] = 0.f;
#pragma unroll
    for (int c = 0; c < ND; ++c) o[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging map: thread -> (key row, 4-element chunk); D/4 chunks per row, 64 rows
  constexpr int CH = D / 4;                            // chunks per row
  constexpr int LOADS = (WA_KT * CH) / WA_THREADS;     // = ND
  V4 rk[LOADS], rv[LOADS];
  const int ntiles = (K + WA_KT - 1) / WA_KT;

  auto issue_loads = [&](int tile) {
#pragma unroll
    for (int u = 0; u < LOADS; ++u) {
      int e = u * WA_THREADS + tid;
      int kk = e / CH, ch = e % CH;
      int key = tile * WA_KT + kk;
      rk[u] = zero4<T>();
      rv[u] = zero4<T>();
      if (key < K) {
        const T* base = qkv + (int64_t)sOrd[key] * C3 + h * D + 4 * ch;
        rk[u] = *reinterpret_cast<const V4*>(base + C);
        rv[u] = *reinterpret_cast<const V4*>(base + 2 * C);
      }
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int u = 0; u < LOADS; ++u) {
      int e = u * WA_THREADS + tid;
      int kk = e / CH, ch = e % CH;
      *reinterpret_cast<V4*>(sK + kk * KS + 4 * ch) = rk[u];
      const T* pv = reinterpret_cast<const T*>(&rv[u]);
#pragma unroll
      for (int q = 0; q < 4; ++q) sV[(4 * ch + q) * VS + kk] = pv[q];
    }
  };

  issue_loads(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    write_lds();
    __syncthreads();
    if (tile + 1 < ntiles) issue_loads(tile + 1);

    // fragments of this key tile shared by all query tiles of the wave
    V4 kf[4][ND], vf[ND][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        kf[kt][c] = *reinterpret_cast<const V4*>(sK + (16 * kt + li) * KS + 16 * c + 4 * g);
        vf[c][kt] = *reinterpret_cast<const V4*>(sV + (16 * c + li) * VS + 16 * kt + 4 * g);
      }
    const int key0 = tile * WA_KT;
    const bool tail = key0 + WA_KT > K;

#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f32x4 s[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < ND; ++c) acc = mma16<T>(kf[kt][c], qf[t][c], acc);
        s[kt] = acc;
      }
      if (rpe != nullptr && qidx[t] < K) {
        const float* rb = rpe + (((int64_t)w * H + h) * K + qidx[t]) * K;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = key0 + 16 * kt + 4 * g + r;
            if (key < K) s[kt][r] += rb[key] * 1.44269504088896340736f;
          }
      }
      if (tail) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * kt + 4 * g + r >= K) s[kt][r] = -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m[t], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
      m[t] = mn;
      float ps = 0.f;
      V4 pf[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        float p0 = __builtin_amdgcn_exp2f(s[kt][0] - mn);
        float p1 = __builtin_amdgcn_exp2f(s[kt][1] - mn);
        float p2 = __builtin_amdgcn_exp2f(s[kt][2] - mn);
        float p3 = __builtin_amdgcn_exp2f(s[kt][3] - mn);
        ps += (p0 + p1) + (p2 + p3);
        if (drop_thr) {   // training with attn_drop > 0: the normaliser keeps every pair, the value sum the kept ones / (1 - p)
          const unsigned long long id = ((unsigned long long)((wbase + qidx[t]) * H + h) << 14) | (unsigned)(key0 + 16 * kt + 4 * g);
          p0 = drop_keep(id, drop_seed, drop_thr) ? p0 * drop_scale : 0.f;
          p1 = drop_keep(id + 1, drop_seed, drop_thr) ? p1 * drop_scale : 0.f;
          p2 = drop_keep(id + 2, drop_seed, drop_thr) ? p2 * drop_scale : 0.f;
          p3 = drop_keep(id + 3, drop_seed, drop_thr) ? p3 * drop_scale : 0.f;
        }
        pf[kt] = pack4<T>(p0, p1, p2, p3);
      }
      l[t] = l[t] * alpha + ps;
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        o[t][c] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) o[t][c] = mma16<T>(vf[c][kt], pf[kt], o[t][c]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: normalise, scatter back through the inverse map (dropping borrowed duplicates)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float lt = l[t];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    // training: log2-domain log-sum-exp of every slot's row (borrowed duplicates included: the backward walks slots)
    if (lse_out && g == 0 && qidx[t] < K) lse_out[(wbase + qidx[t]) * H + h] = m[t] + __log2f(lt);
    if (keep[t]) {
      const int row = sOrd[qidx[t]];
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        V4 v = pack4<T>(o[t][c][0] * inv, o[t][c][1] * inv, o[t][c][2] * inv, o[t][c][3] * inv);
        *reinterpret_cast<V4*>(out + (int64_t)row * C + h * D + 16 * c + 4 * g) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Resident-window variant: the K and V of the whole (window, head) are staged into LDS ONCE
// (bf16, K=1024, head_dim 16: 40 KB + 33 KB), then every wave streams its 64 queries over all key
// tiles with no further barrier.  Softmax rescale is lazy: the cross-lane max exchange and the O / l
// rescale only run (wave-uniformly) when some row maximum grew by more than 2^8 since the last
// rescale; exact (softmax is shift invariant), removes two lane exchanges per tile from the
// dependency chain.  Used whenever the window fits LDS; the tiled kernel above is the fallback.
// ------------------------------------------------------------------------------------------------
constexpr float WA_RESCALE_THR = 8.0f;  // log2 units: probabilities stay <= 256 between rescales

__device__ __forceinline__ float lanes_max_groups(float x) {
  // all-reduce max over the 4 lane groups g = lane >> 4 that share one query (lanes l, l^16, l^32, l^48);
  // only on the rare rescale path, so the LDS-crossbar shuffle is fine here
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}
__device__ __forceinline__ float lanes_sum_groups(float x) {
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}

// fmaxf(fmaxf(a, b), c) lowers to v_max3_f32 (an asm form reading MFMA results directly would need its own
// hazard padding: the compiler does not pad inside asm statements)
__device__ __forceinline__ float max3_raw(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

template <typename T> struct WaFull {
  static constexpr int VPAD = sizeof(T) == 2 ? 8 : 4;  // V^T row padding (elements): conflict-free fragment reads
};

// row sums of P: bf16 puts them on the matrix core (A = all-ones tile: every row of the product is the
// column sum of P^T, already summed over the 4 lane groups), fp32 keeps lane-partial VALU sums
template <typename T> struct RowSum;
template <> struct RowSum<__bf16> {
  static constexpr bool kOnMfma = true;
  static __device__ __forceinline__ s16x4 ones() { return s16x4{0x3F80, 0x3F80, 0x3F80, 0x3F80}; }
};
template <> struct RowSum<float> {
  static constexpr bool kOnMfma = false;
  static __device__ __forceinline__ f32x4 ones() { return f32x4{1.f, 1.f, 1.f, 1.f}; }
};

// RPE: 0 none; 1 dense bias tensor (windows, H, K, K) streamed from memory; 2 the reference's table lookup
// (RPE.forward, point_transformer_v3m1_base.py:29-48 on get_rel_pos :104-112) evaluated in the kernel:
// bias(q, k) = sum_axis table[axis*rpe_num + clamp(grid[q][axis] - grid[k][axis], -bnd, bnd) + bnd][head]
// from the window's voxel coordinates and the head's table column, both resident in LDS.
struct RpeTable { const int32_t* grid; const float* table; int pos_bnd; };

template <typename T, int ND, int RPE, int QT>
__global__ void __launch_bounds__(512)
window_attn_full_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ win_order,
                        const int32_t* __restrict__ win_inverse, T* __restrict__ out, int C, int H, int Kmax,
                        int Kpad, int nwin, int qsplit, float scale_log2e, const float* __restrict__ rpe,
                        RpeTable rt, const int32_t* __restrict__ cu, float* __restrict__ lse_out) {
  typedef typename Vec4<T>::type V4;
  constexpr int D = 16 * ND;
  constexpr int KS = D + 4;
  constexpr bool SUM_MFMA = RowSum<T>::kOnMfma;
  const int VS = Kpad + WaFull<T>::VPAD;
  const int nthreads = blockDim.x;
  const int QB = (nthreads >> 6) * QT * 16;  // queries per workgroup

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sK = reinterpret_cast<T*>(smem);                         // [Kpad][KS]
  T* sV = sK + (size_t)Kpad * KS;                             // [D][VS]
  int32_t* sOrd = reinterpret_cast<int32_t*>(sV + (size_t)D * VS);  // [Kpad]
  int32_t* sGC = sOrd + Kpad;                                       // RPE == 2: [Kpad][3] voxel coordinates
  const int rpe_num = 2 * rt.pos_bnd + 1;
  float* sTab = reinterpret_cast<float*>(sGC + 3 * Kpad);           // RPE == 2: [3*rpe_num] this head's column

  const unsigned nwg = (unsigned)nwin * H * qsplit;
  const unsigned logical = xcd_remap(blockIdx.x, nwg);
  const int w = logical / (H * qsplit);
  const int rem = logical % (H * qsplit);
  const int h = rem / qsplit;
  const int qs = rem % qsplit;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  // varlen (cu != nullptr; flash_attn_varlen semantics, v3m1_base.py:207-215): window w holds the padded slots
  // [cu[w], cu[w+1]), at most Kmax of them; the LDS layout stays that of a Kmax-key window
  const int64_t wbase = cu ? (int64_t)cu[w] : (int64_t)w * Kmax;
  const int K = cu ? cu[w + 1] - cu[w] : Kmax;
  const int Kst = (K + WA_KT - 1) / WA_KT * WA_KT;  // staged rows of THIS window (<= Kpad)
  const int C3 = 3 * C;
  if (qs * QB >= K) return;  // workgroup-uniform: a short window needs fewer query blocks

  for (int i = tid; i < Kst; i += nthreads) sOrd[i] = i < K ? win_order[wbase + i] : -1;
  __syncthreads();
  if constexpr (RPE == 2) {
    for (int i = tid; i < Kst; i += nthreads) {
      const int row = sOrd[i];
#pragma unroll
      for (int d = 0; d < 3; ++d) sGC[3 * i + d] = row >= 0 ? rt.grid[(int64_t)row * 3 + d] : 0;
    }
    for (int j = tid; j < 3 * rpe_num; j += nthreads) sTab[j] = rt.table[(int64_t)j * H + h] * 1.44269504088896340736f;
  }

  // ---- stage the whole window: K row-major, V transposed.  All gathers of a pass are issued before the first LDS
  // write; a 1024-key window at head_dim 16 is ONE pass of 8 loads per thread for an 8-wave workgroup (the gathers
  // are latency-bound: two dependent passes cost two round trips)
  constexpr int CH = D / 4;
  const int total = Kst * CH;
  auto stage_pass = [&](const int e0, auto unroll_tag) {
    constexpr int U = decltype(unroll_tag)::value;
    V4 rk[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * nthreads;
      rk[u] = zero4<T>();
      rv[u] = zero4<T>();
      if (e < total) {
        const int row = sOrd[e / CH];
        if (row >= 0) {
          const T* base = qkv + (int64_t)row * C3 + h * D + 4 * (e % CH);
          rk[u] = *reinterpret_cast<const V4*>(base + C);
          rv[u] = *reinterpret_cast<const V4*>(base + 2 * C);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = e0 + u * nthreads;
      if (e < total) {
        const int kk = e / CH, ch = e % CH;
        *reinterpret_cast<V4*>(sK + (size_t)kk * KS + 4 * ch) = rk[u];
        const T* pv = reinterpret_cast<const T*>(&rv[u]);
#pragma unroll
        for (int q = 0; q < 4; ++q) sV[(size_t)(4 * ch + q) * VS + kk] = pv[q];
      }
    }
  };
  {
    int e0 = tid;
    for (; e0 + 4 * nthreads < total; e0 += 8 * nthreads) stage_pass(e0, std::integral_constant<int, 8>{});
    for (; e0 < total; e0 += 4 * nthreads) stage_pass(e0, std::integral_constant<int, 4>{});
  }

  // ---- Q fragments of this wave (global loads overlap the staging above)
  V4 qf[QT][ND];
  int qidx[QT];
  int qg[QT][3];
  bool keep[QT];   // this slot is the point's own (not a borrowed duplicate): fetched NOW - a load issued between the
                   // stores of the epilogue would wait for every earlier store to be acknowledged (one in-order counter)
  bool any_q = false;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    qidx[t] = qs * QB + (wave * QT + t) * 16 + li;
    const bool qv = qidx[t] < K;
    any_q |= qv;
    const int row = qv ? sOrd[qidx[t]] : 0;
    keep[t] = qv && win_inverse[row] == (int32_t)(wbase + qidx[t]);
    if constexpr (RPE == 2) {
#pragma unroll
      for (int d = 0; d < 3; ++d) qg[t][d] = qv ? rt.grid[(int64_t)row * 3 + d] : 0;
    }
#pragma unroll
    for (int c = 0; c < ND; ++c) {
      V4 raw = zero4<T>();
      if (qv) raw = *reinterpret_cast<const V4*>(qkv + (int64_t)row * C3 + h * D + 16 * c + 4 * g);
      float f[4];
      unpack4<T>(raw, f);
      qf[t][c] = pack4<T>(f[0] * scale_log2e, f[1] * scale_log2e, f[2] * scale_log2e, f[3] * scale_log2e);
    }
  }
  __syncthreads();
  if (!__any(any_q)) return;  // whole wave past the window end: nothing left to synchronise with

  // softmax state per query tile.  negm = -(running reference max) is the INITIAL ACCUMULATOR of the score
  // product, so the matrix core hands back (score - max) and no subtraction is issued per score.
  f32x4 o[QT][ND], negm[QT], lacc[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    negm[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    lacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < ND; ++c) o[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the all-ones A operand of the row-sum product, made opaque once: as a known constant the compiler re-creates it
  // from scalar registers (two v_mov_b64) in front of every use
  V4 ones = RowSum<T>::ones();
  asm volatile("" : "+v"(ones));

  auto tile_body = [&](const int tile, auto mask_tag) {
    constexpr bool MASK = decltype(mask_tag)::value;
    const int key0 = tile * WA_KT;
    V4 kf[4][ND], vf[ND][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        kf[kt][c] = *reinterpret_cast<const V4*>(sK + (size_t)(key0 + 16 * kt + li) * KS + 16 * c + 4 * g);
        vf[c][kt] = *reinterpret_cast<const V4*>(sV + (size_t)(16 * c + li) * VS + key0 + 16 * kt + 4 * g);
      }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f32x4 s[4];  // score - reference max, log2 domain
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 acc = negm[t];
#pragma unroll
        for (int c = 0; c < ND; ++c) acc = mma16<T>(kf[kt][c], qf[t][c], acc);
        s[kt] = acc;
      }
      if constexpr (RPE == 2) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int* kg = sGC + 3 * (key0 + 16 * kt + 4 * g + r);
            float bias = 0.f;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const int rel = min(max(qg[t][d] - kg[d], -rt.pos_bnd), rt.pos_bnd) + rt.pos_bnd;
              bias += sTab[d * rpe_num + rel];
            }
            s[kt][r] += bias;
          }
      }
      if constexpr (RPE == 1) {
        if (qidx[t] < K) {
          const float* rb = rpe + (((int64_t)w * H + h) * K + qidx[t]) * K;
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              int key = key0 + 16 * kt + 4 * g + r;
              if (key < K) s[kt][r] += rb[key] * 1.44269504088896340736f;
            }
        }
      }
      if constexpr (MASK) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * kt + 4 * g + r >= K) s[kt][r] = -INFINITY;
      }
      // 16 values -> 8 x v_max3_f32
      float mx = max3_raw(s[0][0], s[0][1], s[0][2]);
      mx = max3_raw(mx, s[0][3], s[1][0]);
      mx = max3_raw(mx, s[1][1], s[1][2]);
      mx = max3_raw(mx, s[1][3], s[2][0]);
      mx = max3_raw(mx, s[2][1], s[2][2]);
      mx = max3_raw(mx, s[2][3], s[3][0]);
      mx = max3_raw(mx, s[3][1], s[3][2]);
      mx = fmaxf(mx, s[3][3]);
      // Lazy rescale (wave-uniform): the first tile pins the reference to its row maximum; afterwards the
      // exchange + rescale only run when some row maximum outgrew the reference by more than 2^THR.
      if (__builtin_expect(tile == 0 || __any(mx > WA_RESCALE_THR), 0)) {
        float d = lanes_max_groups(mx);      // new max - old reference (same for the 4 lanes of a query)
        if (tile > 0) d = fmaxf(d, 0.f);     // the reference never moves down after the first tile
        if (tile > 0) {
          const float alpha = __builtin_amdgcn_exp2f(-d);
          lacc[t] *= alpha;
#pragma unroll
          for (int c = 0; c < ND; ++c) o[t][c] *= alpha;
        }
        negm[t] -= d;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[kt] -= d;
      }
      float ps = 0.f;
      V4 pf[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const float p0 = __builtin_amdgcn_exp2f(s[kt][0]);
        const float p1 = __builtin_amdgcn_exp2f(s[kt][1]);
        const float p2 = __builtin_amdgcn_exp2f(s[kt][2]);
        const float p3 = __builtin_amdgcn_exp2f(s[kt][3]);
        if constexpr (!SUM_MFMA) ps += (p0 + p1) + (p2 + p3);
        pf[kt] = pack4<T>(p0, p1, p2, p3);
      }
      // P V and the row sums in 32-key steps (bf16: v_mfma_f32_16x16x32_bf16, half the matrix-core issues of the
      // 16-key form; the 32 keys of a step are the two 16-key score tiles side by side, in registers as they are)
      if constexpr (SUM_MFMA) {
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) lacc[t] = mma16x2<T>(ones, ones, pf[2 * kp], pf[2 * kp + 1], lacc[t]);
      } else {
        lacc[t][0] += ps;
      }
#pragma unroll
      for (int c = 0; c < ND; ++c)
#pragma unroll
        for (int kp = 0; kp < 2; ++kp)
          o[t][c] = mma16x2<T>(vf[c][2 * kp], vf[c][2 * kp + 1], pf[2 * kp], pf[2 * kp + 1], o[t][c]);
    }
  };

  const int full_tiles = K / WA_KT;
  for (int tile = 0; tile < full_tiles; ++tile) tile_body(tile, std::false_type{});
  if (full_tiles * WA_KT < K) tile_body(full_tiles, std::true_type{});

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const float lsum = SUM_MFMA ? lacc[t][0] : lanes_sum_groups(lacc[t][0]);
    const float inv = 1.0f / lsum;
    if (lse_out && g == 0 && qidx[t] < K) lse_out[(wbase + qidx[t]) * H + h] = __log2f(lsum) - negm[t][0];   // training
    if (keep[t]) {
      const int row = sOrd[qidx[t]];
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        V4 v = pack4<T>(o[t][c][0] * inv, o[t][c][1] * inv, o[t][c][2] * inv, o[t][c][3] * inv);
        *reinterpret_cast<V4*>(out + (int64_t)row * C + h * D + 16 * c + 4 * g) = v;
      }
    }
  }
}


// ---- launch configuration of the resident-window kernel.
// A wave walks all key tiles of its QT query tiles serially, so the pair (waves per workgroup, QT) trades
// per-wave serial length (deep levels: few windows, the launch lasts as long as ONE wave) against redundant
// staging of the window (every workgroup of a (window, head) pair stages all of K/V) and against the number of
// rounds the grid needs (LDS bounds the workgroups resident per CU).  full_config() evaluates a small cost model
// (cycles on the busiest CU) over the candidates; PTV3_ATTN_WAVES / PTV3_ATTN_QT force a choice for experiments.
template <typename T> struct FullArgs {
  const T* qkv; const int32_t* wo; const int32_t* wi; T* out; int C, H, K, Kpad, nwin; float scale_log2e;
  const float* rpe; RpeTable rt; int waves; size_t lds; hipStream_t s; const int32_t* cu; float* lse;
};

template <typename T, int ND, int RPE, int QT>
static void launch_full(const FullArgs<T>& a) {
  ensure_dynamic_lds(reinterpret_cast<const void*>(&window_attn_full_kernel<T, ND, RPE, QT>), 160 * 1024);
  const int QB = a.waves * QT * 16;
  const int qsplit = (a.K + QB - 1) / QB;
  const unsigned nwg = (unsigned)a.nwin * a.H * qsplit;
  hipLaunchKernelGGL((window_attn_full_kernel<T, ND, RPE, QT>), dim3(nwg), dim3(a.waves * 64), a.lds, a.s, a.qkv, a.wo,
                     a.wi, a.out, a.C, a.H, a.K, a.Kpad, a.nwin, qsplit, a.scale_log2e, a.rpe, a.rt, a.cu, a.lse);
}

template <typename T, int ND, int QT>
static void launch_full_rpe(const FullArgs<T>& a, int rpe_mode) {
  if (rpe_mode == 2) launch_full<T, ND, 2, QT>(a);
  else if (rpe_mode == 1) launch_full<T, ND, 1, QT>(a);
  else launch_full<T, ND, 0, QT>(a);
}

static void full_config(int64_t pairs, int K, int qt_max, size_t lds, int* waves_out, int* qt_out) {
  const char* ew = getenv("PTV3_ATTN_WAVES");
  const char* eq = getenv("PTV3_ATTN_QT");
  const int fw = ew ? atoi(ew) : 0, fq = eq ? atoi(eq) : 0;
  const int cap_lds = (int)std::max<size_t>(1, (160 * 1024) / lds);
  const double tiles = (K + WA_KT - 1) / WA_KT;
  double best = 1e300;
  int bw = 8, bq = qt_max;
  for (int waves = 8; waves >= 4; waves >>= 1) {
    if (fw && waves != fw) continue;
    for (int qt = qt_max; qt >= 1; qt >>= 1) {
      if (fq && qt != fq) continue;
      const int cap = std::min(cap_lds, 32 / waves);  // workgroups resident per CU (LDS, wave slots)
      const int QB = waves * qt * 16;
      const int64_t nwg = pairs * ((K + QB - 1) / QB);
      const int64_t per_round = 256LL * cap;
      const int64_t full_rounds = nwg / per_round;
      const int64_t rest = nwg - full_rounds * per_round;
      // one round with b workgroups on the busiest CU: staging (latency-bound, shared by the b workgroups only
      // partly) + the serial walk of a wave, stretched by the waves sharing its SIMD
      auto round_cost = [&](int64_t b) {
        const double wps = std::max(1.0, (double)b * waves / 4.0);           // waves per SIMD
        const double eff = wps >= 4 ? 0.70 : (wps >= 2 ? 0.50 : 0.35);        // issue efficiency seen in profiles
        const double walk = tiles * qt * 244.0 * wps / eff;                    // cycles
        const double stage = 2500.0 + 4.0 * K * (8.0 / waves) * std::max<double>(1.0, (double)b * 0.5);
        return walk + stage;
      };
      double cost = full_rounds * round_cost(cap);
      if (rest) cost += round_cost((rest + 255) / 256);
      if (cost < best) { best = cost; bw = waves; bq = qt; }
    }
  }
  *waves_out = bw;
  *qt_out = bq;
}

static bool g_last_launch_tiled = false;   // which kernel the last launch_window_attn chose (profiling tag only)

template <typename T, int ND>
static int launch_window_attn(const void* qkv, const int32_t* wo, const int32_t* wi, void* out, int C, int H,
                              int K, int nwin, float scale, const float* rpe, hipStream_t s,
                              RpeTable rt = RpeTable{nullptr, nullptr, 0}, const int32_t* cu = nullptr,
                              float* lse = nullptr) {
  constexpr int D = 16 * ND;
  constexpr int QT = WaCfg<T, ND>::QT;
  const int Kpad = (K + WA_KT - 1) / WA_KT * WA_KT;
  size_t lds_full = ((size_t)Kpad * (D + 4) + (size_t)D * (Kpad + WaFull<T>::VPAD)) * sizeof(T) + (size_t)Kpad * 4;
  if (rt.table) lds_full += (size_t)Kpad * 12 + (size_t)3 * (2 * rt.pos_bnd + 1) * 4;
  if (rt.table && lds_full > 160 * 1024) {
    set_error("window_attn_rpe: window of %d keys does not fit the resident-window kernel", K);
    return PTV3_ERR_UNSUPPORTED;
  }
  g_last_launch_tiled = lds_full > 160 * 1024;
  if (lds_full <= 160 * 1024) {
    int waves, qt;
    full_config(nwin * (int64_t)H, K, WaCfg<T, ND>::QT, lds_full, &waves, &qt);
    const FullArgs<T> a{(const T*)qkv, wo, wi, (T*)out, C, H, K, Kpad, nwin, scale * 1.44269504088896340736f, rpe,
                        rt, waves, lds_full, s, cu, lse};
    const int rpe_mode = rt.table ? 2 : (rpe ? 1 : 0);
    constexpr int QTMAX = WaCfg<T, ND>::QT;
    if (qt >= 4 && QTMAX >= 4) launch_full_rpe<T, ND, (QTMAX >= 4 ? 4 : QTMAX)>(a, rpe_mode);
    else if (qt >= 2 && QTMAX >= 2) launch_full_rpe<T, ND, (QTMAX >= 2 ? 2 : QTMAX)>(a, rpe_mode);
    else launch_full_rpe<T, ND, 1>(a, rpe_mode);
    PTV3_LAUNCH_CHECK();
    return PTV3_OK;
  }
  constexpr int QB = WA_WAVES * QT * 16;
  const int qsplit = (K + QB - 1) / QB;
  const size_t lds = (size_t)(WA_KT * (D + 4) + D * (WA_KT + 4)) * sizeof(T) + (size_t)K * 4;
  const unsigned nwg = (unsigned)nwin * H * qsplit;
  hipLaunchKernelGGL((window_attn_kernel<T, ND>), dim3(nwg), dim3(WA_THREADS), lds, s, (const T*)qkv, wo, wi,
                     (T*)out, C, H, K, nwin, qsplit, scale * 1.44269504088896340736f, rpe, cu, 0u, 0u, 1.f, lse);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

// training forward with attention dropout: always the tiled kernel (the mask costs ~12 integer operations per score; the
// resident-window kernel stays free of it)
template <typename T, int ND>
static int launch_window_attn_drop(const void* qkv, const int32_t* wo, const int32_t* wi, void* out, int C, int H, int K,
                                   int nwin, float scale, const int32_t* cu, unsigned thr, unsigned seed, float dscale,
                                   hipStream_t s) {
  constexpr int D = 16 * ND;
  constexpr int QT = WaCfg<T, ND>::QT;
  constexpr int QB = WA_WAVES * QT * 16;
  const int qsplit = (K + QB - 1) / QB;
  const size_t lds = (size_t)(WA_KT * (D + 4) + D * (WA_KT + 4)) * sizeof(T) + (size_t)K * 4;
  if (lds > 64 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(&window_attn_kernel<T, ND>), 160 * 1024);
  const unsigned nwg = (unsigned)nwin * H * qsplit;
  hipLaunchKernelGGL((window_attn_kernel<T, ND>), dim3(nwg), dim3(WA_THREADS), lds, s, (const T*)qkv, wo, wi, (T*)out, C, H,
                     K, nwin, qsplit, scale * 1.44269504088896340736f, (const float*)nullptr, cu, thr, seed, dscale);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_window_attn_drop_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                         const int32_t* cu_seqlens, int num_windows, void* out, int64_t n, int64_t n_pad,
                                         int c, int heads, int patch, float scale, float p_drop, uint32_t seed, int dtype,
                                         void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_drop: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_drop: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(cu_seqlens != nullptr || n_pad % patch == 0, "window_attn_drop: n_pad=%lld is not a multiple of patch=%d",
               (long long)n_pad, patch);
  PTV3_REQUIRE(p_drop > 0.f && p_drop < 1.f, "window_attn_drop: p_drop %g outside (0,1)", (double)p_drop);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_drop: bad dtype %d", dtype);
  const int d = c / heads;
  if (n == 0) return PTV3_OK;
  if (d != 16 && d != 32 && d != 64) {
    set_error("window_attn_drop: head_dim %d unsupported (16, 32, 64)", d);
    return PTV3_ERR_UNSUPPORTED;
  }
  const int nwin = cu_seqlens ? num_windows : (int)(n_pad / patch);
  const unsigned thr = (unsigned)std::min(4294967295.0, (double)p_drop * 4294967296.0);
  const float dscale = 1.f / (1.f - p_drop);
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_WINDOW_ATTN, 4.0 * n_pad * patch * c,
                              (double)n_pad * 3 * c * esz + (double)n * c * esz + 4.0 * (n_pad + n), nullptr, 0, 0.0);
  prof_kernel(prof, PK_ATTN_TILED);
  int rc = PTV3_ERR_UNSUPPORTED;
#define WAD_CASE(T)                                                                                                       \
  switch (d) {                                                                                                            \
    case 16: rc = launch_window_attn_drop<T, 1>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, cu_seqlens, thr, seed, dscale, s); break; \
    case 32: rc = launch_window_attn_drop<T, 2>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, cu_seqlens, thr, seed, dscale, s); break; \
    default: rc = launch_window_attn_drop<T, 4>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, cu_seqlens, thr, seed, dscale, s); break; \
  }
  if (dtype == PTV3_F32) { WAD_CASE(float) } else { WAD_CASE(__bf16) }
#undef WAD_CASE
  prof_end(prof, s);
  return rc;
}

static int window_attn_fwd_impl(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                const int32_t* cu, int nwin, void* out, int64_t n, int64_t n_pad, int c, int heads,
                                int patch, float scale, const float* rpe_bias, int dtype, double flops, hipStream_t s,
                                float* lse = nullptr) {
  const int d = c / heads;
  if (n == 0) return PTV3_OK;
  if (d != 16 && d != 32 && d != 64) {
    set_error("window_attn: head_dim %d unsupported (16, 32, 64)", d);
    return PTV3_ERR_UNSUPPORTED;
  }
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  // algorithmic work: 4 * sum_w len_w^2 * c flops; qkv read + out write + the two index maps
  const int prof = prof_begin(s, PROF_WINDOW_ATTN, flops,
                              (double)n_pad * 3 * c * esz + (double)n * c * esz + 4.0 * (n_pad + n), nullptr, 0, 0.0);
  int rc = PTV3_ERR_UNSUPPORTED;
  const RpeTable none{nullptr, nullptr, 0};
#define WA_CASE(T)                                                                                          \
  switch (d) {                                                                                              \
    case 16: rc = launch_window_attn<T, 1>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, rpe_bias, s, none, cu, lse); break; \
    case 32: rc = launch_window_attn<T, 2>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, rpe_bias, s, none, cu, lse); break; \
    case 64: rc = launch_window_attn<T, 4>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, rpe_bias, s, none, cu, lse); break; \
    default: break;                                                                                         \
  }
  if (dtype == PTV3_F32) { WA_CASE(float) } else { WA_CASE(__bf16) }
#undef WA_CASE
  prof_kernel(prof, g_last_launch_tiled ? PK_ATTN_TILED : PK_ATTN_FULL);
  prof_end(prof, s);
  return rc;
}

extern "C" int ptv3_window_attn_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                    void* out, int64_t n, int64_t n_pad, int c, int heads, int patch,
                                    float scale, const float* rpe_bias, int dtype, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(n_pad % patch == 0, "window_attn: n_pad=%lld is not a multiple of patch=%d (ragged windows: "
               "ptv3_window_attn_varlen_fwd)", (long long)n_pad, patch);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn: bad dtype %d", dtype);
  return window_attn_fwd_impl(qkv, win_order, win_inverse, nullptr, (int)(n_pad / patch), out, n, n_pad, c, heads,
                              patch, scale, rpe_bias, dtype, 4.0 * n_pad * patch * c, (hipStream_t)stream);
}

extern "C" int ptv3_window_attn_varlen_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                           const int32_t* cu_seqlens, int num_windows, void* out, int64_t n,
                                           int64_t n_pad, int c, int heads, int max_seqlen, float scale,
                                           double sum_len_sq, int dtype, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_varlen: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(max_seqlen >= 1 && max_seqlen <= 16384, "window_attn_varlen: max_seqlen %d outside [1,16384]", max_seqlen);
  PTV3_REQUIRE(cu_seqlens != nullptr && num_windows >= 1, "window_attn_varlen: cu_seqlens with >= 1 window required");
  PTV3_REQUIRE(n_pad <= (int64_t)num_windows * max_seqlen && n_pad >= num_windows,
               "window_attn_varlen: n_pad=%lld cannot be split into %d windows of 1..%d slots", (long long)n_pad,
               num_windows, max_seqlen);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_varlen: bad dtype %d", dtype);
  const double flops = 4.0 * (sum_len_sq > 0 ? sum_len_sq : (double)n_pad * max_seqlen) * c;
  return window_attn_fwd_impl(qkv, win_order, win_inverse, cu_seqlens, num_windows, out, n, n_pad, c, heads,
                              max_seqlen, scale, nullptr, dtype, flops, (hipStream_t)stream);
}

extern "C" int ptv3_window_attn_train_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                          const int32_t* cu_seqlens, int num_windows, void* out, float* lse, int64_t n,
                                          int64_t n_pad, int c, int heads, int patch, float scale, double sum_len_sq,
                                          int dtype, void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_train: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_train: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(cu_seqlens != nullptr || n_pad % patch == 0, "window_attn_train: n_pad=%lld is not a multiple of patch=%d",
               (long long)n_pad, patch);
  PTV3_REQUIRE(lse != nullptr, "window_attn_train: lse (n_pad x heads floats) is required");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_train: bad dtype %d", dtype);
  const int nwin = cu_seqlens ? num_windows : (int)(n_pad / patch);
  const double flops = 4.0 * (cu_seqlens && sum_len_sq > 0 ? sum_len_sq : (double)n_pad * patch) * c;
  return window_attn_fwd_impl(qkv, win_order, win_inverse, cu_seqlens, nwin, out, n, n_pad, c, heads, patch, scale,
                              nullptr, dtype, flops, (hipStream_t)stream, lse);
}

extern "C" int ptv3_window_attn_rpe_fwd(const void* qkv, const int32_t* win_order, const int32_t* win_inverse,
                                        void* out, int64_t n, int64_t n_pad, int c, int heads, int patch, float scale,
                                        const int32_t* grid_coord, const float* rpe_table, int pos_bnd, int dtype,
                                        void* stream) {
  PTV3_REQUIRE(heads > 0 && c % heads == 0, "window_attn_rpe: c=%d not divisible by heads=%d", c, heads);
  PTV3_REQUIRE(patch >= 1 && patch <= 16384, "window_attn_rpe: patch %d outside [1,16384]", patch);
  PTV3_REQUIRE(n_pad % patch == 0, "window_attn_rpe: n_pad=%lld is not a multiple of patch=%d", (long long)n_pad, patch);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "window_attn_rpe: bad dtype %d", dtype);
  PTV3_REQUIRE(grid_coord && rpe_table && pos_bnd >= 0 && pos_bnd <= 4096, "window_attn_rpe: grid_coord, rpe_table and 0 <= pos_bnd <= 4096 are required");
  const int d = c / heads;
  if (n == 0) return PTV3_OK;
  if (d != 16 && d != 32 && d != 64) {
    set_error("window_attn_rpe: head_dim %d unsupported (16, 32, 64)", d);
    return PTV3_ERR_UNSUPPORTED;
  }
  const int nwin = (int)(n_pad / patch);
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_WINDOW_ATTN, 4.0 * n_pad * patch * c,
                              (double)n_pad * 3 * c * esz + (double)n * c * esz + 4.0 * (n_pad + n) + 12.0 * n_pad,
                              nullptr, 0, 0.0);
  const RpeTable rt{grid_coord, rpe_table, pos_bnd};
  int rc = PTV3_ERR_UNSUPPORTED;
#define WAR_CASE(T)                                                                                                    \
  switch (d) {                                                                                                         \
    case 16: rc = launch_window_attn<T, 1>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, nullptr, s, rt); break; \
    case 32: rc = launch_window_attn<T, 2>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, nullptr, s, rt); break; \
    case 64: rc = launch_window_attn<T, 4>(qkv, win_order, win_inverse, out, c, heads, patch, nwin, scale, nullptr, s, rt); break; \
    default: break;                                                                                                    \
  }
  if (dtype == PTV3_F32) { WAR_CASE(float) } else { WAR_CASE(__bf16) }
#undef WAR_CASE
  prof_kernel(prof, PK_ATTN_FULL);
  prof_end(prof, s);
  return rc;
}
