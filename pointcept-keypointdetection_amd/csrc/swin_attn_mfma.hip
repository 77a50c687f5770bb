// cRSE window attention (Swin3D) with every product on the matrix core.
//
// Same arithmetic as swin_attn_kernel (swin_attn.hip; reference call site pointcept/models/swin3d/swin3d_layers.py:556-569,
// restated in oracle/swin3d.py - parity unpinned):
//   e_ij  = q_i . k_j + sum_c ( q_i . T_K[c][idx_c(i,j)] + k_j . T_Q[c][idx_c(i,j)] ),   idx_c = clamp(floor(s_i[c] - s_j[c] + L_c))
//   out_i = sum_j softmax_j(e_ij) ( v_j + sum_c T_V[c][idx_c(i,j)] )
// That kernel fetches 3 S table rows of head_dim floats per (query, key) pair by a data-dependent index: 27 x 64 bytes
// through the vector L1 per pair at S = 9, D = 16 - its bound (64 B/clk/CU), 207 of the 260 ms of a 1M-point Swin3D-S
// forward.  When every axis has few table rows (2 L_c <= 64: quant_size 4 gives 32..56) the table terms are cheaper as
// SCALARS looked up in LDS, one signal axis at a time:
//   QT_c[i][r] = q_i . T_K[c][r]  for the 16 queries of a tile and all rows of the axis   (16 x D) x (D x 2 L_c)  MFMA
//   KT_c[j][r] = k_j . T_Q[c][r]  for 16 keys at a time                                                          MFMA
//   e_ij      += QT_c[i][idx_c] + KT_c[j][idx_c]                                          2 LDS words per pair and axis
//   out_i      = sum_j p_ij v_j + sum_c sum_r H_c[i][r] T_V[c][r],   H_c[i][idx_c(i,j)] += p_ij      both products MFMA;
//                H_c is a histogram of the weights over the rows of the axis: one LDS float add per pair and axis.
// v_mfma_f32_16x16x4_f32 throughout (exact fp32); bf16 only as the storage type of q, k, v, out.
// One WAVE per (window, head) - a workgroup is one wave, so nothing waits at a barrier and ~12 KB of LDS per wave keeps
// a dozen windows in flight per CU; the wave walks the window's queries 16 at a time.
#include <algorithm>
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int SWM_MAX_AXES = 9;
struct SwinTablesM {
  long long start[SWM_MAX_AXES];   // element offset of axis c's slab inside each concatenated table
  int rows[SWM_MAX_AXES];          // 2 L_c
};

// The workgroup is ONE wave and a wave's LDS instructions execute in order, so lanes exchange data through LDS
// without waiting on anything: the only thing to stop is the compiler moving LDS accesses across the hand-over.
// (A fence builtin would also drain vmcnt and with it the global loads issued ahead for the next tile.)
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

// A / B fragment of 16 rows x 16 K of a row-major fp32 or bf16 matrix in global memory.  v_mfma_f32_16x16x4_f32 takes
// K = 4 per step, one element per lane and step: with KE = min(D, 16) / 4 steps per chunk, lane (row = lane & 15, g)
// holds elements 16 nc + KE g .. + KE - 1 of its row (head_dim 8: two steps instead of four half-empty ones).
template <typename T, int D>
__device__ __forceinline__ f32x4 row_frag(const T* __restrict__ row, int nc, int g) {
  constexpr int KE = D >= 16 ? 4 : D / 4;
  const T* p = row + 16 * nc + KE * g;
  if constexpr (KE == 4) {
    float t[4];
    unpack4<T>(*reinterpret_cast<const typename Vec4<T>::type*>(p), t);
    return f32x4{t[0], t[1], t[2], t[3]};
  } else {
    static_assert(KE == 2, "head_dim 8, 16 or 32");
    if constexpr (sizeof(T) == 4) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(p);
      return f32x4{t[0], t[1], 0.f, 0.f};
    } else {
      const unsigned u = *reinterpret_cast<const unsigned*>(p);
      return f32x4{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u), 0.f, 0.f};
    }
  }
}

// acc += A B over NCH chunks of KE steps
template <int NCH, int KE>
__device__ __forceinline__ void mma_chain(const f32x4* a, const f32x4* b, f32x4& acc) {
#pragma unroll
  for (int nc = 0; nc < NCH; ++nc)
#pragma unroll
    for (int e = 0; e < KE; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nc][e], b[nc][e], acc, 0, 0, 0);
}
// acc0 += A0 B, acc1 += A1 B: the two chains alternate, so neither waits for its own result
template <int NCH, int KE>
__device__ __forceinline__ void mma_chain_pair(const f32x4* a0, const f32x4* a1, const f32x4* b, f32x4& acc0, f32x4& acc1) {
#pragma unroll
  for (int nc = 0; nc < NCH; ++nc)
#pragma unroll
    for (int e = 0; e < KE; ++e) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[nc][e], b[nc][e], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[nc][e], b[nc][e], acc1, 0, 0, 0);
    }
}

template <typename T, int D, int S>
__global__ __launch_bounds__(64) void swin_attn_mfma_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, const float* __restrict__ qt,
    const float* __restrict__ kt, const float* __restrict__ vt, SwinTablesM tab, const long long* __restrict__ n2n,
    const int* __restrict__ w_start, const float* __restrict__ crse, T* __restrict__ out, int heads, int mt16, int TS, int nB, int ablate) {
  constexpr int NCH = (D + 15) / 16;       // 16-wide K chunks of the products over head_dim
  constexpr int KE = D >= 16 ? 4 : D / 4;  // matrix-core steps per chunk
  constexpr int ND = (D + 15) / 16;        // 16-wide output column tiles
  constexpr int MAXCT = 4;                 // 16-row tiles of one axis' table (2 L_c <= 64)
  constexpr int VS = D + 4;                // row stride of the T_V slab copy
  extern __shared__ __align__(16) unsigned char swm_smem[];
  const int LP = mt16 + 4;
  float* sE = reinterpret_cast<float*>(swm_smem);     // [16][LP] logits, then weights, of the query tile
  float* sCT = sE + (size_t)16 * LP;                  // [S][mt16] signals of the window's voxels, axis-major
  float* sA = sCT + (size_t)S * mt16;                 // [16][TS] QT_c of the query tile
  float* sB = sA + (size_t)16 * TS;                   // [16][TS] KT_c of the current 16 keys; later [2 L_c][VS] T_V[c]
  int* sRow = reinterpret_cast<int*>(sB + (size_t)nB);   // [mt16] voxel of each token (tail repeats the last one)

  const int w = blockIdx.x, h = blockIdx.y;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const size_t hoff = (size_t)h * D, rstride = (size_t)heads * D;
  const int s0 = w_start[w];
  int m = w_start[w + 1] - s0;
  if (m > mt16) m = mt16;                   // host contract; never true for a valid partition
  if (m <= 0) return;
  const int mp = (m + 15) & ~15;
  for (int t = lane; t < mp; t += 64) sRow[t] = (int)n2n[s0 + min(t, m - 1)];
  for (int e = lane; e < mp * S; e += 64) {
    const int t = e / S, c = e - t * S;
    sCT[c * mt16 + t] = t < m ? crse[(size_t)s0 * S + e] : 0.f;
  }
  wave_lds_fence();

  for (int i0 = 0; i0 < m; i0 += 16) {
    // ---- q . k: rows of the product are keys, columns queries, so lane (i = li, g) ends up with pairs (i, jt + 4 g + e)
    f32x4 qf[NCH];
    {
      const T* qrow = q + ((size_t)sRow[i0 + li] * heads + h) * D;
#pragma unroll
      for (int nc = 0; nc < NCH; ++nc) qf[nc] = row_frag<T, D>(qrow, nc, g);
    }
    for (int jt = 0; jt < mp; jt += 16) {
      const T* krow = k + ((size_t)sRow[jt + li] * heads + h) * D;
      f32x4 kf[NCH], acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nc = 0; nc < NCH; ++nc) kf[nc] = row_frag<T, D>(krow, nc, g);
      mma_chain<NCH, KE>(kf, qf, acc);
      *reinterpret_cast<f32x4*>(sE + li * LP + jt + 4 * g) = acc;
    }
    // ---- table terms of the logits, one signal axis at a time
#pragma unroll 1
    for (int c = 0; c < ((ablate & 1) ? 0 : S); ++c) {
      const int rows = tab.rows[c], nct = (rows + 15) >> 4;
      const float half = (float)(rows >> 1);
      const float ci = sCT[c * mt16 + i0 + li];
      const size_t tbase = (size_t)tab.start[c] + hoff;
      f32x4 tq[MAXCT][NCH];
      // table tiles in pairs: two independent accumulation chains interleave on the matrix core
#pragma unroll
      for (int ct = 0; ct < MAXCT; ct += 2) {
        if (ct < nct) {
          const bool two = ct + 1 < nct;
          const size_t r0 = tbase + (size_t)min(16 * ct + li, rows - 1) * rstride;
          const size_t r1 = tbase + (size_t)min(16 * ct + 16 + li, rows - 1) * rstride;
          f32x4 tk0[NCH], tk1[NCH];
#pragma unroll
          for (int nc = 0; nc < NCH; ++nc) {
            tk0[nc] = row_frag<float, D>(kt + r0, nc, g);
            tq[ct][nc] = row_frag<float, D>(qt + r0, nc, g);
            if (two) {
              tk1[nc] = row_frag<float, D>(kt + r1, nc, g);
              tq[ct + 1][nc] = row_frag<float, D>(qt + r1, nc, g);
            }
          }
          f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
          if (two) mma_chain_pair<NCH, KE>(tk0, tk1, qf, acc0, acc1);
          else mma_chain<NCH, KE>(tk0, qf, acc0);
          *reinterpret_cast<f32x4*>(sA + li * TS + 16 * ct + 4 * g) = acc0;        // QT_c[i = li][16 ct + 4 g ..]
          if (two) *reinterpret_cast<f32x4*>(sA + li * TS + 16 * ct + 16 + 4 * g) = acc1;
        }
      }
      // KT_c of 16 keys at a time.  The products of tile jt + 16 are issued BEFORE the pairs of tile jt are looked up and
      // land in sB after them: the matrix core works through its 8-pass fp32 steps while the wave does the lookups.
      auto key_frags = [&](int jt, f32x4* kf) {
        const T* krow = k + ((size_t)sRow[jt + li] * heads + h) * D;
#pragma unroll
        for (int nc = 0; nc < NCH; ++nc) kf[nc] = row_frag<T, D>(krow, nc, g);
      };
      f32x4 kacc[MAXCT];
      auto key_products = [&](const f32x4* kf) {
#pragma unroll
        for (int ct = 0; ct < MAXCT; ct += 2) {
          kacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
          kacc[ct + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ct + 1 < nct) mma_chain_pair<NCH, KE>(tq[ct], tq[ct + 1], kf, kacc[ct], kacc[ct + 1]);
          else if (ct < nct) mma_chain<NCH, KE>(tq[ct], kf, kacc[ct]);
        }
      };
      auto store_products = [&]() {
#pragma unroll
        for (int ct = 0; ct < MAXCT; ++ct)
          if (ct < nct) *reinterpret_cast<f32x4*>(sB + li * TS + 16 * ct + 4 * g) = kacc[ct];   // KT_c[j = jt + li][16 ct + 4 g ..]
      };
      f32x4 kf[NCH];
      key_frags(0, kf);
      key_products(kf);
      store_products();
      if (mp > 16) key_frags(16, kf);
      wave_lds_fence();
      for (int jt = 0; jt < mp; jt += 16) {
        const bool more = jt + 16 < mp;
        if (more) {
          key_products(kf);
          if (jt + 32 < mp) key_frags(jt + 32, kf);
        }
        if (!(ablate & 8)) {
          const f32x4 sj = *reinterpret_cast<const f32x4*>(sCT + c * mt16 + jt + 4 * g);
          f32x4 ev = *reinterpret_cast<const f32x4*>(sE + li * LP + jt + 4 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int idx = (int)floorf((ci - sj[e]) + half);
            idx = min(max(idx, 0), rows - 1);
            ev[e] += sA[li * TS + idx] + sB[(4 * g + e) * TS + idx];
          }
          *reinterpret_cast<f32x4*>(sE + li * LP + jt + 4 * g) = ev;
        }
        wave_lds_fence();
        if (more) store_products();
        wave_lds_fence();
      }
    }
    // ---- softmax of row i = li: lane (li, g) owns keys 4 g .. 4 g + 3 of every 16; weights stay unnormalised
    float inv;
    {
      float mx = -INFINITY;
      for (int jt = 0; jt < mp; jt += 16) {
        const f32x4 ev = *reinterpret_cast<const f32x4*>(sE + li * LP + jt + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, jt + 4 * g + e < m ? ev[e] : -INFINITY);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float den = 0.f;
      for (int jt = 0; jt < mp; jt += 16) {
        f32x4 ev = *reinterpret_cast<const f32x4*>(sE + li * LP + jt + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ev[e] = jt + 4 * g + e < m ? __expf(ev[e] - mx) : 0.f;
          den += ev[e];
        }
        *reinterpret_cast<f32x4*>(sE + li * LP + jt + 4 * g) = ev;
      }
      den += __shfl_xor(den, 16);
      den += __shfl_xor(den, 32);
      inv = 1.f / den;
    }
    wave_lds_fence();
    // ---- P V: A = weights (lane (i = li, g): keys jt + 4 g ..), B = value rows (lane (d = li, g): keys jt + 4 g + e)
    f32x4 oacc[ND];
#pragma unroll
    for (int nd = 0; nd < ND; ++nd) oacc[nd] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int jt = 0; jt < mp; jt += 16) {
      const f32x4 pf = *reinterpret_cast<const f32x4*>(sE + li * LP + jt + 4 * g);
#pragma unroll
      for (int nd = 0; nd < ND; ++nd) {
        f32x4 vf = f32x4{0.f, 0.f, 0.f, 0.f};
        if (16 * nd + li < D) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vf[e] = to_f32<T>(v[((size_t)sRow[jt + 4 * g + e] * heads + h) * D + 16 * nd + li]);
        }
        oacc[nd] = mma16<float>(pf, vf, oacc[nd]);
      }
    }
    // ---- value-table term: per pair and axis one row of T_V[c], read from an LDS copy of the axis' slab (2 L_c x D
    // floats) and accumulated in registers by the lane that owns the pair.  (Measured and not kept: a histogram of the
    // weights over the rows, H_c[i][idx] += p by ds_add_f32, then H_c T_V[c] on the matrix core - LDS float adds retire
    // about one lane per 3 clocks, 190 clocks per wave instruction: 4.5 of the kernel's 8 ms.)
    float oa[D];                            // (measured and not kept: the same in packed pairs, v_pk_fma_f32 - 5 % slower)
#pragma unroll
    for (int d = 0; d < D; ++d) oa[d] = 0.f;
#pragma unroll 1
    for (int c = 0; c < ((ablate & 2) ? 0 : S); ++c) {
      const int rows = tab.rows[c];
      const float half = (float)(rows >> 1);
      const float ci = sCT[c * mt16 + i0 + li];
      const float* slab = vt + (size_t)tab.start[c] + hoff;
      for (int e = lane; e < rows * (D / 4); e += 64) {
        const int r = e / (D / 4), d4 = (e - r * (D / 4)) * 4;
        *reinterpret_cast<f32x4*>(sB + r * VS + d4) = *reinterpret_cast<const f32x4*>(slab + (size_t)r * rstride + d4);
      }
      wave_lds_fence();
      for (int jt = 0; jt < ((ablate & 4) ? 0 : mp); jt += 16) {
        const f32x4 sj = *reinterpret_cast<const f32x4*>(sCT + c * mt16 + jt + 4 * g);
        const f32x4 pf = *reinterpret_cast<const f32x4*>(sE + li * LP + jt + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int idx = (int)floorf((ci - sj[e]) + half);
          idx = min(max(idx, 0), rows - 1);
          const float* trow = sB + idx * VS;
#pragma unroll
          for (int d = 0; d < D; d += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(trow + d);
            oa[d] = fmaf(pf[e], t[0], oa[d]); oa[d + 1] = fmaf(pf[e], t[1], oa[d + 1]);
            oa[d + 2] = fmaf(pf[e], t[2], oa[d + 2]); oa[d + 3] = fmaf(pf[e], t[3], oa[d + 3]);
          }
        }
      }
      wave_lds_fence();
    }
    // the four lanes (li, g) of a query add their partial rows; lane (li, 0) hands the row to the (d = li, g) layout
#pragma unroll
    for (int d = 0; d < D; ++d) {
      oa[d] += __shfl_xor(oa[d], 16);
      oa[d] += __shfl_xor(oa[d], 32);
    }
    if (g == 0) {
#pragma unroll
      for (int d = 0; d < D; d += 4)
        *reinterpret_cast<f32x4*>(sA + li * VS + d) = f32x4{oa[d], oa[d + 1], oa[d + 2], oa[d + 3]};
    }
    wave_lds_fence();
#pragma unroll
    for (int nd = 0; nd < ND; ++nd)
      if (16 * nd + li < D) {
#pragma unroll
        for (int e = 0; e < 4; ++e) oacc[nd][e] += sA[(4 * g + e) * VS + 16 * nd + li];
      }
    wave_lds_fence();
    // ---- lane (d = li, g) holds out[i0 + 4 g + e][16 nd + li]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float ie = __shfl(inv, 4 * g + e);
      const int i = i0 + 4 * g + e;
#pragma unroll
      for (int nd = 0; nd < ND; ++nd)
        if (i < m && 16 * nd + li < D)
          out[((size_t)sRow[i] * heads + h) * D + 16 * nd + li] = from_f32<T>(oacc[nd][e] * ie);
    }
  }
}

template <typename T, int D, int S>
static int launch_swm(const void* q, const void* k, const void* v, const float* qt, const float* kt, const float* vt,
                      const SwinTablesM& tab, const long long* n2n, const int* w_start, int nwin, const float* crse,
                      void* out, int heads, int max_tokens, hipStream_t s) {
  int maxrows = 0;
  for (int c = 0; c < S; ++c) maxrows = std::max(maxrows, tab.rows[c]);
  if (maxrows > 64) return -1;     // long tables (quant_size 50): the products outweigh the gathers - not served here
  const int TS = ((maxrows + 15) & ~15) + 4;
  const int nB = std::max(16 * TS, maxrows * (D + 4));
  const char* ab = getenv("PTV3_SWIN_ABLATE");      // timing experiments only: wrong results
  const int ablate = ab ? atoi(ab) : 0;
  const int mt16 = (max_tokens + 15) & ~15;
  const size_t lds = ((size_t)16 * (mt16 + 4) + (size_t)S * mt16 + (size_t)16 * TS + nB + mt16) * 4;
  if (lds > 64 * 1024) return -1;
  if (lds > 32 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(&swin_attn_mfma_kernel<T, D, S>), 64 * 1024);
  hipLaunchKernelGGL((swin_attn_mfma_kernel<T, D, S>), dim3((unsigned)nwin, (unsigned)heads), dim3(64), lds, s,
                     (const T*)q, (const T*)k, (const T*)v, qt, kt, vt, tab, n2n, w_start, crse, (T*)out, heads, mt16, TS, nB, ablate);
  return hipGetLastError() == hipSuccess ? PTV3_OK : PTV3_ERR_LAUNCH;
}

template <typename T, int D>
static int swm_axes(int S, const void* q, const void* k, const void* v, const float* qt, const float* kt, const float* vt,
                    const SwinTablesM& tab, const long long* n2n, const int* w_start, int nwin, const float* crse,
                    void* out, int heads, int max_tokens, hipStream_t s) {
  switch (S) {
    case 3: return launch_swm<T, D, 3>(q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
    case 6: return launch_swm<T, D, 6>(q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
    case 9: return launch_swm<T, D, 9>(q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);
  }
  return -1;
}

// -1: shape not served by this kernel (the caller falls back to swin_attn_kernel); otherwise a PTV3_* status
int swin_attn_mfma(int dtype, int D, int S, const void* q, const void* k, const void* v, const float* qt, const float* kt,
                   const float* vt, const long long* start, const int* rows, const long long* n2n, const int* w_start,
                   int nwin, const float* crse, void* out, int heads, int max_tokens, hipStream_t s) {
  SwinTablesM tab;
  for (int c = 0; c < SWM_MAX_AXES; ++c) { tab.start[c] = start[c]; tab.rows[c] = rows[c]; }
#define SWM_DIM(T)                                                                                                  \
  switch (D) {                                                                                                      \
    case 8: return swm_axes<T, 8>(S, q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s);   \
    case 16: return swm_axes<T, 16>(S, q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s); \
    case 32: return swm_axes<T, 32>(S, q, k, v, qt, kt, vt, tab, n2n, w_start, nwin, crse, out, heads, max_tokens, s); \
    default: return -1;                                                                                             \
  }
  if (dtype == PTV3_F32) { SWM_DIM(float) } else { SWM_DIM(__bf16) }
#undef SWM_DIM
}

}  // namespace ptv3
