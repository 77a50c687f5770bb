// Implicit-GEMM kernel shared by torch.nn.Linear and the submanifold sparse convolution:
//   out[i][o] = epilogue( sum_{d<kvol} sum_{c<cin} w[o][d][c] * x[nbr[i][d]][c] )
// (kvol = 1 and nbr = NULL gives the dense linear).  Rows of x are gathered straight into the
// LDS A-tile, weights stream through the LDS B-tile, both K-contiguous, 16 bytes per lane per
// access, so every matrix-core fragment is one ds_read_b128.  The product is computed transposed
// (W_tile * X_tile^T) so each lane ends up with 4 consecutive output channels of ONE point:
// vector epilogue + wide stores.  Small-M / huge-K shapes (deep-stage convolutions: M ~ 10^2..10^3
// rows against K = 27*C up to 13824) are split over K into fp32 slabs that a second kernel sums in
// slab order (bitwise reproducible) and finishes with the same epilogue.
// Reference semantics: include/ptv3_hip.h (ptv3_gemm).
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "profile.h"
#include "block_args.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int GM_THREADS = 256;
constexpr int GM_BM = 64;  // points per workgroup (4 waves x 16)

struct GemmArgs {
  const void* x; const void* w; void* out; void* out2; const void* res;
  const int32_t* nbr; const int32_t* row_order; const int32_t* res_index;
  const float* bias; const float* bn_scale; const float* bn_shift;
  float* slab;  // split-K partial sums [splits][m][cout] fp32 (NULL: direct epilogue)
  int64_t m; int cin; int cout; int kvol; int act; int cin_shift; int steps_per_split;
  int64_t x_bytes;   // size of x in bytes when it is known to be < 2^31 (buffer-descriptor addressing), else 0
  int ablate;        // tools only (PTV3_CONV_ABLATE): bit 0 no matrix-core work / fragment reads, 1 no operand copies, 2 no waits / barriers
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == PTV3_ACT_GELU) return gelu_erf(v);
  if (act == PTV3_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// bias -> folded BN -> activation -> (indexed) residual -> store(s) for 4 consecutive channels of one row
template <typename T>
__device__ __forceinline__ void epilogue_store(const GemmArgs& a, int64_t orow, int ch0, const float* accv) {
  typedef typename Vec4<T>::type V4;
  T* out = reinterpret_cast<T*>(a.out);
  T* out2 = reinterpret_cast<T*>(a.out2);
  const T* res = reinterpret_cast<const T*>(a.res);
  const bool vec_ok = (a.cout & 3) == 0;
  float v[4], v2[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ch = ch0 + r;
    float t = accv[r];
    if (ch < a.cout) {
      if (a.bias) t += a.bias[ch];
      if (a.bn_scale) t = t * a.bn_scale[ch] + a.bn_shift[ch];
      t = apply_act(t, a.act);
    }
    v[r] = t;
    v2[r] = t;
  }
  if (a.res) {
    const int64_t rrow = a.res_index ? (int64_t)a.res_index[orow] : orow;
    if (vec_ok) {
      float rr[4];
      unpack4<T>(*reinterpret_cast<const V4*>(res + rrow * a.cout + ch0), rr);
#pragma unroll
      for (int r = 0; r < 4; ++r) v2[r] = v[r] + rr[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ch0 + r < a.cout) v2[r] = v[r] + to_f32<T>(res[rrow * a.cout + ch0 + r]);
    }
  }
  if (vec_ok) {
    if (out2) {
      *reinterpret_cast<V4*>(out + orow * a.cout + ch0) = pack4<T>(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<V4*>(out2 + orow * a.cout + ch0) = pack4<T>(v2[0], v2[1], v2[2], v2[3]);
    } else {
      *reinterpret_cast<V4*>(out + orow * a.cout + ch0) = pack4<T>(v2[0], v2[1], v2[2], v2[3]);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (ch0 + r < a.cout) {
        if (out2) {
          out[orow * a.cout + ch0 + r] = from_f32<T>(v[r]);
          out2[orow * a.cout + ch0 + r] = from_f32<T>(v2[r]);
        } else {
          out[orow * a.cout + ch0 + r] = from_f32<T>(v2[r]);
        }
      }
  }
}

// ---- coalesced epilogue ---------------------------------------------------------------------------------------
// The matrix-core result leaves a lane with 4 consecutive channels of ONE point: stored from there, a wave
// instruction writes 16 rows x 32 bytes (bf16) - partial lines, measured at ~1 TB/s and 2/3 of a large linear's run
// time.  Instead the tile goes through LDS (the operand buffers are free after the K loop): phase 1 applies
// bias / folded BN / activation in registers and parks the values as T; phase 2 re-reads them row-contiguously,
// 16 bytes per lane, adds the (indexed) residual read the same way and stores whole 128+ byte row segments.
template <typename T> struct Chunk16;   // 16 bytes of T as floats
template <> struct Chunk16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* o) { f32x4 v = *reinterpret_cast<const f32x4*>(p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
  static __device__ __forceinline__ void store(float* p, const float* o) { *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]}; }
};
template <> struct Chunk16<__bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __bf16* p, float* o) {
    const s16x8 v = *reinterpret_cast<const s16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(v[i]);
  }
  static __device__ __forceinline__ void store(__bf16* p, const float* o) {
    const s16x4 lo = pack4<__bf16>(o[0], o[1], o[2], o[3]), hi = pack4<__bf16>(o[4], o[5], o[6], o[7]);
    *reinterpret_cast<s16x8*>(p) = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

// The per-channel epilogue vectors of one accumulator fragment (channels ch0..ch0+3; cout % 4 == 0 on this path), as
// three 16-byte loads.  All fragments' vectors are fetched back to back BEFORE any is used: one memory round trip.
// (Fetched one channel at a time inside the value loop - the first form of this epilogue - every element paid its
// own s_waitcnt vmcnt(0): ~60 dependent round trips per wave, 2/3 of a short-K linear's run time.)
struct EpiVec { f32x4 bias, scale, shift; };
// The tile's bn channels of the three vectors are parked in LDS at kernel start (sEpi: [3][bn] floats, visible after
// the first barrier of the K loop): the epilogue reads them back with three ds_read_b128 per fragment - no registers
// held across the K loop, no global load between the output stores.
__device__ __forceinline__ void park_epi(const GemmArgs& a, float* sEpi, int bn, int n0, int tid, int nthreads) {
  for (int t = tid; t < bn; t += nthreads) {
    const int ch = n0 + t;
    const bool in = ch < a.cout;
    sEpi[t] = (in && a.bias) ? a.bias[ch] : 0.f;
    sEpi[bn + t] = (in && a.bn_scale) ? a.bn_scale[ch] : 1.f;
    sEpi[2 * bn + t] = (in && a.bn_scale) ? a.bn_shift[ch] : 0.f;
  }
}
__device__ __forceinline__ EpiVec load_epi(const float* sEpi, int bn, int cl) {
  return EpiVec{*reinterpret_cast<const f32x4*>(sEpi + cl), *reinterpret_cast<const f32x4*>(sEpi + bn + cl),
                *reinterpret_cast<const f32x4*>(sEpi + 2 * bn + cl)};
}

// phase 1 for one accumulator fragment: channels ch0..ch0+3 of tile row `rl` -> sOut[rl][cl..cl+3]
template <typename T>
__device__ __forceinline__ void stage_values(const GemmArgs& a, T* sOut, int os, int rl, int cl, const EpiVec& e,
                                             f32x4 acc) {
  float v[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = apply_act((acc[r] + e.bias[r]) * e.scale[r] + e.shift[r], a.act);
  *reinterpret_cast<typename Vec4<T>::type*>(sOut + (size_t)rl * os + cl) = pack4<T>(v[0], v[1], v[2], v[3]);
}

// phase 2: `rows` x `bn` tile in sOut (row stride os) -> out / out2, whole row segments, 16 bytes per lane.
// Row lookups, residual loads and stores each go out in batches of UB: loads and stores share one in-order counter
// (s_waitcnt vmcnt), so a lookup issued after a store would wait for that store to be acknowledged - per iteration.
template <typename T, int UB>
__device__ __forceinline__ void store_tile(const GemmArgs& a, const T* sOut, int os, int rows, int bn, int64_t row0,
                                           int n0, int tid, int nthreads) {
  constexpr int N = Chunk16<T>::N;
  typedef typename Frag<T>::type FR;   // 16 bytes, kept packed until used (4 registers per chunk in flight)
  const int cpr = bn / N;
  const int total = rows * cpr;
  T* out = reinterpret_cast<T*>(a.out);
  T* out2 = reinterpret_cast<T*>(a.out2);
  const T* res = reinterpret_cast<const T*>(a.res);
  for (int e0 = tid; e0 < total; e0 += nthreads * UB) {
    int64_t orow[UB];
    int col[UB], rl[UB];
    bool ok[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int e = e0 + u * nthreads;
      rl[u] = e / cpr;
      col[u] = n0 + (e - rl[u] * cpr) * N;
      const int64_t prow = row0 + rl[u];
      ok[u] = e < total && prow < a.m && col[u] < a.cout;
      orow[u] = ok[u] ? (a.row_order ? (int64_t)a.row_order[prow] : prow) : 0;
    }
    FR rraw[UB];
    if (res) {
      int64_t rrow[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) rrow[u] = (ok[u] && a.res_index) ? (int64_t)a.res_index[orow[u]] : orow[u];
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (ok[u]) rraw[u] = *reinterpret_cast<const FR*>(res + rrow[u] * a.cout + col[u]);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (!ok[u]) continue;
      float v[N];
      Chunk16<T>::load(sOut + (size_t)rl[u] * os + (col[u] - n0), v);
      if (res) {
        if (out2) Chunk16<T>::store(out + orow[u] * a.cout + col[u], v);
        float r[N];
        Chunk16<T>::load(reinterpret_cast<const T*>(&rraw[u]), r);
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] += r[i];
        Chunk16<T>::store((out2 ? out2 : out) + orow[u] * a.cout + col[u], v);
      } else {
        Chunk16<T>::store(out + orow[u] * a.cout + col[u], v);
      }
    }
  }
}

// GATHER: rows of x come through the neighbour table (sparse convolution); a separate instantiation, so the dense
// linear carries none of the index arithmetic and the two show up under their own names in rocprofv3 / PMC tables
// PD: K steps fetched ahead into a register ring.  1 for chip-filling grids (occupancy hides the latency); 4 for the
// small grids of the deep levels, where a workgroup is alone on its CU and every K step otherwise costs a full
// global round trip (K = 256: all four stages are in flight before the first one is needed).
template <typename T, int NT, bool GATHER, int PD = 1>
__global__ void __launch_bounds__(GM_THREADS) gemm_kernel(GemmArgs a) {
  constexpr int BN = 16 * NT;  // output channels per workgroup
  typedef Frag<T> F;
  typedef typename F::type FR;
  constexpr int E = F::E;
  constexpr int BK = 2 * F::KC;      // K elements per LDS stage = 128 bytes per row
  constexpr int LS = BK + E;         // LDS row stride (elements): +16 bytes keeps ds_read_b128 conflict-free
  constexpr int CPR = BK / E;        // 16-byte chunks per row (8)
  constexpr int A_LOADS = (GM_BM * CPR) / GM_THREADS;  // 2
  constexpr int B_LOADS = (BN * CPR) / GM_THREADS;     // NT/2
  constexpr int OS = BN + 16 / (int)sizeof(T);  // staged output tile row stride (elements): +16 bytes
  constexpr int SM_OPER = (GM_BM + BN) * LS, SM_OUT = GM_BM * OS;
  __shared__ __attribute__((aligned(16))) T smem_all[SM_OPER > SM_OUT ? SM_OPER : SM_OUT];
  T* sA = smem_all;
  T* sB = smem_all + GM_BM * LS;
  __shared__ __attribute__((aligned(16))) float sEpi[3 * BN];
  // GATHER, kvol <= 27: the neighbour rows of the tile's 64 points, so that a stage's row addresses need no dependent
  // global read (the 5^3 stem conv keeps reading the table from memory)
  __shared__ int32_t sNbr[GATHER ? GM_BM * 27 : 1];

  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GM_BM;
  const int n0 = blockIdx.y * BN;
  const int ktot = a.kvol * a.cin;
  const int nsteps = (ktot + BK - 1) / BK;
  int step_lo = 0, step_hi = nsteps;
  if (a.slab) {
    step_lo = blockIdx.z * a.steps_per_split;
    step_hi = min(nsteps, step_lo + a.steps_per_split);
  }

  // A staging: chunk e -> (row e / CPR, chunk e % CPR); the chunks of one thread sit in one row
  int64_t arow[A_LOADS];
  int a_r[A_LOADS], a_ch[A_LOADS];
#pragma unroll
  for (int u = 0; u < A_LOADS; ++u) {
    int e = tid * A_LOADS + u;
    a_r[u] = e / CPR;
    a_ch[u] = e % CPR;
    int64_t r = row0 + a_r[u];
    if (r < a.m) arow[u] = a.row_order ? (int64_t)a.row_order[r] : r; else arow[u] = -1;
  }
  int b_r[B_LOADS], b_ch[B_LOADS];
#pragma unroll
  for (int u = 0; u < B_LOADS; ++u) {
    int e = tid * B_LOADS + u;
    b_r[u] = e / CPR;
    b_ch[u] = e % CPR;
  }

  if (!a.slab) park_epi(a, sEpi, BN, n0, tid, GM_THREADS);
  const bool nbr_lds = GATHER && a.kvol <= 27;
  if constexpr (GATHER) {
    if (nbr_lds) {
      for (int e = tid; e < GM_BM * a.kvol; e += GM_THREADS) {
        const int pr = e / a.kvol, d = e - pr * a.kvol;
        const int64_t r = row0 + pr;
        sNbr[e] = r < a.m ? a.nbr[(a.row_order ? (int64_t)a.row_order[r] : r) * a.kvol + d] : -1;
      }
      __syncthreads();
    }
  }
  // The operand loads of a stage are issued unconditionally (rows / channels / K past the end and missing neighbours
  // read a valid dummy address and are zeroed when the stage goes to LDS): a load inside a branch makes the compiler
  // wait with s_waitcnt vmcnt(0) in front of the LDS write - for every younger stage of the ring too.  The sparse-conv
  // neighbour index is the one dependent read; it is fetched with the stage and costs its own (L2-hit) round trip.
  FR rra[PD][A_LOADS], rrb[PD][B_LOADS];
  unsigned rok[PD];
  auto issue = [&](int step, FR (&ra)[A_LOADS], FR (&rb)[B_LOADS], unsigned& ok) {
    const int k0 = step * BK;
    ok = 0;
    int64_t src[A_LOADS];
    int cc[A_LOADS];
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      const int kk = k0 + E * a_ch[u];
      const bool in = arow[u] >= 0 && kk < ktot;
      const int kks = in ? kk : 0;
      src[u] = in ? arow[u] : 0;
      cc[u] = kks;
      if constexpr (GATHER) {
        const int d = a.cin_shift >= 0 ? (kks >> a.cin_shift) : (kks / a.cin);
        cc[u] = kks - d * a.cin;
        src[u] = nbr_lds ? (int64_t)sNbr[a_r[u] * a.kvol + d] : (int64_t)a.nbr[src[u] * a.kvol + d];
      }
      const bool v = in && src[u] >= 0;
      ok |= (unsigned)v << u;
      if (!v) src[u] = 0;
    }
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) ra[u] = *reinterpret_cast<const FR*>(x + src[u] * a.cin + cc[u]);
#pragma unroll
    for (int u = 0; u < B_LOADS; ++u) {
      const int kk = k0 + E * b_ch[u];
      const int o = n0 + b_r[u];
      const bool v = o < a.cout && kk < ktot;
      ok |= (unsigned)v << (8 + u);
      rb[u] = *reinterpret_cast<const FR*>(w + (int64_t)(v ? o : 0) * ktot + (v ? kk : 0));
    }
  };
  auto stash = [&](const FR (&ra)[A_LOADS], const FR (&rb)[B_LOADS], unsigned ok) {
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u)
      *reinterpret_cast<FR*>(sA + a_r[u] * LS + E * a_ch[u]) = ((ok >> u) & 1u) ? ra[u] : F::zero();
#pragma unroll
    for (int u = 0; u < B_LOADS; ++u)
      *reinterpret_cast<FR*>(sB + b_r[u] * LS + E * b_ch[u]) = ((ok >> (8 + u)) & 1u) ? rb[u] : F::zero();
  };

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int p = 0; p < PD; ++p)
    if (step_lo + p < step_hi) issue(step_lo + p, rra[p], rrb[p], rok[p]);
  for (int base = step_lo; base < step_hi; base += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int step = base + p;
      if (step >= step_hi) break;   // workgroup-uniform
      stash(rra[p], rrb[p], rok[p]);
      __syncthreads();
      if (step + PD < step_hi) issue(step + PD, rra[p], rrb[p], rok[p]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        FR xf = *reinterpret_cast<const FR*>(sA + (16 * wave + li) * LS + F::KC * ks + E * g);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          FR wf = *reinterpret_cast<const FR*>(sB + (16 * j + li) * LS + F::KC * ks + E * g);
          acc[j] = F::mma(wf, xf, acc[j]);  // D[channel 4g+r][point li]
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: lane owns point (16*wave + li), channels n0 + 16j + 4g .. +3
  if (!a.slab && (a.cout % Chunk16<T>::N) == 0) {
    // whole rows of the tile through LDS (the operand tiles are dead: the K loop ended on a barrier)
#pragma unroll
    for (int j = 0; j < NT; ++j)
      stage_values<T>(a, smem_all, OS, 16 * wave + li, 16 * j + 4 * g, load_epi(sEpi, BN, 16 * j + 4 * g), acc[j]);
    __syncthreads();
    store_tile<T, (GM_BM * BN / Chunk16<T>::N + GM_THREADS - 1) / GM_THREADS>(a, smem_all, OS, GM_BM, BN, row0, n0, tid,
                                                                              GM_THREADS);
    return;
  }
  const int64_t prow = row0 + 16 * wave + li;
  if (prow >= a.m) return;
  const int64_t orow = a.row_order ? (int64_t)a.row_order[prow] : prow;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch0 = n0 + 16 * j + 4 * g;
    if (ch0 >= a.cout) continue;
    if (a.slab) {
      float* dst = a.slab + ((int64_t)blockIdx.z * a.m + orow) * a.cout + ch0;
      if ((a.cout & 3) == 0) {
        *reinterpret_cast<f32x4*>(dst) = acc[j];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch0 + r < a.cout) dst[r] = acc[j][r];
      }
    } else {
      float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
      epilogue_store<T>(a, orow, ch0, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Large-M tile: 128 points x BN (64 | 128) channels per 4-wave workgroup, every wave a (BM/WM) x (BN/WN) sub-tile
// held as MI x NJ accumulator fragments (64 x 64 = 4 x 4 at BN = 128: one LDS fragment read per 2 matrix-core
// steps instead of 1.25 reads per step in gemm_kernel above, whose 16 x 64 wave tile is bound by LDS bandwidth
// at ~40 % of the matrix peak and measured ~10 %).  The two operand tiles are double-buffered in LDS with ONE
// barrier per K step: tile t+1 (fetched into registers during step t-1) is written to the idle buffer at the top
// of step t, the loads of tile t+2 are issued right after, then the fragments of tile t feed the matrix core.
// LDS rows are 128 bytes (8 x 16-byte chunks, chunk c of row r at c ^ ((r >> 1) & 7): conflict-free ds_read_b128 /
// ds_write_b128 without padding); the neighbour table of the tile's 128 points (kvol <= 27) sits in LDS, so the
// gathered row addresses of a sparse convolution cost no dependent global load.  Same operands, epilogue and
// results as gemm_kernel (fp32: bitwise, the accumulation order over K is the same).
// ------------------------------------------------------------------------------------------------------------
constexpr int GB_BM = 128;
constexpr int GB_THREADS = 256;
constexpr int GB_MAX_KVOL = 27;
constexpr int GB_MAX_STEPS = 256;   // K steps a tile can list for tap skipping (27 taps x 512 channels / 64 = 216)

template <typename T, int WM, int WN, int BN, bool GATHER, bool BUF>
__global__ void __launch_bounds__(GB_THREADS) gemm_big_kernel(GemmArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  constexpr int E = F::E;
  constexpr int BK = 2 * F::KC;                 // 128 bytes of K per row and stage
  constexpr int MI = GB_BM / (16 * WM);         // point fragments per wave
  constexpr int NJ = BN / (16 * WN);            // channel fragments per wave
  constexpr int X_LOADS = (GB_BM * 8) / GB_THREADS;  // 4
  constexpr int W_LOADS = (BN * 8) / GB_THREADS;     // 2 | 4
  static_assert(WM * WN == 4, "four waves");
  extern __shared__ __attribute__((aligned(16))) char gb_smem[];
  T* sX = reinterpret_cast<T*>(gb_smem);                         // [2][128][BK]
  T* sW = sX + 2 * GB_BM * BK;                                   // [2][BN][BK]
  int32_t* sNbr = reinterpret_cast<int32_t*>(sW + 2 * BN * BK);  // [128][kvol]
  // GATHER: the K steps that touch at least one active tap of this tile.  One byte each: with the 64 KB of operand
  // buffers, the 13.5 KB neighbour table and the epilogue vectors a workgroup must stay under 80 KB (two per CU)
  __shared__ unsigned char sSteps[GB_MAX_STEPS];
  __shared__ unsigned sTapMask;
  __shared__ int sWaveCnt[4];
  __shared__ __attribute__((aligned(16))) float sEpi[3 * BN];

  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int wr = wave / WN, wc = wave % WN;
  // 1-D grid, channel tile fastest inside a row tile, consecutive logical ids on one XCD: the column tiles of one
  // row tile run together on one XCD (its x rows come out of that L2 once, and the tile's output rows are written
  // whole - 256-byte pieces at a row stride left to different moments cost the memory side its write locality)
  const unsigned ntn = (a.cout + BN - 1) / BN;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned tile_m = logical / ntn;
  const int64_t row0 = (int64_t)tile_m * GB_BM;
  const int n0 = (int)(logical % ntn) * BN;
  const int ktot = a.kvol * a.cin;
  const int nsteps = (ktot + BK - 1) / BK;

  // ---- staging map: thread -> chunk c = tid & 7 of rows (tid >> 3) + 32 u
  const int c = tid & 7;
  const int r0 = tid >> 3;
  int64_t xrow[X_LOADS];   // dense: source row of x; conv: -1 marks rows beyond m
#pragma unroll
  for (int u = 0; u < X_LOADS; ++u) {
    const int64_t r = row0 + r0 + 32 * u;
    xrow[u] = r < a.m ? (a.row_order ? (int64_t)a.row_order[r] : r) : -1;
  }
  int nact = nsteps;   // K steps this tile really runs
  if constexpr (GATHER) {
    // neighbour rows of the tile's points -> LDS (row-major [point][kvol]), and which taps any of them has
    if (tid == 0) sTapMask = 0u;
    __syncthreads();
    unsigned mask = 0u;
    {
      // two threads per point, 14 taps each, all loads of a thread in flight at once (not one dependent
      // row_order -> nbr round trip per element of an element-strided loop)
      const int pr = tid >> 1, d0 = (tid & 1) * 14;
      const int64_t r = row0 + pr;
      const bool in = r < a.m;
      const int64_t orow = in ? (a.row_order ? (int64_t)a.row_order[r] : r) : 0;
      int32_t v[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) {
        const int d = d0 + u;
        v[u] = a.nbr[orow * a.kvol + (d < a.kvol ? d : 0)];
      }
#pragma unroll
      for (int u = 0; u < 14; ++u)
        if (d0 + u < a.kvol) {
          const int32_t nb = in ? v[u] : -1;
          sNbr[pr * a.kvol + d0 + u] = nb;
          if (nb >= 0) mask |= 1u << (d0 + u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mask |= __shfl_xor(mask, o, 64);
    if (lane == 0 && mask) atomicOr(&sTapMask, mask);
    __syncthreads();
    // Points that are neighbours in space (rows come in z-order) miss the same taps: a K step whose taps are empty
    // for the WHOLE tile contributes exact zeros and is skipped - its loads and its matrix-core work (a thin surface
    // or a LiDAR sheet leaves 50-75 % of the 27 taps empty).  The list keeps K order: results are bitwise unchanged.
    if (nsteps <= GB_MAX_STEPS) {
      const unsigned tm = sTapMask;
      bool act = false;
      if (tid < nsteps) {
        const int d0 = (tid * BK) / a.cin, d1 = min(a.kvol - 1, (tid * BK + BK - 1) / a.cin);
        for (int d = d0; d <= d1; ++d) act |= (tm >> d) & 1u;
      }
      const unsigned long long bal = __ballot(act);
      if (lane == 0) sWaveCnt[wave] = __popcll(bal);
      __syncthreads();
      int base = 0;
      for (int wv = 0; wv < wave; ++wv) base += sWaveCnt[wv];
      if (act) sSteps[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned char)tid;
      nact = sWaveCnt[0] + sWaveCnt[1] + sWaveCnt[2] + sWaveCnt[3];
      __syncthreads();
    }
  }
  auto step_of = [&](int i) { return (GATHER && nsteps <= GB_MAX_STEPS) ? (int)sSteps[i] : i; };

  park_epi(a, sEpi, BN, n0, tid, GB_THREADS);
  // ONE register set: tile t+1 is written to LDS at the top of step t, the set then takes tile t+2.  (A second set
  // holding tile t+3 - two tiles in flight - measured slower on every shape, 208 VGPRs: 1.08x -> 1.17x of the
  // 64-point tile on the 56k x 256 x 768 linear, 204 -> 239 us on the C = 256 sparse conv.)
  // Every load of a stage is issued UNCONDITIONALLY (a missing neighbour, a row or channel past the end read a valid
  // dummy address instead and are zeroed when the stage is written to LDS): loads inside branches make the compiler
  // wait with s_waitcnt vmcnt(0) - for the younger stage too - where a counted wait leaves that stage in flight.
  FR rxa[X_LOADS], rwa[W_LOADS], rxb[X_LOADS], rwb[W_LOADS];
  unsigned oka = 0, okb = 0;   // validity bits of the two stages: bit u = x chunk u, bit 8 + u = w chunk u
  // BUF: operands addressed through buffer descriptors (both smaller than 4 GB: the launcher checks).  An element that
  // does not exist - a missing neighbour (three quarters of a LiDAR sheet's 27 taps), a row, channel or K index past
  // the end - gets an offset beyond num_records: the load returns zeros WITHOUT a memory access, and the stage goes to
  // LDS as it stands.  (The pointer form reads a dummy row instead and zeroes it on the way to LDS: the same number of
  // cache requests as a dense tile, on a kernel bound by its operand traffic.)
  __amdgpu_buffer_rsrc_t xrs, wrs;
  if constexpr (BUF) {
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)a.x_bytes, 0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(w), 0, (int)((int64_t)a.cout * ktot * (int)sizeof(T)), 0x00020000);
  }
  auto issue = [&](int step, FR (&rx)[X_LOADS], FR (&rw)[W_LOADS], unsigned& ok) {
    const int kk = step * BK + E * c;
    const bool kin = kk < ktot;
    const int kks = kin ? kk : 0;
    int d = 0, cc = kks;
    if constexpr (GATHER) {
      d = a.cin_shift >= 0 ? (kks >> a.cin_shift) : (kks / a.cin);
      cc = kks - d * a.cin;
    }
    ok = 0;
#pragma unroll
    for (int u = 0; u < X_LOADS; ++u) {
      int64_t src = xrow[u];
      if constexpr (GATHER) src = sNbr[(r0 + 32 * u) * a.kvol + d];
      const bool v = kin && src >= 0;
      if constexpr (BUF) {
        const unsigned off = v ? (unsigned)((src * a.cin + cc) * (int)sizeof(T)) : 0xFFFFFFF0u;
        rx[u] = __builtin_bit_cast(FR, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0));
      } else {
        ok |= (unsigned)v << u;
        rx[u] = *reinterpret_cast<const FR*>(x + (v ? src : 0) * a.cin + cc);
      }
    }
#pragma unroll
    for (int u = 0; u < W_LOADS; ++u) {
      const int o = n0 + r0 + 32 * u;
      const bool v = kin && o < a.cout;
      if constexpr (BUF) {
        const unsigned off = v ? (unsigned)(((int64_t)o * ktot + kks) * (int)sizeof(T)) : 0xFFFFFFF0u;
        rw[u] = __builtin_bit_cast(FR, __builtin_amdgcn_raw_buffer_load_b128(wrs, off, 0, 0));
      } else {
        ok |= (unsigned)v << (8 + u);
        rw[u] = *reinterpret_cast<const FR*>(w + (int64_t)(v ? o : 0) * ktot + kks);
      }
    }
  };
  auto stash = [&](int buf, const FR (&rx)[X_LOADS], const FR (&rw)[W_LOADS], unsigned ok) {
#pragma unroll
    for (int u = 0; u < X_LOADS; ++u) {
      const int r = r0 + 32 * u;
      *reinterpret_cast<FR*>(sX + ((size_t)buf * GB_BM + r) * BK + E * (c ^ ((r >> 1) & 7))) =
          (BUF || ((ok >> u) & 1u)) ? rx[u] : F::zero();
    }
#pragma unroll
    for (int u = 0; u < W_LOADS; ++u) {
      const int r = r0 + 32 * u;
      *reinterpret_cast<FR*>(sW + ((size_t)buf * BN + r) * BK + E * (c ^ ((r >> 1) & 7))) =
          (BUF || ((ok >> (8 + u)) & 1u)) ? rw[u] : F::zero();
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const T* bx = sX + (size_t)buf * GB_BM * BK;
    const T* bw = sW + (size_t)buf * BN * BK;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      FR xf[MI], wf[NJ];
#pragma unroll
      for (int m = 0; m < MI; ++m) {
        const int r = (MI * wr + m) * 16 + li;
        xf[m] = *reinterpret_cast<const FR*>(bx + (size_t)r * BK + E * ((4 * ks + g) ^ ((r >> 1) & 7)));
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int r = (NJ * wc + j) * 16 + li;
        wf[j] = *reinterpret_cast<const FR*>(bw + (size_t)r * BK + E * ((4 * ks + g) ^ ((r >> 1) & 7)));
      }
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[m][j] = F::mma(wf[j], xf[m], acc[m][j]);  // D[channel 4g+r][point li]
    }
  };
  // Pipeline: LDS double buffer + two register sets.  Tile t+1 is written from its set into the idle LDS buffer at the
  // top of step t (its readers, step t-1, are behind the last barrier); that set is then re-issued for tile t+3,
  // while tile t+2 is still on its way into the other set: two tiles in flight per workgroup.  The steady state is
  // branch-free (two steps per iteration, sets named statically) so that the wait in front of a write counts only the
  // loads of ITS set (s_waitcnt vmcnt(n) with the younger set's loads left in flight), not vmcnt(0).
  if (nact > 0) {
    issue(step_of(0), rxa, rwa, oka);
    stash(0, rxa, rwa, oka);
  }
  if (nact > 1) issue(step_of(1), rxa, rwa, oka);
  if (nact > 2) issue(step_of(2), rxb, rwb, okb);
  __syncthreads();
  int step = 0;
  for (; step + 4 < nact; step += 2) {
    stash(1, rxa, rwa, oka);                  // tile step+1
    issue(step_of(step + 3), rxa, rwa, oka);
    compute(0);
    __syncthreads();
    stash(0, rxb, rwb, okb);                  // tile step+2
    issue(step_of(step + 4), rxb, rwb, okb);
    compute(1);
    __syncthreads();
  }
  for (; step < nact; ++step) {               // the last (up to four) steps, with the bounds checked
    const int buf = step & 1;
    if (buf == 0) {
      if (step + 1 < nact) stash(1, rxa, rwa, oka);
      if (step + 3 < nact) issue(step_of(step + 3), rxa, rwa, oka);
    } else {
      if (step + 1 < nact) stash(0, rxb, rwb, okb);
      if (step + 3 < nact) issue(step_of(step + 3), rxb, rwb, okb);
    }
    compute(buf);
    __syncthreads();
  }

  // ---- epilogue: lane owns point (16 (MI wr + m) + li), channels n0 + 16 (NJ wc + j) + 4g .. +3
  if ((a.cout % Chunk16<T>::N) == 0) {
    constexpr int OS = BN + 16 / (int)sizeof(T);
    T* sOut = reinterpret_cast<T*>(gb_smem);   // 128 x (BN + pad) T <= the operand buffers
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const EpiVec ev = load_epi(sEpi, BN, (NJ * wc + j) * 16 + 4 * g);
#pragma unroll
      for (int m = 0; m < MI; ++m)
        stage_values<T>(a, sOut, OS, (MI * wr + m) * 16 + li, (NJ * wc + j) * 16 + 4 * g, ev, acc[m][j]);
    }
    __syncthreads();
    store_tile<T, 8>(a, sOut, OS, GB_BM, BN, row0, n0, tid, GB_THREADS);
    return;
  }
#pragma unroll
  for (int m = 0; m < MI; ++m) {
    const int64_t prow = row0 + (MI * wr + m) * 16 + li;
    if (prow >= a.m) continue;
    const int64_t orow = a.row_order ? (int64_t)a.row_order[prow] : prow;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ch0 = n0 + (NJ * wc + j) * 16 + 4 * g;
      if (ch0 >= a.cout) continue;
      float v[4] = {acc[m][j][0], acc[m][j][1], acc[m][j][2], acc[m][j][3]};
      epilogue_store<T>(a, orow, ch0, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// 256-point tile for the sparse convolutions of the chip-filling deep levels (LiDAR-like scans: C = 256 / 512 with
// 10^4..10^5 sites).  gemm_big_kernel above moves 32 KB of operands per 2.1 MFLOP K-step through registers into LDS
// and sits at ~690 dense-equivalent TFLOP/s whatever the tap occupancy: the LDS STORE path (ds_write_b128: ~79 B/clk
// per CU) and the operand stream L2 -> CU bound it, not the matrix core (27 % busy).  Here:
//   * 256 points x 256 (128) channels per 8-wave workgroup, a wave holds 128 x 64 (64 x 64): 64 (48) KB of operands
//     per 8.4 (4.2) MFLOP K-step - half the bytes per flop;
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write; the
//     copies of step s+1 are issued right after the barrier of step s and land under its matrix-core work;
//   * both operands are addressed through buffer descriptors: a missing neighbour (or a channel row past cout) is an
//     offset beyond num_records, which the DMA turns into zeros in LDS without a memory access (probe:
//     tools/probes/lds_dma_oob.hip) - no validity bits, no selects, no dummy reads;
//   * LDS image as above (128-byte rows, chunk c of row r at c ^ ((r >> 1) & 7)), the swizzle applied to the SOURCE
//     chunk a lane fetches, the DMA destination being lane-linear;
//   * fragments are read one group (4 point fragments x the 4 channel fragments of a K half-step = 16 matrix-core
//     steps) ahead of their use.
// Same K order, epilogue and results as the other two kernels (fp32: bitwise).  cin must be a multiple of the 64 | 32
// channels of a K-step (one tap per step).
// ------------------------------------------------------------------------------------------------------------
constexpr int GC_BM = 256;
constexpr int GC_THREADS = 512;

template <int N, int END, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  if constexpr (N < END) {
    f(std::integral_constant<int, N>{});
    static_for<N + 1, END>(f);
  }
}

#define PTV3_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> __device__ __forceinline__ void gc_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// KB: bytes of K per row and stage (128 | 64); NSTG: stages in the LDS ring.  The copies of stage s + NSTG - 1 are
// issued right after the barrier of stage s, so NSTG - 2 younger stages stay in flight across every barrier (a
// counted s_waitcnt vmcnt in front of it).  Two stages of 128 bytes leave ONE stage in flight, issued when its
// predecessor starts computing: the copies then have one stage's matrix-core time (~0.8 us) to come back from L2 and
// the kernel waits for them (measured 1.77 us per K-step against 0.85 us of matrix-core work); four stages of 64
// bytes hold the same LDS and keep two to three stages in flight.
template <typename T, int BN, int KB, int NSTG>
__global__ void __launch_bounds__(BN >= 128 ? GC_THREADS : GC_THREADS / 2) conv_tile_kernel(GemmArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  constexpr int KS = KB / 64;                    // matrix-core K chunks per stage (16-byte fragments x 4 lane groups)
  constexpr int BK = KS * F::KC;                 // K elements per stage
  constexpr int NW = BN >= 128 ? 8 : 4;          // waves: 64 output channels are one wave column of four 64-point rows
  constexpr int NTH = 64 * NW;
  constexpr int WN = BN >= 128 ? BN / 64 : 1, WM = NW / WN;   // waves across channels / points
  constexpr int MI = GC_BM / (16 * WM), NJ = 4;  // accumulator fragments per wave: MI x NJ (8 x 4 | 4 x 4)
  constexpr int STG = (GC_BM + BN) * KB;         // bytes per stage
  constexpr int RPI = 1024 / KB;                 // rows per copy instruction (8 | 16), KB / 16 lanes per row
  constexpr int XQ = GC_BM / RPI / NW, WQ = BN / RPI / NW;   // copies per wave and stage
  constexpr int DPS = XQ + WQ;
  constexpr int NGRP = KS * (MI / 4);            // fragment groups per stage
  constexpr int AH = NSTG - 1;                   // stages issued ahead
  extern __shared__ __attribute__((aligned(16))) char gc_smem[];
  char* stg = gc_smem;                                                    // [NSTG][STG]
  int32_t* sNbr = reinterpret_cast<int32_t*>(gc_smem + NSTG * STG);     // [256][kvol <= 27]
  // the epilogue's output tile overlays the operand images and the table; the epilogue vectors sit behind both
  constexpr int OPS_BYTES = NSTG * STG + GC_BM * GB_MAX_KVOL * 4;
  constexpr int OUT_BYTES = GC_BM * (BN * (int)sizeof(T) + 16);
  float* sEpi = reinterpret_cast<float*>(gc_smem + (OPS_BYTES > OUT_BYTES ? OPS_BYTES : OUT_BYTES));   // [3][BN]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  const int wr = wave / WN, wc = wave % WN;
  const unsigned ntn = (a.cout + BN - 1) / BN;
  const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t row0 = (int64_t)(logical / ntn) * GC_BM;
  const int n0 = (int)(logical % ntn) * BN;
  const int ktot = a.kvol * a.cin;
  // split-K (a.slab): blockIdx.y takes steps [step_lo, step_lo + nsteps) and leaves raw fp32 partial sums in slab z -
  // the same step boundaries, accumulation order and slab layout as gemm_kernel's splits (bitwise the same slabs)
  const int step_lo = a.slab ? (int)blockIdx.y * a.steps_per_split : 0;
  const int nsteps = a.slab ? min(ktot / BK - step_lo, a.steps_per_split) : ktot / BK;

  {
    // neighbour rows of the tile's 256 points -> LDS: two threads per point, 14 taps each, every load of a thread in
    // flight at once (an element-strided loop makes 14 dependent row_order -> nbr round trips: 20 us of a 190 us launch)
#pragma unroll
    for (int half = tid; half < 2 * GC_BM; half += NTH) {
      const int pr = half >> 1, d0 = (half & 1) * 14;
      const int64_t r = row0 + pr;
      const bool in = r < a.m;
      const int64_t orow = in ? (a.row_order ? (int64_t)a.row_order[r] : r) : 0;
      int32_t v[14];
#pragma unroll
      for (int u = 0; u < 14; ++u) {
        const int d = d0 + u;
        v[u] = a.nbr[orow * a.kvol + (d < a.kvol ? d : 0)];
      }
#pragma unroll
      for (int u = 0; u < 14; ++u)
        if (d0 + u < a.kvol) sNbr[pr * a.kvol + d0 + u] = in ? v[u] : -1;
    }
  }
  if (!a.slab) park_epi(a, sEpi, BN, n0, tid, NTH);

  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.w), 0, (int)((int64_t)a.cout * ktot * (int)sizeof(T)), 0x00020000);
  // LDS image of an operand: KB-byte rows; 16-byte chunk c of row r sits at chunk c ^ swz(r) (conflict-free
  // ds_read_b128 of 16 consecutive rows).  Copy q = wave + 8 i covers rows RPI q .. RPI q + RPI - 1: lane ->
  // (row RPI q + lane / (KB/16), LDS chunk lane % (KB/16)), fetching SOURCE chunk c ^ swz(row).
  // (64-byte rows: a ds_read_b128 is served in four groups of 16 lanes that mix two lane quarters g - per row residue
  // mod 4 the group holds row quarters (0, 1, 2, 3) with g = (a, b, b, a) - and chunk g ^ [0, 3, 2, 1][row quarter]
  // gives its 16 lanes 16 different 16-byte slots of the 256-byte bank row)
  auto swz = [](int r) { return KB == 128 ? ((r >> 1) & 7) : ((0 - (r >> 2)) & 3); };
  constexpr int LPR = KB / 16;
  const int rr = lane / LPR, cch = lane % LPR;
  int xnb[XQ];          // LDS index of the lane's row in the neighbour table (x kvol)
  unsigned xcol[XQ];    // byte offset of the source chunk inside the stage's KB bytes
  unsigned woff[WQ];    // byte offset of (channel row, source chunk) in w, or out of range
#pragma unroll
  for (int i = 0; i < XQ; ++i) {
    const int r = RPI * (wave + NW * i) + rr;
    xnb[i] = r * a.kvol;
    xcol[i] = (unsigned)((cch ^ swz(r)) * 16);
  }
#pragma unroll
  for (int i = 0; i < WQ; ++i) {
    const int r = RPI * (wave + NW * i) + rr;
    const int o = n0 + r;
    woff[i] = o < a.cout ? (unsigned)(((int64_t)o * ktot) * (int)sizeof(T)) + (unsigned)((cch ^ swz(r)) * 16) : 0xFFFFFFF0u;
  }
  const unsigned row_bytes = (unsigned)(a.cin * (int)sizeof(T));
  auto issue = [&](int step) {
    char* buf = stg + (step % NSTG) * STG;
    const int k0 = (step_lo + step) * BK;
    const int d = a.cin_shift >= 0 ? (k0 >> a.cin_shift) : (k0 / a.cin);
    const unsigned kb = (unsigned)((k0 - d * a.cin) * (int)sizeof(T));   // byte offset of the stage's channels in a row of x
#pragma unroll
    for (int i = 0; i < XQ; ++i) {
      const int src = sNbr[xnb[i] + d];
      const unsigned off = src >= 0 ? (unsigned)src * row_bytes + kb + xcol[i] : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, PTV3_LDS_PTR(buf + (wave + NW * i) * 1024), 16, off, 0, 0, 0);
    }
    const unsigned kw = (unsigned)(k0 * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < WQ; ++i) {
      const unsigned off = woff[i] == 0xFFFFFFF0u ? woff[i] : woff[i] + kw;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, PTV3_LDS_PTR(buf + GC_BM * KB + (wave + NW * i) * 1024), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int m = 0; m < MI; ++m)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment of row r, K chunk (ks, g) of an operand image
  auto frag = [&](const char* img, int r, int ks) -> FR {
    return *reinterpret_cast<const FR*>(img + r * KB + 16 * ((4 * ks + g) ^ swz(r)));
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();     // neighbour table + epilogue vectors visible
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < AH; ++i)
    if (i < nsteps) issue(i);
  for (int step = 0; step < nsteps; ++step) {
    // this wave's copies of `step`: all but the copies of the younger stages that exist
    const int rem = nsteps - 1 - step;
    if (!(a.ablate & 4)) {
      if (rem >= AH - 1) gc_wait_vm<(AH - 1) * DPS>();
      else if (AH >= 3 && rem == 1) gc_wait_vm<DPS>();
      else gc_wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // everyone's copies landed; everyone is done with step - 1
      asm volatile("" ::: "memory");
    }
    if (step + AH < nsteps && !(a.ablate & 2)) issue(step + AH);
    if (a.ablate & 1) continue;
    const char* bx = stg + (step % NSTG) * STG;
    const char* bw = bx + GC_BM * KB;
    // group n = (K chunk ks = n / (MI/4), point quarter h = n % (MI/4)): 4 x NJ matrix-core steps
    FR xf[2][4], wf[2][NJ];
    auto load_group = [&](int n, FR (&x4)[4], FR (&w4)[NJ]) {
      const int ks = n / (MI / 4), h = n % (MI / 4);
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) w4[j] = frag(bw, (NJ * wc + j) * 16 + li, ks);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) x4[m] = frag(bx, (MI * wr + 4 * h + m) * 16 + li, ks);
    };
    load_group(0, xf[0], wf[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 4 + NJ, 0);
    static_for<0, NGRP>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int ks = n / (MI / 4), h = n % (MI / 4);
      if constexpr (n + 1 < NGRP) load_group(n + 1, xf[(n + 1) & 1], wf[((n + 1) / (MI / 4)) & 1]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[4 * h + m][j] = F::mma(wf[ks & 1][j], xf[n & 1][m], acc[4 * h + m][j]);
      // the group's first matrix-core step (behind the compiler's wait for the group's fragments) goes ahead of the
      // next group's reads, the rest of the group's steps run under them
      constexpr int MPG = 4 * NJ * (F::E == 8 ? 1 : 4);   // matrix-core instructions per group (fp32: 4 per fragment pair)
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if constexpr (n + 1 < NGRP) __builtin_amdgcn_sched_group_barrier(0x100, ((n + 1) % (MI / 4) == 0) ? 4 + NJ : 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, MPG - 1, 0);
    });
  }
  // ---- epilogue: lane owns point (16 (MI wr + m) + li), channels n0 + 16 (NJ wc + j) + 4g .. +3
  if (a.slab) {
#pragma unroll
    for (int m = 0; m < MI; ++m) {
      const int64_t prow = row0 + (MI * wr + m) * 16 + li;
      if (prow >= a.m) continue;
      const int64_t orow = a.row_order ? (int64_t)a.row_order[prow] : prow;
      float* dst = a.slab + ((int64_t)blockIdx.y * a.m + orow) * a.cout + n0 + 4 * g;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ch = (NJ * wc + j) * 16;
        if (n0 + ch + 4 * g < a.cout) *reinterpret_cast<f32x4*>(dst + ch) = acc[m][j];    // cout % 8 == 0 (policy)
      }
    }
    return;
  }
  __syncthreads();   // operand images are dead: the output tile takes their place
  constexpr int OS = BN + 16 / (int)sizeof(T);
  T* sOut = reinterpret_cast<T*>(gc_smem);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const EpiVec ev = load_epi(sEpi, BN, (NJ * wc + j) * 16 + 4 * g);
#pragma unroll
    for (int m = 0; m < MI; ++m)
      stage_values<T>(a, sOut, OS, (MI * wr + m) * 16 + li, (NJ * wc + j) * 16 + 4 * g, ev, acc[m][j]);
  }
  __syncthreads();
  store_tile<T, 8>(a, sOut, OS, GC_BM, BN, row0, n0, tid, NTH);
}

// the 256-point tile is one workgroup per CU: it pays when its grid fills most of a round of 256 CUs
static bool use_conv_tile(int64_t m, int cin, int cout, int kvol, int dtype, int64_t x_bytes, int* bn_out, int splits) {
  const char* env = getenv("PTV3_CONV_TILE");   // 0 off, 1 policy (default), 2 force; read per call so tests can switch
  const int mode = env ? atoi(env) : 1;
  const int bk = dtype == PTV3_F32 ? 32 : 64, gran = dtype == PTV3_F32 ? 4 : 8;
  // 64 output channels (the C = 64 levels: 10^4 .. 10^5 sites): a four-wave workgroup, 256 points x 64 channels.
  // PTV3_CONV_TILE_64 = 1 to enable (measurement switch)
  const char* e64 = getenv("PTV3_CONV_TILE_64");
  const bool narrow = cout == 64 && dtype == PTV3_BF16 && e64 && atoi(e64) != 0;
  if (mode == 0 || x_bytes <= 0 || kvol > GB_MAX_KVOL || kvol < 2 || cin % bk != 0 || cout % gran != 0 || (cout < 128 && !narrow))
    return false;
  if (narrow) {
    *bn_out = 64;
    return splits > 1 || mode == 2 || cdiv(m, GC_BM) >= 160;
  }
  if (splits > 1) {
    // the deep levels of a 100k-point scene (10^2 .. 10^4 sites at C = 128 .. 512): the K range of every split-K slab
    // through the 256-point tile (128 channels: more workgroups) instead of four 64-point tiles re-reading W -
    // 22 -> 17 us per convolution, the slabs and their consumers unchanged
    // bf16 only by default: with the exact-fp32 matrix path (a quarter of the rate) the forward measured 5.96 vs 5.90 ms
    const char* se = getenv("PTV3_CONV_TILE_SPLIT");   // 0: off, 1: both dtypes (tests), unset: bf16
    *bn_out = 128;
    return cout % 8 == 0 && (se ? atoi(se) != 0 : dtype == PTV3_BF16);
  }
  const int bn = (dtype == PTV3_BF16 && cout >= 256) ? 256 : 128;
  *bn_out = bn;
  if (mode == 2) return true;
  const int64_t tiles = cdiv(m, GC_BM) * cdiv(cout, bn);
  const int64_t rounds = cdiv(tiles, 256);
  return tiles >= 160 && 4 * tiles >= 3 * rounds * 256 && (int64_t)kvol * cin >= 1024;
}

// the large tile pays once its grid fills the chip; below that the 64-point tile (+ split-K) has more workgroups
static bool use_big_tile(int64_t m, int cin, int cout, int kvol, int dtype, int* bn_out) {
  const char* env = getenv("PTV3_GEMM_BIG");  // 0 off, 1 policy (default), 2 force; read per call so tests can switch
  const int mode = env ? atoi(env) : 1;
  // 128-channel variant only: at 64 output channels the 64-point tile measured faster (dec-0 conv 85 vs 100 us,
  // 120k x 64 x 64 linear 11.5 vs 13.2 us) - its grid is four times larger and W is tiny
  if (mode == 0 || kvol > GB_MAX_KVOL || cout < (mode == 2 ? 64 : 128)) return false;
  const int bn = cout >= 128 ? 128 : 64;
  *bn_out = bn;
  const int64_t tiles = cdiv(m, GB_BM) * cdiv(cout, bn);
  const int64_t ktot = (int64_t)kvol * cin;
  if (mode == 2) return true;
  // long K only (sparse convs: K = 27 * cin; fc2 / wide linears: K >= 1024): there the deeper pipeline and the 64 x 64
  // wave tile win (fc2 56k x 1024 x 256: 68 vs 81 us, 28k x 2048 x 512: 105 vs 139 us, C = 256 conv: 160 vs 260-340 us);
  // at K = 256..512 a tile is four to eight steps, the fill / drain of the bigger tile costs more than it saves
  // (qkv 56k x 256 x 768: 79 vs 69 us) and the 64-point tile keeps the job
  return tiles >= 256 && ktot >= 1024;
}

// sums the split-K slabs in slab order and applies the epilogue: one thread per (row, 4 channels)
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(GemmArgs a, int splits) {
  const int c4 = (a.cout + 3) / 4;
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.m * c4) return;
  const int64_t row = t / c4;
  const int ch0 = (int)(t - row * c4) * 4;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec_ok = (a.cout & 3) == 0;
  for (int z = 0; z < splits; ++z) {
    const float* src = a.slab + ((int64_t)z * a.m + row) * a.cout + ch0;
    if (vec_ok) {
      f32x4 p = *reinterpret_cast<const f32x4*>(src);
      v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ch0 + r < a.cout) v[r] += src[r];
    }
  }
  epilogue_store<T>(a, row, ch0, v);
}

static int bk_of(int dtype) { return dtype == PTV3_F32 ? 32 : 64; }

// 16-channel tiles per workgroup column block.  Measured on MI355X over every linear shape of the fork config
// (tools/bench_gemm.py): 64-channel blocks (NT = 4) are fastest or tied everywhere; wider blocks lose to the
// longer per-wave epilogue and the lower workgroup count.  32-channel blocks only for narrow outputs.
static int choose_nt(int cout) {
  static const int opts[] = {2, 4, 6, 8, 12, 16};
  static const bool tuning = getenv("PTV3_GEMM_TUNE") != nullptr;  // tools/bench_gemm.py only
  if (tuning) {
    if (const char* e = getenv("PTV3_GEMM_NT")) {
      int v = atoi(e);
      for (int nt : opts) if (nt == v) return v;
    }
  }
  return cout <= 32 ? 2 : 4;
}

// split decision shared by the workspace query and the launcher
static int choose_splits(int64_t m, int cin, int cout, int kvol, int dtype, int* steps_per_split) {
  const int bk = bk_of(dtype);
  const int64_t ktot = (int64_t)kvol * cin;
  const int nsteps = (int)((ktot + bk - 1) / bk);
  const int bn = 16 * choose_nt(cout);
  const int64_t base = cdiv(m, GM_BM) * cdiv(cout, bn);
  int splits = 1;
  if (nsteps >= 16 && base < 512) {
    int64_t want = 1536 / base;
    if (want > nsteps / 4) want = nsteps / 4;
    if (want >= 4) splits = (int)want;  // a 2-3 way split does not pay for its reduce launch
  }
  int sps = (nsteps + splits - 1) / splits;
  splits = (nsteps + sps - 1) / sps;
  *steps_per_split = sps;
  return splits;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_gemm_splits(int64_t m, int cin, int cout, int kvol, int dtype) {
  int sps;
  return choose_splits(m, cin, cout, kvol, dtype, &sps);
}

extern "C" size_t ptv3_gemm_workspace_bytes(int64_t m, int cin, int cout, int kvol, int dtype) {
  int sps;
  int splits = choose_splits(m, cin, cout, kvol, dtype, &sps);
  return splits > 1 ? (size_t)splits * m * cout * sizeof(float) : 0;
}

extern "C" int ptv3_gemm(const void* x, const void* w, void* out, int64_t m, int cin, int cout, int kvol,
                         const int32_t* nbr, const int32_t* row_order, const float* bias,
                         const float* bn_scale, const float* bn_shift, int act, const void* res,
                         const int32_t* res_index, void* out2, int dtype, void* workspace,
                         size_t workspace_bytes, void* stream) {
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "gemm: bad dtype %d", dtype);
  const int gran = dtype == PTV3_F32 ? 4 : 8;
  PTV3_REQUIRE(cin > 0 && cin % gran == 0, "gemm: cin=%d must be a positive multiple of %d for this dtype", cin, gran);
  PTV3_REQUIRE(cout > 0, "gemm: cout=%d", cout);
  PTV3_REQUIRE(kvol >= 1 && (kvol == 1 || nbr != nullptr), "gemm: kvol=%d needs a neighbour table", kvol);
  PTV3_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "gemm: bn_scale/bn_shift must come together");
  PTV3_REQUIRE(out2 == nullptr || res != nullptr, "gemm: out2 without res");
  if (m == 0) return PTV3_OK;
  int sps;
  int splits = choose_splits(m, cin, cout, kvol, dtype, &sps);
  {
    // a forced tile (tests, tools) also overrides the split-K decision of small shapes
    const char* ct = getenv("PTV3_CONV_TILE");
    const char* bg = getenv("PTV3_GEMM_BIG");
    const char* sk = getenv("PTV3_GEMM_SPLITK");   // 0: single pass over K (the tile tests compare K orders bitwise)
    if (out != nullptr && ((ct && atoi(ct) == 2 && nbr) || (bg && atoi(bg) == 2) || (sk && atoi(sk) == 0))) { splits = 1; sps = 0; }
  }
  int big_bn = 0;
  const bool big = splits <= 1 && out != nullptr && use_big_tile(m, cin, cout, kvol, dtype, &big_bn);
  if (splits > 1 && (workspace == nullptr || workspace_bytes < (size_t)splits * m * cout * sizeof(float))) {
    splits = 1;  // caller gave no slab room: single pass
    sps = 0;
  }
  // out == NULL: leave the raw fp32 slabs [splits][m][cout] in `workspace` (no bias / epilogue); a fused
  // consumer (ptv3_block_head) sums them.  Only valid when the shape really splits (ptv3_gemm_splits > 1).
  PTV3_REQUIRE(out != nullptr || splits > 1, "gemm: out == NULL needs a split-K shape and its workspace");
  int cin_shift = -1;
  if ((cin & (cin - 1)) == 0) { cin_shift = 0; while ((1 << cin_shift) < cin) ++cin_shift; }
  // gathered rows through a buffer descriptor (missing neighbours cost no memory access): x is the (m, cin) feature
  // matrix of the m active sites the neighbour table was built for, and both operands must be addressable with 32-bit
  // byte offsets.  PTV3_GEMM_BUF=0 keeps the pointer form (tools/bench_gemm_big.py compares them).
  const int esz0 = dtype == PTV3_F32 ? 4 : 2;
  const char* buf_env = getenv("PTV3_GEMM_BUF");
  const bool buf_on = !(buf_env && atoi(buf_env) == 0);
  const int64_t xb = m * cin * esz0, wb = (int64_t)cout * kvol * cin * esz0, lim = ((int64_t)1 << 31) - 4096;
  const int64_t x_bytes = (nbr && buf_on && xb < lim && wb < lim) ? xb : 0;
  GemmArgs a{x, w, out, out2, res, nbr, row_order, res_index, bias, bn_scale, bn_shift,
             splits > 1 ? (float*)workspace : nullptr, m, cin, cout, kvol, act, cin_shift, sps, x_bytes, 0};
  if (const char* e = getenv("PTV3_CONV_ABLATE")) a.ablate = atoi(e);
  hipStream_t s = (hipStream_t)stream;
  int nt = choose_nt(cout);
  if (nbr && nt != 2) nt = 4;
  const int bn = 16 * nt;
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, nbr ? PROF_SUBM_CONV : PROF_LINEAR, 2.0 * m * kvol * cin * cout,
                              ((double)m * cin * (nbr ? 1 : kvol) + (double)cout * kvol * cin +
                               (double)m * cout * (1 + (res != nullptr) + (out2 != nullptr))) * esz,
                              nbr, m * kvol, 2.0 * cin * cout);
  auto reduce_slabs = [&]() {
    GemmArgs r = a;
    r.row_order = nullptr;  // slabs are indexed by output row already
    const int64_t work = m * ((cout + 3) / 4);
    if (dtype == PTV3_F32)
      hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, s, r, splits);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<__bf16>, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, s, r, splits);
  };
  int ct_bn = 0;
  const bool ctile = nbr != nullptr && (out != nullptr || splits > 1) && use_conv_tile(m, cin, cout, kvol, dtype, x_bytes, &ct_bn, splits);
  if (ctile) {
    prof_kernel(prof, PK_CONV_TILE);
    // both ring shapes hold (256 + bn) x 256 bytes of operands: 2 stages x 128 bytes of K (default; LiDAR 120k forward
    // 9.34 ms against 9.55 ms) or 4 x 64 (PTV3_CONV_STAGES=4)
    const char* st_env = getenv("PTV3_CONV_STAGES");
    const bool four = st_env && atoi(st_env) == 4 && splits <= 1;   // split-K steps are 128 bytes of K (steps_per_split)
    const size_t lds_ops = (size_t)(GC_BM + ct_bn) * 256 + (size_t)GC_BM * GB_MAX_KVOL * 4;
    const size_t lds_out = (size_t)GC_BM * (ct_bn * esz + 16);
    const size_t lds = std::max(lds_ops, lds_out) + (size_t)3 * ct_bn * sizeof(float);
    PTV3_REQUIRE(lds <= 160 * 1024, "conv tile: %zu bytes of LDS", lds);
    dim3 cgrid((unsigned)(cdiv(m, GC_BM) * cdiv(cout, ct_bn)), (unsigned)std::max(splits, 1));
#define GC_LAUNCH(T, BN_, KB_, NS_)                                                                              \
    do {                                                                                                         \
      ensure_dynamic_lds(reinterpret_cast<const void*>(&conv_tile_kernel<T, BN_, KB_, NS_>), 160 * 1024);        \
      hipLaunchKernelGGL((conv_tile_kernel<T, BN_, KB_, NS_>), cgrid, gc_threads, lds, s, a);              \
    } while (0)
#define GC_PICK(T, BN_) do { if (four) GC_LAUNCH(T, BN_, 64, 4); else GC_LAUNCH(T, BN_, 128, 2); } while (0)
    const dim3 gc_threads(ct_bn == 64 ? GC_THREADS / 2 : GC_THREADS);
    if (dtype == PTV3_F32) GC_PICK(float, 128);
    else if (ct_bn == 256) GC_PICK(__bf16, 256);
    else if (ct_bn == 64) GC_PICK(__bf16, 64);
    else GC_PICK(__bf16, 128);
#undef GC_PICK
#undef GC_LAUNCH
    prof_end(prof, s);
    if (splits > 1 && out != nullptr) reduce_slabs();
    PTV3_LAUNCH_CHECK();
    return PTV3_OK;
  }
  prof_kernel(prof, big ? (nbr ? PK_GEMM_BIG_CONV : PK_GEMM_BIG_DENSE)
                        : nt == 2 ? (nbr ? PK_GEMM32_CONV : PK_GEMM32_DENSE) : (nbr ? PK_GEMM64_CONV : PK_GEMM64_DENSE));
  if (big) {
    const size_t lds_out = (size_t)GB_BM * (big_bn * esz + 16);   // staged output tile of the coalesced epilogue
    const size_t lds = std::max((size_t)2 * (GB_BM + big_bn) * 128 + (nbr ? (size_t)GB_BM * kvol * 4 : 0), lds_out);
    static_assert(2 * (GB_BM + 128) * 128 + GB_BM * GB_MAX_KVOL * 4 + GB_MAX_STEPS + 32 + 3 * 128 * 4 <= 80 * 1024,
                  "gemm_big_kernel: two workgroups per CU need <= 80 KB of LDS each");
    dim3 bgrid((unsigned)(cdiv(m, GB_BM) * cdiv(cout, big_bn)));
#define GB_LAUNCH(T, WM_, WN_, BN_, G_, B_)                                                                      \
    do {                                                                                                         \
      ensure_dynamic_lds(reinterpret_cast<const void*>(&gemm_big_kernel<T, WM_, WN_, BN_, G_, B_>), 96 * 1024);  \
      hipLaunchKernelGGL((gemm_big_kernel<T, WM_, WN_, BN_, G_, B_>), bgrid, dim3(GB_THREADS), lds, s, a);       \
    } while (0)
#define GB_PICK(T)                                                                                               \
    do {                                                                                                         \
      if (nbr && x_bytes) { if (big_bn == 128) GB_LAUNCH(T, 2, 2, 128, true, true); else GB_LAUNCH(T, 4, 1, 64, true, true); } \
      else if (nbr) { if (big_bn == 128) GB_LAUNCH(T, 2, 2, 128, true, false); else GB_LAUNCH(T, 4, 1, 64, true, false); }     \
      else { if (big_bn == 128) GB_LAUNCH(T, 2, 2, 128, false, false); else GB_LAUNCH(T, 4, 1, 64, false, false); }            \
    } while (0)
    if (dtype == PTV3_F32) GB_PICK(float); else GB_PICK(__bf16);
#undef GB_PICK
#undef GB_LAUNCH
    prof_end(prof, s);
    PTV3_LAUNCH_CHECK();
    return PTV3_OK;
  }
  dim3 grid((unsigned)cdiv(m, GM_BM), (unsigned)cdiv(cout, bn), (unsigned)splits);
#define GM_LAUNCH(T, G)                                                                                \
  switch (nt) {                                                                                        \
    case 2: hipLaunchKernelGGL((gemm_kernel<T, 2, G>), grid, dim3(GM_THREADS), 0, s, a); break;        \
    case 4: hipLaunchKernelGGL((gemm_kernel<T, 4, G>), grid, dim3(GM_THREADS), 0, s, a); break;        \
    case 6: hipLaunchKernelGGL((gemm_kernel<T, 6, G>), grid, dim3(GM_THREADS), 0, s, a); break;        \
    case 8: hipLaunchKernelGGL((gemm_kernel<T, 8, G>), grid, dim3(GM_THREADS), 0, s, a); break;        \
    case 12: hipLaunchKernelGGL((gemm_kernel<T, 12, G>), grid, dim3(GM_THREADS), 0, s, a); break;      \
    default: hipLaunchKernelGGL((gemm_kernel<T, 16, G>), grid, dim3(GM_THREADS), 0, s, a); break;      \
  }
  // small grids (the deep levels): four K stages in flight per workgroup instead of one
  // register-ring depth: 4 for the small grids of the deep levels, PTV3_GEMM_PD_LARGE (default 1) for chip-filling ones
  const char* pd_env = getenv("PTV3_GEMM_PD");
  const char* pdl_env = getenv("PTV3_GEMM_PD_LARGE");
  int pd = 1;
  if (nt == 4) {
    const bool small = (int64_t)grid.x * grid.y * grid.z <= 1024;
    pd = pd_env ? atoi(pd_env) : (small ? 4 : (pdl_env ? atoi(pdl_env) : 1));
    if (pd != 2 && pd != 4) pd = 1;
  }
#define GM_PD(T, G)                                                                                            \
  do {                                                                                                         \
    if (pd == 4) hipLaunchKernelGGL((gemm_kernel<T, 4, G, 4>), grid, dim3(GM_THREADS), 0, s, a);               \
    else if (pd == 2) hipLaunchKernelGGL((gemm_kernel<T, 4, G, 2>), grid, dim3(GM_THREADS), 0, s, a);          \
    else hipLaunchKernelGGL((gemm_kernel<T, 4, G, 1>), grid, dim3(GM_THREADS), 0, s, a);                       \
  } while (0)
  if (nbr) {
    // the gathered instantiations only exist for the tile widths the policy picks (2, 4)
    if (dtype == PTV3_F32) {
      if (nt == 2) hipLaunchKernelGGL((gemm_kernel<float, 2, true>), grid, dim3(GM_THREADS), 0, s, a);
      else GM_PD(float, true);
    } else {
      if (nt == 2) hipLaunchKernelGGL((gemm_kernel<__bf16, 2, true>), grid, dim3(GM_THREADS), 0, s, a);
      else GM_PD(__bf16, true);
    }
  } else if (nt == 4) {
    if (dtype == PTV3_F32) GM_PD(float, false); else GM_PD(__bf16, false);
  } else if (dtype == PTV3_F32) { GM_LAUNCH(float, false) } else { GM_LAUNCH(__bf16, false) }
#undef GM_PD
#undef GM_LAUNCH
  prof_end(prof, s);   // the bracket times the GEMM launch alone (the slab reduce below is its own, tiny kernel)
  if (splits > 1 && out != nullptr) reduce_slabs();
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
