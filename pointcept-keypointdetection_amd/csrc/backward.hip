// Backward kernels of the row-local layers of the PTv3 path (training):
//   weight gradients of nn.Linear / SubMConv3d  (dW = dY^T . gather(X)), column reductions (bias, BatchNorm,
//   LayerNorm affine gradients), LayerNorm / BatchNorm / activation input gradients, segment-max and
//   gather backward of SerializedPooling / SerializedUnpooling.
// All reductions over points are deterministic: fixed row chunks -> fp32 slabs -> slab-ordered sum.
#include "common.h"
#include "profile.h"
#include <stdlib.h>
#include "../../include/ptv3_hip.h"

namespace ptv3 {

// ------------------------------------------------------------------------------------------------
// out[j] = sum_s slab[s][j]   (slab order = fixed summation order)
// ------------------------------------------------------------------------------------------------
// block = 64 columns x 16 slab lanes (1024 threads): a wave reads 256 contiguous bytes of one slab row, lane z sums slabs
// z, z+16, ... in order, the 16 lane sums are then added in lane order (fixed tree: deterministic).  (With 16 columns per
// block the reads were 64-byte pieces: 59 us for the ~30 MB of slabs of one block backward.)
constexpr int SS_COLS = 64, SS_LANES = 16;
// V = 4: a thread owns four consecutive columns (16-byte loads: a wave reads 1 KB of one slab row per instruction; the
// scalar form's 256-byte pieces left the reductions of one block backward at ~500 GB/s, 60 us); same per-element order.
template <int V>
__device__ __forceinline__ void slab_sum_body(const float* __restrict__ slab, int nslab, long long n, float* __restrict__ out,
                                              long long n0, float* __restrict__ out1, long long col_block) {
  typedef float VT __attribute__((ext_vector_type(V)));
  __shared__ float red[SS_LANES][V * SS_COLS + 4];
  const int c = threadIdx.x & (SS_COLS - 1), z0 = threadIdx.x / SS_COLS;
  const long long j = (col_block * SS_COLS + c) * V;
  float s[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s[e] = 0.f;
  if (j < n) {
    // eight loads in flight per lane, added in slab order (the sum is the same chain as a plain loop: one load per
    // trip made every addition wait a full memory round trip)
    int z = z0;
    for (; z + 7 * SS_LANES < nslab; z += 8 * SS_LANES) {
      VT v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const VT*>(slab + (long long)(z + u * SS_LANES) * n + j);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += V == 1 ? ((const float*)&v[u])[0] : ((const float*)&v[u])[e];
    }
    for (; z < nslab; z += SS_LANES) {
      const VT v = *reinterpret_cast<const VT*>(slab + (long long)z * n + j);
#pragma unroll
      for (int e = 0; e < V; ++e) s[e] += ((const float*)&v)[e];
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[z0][V * c + e] = s[e];
  __syncthreads();
  if (z0 == 0 && j < n) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float t = 0.f;
#pragma unroll
      for (int z = 0; z < SS_LANES; ++z) t += red[z][V * c + e];
      if (j + e < n0) out[j + e] = t; else out1[j + e - n0] = t;   // entries n0 .. n-1 of a slab row go to a second buffer
    }
  }
}
static inline bool slab_vec4(const float* slab, long long n) { return n % 4 == 0 && ((uintptr_t)slab & 15) == 0; }
static inline int slab_blocks(const float* slab, long long n) { return (int)cdiv(n, SS_COLS * (slab_vec4(slab, n) ? 4 : 1)); }

__global__ void __launch_bounds__(SS_COLS * SS_LANES) slab_sum_kernel(const float* __restrict__ slab, int nslab, int64_t n,
                                                                      float* __restrict__ out, int64_t n0,
                                                                      float* __restrict__ out1, int vec) {
  if (vec) slab_sum_body<4>(slab, nslab, n, out, n0, out1, blockIdx.x);
  else slab_sum_body<1>(slab, nslab, n, out, n0, out1, blockIdx.x);
}

// Deferred reductions: a caller that issues several slab-producing kernels back to back (the native block backward:
// six weight gradients and three LayerNorm backward passes) can collect their reductions and run them as ONE launch at
// the end - each reduction alone is a few microseconds of work behind a full launch (242 such launches were 1.6 ms of
// the fork model's training step).  Every deferred producer must own its slab memory until the flush.
struct SlabSeg { const float* slab; float* out; float* out1; long long n, n0; int nslab, first_block, vec; };
constexpr int SLAB_SEGS = 16;
struct SlabSegs { SlabSeg v[SLAB_SEGS]; };
static thread_local SlabSegs* g_defer = nullptr;
static thread_local int g_defer_count = 0, g_defer_blocks = 0;

__global__ void __launch_bounds__(SS_COLS * SS_LANES) slab_sum_multi_kernel(SlabSegs segs, int nseg) {
  int k = 0;
#pragma unroll 1
  for (int t = 1; t < nseg; ++t)
    if (segs.v[t].first_block <= (int)blockIdx.x) k = t;
  const SlabSeg g = segs.v[k];
  if (g.vec) slab_sum_body<4>(g.slab, g.nslab, g.n, g.out, g.n0, g.out1, (int)blockIdx.x - g.first_block);
  else slab_sum_body<1>(g.slab, g.nslab, g.n, g.out, g.n0, g.out1, (int)blockIdx.x - g.first_block);
}

int slab_defer_flush(hipStream_t s) {
  if (g_defer && g_defer_count > 0)
    hipLaunchKernelGGL(slab_sum_multi_kernel, dim3((unsigned)g_defer_blocks), dim3(SS_COLS * SS_LANES), 0, s, *g_defer,
                       g_defer_count);
  g_defer_count = g_defer_blocks = 0;
  return PTV3_OK;
}
void slab_defer_begin(void* storage) {   // storage: sizeof(SlabSegs) bytes owned by the caller; NULL ends deferral
  g_defer = (SlabSegs*)storage;
  g_defer_count = g_defer_blocks = 0;
}
size_t slab_defer_storage_bytes() { return sizeof(SlabSegs); }

static void slab_sum(const float* slab, int nslab, int64_t n, float* out, hipStream_t s, int64_t n0 = -1,
                     float* out1 = nullptr) {
  if (g_defer) {
    if (g_defer_count == SLAB_SEGS) slab_defer_flush(s);
    SlabSeg& g = g_defer->v[g_defer_count++];
    g.slab = slab; g.out = out; g.out1 = out1; g.n = n; g.n0 = n0 < 0 ? n : n0; g.nslab = nslab;
    g.first_block = g_defer_blocks;
    g.vec = slab_vec4(slab, n) ? 1 : 0;
    g_defer_blocks += slab_blocks(slab, n);
    return;
  }
  hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)slab_blocks(slab, n)), dim3(SS_COLS * SS_LANES), 0, s, slab, nslab, n,
                     out, n0 < 0 ? n : n0, out1, slab_vec4(slab, n) ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// dW[o][tap][c] = sum_i dY[i][o] * X[nbr[i][tap]][c]          (kvol = 1, nbr = NULL: plain dY^T X)
// The contraction runs over POINTS, the slow axis of both operands, while the matrix core wants the
// contraction index contiguous per lane.  Each workgroup therefore stages 64-row blocks of dY and of the
// gathered X through LDS TRANSPOSED ([channel][row]; coalesced 16-byte global loads, element-wise LDS
// stores), after which lane (i, g) reads its 4 consecutive rows of channel i as one fragment and the same
// mma16<T> as the forward kernels applies (bf16: v_mfma_f32_16x16x16_bf16, fp32: exact 16x16x4 chain).
// Workgroup = one (<=64 x <=64) output tile over one row chunk; its 4 waves split the 16-row k-steps of
// every block and are summed through LDS at the end; row chunks -> fp32 slabs -> ordered sum.
// The next block's global loads are issued before the current block's MFMAs.
// ------------------------------------------------------------------------------------------------
struct TnArgs {
  const void* dy; const void* x; const int32_t* nbr; float* out;
  float* dbias;   // NULL, or where chunk z writes its column sums of dy: dbias + z * slab_stride (cout floats)
  int64_t m, rows_per_chunk, slab_stride;
  int cout, cin, kvol, tiles_c;
  int packed;   // narrow convolutions (cin < 64): the 64 columns of a tile run over (tap, channel), several taps per tile
};

// Rows per staged block RB (k-steps of 16 rows, RB/64 per wave) and LDS row stride LS (elements).  One block = one
// round of global loads -> LDS -> barrier -> MFMAs -> barrier.  LS / 2 (bf16 dwords) = 34 modulo 64 keeps the fragment
// reads of a half wave on 32 distinct bank pairs, and = 2 modulo 16 holds the transposing stores to 2-way conflicts
// (an even stride cannot do better: the stores are 8-byte pieces of rows 4 channels apart).
constexpr int TN_RB_MAX = 128;
template <typename T, int RB, int LS>
__device__ __forceinline__ void gemm_tn_body(const TnArgs& a, const int bx, const int by, const int bz) {
  typedef typename Vec4<T>::type V4;
  constexpr int NJ = RB / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char tn_smem[];
  T* sA = reinterpret_cast<T*>(tn_smem);          // [o][row]
  T* sB = sA + 64 * LS;                           // [c][row]
  float* sR = reinterpret_cast<float*>(tn_smem);  // cross-wave reduction, after the last block
  const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int to = bx / a.tiles_c, tc = bx % a.tiles_c;
  const int o0 = 64 * to, c0 = 64 * tc;
  const int ncols = a.packed ? a.kvol * a.cin : a.cin;   // column space of this launch's tiles
  const int nto = (min(64, a.cout - o0) + 15) / 16, ntc = (min(64, ncols - c0) + 15) / 16;
  const int64_t r0 = (int64_t)bz * a.rows_per_chunk;
  const int64_t r1 = r0 + a.rows_per_chunk < a.m ? r0 + a.rows_per_chunk : a.m;
  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: each thread owns NJ 4-row x 4-channel patches of both operands (rows 64j + 4rg .. +3, channels 4cg .. +3),
  // transposes them in registers and stores 4 x (4 consecutive rows of one channel) per patch
  const int rg = threadIdx.x >> 4, cg = threadIdx.x & 15;
  const int mycol = c0 + 4 * cg;                                   // this thread's 4 columns of the gathered operand
  const int tap = a.packed ? mycol / a.cin : by;      // packed: a tile spans 64 / cin taps
  const int mych = a.packed ? mycol % a.cin : mycol;
  const bool oka = o0 + 4 * cg < a.cout, okb = mycol < ncols;
  V4 ra[NJ][4], rb[NJ][4];
  // Every load is issued unconditionally from an in-range address and the value dropped afterwards: a load inside a
  // branch makes the compiler wait for ALL outstanding loads (vmcnt(0)) at the branch, which serialised the
  // index -> row chains of a thread's rows; this way the indices of all rows are in flight together, then all rows.
  const int64_t rlast = r1 - 1;                    // r1 > r0: a chunk is never empty
  const int ca = oka ? o0 + 4 * cg : 0, cb = okb ? mych : 0;
  auto fetch = [&](int64_t rblk) {
    int src[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = rblk + 64 * j + 4 * rg + u;
        const int64_t rc = r < r1 ? r : rlast;
        src[j][u] = a.nbr ? a.nbr[rc * a.kvol + (okb ? tap : 0)] : (int)rc;
        ra[j][u] = *reinterpret_cast<const V4*>(dy + rc * a.cout + ca);
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = rblk + 64 * j + 4 * rg + u;
        const bool live = r < r1;
        const int sidx = src[j][u];
        rb[j][u] = *reinterpret_cast<const V4*>(x + (int64_t)(sidx >= 0 ? sidx : 0) * a.cin + cb);
        if (!(live && oka)) ra[j][u] = zero4<T>();
        if (!(live && okb && sidx >= 0)) rb[j][u] = zero4<T>();
      }
  };
  // bias gradient (column sums of dy) as a by-product of the staging: this thread sees channels o0 + 4cg .. +3 of
  // every row block; only the workgroups of the first input-channel tile / first tap keep them
  const bool want_b = a.dbias != nullptr && tc == 0 && (a.packed || by == 0);
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  fetch(r0);
  for (int64_t rblk = r0; rblk < r1; rblk += RB) {
    if (want_b) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float t4[4];
          unpack4<T>(ra[j][u], t4);
          bs[0] += t4[0]; bs[1] += t4[1]; bs[2] += t4[2]; bs[3] += t4[3];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const T* ea[4] = {reinterpret_cast<const T*>(&ra[j][0]), reinterpret_cast<const T*>(&ra[j][1]),
                        reinterpret_cast<const T*>(&ra[j][2]), reinterpret_cast<const T*>(&ra[j][3])};
      const T* eb[4] = {reinterpret_cast<const T*>(&rb[j][0]), reinterpret_cast<const T*>(&rb[j][1]),
                        reinterpret_cast<const T*>(&rb[j][2]), reinterpret_cast<const T*>(&rb[j][3])};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T ta[4] = {ea[0][e], ea[1][e], ea[2][e], ea[3][e]};
        T tb[4] = {eb[0][e], eb[1][e], eb[2][e], eb[3][e]};
        *reinterpret_cast<V4*>(sA + (4 * cg + e) * LS + 64 * j + 4 * rg) = *reinterpret_cast<const V4*>(ta);
        *reinterpret_cast<V4*>(sB + (4 * cg + e) * LS + 64 * j + 4 * rg) = *reinterpret_cast<const V4*>(tb);
      }
    }
    __syncthreads();
    if (rblk + RB < r1) fetch(rblk + RB);
    // this wave's k-steps: rows 16 * (wave + 4 ks) .. +15 of the block
#pragma unroll
    for (int ks = 0; ks < NJ; ++ks) {
      const int row = 16 * (wave + 4 * ks) + 4 * g;
      V4 fa[4], fb[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (p < nto) fa[p] = *reinterpret_cast<const V4*>(sA + (16 * p + i) * LS + row);
        if (p < ntc) fb[p] = *reinterpret_cast<const V4*>(sB + (16 * p + i) * LS + row);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (p < nto && q < ntc) acc[p][q] = mma16<T>(fa[p], fb[q], acc[p][q]);
    }
    __syncthreads();
  }
  if (want_b) {   // block-uniform
#pragma unroll
    for (int e = 0; e < 4; ++e) sR[rg * 64 + 4 * cg + e] = bs[e];
    __syncthreads();
    if (threadIdx.x < 64 && o0 + (int)threadIdx.x < a.cout) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += sR[r * 64 + threadIdx.x];
      a.dbias[(int64_t)bz * a.slab_stride + o0 + threadIdx.x] = t;
    }
    __syncthreads();
  }
  // sum the four waves' partial tiles: waves 1..3 hand theirs to wave 0 through LDS, one at a time
  // acc[p][q][r] = dW[o0 + 16p + 4g + r][tap][c0 + 16q + i]
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (p < nto && q < ntc)
#pragma unroll
            for (int r = 0; r < 4; ++r) sR[(16 * p + 4 * g + r) * 64 + 16 * q + i] = acc[p][q][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (p < nto && q < ntc)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[p][q][r] += sR[(16 * p + 4 * g + r) * 64 + 16 * q + i];
    }
    __syncthreads();
  }
  if (wave != 0) return;
  float* out = a.out + (int64_t)bz * a.slab_stride;
  const int64_t ld = (int64_t)a.kvol * a.cin;
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (p < nto && q < ntc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = o0 + 16 * p + 4 * g + r, c = c0 + 16 * q + i;
          if (o < a.cout && c < ncols) out[o * ld + (a.packed ? 0 : (int64_t)by * a.cin) + c] = acc[p][q][r];
        }
}

template <typename T, int RB, int LS>
__global__ void __launch_bounds__(256) gemm_tn_kernel(TnArgs a) {
  gemm_tn_body<T, RB, LS>(a, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several dense weight gradients as ONE launch (the five linears of a block: their operands are all alive at the end of
// the block's backward).  At the deep levels a single gradient is 64 - 400 workgroups of a few rounds each, a launch that
// lasts as long as one workgroup; together they fill the chip.  Workgroup -> (problem, tile, row chunk); every problem
// keeps its own chunking and slabs, so the results are bitwise those of separate launches.
constexpr int TN_GROUP = 8;
struct TnGroup { TnArgs v[TN_GROUP]; int first[TN_GROUP]; int tiles[TN_GROUP]; int taps[TN_GROUP]; };
template <typename T, int RB, int LS>
__global__ void __launch_bounds__(256) gemm_tn_group_kernel(TnGroup grp, int count) {
  int k = 0;
#pragma unroll 1
  for (int t = 1; t < count; ++t)
    if (grp.first[t] <= (int)blockIdx.x) k = t;
  const int local = (int)blockIdx.x - grp.first[k];
  const int tt = grp.tiles[k] * grp.taps[k];     // taps: grid.y of a convolution's separate launch (1 for a linear)
  gemm_tn_body<T, RB, LS>(grp.v[k], local % grp.tiles[k], (local % tt) / grp.tiles[k], local / tt);
}

static bool tn_packed(int cin, int kvol) { return kvol > 1 && cin < 64 && 64 % cin == 0; }

static int64_t tn_chunks(int64_t m, int cout, int cin, int kvol, int64_t* rows_per_chunk) {
  // Row chunks (= fp32 slabs, summed in order afterwards).  A workgroup's time is rounds of load -> LDS -> MFMA whose
  // latency nothing inside the workgroup hides, so the chip wants MANY workgroups (tiles x taps x chunks ~ 2048, eight
  // per CU) of a few rounds each rather than a few long ones: with the earlier cap of 64 chunks the weight gradient of a
  // 100k x 96 x 32 linear ran on 126 workgroups (50 us for 26 MB of operands).  The slabs bound it from the other side:
  // every chunk writes and the reduction reads cout x kvol x cin floats, held to ~32 MB per launch.
  const int64_t tiles = tn_packed(cin, kvol) ? cdiv(cout, 64) * cdiv((int64_t)kvol * cin, 64)
                                             : cdiv(cout, 64) * cdiv(cin, 64) * kvol;
  int64_t want = cdiv(2048, tiles);
  const int64_t slab_cap = (32ll << 20) / ((int64_t)cout * cin * kvol * 4);
  if (want > slab_cap) want = slab_cap;
  if (want > 512) want = 512;
  if (want < 1) want = 1;
  int64_t rpc = cdiv(cdiv(m, want), TN_RB_MAX) * TN_RB_MAX;
  if (rpc < 2 * TN_RB_MAX) rpc = 2 * TN_RB_MAX;
  *rows_per_chunk = rpc;
  return cdiv(m, rpc);
}

// ------------------------------------------------------------------------------------------------
// Column reductions over points.  Thread = (column of a 64-wide group, one of 4 row lanes);
// block (bx, by) covers rows [bx*RB, (bx+1)*RB) x columns [64*by, 64*by+64) and writes one slab row.
//   MODE 0: sum a                      MODE 1: sum a, sum a^2
//   MODE 2: sum a, sum a * bhat  with bhat = (b - mu[c]) * rs[c]
//   MODE 3: sum a, sum (a - mu[c])^2
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void __launch_bounds__(256) col_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                          const float* __restrict__ mu, const float* __restrict__ rs,
                                                          float mu_scale, int64_t m, int c, int64_t rb,
                                                          float* __restrict__ slab) {
  constexpr int NQ = MODE == 0 ? 1 : 2;
  __shared__ float red[NQ][4][64];
  const int t = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + t;
  const int64_t r0 = (int64_t)blockIdx.x * rb, r1 = r0 + rb < m ? r0 + rb : m;
  float s0 = 0.f, s1 = 0.f;
  if (col < c) {
    float mc = 0.f, rc = 1.f;
    if (MODE == 2) { mc = mu[col] * mu_scale; rc = rs[col]; }
    if (MODE == 3) mc = mu[col] * mu_scale;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float va = to_f32<T>(a[r * c + col]);
      s0 += va;
      if (MODE == 1) s1 += va * va;
      if (MODE == 2) s1 += va * ((to_f32<T>(b[r * c + col]) - mc) * rc);
      if (MODE == 3) s1 += (va - mc) * (va - mc);
    }
  }
  red[0][rl][t] = s0;
  if (NQ == 2) red[NQ - 1][rl][t] = s1;
  __syncthreads();
  if (rl == 0 && col < c) {
    // slab row layout: [q][c]
    float* o = slab + (int64_t)blockIdx.x * NQ * c;
#pragma unroll
    for (int q = 0; q < NQ; ++q) o[q * c + col] = (red[q][0][t] + red[q][1][t]) + (red[q][2][t] + red[q][3][t]);
  }
}

static int64_t col_chunks(int64_t m, int64_t* rb) {
  // up to 256 row chunks (one workgroup each): small levels still spread over the chip
  int64_t r = cdiv(cdiv(m, 256), 4) * 4;
  if (r < 16) r = 16;
  *rb = r;
  return cdiv(m, r);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d per-channel arithmetic of a training step as ONE launch each (it was ~25 element-wise torch launches
// over (c) vectors per BatchNorm layer and step: 350 launches of the fork model's 14 layers)
//   finalize: mean = sum / m, var = centred_sq / m (biased), rstd, scale = gamma rstd, shift = beta - mean scale,
//             running_mean / running_var updated in place (momentum, unbiased variance) as torch does
//   bwd coeffs: dx = ca dy + cb x + cc  with  ca = gamma rstd, cb = -ca rstd k2, cc = ca (mean rstd k2 - k1),
//             k1 = sum(dpre) / m, k2 = sum(dpre xhat) / m
// ------------------------------------------------------------------------------------------------
__global__ void bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sq, float inv_m,
                                   float unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                   float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ scale, float* __restrict__ shift, int c) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= c) return;
  const float mu = sum[j] * inv_m, var = sq[j] * inv_m;
  const float r = rsqrtf(var + eps);
  const float sc = gamma[j] * r;
  mean[j] = mu; rstd[j] = r; scale[j] = sc; shift[j] = beta[j] - mu * sc;
  if (running_mean) {
    running_mean[j] = running_mean[j] * (1.0f - momentum) + momentum * mu;
    running_var[j] = running_var[j] * (1.0f - momentum) + momentum * (var * unbias);
  }
}

__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ sums, float inv_m, const float* __restrict__ gamma,
                                     const float* __restrict__ rstd, const float* __restrict__ mean,
                                     float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc, int c) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= c) return;
  const float k1 = sums[j] * inv_m, k2 = sums[c + j] * inv_m;
  const float a = gamma[j] * rstd[j];
  ca[j] = a;
  cb[j] = -a * rstd[j] * k2;
  cc[j] = a * (mean[j] * rstd[j] * k2 - k1);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row, lane owns columns lane, lane+64, ...  (c <= 64*NC)
//   xhat = (x - mean) * rstd;  gdy = gamma * dy
//   dx = rstd * (gdy - mean(gdy) - xhat * mean(gdy * xhat));  dgamma += dy * xhat;  dbeta += dy
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

template <typename T, int NC>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             const T* __restrict__ add,
                                                             const float* __restrict__ gamma, int64_t m, int c,
                                                             float eps, int64_t rb, T* __restrict__ dx,
                                                             float* __restrict__ slab) {
  __shared__ float red[2][4][64 * NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rb, r1 = r0 + rb < m ? r0 + rb : m;
  float gm[NC], dg[NC], db[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int col = lane + 64 * k;
    gm[k] = col < c ? gamma[col] : 0.f;
    dg[k] = db[k] = 0.f;
  }
  const float inv_c = 1.0f / (float)c;
  for (int64_t r = r0 + wave; r < r1; r += 4) {
    float xv[NC], dv[NC];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int col = lane + 64 * k;
      xv[k] = col < c ? to_f32<T>(x[r * c + col]) : 0.f;
      dv[k] = col < c ? to_f32<T>(dy[r * c + col]) : 0.f;
      s += xv[k];
    }
    const float mean = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int col = lane + 64 * k;
      const float d = col < c ? xv[k] - mean : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int col = lane + 64 * k;
      xv[k] = col < c ? (xv[k] - mean) * rstd : 0.f;  // xhat
      const float gdy = gm[k] * dv[k];
      s1 += gdy;
      s2 += gdy * xv[k];
      dg[k] += dv[k] * xv[k];
      db[k] += dv[k];
    }
    s1 = wave_sum(s1) * inv_c;
    s2 = wave_sum(s2) * inv_c;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int col = lane + 64 * k;
      if (col < c)
        dx[r * c + col] = from_f32<T>((add ? to_f32<T>(add[r * c + col]) : 0.f) + rstd * (gm[k] * dv[k] - s1 - xv[k] * s2));
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    red[0][wave][lane + 64 * k] = dg[k];
    red[1][wave][lane + 64 * k] = db[k];
  }
  __syncthreads();
  // slab row: [dgamma (c) | dbeta (c)]
  for (int j = threadIdx.x; j < 2 * c; j += 256) {
    const int q = j / c, col = j % c;
    slab[(int64_t)blockIdx.x * 2 * c + j] = (red[q][0][col] + red[q][1][col]) + (red[q][2][col] + red[q][3][col]);
  }
}

// Narrow rows (c = 32 .. 256, a power of two): a lane owns 4 CONSECUTIVE columns, c/4 lanes hold a row and a wave carries
// 256/c rows, so the loads are whole 8/16-byte pieces of one contiguous 512-byte (bf16) stretch and the four row
// reductions run over log2(c/4) shuffle steps for 256/c rows at once.  (The one-wave-per-row form above keeps half of its
// lanes idle at c = 32 and moves 2 bytes per lane per load: 67 us for 100k x 32, this form 15.)
// `add` (optional, same shape as dx): dx = add + (the LayerNorm input gradient) - the residual branch's gradient folded
// into the store instead of a separate element-wise pass.
template <typename T>
__global__ void __launch_bounds__(256) layernorm_bwd_packed_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                    const T* __restrict__ add,
                                                                    const float* __restrict__ gamma, int64_t m, int c,
                                                                    float eps, int64_t rb, T* __restrict__ dx,
                                                                    float* __restrict__ slab) {
  typedef typename Vec4<T>::type V4;
  __shared__ float red[2][1024];                 // [dgamma | dbeta][partial p = wave * rpw + rs][c] : 4 * rpw * c = 1024
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int G = c >> 2, rpw = 64 / G;
  const int gl = lane & (G - 1), rs = lane / G;
  const int64_t r0 = (int64_t)blockIdx.x * rb, r1 = r0 + rb < m ? r0 + rb : m;
  float gm[4], dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) gm[k] = gamma[4 * gl + k];
  const float inv_c = 1.0f / (float)c;
  for (int64_t base = r0; base < r1; base += 4 * rpw) {
    const int64_t r = base + wave * rpw + rs;
    const bool ok = r < r1;
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, dv[4] = {0.f, 0.f, 0.f, 0.f}, av[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      unpack4<T>(*reinterpret_cast<const V4*>(x + r * c + 4 * gl), xv);
      unpack4<T>(*reinterpret_cast<const V4*>(dy + r * c + 4 * gl), dv);
      if (add) unpack4<T>(*reinterpret_cast<const V4*>(add + r * c + 4 * gl), av);
    }
    float s = (xv[0] + xv[1]) + (xv[2] + xv[3]);
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s * inv_c;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { xv[k] -= mean; q += xv[k] * xv[k]; }
    for (int o = G >> 1; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xv[k] *= rstd;  // xhat
      const float gdy = gm[k] * dv[k];
      s1 += gdy;
      s2 += gdy * xv[k];
      dg[k] += dv[k] * xv[k];
      db[k] += dv[k];
    }
    for (int o = G >> 1; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 *= inv_c;
    s2 *= inv_c;
    if (ok) {
      float o4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o4[k] = av[k] + rstd * (gm[k] * dv[k] - s1 - xv[k] * s2);
      *reinterpret_cast<V4*>(dx + r * c + 4 * gl) = pack4<T>(o4[0], o4[1], o4[2], o4[3]);
    }
  }
  const int part = wave * rpw + rs;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][part * c + 4 * gl + k] = dg[k];
    red[1][part * c + 4 * gl + k] = db[k];
  }
  __syncthreads();
  const int parts = 4 * rpw;
  for (int j = threadIdx.x; j < 2 * c; j += 256) {
    const int q = j / c, col = j % c;
    float t = 0.f;
    for (int p2 = 0; p2 < parts; ++p2) t += red[q][p2 * c + col];
    slab[(int64_t)blockIdx.x * 2 * c + j] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// elementwise backward pieces
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == PTV3_ACT_GELU) {
    // d/dz [ z * Phi(z) ] = Phi(z) + z * phi(z)
    const float cdf = 0.5f * (1.0f + erf_fast(z * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.5f * z * z * 1.44269504088896340736f);
    return cdf + z * pdf;
  }
  if (act == PTV3_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return 1.f;
}

// dx = dy * act'(x * scale[c] + shift[c])          (scale == NULL: act'(x))
template <typename T>
__global__ void __launch_bounds__(256) act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int act, T* __restrict__ dx, int64_t total4, int c) {
  typedef typename Vec4<T>::type V4;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= total4) return;
  float d[4], v[4];
  unpack4<T>(reinterpret_cast<const V4*>(dy)[j], d);
  unpack4<T>(reinterpret_cast<const V4*>(x)[j], v);
  const int col = (int)((j * 4) % c);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float z = scale ? v[e] * scale[col + e] + shift[col + e] : v[e];
    d[e] *= act_grad(z, act);
  }
  reinterpret_cast<V4*>(dx)[j] = pack4<T>(d[0], d[1], d[2], d[3]);
}

// dx = ca[c] * dy + cb[c] * x + cc[c]     (BatchNorm input gradient with the batch statistics folded in)
template <typename T>
__global__ void __launch_bounds__(256) affine2_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                                       const float* __restrict__ cc, T* __restrict__ dx,
                                                       int64_t total4, int c) {
  typedef typename Vec4<T>::type V4;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= total4) return;
  float d[4], v[4];
  unpack4<T>(reinterpret_cast<const V4*>(dy)[j], d);
  unpack4<T>(reinterpret_cast<const V4*>(x)[j], v);
  const int col = (int)((j * 4) % c);
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = ca[col + e] * d[e] + cb[col + e] * v[e] + cc[col + e];
  reinterpret_cast<V4*>(dx)[j] = pack4<T>(d[0], d[1], d[2], d[3]);
}

// ------------------------------------------------------------------------------------------------
// SerializedPooling max backward: the first member holding the maximum of (segment, channel) receives
// the gradient, every other member 0 (each source row is a member of exactly one segment: dfeat is
// fully written, no zero-fill needed).  SerializedUnpooling gather backward = segment sum.
// ------------------------------------------------------------------------------------------------
template <typename T, bool MAX>
__global__ void __launch_bounds__(256) segment_bwd_kernel(const T* __restrict__ feat, const T* __restrict__ dy,
                                                           const int64_t* __restrict__ order0,
                                                           const int32_t* __restrict__ seg_start, int64_t n_out, int c,
                                                           T* __restrict__ out) {
  // one thread per (segment, channel); consecutive threads = consecutive channels
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_out * c) return;
  const int64_t j = t / c;
  const int ch = (int)(t % c);
  const int s0 = seg_start[j], s1 = seg_start[j + 1];
  if (MAX) {
    float mx = -INFINITY;
    int arg = s0;
    for (int p = s0; p < s1; ++p) {
      const float v = to_f32<T>(feat[order0[p] * c + ch]);
      if (v > mx) { mx = v; arg = p; }
    }
    const T g = dy[j * c + ch];
    for (int p = s0; p < s1; ++p) out[order0[p] * c + ch] = p == arg ? g : from_f32<T>(0.f);
  } else {
    float s = 0.f;
    for (int p = s0; p < s1; ++p) s += to_f32<T>(dy[order0[p] * c + ch]);
    out[j * c + ch] = from_f32<T>(s);
  }
}

}  // namespace ptv3

namespace ptv3 {
struct TnQueue {
  TnGroup grp; int count, blocks, dtype; double flops, bytes;
  const int32_t* prof_nbr; int64_t prof_pairs; double prof_pair_flops;   // the group's convolution: flops by active pairs
  int ns[TN_GROUP]; int64_t nw[TN_GROUP]; float* dw[TN_GROUP]; float* db[TN_GROUP]; float* ws[TN_GROUP];
};
static thread_local TnQueue* g_tn_defer = nullptr;
size_t tn_defer_storage_bytes() { return sizeof(TnQueue); }
void tn_defer_begin(void* storage) {   // storage: tn_defer_storage_bytes() owned by the caller; NULL ends deferral
  g_tn_defer = (TnQueue*)storage;
  if (g_tn_defer) { g_tn_defer->count = 0; g_tn_defer->blocks = 0; g_tn_defer->flops = g_tn_defer->bytes = 0.0; g_tn_defer->prof_nbr = nullptr; }
}
int tn_defer_flush(hipStream_t s) {
  TnQueue* q = g_tn_defer;
  if (!q || q->count == 0) return PTV3_OK;
  static int rbsel = -1;
  if (rbsel < 0) { const char* e = getenv("PTV3_TN_RB"); rbsel = e ? atoi(e) : 64; }
  const int prof = prof_begin(s, PROF_BACKWARD, q->flops, q->bytes, q->prof_nbr, q->prof_nbr ? q->prof_pairs : 0,
                              q->prof_nbr ? q->prof_pair_flops : 0.0);
  prof_kernel(prof, PK_GEMM_TN);
#define TNG_LAUNCH(T, RB, LS)                                                                                       \
  hipLaunchKernelGGL((gemm_tn_group_kernel<T, RB, LS>), dim3((unsigned)q->blocks), dim3(256), 2 * 64 * (LS) * sizeof(T), s, \
                     q->grp, q->count)
  if (q->dtype == PTV3_F32) TNG_LAUNCH(float, 64, 68);
  else if (rbsel == 128) TNG_LAUNCH(__bf16, 128, 196);
  else TNG_LAUNCH(__bf16, 64, 68);
#undef TNG_LAUNCH
  prof_end(prof, s);
  for (int k = 0; k < q->count; ++k)
    if (q->ns[k] > 1) {
      const int cout = q->grp.v[k].cout;
      if (q->db[k]) slab_sum(q->ws[k], q->ns[k], q->nw[k] + cout, q->dw[k], s, q->nw[k], q->db[k]);
      else slab_sum(q->ws[k], q->ns[k], q->nw[k], q->dw[k], s);
    }
  q->count = 0; q->blocks = 0; q->flops = q->bytes = 0.0; q->prof_nbr = nullptr;
  return hipGetLastError() == hipSuccess ? PTV3_OK : PTV3_ERR_LAUNCH;
}
}  // namespace ptv3

using namespace ptv3;

#define BWD_DTYPE_CHECK(name) PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, name ": bad dtype %d", dtype)

extern "C" size_t ptv3_gemm_tn_workspace_bytes(int64_t m, int cout, int cin, int kvol) {
  int64_t rpc;
  const int64_t ns = tn_chunks(m, cout, cin, kvol, &rpc);
  return ns > 1 ? (size_t)ns * ((size_t)cout * kvol * cin + cout) * sizeof(float) : 0;
}

extern "C" int ptv3_gemm_tn(const void* dy, const void* x, const int32_t* nbr, float* dw, float* dbias, int64_t m,
                            int cout, int cin, int kvol, int dtype, void* workspace, size_t workspace_bytes,
                            void* stream) {
  BWD_DTYPE_CHECK("gemm_tn");
  PTV3_REQUIRE(cout > 0 && cin > 0 && kvol >= 1 && m >= 0, "gemm_tn: bad shape m=%lld cout=%d cin=%d kvol=%d",
               (long long)m, cout, cin, kvol);
  PTV3_REQUIRE((kvol == 1) == (nbr == nullptr), "gemm_tn: nbr must be given exactly when kvol > 1");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nw = (int64_t)cout * kvol * cin;
  if (m == 0) {
    if (hipMemsetAsync(dw, 0, nw * sizeof(float), s) != hipSuccess) return PTV3_ERR_LAUNCH;
    if (dbias && hipMemsetAsync(dbias, 0, (size_t)cout * sizeof(float), s) != hipSuccess) return PTV3_ERR_LAUNCH;
    return PTV3_OK;
  }
  PTV3_REQUIRE(cout % 4 == 0 && cin % 4 == 0, "gemm_tn: cout=%d and cin=%d must be multiples of 4", cout, cin);
  int64_t rpc;
  const int64_t ns = tn_chunks(m, cout, cin, kvol, &rpc);
  PTV3_REQUIRE(workspace_bytes >= ptv3_gemm_tn_workspace_bytes(m, cout, cin, kvol), "gemm_tn: workspace too small");
  TnArgs a;
  a.dy = dy; a.x = x; a.nbr = nbr;
  // several row chunks: slab z = [dW (nw) | column sums of dy (cout)], summed in chunk order afterwards
  a.out = ns > 1 ? (float*)workspace : dw;
  a.dbias = !dbias ? nullptr : (ns > 1 ? (float*)workspace + nw : dbias);
  a.m = m; a.rows_per_chunk = rpc; a.slab_stride = ns > 1 ? nw + (dbias ? cout : 0) : 0;
  a.packed = tn_packed(cin, kvol) ? 1 : 0;
  a.cout = cout; a.cin = cin; a.kvol = kvol; a.tiles_c = (int)(a.packed ? cdiv((int64_t)kvol * cin, 64) : cdiv(cin, 64));
  dim3 grid((unsigned)(cdiv(cout, 64) * a.tiles_c), (unsigned)(a.packed ? 1 : kvol), (unsigned)ns);
  if (g_tn_defer) {   // queued: runs in tn_defer_flush() together with the other weight gradients of the block
    if (g_tn_defer->count == TN_GROUP) { const int rc = tn_defer_flush(s); if (rc != PTV3_OK) return rc; }
    TnQueue& q = *g_tn_defer;
    if (q.count == 0) q.dtype = dtype;
    PTV3_REQUIRE(q.dtype == dtype, "gemm_tn: one dtype per deferred group");
    q.grp.v[q.count] = a;
    q.grp.first[q.count] = q.blocks;
    q.grp.tiles[q.count] = (int)grid.x;
    q.grp.taps[q.count] = (int)grid.y;
    q.blocks += (int)(grid.x * grid.y * grid.z);
    if (nbr && !q.prof_nbr) { q.prof_nbr = nbr; q.prof_pairs = m * kvol; q.prof_pair_flops = 2.0 * cin * cout; }
    else q.flops += 2.0 * m * cout * (double)cin * kvol;   // (a second convolution of a group would count dense)
    q.bytes += ((double)m * (cout + cin)) * (dtype == PTV3_F32 ? 4 : 2) + (double)ns * nw * 4.0;
    q.ns[q.count] = (int)ns; q.nw[q.count] = nw; q.dw[q.count] = dw; q.db[q.count] = dbias; q.ws[q.count] = (float*)workspace;
    ++q.count;
    return PTV3_OK;
  }
  // LDS: two [64][LS] operand blocks; the 64 x 64 fp32 cross-wave reduction reuses them after the last block
  static int rbsel = -1;
  if (rbsel < 0) { const char* e = getenv("PTV3_TN_RB"); rbsel = e ? atoi(e) : 64; }
#define TN_LAUNCH(T, RB, LS)                                                                              \
  do {                                                                                                    \
    constexpr size_t lds = 2 * 64 * (LS) * sizeof(T);                                                     \
    static_assert(lds >= 64 * 64 * sizeof(float) && lds <= 64 * 1024, "gemm_tn LDS");                     \
    hipLaunchKernelGGL((gemm_tn_kernel<T, RB, LS>), grid, dim3(256), lds, s, a);                          \
  } while (0)
  // weight gradient dW = dY^T gather(X): 2 m cout kvol cin flops dense (active pairs counted on the device for a
  // conv); reads dY once and X once (gathered), writes the fp32 gradient slabs
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_BACKWARD, 2.0 * m * cout * (double)kvol * cin,
                              ((double)m * (cout + cin)) * esz + (double)ns * nw * 4.0, nbr, nbr ? m * kvol : 0,
                              2.0 * cin * cout);
  prof_kernel(prof, PK_GEMM_TN);
  if (dtype == PTV3_F32) {
    TN_LAUNCH(float, 64, 68);
  } else {
    if (rbsel == 128) TN_LAUNCH(__bf16, 128, 196); else TN_LAUNCH(__bf16, 64, 68);
  }
  prof_end(prof, s);
#undef TN_LAUNCH
  if (ns > 1) {
    if (dbias) slab_sum((const float*)workspace, (int)ns, nw + cout, dw, s, nw, dbias);
    else slab_sum((const float*)workspace, (int)ns, nw, dw, s);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" size_t ptv3_col_reduce_workspace_bytes(int64_t m, int c) {
  int64_t rb;
  return (size_t)col_chunks(m, &rb) * 2 * c * sizeof(float);
}

extern "C" int ptv3_col_reduce(const void* a, const void* b, const float* mu, const float* rs, float mu_scale, int mode,
                               float* out, int64_t m, int c, int dtype, void* workspace, size_t workspace_bytes,
                               void* stream) {
  BWD_DTYPE_CHECK("col_reduce");
  PTV3_REQUIRE(mode >= 0 && mode <= 3, "col_reduce: mode %d outside [0,3]", mode);
  PTV3_REQUIRE(mode != 3 || mu, "col_reduce: mode 3 needs mu");
  PTV3_REQUIRE(mode != 2 || (b && mu && rs), "col_reduce: mode 2 needs b, mu, rs");
  PTV3_REQUIRE(c > 0 && m >= 0, "col_reduce: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int nq = mode == 0 ? 1 : 2;
  if (m == 0) {
    if (hipMemsetAsync(out, 0, (size_t)nq * c * sizeof(float), s) != hipSuccess) return PTV3_ERR_LAUNCH;
    return PTV3_OK;
  }
  int64_t rb;
  const int64_t ns = col_chunks(m, &rb);
  PTV3_REQUIRE(workspace_bytes >= (size_t)ns * nq * c * sizeof(float), "col_reduce: workspace too small");
  dim3 grid((unsigned)ns, (unsigned)cdiv(c, 64));
#define CR_LAUNCH(T, MODE)                                                                                     \
  hipLaunchKernelGGL((col_reduce_kernel<T, MODE>), grid, dim3(256), 0, s, (const T*)a, (const T*)b, mu, rs, mu_scale, \
                     m, c, rb, (float*)workspace)
  if (dtype == PTV3_F32) {
    if (mode == 0) CR_LAUNCH(float, 0); else if (mode == 1) CR_LAUNCH(float, 1); else if (mode == 2) CR_LAUNCH(float, 2); else CR_LAUNCH(float, 3);
  } else {
    if (mode == 0) CR_LAUNCH(__bf16, 0); else if (mode == 1) CR_LAUNCH(__bf16, 1); else if (mode == 2) CR_LAUNCH(__bf16, 2); else CR_LAUNCH(__bf16, 3);
  }
#undef CR_LAUNCH
  slab_sum((const float*)workspace, (int)ns, (int64_t)nq * c, out, s);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_layernorm_bwd(const void* x, const void* dy, const void* add, const float* gamma, float eps,
                                  void* dx, float* dgamma_dbeta, int64_t m, int c, int dtype, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  BWD_DTYPE_CHECK("layernorm_bwd");
  PTV3_REQUIRE(c > 0 && c <= 1024, "layernorm_bwd: c=%d outside [1,1024]", c);
  hipStream_t s = (hipStream_t)stream;
  if (m == 0) {
    if (hipMemsetAsync(dgamma_dbeta, 0, (size_t)2 * c * sizeof(float), s) != hipSuccess) return PTV3_ERR_LAUNCH;
    return PTV3_OK;
  }
  int64_t rb;
  const int64_t ns = col_chunks(m, &rb);
  PTV3_REQUIRE(workspace_bytes >= (size_t)ns * 2 * c * sizeof(float), "layernorm_bwd: workspace too small");
  const int nc = (int)cdiv(c, 64);
  const bool packed = c >= 32 && c <= 256 && (c & (c - 1)) == 0;
#define LNB_LAUNCH(T, NC)                                                                                      \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T, NC>), dim3((unsigned)ns), dim3(256), 0, s, (const T*)x, (const T*)dy, \
                     (const T*)add, gamma, m, c, eps, rb, (T*)dx, (float*)workspace)
#define LNB_CASE(T)                                                                                                \
  if (packed)                                                                                                      \
    hipLaunchKernelGGL((layernorm_bwd_packed_kernel<T>), dim3((unsigned)ns), dim3(256), 0, s, (const T*)x,         \
                       (const T*)dy, (const T*)add, gamma, m, c, eps, rb, (T*)dx, (float*)workspace);              \
  else if (nc <= 1) LNB_LAUNCH(T, 1);               \
  else if (nc <= 2) LNB_LAUNCH(T, 2);               \
  else if (nc <= 4) LNB_LAUNCH(T, 4);               \
  else if (nc <= 8) LNB_LAUNCH(T, 8);               \
  else LNB_LAUNCH(T, 16);
  if (dtype == PTV3_F32) { LNB_CASE(float) } else { LNB_CASE(__bf16) }
#undef LNB_CASE
#undef LNB_LAUNCH
  slab_sum((const float*)workspace, (int)ns, (int64_t)2 * c, dgamma_dbeta, s);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_act_bwd(const void* dy, const void* x, const float* scale, const float* shift, int act, void* dx,
                            int64_t m, int c, int dtype, void* stream) {
  BWD_DTYPE_CHECK("act_bwd");
  PTV3_REQUIRE(c > 0 && c % 4 == 0, "act_bwd: c=%d must be a multiple of 4", c);
  PTV3_REQUIRE((scale == nullptr) == (shift == nullptr), "act_bwd: scale/shift must come together");
  if (m == 0) return PTV3_OK;
  const int64_t total4 = m * c / 4;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(total4, 256));
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL(act_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)x, scale, shift,
                       act, (float*)dx, total4, c);
  else
    hipLaunchKernelGGL(act_bwd_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)dy, (const __bf16*)x, scale,
                       shift, act, (__bf16*)dx, total4, c);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_affine2(const void* dy, const void* x, const float* ca, const float* cb, const float* cc,
                            void* dx, int64_t m, int c, int dtype, void* stream) {
  BWD_DTYPE_CHECK("affine2");
  PTV3_REQUIRE(c > 0 && c % 4 == 0, "affine2: c=%d must be a multiple of 4", c);
  if (m == 0) return PTV3_OK;
  const int64_t total4 = m * c / 4;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(total4, 256));
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL(affine2_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)x, ca, cb, cc,
                       (float*)dx, total4, c);
  else
    hipLaunchKernelGGL(affine2_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)dy, (const __bf16*)x, ca, cb, cc,
                       (__bf16*)dx, total4, c);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_pool_max_bwd(const void* feat, const void* dy, const int64_t* order0, const int32_t* seg_start,
                                 int64_t n_out, int c, void* dfeat, int dtype, void* stream) {
  BWD_DTYPE_CHECK("pool_max_bwd");
  if (n_out == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(n_out * c, 256));
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL((segment_bwd_kernel<float, true>), grid, dim3(256), 0, s, (const float*)feat, (const float*)dy,
                       order0, seg_start, n_out, c, (float*)dfeat);
  else
    hipLaunchKernelGGL((segment_bwd_kernel<__bf16, true>), grid, dim3(256), 0, s, (const __bf16*)feat,
                       (const __bf16*)dy, order0, seg_start, n_out, c, (__bf16*)dfeat);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_segment_sum(const void* dy, const int64_t* order0, const int32_t* seg_start, int64_t n_out, int c,
                                void* out, int dtype, void* stream) {
  BWD_DTYPE_CHECK("segment_sum");
  if (n_out == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(n_out * c, 256));
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL((segment_bwd_kernel<float, false>), grid, dim3(256), 0, s, (const float*)nullptr,
                       (const float*)dy, order0, seg_start, n_out, c, (float*)out);
  else
    hipLaunchKernelGGL((segment_bwd_kernel<__bf16, false>), grid, dim3(256), 0, s, (const __bf16*)nullptr,
                       (const __bf16*)dy, order0, seg_start, n_out, c, (__bf16*)out);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_bn_finalize(const float* sum, const float* centred_sq, int64_t m, const float* weight,
                                const float* bias, float* running_mean, float* running_var, float momentum, float eps,
                                float* mean, float* rstd, float* scale, float* shift, int c, void* stream) {
  PTV3_REQUIRE(m > 0 && c > 0, "bn_finalize: m=%lld c=%d", (long long)m, c);
  PTV3_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running buffers come together");
  const float unbias = (float)((double)m / (double)(m > 1 ? m - 1 : 1));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, sum, centred_sq,
                     (float)(1.0 / (double)m), unbias, weight, bias, running_mean, running_var, momentum, eps, mean, rstd,
                     scale, shift, c);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_bn_bwd_coeffs(const float* sums, int64_t m, const float* weight, const float* rstd,
                                  const float* mean, float* ca, float* cb, float* cc, int c, void* stream) {
  PTV3_REQUIRE(m > 0 && c > 0, "bn_bwd_coeffs: m=%lld c=%d", (long long)m, c);
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((unsigned)cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, sums,
                     (float)(1.0 / (double)m), weight, rstd, mean, ca, cb, cc, c);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
