// Space-filling-curve codes, stable LSD radix argsort (+ inverse), pad plan, window maps.
// HBM-bound integer work: one pass over coordinates for the codes, (passes x 3) short kernels
// for the sort.  Reference semantics: include/ptv3_hip.h.
#include "common.h"
#include <stdlib.h>
#include "../../include/ptv3_hip.h"

namespace ptv3 {

// ------------------------------------------------------------------------------------------
// curve codes
// ------------------------------------------------------------------------------------------
// spread the low 16 bits of v so that bit i lands on bit 3i
__device__ __forceinline__ uint64_t spread3(uint64_t v) {
  v &= 0xFFFFull;
  v = (v | (v << 32)) & 0x00FF00000000FFFFull;  // not needed for 16 bits but keeps the ladder regular
  v = (v | (v << 16)) & 0x00FF0000FF0000FFull;
  v = (v | (v << 8)) & 0xF00F00F00F00F00Full;
  v = (v | (v << 4)) & 0x30C30C30C30C30C3ull;
  v = (v | (v << 2)) & 0x9249249249249249ull;
  return v;
}

__device__ __forceinline__ uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
}

// Skilling's AxesToTranspose on 3 axes, then Gray -> binary over the interleaved 3*depth bits.
__device__ __forceinline__ uint64_t hilbert3(uint32_t x, uint32_t y, uint32_t z, int depth) {
  uint32_t X0 = x, X1 = y, X2 = z;
  for (uint32_t Q = 1u << (depth - 1); Q > 0; Q >>= 1) {  // every bit plane, MSB first
    uint32_t P = Q - 1;
    // dim 0
    if (X0 & Q) X0 ^= P;
    // dim 1
    if (X1 & Q) X0 ^= P;
    else { uint32_t t = (X0 ^ X1) & P; X0 ^= t; X1 ^= t; }
    // dim 2
    if (X2 & Q) X0 ^= P;
    else { uint32_t t = (X0 ^ X2) & P; X0 ^= t; X2 ^= t; }
  }
  uint64_t g = morton3(X0, X1, X2);
  // prefix xor from the MSB (Gray -> binary); 48 bits at most
  g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16; g ^= g >> 32;
  return g;
}

struct OrderIds { int id[8]; };

template <typename CoordT>
__global__ void sfc_encode_kernel(const CoordT* __restrict__ gc, const int64_t* __restrict__ batch,
                                  int64_t n, int depth, OrderIds ids, int k, int64_t* __restrict__ code) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t mask = depth >= 32 ? 0xFFFFFFFFu : ((1u << depth) - 1u);
  uint32_t x = (uint32_t)gc[i * 3 + 0] & mask;
  uint32_t y = (uint32_t)gc[i * 3 + 1] & mask;
  uint32_t z = (uint32_t)gc[i * 3 + 2] & mask;
  uint64_t b = batch ? ((uint64_t)batch[i] << (3 * depth)) : 0ull;
  for (int r = 0; r < k; ++r) {
    int id = ids.id[r];
    uint32_t a = (id & 1) ? y : x;  // "-trans": swap x and y
    uint32_t c = (id & 1) ? x : y;
    uint64_t key = (id & 2) ? hilbert3(a, c, z, depth) : morton3(a, c, z);
    code[(int64_t)r * n + i] = (int64_t)(b | key);
  }
}

// ------------------------------------------------------------------------------------------
// stable LSD radix argsort, 8-bit digits, rows sorted independently (blockIdx.y = row)
// ------------------------------------------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 8;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // keys per block
constexpr int RS_WAVES = RS_THREADS / 64;

__global__ void __launch_bounds__(RS_THREADS)
radix_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist,
                  int nblk) {
  __shared__ uint32_t h[256];
  const int row = blockIdx.y;
  const uint64_t* k = keys + (int64_t)row * n;
  h[threadIdx.x] = 0;
  __syncthreads();
  int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int it = 0; it < RS_ITEMS; ++it) {
    int64_t i = base + it * RS_THREADS + threadIdx.x;
    if (i < n) atomicAdd(&h[(k[i] >> shift) & 255], 1u);
  }
  __syncthreads();
  // block-major [row][block][digit]: coalesced here, in the scan and in the scatter
  hist[((int64_t)row * nblk + blockIdx.x) * 256 + threadIdx.x] = h[threadIdx.x];
}

// one 1024-thread block per row: exclusive scan, in (digit, block) order, of the [block][digit] counters.
// Thread (part, d) owns digit d of a quarter of the blocks: sums its counters (eight loads in flight), the 256 digit
// totals are scanned across the first four waves, then the thread rewrites its counters as running bases.
// (One thread per digit walking all blocks twice was a chain of 2 nblk dependent round trips: 110 us per pass for
// the 489 blocks of a 1M-voxel Swin3D level, 6 ms of its 96 ms forward.)
constexpr int RSC_PARTS = 4;
__global__ void __launch_bounds__(256 * RSC_PARTS) radix_scan_kernel(uint32_t* __restrict__ hist, int nblk) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t ptot[RSC_PARTS][256];
  __shared__ uint32_t dbase[256];
  const int d = threadIdx.x & 255, part = threadIdx.x >> 8;
  uint32_t* h = hist + (int64_t)blockIdx.x * nblk * 256 + d;  // stride 256 between blocks
  const int q = (nblk + RSC_PARTS - 1) / RSC_PARTS;
  const int b0 = min(part * q, nblk), b1 = min(b0 + q, nblk);
  uint32_t total = 0;
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = h[(int64_t)(b + u) * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) total += v[u];
  }
  for (; b < b1; ++b) total += h[(int64_t)b * 256];
  ptot[part][d] = total;
  __syncthreads();
  if (part == 0) {
    const int lane = d & 63, wave = d >> 6;
    uint32_t tot = 0;
#pragma unroll
    for (int p = 0; p < RSC_PARTS; ++p) tot += ptot[p][d];
    uint32_t x = tot;
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) {
      uint32_t t = __shfl_up(x, k, 64);
      if (lane >= k) x += t;
    }
    if (lane == 63) wsum[wave] = x;
    dbase[d] = x - tot;        // exclusive within the wave; the wave offsets are added below
  }
  __syncthreads();
  uint32_t base = dbase[d];
  for (int w = 0; w < (d >> 6); ++w) base += wsum[w];
  for (int p = 0; p < part; ++p) base += ptot[p][d];
  b = b0;
  for (; b + 8 <= b1; b += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = h[(int64_t)(b + u) * 256];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h[(int64_t)(b + u) * 256] = base;
      base += v[u];
    }
  }
  for (; b < b1; ++b) {
    const uint32_t c = h[(int64_t)b * 256];
    h[(int64_t)b * 256] = base;
    base += c;
  }
}

// FIRST: values are the implicit iota; LAST: writes int64 order + inverse.
// SCAN: `hist` still holds the raw per-block digit counts and every block derives its own digit bases from them
// (digit totals over all blocks, scanned across the 256 threads, plus the counts of the blocks before it) - the
// separate scan launch disappears.  Each block reads nblk*256 counters, so the host only picks this form for
// nblk <= RS_FUSED_SCAN_MAX_BLOCKS (every level of a 100k-point scene: nblk <= 49).
constexpr int RS_FUSED_SCAN_MAX_BLOCKS = 128;

template <bool FIRST, bool LAST, bool SCAN>
__global__ void __launch_bounds__(RS_THREADS)
radix_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                     uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                     int64_t* __restrict__ order, int64_t* __restrict__ inverse, int64_t n, int shift,
                     const uint32_t* __restrict__ hist, int nblk) {
  __shared__ uint32_t cnt[RS_WAVES][256];  // running per-wave digit counts, then wave bases
  __shared__ uint32_t gbase[256];
  __shared__ uint32_t wsum[RS_WAVES];
  if (SCAN) {
    const uint32_t* hr = hist + (int64_t)blockIdx.y * nblk * 256 + threadIdx.x;   // digit = threadIdx.x
    uint32_t total = 0, before = 0;
    for (int b = 0; b < nblk; ++b) {
      const uint32_t c = hr[(int64_t)b * 256];
      total += c;
      if (b < (int)blockIdx.x) before += c;
    }
    uint32_t x = total;
    const int ln = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(x, d, 64);
      if (ln >= d) x += t;
    }
    if (ln == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    uint32_t excl = x - total;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) excl += wsum[w];
    gbase[threadIdx.x] = excl + before;
  }
  const int row = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t* kin = keys_in + (int64_t)row * n;
  for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();

  // wave w owns the contiguous run [base + w*ITEMS*64, +ITEMS*64); round r takes 64 consecutive keys
  const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (RS_ITEMS * 64);
  uint64_t key[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    int64_t i = wbase + r * 64 + lane;
    bool valid = i < n;
    key[r] = valid ? kin[i] : ~0ull;
    uint32_t d = valid ? (uint32_t)((key[r] >> shift) & 255) : 256u;
    // lanes of this wave holding the same digit
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      uint64_t m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    uint32_t before = __popcll(peers & ((1ull << lane) - 1ull));
    uint32_t prev = 0;
    if (valid) prev = cnt[wave][d];
    rank[r] = prev + before;
    // every peer read the counter before the leader updates it: same wave => lockstep, but make
    // the LDS ordering explicit
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // per digit: exclusive prefix over waves -> base of wave w inside the tile
  {
    uint32_t run = 0;
    int d = threadIdx.x;  // RS_THREADS == 256 digits
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      uint32_t c = cnt[w][d];
      cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
  const uint32_t* hrow = hist + ((int64_t)row * nblk + blockIdx.x) * 256;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    int64_t i = wbase + r * 64 + lane;
    if (i < n) {
      uint32_t d = (uint32_t)((key[r] >> shift) & 255);
      int64_t pos = (int64_t)(SCAN ? gbase[d] : hrow[d]) + cnt[wave][d] + rank[r];
      uint32_t v = FIRST ? (uint32_t)i : vals_in[(int64_t)row * n + i];
      if (LAST) {
        order[(int64_t)row * n + pos] = (int64_t)v;
        inverse[(int64_t)row * n + v] = pos;
      } else {
        keys_out[(int64_t)row * n + pos] = key[r];
        vals_out[(int64_t)row * n + pos] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// pad plan + window maps
// ------------------------------------------------------------------------------------------
constexpr int PAD_MAX_SCENES = 64;

__global__ void pad_plan_kernel(const int64_t* __restrict__ offset, int b, int64_t n, int64_t n_pad, int K,
                                int64_t* __restrict__ pad, int64_t* __restrict__ unpad,
                                int32_t* __restrict__ cu) {
  // scene table in LDS: start, padded start, count (b is small: one entry per scene of the batch)
  extern __shared__ int64_t tab[];
  int64_t* s_off = tab;             // b+1
  int64_t* s_offp = tab + (b + 1);  // b+1
  if (threadIdx.x == 0) {
    int64_t prev = 0, prevp = 0;
    s_off[0] = 0; s_offp[0] = 0;
    for (int i = 0; i < b; ++i) {
      int64_t cnt = offset[i] - prev;
      int64_t cp = cnt > K ? (cnt + K - 1) / K * K : cnt;
      prev = offset[i]; prevp += cp;
      s_off[i + 1] = prev; s_offp[i + 1] = prevp;
    }
  }
  __syncthreads();
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_pad) {
    int lo = 0, hi = b;  // largest i with s_offp[i] <= p
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_offp[mid] <= p) lo = mid; else hi = mid; }
    int64_t cnt = s_off[lo + 1] - s_off[lo];
    int64_t cntp = s_offp[lo + 1] - s_offp[lo];
    int64_t local = p - s_offp[lo];
    int64_t src = local;
    if (cnt != cntp) {
      int64_t r = cnt % K;
      if (local >= cntp - K + r) src = local - K;  // borrow the tail of the previous window
    }
    pad[p] = s_off[lo] + src;
  }
  if (p < n) {
    int lo = 0, hi = b;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_off[mid] <= p) lo = mid; else hi = mid; }
    unpad[p] = p + (s_offp[lo] - s_off[lo]);
  }
  // cu_seqlens: concat_i arange(offp_i, offp_{i+1}, K) then n_pad
  if (p <= n_pad && cu) {
    // number of entries = sum_i ceil(cntp_i / K) + 1 ; entry e sits at scene-relative multiples of K
    // enumerate by padded position: p is an entry iff (p - s_offp[scene]) % K == 0 (or p == n_pad)
    if (p == n_pad) {
      int64_t e = 0;
      for (int i = 0; i < b; ++i) e += (s_offp[i + 1] - s_offp[i] + K - 1) / K;
      cu[e] = (int32_t)n_pad;
    } else {
      int lo = 0, hi = b;
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_offp[mid] <= p) lo = mid; else hi = mid; }
      int64_t local = p - s_offp[lo];
      if (local % K == 0) {
        int64_t e = local / K;
        for (int i = 0; i < lo; ++i) e += (s_offp[i + 1] - s_offp[i] + K - 1) / K;
        cu[e] = (int32_t)p;
      }
    }
  }
}

__global__ void window_maps_kernel(const int64_t* __restrict__ order, const int64_t* __restrict__ inverse,
                                   const int64_t* __restrict__ pad, const int64_t* __restrict__ unpad,
                                   int64_t n, int64_t n_pad, int32_t* __restrict__ wo,
                                   int32_t* __restrict__ wi) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) wo[i] = (int32_t)order[pad[i]];
  if (i < n) wi[i] = (int32_t)unpad[inverse[i]];
}

// pad plan and both window maps for all k orders in one pass, without materialising pad / unpad
__global__ void window_plan_kernel(const int64_t* __restrict__ order, const int64_t* __restrict__ inverse,
                                   const int64_t* __restrict__ offset, int b, int k, int64_t n, int64_t n_pad,
                                   int K, int32_t* __restrict__ wo, int32_t* __restrict__ wi,
                                   int32_t* __restrict__ cu) {
  extern __shared__ int64_t tab[];
  int64_t* s_off = tab;
  int64_t* s_offp = tab + (b + 1);
  if (threadIdx.x == 0) {
    int64_t prev = 0, prevp = 0;
    s_off[0] = 0; s_offp[0] = 0;
    for (int i = 0; i < b; ++i) {
      int64_t cnt = offset[i] - prev;
      int64_t cp = cnt > K ? (cnt + K - 1) / K * K : cnt;
      prev = offset[i]; prevp += cp;
      s_off[i + 1] = prev; s_offp[i + 1] = prevp;
    }
  }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n_pad) {
    int lo = 0, hi = b;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_offp[mid] <= p) lo = mid; else hi = mid; }
    const int64_t cnt = s_off[lo + 1] - s_off[lo], cntp = s_offp[lo + 1] - s_offp[lo];
    const int64_t local = p - s_offp[lo];
    int64_t src = local;
    if (cnt != cntp && local >= cntp - K + cnt % K) src = local - K;
    const int64_t pos = s_off[lo] + src;  // = pad[p]
    for (int r = 0; r < k; ++r) wo[(int64_t)r * n_pad + p] = (int32_t)order[(int64_t)r * n + pos];
  }
  if (p < n) {
    for (int r = 0; r < k; ++r) {
      const int64_t j = inverse[(int64_t)r * n + p];  // serialized position of point p in order r
      int lo = 0, hi = b;
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_off[mid] <= j) lo = mid; else hi = mid; }
      wi[(int64_t)r * n + p] = (int32_t)(j + (s_offp[lo] - s_off[lo]));  // = unpad[j]
    }
  }
  if (cu && p <= n_pad) {  // window starts: concat_i arange(offp_i, offp_{i+1}, K), then n_pad (:155-169)
    if (p == n_pad) {
      int64_t e = 0;
      for (int i = 0; i < b; ++i) e += (s_offp[i + 1] - s_offp[i] + K - 1) / K;
      cu[e] = (int32_t)n_pad;
    } else {
      int lo = 0, hi = b;
      while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (s_offp[mid] <= p) lo = mid; else hi = mid; }
      const int64_t local = p - s_offp[lo];
      if (local % K == 0) {
        int64_t e = local / K;
        for (int i = 0; i < lo; ++i) e += (s_offp[i + 1] - s_offp[i] + K - 1) / K;
        cu[e] = (int32_t)p;
      }
    }
  }
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_window_plan(const int64_t* order, const int64_t* inverse, const int64_t* offset, int b, int k,
                                int64_t n, int64_t n_pad, int patch, int32_t* win_order, int32_t* win_inverse,
                                int32_t* cu_seqlens, void* stream) {
  PTV3_REQUIRE(b >= 1 && b <= 4096, "window_plan: batch size %d outside [1,4096]", b);
  PTV3_REQUIRE(k >= 1 && k <= 8 && patch >= 1, "window_plan: bad k / patch");
  int64_t work = (n_pad > n ? n_pad : n) + (cu_seqlens ? 1 : 0);
  if (work == 0) return PTV3_OK;
  hipLaunchKernelGGL(window_plan_kernel, dim3((unsigned)cdiv(work, 256)), dim3(256),
                     (size_t)2 * (b + 1) * sizeof(int64_t), (hipStream_t)stream, order, inverse, offset, b, k, n, n_pad,
                     patch, win_order, win_inverse, cu_seqlens);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_sfc_encode(const void* grid_coord, int coord_is_i64, const int64_t* batch, int64_t n,
                               int depth, const int* order_ids_host, int k, int64_t* code, void* stream) {
  PTV3_REQUIRE(depth >= 1 && depth <= 16, "sfc_encode: depth %d outside [1,16]", depth);
  PTV3_REQUIRE(k >= 1 && k <= 8, "sfc_encode: k=%d outside [1,8]", k);
  if (n == 0) return PTV3_OK;
  OrderIds ids;
  for (int r = 0; r < k; ++r) {
    PTV3_REQUIRE(order_ids_host[r] >= 0 && order_ids_host[r] <= 3, "sfc_encode: bad order id");
    ids.id[r] = order_ids_host[r];
  }
  dim3 grid((unsigned)cdiv(n, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (coord_is_i64)
    hipLaunchKernelGGL(sfc_encode_kernel<int64_t>, grid, block, 0, s, (const int64_t*)grid_coord, batch, n,
                       depth, ids, k, code);
  else
    hipLaunchKernelGGL(sfc_encode_kernel<int32_t>, grid, block, 0, s, (const int32_t*)grid_coord, batch, n,
                       depth, ids, k, code);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t ptv3_argsort_workspace_bytes(int k, int64_t n) {
  int64_t nblk = cdiv(n > 0 ? n : 1, RS_TILE);
  size_t keys = align256((size_t)k * n * 8), vals = align256((size_t)k * n * 4);
  size_t hist = align256((size_t)k * 256 * nblk * 4);
  return 2 * keys + 2 * vals + hist;
}

extern "C" int ptv3_argsort_i64(const int64_t* code, int k, int64_t n, int end_bit, int64_t* order,
                                int64_t* inverse, void* workspace, size_t workspace_bytes, void* stream) {
  PTV3_REQUIRE(end_bit >= 1 && end_bit <= 64, "argsort: end_bit %d outside [1,64]", end_bit);
  PTV3_REQUIRE(n < (1ll << 31), "argsort: n too large");
  PTV3_REQUIRE(workspace_bytes >= ptv3_argsort_workspace_bytes(k, n), "argsort: workspace too small");
  if (n == 0 || k == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)cdiv(n, RS_TILE);
  char* ws = (char*)workspace;
  size_t keys_b = align256((size_t)k * n * 8), vals_b = align256((size_t)k * n * 4);
  uint64_t* kbuf[2] = {(uint64_t*)ws, (uint64_t*)(ws + keys_b)};
  uint32_t* vbuf[2] = {(uint32_t*)(ws + 2 * keys_b), (uint32_t*)(ws + 2 * keys_b + vals_b)};
  uint32_t* hist = (uint32_t*)(ws + 2 * keys_b + 2 * vals_b);
  const int passes = (end_bit + 7) / 8;
  dim3 grid(nblk, k), block(RS_THREADS);
  const uint64_t* kin = (const uint64_t*)code;
  const uint32_t* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const int shift = 8 * p;
    const bool first = p == 0, last = p == passes - 1;
    hipLaunchKernelGGL(radix_hist_kernel, grid, block, 0, s, kin, n, shift, hist, nblk);
    static const bool allow_fused = getenv("PTV3_RADIX_SEPARATE_SCAN") == nullptr;
    const bool fused = allow_fused && nblk <= RS_FUSED_SCAN_MAX_BLOCKS;
    if (!fused) hipLaunchKernelGGL(radix_scan_kernel, dim3(k), dim3(256 * RSC_PARTS), 0, s, hist, nblk);
    uint64_t* kout = kbuf[p & 1];
    uint32_t* vout = vbuf[p & 1];
#define RS_SCATTER(F, L)                                                                                          \
    if (fused) hipLaunchKernelGGL((radix_scatter_kernel<F, L, true>), grid, block, 0, s, kin, vin, kout, vout, order, \
                                  inverse, n, shift, hist, nblk);                                                 \
    else hipLaunchKernelGGL((radix_scatter_kernel<F, L, false>), grid, block, 0, s, kin, vin, kout, vout, order,  \
                            inverse, n, shift, hist, nblk)
    if (first && last) { RS_SCATTER(true, true); }
    else if (first) { RS_SCATTER(true, false); }
    else if (last) { RS_SCATTER(false, true); }
    else { RS_SCATTER(false, false); }
#undef RS_SCATTER
    kin = kout;
    vin = vout;
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_pad_plan(const int64_t* offset, int b, int64_t n, int64_t n_pad, int patch, int64_t* pad,
                             int64_t* unpad, int32_t* cu_seqlens, void* stream) {
  PTV3_REQUIRE(b >= 1 && b <= 4096, "pad_plan: batch size %d outside [1,4096]", b);
  PTV3_REQUIRE(patch >= 1, "pad_plan: patch must be >= 1");
  int64_t work = (n_pad > n ? n_pad : n) + 1;
  size_t lds = (size_t)2 * (b + 1) * sizeof(int64_t);
  hipLaunchKernelGGL(pad_plan_kernel, dim3((unsigned)cdiv(work, 256)), dim3(256), lds, (hipStream_t)stream,
                     offset, b, n, n_pad, patch, pad, unpad, cu_seqlens);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_window_maps(const int64_t* order, const int64_t* inverse, const int64_t* pad,
                                const int64_t* unpad, int64_t n, int64_t n_pad, int32_t* win_order,
                                int32_t* win_inverse, void* stream) {
  int64_t work = n_pad > n ? n_pad : n;
  if (work == 0) return PTV3_OK;
  hipLaunchKernelGGL(window_maps_kernel, dim3((unsigned)cdiv(work, 256)), dim3(256), 0, (hipStream_t)stream,
                     order, inverse, pad, unpad, n, n_pad, win_order, win_inverse);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
