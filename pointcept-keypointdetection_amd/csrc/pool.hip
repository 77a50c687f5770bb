// Serialized pooling ("grid-pool scatter"): cluster ids from runs of equal parent codes in
// serialized order 0, then segmented max / mean / head gathers.
// Replaces torch.unique + torch.sort + cumsum + torch_scatter.segment_csr of
// SerializedPooling.forward (point_transformer_v3m1_base.py:384-428).
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int PS_THREADS = 256;
constexpr int PS_ITEMS = 4;
constexpr int PS_TILE = PS_THREADS * PS_ITEMS;

__device__ __forceinline__ uint32_t pool_flag(const int64_t* code0, const int64_t* order0, int64_t i, int shift) {
  if (i == 0) return 1u;
  return ((code0[order0[i]] >> shift) != (code0[order0[i - 1]] >> shift)) ? 1u : 0u;
}

// block-local inclusive scan of run-start flags; thread t owns items [t*ITEMS, (t+1)*ITEMS)
__global__ void __launch_bounds__(PS_THREADS)
pool_flag_scan_kernel(const int64_t* __restrict__ code0, const int64_t* __restrict__ order0, int64_t n,
                      int shift, uint32_t* __restrict__ local_scan, uint32_t* __restrict__ block_sum) {
  __shared__ uint32_t wsum[PS_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t base = (int64_t)blockIdx.x * PS_TILE + (int64_t)threadIdx.x * PS_ITEMS;
  uint32_t f[PS_ITEMS], tsum = 0;
#pragma unroll
  for (int e = 0; e < PS_ITEMS; ++e) {
    int64_t i = base + e;
    f[e] = i < n ? pool_flag(code0, order0, i, shift) : 0u;
    tsum += f[e];
    f[e] = tsum;  // inclusive within the thread
  }
  uint32_t x = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(x, d, 64);
    if (lane >= d) x += t;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  uint32_t excl = woff + x - tsum;
#pragma unroll
  for (int e = 0; e < PS_ITEMS; ++e) {
    int64_t i = base + e;
    if (i < n) local_scan[i] = excl + f[e];
  }
  if (threadIdx.x == PS_THREADS - 1) block_sum[blockIdx.x] = woff + x;
}

// single block: exclusive scan of block sums (in place), total -> n_out
__global__ void __launch_bounds__(1024) pool_block_scan_kernel(uint32_t* __restrict__ block_sum, int nblk,
                                                                int32_t* __restrict__ n_out) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < nblk; base += 1024) {
    int i = base + threadIdx.x;
    uint32_t v = i < nblk ? block_sum[i] : 0u, x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t t = __shfl_up(x, d, 64);
      if (lane >= d) x += t;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    uint32_t carry = carry_s;
    if (i < nblk) block_sum[i] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = (int32_t)carry_s;
}

__global__ void __launch_bounds__(PS_THREADS)
pool_assign_kernel(const int64_t* __restrict__ code0, const int64_t* __restrict__ order0, int64_t n, int shift,
                   const uint32_t* __restrict__ local_scan, const uint32_t* __restrict__ block_excl,
                   const int32_t* __restrict__ n_out, int64_t* __restrict__ cluster,
                   int32_t* __restrict__ seg_start, const int64_t* __restrict__ batch,
                   int64_t* __restrict__ pooled_offset) {
  int64_t i = (int64_t)blockIdx.x * PS_THREADS + threadIdx.x;
  if (i >= n) return;
  uint32_t cid = local_scan[i] + block_excl[i / PS_TILE] - 1u;
  const int64_t pt = order0[i];
  cluster[pt] = (int64_t)cid;
  if (pool_flag(code0, order0, i, shift)) seg_start[cid] = (int32_t)i;
  if (i == n - 1) seg_start[*n_out] = (int32_t)n;
  if (batch) {
    // codes carry the batch id in their top bits, so scenes are contiguous along order0: the last point
    // of scene b closes it at pooled row cid + 1 (cumulative "offset" of the pooled Point)
    const int64_t b = batch[pt];
    if (i == n - 1 || batch[order0[i + 1]] != b) pooled_offset[b] = (int64_t)cid + 1;
  }
}

// LPR lanes per pooled row over the channel chunks; max over the members, then folded BN + act
template <typename T, int LPR>
__global__ void __launch_bounds__(256)
pool_feat_kernel(const T* __restrict__ feat, const int64_t* __restrict__ order0,
                 const int32_t* __restrict__ seg_start, int64_t n_out, int c,
                 const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
                 T* __restrict__ out) {
  typedef typename Vec4<T>::type V4;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR;
  const int64_t j = ((int64_t)blockIdx.x * 4 + wave) * RPW + lane / LPR;
  if (j >= n_out) return;
  const int s0 = seg_start[j], s1 = seg_start[j + 1];
  const int nch = c / 4;
  for (int ch = sub; ch < nch; ch += LPR) {
    float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int p = s0; p < s1; ++p) {
      float v[4];
      unpack4<T>(*reinterpret_cast<const V4*>(feat + order0[p] * c + 4 * ch), v);
#pragma unroll
      for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = mx[e];
      if (bn_scale) t = t * bn_scale[4 * ch + e] + bn_shift[4 * ch + e];
      if (act == PTV3_ACT_GELU) t = gelu_erf(t);
      else if (act == PTV3_ACT_RELU) t = fmaxf(t, 0.f);
      mx[e] = t;
    }
    *reinterpret_cast<V4*>(out + j * c + 4 * ch) = pack4<T>(mx[0], mx[1], mx[2], mx[3]);
  }
}

struct RowPerm { int v[8]; };

__global__ void pool_meta_kernel(const float* __restrict__ coord, const int64_t* __restrict__ grid_coord,
                                 const int64_t* __restrict__ batch, const int64_t* __restrict__ code, int k,
                                 const int64_t* __restrict__ order0, const int32_t* __restrict__ seg_start,
                                 int64_t n, int64_t n_out, int pooling_depth, RowPerm perm,
                                 float* __restrict__ coord_out,
                                 int64_t* __restrict__ grid_out, int64_t* __restrict__ batch_out,
                                 int64_t* __restrict__ code_out) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_out) return;
  const int s0 = seg_start[j], s1 = seg_start[j + 1];
  const int64_t head = order0[s0];
  if (coord_out) {
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int p = s0; p < s1; ++p) {
      int64_t r = order0[p];
      sx += coord[3 * r]; sy += coord[3 * r + 1]; sz += coord[3 * r + 2];
    }
    float inv = 1.0f / (float)(s1 - s0);
    coord_out[3 * j] = sx * inv; coord_out[3 * j + 1] = sy * inv; coord_out[3 * j + 2] = sz * inv;
  }
  grid_out[3 * j] = grid_coord[3 * head] >> pooling_depth;
  grid_out[3 * j + 1] = grid_coord[3 * head + 1] >> pooling_depth;
  grid_out[3 * j + 2] = grid_coord[3 * head + 2] >> pooling_depth;
  batch_out[j] = batch[head];
  for (int r = 0; r < k; ++r) {
    const int src = perm.v[r];  // output row r takes source row perm[r] (the order shuffle of :408-412)
    code_out[(int64_t)r * n_out + j] = code[(int64_t)src * n + head] >> (3 * pooling_depth);
  }
}

}  // namespace ptv3

using namespace ptv3;

static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t ptv3_pool_workspace_bytes(int64_t n) {
  int64_t nblk = cdiv(n > 0 ? n : 1, PS_TILE);
  return al256((size_t)n * 4) + al256((size_t)nblk * 4);
}

extern "C" int ptv3_pool_segments(const int64_t* code0, const int64_t* order0, int64_t n, int shift_bits,
                                  const int64_t* batch, int64_t* cluster, int32_t* seg_start, int32_t* n_out,
                                  int64_t* pooled_offset, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  PTV3_REQUIRE((batch == nullptr) == (pooled_offset == nullptr), "pool_segments: batch and pooled_offset come together");
  PTV3_REQUIRE(n >= 1 && n < (1ll << 31), "pool_segments: n=%lld", (long long)n);
  PTV3_REQUIRE(shift_bits >= 0 && shift_bits < 63, "pool_segments: shift_bits");
  PTV3_REQUIRE(workspace_bytes >= ptv3_pool_workspace_bytes(n), "pool_segments: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nblk = (int)cdiv(n, PS_TILE);
  uint32_t* local_scan = (uint32_t*)workspace;
  uint32_t* bsum = (uint32_t*)((char*)workspace + al256((size_t)n * 4));
  hipLaunchKernelGGL(pool_flag_scan_kernel, dim3(nblk), dim3(PS_THREADS), 0, s, code0, order0, n, shift_bits,
                     local_scan, bsum);
  hipLaunchKernelGGL(pool_block_scan_kernel, dim3(1), dim3(1024), 0, s, bsum, nblk, n_out);
  hipLaunchKernelGGL(pool_assign_kernel, dim3((unsigned)cdiv(n, PS_THREADS)), dim3(PS_THREADS), 0, s, code0,
                     order0, n, shift_bits, local_scan, bsum, n_out, cluster, seg_start, batch, pooled_offset);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

template <typename T>
static int launch_pool_feat(const void* feat, const int64_t* order0, const int32_t* seg_start, int64_t n_out,
                            int c, const float* bn_scale, const float* bn_shift, int act, void* out,
                            hipStream_t s) {
  const int nch = c / 4;
  int lpr = 1;
  while (lpr < 64 && lpr < nch) lpr <<= 1;
#define PF_LAUNCH(L)                                                                                          \
  hipLaunchKernelGGL((pool_feat_kernel<T, L>), dim3((unsigned)cdiv(n_out, 4 * (64 / L))), dim3(256), 0, s,   \
                     (const T*)feat, order0, seg_start, n_out, c, bn_scale, bn_shift, act, (T*)out);
  switch (lpr) {
    case 1: PF_LAUNCH(1) break;
    case 2: PF_LAUNCH(2) break;
    case 4: PF_LAUNCH(4) break;
    case 8: PF_LAUNCH(8) break;
    case 16: PF_LAUNCH(16) break;
    case 32: PF_LAUNCH(32) break;
    default: PF_LAUNCH(64) break;
  }
#undef PF_LAUNCH
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_pool_reduce(const void* feat, const float* coord, const int64_t* grid_coord,
                                const int64_t* batch, const int64_t* code, int k, const int64_t* order0,
                                const int32_t* seg_start, int64_t n, int64_t n_out, int c, int pooling_depth,
                                const float* bn_scale, const float* bn_shift, int act, const int* row_perm_host,
                                void* feat_out, float* coord_out, int64_t* grid_out, int64_t* batch_out,
                                int64_t* code_out, int dtype, void* stream) {
  PTV3_REQUIRE(k >= 1 && k <= 8, "pool_reduce: k=%d outside [1,8]", k);
  RowPerm perm;
  for (int r = 0; r < 8; ++r) perm.v[r] = r;
  if (row_perm_host)
    for (int r = 0; r < k; ++r) {
      PTV3_REQUIRE(row_perm_host[r] >= 0 && row_perm_host[r] < k, "pool_reduce: bad row permutation");
      perm.v[r] = row_perm_host[r];
    }
  PTV3_REQUIRE(c > 0 && c % 4 == 0, "pool_reduce: c=%d must be a multiple of 4", c);
  PTV3_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "pool_reduce: bn_scale/bn_shift must come together");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "pool_reduce: bad dtype");
  if (n_out == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  if (feat != nullptr) {  // feature half (segmented max + folded BN + act)
    int rc = dtype == PTV3_F32
                 ? launch_pool_feat<float>(feat, order0, seg_start, n_out, c, bn_scale, bn_shift, act, feat_out, s)
                 : launch_pool_feat<__bf16>(feat, order0, seg_start, n_out, c, bn_scale, bn_shift, act, feat_out, s);
    if (rc) return rc;
  }
  if (grid_coord != nullptr) {  // geometry half (coord mean, head gathers, pooled codes)
    hipLaunchKernelGGL(pool_meta_kernel, dim3((unsigned)cdiv(n_out, 256)), dim3(256), 0, s, coord, grid_coord, batch,
                       code, k, order0, seg_start, n, n_out, pooling_depth, perm, coord_out, grid_out, batch_out,
                       code_out);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
