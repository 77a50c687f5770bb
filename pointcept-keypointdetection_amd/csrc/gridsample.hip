// GridSample front half on the device (pointcept/datasets/transform.py:848-860, 926-964): voxel coordinate
// = floor(coord / grid_size) in float64 exactly as numpy evaluates it, per-cloud minimum subtracted, then the
// 64-bit voxel key (FNV-1a or ravel).  The key sort / unique / pick that follow reuse ptv3_argsort_i64 and
// ptv3_pool_segments.  Integer work, HBM-bound: 12 B in, 32 B out per point.
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

__global__ void gs_init_kernel(int64_t* mn, int64_t* mx) {
  if (threadIdx.x < 3) { mn[threadIdx.x] = INT64_MAX; mx[threadIdx.x] = INT64_MIN; }
}

__global__ void __launch_bounds__(256) gs_floor_kernel(const float* __restrict__ coord, int64_t n, double grid_size,
                                                        int64_t* __restrict__ grid, int64_t* __restrict__ mn,
                                                        int64_t* __restrict__ mx) {
  __shared__ long long smn[3][4], smx[3][4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  long long lo[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, hi[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
  if (i < n) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const long long g = (long long)floor((double)coord[3 * i + d] / grid_size);
      grid[3 * i + d] = g;
      lo[d] = hi[d] = g;
    }
  }
  // wave then block reduction (integer min/max: order-free, deterministic)
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int s = 1; s < 64; s <<= 1) {
      lo[d] = min(lo[d], (long long)__shfl_xor(lo[d], s, 64));
      hi[d] = max(hi[d], (long long)__shfl_xor(hi[d], s, 64));
    }
    if ((threadIdx.x & 63) == 0) { smn[d][threadIdx.x >> 6] = lo[d]; smx[d][threadIdx.x >> 6] = hi[d]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int d = threadIdx.x;
    long long a = min(min(smn[d][0], smn[d][1]), min(smn[d][2], smn[d][3]));
    long long b = max(max(smx[d][0], smx[d][1]), max(smx[d][2], smx[d][3]));
    atomicMin(reinterpret_cast<long long*>(mn + d), a);
    atomicMax(reinterpret_cast<long long*>(mx + d), b);
  }
}

__global__ void __launch_bounds__(256) gs_key_kernel(int64_t* __restrict__ grid, int64_t n,
                                                      const int64_t* __restrict__ mn, const int64_t* __restrict__ mx,
                                                      int hash_type, uint64_t* __restrict__ key) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t g[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int64_t v = grid[3 * i + d] - mn[d];
    grid[3 * i + d] = v;
    g[d] = (uint64_t)v;
  }
  uint64_t h;
  if (hash_type == PTV3_HASH_FNV) {
    h = 14695981039346656037ull;  // FNV64-1A over the three coordinates (transform.py:949-964)
#pragma unroll
    for (int d = 0; d < 3; ++d) { h *= 1099511628211ull; h ^= g[d]; }
  } else {
    // ravel (transform.py:926-946): ((x * (max_y + 1)) + y) * (max_z + 1) + z on min-subtracted coordinates
    const uint64_t my = (uint64_t)(mx[1] - mn[1]) + 1, mz = (uint64_t)(mx[2] - mn[2]) + 1;
    h = (g[0] * my + g[1]) * mz + g[2];
  }
  key[i] = h;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_grid_hash(const float* coord, int64_t n, double grid_size, int hash_type, int64_t* grid_coord,
                              int64_t* min_max, uint64_t* key, void* stream) {
  PTV3_REQUIRE(grid_size > 0.0, "grid_hash: grid_size must be positive");
  PTV3_REQUIRE(hash_type == PTV3_HASH_FNV || hash_type == PTV3_HASH_RAVEL, "grid_hash: bad hash type %d", hash_type);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs_init_kernel, dim3(1), dim3(64), 0, s, min_max, min_max + 3);
  if (n > 0) {
    hipLaunchKernelGGL(gs_floor_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, coord, n, grid_size, grid_coord,
                       min_max, min_max + 3);
    hipLaunchKernelGGL(gs_key_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, grid_coord, n, min_max,
                       min_max + 3, hash_type, key);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
