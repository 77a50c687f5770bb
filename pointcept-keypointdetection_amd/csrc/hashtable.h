// Open-addressing table of active sites (batch, x, y, z) -> row: key / probe functions shared by the submanifold
// neighbour search (sparse.hip) and the cell-grid kNN (pointops.hip).  Layout of a table of `slots` entries:
// slots x uint64 keys, then slots x int32 values.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace ptv3 {

constexpr uint64_t HT_EMPTY = ~0ull;

__device__ __forceinline__ uint64_t site_key(int b, int x, int y, int z) {
  return ((uint64_t)(uint32_t)b << 48) | ((uint64_t)(uint32_t)x << 32) | ((uint64_t)(uint32_t)y << 16) |
         (uint64_t)(uint32_t)z;
}
__device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}
// row of site (b, x, y, z), or -1
__device__ __forceinline__ int32_t ht_find(const unsigned long long* __restrict__ keys,
                                           const int32_t* __restrict__ vals, uint64_t mask, int b, int x, int y, int z) {
  if (x < 0 || y < 0 || z < 0 || x >= 65536 || y >= 65536 || z >= 65536) return -1;
  const uint64_t key = site_key(b, x, y, z);
  uint64_t slot = mix64(key) & mask;
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    const unsigned long long kq = keys[slot];
    if (kq == key) return vals[slot];
    if (kq == HT_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

}  // namespace ptv3
