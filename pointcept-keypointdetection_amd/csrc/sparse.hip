// Active-site hash table and neighbour table ("rulebook") for the submanifold convolutions.
// Replaces spconv's indice-pair generation (called lazily per indice_key by SubMConv3d in
// point_transformer_v3m1_base.py:277-284,499-506 on the tensor built in structure.py:111-146).
#include <stdlib.h>
#include "common.h"
#include "hashtable.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

__global__ void ht_insert_kernel(const int32_t* __restrict__ idx, int64_t n, unsigned long long* keys,
                                 int32_t* vals, uint64_t mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key = site_key(idx[4 * i], idx[4 * i + 1], idx[4 * i + 2], idx[4 * i + 3]);
  uint64_t slot = mix64(key) & mask;
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    unsigned long long prev = atomicCAS(&keys[slot], (unsigned long long)HT_EMPTY, (unsigned long long)key);
    if (prev == HT_EMPTY || prev == key) {
      // duplicates (not produced by GridSample) keep the smallest point index deterministically
      atomicMin(&vals[slot], (int32_t)i);  // vals start at 0x7F7F7F7F
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void ht_neighbors_kernel(const int32_t* __restrict__ idx, int64_t n,
                                    const unsigned long long* __restrict__ keys,
                                    const int32_t* __restrict__ vals, uint64_t mask, int ksize, int kvol,
                                    int32_t* __restrict__ nbr) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * kvol) return;
  int64_t i = t / kvol;
  int d = (int)(t - i * kvol);
  int c = d % ksize, b_ = (d / ksize) % ksize, a = d / (ksize * ksize);
  int half = ksize / 2;
  int x = idx[4 * i + 1] + a - half, y = idx[4 * i + 2] + b_ - half, z = idx[4 * i + 3] + c - half;
  int32_t found = -1;
  if (d == kvol / 2) {
    found = (int32_t)i;  // the centre tap is the site itself
  } else if (x >= 0 && y >= 0 && z >= 0 && x < 65536 && y < 65536 && z < 65536) {
    uint64_t key = site_key(idx[4 * i], x, y, z);
    uint64_t slot = mix64(key) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
      unsigned long long kq = keys[slot];
      if (kq == key) { found = vals[slot]; break; }
      if (kq == HT_EMPTY) break;
      slot = (slot + 1) & mask;
    }
  }
  nbr[t] = found;
}

// The neighbour relation of a submanifold convolution is symmetric: j = nbr[i][d]  <=>  i = nbr[j][kvol - 1 - d] (the
// mirrored offset).  Half the taps are probed; a hit fills both entries, the table is pre-filled with -1.  125 probes per
// site at the 5^3 stem are the longest item in front of a forward's first feature kernel (133 us at 100k sites).
__global__ void ht_neighbors_half_kernel(const int32_t* __restrict__ idx, int64_t n,
                                         const unsigned long long* __restrict__ keys, const int32_t* __restrict__ vals,
                                         uint64_t mask, int ksize, int kvol, int slots_log2,
                                         int32_t* __restrict__ nbr) {
  // a workgroup is 256 >> slots_log2 sites x 2^slots_log2 tap slots (>= kvol / 2 + 1): site and tap by shifts - the
  // flat index of the full-probing kernel costs a 64-bit division per thread, more than its probe
  const int hv = kvol / 2 + 1;
  const int64_t i = (int64_t)blockIdx.x * (256 >> slots_log2) + (threadIdx.x >> slots_log2);
  const int d = threadIdx.x & ((1 << slots_log2) - 1);
  if (i >= n || d >= hv) return;
  if (d == kvol / 2) { nbr[i * kvol + d] = (int32_t)i; return; }   // the centre tap is the site itself
  int c = d % ksize, b_ = (d / ksize) % ksize, a = d / (ksize * ksize);
  int half = ksize / 2;
  int x = idx[4 * i + 1] + a - half, y = idx[4 * i + 2] + b_ - half, z = idx[4 * i + 3] + c - half;
  if (!(x >= 0 && y >= 0 && z >= 0 && x < 65536 && y < 65536 && z < 65536)) return;
  uint64_t key = site_key(idx[4 * i], x, y, z);
  uint64_t slot = mix64(key) & mask;
  for (uint64_t probe = 0; probe <= mask; ++probe) {
    unsigned long long kq = keys[slot];
    if (kq == key) {
      const int32_t j = vals[slot];
      nbr[i * kvol + d] = j;
      nbr[(int64_t)j * kvol + (kvol - 1 - d)] = (int32_t)i;
      return;
    }
    if (kq == HT_EMPTY) return;
    slot = (slot + 1) & mask;
  }
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int64_t ptv3_subm_table_slots(int64_t n) {
  int64_t s = 1024;
  while (s < 2 * n) s <<= 1;
  return s;
}

extern "C" int ptv3_subm_build_table(const int32_t* indices, int64_t n, void* table, int64_t slots,
                                     void* stream) {
  PTV3_REQUIRE(slots >= 2 * n && (slots & (slots - 1)) == 0, "subm_build_table: slots must be a power of two >= 2n");
  hipStream_t s = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)table;
  int32_t* vals = (int32_t*)((char*)table + slots * 8);
  if (hipMemsetAsync(keys, 0xFF, (size_t)slots * 8, s) != hipSuccess) {
    set_error("subm_build_table: memset failed");
    return PTV3_ERR_LAUNCH;
  }
  if (hipMemsetAsync(vals, 0x7F, (size_t)slots * 4, s) != hipSuccess) {
    set_error("subm_build_table: memset failed");
    return PTV3_ERR_LAUNCH;
  }
  if (n == 0) return PTV3_OK;
  hipLaunchKernelGGL(ht_insert_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, indices, n, keys, vals,
                     (uint64_t)(slots - 1));
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_subm_neighbors(const int32_t* indices, int64_t n, const void* table, int64_t slots,
                                   int ksize, int32_t* nbr, void* stream) {
  PTV3_REQUIRE(ksize >= 1 && ksize <= 7 && (ksize & 1), "subm_neighbors: ksize %d must be odd and <= 7", ksize);
  if (n == 0) return PTV3_OK;
  const int kvol = ksize * ksize * ksize;
  const unsigned long long* keys = (const unsigned long long*)table;
  const int32_t* vals = (const int32_t*)((const char*)table + slots * 8);
  const char* sym = getenv("PTV3_NBR_SYMMETRIC");      // 0: every tap probed (the checker of the symmetric fill)
  if (sym && atoi(sym) == 0) {
    hipLaunchKernelGGL(ht_neighbors_kernel, dim3((unsigned)cdiv(n * kvol, 256)), dim3(256), 0,
                       (hipStream_t)stream, indices, n, keys, vals, (uint64_t)(slots - 1), ksize, kvol, nbr);
  } else {
    if (hipMemsetAsync(nbr, 0xFF, (size_t)n * kvol * sizeof(int32_t), (hipStream_t)stream) != hipSuccess) {
      set_error("subm_neighbors: memset failed");
      return PTV3_ERR_LAUNCH;
    }
    int sl = 0;
    while ((1 << sl) < kvol / 2 + 1) ++sl;          // 1: 0, 3^3: 4, 5^3: 6, 7^3: 8
    hipLaunchKernelGGL(ht_neighbors_half_kernel, dim3((unsigned)cdiv(n, 256 >> sl)), dim3(256), 0,
                       (hipStream_t)stream, indices, n, keys, vals, (uint64_t)(slots - 1), ksize, kvol, sl, nbr);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
