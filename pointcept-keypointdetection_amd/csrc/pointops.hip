// libs/pointops knn_query / grouping / interpolation as wavefront-primitive HIP.
// knn: one 64-lane wave per group of queries; candidates are scanned 64 at a time (one coalesced load per chunk,
// shared by the queries of the group), a ballot finds the few that beat the current k-th distance.  The running
// k-best of a query is THE REFERENCE'S BINARY MAX-HEAP, held across the lanes of the wave (heap slot e in lane e & 63,
// register e >> 6): an accepted candidate replaces the root and sinks exactly as reheap() does
// (knn_query_cuda_kernel.cu:15-30: larger child, the left one on a tie; the newcomer keeps sinking while it is
// <= that child), with wave-uniform (scalar) control flow and v_readlane for the few slots on the path; the final
// heap_sort (:33-42) runs the same way.  The acceptance test is the reference's strict `d2 < best_dist[0]` in
// candidate-index order, so the sequence of heap operations - and with it the neighbour SET and the ORDER inside
// groups of exactly equal distances, which depend on the whole history of the heap - is the reference's, bit for bit
// (tests: gridded coordinates, where ties are the common case).
// Distances use the reference's expression (dx*dx + dy*dy + dz*dz, left to right, no fma contraction).
// Reference: libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104,
//            grouping/grouping_cuda_kernel.cu:5-25, interpolation/interpolation_cuda_kernel.cu:5-33.
#include "common.h"
#include "hashtable.h"
#include <stdlib.h>
#include "../../include/ptv3_hip.h"

namespace ptv3 {

__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float x, float y, float z) {
  float dx = __fsub_rn(qx, x), dy = __fsub_rn(qy, y), dz = __fsub_rn(qz, z);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- the reference's heap, slot e (wave-uniform index) in lane e & 63 of register e >> 6
template <int KPL>
__device__ __forceinline__ float heap_d(const float (&hd)[KPL], int e) {
  if (KPL == 1 || e < 64) return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hd[0]), e & 63));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hd[KPL - 1]), e & 63));
}
template <int KPL>
__device__ __forceinline__ int heap_i(const int (&hi)[KPL], int e) {
  if (KPL == 1 || e < 64) return __builtin_amdgcn_readlane(hi[0], e & 63);
  return __builtin_amdgcn_readlane(hi[KPL - 1], e & 63);
}
template <int KPL>
__device__ __forceinline__ void heap_set(float (&hd)[KPL], int (&hi)[KPL], int e, float d, int i, int lane) {
  const bool mine = lane == (e & 63);
  if (KPL == 1 || e < 64) { hd[0] = mine ? d : hd[0]; hi[0] = mine ? i : hi[0]; }
  else { hd[KPL - 1] = mine ? d : hd[KPL - 1]; hi[KPL - 1] = mine ? i : hi[KPL - 1]; }
}
// reheap() with (x, xi) placed at the root of a heap of `size` slots (:15-30); everything here is wave-uniform
template <int KPL>
__device__ __forceinline__ void heap_sink_root(float (&hd)[KPL], int (&hi)[KPL], int size, float x, int xi, int lane) {
  int root = 0;
  for (;;) {
    int child = 2 * root + 1;
    if (child >= size) break;
    float dc = heap_d<KPL>(hd, child);
    if (child + 1 < size) {
      const float dr = heap_d<KPL>(hd, child + 1);
      if (dr > dc) { ++child; dc = dr; }
    }
    if (x > dc) break;                                          // `if (dist[root] > dist[child]) return;`
    heap_set<KPL>(hd, hi, root, dc, heap_i<KPL>(hi, child), lane);  // the swap: the child moves up, x goes on
    root = child;
  }
  heap_set<KPL>(hd, hi, root, x, xi, lane);
}

// Scan the candidates [start, end) for NQ queries at once: every 64-candidate chunk is loaded ONCE per wave and
// measured against all NQ queries; each query keeps its own heap across the lanes.
template <int KPL, int NQ>
__device__ __forceinline__ void knn_scan(const int* qs, int nq, int start, int end, int nsample,
                                         const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                         int* __restrict__ idx, float* __restrict__ dist2, int lane) {
  float qx[NQ], qy[NQ], qz[NQ], tau[NQ];
  float bd[NQ][KPL];
  int bi[NQ][KPL];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int qi = q < nq ? qs[q] : qs[0];
    qx[q] = new_xyz[3 * qi]; qy[q] = new_xyz[3 * qi + 1]; qz[q] = new_xyz[3 * qi + 2];
    tau[q] = q < nq ? 1e10f : -1.0f;   // inactive slots never accept a candidate
#pragma unroll
    for (int e = 0; e < KPL; ++e) { bd[q][e] = 1e10f; bi[q][e] = -1; }
  }
  for (int base = start; base < end; base += 64) {
    const int i = base + lane;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    const bool cv = i < end;
    if (cv) { cx = xyz[3 * i]; cy = xyz[3 * i + 1]; cz = xyz[3 * i + 2]; }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float d = 3.0e38f;
      if (cv) d = dist2_ref(qx[q], qy[q], qz[q], cx, cy, cz);
      unsigned long long mask = __ballot(d < tau[q]);
      while (mask) {   // the few candidates of this chunk that may enter, in index order (as the reference's loop)
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
        if (!(dc < tau[q])) continue;   // `if (d2 < best_dist[0])` against the root as it is NOW
        heap_sink_root<KPL>(bd[q], bi[q], nsample, dc, base + src, lane);
        tau[q] = heap_d<KPL>(bd[q], 0);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q >= nq) continue;
    // heap_sort (:33-42): slot i takes the root, the old slot i sinks from the root of the remaining i slots
    for (int i = nsample - 1; i > 0; --i) {
      const float xd = heap_d<KPL>(bd[q], i);
      const int xi = heap_i<KPL>(bi[q], i);
      heap_set<KPL>(bd[q], bi[q], i, heap_d<KPL>(bd[q], 0), heap_i<KPL>(bi[q], 0), lane);
      heap_sink_root<KPL>(bd[q], bi[q], i, xd, xi, lane);
    }
#pragma unroll
    for (int e = 0; e < KPL; ++e) {
      const int p = lane + 64 * e;
      if (p < nsample) {
        idx[(int64_t)qs[q] * nsample + p] = bi[q][e];
        dist2[(int64_t)qs[q] * nsample + p] = bd[q][e];
      }
    }
  }
}

__device__ __forceinline__ int knn_scene_of(int q, const int* __restrict__ new_offset, int b) {
  int bt = 0;
  while (bt < b - 1 && q >= new_offset[bt]) ++bt;
  return bt;
}

// one wave per NQ consecutive queries; a group that straddles a scene boundary is scanned query by query
template <int KPL, int NQ>
__global__ void __launch_bounds__(256)
knn_query_kernel(int m, int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                 const int* __restrict__ offset, const int* __restrict__ new_offset, int b,
                 int* __restrict__ idx, float* __restrict__ dist2) {
  const int lane = threadIdx.x & 63;
  const int q0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NQ;
  if (q0 >= m) return;  // whole wave leaves together
  const int nq = min(NQ, m - q0);
  int qs[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) qs[q] = q0 + (q < nq ? q : 0);
  const int b0 = knn_scene_of(q0, new_offset, b), b1 = knn_scene_of(q0 + nq - 1, new_offset, b);
  if (b0 == b1) {
    knn_scan<KPL, NQ>(qs, nq, b0 == 0 ? 0 : offset[b0 - 1], offset[b0], nsample, xyz, new_xyz, idx, dist2, lane);
  } else {
    for (int q = 0; q < nq; ++q) {
      const int bt = knn_scene_of(q0 + q, new_offset, b);
      const int one[1] = {q0 + q};
      knn_scan<KPL, 1>(one, 1, bt == 0 ? 0 : offset[bt - 1], offset[bt], nsample, xyz, new_xyz, idx, dist2, lane);
    }
  }
}

// ---- cell-grid kNN ----------------------------------------------------------------------------------------------
// The k best of a query by the total order (distance, candidate index), found by walking the occupied cells of a
// uniform grid in Chebyshev shells around the query's cell instead of scanning the whole scene.  A point in a cell at
// Chebyshev distance >= r + 1 lies at least r * cell away from any point of the query's cell, so after shell r the
// search stops once the k-th distance is below (r * cell)^2.  Against the reference's scan the DISTANCES of a row are
// identical and so is every neighbour strictly closer than the k-th distance; among several candidates AT the k-th
// distance the reference keeps whichever its heap did not happen to hold at the root when a closer candidate evicted
// it, this search keeps the lowest indices, and rows ascend in (distance, index).  nsample = 1 leaves no such
// freedom (strict `<` in index order keeps the first of equals in both): bit-identical.
__device__ __forceinline__ bool pair_gt(float d1, int i1, float d2, int i2) { return d1 > d2 || (d1 == d2 && i1 > i2); }

template <int KPL>
__device__ __forceinline__ void pheap_sink_root(float (&hd)[KPL], int (&hi)[KPL], int size, float x, int xi, int lane) {
  int root = 0;
  for (;;) {
    int child = 2 * root + 1;
    if (child >= size) break;
    float dc = heap_d<KPL>(hd, child);
    int ic = heap_i<KPL>(hi, child);
    if (child + 1 < size) {
      const float dr = heap_d<KPL>(hd, child + 1);
      const int ir = heap_i<KPL>(hi, child + 1);
      if (pair_gt(dr, ir, dc, ic)) { ++child; dc = dr; ic = ir; }
    }
    if (pair_gt(x, xi, dc, ic)) break;
    heap_set<KPL>(hd, hi, root, dc, ic, lane);
    root = child;
  }
  heap_set<KPL>(hd, hi, root, x, xi, lane);
}

// offer candidate (d, i) of every lane with `has` to the wave's heap, in lane order
template <int KPL>
__device__ __forceinline__ void pheap_offer(float (&bd)[KPL], int (&bi)[KPL], int nsample, bool has, float d, int i,
                                            float& tau_d, int& tau_i, int lane) {
  unsigned long long hit = __ballot(has && pair_gt(tau_d, tau_i, d, i));
  while (hit) {
    const int src = __ffsll((long long)hit) - 1;
    hit &= hit - 1;
    const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), src));
    const int ic = __builtin_amdgcn_readlane(i, src);
    if (!pair_gt(tau_d, tau_i, dc, ic)) continue;   // against the root as it is NOW
    pheap_sink_root<KPL>(bd, bi, nsample, dc, ic, lane);
    tau_d = heap_d<KPL>(bd, 0);
    tau_i = heap_i<KPL>(bi, 0);
  }
}

constexpr int KNN_CELL_SHELLS = 8;   // shells walked before a query falls back to scanning its scene

template <int KPL>
__global__ void __launch_bounds__(256)
knn_cells_kernel(int m, int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                 const int* __restrict__ qcell, const unsigned long long* __restrict__ keys,
                 const int* __restrict__ vals, unsigned long long mask, const long long* __restrict__ order,
                 const int* __restrict__ seg_start, const int* __restrict__ offset, float cell,
                 int* __restrict__ idx, float* __restrict__ dist2) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= m) return;  // whole wave leaves together
  const float qx = new_xyz[3 * q], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];
  const int cb = qcell[4 * q], cx = qcell[4 * q + 1], cy = qcell[4 * q + 2], cz = qcell[4 * q + 3];
  float bd[KPL];
  int bi[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) { bd[e] = 1e10f; bi[e] = 0x7fffffff; }   // sentinels lose every comparison
  float tau_d = 1e10f;
  int tau_i = 0x7fffffff;
  bool done = false;
  for (int r = 0; r <= KNN_CELL_SHELLS && !done; ++r) {
    const int side = 2 * r + 1, ncube = side * side * side;
    for (int base = 0; base < ncube; base += 64) {
      const int t = base + lane;
      int ci = -1;
      if (t < ncube) {
        const int dx = t / (side * side) - r, dy = (t / side) % side - r, dz = t % side - r;
        const int ch = max(abs(dx), max(abs(dy), abs(dz)));
        if (ch == r) ci = ht_find(keys, vals, mask, cb, cx + dx, cy + dy, cz + dz);
      }
      const int s0 = ci >= 0 ? seg_start[ci] : 0;
      const int cnt = ci >= 0 ? seg_start[ci + 1] - s0 : 0;
      for (int u = 0; __ballot(u < cnt); ++u) {
        const bool has = u < cnt;
        const int i = has ? (int)order[s0 + u] : 0x7fffffff;
        float d = 3.0e38f;
        if (has) d = dist2_ref(qx, qy, qz, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        pheap_offer<KPL>(bd, bi, nsample, has, d, i, tau_d, tau_i, lane);
      }
    }
    // everything outside shell r is farther than r cells - less the rounding of the cell assignment: a point's cell
    // is floor((x - lo) / cell) in fp32 (two roundings, each 2^-24 relative to a quotient below the 65536-cell grid
    // extent), so a point can sit up to 2 * 65536 * 2^-23 = 0.016 cells across a boundary, on either end of the pair
    const float reach = ((float)r - 0.04f) * cell;
    done = tau_d < 1e10f && tau_d < reach * reach;
  }
  if (!done) {
    // sparse neighbourhood (or a scene with fewer than nsample candidates): start over with a scan of the scene
#pragma unroll
    for (int e = 0; e < KPL; ++e) { bd[e] = 1e10f; bi[e] = 0x7fffffff; }
    tau_d = 1e10f;
    tau_i = 0x7fffffff;
    const int start = cb == 0 ? 0 : offset[cb - 1], end = offset[cb];
    for (int base = start; base < end; base += 64) {
      const int i = base + lane;
      const bool has = i < end;
      float d = 3.0e38f;
      if (has) d = dist2_ref(qx, qy, qz, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      pheap_offer<KPL>(bd, bi, nsample, has, d, has ? i : 0x7fffffff, tau_d, tau_i, lane);
    }
  }
  // ascending (distance, index): repeatedly move the root behind the shrinking heap
  for (int i = nsample - 1; i > 0; --i) {
    const float xd = heap_d<KPL>(bd, i);
    const int xi = heap_i<KPL>(bi, i);
    heap_set<KPL>(bd, bi, i, heap_d<KPL>(bd, 0), heap_i<KPL>(bi, 0), lane);
    pheap_sink_root<KPL>(bd, bi, i, xd, xi, lane);
  }
#pragma unroll
  for (int e = 0; e < KPL; ++e) {
    const int p = lane + 64 * e;
    if (p < nsample) {
      idx[(int64_t)q * nsample + p] = bi[e] == 0x7fffffff ? -1 : bi[e];
      dist2[(int64_t)q * nsample + p] = bd[e];
    }
  }
}

__global__ void grouping_forward_kernel(int64_t total, int nsample, int c, const float* __restrict__ input,
                                        const int* __restrict__ idx, float* __restrict__ output) {
  int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= total) return;
  const int c_idx = (int)(index % c);
  const int64_t pair = index / c;  // m_idx * nsample + nsample_idx
  const int src = idx[pair];
  output[index] = src >= 0 ? input[(int64_t)src * c + c_idx] : 0.f;
}

__global__ void grouping_backward_kernel(int64_t total, int nsample, int c, const float* __restrict__ grad_output,
                                         const int* __restrict__ idx, float* __restrict__ grad_input) {
  int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= total) return;
  const int c_idx = (int)(index % c);
  const int src = idx[index / c];
  if (src >= 0) atomicAdd(grad_input + (int64_t)src * c + c_idx, grad_output[index]);
}

__global__ void interpolation_forward_kernel(int64_t total, int c, int k, const float* __restrict__ input,
                                             const int* __restrict__ idx, const float* __restrict__ weight,
                                             float* __restrict__ output) {
  int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= total) return;
  const int c_idx = (int)(index % c);
  const int64_t n_idx = index / c;
  float acc = output[index];
  for (int i = 0; i < k; ++i) {
    const int src = idx[n_idx * k + i];
    if (src >= 0) acc += input[(int64_t)src * c + c_idx] * weight[n_idx * k + i];
  }
  output[index] = acc;
}

__global__ void interpolation_backward_kernel(int64_t total, int c, int k, const float* __restrict__ grad_output,
                                              const int* __restrict__ idx, const float* __restrict__ weight,
                                              float* __restrict__ grad_input) {
  int64_t index = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= total) return;
  const int c_idx = (int)(index % c);
  const int64_t n_idx = index / c;
  const float go = grad_output[index];
  for (int i = 0; i < k; ++i) {
    const int src = idx[n_idx * k + i];
    if (src >= 0) atomicAdd(grad_input + (int64_t)src * c + c_idx, go * weight[n_idx * k + i]);
  }
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_knn_query(int m, int nsample, const float* xyz, const float* new_xyz, const int* offset,
                              const int* new_offset, int b, int* idx, float* dist2, void* stream) {
  PTV3_REQUIRE(nsample >= 1 && nsample <= 128, "knn_query: nsample=%d outside [1,128] (reference limit)", nsample);
  PTV3_REQUIRE(b >= 1, "knn_query: empty offset");
  if (m == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("PTV3_KNN_QUERIES_PER_WAVE"); forced = e ? atoi(e) : 0; }
  // 4 queries per wave once there are enough waves to fill the chip (256 CUs x 4 SIMDs); measured: 25k queries 1.04 ms at 4 per wave, 1.14 at 2
  int nq = forced == 1 || forced == 2 || forced == 4 ? forced : (m >= 16384 ? 4 : m >= 4096 ? 2 : 1);
  if (nsample > 64 && nq > 2) nq = 2;
  dim3 grid((unsigned)cdiv(m, 4 * nq)), block(256);
#define KNN_LAUNCH(KPL, NQ) \
  hipLaunchKernelGGL((knn_query_kernel<KPL, NQ>), grid, block, 0, s, m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2)
  if (nsample <= 64) {
    if (nq == 4) KNN_LAUNCH(1, 4); else if (nq == 2) KNN_LAUNCH(1, 2); else KNN_LAUNCH(1, 1);
  } else {
    if (nq == 2) KNN_LAUNCH(2, 2); else KNN_LAUNCH(2, 1);
  }
#undef KNN_LAUNCH
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_knn_query_cells(int m, int nsample, const float* xyz, const float* new_xyz, const int32_t* qcell,
                                    const void* table, int64_t slots, const int64_t* order, const int32_t* seg_start,
                                    const int* offset, float cell, int* idx, float* dist2, void* stream) {
  PTV3_REQUIRE(nsample >= 1 && nsample <= 128, "knn_query_cells: nsample=%d outside [1,128]", nsample);
  PTV3_REQUIRE(slots > 0 && (slots & (slots - 1)) == 0, "knn_query_cells: slots must be a power of two");
  PTV3_REQUIRE(cell > 0.f, "knn_query_cells: cell %g", (double)cell);
  if (m == 0) return PTV3_OK;
  const unsigned long long* keys = (const unsigned long long*)table;
  const int32_t* vals = (const int32_t*)((const char*)table + slots * 8);
  dim3 grid((unsigned)cdiv(m, 4)), block(256);
  if (nsample <= 64)
    hipLaunchKernelGGL((knn_cells_kernel<1>), grid, block, 0, (hipStream_t)stream, m, nsample, xyz, new_xyz, qcell, keys,
                       vals, (unsigned long long)(slots - 1), (const long long*)order, seg_start, offset, cell, idx, dist2);
  else
    hipLaunchKernelGGL((knn_cells_kernel<2>), grid, block, 0, (hipStream_t)stream, m, nsample, xyz, new_xyz, qcell, keys,
                       vals, (unsigned long long)(slots - 1), (const long long*)order, seg_start, offset, cell, idx, dist2);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_grouping_forward(int m, int nsample, int c, const float* input, const int* idx, float* output,
                                     void* stream) {
  int64_t total = (int64_t)m * nsample * c;
  if (total == 0) return PTV3_OK;
  hipLaunchKernelGGL(grouping_forward_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     total, nsample, c, input, idx, output);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_grouping_backward(int m, int nsample, int c, const float* grad_output, const int* idx,
                                      float* grad_input, void* stream) {
  int64_t total = (int64_t)m * nsample * c;
  if (total == 0) return PTV3_OK;
  hipLaunchKernelGGL(grouping_backward_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     total, nsample, c, grad_output, idx, grad_input);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_interpolation_forward(int n, int c, int k, const float* input, const int* idx,
                                          const float* weight, float* output, void* stream) {
  int64_t total = (int64_t)n * c;
  if (total == 0) return PTV3_OK;
  hipLaunchKernelGGL(interpolation_forward_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, total, c, k, input, idx, weight, output);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_interpolation_backward(int n, int c, int k, const float* grad_output, const int* idx,
                                           const float* weight, float* grad_input, void* stream) {
  int64_t total = (int64_t)n * c;
  if (total == 0) return PTV3_OK;
  hipLaunchKernelGGL(interpolation_backward_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, total, c, k, grad_output, idx, weight, grad_input);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
