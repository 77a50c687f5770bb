// libs/pointops knn_query / grouping / interpolation as wavefront-primitive HIP.
// knn: one 64-lane wave per query; the running top-k lives sorted across the lanes of the wave
// (lane l holds ranks l and l+64), candidates are scanned 64 at a time, a ballot finds the few that
// beat the current k-th distance and each is inserted with one ballot + one lane shift.
// Distances use the reference's expression (dx*dx + dy*dy + dz*dz, left to right, no fma contraction)
// so distances and indices match a plain-C restatement of the reference bit for bit - except for the ORDER inside a
// group of exactly equal fp32 distances, which the reference's heap sort leaves implementation-defined (here:
// ascending index).  The neighbour set is identical (strict `<` at the k-th distance in both).
// Reference: libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104,
//            grouping/grouping_cuda_kernel.cu:5-25, interpolation/interpolation_cuda_kernel.cu:5-33.
#include "common.h"
#include <stdlib.h>
#include "../../include/ptv3_hip.h"

namespace ptv3 {

__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float x, float y, float z) {
  float dx = __fsub_rn(qx, x), dy = __fsub_rn(qy, y), dz = __fsub_rn(qz, z);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// Scan the candidates [start, end) for NQ queries at once: every 64-candidate chunk is loaded ONCE per wave and
// measured against all NQ queries (the loads and the loop overhead were ~half of the single-query kernel's
// instruction stream); each query keeps its own sorted top-k across the lanes.
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
  const int tpos = nsample - 1;
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
      while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const float dc = __shfl(d, src, 64);
        if (!(dc < tau[q])) continue;
        const int ci = base + src;
        // rank of the newcomer: behind every entry with distance <= dc (earlier index wins ties)
        int pos = 0;
#pragma unroll
        for (int e = 0; e < KPL; ++e) pos += __popcll(__ballot(bd[q][e] <= dc));
#pragma unroll
        for (int e = KPL - 1; e >= 0; --e) {
          const int p = lane + 64 * e;
          float ud = __shfl_up(bd[q][e], 1, 64);
          int ui = __shfl_up(bi[q][e], 1, 64);
          if (e > 0) {
            float wd = __shfl(bd[q][e - 1], 63, 64);
            int wi = __shfl(bi[q][e - 1], 63, 64);
            if (lane == 0) { ud = wd; ui = wi; }
          }
          if (p > pos) { bd[q][e] = ud; bi[q][e] = ui; }
          else if (p == pos) { bd[q][e] = dc; bi[q][e] = ci; }
        }
        const float t0 = __shfl(bd[q][0], tpos & 63, 64);
        if (KPL == 2) {
          const float t1 = __shfl(bd[q][KPL - 1], tpos & 63, 64);
          tau[q] = tpos >= 64 ? t1 : t0;
        } else {
          tau[q] = t0;
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q >= nq) continue;
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
