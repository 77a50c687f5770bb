// libs/pointops knn_query / grouping / interpolation as wavefront-primitive HIP.
// knn: one 64-lane wave per query; the running top-k lives sorted across the lanes of the wave
// (lane l holds ranks l and l+64), candidates are scanned 64 at a time, a ballot finds the few that
// beat the current k-th distance and each is inserted with one ballot + one lane shift.
// Distances use the reference's expression (dx*dx + dy*dy + dz*dz, left to right, no fma contraction)
// so indices match a plain-C restatement bit for bit.
// Reference: libs/pointops/src/knn_query/knn_query_cuda_kernel.cu:60-104,
//            grouping/grouping_cuda_kernel.cu:5-25, interpolation/interpolation_cuda_kernel.cu:5-33.
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

__device__ __forceinline__ float dist2_ref(float qx, float qy, float qz, float x, float y, float z) {
  float dx = __fsub_rn(qx, x), dy = __fsub_rn(qy, y), dz = __fsub_rn(qz, z);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int KPL>
__global__ void __launch_bounds__(256)
knn_query_kernel(int m, int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                 const int* __restrict__ offset, const int* __restrict__ new_offset, int b,
                 int* __restrict__ idx, float* __restrict__ dist2) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= m) return;  // whole wave leaves together
  int bt = 0;
  while (bt < b - 1 && q >= new_offset[bt]) ++bt;
  const int start = bt == 0 ? 0 : offset[bt - 1];
  const int end = offset[bt];
  const float qx = new_xyz[3 * q], qy = new_xyz[3 * q + 1], qz = new_xyz[3 * q + 2];

  float bd[KPL];
  int bi[KPL];
#pragma unroll
  for (int e = 0; e < KPL; ++e) { bd[e] = 1e10f; bi[e] = -1; }
  float tau = 1e10f;
  const int tpos = nsample - 1;

  for (int base = start; base < end; base += 64) {
    const int i = base + lane;
    float d = 3.0e38f;
    if (i < end) d = dist2_ref(qx, qy, qz, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    unsigned long long mask = __ballot(d < tau);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const float dc = __shfl(d, src, 64);
      if (!(dc < tau)) continue;
      const int ci = base + src;
      // rank of the newcomer: behind every entry with distance <= dc (earlier index wins ties)
      int pos = 0;
#pragma unroll
      for (int e = 0; e < KPL; ++e) pos += __popcll(__ballot(bd[e] <= dc));
#pragma unroll
      for (int e = KPL - 1; e >= 0; --e) {
        const int p = lane + 64 * e;
        float ud = __shfl_up(bd[e], 1, 64);
        int ui = __shfl_up(bi[e], 1, 64);
        if (e > 0) {
          float wd = __shfl(bd[e - 1], 63, 64);
          int wi = __shfl(bi[e - 1], 63, 64);
          if (lane == 0) { ud = wd; ui = wi; }
        }
        if (p > pos) { bd[e] = ud; bi[e] = ui; }
        else if (p == pos) { bd[e] = dc; bi[e] = ci; }
      }
      const float t0 = __shfl(bd[0], tpos & 63, 64);
      if (KPL == 2) {
        const float t1 = __shfl(bd[KPL - 1], tpos & 63, 64);
        tau = tpos >= 64 ? t1 : t0;
      } else {
        tau = t0;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < KPL; ++e) {
    const int p = lane + 64 * e;
    if (p < nsample) {
      idx[(int64_t)q * nsample + p] = bi[e];
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
  dim3 grid((unsigned)cdiv(m, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (nsample <= 64)
    hipLaunchKernelGGL(knn_query_kernel<1>, grid, block, 0, s, m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2);
  else
    hipLaunchKernelGGL(knn_query_kernel<2>, grid, block, 0, s, m, nsample, xyz, new_xyz, offset, new_offset, b, idx, dist2);
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
