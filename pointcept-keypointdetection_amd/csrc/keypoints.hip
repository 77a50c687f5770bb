// Keypoint aggregation: the per-sample / per-keypoint python loops either side of the model
// (engines/hooks/offset_keypoint_evaluator.py:46-84 and tools/infer_offset.py:555-597) as one launch:
// workgroup (k, b) reduces the points of scene b for keypoint k.  Reductions use a fixed tree (deterministic);
// ties of the argmax go to the lowest point index like torch.argmax.
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

struct KpPartial {
  float best; int best_idx;       // arg max of the score
  float w, x, y, z;               // weighted sums (or plain sums for the GT mean)
  int count; int first;           // valid points, lowest valid index
};

__device__ __forceinline__ void kp_merge(KpPartial& a, const KpPartial& b) {
  if (b.best > a.best || (b.best == a.best && b.best_idx < a.best_idx)) { a.best = b.best; a.best_idx = b.best_idx; }
  a.w += b.w; a.x += b.x; a.y += b.y; a.z += b.z;
  a.count += b.count;
  a.first = min(a.first, b.first);
}

// position of point i in the output frame: coord * s + centroid, each step rounded like the torch statements
__device__ __forceinline__ void kp_point(const float* coord, const float* v4, int64_t i, float s, const float* cen,
                                         float* out) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float pos = __fadd_rn(__fmul_rn(coord[3 * i + d], s), cen[d]);
    out[d] = __fadd_rn(pos, __fmul_rn(v4[d], s));
  }
}

__global__ void __launch_bounds__(256) keypoint_aggregate_kernel(const float* __restrict__ coord,
                                                                  const float* __restrict__ pred,
                                                                  const int64_t* __restrict__ offset, int nkp,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ centroid, int mode,
                                                                  float thresh, float* __restrict__ kp_out,
                                                                  int32_t* __restrict__ aux_out) {
  __shared__ KpPartial red[256];
  const int k = blockIdx.x, b = blockIdx.y;
  const int64_t start = b == 0 ? 0 : offset[b - 1], end = offset[b];
  const float s = scale ? scale[b] : 1.0f;
  float cen[3] = {0.f, 0.f, 0.f};
  if (centroid) { cen[0] = centroid[3 * b]; cen[1] = centroid[3 * b + 1]; cen[2] = centroid[3 * b + 2]; }
  KpPartial p;
  p.best = -INFINITY; p.best_idx = 0x7fffffff; p.w = p.x = p.y = p.z = 0.f; p.count = 0; p.first = 0x7fffffff;
  for (int64_t i = start + threadIdx.x; i < end; i += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(pred + (i * nkp + k) * 4);
    const int li = (int)(i - start);
    if (mode == PTV3_KP_ARGMAX || mode == PTV3_KP_WEIGHTED) {
      if (v[3] > p.best) { p.best = v[3]; p.best_idx = li; }   // ascending i per thread: first max kept
    }
    const bool valid = mode == PTV3_KP_WEIGHTED ? v[3] > thresh
                       : mode == PTV3_KP_GT_MEAN ? v[3] > 0.f
                       : mode == PTV3_KP_GT_FIRST ? v[3] > 0.5f : false;
    if (valid) {
      p.count += 1;
      p.first = min(p.first, li);
      if (mode != PTV3_KP_GT_FIRST) {
        float q[3];
        const float vv[3] = {v[0], v[1], v[2]};
        kp_point(coord, vv, i, s, cen, q);
        const float wgt = mode == PTV3_KP_WEIGHTED ? v[3] : 1.0f;
        p.w += wgt; p.x += wgt * q[0]; p.y += wgt * q[1]; p.z += wgt * q[2];
      }
    }
  }
  red[threadIdx.x] = p;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) kp_merge(red[threadIdx.x], red[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  p = red[0];
  float out[3] = {NAN, NAN, NAN};
  int aux = 0;
  const int n = (int)(end - start);
  if (mode == PTV3_KP_ARGMAX || (mode == PTV3_KP_WEIGHTED && p.count == 0)) {
    aux = mode == PTV3_KP_ARGMAX ? p.best_idx : 0;
    if (n > 0) {
      const int64_t i = start + p.best_idx;
      const float vv[3] = {pred[(i * nkp + k) * 4], pred[(i * nkp + k) * 4 + 1], pred[(i * nkp + k) * 4 + 2]};
      kp_point(coord, vv, i, s, cen, out);
    } else {
      aux = -1;
      if (mode == PTV3_KP_WEIGHTED) out[0] = out[1] = out[2] = 0.f;  // infer_offset leaves the zero initialiser
    }
    if (mode == PTV3_KP_ARGMAX && n == 0) out[0] = out[1] = out[2] = 0.f;
  } else if (mode == PTV3_KP_WEIGHTED || mode == PTV3_KP_GT_MEAN) {
    aux = p.count;
    if (p.count > 0) { out[0] = p.x / p.w; out[1] = p.y / p.w; out[2] = p.z / p.w; }
  } else {  // GT_FIRST
    aux = p.count > 0 ? p.first : -1;
    if (p.count > 0) {
      const int64_t i = start + p.first;
      const float vv[3] = {pred[(i * nkp + k) * 4], pred[(i * nkp + k) * 4 + 1], pred[(i * nkp + k) * 4 + 2]};
      kp_point(coord, vv, i, s, cen, out);
    }
  }
  float* o = kp_out + ((int64_t)b * nkp + k) * 3;
  o[0] = out[0]; o[1] = out[1]; o[2] = out[2];
  aux_out[(int64_t)b * nkp + k] = aux;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_keypoint_aggregate(const float* coord, const float* pred, const int64_t* offset, int nscenes,
                                       int nkp, const float* scale, const float* centroid, int mode, float thresh,
                                       float* kp_out, int32_t* aux_out, void* stream) {
  PTV3_REQUIRE(mode >= PTV3_KP_ARGMAX && mode <= PTV3_KP_GT_FIRST, "keypoint_aggregate: mode %d outside [0,3]", mode);
  PTV3_REQUIRE(nscenes >= 0 && nkp >= 1 && nkp <= 65535, "keypoint_aggregate: bad nscenes=%d / nkp=%d", nscenes, nkp);
  if (nscenes == 0) return PTV3_OK;
  hipLaunchKernelGGL(keypoint_aggregate_kernel, dim3((unsigned)nkp, (unsigned)nscenes), dim3(256), 0,
                     (hipStream_t)stream, coord, pred, offset, nkp, scale, centroid, mode, thresh, kp_out, aux_out);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
