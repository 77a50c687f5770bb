// LayerNorm (+ residual, + chained second LayerNorm), folded eval-BatchNorm + activation, dtype casts.
// HBM-bound: one read and one write of the (m, c) matrix; statistics in fp32.
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int LN_MAXCH = 8;  // 4-element chunks held per lane -> c <= 64 lanes * 8 * 4 = 2048

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// LPR lanes cooperate on one row (64/LPR rows per wave), 4 waves per block
template <typename T, int LPR>
__global__ void __launch_bounds__(256)
layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                 const T* __restrict__ res, T* __restrict__ y, const float* __restrict__ gamma2,
                 const float* __restrict__ beta2, T* __restrict__ y2, int64_t m, int c, float eps,
                 const float* __restrict__ slab, int splits, const float* __restrict__ slab_bias) {
  typedef typename Vec4<T>::type V4;
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * RPW + lane / LPR;
  const bool rv = row < m;
  const int nch = c / 4;  // chunks per row; lane takes chunks sub, sub+LPR, ...
  float v[LN_MAXCH][4];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    int ch = sub + k * LPR;
    v[k][0] = v[k][1] = v[k][2] = v[k][3] = 0.f;
    if (rv && ch < nch) {
      if (slab) {
        // the producing GEMM left split-K fp32 slabs: sum them in slab order, add its bias, round to T as the
        // separate reduce kernel would have stored it
        f32x4 acc = *reinterpret_cast<const f32x4*>(slab_bias + 4 * ch);
        for (int z = 0; z < splits; ++z)
          acc += *reinterpret_cast<const f32x4*>(slab + ((int64_t)z * m + row) * c + 4 * ch);
        unpack4<T>(pack4<T>(acc[0], acc[1], acc[2], acc[3]), v[k]);
      } else {
        unpack4<T>(*reinterpret_cast<const V4*>(x + row * c + 4 * ch), v[k]);
      }
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
    }
  }
  const float inv_c = 1.0f / (float)c;
  float mean = row_sum<LPR>(s) * inv_c;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    int ch = sub + k * LPR;
    if (ch < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[k][e] - mean; q += d * d; }
    }
  }
  float rstd = rsqrtf(row_sum<LPR>(q) * inv_c + eps);
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    int ch = sub + k * LPR;
    if (rv && ch < nch) {
      f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * ch);
      f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 4 * ch);
      float r[4] = {0.f, 0.f, 0.f, 0.f};
      if (res) unpack4<T>(*reinterpret_cast<const V4*>(res + row * c + 4 * ch), r);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[k][e] = (v[k][e] - mean) * rstd * gm[e] + bt[e] + r[e];
      *reinterpret_cast<V4*>(y + row * c + 4 * ch) = pack4<T>(v[k][0], v[k][1], v[k][2], v[k][3]);
      if (y2) {
        // the chained norm sees what was stored (rounded to T), as a separate kernel would
        unpack4<T>(pack4<T>(v[k][0], v[k][1], v[k][2], v[k][3]), v[k]);
        s2 += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
      }
    }
  }
  if (y2 == nullptr) return;
  mean = row_sum<LPR>(s2) * inv_c;
  q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    int ch = sub + k * LPR;
    if (rv && ch < nch) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float d = v[k][e] - mean; q += d * d; }
    }
  }
  rstd = rsqrtf(row_sum<LPR>(q) * inv_c + eps);
#pragma unroll
  for (int k = 0; k < LN_MAXCH; ++k) {
    int ch = sub + k * LPR;
    if (rv && ch < nch) {
      f32x4 gm = *reinterpret_cast<const f32x4*>(gamma2 + 4 * ch);
      f32x4 bt = *reinterpret_cast<const f32x4*>(beta2 + 4 * ch);
      *reinterpret_cast<V4*>(y2 + row * c + 4 * ch) =
          pack4<T>((v[k][0] - mean) * rstd * gm[0] + bt[0], (v[k][1] - mean) * rstd * gm[1] + bt[1],
                   (v[k][2] - mean) * rstd * gm[2] + bt[2], (v[k][3] - mean) * rstd * gm[3] + bt[3]);
    }
  }
}

template <typename T>
__global__ void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, int act, T* __restrict__ y, int64_t total4,
                                  int c) {
  typedef typename Vec4<T>::type V4;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  int ch = (int)((i * 4) % c);
  float v[4];
  unpack4<T>(*reinterpret_cast<const V4*>(x + 4 * i), v);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float t = v[e];
    if (scale) t = t * scale[ch + e] + shift[ch + e];
    if (act == PTV3_ACT_GELU) t = gelu_erf(t);
    else if (act == PTV3_ACT_RELU) t = fmaxf(t, 0.f);
    v[e] = t;
  }
  *reinterpret_cast<V4*>(y + 4 * i) = pack4<T>(v[0], v[1], v[2], v[3]);
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ x, D* __restrict__ y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = from_f32<D>(to_f32<S>(x[i]));
}

template <typename T>
static int launch_ln(const void* x, const float* gamma, const float* beta, const void* res, void* y,
                     const float* gamma2, const float* beta2, void* y2, int64_t m, int c, float eps,
                     hipStream_t s, const float* slab = nullptr, int splits = 0, const float* slab_bias = nullptr) {
  const int nch = c / 4;
  int lpr = 1;
  while (lpr < 64 && lpr < nch) lpr <<= 1;
#define LN_LAUNCH(L)                                                                                     \
  {                                                                                                      \
    int64_t rows_per_block = 4 * (64 / L);                                                               \
    hipLaunchKernelGGL((layernorm_kernel<T, L>), dim3((unsigned)cdiv(m, rows_per_block)), dim3(256), 0, s, \
                       (const T*)x, gamma, beta, (const T*)res, (T*)y, gamma2, beta2, (T*)y2, m, c, eps, slab,   \
                       splits, slab_bias);                                                                \
  }
  switch (lpr) {
    case 1: LN_LAUNCH(1) break;
    case 2: LN_LAUNCH(2) break;
    case 4: LN_LAUNCH(4) break;
    case 8: LN_LAUNCH(8) break;
    case 16: LN_LAUNCH(16) break;
    case 32: LN_LAUNCH(32) break;
    default: LN_LAUNCH(64) break;
  }
#undef LN_LAUNCH
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_layernorm(const void* x, const float* gamma, const float* beta, const void* res, void* y,
                              const float* gamma2, const float* beta2, void* y2, int64_t m, int c, float eps,
                              int dtype, void* stream) {
  PTV3_REQUIRE(c > 0 && c % 4 == 0 && c <= 64 * LN_MAXCH * 4, "layernorm: c=%d must be a multiple of 4, <= 2048", c);
  PTV3_REQUIRE((y2 == nullptr) || (gamma2 && beta2), "layernorm: y2 needs gamma2/beta2");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "layernorm: bad dtype");
  if (m == 0) return PTV3_OK;
  if (dtype == PTV3_F32) return launch_ln<float>(x, gamma, beta, res, y, gamma2, beta2, y2, m, c, eps, (hipStream_t)stream);
  return launch_ln<__bf16>(x, gamma, beta, res, y, gamma2, beta2, y2, m, c, eps, (hipStream_t)stream);
}

extern "C" int ptv3_layernorm_slabs(const float* slab, int splits, const float* slab_bias, const float* gamma,
                                    const float* beta, const void* res, void* y, const float* gamma2,
                                    const float* beta2, void* y2, int64_t m, int c, float eps, int dtype,
                                    void* stream) {
  PTV3_REQUIRE(c > 0 && c % 4 == 0 && c <= 64 * LN_MAXCH * 4, "layernorm_slabs: c=%d must be a multiple of 4, <= 2048", c);
  PTV3_REQUIRE(slab && slab_bias && splits >= 1, "layernorm_slabs: slab, slab_bias and splits >= 1 are required");
  PTV3_REQUIRE((y2 == nullptr) || (gamma2 && beta2), "layernorm_slabs: y2 needs gamma2/beta2");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "layernorm_slabs: bad dtype");
  if (m == 0) return PTV3_OK;
  if (dtype == PTV3_F32)
    return launch_ln<float>(nullptr, gamma, beta, res, y, gamma2, beta2, y2, m, c, eps, (hipStream_t)stream, slab, splits,
                            slab_bias);
  return launch_ln<__bf16>(nullptr, gamma, beta, res, y, gamma2, beta2, y2, m, c, eps, (hipStream_t)stream, slab, splits,
                           slab_bias);
}

extern "C" int ptv3_affine_act(const void* x, const float* scale, const float* shift, int act, void* y,
                               int64_t m, int c, int dtype, void* stream) {
  PTV3_REQUIRE(c > 0 && c % 4 == 0, "affine_act: c=%d must be a multiple of 4", c);
  PTV3_REQUIRE((scale == nullptr) == (shift == nullptr), "affine_act: scale/shift must come together");
  if (m == 0) return PTV3_OK;
  int64_t total4 = m * c / 4;
  dim3 grid((unsigned)cdiv(total4, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == PTV3_F32)
    hipLaunchKernelGGL(affine_act_kernel<float>, grid, block, 0, s, (const float*)x, scale, shift, act, (float*)y, total4, c);
  else
    hipLaunchKernelGGL(affine_act_kernel<__bf16>, grid, block, 0, s, (const __bf16*)x, scale, shift, act, (__bf16*)y, total4, c);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_cast(const void* x, int src_dtype, void* y, int dst_dtype, int64_t count, void* stream) {
  if (count == 0) return PTV3_OK;
  dim3 grid((unsigned)cdiv(count, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (src_dtype == PTV3_F32 && dst_dtype == PTV3_BF16)
    hipLaunchKernelGGL((cast_kernel<float, __bf16>), grid, block, 0, s, (const float*)x, (__bf16*)y, count);
  else if (src_dtype == PTV3_BF16 && dst_dtype == PTV3_F32)
    hipLaunchKernelGGL((cast_kernel<__bf16, float>), grid, block, 0, s, (const __bf16*)x, (float*)y, count);
  else {
    set_error("cast: unsupported dtype pair %d -> %d", src_dtype, dst_dtype);
    return PTV3_ERR_UNSUPPORTED;
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
