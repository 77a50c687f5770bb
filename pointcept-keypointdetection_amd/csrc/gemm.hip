// Implicit-GEMM kernel shared by torch.nn.Linear and the submanifold sparse convolution:
//   out[i][o] = epilogue( sum_{d<kvol} sum_{c<cin} w[o][d][c] * x[nbr[i][d]][c] )
// (kvol = 1 and nbr = NULL gives the dense linear).  Rows of x are gathered straight into the
// LDS A-tile, weights stream through the LDS B-tile, both K-contiguous, so every matrix-core
// fragment is one 4-element LDS read.  Product is computed transposed (W_tile * X_tile^T) so each
// lane ends up with 4 consecutive output channels of ONE point: vector epilogue + 16-B stores.
// Reference semantics: include/ptv3_hip.h (ptv3_gemm).
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

constexpr int GM_THREADS = 256;
constexpr int GM_BM = 64;   // points per workgroup (4 waves x 16)
constexpr int GM_BK = 32;   // K elements per LDS stage
constexpr int GM_LS = GM_BK + 4;  // LDS row stride (elements)

struct GemmArgs {
  const void* x; const void* w; void* out; void* out2; const void* res;
  const int32_t* nbr; const int32_t* row_order; const int32_t* res_index;
  const float* bias; const float* bn_scale; const float* bn_shift;
  int64_t m; int cin; int cout; int kvol; int act;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == PTV3_ACT_GELU) return gelu_erf(v);
  if (act == PTV3_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

template <typename T, int BN>
__global__ void __launch_bounds__(GM_THREADS) gemm_kernel(GemmArgs a) {
  typedef typename Vec4<T>::type V4;
  constexpr int NT = BN / 16;                                  // 16-channel tiles per wave
  constexpr int A_LOADS = (GM_BM * (GM_BK / 4)) / GM_THREADS;  // 2
  constexpr int B_LOADS = (BN * (GM_BK / 4)) / GM_THREADS;     // BN/32
  __shared__ __attribute__((aligned(16))) T sA[GM_BM * GM_LS];
  __shared__ __attribute__((aligned(16))) T sB[BN * GM_LS];

  const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GM_BM;
  const int n0 = blockIdx.y * BN;
  const int ktot = a.kvol * a.cin;
  const int nsteps = (ktot + GM_BK - 1) / GM_BK;

  // A staging: chunk e -> (row e/8, 4-element chunk e%8); both chunks of a thread sit in one row
  int64_t arow[A_LOADS];
  int a_r[A_LOADS], a_ch[A_LOADS];
#pragma unroll
  for (int u = 0; u < A_LOADS; ++u) {
    int e = tid * A_LOADS + u;
    a_r[u] = e / (GM_BK / 4);
    a_ch[u] = e % (GM_BK / 4);
    int64_t r = row0 + a_r[u];
    if (r < a.m) arow[u] = a.row_order ? (int64_t)a.row_order[r] : r; else arow[u] = -1;
  }
  int b_r[B_LOADS], b_ch[B_LOADS];
#pragma unroll
  for (int u = 0; u < B_LOADS; ++u) {
    int e = tid * B_LOADS + u;
    b_r[u] = e / (GM_BK / 4);
    b_ch[u] = e % (GM_BK / 4);
  }

  V4 ra[A_LOADS], rb[B_LOADS];
  auto issue = [&](int step) {
    const int k0 = step * GM_BK;
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) {
      ra[u] = zero4<T>();
      int kk = k0 + 4 * a_ch[u];
      if (arow[u] >= 0 && kk < ktot) {
        int64_t src = arow[u];
        int c = kk;
        if (a.nbr) {
          int d = kk / a.cin;
          c = kk - d * a.cin;
          src = a.nbr[arow[u] * a.kvol + d];
        }
        if (src >= 0) ra[u] = *reinterpret_cast<const V4*>(x + src * a.cin + c);
      }
    }
#pragma unroll
    for (int u = 0; u < B_LOADS; ++u) {
      rb[u] = zero4<T>();
      int kk = k0 + 4 * b_ch[u];
      int o = n0 + b_r[u];
      if (o < a.cout && kk < ktot) rb[u] = *reinterpret_cast<const V4*>(w + (int64_t)o * ktot + kk);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < A_LOADS; ++u) *reinterpret_cast<V4*>(sA + a_r[u] * GM_LS + 4 * a_ch[u]) = ra[u];
#pragma unroll
    for (int u = 0; u < B_LOADS; ++u) *reinterpret_cast<V4*>(sB + b_r[u] * GM_LS + 4 * b_ch[u]) = rb[u];
  };

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0);
  for (int step = 0; step < nsteps; ++step) {
    stash();
    __syncthreads();
    if (step + 1 < nsteps) issue(step + 1);
#pragma unroll
    for (int ks = 0; ks < GM_BK / 16; ++ks) {
      V4 xf = *reinterpret_cast<const V4*>(sA + (16 * wave + li) * GM_LS + 16 * ks + 4 * g);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        V4 wf = *reinterpret_cast<const V4*>(sB + (16 * j + li) * GM_LS + 16 * ks + 4 * g);
        acc[j] = mma16<T>(wf, xf, acc[j]);  // D[channel 4g+r][point li]
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns point (16*wave + li), channels n0 + 16j + 4g .. +3
  const int64_t prow = row0 + 16 * wave + li;
  if (prow >= a.m) return;
  const int64_t orow = a.row_order ? (int64_t)a.row_order[prow] : prow;
  T* out = reinterpret_cast<T*>(a.out);
  T* out2 = reinterpret_cast<T*>(a.out2);
  const T* res = reinterpret_cast<const T*>(a.res);
  const int64_t rrow = a.res ? (a.res_index ? (int64_t)a.res_index[orow] : orow) : 0;
  const bool vec_ok = (a.cout & 3) == 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch0 = n0 + 16 * j + 4 * g;
    if (ch0 >= a.cout) continue;
    float v[4], v2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = ch0 + r;
      float t = acc[j][r];
      if (ch < a.cout) {
        if (a.bias) t += a.bias[ch];
        if (a.bn_scale) t = t * a.bn_scale[ch] + a.bn_shift[ch];
        t = apply_act(t, a.act);
      }
      v[r] = t;
      v2[r] = t;
    }
    if (a.res) {
      if (vec_ok) {
        float rr[4];
        unpack4<T>(*reinterpret_cast<const V4*>(res + rrow * a.cout + ch0), rr);
#pragma unroll
        for (int r = 0; r < 4; ++r) v2[r] = v[r] + rr[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch0 + r < a.cout) v2[r] = v[r] + to_f32<T>(res[rrow * a.cout + ch0 + r]);
      }
    }
    if (vec_ok) {
      if (out2) {
        *reinterpret_cast<V4*>(out + orow * a.cout + ch0) = pack4<T>(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<V4*>(out2 + orow * a.cout + ch0) = pack4<T>(v2[0], v2[1], v2[2], v2[3]);
      } else {
        *reinterpret_cast<V4*>(out + orow * a.cout + ch0) = pack4<T>(v2[0], v2[1], v2[2], v2[3]);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ch0 + r < a.cout) {
          if (out2) {
            out[orow * a.cout + ch0 + r] = from_f32<T>(v[r]);
            out2[orow * a.cout + ch0 + r] = from_f32<T>(v2[r]);
          } else {
            out[orow * a.cout + ch0 + r] = from_f32<T>(v2[r]);
          }
        }
    }
  }
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_gemm(const void* x, const void* w, void* out, int64_t m, int cin, int cout, int kvol,
                         const int32_t* nbr, const int32_t* row_order, const float* bias,
                         const float* bn_scale, const float* bn_shift, int act, const void* res,
                         const int32_t* res_index, void* out2, int dtype, void* stream) {
  PTV3_REQUIRE(cin > 0 && cin % 4 == 0, "gemm: cin=%d must be a positive multiple of 4", cin);
  PTV3_REQUIRE(cout > 0, "gemm: cout=%d", cout);
  PTV3_REQUIRE(kvol >= 1 && (kvol == 1 || nbr != nullptr), "gemm: kvol=%d needs a neighbour table", kvol);
  PTV3_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "gemm: bn_scale/bn_shift must come together");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "gemm: bad dtype %d", dtype);
  PTV3_REQUIRE(out2 == nullptr || res != nullptr, "gemm: out2 without res");
  if (m == 0) return PTV3_OK;
  GemmArgs a{x, w, out, out2, res, nbr, row_order, res_index, bias, bn_scale, bn_shift, m, cin, cout, kvol, act};
  hipStream_t s = (hipStream_t)stream;
  const int bn = (cout % 64 == 0) ? 64 : 32;
  dim3 grid((unsigned)cdiv(m, GM_BM), (unsigned)cdiv(cout, bn));
  if (dtype == PTV3_F32) {
    if (bn == 64) hipLaunchKernelGGL((gemm_kernel<float, 64>), grid, dim3(GM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<float, 32>), grid, dim3(GM_THREADS), 0, s, a);
  } else {
    if (bn == 64) hipLaunchKernelGGL((gemm_kernel<__bf16, 64>), grid, dim3(GM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<__bf16, 32>), grid, dim3(GM_THREADS), 0, s, a);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
