// Shared device helpers for the PTv3 hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#define PTV3_OK 0
#define PTV3_ERR_ARG 1
#define PTV3_ERR_LAUNCH 2
#define PTV3_ERR_UNSUPPORTED 3

#define PTV3_F32 0
#define PTV3_BF16 1

namespace ptv3 {

void set_error(const char* fmt, ...);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 4 contiguous elements of T as one register group
template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef f32x4 type; };
template <> struct Vec4<__bf16> { typedef s16x4 type; };

template <typename T> __device__ __forceinline__ typename Vec4<T>::type zero4();
template <> __device__ __forceinline__ f32x4 zero4<float>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
template <> __device__ __forceinline__ s16x4 zero4<__bf16>() { return s16x4{0, 0, 0, 0}; }

__device__ __forceinline__ short bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<short*>(&b);
}
__device__ __forceinline__ float bf16_to_f32(short s) {
  unsigned u = ((unsigned)(unsigned short)s) << 16;
  return __builtin_bit_cast(float, u);
}

template <typename T> __device__ __forceinline__ typename Vec4<T>::type pack4(float a, float b, float c, float d);
template <> __device__ __forceinline__ f32x4 pack4<float>(float a, float b, float c, float d) { return f32x4{a, b, c, d}; }
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <> __device__ __forceinline__ s16x4 pack4<__bf16>(float a, float b, float c, float d) {
  // two v_cvt_pk_bf16_f32 (vector conversion), not four scalar converts + permutes
  bf16x2 lo = __builtin_convertvector(f32x2{a, b}, bf16x2);
  bf16x2 hi = __builtin_convertvector(f32x2{c, d}, bf16x2);
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 w = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
  return __builtin_bit_cast(s16x4, w);
}
template <typename T> __device__ __forceinline__ void unpack4(typename Vec4<T>::type v, float* o);
template <> __device__ __forceinline__ void unpack4<float>(f32x4 v, float* o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
template <> __device__ __forceinline__ void unpack4<__bf16>(s16x4 v, float* o) {
  o[0] = bf16_to_f32(v[0]); o[1] = bf16_to_f32(v[1]); o[2] = bf16_to_f32(v[2]); o[3] = bf16_to_f32(v[3]);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float v) { return (__bf16)v; }

// One 16x16x16 matrix-core step, D = A*B + C, shared lane map for both dtypes:
// lane l, i = l & 15, g = l >> 4 holds A[i][4g..4g+3], B[4g..4g+3][i]; acc reg r is D[4g+r][i].
// f32: 4 x v_mfma_f32_16x16x4_f32 (exact fmaf chain); bf16: 1 x v_mfma_f32_16x16x16_bf16.
template <typename T>
__device__ __forceinline__ f32x4 mma16(typename Vec4<T>::type a, typename Vec4<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma16<float>(f32x4 a, f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  return c;
}
template <>
__device__ __forceinline__ f32x4 mma16<__bf16>(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Two consecutive mma16 steps over K as ONE call: D = A0*B0 + A1*B1 + C.  bf16: a single v_mfma_f32_16x16x32_bf16
// (lane (i, g) holds K elements 8g..8g+7 = the two 4-element fragments back to back: the K order inside the 32
// is free as long as A and B agree), 16 cycles for twice the work of the 16x16x16 form; fp32: the same two exact
// 16x16x4 chains as two mma16 calls (bitwise identical to them).
template <typename T>
__device__ __forceinline__ f32x4 mma16x2(typename Vec4<T>::type a0, typename Vec4<T>::type a1,
                                         typename Vec4<T>::type b0, typename Vec4<T>::type b1, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma16x2<float>(f32x4 a0, f32x4 a1, f32x4 b0, f32x4 b1, f32x4 c) {
  return mma16<float>(a1, b1, mma16<float>(a0, b0, c));
}
template <>
__device__ __forceinline__ f32x4 mma16x2<__bf16>(s16x4 a0, s16x4 a1, s16x4 b0, s16x4 b1, f32x4 c) {
  const s16x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
  const s16x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 16-byte fragment of T: 4 fp32 or 8 bf16 consecutive K elements
template <typename T> struct Frag;
template <> struct Frag<float> {
  typedef f32x4 type;
  static constexpr int E = 4;    // elements per lane per fragment
  static constexpr int KC = 16;  // K covered by one matrix-core chunk (4 lane groups x E)
  static __device__ __forceinline__ f32x4 zero() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
  // 4 x v_mfma_f32_16x16x4_f32: exact fp32 fma chain
  static __device__ __forceinline__ f32x4 mma(f32x4 a, f32x4 b, f32x4 c) { return mma16<float>(a, b, c); }
};
template <> struct Frag<__bf16> {
  typedef s16x8 type;
  static constexpr int E = 8;
  static constexpr int KC = 32;
  static __device__ __forceinline__ s16x8 zero() { return s16x8{0, 0, 0, 0, 0, 0, 0, 0}; }
  // 1 x v_mfma_f32_16x16x32_bf16: lane (i, g) holds A[i][8g..8g+7], B[8g..8g+7][i]
  static __device__ __forceinline__ f32x4 mma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
  }
};

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, far inside the 1e-4 parity budget): one v_rcp and
// one v_exp instead of libm's ~45-instruction erff -- the exact-GELU epilogue of fc1 was VALU-bound on erff.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  return copysignf(fmaf(-p, e, 1.0f), x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each
// XCD a contiguous run of logical ids (speed only, never correctness).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  if (nwg < nx * 2) return bid;
  unsigned xcd = bid % nx, slot = bid / nx;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// Attention dropout (reference: torch.nn.Dropout on the attention probabilities, point_transformer_v3m1_base.py:203, or
// flash_attn's dropout_p, :211): keep / drop of the pair (query slot, key slot) is a counter-based hash of
// id = ((padded query slot * heads + head) << 14 | key slot in its window) and the call's seed, so the forward kernel
// and the two backward kernels regenerate the same mask without storing it (the reference's Philox stream is not
// reproduced: the mask is equal in distribution, not in value; tests rebuild it on the host from this function).
__device__ __forceinline__ bool drop_keep(unsigned long long id, unsigned seed, unsigned thr) {
  unsigned x = ((unsigned)id * 0x9E3779B1u) ^ ((unsigned)(id >> 32) * 0x85EBCA77u) ^ seed;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x >= thr;   // P(drop) = thr / 2^32
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a function: set it once per (function,
// device), under a lock (a process may drive several GPUs, and several host threads one GPU).
inline void ensure_dynamic_lds(const void* fn, int bytes) {
  constexpr int MAX_DEV = 64, MAX_FN = 256;
  static std::mutex mu;
  static const void* fns[MAX_FN];
  static int maxb[MAX_FN][MAX_DEV];
  static int nfn = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    return;
  }
  std::lock_guard<std::mutex> lock(mu);
  int i = 0;
  while (i < nfn && fns[i] != fn) ++i;
  if (i == nfn) {
    if (nfn == MAX_FN) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); return; }
    fns[nfn++] = fn;
  }
  if (maxb[i][dev] < bytes) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    maxb[i][dev] = bytes;
  }
}

}  // namespace ptv3

#define PTV3_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      ptv3::set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return PTV3_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

#define PTV3_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ptv3::set_error(__VA_ARGS__);        \
      return PTV3_ERR_ARG;                 \
    }                                      \
  } while (0)
