// Argument blocks and register-chain helpers shared by the fused halves of Block.forward
// (block_fused.hip: wave-local / cooperative variants; block_wide.hip: the LDS-DMA weight-streaming variant).
#pragma once
#include "common.h"

namespace ptv3 {

struct HeadArgs {
  const void* x; const float* slab; int splits; const float* conv_bias;  // x XOR slab
  const void* shortcut;
  const float *g0, *b0, *g1, *b1;
  const void* wqkv; const float* bqkv;
  void* f1; void* qkv;
  int64_t m; float eps;
};

struct TailArgs {
  const void* attn; const void* f1;
  const void* wproj; const float* bproj;
  const float *g2, *b2;
  const void* w1; const float* bias1; const void* w2; const float* bias2;
  void* out;
  int64_t m; int hidden; float eps;
  int ablate = 0;   // tools only (PTV3_WIDE_ABLATE): bit 0 no GELU, 1 no matrix-core work, 2 no weight copies, 3 no waits / barriers in the MLP loop
};

__device__ __forceinline__ float groups_sum(float x) {
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}

// the B fragment of K-chunk `kc` from accumulator-layout tiles t[] (already rounded to T)
template <typename T, int NT> struct ChainFrag;
template <int NT> struct ChainFrag<float, NT> {
  static constexpr int NKC = NT;  // 16-channel chunks
  static __device__ __forceinline__ f32x4 get(const f32x4* t, int kc) { return t[kc]; }
};
template <int NT> struct ChainFrag<__bf16, NT> {
  static constexpr int NKC = NT / 2;  // 32-channel chunks = two tiles
  static __device__ __forceinline__ s16x8 get(const f32x4* t, int kc) {
    s16x4 lo = pack4<__bf16>(t[2 * kc][0], t[2 * kc][1], t[2 * kc][2], t[2 * kc][3]);
    s16x4 hi = pack4<__bf16>(t[2 * kc + 1][0], t[2 * kc + 1][1], t[2 * kc + 1][2], t[2 * kc + 1][3]);
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
};

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__bf16>(float v) { return (float)(__bf16)v; }

// LayerNorm over the C = 16*NT channels of the lane's point; values in accumulator layout v[j][r] = channel 16j+4g+r
// (any layout in which the four lanes li, li+16, li+32, li+48 hold the row between them works the same)
template <int NT>
__device__ __forceinline__ void row_norm(const f32x4* v, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j) s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
  mean = groups_sum(s) * (1.0f / (16 * NT));
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { float d = v[j][r] - mean; q += d * d; }
  rstd = rsqrtf(groups_sum(q) * (1.0f / (16 * NT)) + eps);
}

// block_wide.hip: the weight-streaming variant for c in {128, 256} at large m
bool wide_capable(int c, int hidden, int dtype);
int64_t wide_min_rows(int c);
void launch_block_head_wide(const HeadArgs& a, int c, int dtype, hipStream_t s);
void launch_block_tail_wide(const TailArgs& a, int c, int dtype, hipStream_t s);

}  // namespace ptv3
