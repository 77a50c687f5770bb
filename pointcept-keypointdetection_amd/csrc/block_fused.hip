// Row-local halves of Block.forward fused around the two non-local ops (sparse conv, window attention):
//
//   block_head:  x = conv output (or the sum of its split-K slabs + bias)
//                f1 = LayerNorm_cpe(x) + shortcut ; qkv = Linear_qkv(LayerNorm_1(f1))          (:319-324, 188)
//   block_tail:  f2 = Linear_proj(attn) + f1 ; out = f2 + fc2(GELU(fc1(LayerNorm_2(f2))))       (:219, 326-334)
//
// One wave owns 16 points end to end; there is no LDS and no barrier.  GEMMs are chained in registers:
// with the product computed as W_tile * X^T a lane holds 4 consecutive channels of ONE point per 16-channel
// tile, and two such accumulator tiles, packed to bf16, ARE the 8-element B fragment of the next
// v_mfma_f32_16x16x32_bf16 -- provided the next weight matrix has its input channels permuted inside every
// 32-chunk as [0-3,16-19,4-7,20-23,8-11,24-27,12-15,28-31] (done once per weight version on the host;
// fp32 uses 16-wide chunks where the order is the identity).  LayerNorm statistics are an in-lane sum plus
// two cross-lane-group exchanges.  Weight fragments stream from L2 straight into registers.
// Used for C in {32, 64}: there M is large and the chain is bandwidth/launch bound (measured at 100k points,
// C=64: head 30 us vs 37 us unfused, tail 49 us vs 75 us).  At C >= 128 the levels are small (M ~ 10^2..10^3
// rows): a wave would walk > 10^3 dependent weight-fragment loads alone and parallelism has to come from the
// output-channel dimension instead, i.e. from the tiled GEMM kernel (measured C=256, M=1388: 299 us fused vs
// 40 us unfused), so wider blocks keep the unfused kernels.
#include "common.h"
#include "block_args.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "profile.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

template <typename T> struct WPad { static constexpr int V = 16 / sizeof(T); };

template <typename T>
__device__ __forceinline__ void stage_matrix(T* dst, const T* __restrict__ src, int rows, int cols, int ld) {
  typedef typename Frag<T>::type FR;
  constexpr int E = Frag<T>::E;
  const int cpr = cols / E;
  for (int u = threadIdx.x; u < rows * cpr; u += blockDim.x) {
    const int r = u / cpr, ch = u % cpr;
    *reinterpret_cast<FR*>(dst + r * ld + E * ch) = *reinterpret_cast<const FR*>(src + (int64_t)r * cols + E * ch);
  }
}

template <typename T, int NT, bool WLDS>
__global__ void __launch_bounds__(256) block_head_kernel(HeadArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  typedef typename Vec4<T>::type V4;
  constexpr int C = 16 * NT, E = F::E, KC = F::KC;
  constexpr int NKC = ChainFrag<T, NT>::NKC;
  extern __shared__ __attribute__((aligned(16))) char head_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const T* w = reinterpret_cast<const T*>(a.wqkv);
  int ldc = C;
  // The per-channel vectors (two LayerNorm affines, qkv bias, conv bias) live in LDS next to the weights: read from
  // global inside the row loop, each read that follows a store waits for that store to be acknowledged (loads and
  // stores retire through one in-order counter) - LDS reads do not.
  float* sVec = reinterpret_cast<float*>(head_smem + (WLDS ? (size_t)3 * C * (C + WPad<T>::V) * sizeof(T) : 0));
  const float *vg0 = sVec, *vb0 = sVec + C, *vg1 = sVec + 2 * C, *vb1 = sVec + 3 * C, *vbq = sVec + 4 * C,
              *vcb = sVec + 7 * C;
  for (int i = threadIdx.x; i < C; i += 256) {
    sVec[i] = a.g0[i]; sVec[C + i] = a.b0[i]; sVec[2 * C + i] = a.g1[i]; sVec[3 * C + i] = a.b1[i];
    sVec[7 * C + i] = a.conv_bias ? a.conv_bias[i] : 0.f;
  }
  for (int i = threadIdx.x; i < 3 * C; i += 256) sVec[4 * C + i] = a.bqkv[i];
  if (WLDS) {   // the 3C x C qkv weight resident in LDS for all row blocks of this workgroup
    ldc = C + WPad<T>::V;
    T* sW = reinterpret_cast<T*>(head_smem);
    stage_matrix<T>(sW, w, 3 * C, C, ldc);
    w = sW;
  }
  __syncthreads();
  const int64_t nblocks = (a.m + 63) / 64;
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
  const int64_t row = (blk * 4 + wave) * 16 + li;
  if ((blk * 4 + wave) * 16 >= a.m) continue;  // whole wave out of range
  const bool valid = row < a.m;
  const int64_t rc = valid ? row : a.m - 1;
  const T* sc = reinterpret_cast<const T*>(a.shortcut);

  f32x4 v[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch = 16 * j + 4 * g;
    if (a.slab) {
      f32x4 acc = *reinterpret_cast<const f32x4*>(vcb + ch);
      for (int z = 0; z < a.splits; ++z)
        acc += *reinterpret_cast<const f32x4*>(a.slab + ((int64_t)z * a.m + rc) * C + ch);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j][r] = round_to<T>(acc[r]);
    } else {
      float t[4];
      unpack4<T>(*reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.x) + rc * C + ch), t);
      v[j] = f32x4{t[0], t[1], t[2], t[3]};
    }
  }
  float mean, rstd;
  row_norm<NT>(v, a.eps, mean, rstd);
  T* f1 = reinterpret_cast<T*>(a.f1);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch = 16 * j + 4 * g;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(vg0 + ch), bt = *reinterpret_cast<const f32x4*>(vb0 + ch);
    float s[4];
    unpack4<T>(*reinterpret_cast<const V4*>(sc + rc * C + ch), s);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[j][r] = round_to<T>((v[j][r] - mean) * rstd * gm[r] + bt[r] + s[r]);
    if (valid) *reinterpret_cast<V4*>(f1 + row * C + ch) = pack4<T>(v[j][0], v[j][1], v[j][2], v[j][3]);
  }
  row_norm<NT>(v, a.eps, mean, rstd);
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ch = 16 * j + 4 * g;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(vg1 + ch), bt = *reinterpret_cast<const f32x4*>(vb1 + ch);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[j][r] = round_to<T>((v[j][r] - mean) * rstd * gm[r] + bt[r]);
  }
  FR xf[NKC];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) xf[kc] = ChainFrag<T, NT>::get(v, kc);

  // qkv = t3 @ Wqkv^T + b : 3*NT output tiles, 4 at a time
  T* qkv = reinterpret_cast<T*>(a.qkv);
  constexpr int OT = 3 * NT;
  for (int o0 = 0; o0 < OT; o0 += 4) {
    f32x4 acc[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (o0 + jj < OT) {
          FR wa = *reinterpret_cast<const FR*>(w + (16 * (o0 + jj) + li) * ldc + KC * kc + E * g);
          acc[jj] = F::mma(wa, xf[kc], acc[jj]);
        }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (o0 + jj < OT && valid) {
        const int ch = 16 * (o0 + jj) + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4*>(vbq + ch);
        *reinterpret_cast<V4*>(qkv + row * (3 * C) + ch) =
            pack4<T>(acc[jj][0] + b[0], acc[jj][1] + b[1], acc[jj][2] + b[2], acc[jj][3] + b[3]);
      }
  }
  }  // row blocks
}

// RT = 16-point row tiles per wave: every weight fragment fetched from L1/L2 feeds RT matrix-core steps and the
// RT chains are independent (the single-tile form is latency-bound: one MFMA per fragment load).
// WLDS: the 9*C^2 weights are staged once per workgroup into LDS (rows padded by 16 bytes) and the workgroup walks
// row blocks in a grid-stride loop: without it every wave re-streams all weights from L2 (72 KB per 32 points at
// C = 64: 225 MB of L2 traffic per 100k-point call, the real bound of the first version).
template <typename T, int NT, int RT, bool WLDS>
__global__ void __launch_bounds__(256) block_tail_kernel(TailArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  typedef typename Vec4<T>::type V4;
  constexpr int C = 16 * NT, E = F::E, KC = F::KC;
  constexpr int NKC = ChainFrag<T, NT>::NKC;
  constexpr int HKC = 64 / KC;  // K-chunks per 64-wide hidden slice (2 bf16 / 4 fp32)
  extern __shared__ __attribute__((aligned(16))) char tail_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const T* attn = reinterpret_cast<const T*>(a.attn);
  const T* f1 = reinterpret_cast<const T*>(a.f1);
  const T* wp = reinterpret_cast<const T*>(a.wproj);
  const T* w1 = reinterpret_cast<const T*>(a.w1);
  const T* w2 = reinterpret_cast<const T*>(a.w2);
  int ldc = C, ldh = a.hidden;   // row strides of (wproj, w1) and of w2
  // per-channel vectors in LDS (see block_head_kernel): no global read between the stores of the row loop
  const float *vbp = a.bproj, *vg2 = a.g2, *vb2 = a.b2, *vbias1 = a.bias1, *vbias2 = a.bias2;
  if (WLDS) {
    ldc = C + WPad<T>::V;
    ldh = a.hidden + WPad<T>::V;
    T* sWp = reinterpret_cast<T*>(tail_smem);
    T* sW1 = sWp + C * ldc;
    T* sW2 = sW1 + a.hidden * ldc;
    float* sVec = reinterpret_cast<float*>(sW2 + C * ldh);   // 16-byte aligned: every part above is a multiple of 16 bytes
    stage_matrix<T>(sWp, wp, C, C, ldc);
    stage_matrix<T>(sW1, w1, a.hidden, C, ldc);
    stage_matrix<T>(sW2, w2, C, a.hidden, ldh);
    for (int i = threadIdx.x; i < C; i += 256) {
      sVec[i] = a.bproj[i]; sVec[C + i] = a.g2[i]; sVec[2 * C + i] = a.b2[i]; sVec[3 * C + i] = a.bias2[i];
    }
    for (int i = threadIdx.x; i < a.hidden; i += 256) sVec[4 * C + i] = a.bias1[i];
    __syncthreads();
    wp = sWp; w1 = sW1; w2 = sW2;
    vbp = sVec; vg2 = sVec + C; vb2 = sVec + 2 * C; vbias2 = sVec + 3 * C; vbias1 = sVec + 4 * C;
  }
  const int64_t nblocks = (a.m + 64 * RT - 1) / (64 * RT);
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
  const int64_t base = (blk * 4 + wave) * (16 * RT);
  if (base >= a.m) continue;
  int64_t row[RT], rc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    row[t] = base + 16 * t + li;
    rc[t] = row[t] < a.m ? row[t] : a.m - 1;
  }

  // ---- f2 = attn @ Wproj^T + b + f1   (attn rows are in natural channel order: natural Wproj)
  f32x4 f2[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) f2[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < C / KC; ++kc) {
    FR xb[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) xb[t] = *reinterpret_cast<const FR*>(attn + rc[t] * C + KC * kc + E * g);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      FR wa = *reinterpret_cast<const FR*>(wp + (16 * j + li) * ldc + KC * kc + E * g);
#pragma unroll
      for (int t = 0; t < RT; ++t) f2[t][j] = F::mma(wa, xb[t], f2[t][j]);
    }
  }
  FR xf[RT][NKC];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    f32x4 t5[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = 16 * j + 4 * g;
      const f32x4 b = *reinterpret_cast<const f32x4*>(vbp + ch);
      float s[4];
      unpack4<T>(*reinterpret_cast<const V4*>(f1 + rc[t] * C + ch), s);
#pragma unroll
      for (int r = 0; r < 4; ++r) f2[t][j][r] = round_to<T>(f2[t][j][r] + b[r] + s[r]);
    }
    float mean, rstd;
    row_norm<NT>(f2[t], a.eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = 16 * j + 4 * g;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(vg2 + ch), bt = *reinterpret_cast<const f32x4*>(vb2 + ch);
#pragma unroll
      for (int r = 0; r < 4; ++r) t5[j][r] = round_to<T>((f2[t][j][r] - mean) * rstd * gm[r] + bt[r]);
    }
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) xf[t][kc] = ChainFrag<T, NT>::get(t5, kc);
  }

  // ---- out = f2 + fc2(GELU(fc1(t5))) : the hidden layer goes through registers 64 channels at a time
  f32x4 o[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) o[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weight fragments are fetched one stage ahead of their MFMAs (the kernel is otherwise parked on s_waitcnt:
  // 70 % of its wave-cycles in the PMC profile): fc1's for slice h0+64 while slice h0 is computed
  FR w1f[NKC][4];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      w1f[kc][jj] = *reinterpret_cast<const FR*>(w1 + (16 * jj + li) * ldc + KC * kc + E * g);
  for (int h0 = 0; h0 < a.hidden; h0 += 64) {
    FR w2f[HKC][NT];
#pragma unroll
    for (int mm = 0; mm < HKC; ++mm)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        w2f[mm][j] = *reinterpret_cast<const FR*>(w2 + (16 * j + li) * ldh + h0 + KC * mm + E * g);
    f32x4 bias1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) bias1[jj] = *reinterpret_cast<const f32x4*>(vbias1 + h0 + 16 * jj + 4 * g);
    f32x4 h[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) h[t][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < RT; ++t) h[t][jj] = F::mma(w1f[kc][jj], xf[t][kc], h[t][jj]);
    const int hn = h0 + 64 < a.hidden ? h0 + 64 : h0;   // the last slice re-reads itself (result unused)
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        w1f[kc][jj] = *reinterpret_cast<const FR*>(w1 + (hn + 16 * jj + li) * ldc + KC * kc + E * g);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][jj][r] = round_to<T>(gelu_erf(h[t][jj][r] + bias1[jj][r]));
#pragma unroll
    for (int mm = 0; mm < HKC; ++mm) {
      FR hf[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) hf[t] = ChainFrag<T, 4>::get(h[t], mm);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < RT; ++t) o[t][j] = F::mma(w2f[mm][j], hf[t], o[t][j]);
    }
  }
  T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    if (row[t] >= a.m) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = 16 * j + 4 * g;
      const f32x4 b = *reinterpret_cast<const f32x4*>(vbias2 + ch);
      *reinterpret_cast<V4*>(out + row[t] * C + ch) =
          pack4<T>(o[t][j][0] + b[0] + f2[t][j][0], o[t][j][1] + b[1] + f2[t][j][1], o[t][j][2] + b[2] + f2[t][j][2],
                   o[t][j][3] + b[3] + f2[t][j][3]);
    }
  }
  }  // row blocks
}

// ------------------------------------------------------------------------------------------------
// Row-local two-layer MLP with the hidden layer kept in registers (the dense keypoint head,
// offset_keypoint_ptv3.py:26-31: Linear -> BatchNorm1d -> ReLU -> Linear): per wave 16 points,
//   h = act((x W1^T + b1) * s1 + t1)  64 hidden channels at a time,  out += h W2^T,  out + b2.
// The (m, hidden) intermediate (51 MB at 100k points x 256 bf16) never exists.
// ------------------------------------------------------------------------------------------------
struct Mlp2Args {
  const void* x; const void* w1; const float* b1; const float* s1; const float* t1;
  const void* w2; const float* b2; void* out;
  int64_t m; int hidden, cout, act, out_f32;
};

template <typename T, int NTI, int NTO, int RT, bool WLDS>
__global__ void __launch_bounds__(256) mlp2_kernel(Mlp2Args a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  typedef typename Vec4<T>::type V4;
  constexpr int CI = 16 * NTI, E = F::E, KC = F::KC;
  constexpr int HKC = 64 / KC;
  extern __shared__ __attribute__((aligned(16))) char mlp_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* w1 = reinterpret_cast<const T*>(a.w1);
  const T* w2 = reinterpret_cast<const T*>(a.w2);
  int ld1 = CI, ld2 = a.hidden;
  if (WLDS) {   // both weight matrices resident in LDS for all row blocks of this workgroup
    ld1 = CI + WPad<T>::V;
    ld2 = a.hidden + WPad<T>::V;
    T* sW1 = reinterpret_cast<T*>(mlp_smem);
    T* sW2 = sW1 + a.hidden * ld1;
    stage_matrix<T>(sW1, w1, a.hidden, CI, ld1);
    stage_matrix<T>(sW2, w2, 16 * NTO, a.hidden, ld2);
    __syncthreads();
    w1 = sW1; w2 = sW2;
  }
  const int64_t nblocks = (a.m + 64 * RT - 1) / (64 * RT);
  for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
  const int64_t base = (blk * 4 + wave) * (16 * RT);
  if (base >= a.m) continue;
  int64_t row[RT];
  FR xf[RT][CI / KC];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    row[t] = base + 16 * t + li;
    const int64_t rc = row[t] < a.m ? row[t] : a.m - 1;
#pragma unroll
    for (int kc = 0; kc < CI / KC; ++kc) xf[t][kc] = *reinterpret_cast<const FR*>(x + rc * CI + KC * kc + E * g);
  }
  f32x4 o[RT][NTO];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < NTO; ++j) o[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  FR w1f[CI / KC][4];
#pragma unroll
  for (int kc = 0; kc < CI / KC; ++kc)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      w1f[kc][jj] = *reinterpret_cast<const FR*>(w1 + (16 * jj + li) * ld1 + KC * kc + E * g);
  for (int h0 = 0; h0 < a.hidden; h0 += 64) {
    FR w2f[HKC][NTO];
#pragma unroll
    for (int mm = 0; mm < HKC; ++mm)
#pragma unroll
      for (int j = 0; j < NTO; ++j)
        w2f[mm][j] = *reinterpret_cast<const FR*>(w2 + (16 * j + li) * ld2 + h0 + KC * mm + E * g);
    f32x4 h[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) h[t][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < CI / KC; ++kc)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < RT; ++t) h[t][jj] = F::mma(w1f[kc][jj], xf[t][kc], h[t][jj]);
    const int hn = h0 + 64 < a.hidden ? h0 + 64 : h0;
#pragma unroll
    for (int kc = 0; kc < CI / KC; ++kc)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        w1f[kc][jj] = *reinterpret_cast<const FR*>(w1 + (hn + 16 * jj + li) * ld1 + KC * kc + E * g);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int ch = h0 + 16 * jj + 4 * g;
      const f32x4 b = a.b1 ? *reinterpret_cast<const f32x4*>(a.b1 + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 sc = a.s1 ? *reinterpret_cast<const f32x4*>(a.s1 + ch) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 sh = a.t1 ? *reinterpret_cast<const f32x4*>(a.t1 + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = (h[t][jj][r] + b[r]) * sc[r] + sh[r];
          if (a.act == PTV3_ACT_GELU) v = gelu_erf(v);
          else if (a.act == PTV3_ACT_RELU) v = fmaxf(v, 0.f);
          h[t][jj][r] = round_to<T>(v);
        }
    }
#pragma unroll
    for (int mm = 0; mm < HKC; ++mm) {
      FR hf[RT];
#pragma unroll
      for (int t = 0; t < RT; ++t) hf[t] = ChainFrag<T, 4>::get(h[t], mm);
#pragma unroll
      for (int j = 0; j < NTO; ++j)
#pragma unroll
        for (int t = 0; t < RT; ++t) o[t][j] = F::mma(w2f[mm][j], hf[t], o[t][j]);
    }
  }
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    if (row[t] >= a.m) continue;
#pragma unroll
    for (int j = 0; j < NTO; ++j) {
      const int ch = 16 * j + 4 * g;
      if (ch >= a.cout) continue;   // cout is a multiple of 4
      const f32x4 b = a.b2 ? *reinterpret_cast<const f32x4*>(a.b2 + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 v = f32x4{o[t][j][0] + b[0], o[t][j][1] + b[1], o[t][j][2] + b[2], o[t][j][3] + b[3]};
      if (a.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + row[t] * a.cout + ch) = v;
      else *reinterpret_cast<V4*>(reinterpret_cast<T*>(a.out) + row[t] * a.cout + ch) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
  }
  }  // row blocks
}

// ------------------------------------------------------------------------------------------------
// Workgroup-cooperative variants for the SMALL levels (C = 128 / 256 / 512, M <= a few thousand rows):
// one workgroup owns 16 points; its NW = min(16, C/16) waves split the OUTPUT channels of every GEMM
// of the chain (weights are read once per workgroup, each wave its share), activations hop between
// the GEMMs through LDS in natural channel order (natural weight layout), 2-3 barriers per kernel.
// With so few rows the chain is latency-bound: every wave issues 16 weight fragments before its
// first MFMA and 4 waves per SIMD cover the rest.
// ------------------------------------------------------------------------------------------------
template <typename T> struct CoopPad { static constexpr int V = 16 / sizeof(T); };  // +16 bytes per LDS row
template <int C> struct Coop {
  static constexpr int NW = C / 16 < 16 ? C / 16 : 16;  // waves per workgroup
  static constexpr int LPR = 4 * NW;                    // lanes per row in the LayerNorm stages
  static constexpr int CPL = C / LPR;                   // channels per lane there (4 or 8)
};

// acc[t] (t < NTW) += W[tile q0 + NW*t][K] . act[16 rows][K]; act rows have stride `as` (LDS or global)
template <typename T, int NTW, int NW>
__device__ __forceinline__ void coop_tiles(f32x4 (&acc)[NTW], const T* __restrict__ w, int K, int ntiles, int q0,
                                           const T* act, int as, int li, int g) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  constexpr int E = F::E, KC = F::KC;
  constexpr int U = NTW >= 5 ? 2 : NTW >= 3 ? 4 : NTW == 2 ? 8 : 16;  // <= 16 fragments in flight
  const T* wp[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int q = q0 + NW * t < ntiles ? q0 + NW * t : q0;  // inactive tiles alias an active one (results dropped)
    wp[t] = w + (int64_t)(16 * q + li) * K + E * g;
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const T* ap = act + li * as + E * g;
  const int nkc = K / KC;
  for (int kc0 = 0; kc0 < nkc; kc0 += U) {
    FR wa[NTW][U];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        wa[t][u] = *reinterpret_cast<const FR*>(wp[t] + KC * (kc0 + u < nkc ? kc0 + u : kc0));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (kc0 + u < nkc) {
        const FR xf = *reinterpret_cast<const FR*>(ap + KC * (kc0 + u));
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = F::mma(wa[t][u], xf, acc[t]);
      }
    }
  }
}

template <int CPL, int LPR>
__device__ __forceinline__ void coop_row_stats(const float (&v)[CPL], int C, float eps, float& mean, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < CPL; ++e) s += v[e];
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) s += __shfl_xor(s, d, 64);
  mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < CPL; ++e) { float dd = v[e] - mean; q += dd * dd; }
#pragma unroll
  for (int d = 1; d < LPR; d <<= 1) q += __shfl_xor(q, d, 64);
  rstd = rsqrtf(q / (float)C + eps);
}

template <typename T, int C>
__global__ void __launch_bounds__(64 * Coop<C>::NW) block_head_coop_kernel(HeadArgs a) {
  typedef typename Vec4<T>::type V4;
  constexpr int NW = Coop<C>::NW, LPR = Coop<C>::LPR, CPL = Coop<C>::CPL;
  constexpr int TS = C + CoopPad<T>::V;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* sT = reinterpret_cast<T*>(smem);  // [16][TS]  LN1(f1)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  // ---- stage 0: x (or slab sum + bias) -> LN0 + shortcut -> f1 ; LN1(f1) -> sT
  {
    const int rr = threadIdx.x / LPR, ll = threadIdx.x % LPR;  // row of the tile, lane inside the row
    const int64_t row = r0 + rr;
    const bool valid = row < a.m;
    const int64_t rc = valid ? row : a.m - 1;
    const int c0 = ll * CPL;
    float v[CPL];
#pragma unroll
    for (int e4 = 0; e4 < CPL / 4; ++e4) {
      const int ch = c0 + 4 * e4;
      if (a.slab) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(a.conv_bias + ch);
        for (int z = 0; z < a.splits; ++z)
          acc += *reinterpret_cast<const f32x4*>(a.slab + ((int64_t)z * a.m + rc) * C + ch);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * e4 + r] = round_to<T>(acc[r]);
      } else {
        float t[4];
        unpack4<T>(*reinterpret_cast<const V4*>(reinterpret_cast<const T*>(a.x) + rc * C + ch), t);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * e4 + r] = t[r];
      }
    }
    float mean, rstd;
    coop_row_stats<CPL, LPR>(v, C, a.eps, mean, rstd);
    const T* sc = reinterpret_cast<const T*>(a.shortcut);
    T* f1 = reinterpret_cast<T*>(a.f1);
#pragma unroll
    for (int e4 = 0; e4 < CPL / 4; ++e4) {
      const int ch = c0 + 4 * e4;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.g0 + ch), bt = *reinterpret_cast<const f32x4*>(a.b0 + ch);
      float s4[4];
      unpack4<T>(*reinterpret_cast<const V4*>(sc + rc * C + ch), s4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * e4 + r] = round_to<T>((v[4 * e4 + r] - mean) * rstd * gm[r] + bt[r] + s4[r]);
      if (valid) *reinterpret_cast<V4*>(f1 + row * C + ch) = pack4<T>(v[4 * e4], v[4 * e4 + 1], v[4 * e4 + 2], v[4 * e4 + 3]);
    }
    coop_row_stats<CPL, LPR>(v, C, a.eps, mean, rstd);
#pragma unroll
    for (int e4 = 0; e4 < CPL / 4; ++e4) {
      const int ch = c0 + 4 * e4;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.g1 + ch), bt = *reinterpret_cast<const f32x4*>(a.b1 + ch);
      *reinterpret_cast<V4*>(sT + rr * TS + ch) =
          pack4<T>((v[4 * e4] - mean) * rstd * gm[0] + bt[0], (v[4 * e4 + 1] - mean) * rstd * gm[1] + bt[1],
                   (v[4 * e4 + 2] - mean) * rstd * gm[2] + bt[2], (v[4 * e4 + 3] - mean) * rstd * gm[3] + bt[3]);
    }
  }
  __syncthreads();
  // ---- stage 1: qkv, 3C/16 tiles dealt round-robin to the waves
  constexpr int OT = 3 * C / 16, NTW = OT / NW;
  static_assert(OT % NW == 0, "qkv tiles must divide over the waves");
  f32x4 acc[NTW];
  coop_tiles<T, NTW, NW>(acc, reinterpret_cast<const T*>(a.wqkv), C, OT, wave, sT, TS, li, g);
  const int64_t row = r0 + li;
  if (row < a.m) {
    T* qkv = reinterpret_cast<T*>(a.qkv);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int ch = 16 * (wave + NW * t) + 4 * g;
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.bqkv + ch);
      *reinterpret_cast<V4*>(qkv + row * (3 * C) + ch) =
          pack4<T>(acc[t][0] + b[0], acc[t][1] + b[1], acc[t][2] + b[2], acc[t][3] + b[3]);
    }
  }
}

// HT = hidden / C (the mlp ratio, 1..4): tiles per wave of fc1 = HT * C / 16 / NW
template <typename T, int C, int HT>
__global__ void __launch_bounds__(64 * Coop<C>::NW) block_tail_coop_kernel(TailArgs a) {
  typedef typename Vec4<T>::type V4;
  constexpr int NW = Coop<C>::NW, LPR = Coop<C>::LPR, CPL = Coop<C>::CPL;
  constexpr int HID = HT * C;
  constexpr int FS = C + 4;                      // sF2 row stride (floats)
  constexpr int TS = C + CoopPad<T>::V;          // sT row stride
  constexpr int HS = HID + CoopPad<T>::V;        // sH row stride
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sF2 = reinterpret_cast<float*>(smem);   // [16][FS]  f2 (residual of the MLP)
  T* sT = reinterpret_cast<T*>(sF2 + 16 * FS);   // [16][TS]  LN2(f2)
  T* sH = sT + 16 * TS;                          // [16][HS]  GELU(fc1)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int64_t row = r0 + li;
  const bool valid = row < a.m;
  const int64_t rc = valid ? row : a.m - 1;
  constexpr int NTC = C / 16, TWC = NTC / NW;    // channel tiles, per wave
  // ---- stage A: f2 = attn @ Wproj^T + b + f1 (the activation rows come straight from global memory)
  {
    f32x4 acc[TWC];
    coop_tiles<T, TWC, NW>(acc, reinterpret_cast<const T*>(a.wproj), C, NTC, wave,
                           reinterpret_cast<const T*>(a.attn) + (rc - li) * C, C, li, g);
    const T* f1 = reinterpret_cast<const T*>(a.f1);
#pragma unroll
    for (int t = 0; t < TWC; ++t) {
      const int ch = 16 * (wave + NW * t) + 4 * g;
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.bproj + ch);
      float s4[4];
      unpack4<T>(*reinterpret_cast<const V4*>(f1 + rc * C + ch), s4);
      *reinterpret_cast<f32x4*>(sF2 + li * FS + ch) =
          f32x4{round_to<T>(acc[t][0] + b[0] + s4[0]), round_to<T>(acc[t][1] + b[1] + s4[1]),
                round_to<T>(acc[t][2] + b[2] + s4[2]), round_to<T>(acc[t][3] + b[3] + s4[3])};
    }
  }
  __syncthreads();
  // ---- LN2
  {
    const int rr = threadIdx.x / LPR, ll = threadIdx.x % LPR;
    const int c0 = ll * CPL;
    float v[CPL];
#pragma unroll
    for (int e4 = 0; e4 < CPL / 4; ++e4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(sF2 + rr * FS + c0 + 4 * e4);
      v[4 * e4] = t[0]; v[4 * e4 + 1] = t[1]; v[4 * e4 + 2] = t[2]; v[4 * e4 + 3] = t[3];
    }
    float mean, rstd;
    coop_row_stats<CPL, LPR>(v, C, a.eps, mean, rstd);
#pragma unroll
    for (int e4 = 0; e4 < CPL / 4; ++e4) {
      const int ch = c0 + 4 * e4;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.g2 + ch), bt = *reinterpret_cast<const f32x4*>(a.b2 + ch);
      *reinterpret_cast<V4*>(sT + rr * TS + ch) =
          pack4<T>((v[4 * e4] - mean) * rstd * gm[0] + bt[0], (v[4 * e4 + 1] - mean) * rstd * gm[1] + bt[1],
                   (v[4 * e4 + 2] - mean) * rstd * gm[2] + bt[2], (v[4 * e4 + 3] - mean) * rstd * gm[3] + bt[3]);
    }
  }
  __syncthreads();
  // ---- stage B: h = GELU(t5 @ W1^T + b1) -> sH
  {
    constexpr int NTH = HID / 16, TWH = NTH / NW;
    f32x4 acc[TWH];
    coop_tiles<T, TWH, NW>(acc, reinterpret_cast<const T*>(a.w1), C, NTH, wave, sT, TS, li, g);
#pragma unroll
    for (int t = 0; t < TWH; ++t) {
      const int ch = 16 * (wave + NW * t) + 4 * g;
      const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias1 + ch);
      *reinterpret_cast<V4*>(sH + li * HS + ch) =
          pack4<T>(gelu_erf(acc[t][0] + b[0]), gelu_erf(acc[t][1] + b[1]), gelu_erf(acc[t][2] + b[2]),
                   gelu_erf(acc[t][3] + b[3]));
    }
  }
  __syncthreads();
  // ---- stage C: out = h @ W2^T + b2 + f2
  {
    f32x4 acc[TWC];
    coop_tiles<T, TWC, NW>(acc, reinterpret_cast<const T*>(a.w2), HID, NTC, wave, sH, HS, li, g);
    if (valid) {
      T* out = reinterpret_cast<T*>(a.out);
#pragma unroll
      for (int t = 0; t < TWC; ++t) {
        const int ch = 16 * (wave + NW * t) + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias2 + ch);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(sF2 + li * FS + ch);
        *reinterpret_cast<V4*>(out + row * C + ch) =
            pack4<T>(acc[t][0] + b[0] + r2[0], acc[t][1] + b[1] + r2[1], acc[t][2] + b[2] + r2[2], acc[t][3] + b[3] + r2[3]);
      }
    }
  }
}

}  // namespace ptv3

using namespace ptv3;

// two 16-point row tiles per wave once there are enough waves to fill the chip several times over
static int row_tiles(int64_t m) {
  static int forced = -1;
  if (forced < 0) { const char* e = getenv("PTV3_FUSED_ROW_TILES"); forced = e ? atoi(e) : 0; }
  if (forced == 1 || forced == 2) return forced;
  return m >= 32768 ? 2 : 1;
}

static size_t tail_wlds_bytes(int c, int hidden, int esz) {
  const int pad = 16 / esz;
  return ((size_t)(c + hidden) * (c + pad) + (size_t)c * (hidden + pad)) * esz + (size_t)(4 * c + hidden) * sizeof(float);
}

template <typename T, int NT, int RT>
static void launch_tail(const TailArgs& a, int c, hipStream_t s) {
  const size_t wb = tail_wlds_bytes(c, a.hidden, sizeof(T));
  const int64_t nblocks = cdiv(a.m, 64 * RT);
  if (wb <= 96 * 1024) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(&block_tail_kernel<T, NT, RT, true>), 96 * 1024);
    // weights resident per workgroup: a few workgroups per CU, each walking many row blocks
    const int64_t grid = std::min<int64_t>(nblocks, 2 * 256);
    hipLaunchKernelGGL((block_tail_kernel<T, NT, RT, true>), dim3((unsigned)grid), dim3(256), wb, s, a);
  } else {
    hipLaunchKernelGGL((block_tail_kernel<T, NT, RT, false>), dim3((unsigned)nblocks), dim3(256), 0, s, a);
  }
}

#define TAIL_LAUNCH(ARGS)                                                                                \
  {                                                                                                      \
    const int rt = row_tiles(m);                                                                         \
    if (dtype == PTV3_F32) {                                                                             \
      if (c == 32) { if (rt == 2) launch_tail<float, 2, 2>(ARGS, c, s); else launch_tail<float, 2, 1>(ARGS, c, s); } \
      else { if (rt == 2) launch_tail<float, 4, 2>(ARGS, c, s); else launch_tail<float, 4, 1>(ARGS, c, s); }         \
    } else {                                                                                             \
      if (c == 32) { if (rt == 2) launch_tail<__bf16, 2, 2>(ARGS, c, s); else launch_tail<__bf16, 2, 1>(ARGS, c, s); } \
      else { if (rt == 2) launch_tail<__bf16, 4, 2>(ARGS, c, s); else launch_tail<__bf16, 4, 1>(ARGS, c, s); }         \
    }                                                                                                    \
  }

#define FUSED_LAUNCH(KERNEL, ARGS)                                                                     \
  {                                                                                                    \
    dim3 grid((unsigned)cdiv(m, 64)), block(256);                                                      \
    if (dtype == PTV3_F32) {                                                                           \
      if (c == 32) hipLaunchKernelGGL((KERNEL<float, 2>), grid, block, 0, s, ARGS);                    \
      else hipLaunchKernelGGL((KERNEL<float, 4>), grid, block, 0, s, ARGS);                            \
    } else {                                                                                           \
      if (c == 32) hipLaunchKernelGGL((KERNEL<__bf16, 2>), grid, block, 0, s, ARGS);                   \
      else hipLaunchKernelGGL((KERNEL<__bf16, 4>), grid, block, 0, s, ARGS);                           \
    }                                                                                                  \
  }

// 0: not fusable, 1: wave-local register chain (C in {32,64}; bf16 wants chain-permuted weights),
// 2: workgroup-cooperative (C in {128,256,512}; natural weights)
static size_t coop_head_lds(int c, int esz) { return (size_t)16 * (c + 16 / esz) * esz; }
static size_t coop_tail_lds(int c, int hidden, int esz) {
  return (size_t)16 * (c + 4) * 4 + (size_t)16 * (c + 16 / esz) * esz + (size_t)16 * (hidden + 16 / esz) * esz;
}

// the cooperative variant re-reads the block's weights once per 16 points: past this many rows the
// 64-row tiles of ptv3_gemm stream less
static int64_t coop_max_rows(int c) {
  static int64_t lim[3] = {-1, -1, -1};
  const int i = c == 128 ? 0 : c == 256 ? 1 : 2;
  if (lim[i] < 0) {
    // measured on the 100k-point scene (tools/coop_sweep.sh): c=128 gains 3%, c=256 is neutral, c=512 loses --
    // one CU cannot stream 9*c^2 weights fast enough, the separate GEMM launches spread them over the chip
    const int64_t dflt[3] = {16384, 0, 0};
    char name[32];
    snprintf(name, sizeof name, "PTV3_COOP_ROWS_%d", c);
    const char* e = getenv(name);
    lim[i] = e ? atoll(e) : dflt[i];
  }
  return lim[i];
}

// 3: the weight-streaming variant of block_wide.hip (its buffer stores address rows with 32-bit byte offsets)
static bool wide_rows(int c, int hidden, int dtype, int64_t m) {
  return wide_capable(c, hidden, dtype) && m >= wide_min_rows(c) &&
         m * 3 * c * (dtype == PTV3_F32 ? 4 : 2) < ((int64_t)1 << 31) - (1 << 20);
}

extern "C" int ptv3_block_fusable(int c, int hidden, int dtype, int64_t m) {
  if (hidden <= 0 || hidden % 64 != 0 || (dtype != PTV3_F32 && dtype != PTV3_BF16)) return 0;
  if (c == 32 || c == 64) return 1;
  if (wide_rows(c, hidden, dtype, m)) return 3;
  if ((c == 128 || c == 256 || c == 512) && hidden == 4 * c && m <= coop_max_rows(c) &&
      coop_tail_lds(c, hidden, dtype == PTV3_F32 ? 4 : 2) <= 160 * 1024)
    return 2;
  return 0;
}

template <typename T, int C>
static void launch_head_coop(const HeadArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((block_head_coop_kernel<T, C>), dim3((unsigned)cdiv(a.m, 16)), dim3(64 * Coop<C>::NW),
                     coop_head_lds(C, sizeof(T)), s, a);
}
template <typename T, int C>
static void launch_tail_coop(const TailArgs& a, hipStream_t s) {
  ensure_dynamic_lds(reinterpret_cast<const void*>(&block_tail_coop_kernel<T, C, 4>), 160 * 1024);
  hipLaunchKernelGGL((block_tail_coop_kernel<T, C, 4>), dim3((unsigned)cdiv(a.m, 16)), dim3(64 * Coop<C>::NW),
                     coop_tail_lds(C, 4 * C, sizeof(T)), s, a);
}
#define COOP_LAUNCH(FN, ARGS)                                              \
  if (dtype == PTV3_F32) {                                                 \
    if (c == 128) FN<float, 128>(ARGS, s);                                 \
    else if (c == 256) FN<float, 256>(ARGS, s);                            \
    else FN<float, 512>(ARGS, s);                                          \
  } else {                                                                 \
    if (c == 128) FN<__bf16, 128>(ARGS, s);                                \
    else if (c == 256) FN<__bf16, 256>(ARGS, s);                           \
    else FN<__bf16, 512>(ARGS, s);                                         \
  }

extern "C" int ptv3_mlp2_fusable(int cin, int hidden, int cout, int dtype) {
  return (cin == 32 || cin == 64) && hidden > 0 && hidden % 64 == 0 && cout > 0 && cout % 4 == 0 && cout <= 64 &&
         (dtype == PTV3_F32 || dtype == PTV3_BF16);
}

extern "C" int ptv3_mlp2(const void* x, const void* w1, const float* b1, const float* s1, const float* t1, int act,
                         const void* w2, const float* b2, void* out, int out_f32, int64_t m, int cin, int hidden,
                         int cout, int dtype, void* stream) {
  PTV3_REQUIRE(ptv3_mlp2_fusable(cin, hidden, cout, dtype), "mlp2: cin=%d hidden=%d cout=%d not fusable", cin, hidden,
               cout);
  PTV3_REQUIRE((s1 == nullptr) == (t1 == nullptr), "mlp2: s1/t1 must come together");
  if (m == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  Mlp2Args a;
  a.x = x; a.w1 = w1; a.b1 = b1; a.s1 = s1; a.t1 = t1; a.w2 = w2; a.b2 = b2; a.out = out;
  a.m = m; a.hidden = hidden; a.cout = cout; a.act = act; a.out_f32 = out_f32;
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_LINEAR, 2.0 * m * hidden * (cin + cout),
                              ((double)m * cin + (double)hidden * (cin + cout)) * esz + (double)m * cout * (out_f32 ? 4 : esz),
                              nullptr, 0, 0.0);
  prof_kernel(prof, PK_MLP2);
  const int rt = row_tiles(m);
  const int nto = cout <= 16 ? 1 : cout <= 32 ? 2 : 4;
  const size_t wb = ((size_t)hidden * (cin + 16 / esz) + (size_t)16 * nto * (hidden + 16 / esz)) * esz;
  const bool wlds = wb <= 128 * 1024;
  const int64_t nblocks = cdiv(m, 64 * rt);
  dim3 grid((unsigned)(wlds ? std::min<int64_t>(nblocks, 2 * 256) : nblocks)), block(256);
#define MLP2_GO(KERNEL_T, KERNEL_F)                                                               \
  if (wlds) {                                                                                     \
    ensure_dynamic_lds(reinterpret_cast<const void*>(&KERNEL_T), 128 * 1024);                     \
    hipLaunchKernelGGL(KERNEL_T, grid, block, wb, s, a);                                          \
  } else {                                                                                        \
    hipLaunchKernelGGL(KERNEL_F, grid, block, 0, s, a);                                           \
  }
#define MLP2_RT(T, NTI, NTO)                                                                      \
  if (rt == 2) { MLP2_GO((mlp2_kernel<T, NTI, NTO, 2, true>), (mlp2_kernel<T, NTI, NTO, 2, false>)) } \
  else { MLP2_GO((mlp2_kernel<T, NTI, NTO, 1, true>), (mlp2_kernel<T, NTI, NTO, 1, false>)) }
#define MLP2_CASE(T, NTI)                                                                         \
  if (nto == 1) { MLP2_RT(T, NTI, 1) }                                                            \
  else if (nto == 2) { MLP2_RT(T, NTI, 2) }                                                       \
  else { MLP2_RT(T, NTI, 4) }
  if (dtype == PTV3_F32) {
    if (cin == 32) { MLP2_CASE(float, 2) } else { MLP2_CASE(float, 4) }
  } else {
    if (cin == 32) { MLP2_CASE(__bf16, 2) } else { MLP2_CASE(__bf16, 4) }
  }
#undef MLP2_CASE
#undef MLP2_RT
#undef MLP2_GO
  prof_end(prof, s);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_block_head(const void* x, const float* slab, int splits, const float* conv_bias,
                               const void* shortcut, const float* g0, const float* b0, const float* g1,
                               const float* b1, const void* wqkv, const float* bqkv, void* f1, void* qkv, int64_t m,
                               int c, float eps, int dtype, void* stream) {
  // m = 0: capability only, the row limits of the cooperative variant are the caller's policy; the streaming
  // variant takes over from its own row threshold
  const int mode = wide_rows(c, 4 * c, dtype, m) && x != nullptr ? 3 : ptv3_block_fusable(c, 4 * c, dtype, 0);
  PTV3_REQUIRE(mode != 0, "block_head: c=%d not fusable", c);
  PTV3_REQUIRE((x != nullptr) != (slab != nullptr), "block_head: give the conv output OR its split-K slabs");
  PTV3_REQUIRE(slab == nullptr || (splits >= 1 && conv_bias != nullptr), "block_head: slabs need splits and the conv bias");
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "block_head: bad dtype");
  if (m == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  HeadArgs a{x, slab, splits, conv_bias, shortcut, g0, b0, g1, b1, wqkv, bqkv, f1, qkv, m, eps};
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_LINEAR, 2.0 * m * c * 3 * c, ((double)m * c * 6 + 3.0 * c * c) * esz, nullptr, 0, 0.0);
  prof_kernel(prof, mode == 1 ? PK_BLOCK_HEAD : mode == 3 ? PK_BLOCK_HEAD_WIDE : PK_BLOCK_HEAD_COOP);
  if (mode == 3) {
    launch_block_head_wide(a, c, dtype, s);
  } else if (mode == 1) {
    const int64_t nblocks = cdiv(m, 64);
    const size_t wb = (size_t)3 * c * (c + 16 / esz) * esz + (size_t)8 * c * sizeof(float);  // weights + vectors
    const dim3 grid_l((unsigned)std::min<int64_t>(nblocks, 4 * 256)), block(256);
    if (dtype == PTV3_F32) {
      if (c == 32) hipLaunchKernelGGL((block_head_kernel<float, 2, true>), grid_l, block, wb, s, a);
      else hipLaunchKernelGGL((block_head_kernel<float, 4, true>), grid_l, block, wb, s, a);
    } else {
      if (c == 32) hipLaunchKernelGGL((block_head_kernel<__bf16, 2, true>), grid_l, block, wb, s, a);
      else hipLaunchKernelGGL((block_head_kernel<__bf16, 4, true>), grid_l, block, wb, s, a);
    }
  } else {
    COOP_LAUNCH(launch_head_coop, a)
  }
  prof_end(prof, s);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_block_tail(const void* attn, const void* f1, const void* wproj, const float* bproj,
                               const float* g2, const float* b2, const void* w1, const float* bias1, const void* w2,
                               const float* bias2, void* out, int64_t m, int c, int hidden, float eps, int dtype,
                               void* stream) {
  const int mode = wide_rows(c, hidden, dtype, m) ? 3 : ptv3_block_fusable(c, hidden, dtype, 0);
  PTV3_REQUIRE(mode != 0, "block_tail: c=%d / hidden=%d not fusable", c, hidden);
  PTV3_REQUIRE(dtype == PTV3_F32 || dtype == PTV3_BF16, "block_tail: bad dtype");
  if (m == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  TailArgs a{attn, f1, wproj, bproj, g2, b2, w1, bias1, w2, bias2, out, m, hidden, eps};
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_LINEAR, 2.0 * m * c * (c + 2.0 * hidden),
                              ((double)m * c * 3 + (double)c * c + 2.0 * c * hidden) * esz, nullptr, 0, 0.0);
  prof_kernel(prof, mode == 1 ? PK_BLOCK_TAIL : mode == 3 ? PK_BLOCK_TAIL_WIDE : PK_BLOCK_TAIL_COOP);
  if (mode == 3) {
    launch_block_tail_wide(a, c, dtype, s);
  } else if (mode == 1) {
    TAIL_LAUNCH(a)
  } else {
    COOP_LAUNCH(launch_tail_coop, a)
  }
  prof_end(prof, s);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
