// Row-local halves of Block.forward (point_transformer_v3m1_base.py:318-338) for the WIDE levels at LARGE M
// (C = 128 / 256 with 10^4..10^5 rows: the deep levels of a LiDAR-like scan, where every level fills the chip):
//
//   head:  f1 = LayerNorm_cpe(x) + shortcut ; qkv = Linear_qkv(LayerNorm_1(f1))                  (:319-324, :188)
//   tail:  f2 = Linear_proj(attn) + f1 ; out = f2 + fc2(GELU(fc1(LayerNorm_2(f2))))               (:219, :326-334)
//
// Same arithmetic as block_fused.hip's wave-local chain (activations stay in registers from the first load to the
// last store: a wave owns RT x 16 rows, the accumulator of one GEMM is the B operand of the next), but the 12 C^2
// weights no longer fit LDS (393 KB at C = 128, 1.5 MB at C = 256).  They STREAM through it instead:
//   * a unit = 128 C bytes of weights = one 64 | 32-row slice (bf16 | fp32) of wproj / w1 / wqkv over all C inputs,
//     or all C rows of w2 over one 64 | 32-wide slice of the hidden layer;
//   * units are copied global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: no registers, no ds_write) into a
//     ring of three buffers, two units ahead of the one being multiplied; ONE workgroup barrier per unit, in front
//     of which every wave waits for its own copies of that unit with a COUNTED s_waitcnt vmcnt (the younger unit
//     stays in flight across the barrier);
//   * LDS image = 128-byte rows, 16-byte chunk c of row r stored at c ^ ((r >> 1) & 7) (the swizzle is applied to the
//     SOURCE address, the DMA destination is lane-linear): conflict-free ds_read_b128 of the A fragments;
//   * bf16: a GEMM whose result feeds the next one from registers gets its weight ROWS permuted on their way into
//     LDS (again a source-address matter), so that two accumulator tiles side by side are 8 CONSECUTIVE channels per
//     lane - the natural B fragment of the next GEMM, whose weights can then be read in natural K order too
//     (block_fused.hip permutes the K order of the next weight instead, on the host; here one weight set serves this
//     kernel, the cooperative one and the tiled GEMMs).
// The (M, 4C) hidden layer, LayerNorm outputs and the proj output never exist in memory: the tail reads attn + f1 and
// writes out (3 M C elements instead of 17 M C for proj / LayerNorm / fc1 / fc2 as four launches), the head reads
// x + shortcut and writes f1 + qkv (6 M C instead of 9 M C).  Outputs leave through a wave-private LDS tile so that
// every store instruction writes whole 128-byte row pieces.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "block_args.h"
#include "profile.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

template <typename T> struct Wide {
  static constexpr int PK = 128 / (int)sizeof(T);   // K elements in one 128-byte panel row (64 | 32)
  static constexpr int HSL = PK;                     // rows of a row-slice unit = K extent of a column-slice unit
  static constexpr int NH = HSL / 16;                // accumulator tiles per slice (4 | 2)
};

// Accumulator tile j of a chained GEMM: lane (i, g) register r holds channel ch(j, g) + r of point i.  fp32: the
// natural 16j + 4g.  bf16: tiles 2q and 2q+1 interleave in groups of four, so that the lane's eight values of the pair
// are channels 32q + 8g .. + 7; the A fragment of tile j therefore carries weight rows perm_row(16j + i) in lane i.
template <typename T> struct ChMap;
template <> struct ChMap<float> {
  static __device__ __forceinline__ int ch(int j, int g) { return 16 * j + 4 * g; }
  static __device__ __forceinline__ int perm_row(int l) { return l; }
};
template <> struct ChMap<__bf16> {
  static __device__ __forceinline__ int ch(int j, int g) { return 32 * (j >> 1) + 8 * g + 4 * (j & 1); }
  static __device__ __forceinline__ int perm_row(int l) { return (l & ~31) + 8 * ((l & 15) >> 2) + 4 * ((l >> 4) & 1) + (l & 3); }
};

#define PTV3_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define PTV3_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// `nrows` weight rows x 128 bytes (k0 .. k0 + PK of every row) -> one LDS panel.  Wave-instruction q copies rows
// 8q .. 8q+7: lane -> (row 8q + lane / 8, LDS chunk lane % 8), whose bytes come from source chunk c ^ swizzle(row).
// nrows / 8 is a multiple of 4: every wave issues exactly nrows / 32 copies.
// PERM: LDS row L takes weight row perm_row(L) (see ChMap)
template <typename T, int NROWS, bool PERM>
__device__ __forceinline__ void dma_panel(char* panel, const T* __restrict__ src, int64_t ld, int wave, int lane) {
  static_assert(NROWS % 32 == 0, "a panel is copied by four waves, eight rows per instruction");
  const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
  for (int i = 0; i < NROWS / 32; ++i) {
    const int q = wave + 4 * i;
    const int r = 8 * q + rr;
    const T* g = src + (int64_t)(PERM ? ChMap<T>::perm_row(r) : r) * ld + (cc ^ ((r >> 1) & 7)) * Frag<T>::E;
    __builtin_amdgcn_global_load_lds(PTV3_GLOBAL_PTR(g), PTV3_LDS_PTR(panel + q * 1024), 16, 0, 0);
  }
}

// n floats (a multiple of 64) global -> LDS by 4-byte LDS-DMA pieces, 64 floats per wave-instruction, dealt to the four
// waves: no register round trip and no wait in the prologue (the first unit's wait, which is younger, covers them)
__device__ __forceinline__ void dma_floats(float* sdst, const float* __restrict__ src, int n, int wave, int lane) {
  for (int q = wave; q < n / 64; q += 4)
    __builtin_amdgcn_global_load_lds(PTV3_GLOBAL_PTR(src + 64 * q + lane), PTV3_LDS_PTR(sdst + 64 * q), 4, 0, 0);
}

// A fragment (16 weight rows x one K chunk) for a B operand in NATURAL channel order
template <typename T>
__device__ __forceinline__ typename Frag<T>::type afrag_nat(const char* panel, int r, int ks, int g) {
  return *reinterpret_cast<const typename Frag<T>::type*>(panel + r * 128 + 16 * ((4 * ks + g) ^ ((r >> 1) & 7)));
}
template <typename T> struct FragF;   // a 16-byte fragment <-> floats
template <> struct FragF<float> {
  static __device__ __forceinline__ void unpack(f32x4 v, float* o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
  static __device__ __forceinline__ f32x4 pack(const float* o) { return f32x4{o[0], o[1], o[2], o[3]}; }
};
template <> struct FragF<__bf16> {
  static __device__ __forceinline__ void unpack(s16x8 v, float* o) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(v[i]);
  }
  static __device__ __forceinline__ s16x8 pack(const float* o) {
    const s16x4 lo = pack4<__bf16>(o[0], o[1], o[2], o[3]), hi = pack4<__bf16>(o[4], o[5], o[6], o[7]);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

constexpr int WIDE_STAGE_ROW = 128 + 16;   // bytes: one 128-byte output piece per row, +16 keeps the reads conflict-free

template <int RT> __host__ __device__ constexpr int wide_stage_bytes() { return RT * 16 * WIDE_STAGE_ROW; }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The ring holds NBUF units: one being multiplied, up to AH - 1 = NBUF - 2 younger ones in flight, and the buffer of
// the unit just finished, which the copies issued after the barrier overwrite.  Every wave waits for its OWN copies of
// unit u - all but the youngest memory operations of the wave: DPU copies for each younger unit that exists, plus,
// where every younger unit exists, EXTRA operations known to have been issued since (the head's output stores) - and
// then joins the barrier that (i) publishes unit u and (ii) says every wave is done reading unit u - 1.
template <int DPU, int AH, int EXTRA>
__device__ __forceinline__ void wide_sync(int u, int nu) {
  static_assert((AH - 1) * DPU + EXTRA <= 63, "vmcnt is a 6-bit field");
  const int rem = nu - 1 - u;   // younger units
  if (rem >= AH - 1) wait_vm<(AH - 1) * DPU + EXTRA>();
  else if (AH >= 4 && rem == 2) wait_vm<2 * DPU>();
  else if (AH >= 3 && rem == 1) wait_vm<DPU>();
  else if (rem >= 3) wait_vm<3 * DPU>();   // AH > 4: stricter than needed for the few last units
  else wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// GELU of the hidden layer.  fp32 (the parity mode): the erf form every other kernel uses (common.h).  bf16: at 20
// vector instructions per element - two of them quarter-rate (rcp, exp) - that form is the kernel's largest cost
// (one wave per SIMD carries 32 hidden values per lane and slice next to 128 matrix-core steps).  Here the result is
// rounded to bf16 (rms rounding error 1.7e-3 on N(0, 1.5) inputs), so erf(t), t = x / sqrt 2 clamped to +-2.75, is an
// odd polynomial of degree 13 fitted (minimax, constrained to reach exactly 1 at the clamp) to 3.1e-4 max / 8e-5 rms
// absolute error of the GELU value - multiply-adds only, evaluated two values per instruction (v_pk_fma_f32).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
  f32x2 t = x * 0.70710678f;
  t[0] = __builtin_amdgcn_fmed3f(t[0], -2.75f, 2.75f);
  t[1] = __builtin_amdgcn_fmed3f(t[1], -2.75f, 2.75f);
  const f32x2 u = t * t;
  f32x2 p = u * 5.844273399e-06f + -0.0001883886428f;
  p = p * u + 0.002602441612f;
  p = p * u + -0.02051126691f;
  p = p * u + 0.1043352928f;
  p = p * u + -0.3709193342f;
  p = p * u + 1.127441542f;
  const f32x2 h = x * 0.5f;
  return h * (t * p) + h;
}
template <typename T> __device__ __forceinline__ void gelu_bias4(f32x4& v, const f32x4& b);
template <> __device__ __forceinline__ void gelu_bias4<float>(f32x4& v, const f32x4& b) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r] + b[r]);
}
template <> __device__ __forceinline__ void gelu_bias4<__bf16>(f32x4& v, const f32x4& b) {
  const f32x2 lo = gelu_poly2(f32x2{v[0] + b[0], v[1] + b[1]}), hi = gelu_poly2(f32x2{v[2] + b[2], v[3] + b[3]});
  v = f32x4{lo[0], lo[1], hi[0], hi[1]};   // rounded to bf16 when the chain fragment is packed
}


// One unit's worth of matrix-core work with the A fragments fetched ONE GROUP AHEAD of their use: group n+1's four
// ds_read_b128 are issued in front of group n's 4 RT matrix-core steps (left to itself the compiler issued every
// fragment read right in front of its first use - a full LDS round trip exposed per pair of steps, with one wave per
// SIMD and nothing else to run).  sched_group_barrier pins that interleave.
//   ROWSLICE (wproj / w1 / wqkv units): fragment f = (K chunk f / NJ, tile f % NJ) -> acc[t][f % NJ] += A * x[t][f / NJ]
//   otherwise (w2 units: all NJ = C / 16 tiles over the slice's two K chunks): f = (tile f / 2, chunk f % 2)
template <typename T, int RT, int NJ, int NKCU, bool ROWSLICE, int XS, int AS>
__device__ __forceinline__ void unit_gemm(const char* buf, int slice_rows, const typename Frag<T>::type (&x)[RT][XS],
                                          f32x4 (&acc)[RT][AS], int acc0, int li, int g) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  constexpr int NF = NJ * NKCU, G = 4, NG = NF / G;
  static_assert(NF % G == 0, "fragments come in groups of four");
  auto frag = [&](int f) -> FR {
    if constexpr (ROWSLICE) {
      const int kc = f / NJ, jj = f % NJ;
      return afrag_nat<T>(buf + (kc >> 1) * slice_rows * 128, 16 * jj + li, kc & 1, g);
    } else {
      return afrag_nat<T>(buf, 16 * (f >> 1) + li, f & 1, g);
    }
  };
  FR wa[2][G];
#pragma unroll
  for (int e = 0; e < G; ++e) wa[0][e] = frag(e);
  __builtin_amdgcn_sched_group_barrier(0x100, G, 0);
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    if (n + 1 < NG) {
#pragma unroll
      for (int e = 0; e < G; ++e) wa[(n + 1) & 1][e] = frag((n + 1) * G + e);
    }
#pragma unroll
    for (int e = 0; e < G; ++e) {
      const int f = n * G + e;
      const int kc = ROWSLICE ? f / NJ : (f & 1), jt = ROWSLICE ? f % NJ : (f >> 1);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t][acc0 + jt] = F::mma(wa[n & 1][e], x[t][kc], acc[t][acc0 + jt]);
    }
    // the group's first matrix-core step (with the compiler's wait for the group's fragments in front of it - a wait
    // for ALL outstanding LDS reads, as hipcc writes it) goes ahead of the next group's reads
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (n + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, G, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, G * RT - 1, 0);
  }
}


// Output rows leave through BUFFER stores: a row at or past m lands beyond the descriptor's num_records and is dropped
// by the hardware, so every store instruction is issued by every wave, branch-free - the counted waits of the unit
// loop rely on that count.  (m x ld x sizeof(T) + one tile must stay below 4 GB: checked by the launcher.)
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(T* base, int64_t m, int64_t ld) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(m * ld * (int64_t)sizeof(T)), 0x00020000);
}
template <typename T>
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t rs, int64_t row, int64_t ld, int col,
                                        typename Frag<T>::type v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs,
                                         (unsigned)((row * ld + col) * (int64_t)sizeof(T)), 0, 0);
}

// RT x 16 rows x (NH x 16) channels from accumulator layout to memory as whole 128-byte row pieces (columns col0 ..),
// through the wave's own LDS tile (same-wave LDS operations execute in order: no barrier).  2 RT store instructions.
template <typename T, int RT, int NH, bool PERM>
__device__ __forceinline__ void wave_store_tile(char* stage, const f32x4 (&v)[RT][NH], __amdgpu_buffer_rsrc_t rs,
                                                int64_t ld, int col0, int64_t base, int li, int g, int lane) {
  typedef typename Vec4<T>::type V4;
  typedef typename Frag<T>::type FR;
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int jj = 0; jj < NH; ++jj)
      *reinterpret_cast<V4*>(stage + (16 * t + li) * WIDE_STAGE_ROW +
                             (PERM ? ChMap<T>::ch(jj, g) : 16 * jj + 4 * g) * (int)sizeof(T)) =
          pack4<T>(v[t][jj][0], v[t][jj][1], v[t][jj][2], v[t][jj][3]);
  const int rr = lane >> 3, cc = lane & 7;
#pragma unroll
  for (int it = 0; it < 2 * RT; ++it) {
    const int r = 8 * it + rr;
    const FR piece = *reinterpret_cast<const FR*>(stage + r * WIDE_STAGE_ROW + 16 * cc);
    store16<T>(rs, base + r, ld, col0 + cc * Frag<T>::E, piece);
  }
}

// ------------------------------------------------------------------------------------------------------------
template <typename T, int NT, int RT, int NBUF, int WGS = 1>
__global__ void __launch_bounds__(256, WGS) block_tail_wide_kernel(TailArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  typedef typename Vec4<T>::type V4;
  typedef Wide<T> W;
  constexpr int C = 16 * NT, KC = F::KC, NKC = C / KC, HSL = W::HSL, NH = W::NH;
  constexpr int NP = C / W::PK;          // panels of a row-slice unit
  constexpr int UB = 128 * C;            // bytes per unit
  constexpr int DPU = C / 32;            // copies per wave and unit
  constexpr int NPU = C / HSL;           // proj units
  constexpr int AH = NBUF - 1;           // units issued ahead of the one being multiplied
  extern __shared__ __attribute__((aligned(16))) char wide_smem[];
  char* ring = wide_smem;
  float* sVec = reinterpret_cast<float*>(wide_smem + NBUF * UB);   // bproj, g2, b2, bias2 [C each], bias1 [hidden]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  char* stage = reinterpret_cast<char*>(sVec + 4 * C + a.hidden) + wave * wide_stage_bytes<RT>();
  const T* attn = reinterpret_cast<const T*>(a.attn);
  const T* f1 = reinterpret_cast<const T*>(a.f1);
  const T* wp = reinterpret_cast<const T*>(a.wproj);
  const T* w1 = reinterpret_cast<const T*>(a.w1);
  const T* w2 = reinterpret_cast<const T*>(a.w2);
  const int NS = a.hidden / HSL;
  const int NU = NPU + 2 * NS;

  // unit u: u < NPU: rows u HSL.. of wproj; then, per hidden slice s, rows s HSL.. of w1 and columns s HSL.. of w2
  auto issue_rows = [&](int u, const T* src) {
    char* buf = ring + (u % NBUF) * UB;
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_panel<T, HSL, true>(buf + p * HSL * 128, src + p * W::PK, C, wave, lane);
  };
  auto issue_w2 = [&](int u, int s) {
    dma_panel<T, C, true>(ring + (u % NBUF) * UB, w2 + (int64_t)s * HSL, a.hidden, wave, lane);
  };
  auto issue_any = [&](int u) {
    if (u < NPU) issue_rows(u, wp + (int64_t)u * HSL * C);
    else if (((u - NPU) & 1) == 0) issue_rows(u, w1 + (int64_t)((u - NPU) >> 1) * HSL * C);
    else issue_w2(u, (u - NPU) >> 1);
  };
  dma_floats(sVec, a.bproj, C, wave, lane);
  dma_floats(sVec + C, a.g2, C, wave, lane);
  dma_floats(sVec + 2 * C, a.b2, C, wave, lane);
  dma_floats(sVec + 3 * C, a.bias2, C, wave, lane);
  dma_floats(sVec + 4 * C, a.bias1, a.hidden, wave, lane);
#pragma unroll
  for (int i = 0; i < AH; ++i) issue_any(i);
  const float *vbp = sVec, *vg2 = sVec + C, *vb2 = sVec + 2 * C, *vbias2 = sVec + 3 * C, *vbias1 = sVec + 4 * C;

  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);
  int64_t rc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int64_t row = base + 16 * t + li;
    rc[t] = row < a.m ? row : a.m - 1;
  }
  // ---- f2 = attn @ Wproj^T + b + f1   (attn rows in natural channel order)
  FR xa[RT][NKC];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) xa[t][kc] = *reinterpret_cast<const FR*>(attn + rc[t] * C + KC * kc + F::E * g);
  f32x4 f2[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) f2[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int u = 0;
#pragma unroll
  for (int pt = 0; pt < NPU; ++pt, ++u) {
    wide_sync<DPU, AH, 0>(u, NU);
    if (u + AH < NU) issue_any(u + AH);
    unit_gemm<T, RT, NH, NKC, true>(ring + (u % NBUF) * UB, HSL, xa, f2, pt * NH, li, g);
  }
  // ---- + bias + f1, LayerNorm_2 -> B fragments of fc1; f2 stays (rounded to T, packed) as the MLP's residual
  FR xf[RT][NKC];
  V4 f2p[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = ChMap<T>::ch(j, g);
      const f32x4 b = *reinterpret_cast<const f32x4*>(vbp + ch);
      float s[4];
      unpack4<T>(*reinterpret_cast<const V4*>(f1 + rc[t] * C + ch), s);
#pragma unroll
      for (int r = 0; r < 4; ++r) f2[t][j][r] = round_to<T>(f2[t][j][r] + b[r] + s[r]);
      f2p[t][j] = pack4<T>(f2[t][j][0], f2[t][j][1], f2[t][j][2], f2[t][j][3]);
    }
    float mean, rstd;
    row_norm<NT>(f2[t], a.eps, mean, rstd);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ch = ChMap<T>::ch(j, g);
      const f32x4 gm = *reinterpret_cast<const f32x4*>(vg2 + ch), bt = *reinterpret_cast<const f32x4*>(vb2 + ch);
#pragma unroll
      for (int r = 0; r < 4; ++r) f2[t][j][r] = round_to<T>((f2[t][j][r] - mean) * rstd * gm[r] + bt[r]);
    }
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) xf[t][kc] = ChainFrag<T, NT>::get(f2[t], kc);
  }
  // ---- out = f2 + fc2(GELU(fc1(t5))): the hidden layer passes through registers one slice at a time
  f32x4 o[RT][NT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) o[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ab = a.ablate;
  for (int s = 0; s < NS; ++s) {
    if (!(ab & 8)) wide_sync<DPU, AH, 0>(u, NU);
    if (u + AH < NU && !(ab & 4)) issue_any(u + AH);
    const char* buf = ring + (u % NBUF) * UB;
    ++u;
    f32x4 h[RT][NH];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int jj = 0; jj < NH; ++jj) h[t][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(ab & 2)) unit_gemm<T, RT, NH, NKC, true>(buf, HSL, xf, h, 0, li, g);
#pragma unroll
    for (int jj = 0; jj < NH; ++jj) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(vbias1 + s * HSL + ChMap<T>::ch(jj, g));
      if (ab & 1) {
#pragma unroll
        for (int t = 0; t < RT; ++t) h[t][jj] += b;
      } else {
#pragma unroll
        for (int t = 0; t < RT; ++t) gelu_bias4<T>(h[t][jj], b);
      }
    }
    FR hf[RT][2];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int mm = 0; mm < 2; ++mm) hf[t][mm] = ChainFrag<T, NH>::get(h[t], mm);
    if (!(ab & 8)) wide_sync<DPU, AH, 0>(u, NU);
    if (u + AH < NU && !(ab & 4)) issue_any(u + AH);
    buf = ring + (u % NBUF) * UB;
    ++u;
    if (!(ab & 2)) unit_gemm<T, RT, NT, 2, false>(buf, C, hf, o, 0, li, g);
  }
  // ---- + bias2 + f2, out in pieces of HSL channels
  const __amdgpu_buffer_rsrc_t ors = out_rsrc<T>(reinterpret_cast<T*>(a.out), a.m, C);
#pragma unroll
  for (int jg = 0; jg < NT / NH; ++jg) {
    f32x4 v[RT][NH];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int jj = 0; jj < NH; ++jj) {
        const int j = jg * NH + jj;
        const f32x4 b = *reinterpret_cast<const f32x4*>(vbias2 + ChMap<T>::ch(j, g));
        float r2[4];
        unpack4<T>(f2p[t][j], r2);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][jj][r] = o[t][j][r] + b[r] + r2[r];
      }
    wave_store_tile<T, RT, NH, true>(stage, v, ors, C, jg * HSL, base, li, g, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------
template <typename T, int NT, int RT, int NBUF, int WGS = 1>
__global__ void __launch_bounds__(256, WGS) block_head_wide_kernel(HeadArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  typedef Wide<T> W;
  constexpr int C = 16 * NT, E = F::E, KC = F::KC, NKC = C / KC, HSL = W::HSL, NH = W::NH;
  constexpr int NP = C / W::PK;
  constexpr int UB = 128 * C;
  constexpr int DPU = C / 32;
  constexpr int NU = 3 * C / HSL;
  constexpr int AH = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) char wide_smem[];
  char* ring = wide_smem;
  float* sVec = reinterpret_cast<float*>(wide_smem + NBUF * UB);   // g0, b0, g1, b1 [C each], bqkv [3C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  char* stage = reinterpret_cast<char*>(sVec + 7 * C) + wave * wide_stage_bytes<RT>();
  const T* w = reinterpret_cast<const T*>(a.wqkv);

  auto issue = [&](int u) {
    char* buf = ring + (u % NBUF) * UB;
    const T* src = w + (int64_t)u * HSL * C;
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_panel<T, HSL, false>(buf + p * HSL * 128, src + p * W::PK, C, wave, lane);
  };
  dma_floats(sVec, a.g0, C, wave, lane);
  dma_floats(sVec + C, a.b0, C, wave, lane);
  dma_floats(sVec + 2 * C, a.g1, C, wave, lane);
  dma_floats(sVec + 3 * C, a.b1, C, wave, lane);
  dma_floats(sVec + 4 * C, a.bqkv, 3 * C, wave, lane);
#pragma unroll
  for (int i = 0; i < AH; ++i) issue(i);
  const float *vg0 = sVec, *vb0 = sVec + C, *vg1 = sVec + 2 * C, *vb1 = sVec + 3 * C, *vbq = sVec + 4 * C;

  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* sc = reinterpret_cast<const T*>(a.shortcut);
  const __amdgpu_buffer_rsrc_t f1rs = out_rsrc<T>(reinterpret_cast<T*>(a.f1), a.m, C);
  const __amdgpu_buffer_rsrc_t qrs = out_rsrc<T>(reinterpret_cast<T*>(a.qkv), a.m, 3 * C);
  // rows in NATURAL channel order: lane (i, g) holds channels KC kc + E g .. + E - 1 of row i (16-byte loads / stores)
  FR xr[RT][NKC], sr[RT][NKC];
  int64_t rc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int64_t row = base + 16 * t + li;
    rc[t] = row < a.m ? row : a.m - 1;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      xr[t][kc] = *reinterpret_cast<const FR*>(x + rc[t] * C + KC * kc + E * g);
      sr[t][kc] = *reinterpret_cast<const FR*>(sc + rc[t] * C + KC * kc + E * g);
    }
  }
  // the vectors parked above are read below: make them visible (the first unit is synchronised here too)
  wide_sync<DPU, AH, 0>(0, NU);
  issue(AH);
  FR xf[RT][NKC];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    float v[NKC][E];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) FragF<T>::unpack(xr[t][kc], v[kc]);
    auto stats = [&](float& mean, float& rstd) {
      float s = 0.f;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int e = 0; e < E; ++e) s += v[kc][e];
      mean = groups_sum(s) * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int e = 0; e < E; ++e) { const float d = v[kc][e] - mean; q += d * d; }
      rstd = rsqrtf(groups_sum(q) * (1.0f / C) + a.eps);
    };
    float mean, rstd;
    stats(mean, rstd);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      float s[E];
      FragF<T>::unpack(sr[t][kc], s);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int ch = KC * kc + E * g + e;
        v[kc][e] = round_to<T>((v[kc][e] - mean) * rstd * vg0[ch] + vb0[ch] + s[e]);
      }
      store16<T>(f1rs, base + 16 * t + li, C, KC * kc + E * g, FragF<T>::pack(v[kc]));
    }
    stats(mean, rstd);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int ch = KC * kc + E * g + e;
        v[kc][e] = (v[kc][e] - mean) * rstd * vg1[ch] + vb1[ch];
      }
      xf[t][kc] = FragF<T>::pack(v[kc]);
    }
  }
  // ---- qkv = t3 @ Wqkv^T + b, one HSL-channel slice per unit
  // younger than the copies of unit u when its wait comes: the copies of the AH - 1 following units and the output
  // stores of the two units before (2 RT each, always issued; for the first units the RT NKC stores of f1 instead)
  for (int u = 0; u < NU; ++u) {
    if (u > 0) {
      wide_sync<DPU, AH, 4 * RT>(u, NU);
      if (u + AH < NU) issue(u + AH);
    }
    const char* buf = ring + (u % NBUF) * UB;
    f32x4 acc[RT][NH];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int jj = 0; jj < NH; ++jj) acc[t][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    unit_gemm<T, RT, NH, NKC, true>(buf, HSL, xf, acc, 0, li, g);
#pragma unroll
    for (int jj = 0; jj < NH; ++jj) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(vbq + u * HSL + 16 * jj + 4 * g);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t][jj] += b;
    }
    wave_store_tile<T, RT, NH, false>(stage, acc, qrs, 3 * C, u * HSL, base, li, g, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------
// One linear layer on whole rows with its neighbours folded in (the same weight-streaming machinery): for widths the
// fused halves above do not cover (C = 512: 12 C^2 weights per 128 rows would be 6.3 MB of stream) and for the
// linears outside a block.
//   out = act(prologue(x) @ w^T + bias) [+ res]
//   prologue  PRO = 0: x as it is;  1: LayerNorm(x; g1, b1);
//             2: f1 = LayerNorm(x; g0, b0) + shortcut (stored), then LayerNorm(f1; g1, b1)   (Block.forward :319-324)
// A workgroup owns 64 RT rows and the output channels [y, y + 1) x cout / gridDim.y (column groups only for small m);
// the rows are read ONCE (the tiled GEMM re-reads them per 64-channel column tile: 4x the algorithmic traffic on the
// C = 512 level of a LiDAR scan), LayerNorm runs on the packed fragments (statistics by a second unpack, nothing
// but the fragments stays live), units of HSLR weight rows x C stream through the LDS ring.
// ------------------------------------------------------------------------------------------------------------
// a 16-byte fragment the compiler may not see through (no instruction: keeps it from caching values derived from it)
template <typename FR> __device__ __forceinline__ FR opaque16(FR f) {
  u32x4 u = __builtin_bit_cast(u32x4, f);
  asm volatile("" : "+v"(u));
  return __builtin_bit_cast(FR, u);
}

struct RowsLinArgs {
  const void* x; const void* shortcut;
  const float *g0, *b0, *g1, *b1;
  const void* w; const float* bias; const void* res;
  void* f1; void* out;
  int64_t m; int cout; int act; float eps;
};

template <typename T, int NT, int RT, int HSLR, int PRO>
__global__ void __launch_bounds__(256) rows_linear_kernel(RowsLinArgs a) {
  typedef Frag<T> F;
  typedef typename F::type FR;
  typedef Wide<T> W;
  constexpr int C = 16 * NT, E = F::E, KC = F::KC, NKC = C / KC, NH = HSLR / 16;
  constexpr int NP = C / W::PK;
  constexpr int UB = HSLR * C * (int)sizeof(T);
  constexpr int DPU = UB / 4096;
  constexpr int NBUF = 3, AH = NBUF - 1;
  static_assert(UB <= 32 * 1024 && UB % 4096 == 0, "unit size");
  extern __shared__ __attribute__((aligned(16))) char wide_smem[];
  char* ring = wide_smem;
  float* sVec = reinterpret_cast<float*>(wide_smem + NBUF * UB);   // g0, b0, g1, b1 [C each], bias [cout]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, g = lane >> 4;
  char* stage = reinterpret_cast<char*>(sVec + 4 * C + a.cout) + wave * wide_stage_bytes<RT>();
  const T* w = reinterpret_cast<const T*>(a.w);
  const int units = a.cout / HSLR;
  const int upw = (units + gridDim.y - 1) / gridDim.y;
  const int u0 = blockIdx.y * upw;
  const int nu = min(units, u0 + upw) - u0;        // units of this workgroup (>= 1 by the launcher's split)

  auto issue = [&](int u) {
    char* buf = ring + (u % NBUF) * UB;
    const T* src = w + (int64_t)(u0 + u) * HSLR * C;
#pragma unroll
    for (int p = 0; p < NP; ++p) dma_panel<T, HSLR, false>(buf + p * HSLR * 128, src + p * W::PK, C, wave, lane);
  };
  if constexpr (PRO == 2) { dma_floats(sVec, a.g0, C, wave, lane); dma_floats(sVec + C, a.b0, C, wave, lane); }
  if constexpr (PRO >= 1) { dma_floats(sVec + 2 * C, a.g1, C, wave, lane); dma_floats(sVec + 3 * C, a.b1, C, wave, lane); }
  dma_floats(sVec + 4 * C, a.bias, a.cout, wave, lane);
#pragma unroll
  for (int i = 0; i < AH; ++i)
    if (i < nu) issue(i);
  const float *vg0 = sVec, *vb0 = sVec + C, *vg1 = sVec + 2 * C, *vb1 = sVec + 3 * C, *vbias = sVec + 4 * C;

  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * (16 * RT);
  const T* x = reinterpret_cast<const T*>(a.x);
  const __amdgpu_buffer_rsrc_t ors = out_rsrc<T>(reinterpret_cast<T*>(a.out), a.m, a.cout);
  // rows in NATURAL channel order: lane (i, g) holds channels KC kc + E g .. + E - 1 of row i
  FR xf[RT][NKC];
  int64_t rc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int64_t row = base + 16 * t + li;
    rc[t] = row < a.m ? row : a.m - 1;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) xf[t][kc] = *reinterpret_cast<const FR*>(x + rc[t] * C + KC * kc + E * g);
  }
  // vectors + first unit visible.  (nu < AH leaves fewer copies in flight than the counted wait assumes: wait for all.)
  if (nu >= AH) wide_sync<DPU, AH, 0>(0, nu); else wide_sync<DPU, 1, 0>(0, 1);
  if (AH < nu) issue(AH);
  if constexpr (PRO >= 1) {
    // LayerNorm on the packed fragments: mean, then centred second moment (two unpacks), then normalise in place
    auto stats = [&](const FR (&fr)[NKC], float& mean, float& rstd) {
      float s = 0.f;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        float v[E];
        FragF<T>::unpack(fr[kc], v);
#pragma unroll
        for (int e = 0; e < E; ++e) s += v[e];
      }
      mean = groups_sum(s) * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        float v[E];
        // (the fragment is made opaque first: otherwise the compiler keeps the floats of the first pass - C / 4 live
        // registers per row tile, spilled at C = 512 - instead of unpacking again)
        FragF<T>::unpack(opaque16(fr[kc]), v);
#pragma unroll
        for (int e = 0; e < E; ++e) { const float d = v[e] - mean; q += d * d; }
      }
      rstd = rsqrtf(groups_sum(q) * (1.0f / C) + a.eps);
    };
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      float mean, rstd;
      if constexpr (PRO == 2) {
        const T* sc = reinterpret_cast<const T*>(a.shortcut);
        const __amdgpu_buffer_rsrc_t f1rs = out_rsrc<T>(reinterpret_cast<T*>(a.f1), a.m, C);
        FR sr[NKC];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) sr[kc] = *reinterpret_cast<const FR*>(sc + rc[t] * C + KC * kc + E * g);
        stats(xf[t], mean, rstd);
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          float v[E], sh[E];
          FragF<T>::unpack(opaque16(xf[t][kc]), v);
          FragF<T>::unpack(sr[kc], sh);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int ch = KC * kc + E * g + e;
            v[e] = (v[e] - mean) * rstd * vg0[ch] + vb0[ch] + sh[e];
          }
          xf[t][kc] = FragF<T>::pack(v);      // rounded to T: f1 as stored, and what LayerNorm_1 sees
          if (blockIdx.y == 0) store16<T>(f1rs, base + 16 * t + li, C, KC * kc + E * g, xf[t][kc]);
          // one K chunk at a time: hoisting every chunk's vector reads ahead costs 2 C / 4 registers per row tile
          if constexpr (C >= 512) __builtin_amdgcn_sched_barrier(0);
        }
      }
      stats(xf[t], mean, rstd);
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc) {
        float v[E];
        FragF<T>::unpack(opaque16(xf[t][kc]), v);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int ch = KC * kc + E * g + e;
          v[e] = (v[e] - mean) * rstd * vg1[ch] + vb1[ch];
        }
        xf[t][kc] = FragF<T>::pack(v);
        if constexpr (C >= 512) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const T* res = reinterpret_cast<const T*>(a.res);
  for (int u = 0; u < nu; ++u) {
    if (u > 0) {
      // stricter than needed while output (and f1) stores are in flight: their count varies with PRO and the split
      wide_sync<DPU, AH, 0>(u, nu);
      if (u + AH < nu) issue(u + AH);
    }
    const char* buf = ring + (u % NBUF) * UB;
    f32x4 acc[RT][NH];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int jj = 0; jj < NH; ++jj) acc[t][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    unit_gemm<T, RT, NH, NKC, true>(buf, HSLR, xf, acc, 0, li, g);
    const int col0 = (u0 + u) * HSLR;
#pragma unroll
    for (int jj = 0; jj < NH; ++jj) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(vbias + col0 + 16 * jj + 4 * g);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        if (a.act == PTV3_ACT_GELU) gelu_bias4<T>(acc[t][jj], b);
        else if (a.act == PTV3_ACT_RELU) { acc[t][jj] += b; for (int r = 0; r < 4; ++r) acc[t][jj][r] = fmaxf(acc[t][jj][r], 0.f); }
        else acc[t][jj] += b;
      }
    }
    // staged tile -> whole 128-byte row pieces (HSLR = 32 bf16 channels: 64-byte pieces), residual added on the way out
    {
      typedef typename Vec4<T>::type V4;
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int jj = 0; jj < NH; ++jj)
          *reinterpret_cast<V4*>(stage + (16 * t + li) * WIDE_STAGE_ROW + (16 * jj + 4 * g) * (int)sizeof(T)) =
              pack4<T>(acc[t][jj][0], acc[t][jj][1], acc[t][jj][2], acc[t][jj][3]);
      constexpr int CPR = HSLR * (int)sizeof(T) / 16;      // 16-byte pieces per row of the tile (8 | 4)
      constexpr int RPI = 64 / CPR;                         // rows per store instruction
      const int rr = lane / CPR, cc = lane % CPR;
#pragma unroll
      for (int it = 0; it < 16 * RT / RPI; ++it) {
        const int r = RPI * it + rr;
        FR piece = *reinterpret_cast<const FR*>(stage + r * WIDE_STAGE_ROW + 16 * cc);
        const int64_t row = base + r;
        if (res) {
          const int64_t rrow = row < a.m ? row : a.m - 1;
          const FR rp = *reinterpret_cast<const FR*>(res + rrow * a.cout + col0 + cc * E);
          float v[E], rv[E];
          FragF<T>::unpack(piece, v);
          FragF<T>::unpack(rp, rv);
#pragma unroll
          for (int e = 0; e < E; ++e) v[e] += rv[e];
          piece = FragF<T>::pack(v);
        }
        store16<T>(ors, row, a.cout, col0 + cc * E, piece);
      }
    }
  }
}

template <typename T, int RT, int NBUF> static size_t wide_tail_lds(int c, int hidden) {
  return (size_t)NBUF * 128 * c + (size_t)(4 * c + hidden) * sizeof(float) + 4 * wide_stage_bytes<RT>();
}
template <typename T, int RT, int NBUF> static size_t wide_head_lds(int c) {
  return (size_t)NBUF * 128 * c + (size_t)7 * c * sizeof(float) + 4 * wide_stage_bytes<RT>();
}

template <typename T, int NT, int RT, int NBUF, int WGS = 1>
static void launch_tail_wide(const TailArgs& a, hipStream_t s) {
  const size_t lds = wide_tail_lds<T, RT, NBUF>(16 * NT, a.hidden);
  ensure_dynamic_lds(reinterpret_cast<const void*>(&block_tail_wide_kernel<T, NT, RT, NBUF, WGS>), 160 * 1024);
  hipLaunchKernelGGL((block_tail_wide_kernel<T, NT, RT, NBUF, WGS>), dim3((unsigned)cdiv(a.m, 64 * RT)), dim3(256), lds, s, a);
}
template <typename T, int NT, int RT, int NBUF, int WGS = 1>
static void launch_head_wide(const HeadArgs& a, hipStream_t s) {
  const size_t lds = wide_head_lds<T, RT, NBUF>(16 * NT);
  ensure_dynamic_lds(reinterpret_cast<const void*>(&block_head_wide_kernel<T, NT, RT, NBUF, WGS>), 160 * 1024);
  hipLaunchKernelGGL((block_head_wide_kernel<T, NT, RT, NBUF, WGS>), dim3((unsigned)cdiv(a.m, 64 * RT)), dim3(256), lds, s, a);
}

// rows from which the streaming variant takes over (below, its grid of m / 128 workgroups leaves CUs idle and the
// tiled GEMM launches spread the same weights over more of them); env PTV3_WIDE_ROWS_<c>, 0 = off
int64_t wide_min_rows(int c) {
  static int64_t lim[2] = {-1, -1};
  const int i = c == 128 ? 0 : 1;
  if (lim[i] < 0) {
    char name[32];
    snprintf(name, sizeof name, "PTV3_WIDE_ROWS_%d", c);
    const char* e = getenv(name);
    lim[i] = e ? atoll(e) : 24576;
    if (lim[i] <= 0) lim[i] = INT64_MAX;
  }
  return lim[i];
}

bool wide_capable(int c, int hidden, int dtype) {
  return (c == 128 || c == 256) && hidden == 4 * c && (dtype == PTV3_F32 || dtype == PTV3_BF16);
}

// Ring depth 3 (two units in flight).  Measured equal to the deeper rings PTV3_WIDE_NBUF=4 selects (4 x 32 KB at C = 256,
// 6 x 16 KB at C = 128): the copies are not what the kernel waits for.  C = 128 bf16 runs two workgroups per CU
// (71 KB of LDS, 256 registers: PTV3_WIDE_WGS=1 for one).
static bool deep_ring() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PTV3_WIDE_NBUF"); v = (e && atoi(e) > 3) ? 1 : 0; }
  return v == 1;
}
static bool two_wgs() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("PTV3_WIDE_WGS"); v = (e && atoi(e) == 1) ? 0 : 1; }
  return v == 1;
}
void launch_block_head_wide(const HeadArgs& a, int c, int dtype, hipStream_t s) {
  const bool deep = deep_ring();
  if (dtype == PTV3_F32) {
    if (c == 128) { if (deep) launch_head_wide<float, 8, 1, 6>(a, s); else launch_head_wide<float, 8, 1, 3>(a, s); }
    else { if (deep) launch_head_wide<float, 16, 1, 4>(a, s); else launch_head_wide<float, 16, 1, 3>(a, s); }
  } else {
    if (c == 128) {
      if (deep) launch_head_wide<__bf16, 8, 2, 6>(a, s);
      else if (two_wgs()) launch_head_wide<__bf16, 8, 2, 3, 2>(a, s);
      else launch_head_wide<__bf16, 8, 2, 3>(a, s);
    }
    else { if (deep) launch_head_wide<__bf16, 16, 2, 4>(a, s); else launch_head_wide<__bf16, 16, 2, 3>(a, s); }
  }
}
void launch_block_tail_wide(const TailArgs& a0, int c, int dtype, hipStream_t s) {
  const bool deep = deep_ring();
  TailArgs a = a0;
  if (const char* e = getenv("PTV3_WIDE_ABLATE")) a.ablate = atoi(e);
  if (dtype == PTV3_F32) {
    if (c == 128) { if (deep) launch_tail_wide<float, 8, 1, 6>(a, s); else launch_tail_wide<float, 8, 1, 3>(a, s); }
    else { if (deep) launch_tail_wide<float, 16, 1, 4>(a, s); else launch_tail_wide<float, 16, 1, 3>(a, s); }
  } else {
    if (c == 128) {
      if (deep) launch_tail_wide<__bf16, 8, 2, 6>(a, s);
      else if (two_wgs()) launch_tail_wide<__bf16, 8, 2, 3, 2>(a, s);
      else launch_tail_wide<__bf16, 8, 2, 3>(a, s);
    }
    else { if (deep) launch_tail_wide<__bf16, 16, 2, 4>(a, s); else launch_tail_wide<__bf16, 16, 2, 3>(a, s); }
  }
}

// ---- ptv3_rows_linear: capability and launch
static int rows_hslr(int c, int dtype) {
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  int h = 64;
  while (h > 16 && h * c * esz > 32 * 1024) h >>= 1;
  return h;
}
bool rows_linear_capable(int c, int cout, int dtype, int64_t m) {
  if (!(c == 128 || c == 256 || c == 512) || (dtype != PTV3_F32 && dtype != PTV3_BF16)) return false;
  const int h = rows_hslr(c, dtype);
  if (h < 32 || cout % h != 0 || cout > 8192) return false;     // fp32 at c = 512: a 32-row unit would be 64 KB
  return m * (int64_t)std::max(cout, c) * (dtype == PTV3_F32 ? 4 : 2) < ((int64_t)1 << 31) - (1 << 20);
}

template <typename T, int NT, int RT, int HSLR, int PRO>
static void launch_rows_pro(const RowsLinArgs& a, hipStream_t s) {
  const int c = 16 * NT;
  const size_t lds = (size_t)3 * HSLR * c * sizeof(T) + (size_t)(4 * c + a.cout) * sizeof(float) + 4 * wide_stage_bytes<RT>();
  const int64_t tiles = cdiv(a.m, 64 * RT);
  const int units = a.cout / HSLR;
  // column groups only when the row tiles alone leave CUs idle (every group re-reads and re-normalises the rows)
  const int split = tiles >= 192 ? 1 : (int)std::min<int64_t>(units, cdiv(256, tiles));
  ensure_dynamic_lds(reinterpret_cast<const void*>(&rows_linear_kernel<T, NT, RT, HSLR, PRO>), 160 * 1024);
  hipLaunchKernelGGL((rows_linear_kernel<T, NT, RT, HSLR, PRO>), dim3((unsigned)tiles, (unsigned)split), dim3(256), lds, s, a);
}

// RT: two 16-row tiles per wave halve the weight stream per row; at C = 512 the LayerNorm prologues of two tiles do not
// fit the register file next to the 128 fragment registers (the compiler spills ~1 KB per lane): one tile there
template <typename T, int NT, int RT, int HSLR>
static void launch_rows(const RowsLinArgs& a, int pro, hipStream_t s) {
  constexpr int RTP = (NT >= 32) ? 1 : RT;
  if (pro == 2) launch_rows_pro<T, NT, RTP, HSLR, 2>(a, s);
  else if (pro == 1) launch_rows_pro<T, NT, RTP, HSLR, 1>(a, s);
  else launch_rows_pro<T, NT, RT, HSLR, 0>(a, s);
}

void launch_rows_linear(const RowsLinArgs& a, int c, int pro, int dtype, hipStream_t s) {
  if (dtype == PTV3_F32) {
    if (c == 128) launch_rows<float, 8, 1, 32>(a, pro, s); else launch_rows<float, 16, 1, 32>(a, pro, s);
  } else {
    if (c == 128) launch_rows<__bf16, 8, 2, 64>(a, pro, s);
    else if (c == 256) launch_rows<__bf16, 16, 2, 64>(a, pro, s);
    else launch_rows<__bf16, 32, 2, 32>(a, pro, s);
  }
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_rows_linear_capable(int c, int cout, int dtype, int64_t m) {
  return rows_linear_capable(c, cout, dtype, m) ? 1 : 0;
}

extern "C" int ptv3_rows_linear(const void* x, const void* shortcut, const float* g0, const float* b0, const float* g1,
                                const float* b1, const void* w, const float* bias, int act, const void* res, void* f1,
                                void* out, int64_t m, int c, int cout, float eps, int dtype, void* stream) {
  PTV3_REQUIRE(rows_linear_capable(c, cout, dtype, m), "rows_linear: c=%d cout=%d dtype=%d m=%lld not served", c, cout,
               dtype, (long long)m);
  PTV3_REQUIRE(x && w && bias && out, "rows_linear: x, w, bias and out are required");
  PTV3_REQUIRE((g1 == nullptr) == (b1 == nullptr) && (g0 == nullptr) == (b0 == nullptr), "rows_linear: LayerNorm weight and bias come together");
  const int pro = g0 ? 2 : (g1 ? 1 : 0);
  PTV3_REQUIRE(pro != 2 || (g1 && shortcut && f1), "rows_linear: the chained prologue needs g1/b1, shortcut and f1");
  if (m == 0) return PTV3_OK;
  hipStream_t s = (hipStream_t)stream;
  const RowsLinArgs a{x, shortcut, g0, b0, g1, b1, w, bias, res, f1, out, m, cout, act, eps};
  const int esz = dtype == PTV3_F32 ? 4 : 2;
  const int prof = prof_begin(s, PROF_LINEAR, 2.0 * m * c * (double)cout,
                              ((double)m * c * (1 + 2 * (pro == 2)) + (double)c * cout + (double)m * cout * (1 + (res != nullptr))) * esz,
                              nullptr, 0, 0.0);
  prof_kernel(prof, PK_ROWS_LINEAR);
  launch_rows_linear(a, c, pro, dtype, s);
  prof_end(prof, s);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
