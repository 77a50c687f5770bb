// Fused multi-tensor AdamW step (torch.optim.AdamW semantics, decoupled weight decay) for the training glue
// (the reference builds torch.optim.AdamW through pointcept/utils/optimizer.py with one extra
// parameter group for the "block" keyword).  One launch updates every parameter: a device table lists
// (param, grad, exp_avg, exp_avg_sq, numel, group) per tensor, blocks find their tensor by binary search.
#include "common.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

struct AdamWTensor {
  float* p; const float* g; float* m; float* v;
  int64_t numel; int32_t group; int32_t first_block;
  // optional compute-dtype shadows the training Functions read instead of casting / transposing every step:
  // shadow = the parameter as is; shadow_t = (rows, kvol, cols) -> (cols, kvol mirrored, rows), i.e. W^T for a
  // Linear (kvol = 1) and the mirrored-tap transposed weight of a SubMConv3d input gradient
  void* shadow; void* shadow_t;
  int32_t rows, cols, kvol, shadow_dtype;
  // steps this tensor has taken fewer than the launch's `step` argument (a parameter that joined the table late, or
  // was restored from a checkpoint with its own torch.optim state["step"]): bias corrections use step - step_lag
  int32_t step_lag;
  int32_t first_tile;   // first 64 x 64 tile of this tensor in the transposed-shadow pass (tensors without shadow_t own none)
};

struct AdamWGroups { float lr[8], wd[8]; };
// gradient addresses of up to 256 consecutive table entries, handed over as a kernel argument: autograd leaves a NEW
// gradient tensor on every parameter each step (after zero_grad(set_to_none=True)), and a by-value argument needs neither
// a table rebuild nor a host-to-device copy that the next step's host code could race with
constexpr int ADAMW_PTRS = 256;
struct AdamWGrads { const float* g[ADAMW_PTRS]; };

constexpr int ADAMW_CHUNK = 4096;  // elements per block

template <bool PTRS>
__global__ void __launch_bounds__(256) adamw_kernel(const AdamWTensor* __restrict__ tab, int tensor_lo, int tensor_hi,
                                                     int block_base, AdamWGrads gp, AdamWGroups grp, float beta1,
                                                     float beta2, float eps, float bc1, float rsqrt_bc2,
                                                     float grad_scale, float step_no) {
  // find the tensor owning this block (entries tensor_lo .. tensor_hi of the table)
  const int blk = block_base + (int)blockIdx.x;
  int lo = tensor_lo, hi = tensor_hi;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  AdamWTensor t = tab[lo];
  if (PTRS) t.g = gp.g[lo - tensor_lo];
  const int64_t base = (int64_t)(blk - t.first_block) * ADAMW_CHUNK;
  const float lr = grp.lr[t.group], wd = grp.wd[t.group];
  if (t.step_lag != 0) {  // this tensor's own step count (block-uniform branch)
    const float own = step_no - (float)t.step_lag;
    bc1 = 1.0f - powf(beta1, own);
    rsqrt_bc2 = rsqrtf(1.0f - powf(beta2, own));
  }
  const float step = lr / bc1;
  const float decay = 1.0f - lr * wd;
  auto update = [&](float g, float& p, float& m, float& v) {
    g *= grad_scale;
    p *= decay;
    m = beta1 * m + (1.0f - beta1) * g;
    v = beta2 * v + (1.0f - beta2) * g * g;
    p -= step * (m / (sqrtf(v) * rsqrt_bc2 + eps));
  };
  const int64_t end = base + ADAMW_CHUNK < t.numel ? base + ADAMW_CHUNK : t.numel;
  const bool vec = (t.numel & 3) == 0 &&
                   ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0) &&
                   (!t.shadow || ((uintptr_t)t.shadow & 15) == 0);
  if (vec) {   // 16-byte streams: four per element in, three (+ the shadow) out
    for (int64_t j = base + 4 * threadIdx.x; j < end; j += 4 * 256) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(t.g + j);
      f32x4 p4 = *reinterpret_cast<const f32x4*>(t.p + j);
      f32x4 m4 = *reinterpret_cast<const f32x4*>(t.m + j);
      f32x4 v4 = *reinterpret_cast<const f32x4*>(t.v + j);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float p = p4[k], m = m4[k], v = v4[k];
        update(g4[k], p, m, v);
        p4[k] = p; m4[k] = m; v4[k] = v;
      }
      *reinterpret_cast<f32x4*>(t.p + j) = p4;
      *reinterpret_cast<f32x4*>(t.m + j) = m4;
      *reinterpret_cast<f32x4*>(t.v + j) = v4;
      if (t.shadow) {
        if (t.shadow_dtype == PTV3_BF16)
          *reinterpret_cast<s16x4*>(reinterpret_cast<__bf16*>(t.shadow) + j) = pack4<__bf16>(p4[0], p4[1], p4[2], p4[3]);
        else
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(t.shadow) + j) = p4;
      }
    }
    return;
  }
  for (int64_t j = base + threadIdx.x; j < end; j += 256) {
    float p = t.p[j], m = t.m[j], v = t.v[j];
    update(t.g[j], p, m, v);
    t.p[j] = p; t.m[j] = m; t.v[j] = v;
    if (t.shadow) {
      if (t.shadow_dtype == PTV3_BF16) reinterpret_cast<__bf16*>(t.shadow)[j] = (__bf16)p;
      else reinterpret_cast<float*>(t.shadow)[j] = p;
    }
  }
}

// shadow_t[(c * kvol + (kvol - 1 - d)) * rows + o] = shadow[(o * kvol + d) * cols + c]: the transposed (mirrored-tap)
// copy of every weight as its own pass of 64 x 64 tiles through LDS.  Written element by element from the update
// kernel the transposed copy was 46 M two-byte stores a whole weight row apart (each its own cache line; 1.0 ms of the
// step for 1.4 GB of streams); tiled, both sides move 128-byte row segments.
constexpr int SHT = 64;
template <typename E>
__global__ void __launch_bounds__(256) shadow_transpose_kernel(const AdamWTensor* __restrict__ tab, int ntensors) {
  __shared__ E tile[SHT][SHT + 2];
  int lo = 0, hi = ntensors - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const AdamWTensor t = tab[lo];
  if (!t.shadow_t) return;   // cannot happen for a consistent table
  const int tiles_o = (t.rows + SHT - 1) / SHT, tiles_c = (t.cols + SHT - 1) / SHT;
  int tl = (int)blockIdx.x - t.first_tile;
  const int tc = tl % tiles_c; tl /= tiles_c;
  const int to = tl % tiles_o;
  const int d = tl / tiles_o;
  const E* __restrict__ src = reinterpret_cast<const E*>(t.shadow);
  E* __restrict__ dst = reinterpret_cast<E*>(t.shadow_t);
  const int x = threadIdx.x & 63, y0 = threadIdx.x >> 6;
  for (int y = y0; y < SHT; y += 4) {          // rows o of the tile, lanes along c
    const int o = to * SHT + y, c = tc * SHT + x;
    if (o < t.rows && c < t.cols) tile[y][x] = src[((int64_t)o * t.kvol + d) * t.cols + c];
  }
  __syncthreads();
  for (int y = y0; y < SHT; y += 4) {          // rows c of the tile, lanes along o
    const int c = tc * SHT + y, o = to * SHT + x;
    if (o < t.rows && c < t.cols) dst[((int64_t)c * t.kvol + (t.kvol - 1 - d)) * t.rows + o] = tile[x][y];
  }
}

// sum of squares of every gradient (for clip_grad_norm_): per-block partials -> one value, deterministic
template <bool PTRS>
__global__ void __launch_bounds__(256) grad_sq_kernel(const AdamWTensor* __restrict__ tab, int tensor_lo, int tensor_hi,
                                                       int block_base, AdamWGrads gp, float* __restrict__ partial) {
  __shared__ float red[256];
  const int blk = block_base + (int)blockIdx.x;
  int lo = tensor_lo, hi = tensor_hi;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= blk) lo = mid; else hi = mid - 1;
  }
  AdamWTensor t = tab[lo];
  if (PTRS) t.g = gp.g[lo - tensor_lo];
  const int64_t base = (int64_t)(blk - t.first_block) * ADAMW_CHUNK;
  float s = 0.f;
  for (int64_t j = base + threadIdx.x; j < base + ADAMW_CHUNK && j < t.numel; j += 256) s += t.g[j] * t.g[j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blk] = red[0];
}

__global__ void __launch_bounds__(256) sum_partials_kernel(const float* __restrict__ partial, int n, float* out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) s += partial[j];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace ptv3

using namespace ptv3;

extern "C" size_t ptv3_adamw_entry_bytes(void) { return sizeof(AdamWTensor); }
extern "C" int ptv3_adamw_chunk(void) { return ADAMW_CHUNK; }

/* host helper: fill one table entry (the table is built on the host, then copied to the device by the caller) */
extern "C" int ptv3_adamw_fill_entry(void* entry_host, void* param, const void* grad, void* exp_avg, void* exp_avg_sq,
                                     int64_t numel, int group, int first_block) {
  PTV3_REQUIRE(group >= 0 && group < 8, "adamw: group %d outside [0,8)", group);
  AdamWTensor* e = (AdamWTensor*)entry_host;
  e->p = (float*)param; e->g = (const float*)grad; e->m = (float*)exp_avg; e->v = (float*)exp_avg_sq;
  e->numel = numel; e->group = group; e->first_block = first_block;
  e->shadow = nullptr; e->shadow_t = nullptr; e->rows = e->cols = e->kvol = 0; e->shadow_dtype = PTV3_F32;
  e->step_lag = 0; e->first_tile = 0;
  return PTV3_OK;
}

extern "C" int ptv3_adamw_fill_step_lag(void* entry_host, int64_t step_lag) {
  PTV3_REQUIRE(step_lag >= 0 && step_lag < (1ll << 31), "adamw: step_lag %lld outside [0, 2^31)", (long long)step_lag);
  ((AdamWTensor*)entry_host)->step_lag = (int32_t)step_lag;
  return PTV3_OK;
}

extern "C" int ptv3_adamw_shadow_tiles(int rows, int cols, int kvol) {
  return kvol * ((rows + SHT - 1) / SHT) * ((cols + SHT - 1) / SHT);
}

extern "C" int ptv3_adamw_fill_first_tile(void* entry_host, int first_tile) {
  PTV3_REQUIRE(first_tile >= 0, "adamw: first_tile %d", first_tile);
  ((AdamWTensor*)entry_host)->first_tile = first_tile;
  return PTV3_OK;
}

extern "C" int ptv3_adamw_fill_shadow(void* entry_host, void* shadow, void* shadow_t, int rows, int cols, int kvol,
                                      int shadow_dtype) {
  PTV3_REQUIRE(shadow != nullptr, "adamw: shadow pointer required");
  PTV3_REQUIRE(shadow_dtype == PTV3_F32 || shadow_dtype == PTV3_BF16, "adamw: bad shadow dtype");
  AdamWTensor* e = (AdamWTensor*)entry_host;
  PTV3_REQUIRE(shadow_t == nullptr || (rows > 0 && cols > 0 && kvol > 0 && (int64_t)rows * cols * kvol == e->numel),
               "adamw: shadow_t needs rows * kvol * cols == numel");
  e->shadow = shadow; e->shadow_t = shadow_t; e->rows = rows; e->cols = cols; e->kvol = kvol;
  e->shadow_dtype = shadow_dtype;
  return PTV3_OK;
}

// launches over the table in runs of <= ADAMW_PTRS tensors when gradient addresses come from the host
template <typename F>
static int adamw_runs(int ntensors, int total_blocks, const int32_t* first_block_host, const void* const* grads_host,
                      F&& launch) {
  if (!grads_host) {
    AdamWGrads none{};
    launch(false, 0, ntensors - 1, 0, total_blocks, none);
    return PTV3_OK;
  }
  PTV3_REQUIRE(first_block_host != nullptr, "adamw: gradient pointers need the tensors' first blocks");
  for (int t0 = 0; t0 < ntensors; t0 += ADAMW_PTRS) {
    const int t1 = t0 + ADAMW_PTRS < ntensors ? t0 + ADAMW_PTRS : ntensors;
    AdamWGrads gp{};
    for (int i = t0; i < t1; ++i) {
      PTV3_REQUIRE(grads_host[i] != nullptr, "adamw: gradient %d is NULL", i);
      gp.g[i - t0] = (const float*)grads_host[i];
    }
    const int b0 = first_block_host[t0], b1 = t1 < ntensors ? first_block_host[t1] : total_blocks;
    if (b1 > b0) launch(true, t0, t1 - 1, b0, b1 - b0, gp);
  }
  return PTV3_OK;
}

extern "C" int ptv3_adamw_step(const void* table_dev, int ntensors, int total_blocks, const float* lr_host,
                               const float* wd_host, int ngroups, float beta1, float beta2, float eps, int64_t step,
                               float grad_scale, const int32_t* first_block_host, const void* const* grads_host,
                               int total_tiles, int shadow_dtype, void* stream) {
  PTV3_REQUIRE(ngroups >= 1 && ngroups <= 8, "adamw: ngroups %d outside [1,8]", ngroups);
  PTV3_REQUIRE(step >= 1, "adamw: step must be >= 1");
  if (ntensors == 0 || total_blocks == 0) return PTV3_OK;
  AdamWGroups grp;
  for (int i = 0; i < 8; ++i) { grp.lr[i] = i < ngroups ? lr_host[i] : 0.f; grp.wd[i] = i < ngroups ? wd_host[i] : 0.f; }
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const AdamWTensor* tab = (const AdamWTensor*)table_dev;
  hipStream_t s = (hipStream_t)stream;
  const int rc = adamw_runs(ntensors, total_blocks, first_block_host, grads_host,
                            [&](bool ptrs, int lo, int hi, int b0, int nb, const AdamWGrads& gp) {
    if (ptrs)
      hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, tab, lo, hi, b0, gp, grp, beta1, beta2,
                         eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, (float)step);
    else
      hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, tab, lo, hi, b0, gp, grp, beta1, beta2,
                         eps, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, (float)step);
  });
  if (rc != PTV3_OK) return rc;
  if (total_tiles > 0) {
    if (shadow_dtype == PTV3_BF16)
      hipLaunchKernelGGL(shadow_transpose_kernel<short>, dim3((unsigned)total_tiles), dim3(256), 0, s, tab, ntensors);
    else
      hipLaunchKernelGGL(shadow_transpose_kernel<float>, dim3((unsigned)total_tiles), dim3(256), 0, s, tab, ntensors);
  }
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}

extern "C" int ptv3_grad_sqnorm(const void* table_dev, int ntensors, int total_blocks, float* partial_ws,
                                float* out, const int32_t* first_block_host, const void* const* grads_host,
                                void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (ntensors == 0 || total_blocks == 0) {
    if (hipMemsetAsync(out, 0, sizeof(float), s) != hipSuccess) return PTV3_ERR_LAUNCH;
    return PTV3_OK;
  }
  const AdamWTensor* tab = (const AdamWTensor*)table_dev;
  const int rc = adamw_runs(ntensors, total_blocks, first_block_host, grads_host,
                            [&](bool ptrs, int lo, int hi, int b0, int nb, const AdamWGrads& gp) {
    if (ptrs)
      hipLaunchKernelGGL(grad_sq_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, tab, lo, hi, b0, gp, partial_ws);
    else
      hipLaunchKernelGGL(grad_sq_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, tab, lo, hi, b0, gp, partial_ws);
  });
  if (rc != PTV3_OK) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial_ws, total_blocks, out);
  PTV3_LAUNCH_CHECK();
  return PTV3_OK;
}
