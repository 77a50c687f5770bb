// Per-launch HIP-event timing behind ptv3_profile_enable / ptv3_profile_collect (include/ptv3_hip.h).
#include <vector>
#include "common.h"
#include "profile.h"
#include "../../include/ptv3_hip.h"

namespace ptv3 {

struct ProfRec { int family; double flops, bytes, per_valid; hipEvent_t e0, e1; int slot; int kernel; };
static const char* const g_kernel_names[PK_KERNELS] = {
    "gemm_kernel<64ch> dense", "gemm_kernel<64ch> gather (sparse conv)", "gemm_kernel<32ch> dense",
    "gemm_kernel<32ch> gather (sparse conv)", "gemm_big_kernel dense", "gemm_big_kernel gather (sparse conv)",
    "block_head_kernel", "block_tail_kernel", "block_head_coop_kernel", "block_tail_coop_kernel", "mlp2_kernel",
    "window_attn_full_kernel", "window_attn_kernel", "block_head_wide_kernel", "block_tail_wide_kernel",
    "conv_tile_kernel (sparse conv)", "gemm_tn_kernel", "attn_bwd_dq_kernel", "attn_bwd_dkv_kernel",
    "swin_attn_kernel", "rows_linear_kernel", "swin_attn_mfma_kernel"};
static bool g_on = false;
static double g_hint_flops = -1.0;   // flops of the NEXT bracket, set by the caller that knows them (ptv3_profile_hint_flops)
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static unsigned long long* g_slots = nullptr;
static int g_nslots = 0;
constexpr int MAX_SLOTS = 1 << 16;

__global__ void count_valid_kernel(const int32_t* __restrict__ nbr, int64_t n, unsigned long long* out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long c = 0;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) c += nbr[i] >= 0;
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

bool prof_on() { return g_on; }

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

int prof_begin(hipStream_t s, int family, double flops, double bytes, const int32_t* nbr, int64_t nbr_count,
               double flops_per_valid) {
  if (!g_on) return -1;
  static const int default_kernel[PROF_FAMILIES] = {PK_GEMM64_DENSE, PK_GEMM64_CONV, PK_ATTN_FULL, PK_GEMM_TN};
  if (g_hint_flops >= 0.0) { flops = g_hint_flops; g_hint_flops = -1.0; }
  ProfRec r{family, flops, bytes, flops_per_valid, get_event(), get_event(), -1, default_kernel[family]};
  if (nbr && g_slots && g_nslots < MAX_SLOTS) {
    r.slot = g_nslots++;
    hipLaunchKernelGGL(count_valid_kernel, dim3(512), dim3(256), 0, s, nbr, nbr_count, g_slots + r.slot);
  }
  (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void prof_end(int rec, hipStream_t s) {
  if (rec >= 0) (void)hipEventRecord(g_recs[rec].e1, s);
}

void prof_kernel(int rec, int kernel) {
  if (rec >= 0 && kernel >= 0 && kernel < PK_KERNELS) g_recs[rec].kernel = kernel;
}

}  // namespace ptv3

using namespace ptv3;

extern "C" int ptv3_profile_enable(int on) {
  if (on && !g_slots) {
    if (hipMalloc(&g_slots, (size_t)MAX_SLOTS * 8) != hipSuccess) { set_error("profile: hipMalloc failed"); return PTV3_ERR_LAUNCH; }
  }
  if (on) {
    (void)hipMemset(g_slots, 0, (size_t)MAX_SLOTS * 8);
    g_nslots = 0;
    for (auto& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
    g_recs.clear();
  }
  g_on = on != 0;
  return PTV3_OK;
}

extern "C" int ptv3_profile_hint_flops(double flops) {
  g_hint_flops = flops;
  return PTV3_OK;
}

extern "C" int ptv3_profile_kernel_count(void) { return PK_KERNELS; }
extern "C" const char* ptv3_profile_kernel_name(int kernel) {
  return kernel >= 0 && kernel < PK_KERNELS ? g_kernel_names[kernel] : "";
}

static int collect(int by_kernel, int n, double* ms, double* flops, double* bytes, int64_t* launches) {
  if (hipDeviceSynchronize() != hipSuccess) { set_error("profile: synchronize failed"); return PTV3_ERR_LAUNCH; }
  std::vector<unsigned long long> slots(g_nslots > 0 ? g_nslots : 1);
  if (g_nslots > 0) (void)hipMemcpy(slots.data(), g_slots, (size_t)g_nslots * 8, hipMemcpyDeviceToHost);
  for (int f = 0; f < n; ++f) { ms[f] = flops[f] = bytes[f] = 0.0; launches[f] = 0; }
  for (auto& r : g_recs) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    const int k = by_kernel ? r.kernel : r.family;
    ms[k] += t;
    flops[k] += r.slot >= 0 ? r.per_valid * (double)slots[r.slot] : r.flops;
    bytes[k] += r.bytes;
    launches[k] += 1;
  }
  return PTV3_OK;
}

extern "C" int ptv3_profile_collect_kernels(double* ms, double* flops, double* bytes, int64_t* launches) {
  return collect(1, PK_KERNELS, ms, flops, bytes, launches);   // does not reset: call before ptv3_profile_collect
}

extern "C" int ptv3_profile_collect(double* ms, double* flops, double* bytes, int64_t* launches) {
  if (int rc = collect(0, PROF_FAMILIES, ms, flops, bytes, launches)) return rc;
  for (auto& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
  g_recs.clear();
  if (g_slots) (void)hipMemset(g_slots, 0, (size_t)MAX_SLOTS * 8);
  g_nslots = 0;
  return PTV3_OK;
}
