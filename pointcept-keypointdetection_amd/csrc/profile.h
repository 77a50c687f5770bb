// Optional per-launch device timing (HIP events on the launch stream) for bench.py's roofline section.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptv3 {
enum { PROF_LINEAR = 0, PROF_SUBM_CONV = 1, PROF_WINDOW_ATTN = 2, PROF_BACKWARD = 3, PROF_FAMILIES = 4 };
// which KERNEL a bracket times (one launch per bracket), so that a kernel's own algorithmic bytes / flops per launch
// can be set against its average duration in a rocprofv3 --kernel-trace --stats summary of the same command
enum {
  PK_GEMM64_DENSE = 0, PK_GEMM64_CONV, PK_GEMM32_DENSE, PK_GEMM32_CONV, PK_GEMM_BIG_DENSE, PK_GEMM_BIG_CONV,
  PK_BLOCK_HEAD, PK_BLOCK_TAIL, PK_BLOCK_HEAD_COOP, PK_BLOCK_TAIL_COOP, PK_MLP2, PK_ATTN_FULL, PK_ATTN_TILED,
  PK_BLOCK_HEAD_WIDE, PK_BLOCK_TAIL_WIDE, PK_CONV_TILE, PK_GEMM_TN, PK_ATTN_BWD_DQ, PK_ATTN_BWD_DKV, PK_SWIN_ATTN, PK_ROWS_LINEAR, PK_SWIN_ATTN_MFMA, PK_KERNELS
};
void prof_kernel(int rec, int kernel);   // tag the bracket opened by prof_begin (default: by family)
bool prof_on();
// nbr != NULL: algorithmic flops = flops_per_valid * (#entries >= 0), counted on the device outside the bracket
int prof_begin(hipStream_t s, int family, double flops, double bytes, const int32_t* nbr, int64_t nbr_count,
               double flops_per_valid);
void prof_end(int rec, hipStream_t s);
}  // namespace ptv3
