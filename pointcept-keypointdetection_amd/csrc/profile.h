// Optional per-launch device timing (HIP events on the launch stream) for bench.py's roofline section.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptv3 {
enum { PROF_LINEAR = 0, PROF_SUBM_CONV = 1, PROF_WINDOW_ATTN = 2, PROF_FAMILIES = 3 };
bool prof_on();
// nbr != NULL: algorithmic flops = flops_per_valid * (#entries >= 0), counted on the device outside the bracket
int prof_begin(hipStream_t s, int family, double flops, double bytes, const int32_t* nbr, int64_t nbr_count,
               double flops_per_valid);
void prof_end(int rec, hipStream_t s);
}  // namespace ptv3
