// Issue rate of v_exp_f32 against v_mul_f32 / v_cvt_pk_bf16_f32 / v_max3_f32 on gfx950: what bounds the
// softmax part of the window-attention kernel (one exp per score at head_dim 16).
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o pointcept-keypointdetection_amd/build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(1024) rate_kernel(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                   "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (OP == 1) {
      asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                   "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (OP == 2) {
      asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n"
                   "v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (OP == 3) {
      asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                   "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else {  // 4 exp interleaved with 4 mul: does the transcendental unit run beside the main VALU?
      asm volatile("v_exp_f32 %0, %0\n v_mul_f32 %4, %4, %4\n v_exp_f32 %1, %1\n v_mul_f32 %5, %5, %5\n"
                   "v_exp_f32 %2, %2\n v_mul_f32 %6, %6, %6\n v_exp_f32 %3, %3\n v_mul_f32 %7, %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
static float run(float* d, int wgs, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(wgs), dim3(threads), 0, 0, d, 100, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(wgs), dim3(threads), 0, 0, d, iters, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* d;
  hipMalloc(&d, 4 << 20);
  const int iters = 200000;
  const char* names[5] = {"v_exp_f32", "v_mul_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "exp+mul interleaved"};
  for (int wps = 1; wps <= 4; wps *= 2) {
    // 256 workgroups x (4 * wps) waves: wps waves on every SIMD of a 256-CU part
    float ms[5];
    ms[0] = run<0>(d, 256, 256 * wps, iters); ms[1] = run<1>(d, 256, 256 * wps, iters); ms[2] = run<2>(d, 256, 256 * wps, iters);
    ms[3] = run<3>(d, 256, 256 * wps, iters); ms[4] = run<4>(d, 256, 256 * wps, iters);
    for (int k = 0; k < 5; ++k)
      printf("%d wave(s)/SIMD  %-22s %8.3f ms  -> %.2f ns per wave-instruction, %.2fx v_mul\n", wps, names[k], ms[k],
             ms[k] * 1e6 / (8.0 * iters * wps), ms[k] / ms[1]);
  }
  return 0;
}
