// Probe: does an LDS-DMA buffer load (buffer_load_dwordx4 ... lds) whose offset lies beyond num_records write ZEROS to
// its LDS destination (the register form returns zeros), or leave the destination untouched?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = -7.0f;   // sentinel
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 4, 0x00020000);
  // lanes 0..39 in range, lanes 40..63 beyond num_records
  const unsigned off = threadIdx.x < 40 ? threadIdx.x * 16 : 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += blockDim.x) out[i] = s[i];
}
int main() {
  float h[256], *din, *dout;
  for (int i = 0; i < 256; ++i) h[i] = 1.0f + i;
  hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof h);
  hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, din, dout, 256);
  hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
  printf("in-range lane 39: %g %g %g %g\n", h[156], h[157], h[158], h[159]);
  printf("out-of-range lane 40: %g %g %g %g   lane 63: %g %g %g %g\n", h[160], h[161], h[162], h[163], h[252], h[253], h[254], h[255]);
  printf("%s\n", h[160] == 0.f && h[255] == 0.f ? "OOB_WRITES_ZEROS" : (h[160] == -7.f ? "OOB_LEAVES_LDS_UNTOUCHED" : "OOB_OTHER"));
  return 0;
}
