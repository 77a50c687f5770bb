// Host cost of HIP launch calls on this box: what bounds the executor's ~270 launches per forward.
// build: hipcc --offload-arch=gfx950 -O2 -pthread tools/launch_rate.hip -o pointcept-keypointdetection_amd/build/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void tiny(int* p, int n) { if (p && threadIdx.x == 0 && blockIdx.x == 0 && n < 0) *p = n; }
struct Big { int v[24]; };
__global__ void tiny_big(Big b, int* p) { if (p && b.v[0] < 0) *p = b.v[1]; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s[4];
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  hipEvent_t ev[8];
  for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  const int N = 4000;
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], nullptr, 0);
  hipDeviceSynchronize();
  double t = now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], nullptr, 0);
  double t1 = now();
  hipDeviceSynchronize();
  printf("one stream, tiny args:        %.2f us/launch host (%.2f us/launch incl. drain)\n", (t1 - t) / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_big, dim3(1), dim3(64), 0, s[0], Big{}, nullptr);
  t1 = now();
  hipDeviceSynchronize();
  printf("one stream, 96-byte struct:   %.2f us/launch host (%.2f incl. drain)\n", (t1 - t) / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[i & 1], nullptr, 0);
  t1 = now();
  hipDeviceSynchronize();
  printf("alternating two streams:      %.2f us/launch host (%.2f incl. drain)\n", (t1 - t) / N, (now() - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) {
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[0], nullptr, 0);
    if ((i & 7) == 0) { hipEventRecord(ev[0], s[0]); hipStreamWaitEvent(s[1], ev[0], 0); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[1], nullptr, 0); }
  }
  t1 = now();
  hipDeviceSynchronize();
  printf("launch + (record,wait,launch other stream)/8: %.2f us/iteration host\n", (t1 - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) { hipEventRecord(ev[i & 7], s[0]); }
  t1 = now();
  hipDeviceSynchronize();
  printf("hipEventRecord:               %.2f us host\n", (t1 - t) / N);
  t = now();
  for (int i = 0; i < N; ++i) { hipStreamWaitEvent(s[1], ev[i & 7], 0); }
  t1 = now();
  hipDeviceSynchronize();
  printf("hipStreamWaitEvent:           %.2f us host\n", (t1 - t) / N);
  int* d; hipMalloc(&d, 1 << 20);
  t = now();
  for (int i = 0; i < N; ++i) hipMemsetAsync(d, 0, 4096, s[0]);
  t1 = now();
  hipDeviceSynchronize();
  printf("hipMemsetAsync 4 KB:          %.2f us host\n", (t1 - t) / N);
  // two host threads, one stream each
  for (int nt = 2; nt <= 3; ++nt) {
    t = now();
    std::vector<std::thread> th;
    for (int k = 0; k < nt; ++k)
      th.emplace_back([&, k] { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s[k], nullptr, 0); });
    for (auto& x : th) x.join();
    t1 = now();
    hipDeviceSynchronize();
    printf("%d threads x own stream:       %.2f us per launch per thread host, aggregate %.2f us/launch (%.2f incl. drain)\n", nt,
           (t1 - t) / N, (t1 - t) / N / nt, (now() - t) / N / nt);
  }
  // longer kernels: does a busy GPU slow the launch call?
  return 0;
}
