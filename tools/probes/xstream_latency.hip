// Latency of a cross-stream dependency (hipEventRecord on one stream, hipStreamWaitEvent on the other) against a
// same-stream dependency, with short kernels: what a look-ahead schedule on two streams would pay per hand-over.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/xstream_latency.hip -o /tmp/xs && /tmp/xs
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(double* p, int n) {
  double a = p[threadIdx.x];
  for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9;
  p[threadIdx.x] = a;
}
int main() {
  double* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int N = 2000;
  hipEvent_t ev[2]; CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  for (int work : {10, 2000}) {
    for (int mode = 0; mode < 2; ++mode) {
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (mode == 0) {
          hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, work);
          hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d + 64, work);
        } else {
          hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, work);
          CK(hipEventRecord(ev[0], s1)); CK(hipStreamWaitEvent(s2, ev[0], 0));
          hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, d + 64, work);
          CK(hipEventRecord(ev[1], s2)); CK(hipStreamWaitEvent(s1, ev[1], 0));
        }
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("work %5d  %s: %.2f us per pair of dependent kernels\n", work, mode ? "two streams + events" : "one stream          ", us / N);
    }
  }
  return 0;
}
