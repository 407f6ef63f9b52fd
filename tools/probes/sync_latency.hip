// How a few hundred bytes of kernel results reach the host: (a) hipMemcpyAsync D2H + hipStreamSynchronize (what the Krylov
// loop does), (b) the kernel writes host-mapped memory, hipStreamSynchronize, (c) the kernel writes host-mapped memory and a
// sequence word, the host spins on the word.  Microseconds per round trip (launch of a short kernel included).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/sync_latency.hip -o /tmp/sl && /tmp/sl
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void produce(double* dev, double* host, volatile unsigned* seq, unsigned s, int n) {
  const int t = threadIdx.x;
  double a = dev[t];
  for (int i = 0; i < n; ++i) a = a * 1.0000001 + 1e-9;
  dev[t] = a;
  if (host) host[t] = a;
  if (seq) {
    __threadfence_system();
    __syncthreads();
    if (t == 0) *seq = s;
  }
}
int main() {
  double *d, *h, *hm; unsigned* sq;
  CK(hipMalloc(&d, 64 * 8)); CK(hipMemset(d, 0, 64 * 8));
  CK(hipHostMalloc((void**)&h, 64 * 8, hipHostMallocDefault));
  CK(hipHostMalloc((void**)&hm, 64 * 8, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc((void**)&sq, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *sq = 0;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int N = 3000;
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 1; i <= N; ++i) {
      if (mode == 0) {
        hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, st, d, (double*)nullptr, (volatile unsigned*)nullptr, 0u, 10);
        CK(hipMemcpyAsync(h, d, 64 * 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
      } else if (mode == 1) {
        hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, st, d, hm, (volatile unsigned*)nullptr, 0u, 10);
        CK(hipStreamSynchronize(st));
      } else {
        hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, st, d, hm, (volatile unsigned*)sq, (unsigned)i, 10);
        while (*(volatile unsigned*)sq != (unsigned)i) { }
      }
    }
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    const char* names[3] = {"memcpyAsync D2H + streamSynchronize", "host-mapped write + streamSynchronize", "host-mapped write + spin on a sequence word"};
    printf("%-46s %.2f us per round trip\n", names[mode], us / N);
  }
  return 0;
}
