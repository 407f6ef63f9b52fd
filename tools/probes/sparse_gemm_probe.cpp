// Times the block-sparse W-stage GEMM (batched 512 x 1024 x 512, batch 1024: the C4 interior S2) for several
// K-tile lists: all row tiles light (1 tile), the finite-state-machine pattern (7 light + 1 full), all full, dense path.
// Build on the GPU box:  hipcc -O2 --offload-arch=gfx950 -Ipytdscf_amd/csrc tools/probes/sparse_gemm_probe.cpp -Lpytdscf_amd/csrc -lmitdvp -Wl,-rpath,$PWD/pytdscf_amd/csrc -o /tmp/sgp
#include <cstdio>
#include <vector>
#include "common.h"
using namespace mitdvp;
int main() {
  const int M = 512, N = 1024, K = 512, batch = 1024, d = 16, mr = 32;
  zc *A, *B, *C; int* kl;
  HIP_CHECK(hipMalloc(&A, (size_t)M * K * sizeof(zc)));
  HIP_CHECK(hipMalloc(&B, (size_t)batch * K * N * sizeof(zc)));
  HIP_CHECK(hipMalloc(&C, (size_t)batch * M * N * sizeof(zc)));
  HIP_CHECK(hipMemset(A, 0, (size_t)M * K * sizeof(zc)));
  HIP_CHECK(hipMemset(B, 0, (size_t)batch * K * N * sizeof(zc)));
  const int ntm = M / 64, nkt = K / 16, stride = nkt + 1;
  HIP_CHECK(hipMalloc(&kl, ntm * stride * sizeof(int)));
  hipStream_t st; HIP_CHECK(hipStreamCreate(&st));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int mode, bool rowmap) {
    std::vector<int> h(ntm * stride, 0);
    for (int tm = 0; tm < ntm; ++tm) {
      int cnt = mode == 0 ? 1 : (mode == 1 ? (tm == ntm - 1 ? nkt : 1) : nkt);
      h[tm * stride] = cnt;
      for (int q = 0; q < cnt; ++q) h[tm * stride + 1 + q] = q;
    }
    HIP_CHECK(hipMemcpy(kl, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    ZgemmDesc g = zgemm_desc(A, B, C, M, N, K);
    g.batch = batch; g.strideA = 0; g.strideB = (long)K * N; g.strideC = (long)M * N;
    if (mode >= 0) { g.klist = kl; g.klist_stride = stride; g.tile_cfg = 1; if (rowmap) { g.rowmap_p = d; g.rowmap_s1 = (long)mr * N; g.rowmap_s2 = N; } }
    zgemm(st, g);
    HIP_CHECK(hipStreamSynchronize(st));
    hipEventRecord(e0, st);
    for (int r = 0; r < 3; ++r) zgemm(st, g);
    hipEventRecord(e1, st);
    HIP_CHECK(hipStreamSynchronize(st));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms\n", name, ms / 3);
  };
  run("dense path", -1, false);
  run("list: all full", 2, false);
  run("list: FSM (7 light + 1 full)", 1, false);
  run("list: FSM + row map", 1, true);
  run("list: all light", 0, false);
  run("list: all light + row map", 0, true);
  return 0;
}
