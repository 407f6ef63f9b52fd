// Lane layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction), found by one-hot inputs:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_4x4x4_layout.hip -o /tmp/mfma4 && /tmp/mfma4
// prints, for every lane, which (block, row, k) its A operand is, which (block, k, column) its B operand is and which
// (block, row, column) its result is.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k(const double* A, const double* B, double* D) {
  const int l = threadIdx.x;
  D[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
}

int main() {
  double *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
  std::vector<double> a(64), b(64), d(64);
  // pairs[la][lb] = result lane that receives A[la] * B[lb] (or -1)
  std::vector<int> pair(64 * 64, -1);
  for (int la = 0; la < 64; ++la) {
    for (int i = 0; i < 64; ++i) { a[i] = (i == la) ? 1.0 : 0.0; b[i] = 1.0 + i; }  // distinct B values identify lb
    hipMemcpy(dA, a.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dB, b.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(d.data(), dD, 512, hipMemcpyDeviceToHost);
    for (int ld = 0; ld < 64; ++ld)
      if (d[ld] != 0.0) {
        const int lb = (int)(d[ld] + 0.5) - 1;
        if (lb >= 0 && lb < 64 && d[ld] == 1.0 + lb) pair[la * 64 + lb] = ld;
        else printf("lane %d: A one-hot gives a sum %g in result lane %d (more than one term?)\n", la, d[ld], ld);
      }
  }
  // every A lane pairs with 4 B lanes (the 4 columns j at its k) and lands in 4 result lanes (its row i, columns j)
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d ->", la);
    for (int lb = 0; lb < 64; ++lb)
      if (pair[la * 64 + lb] >= 0) printf("  (B lane %2d -> D lane %2d)", lb, pair[la * 64 + lb]);
    printf("\n");
  }
  // the layout csrc/zgemm.hip assumes (MITDVP_B4_*): block = (l / 4) % 4; A(i = l % 4, k = l / 16); B(j = l % 4, k = l / 16);
  // D(i = l / 16, j = l % 4)
  bool ok = true;
  for (int la = 0; la < 64 && ok; ++la)
    for (int lb = 0; lb < 64 && ok; ++lb) {
      const int ba = (la / 4) % 4, ia = la % 4, ka = la / 16, bb = (lb / 4) % 4, jb = lb % 4, kb = lb / 16;
      const int want = (ba == bb && ka == kb) ? ia * 16 + ba * 4 + jb : -1;
      if (pair[la * 64 + lb] != want) ok = false;
    }
  printf("layout assumed by csrc/zgemm.hip: %s\n", ok ? "FITS" : "does NOT fit");
  return 0;
}
