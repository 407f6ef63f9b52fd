#include "../../pytdscf_amd/csrc/small_linalg.h"
#include <cstdio>
extern "C" {
int sl_tridiag_eigvec(const double* a, const double* b, int k, int root, double* vec, double* val) {
  std::vector<double> A(a, a + k), B(b, b + (k > 1 ? k - 1 : 0));
  auto v = mitdvp::tridiag_eigvec(A, B, k, root, val);
  if (v.empty()) return 1;
  for (int i = 0; i < k; ++i) vec[i] = v[i];
  return 0;
}
int sl_expm_tridiag(const double* a, const double* b, int k, double sre, double sim, double* out) {
  std::vector<double> A(a, a + k), B(b, b + k);
  auto c = mitdvp::expm_tridiag_e0(A, B, k, mitdvp::hzc(sre, sim));
  for (int i = 0; i < k; ++i) { out[2 * i] = c[i].real(); out[2 * i + 1] = c[i].imag(); }
  return 0;
}
int sl_expm_col0(const double* m, int k, double* out) {
  std::vector<mitdvp::hzc> A(k * k);
  for (int i = 0; i < k * k; ++i) A[i] = mitdvp::hzc(m[2 * i], m[2 * i + 1]);
  auto c = mitdvp::expm_col0(A, k);
  for (int i = 0; i < k; ++i) { out[2 * i] = c[i].real(); out[2 * i + 1] = c[i].imag(); }
  return 0;
}
}
