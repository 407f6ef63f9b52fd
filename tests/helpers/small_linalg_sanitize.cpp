// Standalone driver of the host-side Krylov-space algebra for AddressSanitizer /
// UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool: the host code is
// what can be sanitised).  Exits 0 when every identity holds.
#include <cstdio>
#include <random>

#include "../../pytdscf_amd/csrc/small_linalg.h"

using namespace mitdvp;

int main() {
  std::mt19937_64 rng(7);
  std::normal_distribution<double> g(0.0, 1.0);
  int bad = 0;
  for (int k : {1, 2, 3, 7, 20, 21, 64}) {
    std::vector<double> a(k), b(k);
    for (int i = 0; i < k; ++i) { a[i] = g(rng); b[i] = std::fabs(g(rng)) + 0.1; }
    // eigenvector residual of the tridiagonal matrix
    for (int root : {0, -1}) {
      double val = 0;
      auto v = tridiag_eigvec(a, b, k, root, &val);
      if ((int)v.size() != k) { ++bad; continue; }
      double res = 0, nrm = 0;
      for (int i = 0; i < k; ++i) {
        double r = a[i] * v[i] - val * v[i];
        if (i > 0) r += b[i - 1] * v[i - 1];
        if (i + 1 < k) r += b[i] * v[i + 1];
        res += r * r; nrm += v[i] * v[i];
      }
      if (!(std::sqrt(res) < 1e-10 && std::fabs(nrm - 1) < 1e-10)) ++bad;
    }
    // exp(i t T) e0 has unit norm for real symmetric T
    auto c = expm_tridiag_e0(a, b, k, hzc(0.0, -0.37));
    double n2 = 0;
    for (auto& x : c) n2 += std::norm(x);
    if (!(std::fabs(n2 - 1) < 1e-12)) ++bad;
    // general complex matrix: expm_col0(A) and expm_col0(-A) are columns of inverse matrices
    std::vector<hzc> A((size_t)k * k), mA((size_t)k * k);
    for (size_t i = 0; i < A.size(); ++i) { A[i] = 0.3 * hzc(g(rng), g(rng)); mA[i] = -A[i]; }
    auto e0 = expm_col0(A, k);
    if ((int)e0.size() != k) ++bad;
    // skew-Hermitian part only -> unitary -> unit-norm column
    std::vector<hzc> S((size_t)k * k);
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) S[(size_t)i * k + j] = 0.5 * (A[(size_t)i * k + j] - std::conj(A[(size_t)j * k + i]));
    auto u0 = expm_col0(S, k);
    n2 = 0;
    for (auto& x : u0) n2 += std::norm(x);
    if (!(std::fabs(n2 - 1) < 1e-11)) ++bad;
    // symmetric eigen-decomposition: V diag V^T reproduces the matrix
    std::vector<double> M((size_t)k * k), M0, V;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j <= i; ++j) M[(size_t)i * k + j] = M[(size_t)j * k + i] = g(rng);
    M0 = M;
    jacobi_eigh(k, M, V);
    double err = 0;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) {
        double s = 0;
        for (int e = 0; e < k; ++e) s += V[(size_t)i * k + e] * M[(size_t)e * k + e] * V[(size_t)j * k + e];
        err = std::max(err, std::fabs(s - M0[(size_t)i * k + j]));
      }
    if (!(err < 1e-10)) ++bad;
  }
  std::printf("small_linalg sanitize run: %d failures\n", bad);
  return bad ? 1 : 0;
}
