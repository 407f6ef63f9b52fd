// small_linalg.h -- host-side dense algebra on the k x k (k <= 21) Krylov
// projections.  The reference does this on the host too
// (scipy.linalg.eigh_tridiagonal / eig + solve, _integrator.py:401-409,
// :617-637); here it is self-contained C++ (no LAPACK in the C-ABI library).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

namespace mitdvp {

typedef std::complex<double> hzc;

// Cyclic Jacobi eigen-decomposition of a real symmetric matrix (row-major n x n).
// On return a holds the eigenvalues on its diagonal, v the eigenvectors (columns).
inline void jacobi_eigh(int n, std::vector<double>& a, std::vector<double>& v) {
  v.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag += a[(size_t)p * n + p] * a[(size_t)p * n + p];
      for (int q = p + 1; q < n; ++q) off += a[(size_t)p * n + q] * a[(size_t)p * n + q];
    }
    if (off <= 1e-34 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {  // columns p, q
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - s * akq;
          a[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {  // rows p, q
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - s * aqk;
          a[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
          v[(size_t)k * n + p] = c * vkp - s * vkq;
          v[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
}

// Eigen-decomposition of a real symmetric tridiagonal matrix by QL with implicit
// shifts (the classic tql2/tqli algorithm): d = diagonal (n), e = sub-diagonal
// (n-1).  On return d holds the eigenvalues (unsorted) and z (row-major n x n)
// the eigenvectors in its columns.  Returns false if an eigenvalue needs > 100 sweeps.
inline bool tridiag_ql(int n, std::vector<double>& d, std::vector<double> e, std::vector<double>& z) {
  z.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) z[(size_t)i * n + i] = 1.0;
  e.resize(n, 0.0);
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
        if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
      }
      if (m != l) {
        if (iter++ == 100) return false;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = std::hypot(g, 1.0);
        g = d[m] - d[l] + e[l] / (g + (g >= 0 ? std::fabs(r) : -std::fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i];
          const double b = c * e[i];
          e[i + 1] = (r = std::hypot(f, g));
          if (r == 0.0) {
            d[i + 1] -= p;
            e[m] = 0.0;
            break;
          }
          s = f / r;
          c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
          for (int k = 0; k < n; ++k) {
            f = z[(size_t)k * n + i + 1];
            z[(size_t)k * n + i + 1] = s * z[(size_t)k * n + i] + c * f;
            z[(size_t)k * n + i] = c * z[(size_t)k * n + i] - s * f;
          }
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p;
        e[l] = g;
        e[m] = 0.0;
      }
    } while (m != l);
  }
  return true;
}

// eigenvector of the `root`-th smallest eigenvalue of T(alpha, beta), with the
// sign fixed so that its first component is non-negative
inline std::vector<double> tridiag_eigvec(const std::vector<double>& alpha, const std::vector<double>& beta, int k,
                                          int root, double* eigval) {
  std::vector<double> d(alpha.begin(), alpha.begin() + k), e(beta.begin(), beta.begin() + std::max(k - 1, 0)), z;
  if (!tridiag_ql(k, d, e, z)) return {};
  std::vector<int> idx(k);
  for (int i = 0; i < k; ++i) idx[i] = i;
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return d[a] < d[b]; });
  const int col = idx[root < 0 ? k + root : root];
  std::vector<double> v(k);
  for (int i = 0; i < k; ++i) v[i] = z[(size_t)i * k + col];
  if (v[0] < 0)
    for (auto& x : v) x = -x;
  if (eigval) *eigval = d[col];
  return v;
}

// coef = exp(scale * T) e0 for the real symmetric tridiagonal T(alpha, beta):
// Phi (exp(scale*lambda) * Phi^T e0), like _integrator.py:617-621,635.
inline std::vector<hzc> expm_tridiag_e0(const std::vector<double>& alpha, const std::vector<double>& beta, int k,
                                        hzc scale) {
  std::vector<double> a((size_t)k * k, 0.0), v;
  for (int i = 0; i < k; ++i) {
    a[(size_t)i * k + i] = alpha[i];
    if (i + 1 < k) a[(size_t)i * k + i + 1] = a[(size_t)(i + 1) * k + i] = beta[i];
  }
  jacobi_eigh(k, a, v);
  std::vector<hzc> coef(k, hzc(0, 0));
  for (int e = 0; e < k; ++e) {
    const hzc w = std::exp(scale * a[(size_t)e * k + e]) * v[(size_t)0 * k + e];
    for (int i = 0; i < k; ++i) coef[i] += v[(size_t)i * k + e] * w;
  }
  return coef;
}

// first column of exp(A) for a general complex k x k matrix (row-major):
// scaling and squaring with a degree-20 Taylor polynomial.  Mathematically the
// eig + solve form of _integrator.py:402-408 / :623-635.
inline std::vector<hzc> expm_col0(std::vector<hzc> A, int k) {
  double nrm = 0.0;
  for (int j = 0; j < k; ++j) {
    double s = 0.0;
    for (int i = 0; i < k; ++i) s += std::abs(A[(size_t)i * k + j]);
    nrm = std::max(nrm, s);
  }
  int sq = 0;
  while (nrm > 0.5 && sq < 60) { nrm *= 0.5; ++sq; }
  const double sc = std::ldexp(1.0, -sq);
  for (auto& x : A) x *= sc;
  auto matmul = [k](const std::vector<hzc>& X, const std::vector<hzc>& Y) {
    std::vector<hzc> Z((size_t)k * k, hzc(0, 0));
    for (int i = 0; i < k; ++i)
      for (int l = 0; l < k; ++l) {
        const hzc x = X[(size_t)i * k + l];
        if (x == hzc(0, 0)) continue;
        for (int j = 0; j < k; ++j) Z[(size_t)i * k + j] += x * Y[(size_t)l * k + j];
      }
    return Z;
  };
  std::vector<hzc> E((size_t)k * k, hzc(0, 0)), P((size_t)k * k, hzc(0, 0));
  for (int i = 0; i < k; ++i) E[(size_t)i * k + i] = P[(size_t)i * k + i] = 1.0;
  for (int deg = 1; deg <= 20; ++deg) {
    P = matmul(P, A);
    const double inv = 1.0 / deg;
    for (auto& x : P) x *= inv;
    for (size_t i = 0; i < E.size(); ++i) E[i] += P[i];
  }
  for (int s = 0; s < sq; ++s) E = matmul(E, E);
  std::vector<hzc> col(k);
  for (int i = 0; i < k; ++i) col[i] = E[(size_t)i * k];
  return col;
}

}  // namespace mitdvp
