// svd.hip -- complex SVD of bond matrices on the GPU (one-sided Jacobi).
//
// Reference use: SiteCoef-level bond truncation ``truncate_sigvec``
// (_site_cls.py:586-690, scipy.linalg.svd of the bond matrix sigma), two-site
// canonicalisation and the Kraus / MPI joint-bond re-splits.
//
// Algorithm: Hestenes one-sided Jacobi on the ROWS of M (rows are contiguous in
// the row-major layout the engine uses everywhere): unitary 2x2 rotations are
// applied to row pairs until all rows are mutually orthogonal,
//     W M = S Q   (Q with orthonormal rows)   =>   M = W^H S Q,
// i.e. U = W^H, Vh = Q, singular values = row norms.  A round-robin tournament
// gives n/2 independent row pairs per step: one workgroup per pair, n-1 steps
// per sweep, convergence when every normalised inner product is below 1e-15.
// Unconditionally stable and accurate to working precision for small singular
// values as well, which is what a truncation criterion looks at.
#include "svd.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "vecops.h"

namespace mitdvp {

__device__ __forceinline__ double svd_block_sum(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return sh[4];
}

// one tournament step: workgroup k rotates the row pair it is assigned in round r
// tiny2: squared norm below which a row counts as numerically null (1e-28 x the largest squared row norm
// of the input, i.e. a singular value 1e-14 below the largest): such rows carry rounding noise only, their
// normalised inner products are O(1) for ever, so they neither rotate nor hold up the convergence test
// (rank-deficient matrices: direct sums with repeated channels, zero-padded bond matrices).
__global__ __launch_bounds__(256) void k_jacobi_step(zc* __restrict__ M, zc* __restrict__ W, int nrow, int ncol, int np,
                                                     int round, unsigned long long* __restrict__ offmax, double tiny2) {
  __shared__ double sh[5];
  // round-robin pairing of np (even) players; player np-1 is fixed
  const int k = blockIdx.x;
  int p, q;
  if (k == 0) { p = np - 1; q = round; }
  else { p = (round + k) % (np - 1); q = (round - k + (np - 1)) % (np - 1); }
  if (p >= nrow || q >= nrow) return;  // padding player of an odd tournament
  zc* x = M + (size_t)p * ncol;
  zc* y = M + (size_t)q * ncol;
  double a = 0, b = 0, gr = 0, gi = 0;
  for (int c = threadIdx.x; c < ncol; c += 256) {
    const zc xv = x[c], yv = y[c];
    a += xv.x * xv.x + xv.y * xv.y;
    b += yv.x * yv.x + yv.y * yv.y;
    gr += xv.x * yv.x + xv.y * yv.y;  // x conj(y)
    gi += xv.y * yv.x - xv.x * yv.y;
  }
  a = svd_block_sum(a, sh);
  b = svd_block_sum(b, sh);
  gr = svd_block_sum(gr, sh);
  gi = svd_block_sum(gi, sh);
  const double g2 = gr * gr + gi * gi;
  if (!(g2 > 0.0) || !(a > tiny2) || !(b > tiny2)) return;
  const double rel = g2 / (a * b);
  if (threadIdx.x == 0) atomicMax(offmax, (unsigned long long)__double_as_longlong(rel));
  if (rel <= 1e-32) return;
  const double gabs = sqrt(g2);
  const double er = gr / gabs, ei = gi / gabs;  // e^{i phi}
  const double zeta = (b - a) / (2.0 * gabs);
  const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
  // x' = cs x - sn e^{i phi} y ;  y' = sn x + cs e^{i phi} y
  auto rot = [&](zc* u, zc* v, int len) {
    for (int c = threadIdx.x; c < len; c += 256) {
      const zc uv = u[c], vv = v[c];
      const double pr = er * vv.x - ei * vv.y, pi = er * vv.y + ei * vv.x;  // e^{i phi} v
      u[c] = make_double2(cs * uv.x - sn * pr, cs * uv.y - sn * pi);
      v[c] = make_double2(sn * uv.x + cs * pr, sn * uv.y + cs * pi);
    }
  };
  rot(x, y, ncol);
  if (W) rot(W + (size_t)p * nrow, W + (size_t)q * nrow, nrow);
}

// row norms -> s[i]
__global__ __launch_bounds__(256) void k_row_norms(const zc* __restrict__ M, int ncol, double* __restrict__ s) {
  __shared__ double sh[5];
  const zc* x = M + (size_t)blockIdx.x * ncol;
  double a = 0;
  for (int c = threadIdx.x; c < ncol; c += 256) {
    const zc v = x[c];
    a += v.x * v.x + v.y * v.y;
  }
  a = svd_block_sum(a, sh);
  if (threadIdx.x == 0) s[blockIdx.x] = sqrt(a);
}

// Vh[k][:] = M[idx[k]][:] / s[idx[k]] ;  U[r][k] = conj(W[idx[k]][r])
__global__ __launch_bounds__(256) void k_svd_gather(const zc* __restrict__ M, const zc* __restrict__ W,
                                                    const int* __restrict__ idx, const double* __restrict__ s, int nrow,
                                                    int ncol, zc* __restrict__ U, zc* __restrict__ Vh) {
  const int k = blockIdx.x;
  const int src = idx[k];
  const double sv = s[src];
  const double inv = sv > 0.0 ? 1.0 / sv : 0.0;
  for (int c = threadIdx.x; c < ncol; c += 256) {
    const zc v = M[(size_t)src * ncol + c];
    Vh[(size_t)k * ncol + c] = make_double2(v.x * inv, v.y * inv);
  }
  for (int r = threadIdx.x; r < nrow; r += 256) {
    const zc w = W[(size_t)src * nrow + r];
    U[(size_t)r * nrow + k] = make_double2(w.x, -w.y);
  }
}

// Orthogonalise the rows of M (nr x nc) in place by Jacobi rotations (optionally accumulating
// them in W); returns the number of sweeps.  off_dev: one device word for the convergence flag.
static int jacobi_rows(hipStream_t st, zc* M, zc* W, int nr, int nc, unsigned long long* off_dev, double* s_dev) {
  const int np = nr + (nr & 1);
  int sweeps = 0;
  if (nr <= 1) return 0;
  // scale of the input: the largest row norm (rotations only move norm between rows, they never create it)
  hipLaunchKernelGGL(k_row_norms, dim3(nr), dim3(256), 0, st, M, nc, s_dev);
  std::vector<double> s0(nr);
  HIP_CHECK(hipMemcpyAsync(s0.data(), s_dev, nr * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  double smax = 0.0;
  for (double v : s0) smax = std::max(smax, v);
  const double tiny2 = 1e-28 * smax * smax;
  for (; sweeps < 60; ++sweeps) {
    HIP_CHECK(hipMemsetAsync(off_dev, 0, sizeof(unsigned long long), st));
    for (int round = 0; round < np - 1; ++round)
      hipLaunchKernelGGL(k_jacobi_step, dim3(np / 2), dim3(256), 0, st, M, W, nr, nc, np, round, off_dev, tiny2);
    HIP_CHECK(hipGetLastError());
    unsigned long long bits = 0;
    HIP_CHECK(hipMemcpyAsync(&bits, off_dev, sizeof(bits), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    double off;
    static_assert(sizeof(double) == sizeof(unsigned long long), "bit cast");
    std::memcpy(&off, &bits, sizeof(off));
    if (off <= 1e-30) return sweeps + 1;  // every |<x,y>| <= 1e-15 ||x|| ||y||
  }
  throw NotConverged("Jacobi SVD did not converge in 60 sweeps");
}

// Rows of M -> mutually orthogonal rows s_i q_i (the "U S" factor of M^T = Q^T S W^*), their
// norms (host, descending) and the sorting permutation (device idx, nr ints).  This is all a
// rank truncation "keep U S of the leading singular values" needs (Kraus maps, kraus.py:195-207).
void svd_rows_us(hipStream_t st, zc* M, int nr, int nc, double* S_host, int* idx_dev, zc* work, int* sweeps_out) {
  double* s_dev = reinterpret_cast<double*>(work);
  unsigned long long* off_dev = reinterpret_cast<unsigned long long*>(work + (nr + 1) / 2 + 1);
  const int sw = jacobi_rows(st, M, nullptr, nr, nc, off_dev, s_dev);
  hipLaunchKernelGGL(k_row_norms, dim3(nr), dim3(256), 0, st, M, nc, s_dev);
  std::vector<double> s(nr);
  HIP_CHECK(hipMemcpyAsync(s.data(), s_dev, nr * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  std::vector<int> idx(nr);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return s[a] > s[b]; });
  HIP_CHECK(hipMemcpyAsync(idx_dev, idx.data(), nr * sizeof(int), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));
  for (int k = 0; k < nr; ++k) S_host[k] = s[idx[k]];
  if (sweeps_out) *sweeps_out = sw;
}

size_t svd_work_elems(int r, int c) {
  const int nr = std::min(r, c), nc = std::max(r, c);
  return (size_t)nr * nc + (size_t)nr * nr + nr /*s*/ + nr /*idx*/ + 8 + (size_t)nr * nr + (size_t)nr * nc;
}

// A (r x c, row-major) = U (r x k) diag(S) Vh (k x c), k = min(r, c); S descending (host array).
void svd_jacobi(hipStream_t st, const zc* A, int r, int c, zc* U, double* S_host, zc* Vh, zc* work, int* sweeps_out) {
  if (r < 1 || c < 1) throw ArgError("svd: bad shape");
  const bool tr = r > c;  // work on the transpose so that the rotated vectors are the (fewer) rows
  const int nr = tr ? c : r, nc = tr ? r : c;
  zc* M = work;
  zc* W = M + (size_t)nr * nc;
  double* s_dev = reinterpret_cast<double*>(W + (size_t)nr * nr);
  int* idx_dev = reinterpret_cast<int*>(W + (size_t)nr * nr + nr);
  unsigned long long* off_dev = reinterpret_cast<unsigned long long*>(W + (size_t)nr * nr + 2 * (size_t)nr);
  zc* Ut = W + (size_t)nr * nr + 2 * (size_t)nr + 8;  // (nr x nr)
  zc* Vt = Ut + (size_t)nr * nr;                       // (nr x nc)
  if (tr) transpose_batched(st, A, M, r, c, c, r, 1, 0, 0);
  else HIP_CHECK(hipMemcpyAsync(M, A, (size_t)r * c * sizeof(zc), hipMemcpyDeviceToDevice, st));
  set_identity(st, W, nr, nr, nr);
  const int sweeps = jacobi_rows(st, M, W, nr, nc, off_dev, s_dev);
  hipLaunchKernelGGL(k_row_norms, dim3(nr), dim3(256), 0, st, M, nc, s_dev);
  std::vector<double> s(nr);
  HIP_CHECK(hipMemcpyAsync(s.data(), s_dev, nr * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  std::vector<int> idx(nr);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return s[a] > s[b]; });
  HIP_CHECK(hipMemcpyAsync(idx_dev, idx.data(), nr * sizeof(int), hipMemcpyHostToDevice, st));
  for (int k = 0; k < nr; ++k) S_host[k] = s[idx[k]];
  if (!tr) {
    hipLaunchKernelGGL(k_svd_gather, dim3(nr), dim3(256), 0, st, M, W, idx_dev, s_dev, nr, nc, U, Vh);
  } else {
    // A^T = Ut S Vt  =>  A = Vt^T S Ut^T
    hipLaunchKernelGGL(k_svd_gather, dim3(nr), dim3(256), 0, st, M, W, idx_dev, s_dev, nr, nc, Ut, Vt);
    transpose_batched(st, Vt, U, nr, nc, nc, nr, 1, 0, 0);   // U  (r x k) = Vt^T
    transpose_batched(st, Ut, Vh, nr, nr, nr, nr, 1, 0, 0);  // Vh (k x c) = Ut^T
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));  // idx / s host vectors go out of scope
  if (sweeps_out) *sweeps_out = sweeps;
}

}  // namespace mitdvp
