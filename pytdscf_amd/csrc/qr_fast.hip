// qr_fast.hip -- panel factorisation of the blocked Householder QR (qr.hip) WITHOUT one launch per column:
// the 32-column panel is orthogonalised by CholeskyQR2 (two rounds of Gram matrix -> Cholesky factor -> triangular
// solve: O(m) work in a handful of launches whose width is the whole panel), and LAPACK's Householder representation
// of exactly that factorisation -- the unit-lower vectors V, the compact-WY factor T, tau and the REAL diagonal of R
// with zlarfg's signs -- is then RECONSTRUCTED from the orthonormal panel by an LU factorisation with on-the-fly sign
// choices (Ballard, Demmel, Grigori, Jacquelin, Nguyen, Solomonik, "Reconstructing Householder vectors from
// Tall-Skinny QR", IPDPS 2014; complex form: the diagonal of R stays real, so the per-column factor is +-1,
// D_j = -sign(Re pivot_j)).  Everything downstream of a panel (zlarfb trailing update, zungqr, the extended Q of the
// adaptive sweep) sees the same V / T / tau / R as after zgeqr2, to rounding: the reference's gauge move
// (SiteCoef.gauge_trf, _site_cls.py:138-292 -> scipy.linalg.qr -> zgeqrf + zungqr) is reproduced including diag(R)'s
// signs and the order of the orthogonal complement.
//
// CholeskyQR2 needs cond(panel) well below eps^-1/2; the zero-padded, rank-deficient states the reference starts from
// are not that.  Every Cholesky pivot and every LU pivot is checked on the device; a failed check raises a sticky flag
// and the caller redoes the factorisation with the per-column Householder kernels (qr.hip), which are unconditionally
// stable.  One flag read per QR, no host round trip per panel.
#include <cstdlib>

#include "qr.h"
#include "vecops.h"

namespace mitdvp {

namespace {
constexpr int NB = QR_NB;          // 32
constexpr int FQ_ROWS = 128;       // panel rows per workgroup of the tall-skinny kernels
constexpr double CHOL_TOL = 1e-11;  // pivot / diagonal below this: the panel is too ill-conditioned for CholeskyQR2

__device__ __forceinline__ zc cmul(zc a, zc b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ zc cmulc(zc a, zc b) {  // conj(a) * b
  return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ zc cadd(zc a, zc b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ zc csub(zc a, zc b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ zc cdiv(zc a, zc b) {
  const double s = 1.0 / (b.x * b.x + b.y * b.y);
  return make_double2((a.x * b.x + a.y * b.y) * s, (a.y * b.x - a.x * b.y) * s);
}
}  // namespace

// partial Gram matrices of a tall panel: part[blk][i][j] = sum_{rows of blk} conj(P[r][i]) P[r][j]
// (P = mp x nb, leading dimension ld; the rows of a workgroup are summed in a fixed order: deterministic)
__global__ __launch_bounds__(256) void k_fq_gram(const zc* __restrict__ P, long ld, int mp, int nb, zc* __restrict__ part) {
  __shared__ zc tile[32][NB + 1];
  const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
  const long r0 = (long)blockIdx.x * FQ_ROWS;
  const int nr = (int)min((long)FQ_ROWS, mp - r0);
  zc acc[2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};
  for (int base = 0; base < nr; base += 32) {
    const int rows = min(32, nr - base);
    __syncthreads();
    for (int e = t; e < 32 * NB; e += 256) {
      const int r = e >> 5, c = e & 31;
      tile[r][c] = (r < rows && c < nb) ? P[(r0 + base + r) * ld + c] : make_double2(0.0, 0.0);
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 32; ++k) {
      const zc a0 = tile[k][ti], a1 = tile[k][ti + 16], b0 = tile[k][tj], b1 = tile[k][tj + 16];
      acc[0][0] = cadd(acc[0][0], cmulc(a0, b0));
      acc[0][1] = cadd(acc[0][1], cmulc(a0, b1));
      acc[1][0] = cadd(acc[1][0], cmulc(a1, b0));
      acc[1][1] = cadd(acc[1][1], cmulc(a1, b1));
    }
  }
  zc* o = part + (size_t)blockIdx.x * NB * NB;
  o[ti * NB + tj] = acc[0][0];
  o[ti * NB + tj + 16] = acc[0][1];
  o[(ti + 16) * NB + tj] = acc[1][0];
  o[(ti + 16) * NB + tj + 16] = acc[1][1];
}

// G = sum of the partials; G = R^H R (upper Cholesky factor, real positive diagonal); Rinv = R^-1.
// second != 0: additionally Rtot = R * Rprev (the panel's triangular factor after both rounds).
// A pivot that is not clearly positive raises *flag (sticky) and leaves the identity behind.
__global__ __launch_bounds__(1024) void k_fq_chol(const zc* __restrict__ part, int nblk, int nb, zc* __restrict__ Rinv_out,
                                                  const zc* __restrict__ Rprev, zc* __restrict__ Rtot, int second,
                                                  int* __restrict__ flag) {
  __shared__ zc G[NB][NB + 1];
  __shared__ zc R[NB][NB + 1];
  __shared__ zc Ri[NB][NB + 1];
  __shared__ double d0[NB];
  __shared__ int bad;
  const int i = threadIdx.x >> 5, j = threadIdx.x & 31;
  zc g = make_double2(0.0, 0.0);
  if (i < nb && j < nb)
    for (int b = 0; b < nblk; ++b) g = cadd(g, part[(size_t)b * NB * NB + i * NB + j]);
  G[i][j] = g;
  R[i][j] = make_double2(0.0, 0.0);
  Ri[i][j] = make_double2(0.0, 0.0);
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  if (i == j) d0[i] = G[i][i].x;
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    const double d = G[k][k].x;
    const bool ok = second ? (d > 0.5 && d < 2.0) : (d > CHOL_TOL * d0[k] && d0[k] > 0.0);
    if (!ok) {
      if (threadIdx.x == 0) bad = 1;
      break;  // uniform: every thread reads the same G[k][k]
    }
    const double r = sqrt(d), ir = 1.0 / r;
    __syncthreads();
    if (i == k && j >= k && j < nb) R[k][j] = j == k ? make_double2(r, 0.0) : make_double2(G[k][j].x * ir, G[k][j].y * ir);
    __syncthreads();
    if (i > k && j > k && i < nb && j < nb) G[i][j] = csub(G[i][j], cmulc(R[k][i], R[k][j]));
    __syncthreads();
  }
  __syncthreads();
  if (bad) {
    if (threadIdx.x == 0) atomicExch(flag, 1);
    Rinv_out[i * NB + j] = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
    if (second) Rtot[i * NB + j] = Rprev[i * NB + j];
    return;
  }
  // inverse of the upper-triangular factor, column j by back substitution (thread (0, j) walks its column)
  if (i == 0 && j < nb) {
    Ri[j][j] = make_double2(1.0 / R[j][j].x, 0.0);
    for (int r = j - 1; r >= 0; --r) {
      zc s = make_double2(0.0, 0.0);
      for (int k = r + 1; k <= j; ++k) s = cadd(s, cmul(R[r][k], Ri[k][j]));
      const double ir = -1.0 / R[r][r].x;
      Ri[r][j] = make_double2(s.x * ir, s.y * ir);
    }
  }
  __syncthreads();
  Rinv_out[i * NB + j] = (i < nb && j < nb) ? Ri[i][j] : (i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
  if (second) {
    zc s = make_double2(0.0, 0.0);
    if (i < nb && j < nb)
      for (int k = i; k <= j; ++k) s = cadd(s, cmul(R[i][k], Rprev[k * NB + j]));
    Rtot[i * NB + j] = s;
  } else if (Rtot) {
    Rtot[i * NB + j] = (i < nb && j < nb) ? R[i][j] : make_double2(0.0, 0.0);
  }
}

// out[r][:] = in[r][:] * M  for the rows [row_off, mp) of a tall panel (M: nb x nb in a 32 x 32 buffer);
// out2 (nullable) receives the same values with its own leading dimension
__global__ __launch_bounds__(256) void k_fq_apply(const zc* __restrict__ in, long ldi, int mp, int nb, const zc* __restrict__ M,
                                                  zc* __restrict__ out, long ldo, zc* __restrict__ out2, long ldo2, int row_off) {
  __shared__ zc Ms[NB][NB + 1];
  __shared__ zc tile[8][NB + 1];
  const int t = threadIdx.x, tr = t >> 5, tc = t & 31;
  for (int e = t; e < NB * NB; e += 256) Ms[e >> 5][e & 31] = M[e];
  const long r0 = row_off + (long)blockIdx.x * FQ_ROWS;
  const int nr = (int)min((long)FQ_ROWS, mp - r0);
  for (int base = 0; base < nr; base += 8) {
    const long r = r0 + base + tr;
    const bool live = base + tr < nr;
    __syncthreads();
    tile[tr][tc] = (live && tc < nb) ? in[r * ldi + tc] : make_double2(0.0, 0.0);
    __syncthreads();
    zc s = make_double2(0.0, 0.0);
#pragma unroll 8
    for (int k = 0; k < NB; ++k) s = cadd(s, cmul(tile[tr][k], Ms[k][tc]));
    if (live && tc < nb) {
      out[r * ldo + tc] = s;
      if (out2) out2[r * ldo2 + tc] = s;
    }
  }
}

// Householder reconstruction of one panel from its orthonormal factor (top nb x nb block Qt of Q, row stride ldq) and
// its triangular factor Rtot (positive diagonal): LU of (Q - [D; 0]) with D_j = -sign(Re pivot_j) chosen when column j
// is eliminated.  Writes, in LAPACK's layout, into the panel of A: R' = D Rtot on and above the diagonal, the unit-lower
// V1 = L below it; Vp's top block (unit diagonal, zeros above); T (compact WY: T V1^H = -U D); tau = diag(T); Uinv
// (for V2 = Q2 U^-1).  An LU pivot smaller than 1/2 (it is >= 1 for an orthonormal Q) raises *flag.
__global__ __launch_bounds__(1024) void k_fq_reconstruct(const zc* __restrict__ Qt, long ldq, const zc* __restrict__ Rtot, int nb,
                                                         zc* __restrict__ A, long lda, zc* __restrict__ Vp, zc* __restrict__ T,
                                                         zc* __restrict__ tau, zc* __restrict__ Uinv, int* __restrict__ flag) {
  __shared__ zc Q[NB][NB + 1];
  __shared__ zc Y[NB][NB + 1];
  __shared__ zc Ui[NB][NB + 1];
  __shared__ zc Ts[NB][NB + 1];
  __shared__ double D[NB];
  __shared__ int bad;
  const int i = threadIdx.x >> 5, j = threadIdx.x & 31;
  Q[i][j] = (i < nb && j < nb) ? Qt[(long)i * ldq + j] : make_double2(0.0, 0.0);
  Y[i][j] = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
  Ui[i][j] = make_double2(0.0, 0.0);
  Ts[i][j] = make_double2(0.0, 0.0);
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  for (int k = 0; k < nb; ++k) {
    if (threadIdx.x == 0) {
      const zc piv = Q[k][k];
      const double dk = piv.x >= 0.0 ? -1.0 : 1.0;  // zlarfg: beta = -sign(Re alpha) |x|, sign(0) = +
      D[k] = dk;
      const zc u = make_double2(piv.x - dk, piv.y);
      Q[k][k] = u;
      if (!(u.x * u.x + u.y * u.y > 0.25)) bad = 1;
    }
    __syncthreads();
    if (bad) break;
    if (j == k && i > k && i < nb) Y[i][k] = cdiv(Q[i][k], Q[k][k]);
    __syncthreads();
    if (i > k && j > k && i < nb && j < nb) Q[i][j] = csub(Q[i][j], cmul(Y[i][k], Q[k][j]));
    __syncthreads();
  }
  __syncthreads();
  if (bad) {
    if (threadIdx.x == 0) atomicExch(flag, 1);
    return;
  }
  // U = upper triangle of Q now; its inverse by back substitution (thread (0, j): column j)
  if (i == 0 && j < nb) {
    Ui[j][j] = cdiv(make_double2(1.0, 0.0), Q[j][j]);
    for (int r = j - 1; r >= 0; --r) {
      zc s = make_double2(0.0, 0.0);
      for (int k = r + 1; k <= j; ++k) s = cadd(s, cmul(Q[r][k], Ui[k][j]));
      Ui[r][j] = cdiv(make_double2(-s.x, -s.y), Q[r][r]);
    }
  }
  // T V1^H = -U D: row i of T left to right (thread (i, 0))
  if (j == 0 && i < nb) {
    for (int c = i; c < nb; ++c) {
      zc s = make_double2(-Q[i][c].x * D[c], -Q[i][c].y * D[c]);
      for (int k = i; k < c; ++k) s = csub(s, cmul(Ts[i][k], make_double2(Y[c][k].x, -Y[c][k].y)));
      Ts[i][c] = s;
    }
  }
  __syncthreads();
  T[i * NB + j] = Ts[i][j];
  Uinv[i * NB + j] = (i < nb && j < nb) ? Ui[i][j] : (i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
  if (i == 0 && j < nb) tau[j] = Ts[j][j];
  if (i < nb && j < nb) {
    const zc r = Rtot[i * NB + j];
    A[(long)i * lda + j] = j >= i ? make_double2(D[i] * r.x, D[i] * r.y) : Y[i][j];
    Vp[i * nb + j] = j > i ? make_double2(0.0, 0.0) : Y[i][j];
  }
}

size_t qr_fast_work_elems(int m, int n) {
  const size_t nblk = ((size_t)m + FQ_ROWS - 1) / FQ_ROWS;
  return (size_t)m * n            // copy of the input (the factorisation is redone from it when a check fails)
         + 2 * (size_t)m * NB     // Q1, Q2: the panel after the first / second round
         + nblk * NB * NB         // partial Gram matrices
         + 5 * (size_t)NB * NB    // R1^-1, R2^-1, R1, R2 R1, U^-1
         + 8;                     // flag
}

// One panel: A[j0:m, j0:j0+nbp] -> R' / V in place (LAPACK layout), Vp (mp x nbp unit lower trapezoid, contiguous),
// Tp, tau[j0..].  `ws` = the part of the workspace behind the input copy (see qr_fast_work_elems).
int qr_fast_panel(hipStream_t st, zc* A, long lda, int m, int j0, int nbp, zc* Vp, zc* Tp, zc* tau, zc* ws, int* flag) {
  const int mp = m - j0;
  zc* P = A + (long)j0 * lda + j0;
  zc* Q1 = ws;
  zc* Q2 = Q1 + (size_t)m * NB;
  const int nblk = (mp + FQ_ROWS - 1) / FQ_ROWS;
  zc* part = Q2 + (size_t)m * NB;
  zc* small = part + (size_t)((m + FQ_ROWS - 1) / FQ_ROWS) * NB * NB;
  zc* R1inv = small;
  zc* R2inv = small + NB * NB;
  zc* R1 = small + 2 * NB * NB;
  zc* Rtot = small + 3 * NB * NB;
  zc* Uinv = small + 4 * NB * NB;
  // round 1
  hipLaunchKernelGGL(k_fq_gram, dim3(nblk), dim3(256), 0, st, P, lda, mp, nbp, part);
  hipLaunchKernelGGL(k_fq_chol, dim3(1), dim3(1024), 0, st, part, nblk, nbp, R1inv, (const zc*)nullptr, R1, 0, flag);
  hipLaunchKernelGGL(k_fq_apply, dim3(nblk), dim3(256), 0, st, P, lda, mp, nbp, R1inv, Q1, (long)NB, (zc*)nullptr, 0L, 0);
  // round 2
  hipLaunchKernelGGL(k_fq_gram, dim3(nblk), dim3(256), 0, st, Q1, (long)NB, mp, nbp, part);
  hipLaunchKernelGGL(k_fq_chol, dim3(1), dim3(1024), 0, st, part, nblk, nbp, R2inv, R1, Rtot, 1, flag);
  hipLaunchKernelGGL(k_fq_apply, dim3(nblk), dim3(256), 0, st, Q1, (long)NB, mp, nbp, R2inv, Q2, (long)NB, (zc*)nullptr, 0L, 0);
  // Householder reconstruction: top block, then V2 = Q2[nbp:] U^-1 into the panel and into Vp
  hipLaunchKernelGGL(k_fq_reconstruct, dim3(1), dim3(1024), 0, st, Q2, (long)NB, Rtot, nbp, P, lda, Vp, Tp, tau + j0, Uinv, flag);
  int nl = 7;
  if (mp > nbp) {
    const int nb2 = (mp - nbp + FQ_ROWS - 1) / FQ_ROWS;
    hipLaunchKernelGGL(k_fq_apply, dim3(nb2), dim3(256), 0, st, Q2, (long)NB, mp, nbp, Uinv, P, lda, Vp, (long)nbp, nbp);
    ++nl;
  }
  HIP_CHECK(hipGetLastError());
  return nl;
}

}  // namespace mitdvp
