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
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "qr.h"
#include "vecops.h"

namespace mitdvp {

namespace {
constexpr int NB = QR_NB;          // 32
constexpr int FQ_NQ = 1;             // matrix elements per thread of the 32 x 32 kernels (1024 / FQ_NQ threads)
constexpr int FQ_RS = NB / FQ_NQ;     // row stride between a thread's elements
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
__global__ __launch_bounds__(256) void k_fq_gram(const zc* __restrict__ P, long ld, int mp, int nb, zc* __restrict__ part,
                                                 int rows_per_blk) {
  __shared__ zc tile[32][NB + 1];
  const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
  const long r0 = (long)blockIdx.x * rows_per_blk;
  const int nr = (int)min((long)rows_per_blk, mp - r0);
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
// second != 0: additionally Rtot = R * Rprev (the panel's triangular factor after both rounds), else Rtot = R.
// One workgroup, FQ_NQ matrix elements per thread.  The 32 pivots are a chain of rank-1 updates in LDS
// with ONE barrier per step (row k is used unscaled, G[k][i]* G[k][j] / G[k][k], and scaled once at the end); the
// inverse is the same chain run backwards on [R | I] (Gauss-Jordan), again one barrier per step.  (Measured on the
// way here: a single wave holding the matrix in registers, 94 us per call -- ~12 000 dependent instructions; one
// element per thread with three barriers per step, 45 us -- a 16-wave barrier costs ~0.3 us.)
// A pivot that is not clearly positive raises *flag (sticky) and leaves the identity behind.
__global__ __launch_bounds__(1024 / FQ_NQ) void k_fq_chol(const zc* __restrict__ part, int nblk, int nb, zc* __restrict__ Rinv_out,
                                                 const zc* __restrict__ Rprev, zc* __restrict__ Rtot, int second,
                                                 int* __restrict__ flag) {
  __shared__ zc G[NB][NB + 1];   // Gram matrix -> R (upper)
  __shared__ zc W[NB][NB + 1];   // inverse being built
  __shared__ zc P1[NB][NB + 1];  // Rprev (a global load inside the product loop below costs a memory latency per term)
  __shared__ double d0[NB];
  const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;  // my elements: (i0 + 8 q, j), q = 0..3
#pragma unroll
  for (int q = 0; q < FQ_NQ; ++q) {
    const int i = i0 + FQ_RS * q;
    P1[i][j] = second ? Rprev[i * NB + j] : make_double2(0.0, 0.0);
    zc g = make_double2(0.0, 0.0);
    if (i < nb && j < nb) {
      const zc* pp = part + i * NB + j;
      int b = 0;
      for (; b + 8 <= nblk; b += 8) {  // eight loads in flight (one exposed memory latency per batch, not per partial)
        zc v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pp[(size_t)(b + u) * NB * NB];
#pragma unroll
        for (int u = 0; u < 8; ++u) g = cadd(g, v[u]);
      }
      for (; b < nblk; ++b) g = cadd(g, pp[(size_t)b * NB * NB]);
    }
    if (i >= nb || j >= nb) g = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);  // pad with the identity
    if (i == j) g.y = 0.0;
    G[i][j] = g;
    W[i][j] = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
    if (i == j) d0[i] = g.x;
  }
  __syncthreads();
  if (second) {
    // After the first round the panel is orthonormal to eps cond^2: G = I + E with a tiny E, and to first order
    // chol(I + E) = I + U, U = triu(E, 1) + diag(E) / 2, (I + U)^-1 = I - U; the neglected terms are O(|E|^2), i.e.
    // below rounding when |E| < 1e-8 -- no chain of 32 pivots, no inverse.  Larger E: the full factorisation below.
    __shared__ double emax_s[16];
    double e = 0.0;
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      const zc g = G[i][j];
      e = fmax(e, fmax(fabs(g.x - (i == j ? 1.0 : 0.0)), fabs(g.y)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = fmax(e, __shfl_xor(e, o, 64));
    if ((threadIdx.x & 63) == 0) emax_s[threadIdx.x >> 6] = e;
    __syncthreads();
    double emax = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) emax = fmax(emax, emax_s[w]);
    if (emax < 1e-8) {
#pragma unroll
      for (int q = 0; q < FQ_NQ; ++q) {
        const int i = i0 + FQ_RS * q;
        const zc g = G[i][j];
        const zc u = j > i ? g : (j == i ? make_double2(0.5 * (g.x - 1.0), 0.0) : make_double2(0.0, 0.0));
        W[i][j] = u;  // U
        Rinv_out[i * NB + j] = make_double2((i == j ? 1.0 : 0.0) - u.x, -u.y);
      }
      __syncthreads();
      if (Rtot) {
#pragma unroll
        for (int q = 0; q < FQ_NQ; ++q) {
          const int i = i0 + FQ_RS * q;
          zc t = P1[i][j];  // (I + U) Rprev
          for (int k = i; k <= j; ++k) t = cadd(t, cmul(W[i][k], P1[k][j]));
          Rtot[i * NB + j] = t;
        }
      }
      return;
    }
    __syncthreads();
  }
  bool bad = false;
  for (int k = 0; k < NB; ++k) {
    const double d = G[k][k].x;
    const bool ok = second ? (d > 0.5 && d < 2.0) : (d > CHOL_TOL * d0[k] && d0[k] > 0.0);
    if (!ok) { bad = true; break; }  // uniform: every thread read the same word
    const double id = fast_rcp(d);
    const zc gkj = G[k][j];
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      if (i > k && j >= i) {
        const zc gki = G[k][i];
        G[i][j] = csub(G[i][j], cmulc(gki, make_double2(gkj.x * id, gkj.y * id)));
      }
    }
    __syncthreads();
  }
  if (bad) {
    if (threadIdx.x == 0) atomicExch(flag, 1);
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      Rinv_out[i * NB + j] = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
      if (Rtot) Rtot[i * NB + j] = second ? P1[i][j] : (i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
    }
    return;
  }
  // rows scaled: R[k][j] = G[k][j] / sqrt(G[k][k])
  {
    zc r[FQ_NQ];
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      const double dd = G[i][i].x;
      const double is = 1.0 / sqrt(dd);
      r[q] = j > i ? make_double2(G[i][j].x * is, G[i][j].y * is) : (j == i ? make_double2(dd * is, 0.0) : make_double2(0.0, 0.0));
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) G[i0 + FQ_RS * q][j] = r[q];
    __syncthreads();
    if (threadIdx.x < NB) d0[threadIdx.x] = 1.0 / G[threadIdx.x][threadIdx.x].x;  // 1 / R[k][k] for the chain below
    __syncthreads();
  }
  if (Rtot) {
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      zc t = G[i][j];
      if (second) {
        t = make_double2(0.0, 0.0);
        for (int k = i; k <= j; ++k) t = cadd(t, cmul(G[i][k], P1[k][j]));
      }
      Rtot[i * NB + j] = t;
    }
  }
  // W <- R^-1 by Gauss-Jordan on [R | I], rows from the bottom up; row k is final when its turn comes and enters
  // scaled by 1 / R[k][k] on the fly (rows are scaled once at the end)
  for (int k = NB - 1; k > 0; --k) {
    const double ir = d0[k];
    const zc wkj = make_double2(W[k][j].x * ir, W[k][j].y * ir);
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      if (i < k && j >= k) W[i][j] = csub(W[i][j], cmul(G[i][k], wkj));
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < FQ_NQ; ++q) {
    const int i = i0 + FQ_RS * q;
    const double ir = d0[i];
    Rinv_out[i * NB + j] = make_double2(W[i][j].x * ir, W[i][j].y * ir);
  }
}

// out[r][:] = in[r][:] * M  for the rows [row_off, mp) of a tall panel (M: nb x nb in a 32 x 32 buffer);
// out2 (nullable) receives the same values with its own leading dimension
__global__ __launch_bounds__(256) void k_fq_apply(const zc* __restrict__ in, long ldi, int mp, int nb, const zc* __restrict__ M,
                                                  zc* __restrict__ out, long ldo, zc* __restrict__ out2, long ldo2, int row_off,
                                                  int rows_per_blk) {
  __shared__ zc Ms[NB][NB + 1];
  __shared__ zc tile[8][NB + 1];
  const int t = threadIdx.x, tr = t >> 5, tc = t & 31;
  for (int e = t; e < NB * NB; e += 256) Ms[e >> 5][e & 31] = M[e];
  const long r0 = row_off + (long)blockIdx.x * rows_per_blk;
  const int nr = (int)min((long)rows_per_blk, mp - r0);
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
// V1 = L below it; Vp's top block (unit diagonal, zeros above); T (compact WY: T = -U D V1^-H, leading dimension nb);
// tau = diag(T); Uinv (for V2 = Q2 U^-1).  An LU pivot smaller than 1/2 (it is >= 1 for an orthonormal Q) raises *flag.
// One workgroup, FQ_NQ elements per thread, one barrier per step (see k_fq_chol): 32 elimination steps
// (multipliers formed on the fly, column k scaled at the end), then U^-1 and V1^-H together in one backward
// Gauss-Jordan chain, then the 32 x 32 product for T.
__global__ __launch_bounds__(1024 / FQ_NQ) void k_fq_reconstruct(const zc* __restrict__ Qt, long ldq, const zc* __restrict__ Rtot, int nb,
                                                        zc* __restrict__ A, long lda, zc* __restrict__ Vp, zc* __restrict__ T,
                                                        zc* __restrict__ tau, zc* __restrict__ Uinv, int* __restrict__ flag) {
  __shared__ zc Q[NB][NB + 1];    // -> L (strictly lower, unscaled until the end) and U (strictly upper); diagonal: pivots
  __shared__ zc Wu[NB][NB + 1];   // U^-1
  __shared__ zc Wy[NB][NB + 1];   // (V1^H)^-1, V1^H unit upper triangular
  __shared__ zc Ud[NB];           // diagonal of U
  __shared__ double D[NB];
  const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;
#pragma unroll
  for (int q = 0; q < FQ_NQ; ++q) {
    const int i = i0 + FQ_RS * q;
    Q[i][j] = (i < nb && j < nb) ? Qt[(long)i * ldq + j] : (i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
    Wu[i][j] = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
    Wy[i][j] = Wu[i][j];
  }
  __syncthreads();
  bool bad = false;
  for (int k = 0; k < NB; ++k) {
    const zc piv = Q[k][k];
    const double dk = piv.x >= 0.0 ? -1.0 : 1.0;  // zlarfg: beta = -sign(Re alpha) |x|, sign(0) = +
    const zc u = make_double2(piv.x - dk, piv.y);
    const double un = u.x * u.x + u.y * u.y;
    if (!(un > 0.25)) { bad = true; break; }  // uniform
    if (threadIdx.x == 0) { D[k] = dk; Ud[k] = u; }
    const double iun = fast_rcp(un);
    const zc iu = make_double2(u.x * iun, -u.y * iun);
    const zc qkj = Q[k][j];
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      if (i > k && j > k) Q[i][j] = csub(Q[i][j], cmul(cmul(Q[i][k], iu), qkj));
    }
    __syncthreads();
  }
  if (bad) {
    if (threadIdx.x == 0) atomicExch(flag, 1);
    return;
  }
  {  // column k of L scaled by 1 / U[k][k]; the diagonal takes U's
    zc v[FQ_NQ];
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      const zc ud = Ud[j];
      const double un = ud.x * ud.x + ud.y * ud.y;
      v[q] = i > j ? cmul(Q[i][j], make_double2(ud.x / un, -ud.y / un)) : (i == j ? ud : Q[i][j]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) Q[i0 + FQ_RS * q][j] = v[q];
    if (threadIdx.x < NB) {  // Ud <- 1 / U[k][k] for the chain below
      const zc ud = Ud[threadIdx.x];
      const double un = ud.x * ud.x + ud.y * ud.y;
      Ud[threadIdx.x] = make_double2(ud.x / un, -ud.y / un);
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < FQ_NQ; ++q) {
    const int i = i0 + FQ_RS * q;
    if (i < nb && j < nb) {
      const zc r = Rtot[i * NB + j];
      A[(long)i * lda + j] = j >= i ? make_double2(D[i] * r.x, D[i] * r.y) : Q[i][j];
      Vp[i * nb + j] = i > j ? Q[i][j] : (i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
    }
  }
  // Wu <- U^-1 and Wy <- (V1^H)^-1 (V1^H[i][k] = conj(L[k][i]) for k > i, unit diagonal), rows from the bottom up
  for (int k = NB - 1; k > 0; --k) {
    const zc wkj = cmul(Wu[k][j], Ud[k]);
    const zc ykj = Wy[k][j];
#pragma unroll
    for (int q = 0; q < FQ_NQ; ++q) {
      const int i = i0 + FQ_RS * q;
      if (i < k && j >= k) {
        Wu[i][j] = csub(Wu[i][j], cmul(Q[i][k], wkj));
        const zc l = Q[k][i];  // L[k][i]
        Wy[i][j] = csub(Wy[i][j], cmul(make_double2(l.x, -l.y), ykj));
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < FQ_NQ; ++q) {
    const int i = i0 + FQ_RS * q;
    Uinv[i * NB + j] = cmul(Wu[i][j], Ud[i]);
    // T = -(U D) (V1^H)^-1
    zc t = make_double2(0.0, 0.0);
    if (j >= i)
      for (int k = i; k <= j; ++k) t = csub(t, cmul(make_double2(Q[i][k].x * D[k], Q[i][k].y * D[k]), Wy[k][j]));
    if (i < nb && j < nb) {
      T[i * nb + j] = t;
      if (i == j) tau[j] = t;
    }
  }
}

static int fq_rows_per_blk(int m) {  // at most 128 workgroups (= partial Gram matrices), at least 32 rows each
  int r = (m + 127) / 128;
  r = (r + 31) / 32 * 32;
  return r < 32 ? 32 : r;
}
// (Round 4, measured and dropped -- profiles/r04_qr_chain_ab.txt: fewer, larger Gram workgroups (sqrt(0.22 m) partials
// instead of m / 32) take 13 us off the first-round Cholesky kernel, whose one workgroup streams the partials, and put
// 12 us onto the two Gram launches, whose plain-FMA tiles cost ~5 us per 32 rows; the same chain kernels with 512
// threads and the triangular inverses by recursive doubling (the building blocks of k_qr_small_fast below) run the
// reconstruction in 27.0 instead of 23.7 us: a 31-step Gauss-Jordan chain builds both inverses at once.)

size_t qr_fast_work_elems(int m, int n) {
  const size_t nblk = 128;
  return (size_t)m * n            // copy of the input (the factorisation is redone from it when a check fails)
         + (size_t)m * n          // the V panels of all block reflectors (the Q formation reuses them)
         + 2 * (size_t)m * NB     // Q1, Q2: the panel after the first / second round
         + nblk * NB * NB         // partial Gram matrices
         + 5 * (size_t)NB * NB    // R1^-1, R2^-1, R1, R2 R1, U^-1
         + 8;                     // flag
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole factorisation of a SMALL matrix (m <= 320, n <= 32: the gauge moves of the small-bond regime, C2's 320 x 32)
// in ONE workgroup: CholeskyQR2 with the two Gram matrices and the two triangular applies on the matrix cores, and
// LAPACK's signs recovered by the LU chain of the reconstruction (only D is needed: Q' = Q D, R' = D R -- the caller of the
// thin factorisation never sees V, T or tau).  The per-column Householder kernel of qr.hip spends 4.3 us per column on
// two workgroup barriers and the reflector's scalar chain, 137 us for 320 x 32; here the only chains left are the 32
// Cholesky pivots and the 32 LU pivots on a 32 x 32 matrix in LDS (the inverse of the triangular factor is built by
// recursive doubling: 5 levels of small products instead of 31 substitution steps).
//   Gram:   wave (row quarter, half) accumulates three of the six real streams {00re, 00im, 01re | 01im, 11re, 11im} of
//           the 2 x 2 blocks of 16 x 16 with v_mfma_f64_16x16x4_f64 straight from global loads in operand layout
//           (lane: row 4 s + lane / 16, column lane % 16); the four quarter partials are summed in a fixed order.
//   apply:  each wave owns 16-row blocks (w, w + 8, w + 16); A-operand from global (lane: row lane % 16, column
//           4 s + lane / 16), B-operand = the triangular inverse from LDS, kept in registers over the wave's blocks.
// A failed conditioning check (same thresholds as the panels above) sets *fail and leaves Q / R untouched: the caller
// has queued the Householder kernel behind this one, which runs only then.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int SF_T = 512;
typedef double sf_d4 __attribute__((ext_vector_type(4)));

typedef zc SfMat[NB][NB + 1];
typedef double SfPart[4][6][4][64];
struct SfSmem {
  SfMat G;   // Gram matrix -> R (upper)
  SfMat W;   // R^-1
  SfMat P1;  // R1, then R2 R1
  SfMat Qt;  // scratch of the inverse; then the top block of Q for the LU chain
  SfPart pb;  // Gram partials [row quarter][stream][accumulator register][lane]
  double d0[NB];
  double Dg[NB];
  double emax[SF_T / 64];
};

__device__ __forceinline__ sf_d4 sf_mfma(double a, double b, sf_d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// the six real streams of the upper 2 x 2 blocks of src^H src, one partial per row quarter, left in pb (barrier included)
__device__ void sf_gram_partials(const zc* __restrict__ src, long ld, int m, int n, SfPart& pb) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
  const int rq = w >> 1, half = w & 1;
  const int nks = (m + 3) / 4, per = (nks + 3) / 4;
  const int s0 = rq * per, s1 = min(nks, s0 + per);
  sf_d4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (sf_d4){0.0, 0.0, 0.0, 0.0};
  for (int s = s0; s < s1; s += 4) {
    zc x0[4], x1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int row = 4 * (s + u) + lk;
      const bool ok = (s + u < s1) && row < m;
      x0[u] = (ok && li < n) ? src[(long)row * ld + li] : make_double2(0.0, 0.0);
      x1[u] = (ok && li + 16 < n) ? src[(long)row * ld + 16 + li] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // a diagonal block's imaginary part is P - P^T with P = Xr^T Xi: one product instead of two (the transpose is
      // taken when the partials are gathered)
      if (half == 0) {
        acc[0] = sf_mfma(x0[u].x, x0[u].x, acc[0]);  acc[0] = sf_mfma(x0[u].y, x0[u].y, acc[0]);   // 00 re
        acc[1] = sf_mfma(x0[u].x, x0[u].y, acc[1]);                                                // 00: P
        acc[2] = sf_mfma(x0[u].x, x1[u].x, acc[2]);  acc[2] = sf_mfma(x0[u].y, x1[u].y, acc[2]);   // 01 re
      } else {
        acc[0] = sf_mfma(x0[u].x, x1[u].y, acc[0]);  acc[0] = sf_mfma(-x0[u].y, x1[u].x, acc[0]);  // 01 im
        acc[1] = sf_mfma(x1[u].x, x1[u].x, acc[1]);  acc[1] = sf_mfma(x1[u].y, x1[u].y, acc[1]);   // 11 re
        acc[2] = sf_mfma(x1[u].x, x1[u].y, acc[2]);                                                // 11: P
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) pb[rq][half * 3 + t][r][l] = acc[t][r];
  __syncthreads();
}
// element (i, j), j >= i, of the Gram matrix from the quarter partials (summed in a fixed order)
__device__ __forceinline__ zc sf_gram_element(const SfPart& pb, int i, int j, int cd_mode) {
  const int blk = (i >> 4) + (j >> 4), ii = i & 15, jj = j & 15;
  const int lk_ = cd_mode == 0 ? (ii & 3) : (ii >> 2), r_ = cd_mode == 0 ? (ii >> 2) : (ii & 3);
  const int ln = lk_ * 16 + jj;
  // streams: 0 = 00 re, 1 = 00 P, 2 = 01 re, 3 = 01 im, 4 = 11 re, 5 = 11 P
  const int sre = blk == 0 ? 0 : (blk == 1 ? 2 : 4), sim = blk == 0 ? 1 : (blk == 1 ? 3 : 5);
  const int lkt = cd_mode == 0 ? (jj & 3) : (jj >> 2), rt = cd_mode == 0 ? (jj >> 2) : (jj & 3);
  const int lnt = lkt * 16 + ii;  // element (jj, ii) of the same block
  zc g = make_double2(0.0, 0.0);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    g.x += pb[p][sre][r_][ln];
    g.y += blk == 1 ? pb[p][sim][r_][ln] : pb[p][sim][r_][ln] - pb[p][sim][rt][lnt];
  }
  return g;
}
// G = src^H src (upper triangle), padded with the identity beyond n; d0 = diag
__device__ void sf_gram(const zc* __restrict__ src, long ld, int m, int n, SfSmem& S, int cd_mode) {
  sf_gram_partials(src, ld, m, n, S.pb);
  const int tid = threadIdx.x, j = tid & 31, i0 = tid >> 5;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = i0 + 16 * q;
    zc g = make_double2(0.0, 0.0);
    if (j >= i) g = sf_gram_element(S.pb, i, j, cd_mode);
    if (i >= n || j >= n) g = i == j ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0);
    if (i == j) { g.y = 0.0; S.d0[i] = g.x; }
    S.G[i][j] = g;
  }
  __syncthreads();
}

// G (upper) -> its Cholesky factor R (upper, real positive diagonal); d0 <- 1 / diag(R).  Returns true (uniformly) when
// a pivot fails the conditioning check.
// (512 threads, two elements each: a barrier of 8 waves and 0.30 us per pivot, against 0.7 us with one element per thread
// of a 1024-thread workgroup)
__device__ bool sf_chol(SfMat& G, const double* d0, bool second) {
  // Two pivots per barrier: a thread forms the entries of row k + 1 it needs from rows k and k + 1 as they stand (the very
  // operations the one-pivot step would have stored), so the results are those of 32 single steps; rows k and k + 1
  // are not written (row k + 1 is finished in the scaling pass below): no read of a step races with a write of it.
  const int j = threadIdx.x & 31, i0 = threadIdx.x >> 5;
  for (int k = 0; k < NB; k += 2) {
    const double da = G[k][k].x;
    const bool oka = second ? (da > 0.5 && da < 2.0) : (da > CHOL_TOL * d0[k] && d0[k] > 0.0);
    if (!oka) return true;  // uniform: every thread read the same words
    const double ida = fast_rcp(da);
    const zc g01 = G[k][k + 1];
    const zc s01 = make_double2(g01.x * ida, g01.y * ida);
    const double db = csub(G[k + 1][k + 1], cmulc(g01, s01)).x;
    const bool okb = second ? (db > 0.5 && db < 2.0) : (db > CHOL_TOL * d0[k + 1] && d0[k + 1] > 0.0);
    if (!okb) return true;
    const double idb = fast_rcp(db);
    const zc gkj = G[k][j];
    const zc sa = make_double2(gkj.x * ida, gkj.y * ida);
    const zc r1j = csub(G[k + 1][j], cmulc(g01, sa));  // row k + 1 after pivot k
    const zc sb = make_double2(r1j.x * idb, r1j.y * idb);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = i0 + 16 * q;
      if (i > k + 1 && j >= i) {
        const zc gki = G[k][i];
        const zc r1i = csub(G[k + 1][i], cmulc(g01, make_double2(gki.x * ida, gki.y * ida)));
        G[i][j] = csub(csub(G[i][j], cmulc(gki, sa)), cmulc(r1i, sb));
      }
    }
    __syncthreads();
  }
  zc r[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = i0 + 16 * q;
    zc raw = G[i][j], rd = G[i][i];
    if (i & 1) {  // odd rows still lack the update by the pivot above them
      const double idp = fast_rcp(G[i - 1][i - 1].x);
      const zc gp = G[i - 1][i];
      const zc gpj = G[i - 1][j];
      raw = csub(raw, cmulc(gp, make_double2(gpj.x * idp, gpj.y * idp)));
      rd = csub(rd, cmulc(gp, make_double2(gp.x * idp, gp.y * idp)));
    }
    const double dd = rd.x;
    const double is = 1.0 / sqrt(dd);
    r[q] = j > i ? make_double2(raw.x * is, raw.y * is) : (j == i ? make_double2(dd * is, 0.0) : make_double2(0.0, 0.0));
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) G[i0 + 16 * q][j] = r[q];
  __syncthreads();
  return false;
}

// W = U^-1 for an upper-triangular U by recursive doubling: the inverse of [[U00, U01], [0, U11]] is
// [[U00^-1, -U00^-1 U01 U11^-1], [0, U11^-1]]; 5 levels, two small products each (T is the scratch) -- 10 barriers where a
// substitution chain needs 31.  KIND 0: U = upper triangle of M, real diagonal; 1: the same with a complex diagonal;
// 2: U = unit upper triangular, U[i][k] = conj(M[k][i]) (the V1^H of a block reflector stored as the unit-lower L).
template <int KIND>
__device__ void sf_tri_inverse(const SfMat& M, SfMat& W, SfMat& T) {
  const int tid = threadIdx.x, j = tid & 31, i0 = tid >> 5;
  auto el = [&](int i, int k) __attribute__((always_inline)) -> zc {
    if (KIND == 2) { const zc v = M[k][i]; return make_double2(v.x, -v.y); }
    return M[i][k];
  };
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = i0 + 16 * q;
    zc dinv = make_double2(1.0, 0.0);
    if (KIND == 0) dinv = make_double2(1.0 / M[i][i].x, 0.0);
    if (KIND == 1) dinv = cdiv(make_double2(1.0, 0.0), M[i][i]);
    W[i][j] = i == j ? dinv : make_double2(0.0, 0.0);
  }
  __syncthreads();
  for (int b = 1; b < NB; b <<= 1) {
    const int ne = 16 * b;  // (NB / 2b) pairs x b x b elements
    int ii = 0, jj = 0, c0 = 0;
    if (tid < ne) {
      const int p = tid / (b * b), rem = tid - p * b * b;
      ii = rem / b; jj = rem - ii * b; c0 = p * 2 * b;
      zc t = make_double2(0.0, 0.0);
      for (int k = 0; k <= jj; ++k) t = cadd(t, cmul(el(c0 + ii, c0 + b + k), W[c0 + b + k][c0 + b + jj]));
      T[c0 + ii][c0 + b + jj] = t;
    }
    __syncthreads();
    if (tid < ne) {
      zc x = make_double2(0.0, 0.0);
      for (int k = ii; k < b; ++k) x = csub(x, cmul(W[c0 + ii][c0 + k], T[c0 + k][c0 + b + jj]));
      W[c0 + ii][c0 + b + jj] = x;
    }
    __syncthreads();
  }
}

// (rows of this wave's 16-row blocks) x W: results stay in qre / qim (accumulator layout: lane (li, lk) holds rows
// cd_row(lk, r), column li of either column block); DST: also written to dst (ld = NB)
template <bool DST>
__device__ __forceinline__ void sf_apply(const zc* __restrict__ src, long ld, int m, int ncol, const SfSmem& S, int cd_mode,
                                         zc* __restrict__ dst, sf_d4 (&qre)[3][2], sf_d4 (&qim)[3][2]) {
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
  // complex products in the three-multiplication form (as the large GEMMs): T1 = Ar Br, T2 = Ai Bi, T3 = (Ar + Ai)(Br + Bi),
  // Re = T1 - T2, Im = T3 - T1 - T2 -- three matrix-core products per k-step instead of four, normwise the same error
  zc b0[4], b1[8];
  double bs0[4], bs1[8];
#pragma unroll
  for (int s = 0; s < 4; ++s) { b0[s] = S.W[4 * s + lk][li]; bs0[s] = b0[s].x + b0[s].y; }
#pragma unroll
  for (int s = 0; s < 8; ++s) { b1[s] = S.W[4 * s + lk][16 + li]; bs1[s] = b1[s].x + b1[s].y; }
#pragma unroll
  for (int bl = 0; bl < 3; ++bl) {
    const int blk = w + 8 * bl;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) { qre[bl][cb] = (sf_d4){0.0, 0.0, 0.0, 0.0}; qim[bl][cb] = (sf_d4){0.0, 0.0, 0.0, 0.0}; }
    if (blk * 16 >= m) continue;  // wave-uniform
    const int row = blk * 16 + li;
    zc a[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) a[s] = (row < m && 4 * s + lk < ncol) ? src[(long)row * ld + 4 * s + lk] : make_double2(0.0, 0.0);
    sf_d4 t1[2], t2[2], t3[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) { t1[cb] = (sf_d4){0.0, 0.0, 0.0, 0.0}; t2[cb] = t1[cb]; t3[cb] = t1[cb]; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const double as = a[s].x + a[s].y;
      if (s < 4) {
        t1[0] = sf_mfma(a[s].x, b0[s].x, t1[0]);  t2[0] = sf_mfma(a[s].y, b0[s].y, t2[0]);  t3[0] = sf_mfma(as, bs0[s], t3[0]);
      }
      t1[1] = sf_mfma(a[s].x, b1[s].x, t1[1]);  t2[1] = sf_mfma(a[s].y, b1[s].y, t2[1]);  t3[1] = sf_mfma(as, bs1[s], t3[1]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      qre[bl][cb] = t1[cb] - t2[cb];
      qim[bl][cb] = t3[cb] - t1[cb] - t2[cb];
    }
    if (DST) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int orow = blk * 16 + (cd_mode == 0 ? lk + 4 * r : 4 * lk + r);
          if (orow < m) dst[(long)orow * NB + 16 * cb + li] = make_double2(qre[bl][cb][r], qim[bl][cb][r]);
        }
    }
  }
}

__global__ __launch_bounds__(SF_T) void k_qr_small_fast(const zc* __restrict__ A, int m, int n, zc* __restrict__ Q, zc* __restrict__ R,
                                                        zc* __restrict__ Q1, int cd_mode, int* __restrict__ fail,
                                                        long long* __restrict__ trace, int gauge_free) {
  extern __shared__ __attribute__((aligned(16))) char sf_raw[];
  SfSmem& S = *reinterpret_cast<SfSmem*>(sf_raw);
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, li = l & 15, lk = l >> 4;
  const int j = tid & 31, i0 = tid >> 5;
  sf_d4 qre[3][2], qim[3][2];
  // debugging (MITDVP_QR_TRACE): thread 0 stamps s_memrealtime (10 ns ticks) at the phase boundaries
  auto stamp = [&](int k) __attribute__((always_inline)) { if (trace && tid == 0) trace[k] = (long long)__builtin_amdgcn_s_memrealtime(); };
  stamp(0);
  // ---- round 1 ----
  sf_gram(A, n, m, n, S, cd_mode);
  stamp(1);
  if (sf_chol(S.G, S.d0, false)) { if (tid == 0) *fail = 1; return; }
#pragma unroll
  for (int q = 0; q < 2; ++q) S.P1[i0 + 16 * q][j] = S.G[i0 + 16 * q][j];  // R1 (zero below the diagonal)
  stamp(2);
  sf_tri_inverse<0>(S.G, S.W, S.Qt);
  stamp(3);
  sf_apply<true>(A, n, m, n, S, cd_mode, Q1, qre, qim);
  __threadfence_block();
  __syncthreads();
  stamp(4);
  // ---- round 2 ----
  sf_gram(Q1, NB, m, n, S, cd_mode);
  stamp(5);
  {
    // G = I + E: to first order chol(I + E) = I + U, U = triu(E, 1) + diag(E) / 2, (I + U)^-1 = I - U (see k_fq_chol)
    double e = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = i0 + 16 * q;
      if (j >= i) {
        const zc g = S.G[i][j];
        const double v = fmax(fabs(g.x - (i == j ? 1.0 : 0.0)), fabs(g.y));
        e = (g.x != g.x || g.y != g.y) ? 1e308 : fmax(e, v);  // fmax drops a NaN: keep it visible
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e = fmax(e, __shfl_xor(e, o, 64));
    if (l == 0) S.emax[w] = e;
    __syncthreads();
    double emax = 0.0;
#pragma unroll
    for (int u = 0; u < SF_T / 64; ++u) emax = fmax(emax, S.emax[u]);
    zc rt[2];
    if (emax < 1e-8) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = i0 + 16 * q;
        const zc g = S.G[i][j];
        const zc u = j > i ? g : (j == i ? make_double2(0.5 * (g.x - 1.0), 0.0) : make_double2(0.0, 0.0));
        S.Qt[i][j] = u;
        S.W[i][j] = make_double2((i == j ? 1.0 : 0.0) - u.x, -u.y);
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = i0 + 16 * q;
        zc t = S.P1[i][j];  // (I + U) R1
        for (int k = i; k <= j; ++k) t = cadd(t, cmul(S.Qt[i][k], S.P1[k][j]));
        rt[q] = t;
      }
    } else {
      if (!(emax < 1e300)) { if (tid == 0) *fail = 1; return; }  // NaN / overflow in the first round
      if (sf_chol(S.G, S.d0, true)) { if (tid == 0) *fail = 1; return; }
      sf_tri_inverse<0>(S.G, S.W, S.Qt);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int i = i0 + 16 * q;
        zc t = make_double2(0.0, 0.0);
        for (int k = i; k <= j; ++k) t = cadd(t, cmul(S.G[i][k], S.P1[k][j]));
        rt[q] = t;
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) S.P1[i0 + 16 * q][j] = rt[q];  // R2 R1
    __syncthreads();
  }
  stamp(6);
  sf_apply<false>(Q1, NB, m, NB, S, cd_mode, nullptr, qre, qim);
  // gauge_free (the sweep's moves, qr_thin): R2 R1 has a positive diagonal and Q is orthonormal -- that IS a thin
  // factorisation; the chain below only reproduces LAPACK's signs of diag(R) (9.4 of this kernel's 66 us at 320 x 32)
  if (gauge_free) {
    if (tid < NB) S.Dg[tid] = 1.0;
    __syncthreads();
  }
  // top block -> LDS for the sign chain (rows / columns beyond n: identity)
  if (!gauge_free && w < 2) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = w * 16 + (cd_mode == 0 ? lk + 4 * r : 4 * lk + r), col = 16 * cb + li;
        S.Qt[row][col] = (row < n && col < n) ? make_double2(qre[0][cb][r], qim[0][cb][r])
                                              : (row == col ? make_double2(1.0, 0.0) : make_double2(0.0, 0.0));
      }
  }
  __syncthreads();
  stamp(7);
  // ---- LAPACK's signs: LU of (Q - [D; 0]) with D_k = -sign(Re pivot_k) (k_fq_reconstruct's first chain) ----
  for (int k = 0; k < (gauge_free ? 0 : NB); k += 2) {  // two pivots per barrier (see sf_chol); only D is kept, row / column k + 1 stay as they are
    const zc piva = S.Qt[k][k];
    const double dka = piva.x >= 0.0 ? -1.0 : 1.0;
    const zc ua = make_double2(piva.x - dka, piva.y);
    const double una = ua.x * ua.x + ua.y * ua.y;
    if (!(una > 0.25)) { if (tid == 0) *fail = 1; return; }  // uniform
    const double iuna = fast_rcp(una);
    const zc iua = make_double2(ua.x * iuna, -ua.y * iuna);
    const zc q01 = S.Qt[k][k + 1];
    const zc l10 = cmul(S.Qt[k + 1][k], iua);
    const zc pivb = csub(S.Qt[k + 1][k + 1], cmul(l10, q01));
    const double dkb = pivb.x >= 0.0 ? -1.0 : 1.0;
    const zc ub = make_double2(pivb.x - dkb, pivb.y);
    const double unb = ub.x * ub.x + ub.y * ub.y;
    if (!(unb > 0.25)) { if (tid == 0) *fail = 1; return; }
    if (tid == 0) { S.Dg[k] = dka; S.Dg[k + 1] = dkb; }
    const double iunb = fast_rcp(unb);
    const zc iub = make_double2(ub.x * iunb, -ub.y * iunb);
    const zc qkj = S.Qt[k][j];
    const zc r1j = csub(S.Qt[k + 1][j], cmul(l10, qkj));  // row k + 1 after pivot k
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = i0 + 16 * q;
      if (i > k + 1 && j > k + 1) {
        const zc lia = cmul(S.Qt[i][k], iua);
        const zc c1i = csub(S.Qt[i][k + 1], cmul(lia, q01));  // column k + 1 after pivot k
        S.Qt[i][j] = csub(csub(S.Qt[i][j], cmul(lia, qkj)), cmul(cmul(c1i, iub), r1j));
      }
    }
    __syncthreads();
  }
  stamp(8);
  // ---- Q' = Q D, R' = D R ----
  if (Q) {
#pragma unroll
    for (int bl = 0; bl < 3; ++bl) {
      const int blk = w + 8 * bl;
      if (blk * 16 >= m) continue;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int col = 16 * cb + li;
        const double dc = S.Dg[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = blk * 16 + (cd_mode == 0 ? lk + 4 * r : 4 * lk + r);
          if (row < m && col < n) Q[(long)row * n + col] = make_double2(dc * qre[bl][cb][r], dc * qim[bl][cb][r]);
        }
      }
    }
  }
  if (R) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = i0 + 16 * q;
      if (i < n && j < n) {
        const zc r = S.P1[i][j];
        const double di = S.Dg[i];
        R[(long)i * n + j] = j >= i ? make_double2(di * r.x, di * r.y) : make_double2(0.0, 0.0);
      }
    }
  }
  if (tid == 0) *fail = 0;
  stamp(9);
}
// Partial Gram matrices of a tall panel on the matrix cores: workgroup b takes rows_per_blk rows (eight waves = four row
// quarters x two halves of the six streams, as in the small kernel) and writes ONE 32 x 32 partial, Hermitian, zero beyond nb.
// 128 rows cost about what k_fq_gram's plain-FMA tiles need for 32, so a 4096-row panel leaves 32 partials for the one
// workgroup of the Cholesky kernel to stream instead of 128 (27 of that kernel's 38 us were that stream).
__global__ __launch_bounds__(SF_T) void k_fq_gram_mfma(const zc* __restrict__ P, long ld, int mp, int nb, zc* __restrict__ part,
                                                       int rows_per_blk, int cd_mode) {
  __shared__ SfPart pb;
  const long r0 = (long)blockIdx.x * rows_per_blk;
  const int nr = (int)min((long)rows_per_blk, mp - r0);
  sf_gram_partials(P + r0 * ld, ld, nr, nb, pb);
  const int tid = threadIdx.x, j = tid & 31, i0 = tid >> 5;
  zc* o = part + (size_t)blockIdx.x * NB * NB;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = i0 + 16 * q;
    zc g = make_double2(0.0, 0.0);
    if (i < nb && j < nb) {
      if (j >= i) g = sf_gram_element(pb, i, j, cd_mode);
      else { g = sf_gram_element(pb, j, i, cd_mode); g.y = -g.y; }
    }
    o[i * NB + j] = g;
  }
}
}  // namespace

// MITDVP_QR_GRAM_MFMA=0: the plain-FMA Gram kernel with 32 rows per workgroup (A/B runs)
static bool fq_gram_mfma_on() {
  static const bool on = !(std::getenv("MITDVP_QR_GRAM_MFMA") && std::atoi(std::getenv("MITDVP_QR_GRAM_MFMA")) == 0);
  return on;
}
static int fq_rows_per_blk_mfma(int m) {  // 128 rows per workgroup (MITDVP_QR_GRAM_ROWS; QR ms per 6 sweeps at 64 / 128 / 256 / 512 rows: C3 12.1 / 11.6 / 11.9 / 12.9, C5 1565 / 1557 / 1602 / 1758; 13.2 / 1648 with k_fq_gram), at most 128 workgroups
  static const int rows_env = [] { const char* e = std::getenv("MITDVP_QR_GRAM_ROWS"); return e ? std::max(32, std::atoi(e) / 4 * 4) : 128; }();
  int r = rows_env;
  if ((m + r - 1) / r > 128) r = ((m + 127) / 128 + 3) / 4 * 4;
  return r;
}
// number of partials written
static int fq_gram_launch(hipStream_t st, const zc* P, long ld, int mp, int nb, zc* part) {
  if (fq_gram_mfma_on()) {
    const int rpb = fq_rows_per_blk_mfma(mp), nblk = (mp + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_fq_gram_mfma, dim3(nblk), dim3(SF_T), 0, st, P, ld, mp, nb, part, rpb, zgemm_cd_mode(st));
    return nblk;
  }
  const int rpb = fq_rows_per_blk(mp), nblk = (mp + rpb - 1) / rpb;
  hipLaunchKernelGGL(k_fq_gram, dim3(nblk), dim3(256), 0, st, P, ld, mp, nb, part, rpb);
  return nblk;
}

size_t qr_small_fast_lds() { return sizeof(SfSmem); }

// Q1: m x 32 scratch, fail: one word.  Always queues the launch; the caller queues the conditional Householder kernel.
void qr_small_fast_launch(hipStream_t st, const zc* A, int m, int n, zc* Q, zc* R, zc* Q1, int* fail, bool gauge_free) {
  static std::mutex mu;
  static bool attr_done[64] = {};
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || !attr_done[dev]) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_qr_small_fast), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sizeof(SfSmem)));
      if (dev >= 0 && dev < 64) attr_done[dev] = true;
    }
  }
  const int cd_mode = zgemm_cd_mode(st);
  static const bool tracing = std::getenv("MITDVP_QR_TRACE") != nullptr;
  static long long* trace = nullptr;
  if (tracing && !trace) {
    HIP_CHECK(hipMalloc(&trace, 16 * sizeof(long long)));
    HIP_CHECK(hipMemset(trace, 0, 16 * sizeof(long long)));
  }
  hipLaunchKernelGGL(k_qr_small_fast, dim3(1), dim3(SF_T), sizeof(SfSmem), st, A, m, n, Q, R, Q1, cd_mode, fail, trace, gauge_free ? 1 : 0);
  HIP_CHECK(hipGetLastError());
  if (tracing) {  // phase durations in us: gram, chol, inverse, apply, gram, round 2, apply, sign chain, stores
    long long h[16];
    HIP_CHECK(hipMemcpyAsync(h, trace, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    fprintf(stderr, "[qr_trace] fast m=%d n=%d total %.2f us:", m, n, (double)(h[9] - h[0]) * 0.01);
    for (int k = 0; k < 9; ++k) fprintf(stderr, " %.2f", (double)(h[k + 1] - h[k]) * 0.01);
    fprintf(stderr, "\n");
  }
}


// One panel: A[j0:m, j0:j0+nbp] -> R' / V in place (LAPACK layout), Vp (mp x nbp unit lower trapezoid, contiguous),
// Tp, tau[j0..].  `ws` = the part of the workspace behind the input copy (see qr_fast_work_elems).
int qr_fast_panel(hipStream_t st, zc* A, long lda, int m, int j0, int nbp, zc* Vp, zc* Tp, zc* tau, zc* ws, int* flag) {
  const int mp = m - j0;
  zc* P = A + (long)j0 * lda + j0;
  zc* Q1 = ws;
  zc* Q2 = Q1 + (size_t)m * NB;
  zc* part = Q2 + (size_t)m * NB;
  zc* small = part + (size_t)128 * NB * NB;
  zc* R1inv = small;
  zc* R2inv = small + NB * NB;
  zc* R1 = small + 2 * NB * NB;
  zc* Rtot = small + 3 * NB * NB;
  zc* Uinv = small + 4 * NB * NB;
  // round 1
  int nblk = fq_gram_launch(st, P, lda, mp, nbp, part);
  hipLaunchKernelGGL(k_fq_chol, dim3(1), dim3(1024 / FQ_NQ), 0, st, part, nblk, nbp, R1inv, (const zc*)nullptr, R1, 0, flag);
  // (the applies have no partial results to keep few: 32 rows per workgroup, four 8-row passes each)
  const int arows = 32, ablk = (mp + arows - 1) / arows;
  hipLaunchKernelGGL(k_fq_apply, dim3(ablk), dim3(256), 0, st, P, lda, mp, nbp, R1inv, Q1, (long)NB, (zc*)nullptr, 0L, 0, arows);
  // round 2
  nblk = fq_gram_launch(st, Q1, (long)NB, mp, nbp, part);
  hipLaunchKernelGGL(k_fq_chol, dim3(1), dim3(1024 / FQ_NQ), 0, st, part, nblk, nbp, R2inv, R1, Rtot, 1, flag);
  hipLaunchKernelGGL(k_fq_apply, dim3(ablk), dim3(256), 0, st, Q1, (long)NB, mp, nbp, R2inv, Q2, (long)NB, (zc*)nullptr, 0L, 0, arows);
  // Householder reconstruction: top block, then V2 = Q2[nbp:] U^-1 into the panel and into Vp
  hipLaunchKernelGGL(k_fq_reconstruct, dim3(1), dim3(1024 / FQ_NQ), 0, st, Q2, (long)NB, Rtot, nbp, P, lda, Vp, Tp, tau + j0, Uinv, flag);
  int nl = 7;
  if (mp > nbp) {
    const int nb2 = (mp - nbp + arows - 1) / arows;
    hipLaunchKernelGGL(k_fq_apply, dim3(nb2), dim3(256), 0, st, Q2, (long)NB, mp, nbp, Uinv, P, lda, Vp, (long)nbp, nbp, arows);
    ++nl;
  }
  HIP_CHECK(hipGetLastError());
  return nl;
}


}  // namespace mitdvp
