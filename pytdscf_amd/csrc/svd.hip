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
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "grid_exchange.h"
#include "qr.h"
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


// ---------------------------------------------------------------------------
// Blocked tournament step (round 2).  The row-pair step above moves the whole matrix (M and W, 32 MB at
// 1024 x 1024) through the chip once per pairing: n - 1 = 1023 times per sweep, which is what its 19 us per
// launch are.  Here a pairing is between BLOCKS of JB = 8 rows, so a sweep needs n / 8 - 1 of them, two launches each:
//   k_jacobi_block_rot (one workgroup per block pair, 16 rows):
//     1. Gram matrix G = X X^H (16 x 16) on the matrix cores: with lane (i, k) holding X[i][c0 + k] the A and the B
//        operand of v_mfma_f64_16x16x4_f64 are the SAME register (B[k][j] = X[j][c0 + k]), four real MFMAs per
//        group of four columns; the waves split the columns;
//     2. one cyclic Jacobi sweep on the Hermitian G in LDS: 8 independent rotations per step, 15 steps; every thread
//        (i, j) derives the two rotations it needs itself and updates G <- J G J^H and V <- J V entry-wise (one
//        barrier per step).  Same rotation formulas as the row-pair step, taken from G instead of from the rows:
//        the angles depend on G_pp, G_qq, G_pq through scale-invariant ratios only, and G is formed afresh from the
//        rows at every visit, so the relative accuracy of the one-sided method is kept;
//   k_jacobi_block_apply (eight workgroups per block pair): rows <- V rows for M and W, again on the matrix cores
//        (V is the A operand, a 16-column chunk of the rows the B operand), in place.
// The convergence measure is the largest normalised |G_pq|^2 seen BEFORE its rotation.
// ---------------------------------------------------------------------------
constexpr int JB = 8, J2 = 2 * JB, JT = 512, JW = JT / 64, JAPPLY_Y = 8;
// rotations are skipped two decades below the convergence threshold (1e-30).  Measured and dropped (round 3): skipping at
// the threshold itself costs sweeps (pairs left just under it are pushed over it again by their neighbours' rotations:
// 12 -> 14 sweeps at 1024^2), and a pre-check of the Gram matrix that lets a converged block pair leave before the
// 15-step sweep costs 3 us on every launch for 13 us saved in the last two sweeps only (34.5 / 18.3 us against 31.6).
constexpr double JROT_SKIP = 1e-32;
typedef double jd4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void jblock_pair(int k, int nblk, int round, int& P, int& Q) {
  if (k == 0) { P = nblk - 1; Q = round; }
  else { P = (round + k) % (nblk - 1); Q = (round - k + (nblk - 1)) % (nblk - 1); }
}

// rotation of the pair index i belongs to, as the row map
// x_i' = d x_i + o x_partner;  rel = |G_pq|^2 / (G_pp G_qq) before the rotation (0 when the pair is skipped)
// pairing of index i in step rnd of the 16-player tournament: partner | is_p << 7 (is_p: i is the first of the pair)
__device__ __forceinline__ unsigned jblock_pairing(int i, int rnd) {
  int p, q;
  bool is_p;
  if (i == J2 - 1) { p = i; q = rnd; is_p = true; }
  else if (i == rnd) { p = J2 - 1; q = i; is_p = false; }
  else {
    const int kk = (i - rnd + (J2 - 1)) % (J2 - 1);
    if (kk <= JB - 1) { p = i; q = (rnd - kk + (J2 - 1)) % (J2 - 1); is_p = true; }
    else { const int k2 = (J2 - 1) - kk; p = (rnd + k2) % (J2 - 1); q = i; is_p = false; }
  }
  return (unsigned)(is_p ? q : p) | (is_p ? 128u : 0u);
}

__device__ __forceinline__ void jblock_rot(const zc (*Gc)[J2 + 1], int i, unsigned pairing, double tiny2, int& partner, zc& d, zc& o,
                                           double& rel) {
  partner = (int)(pairing & 31u);
  const bool is_p = (pairing & 128u) != 0u;
  const int p = is_p ? i : partner, q = is_p ? partner : i;
  d = make_double2(1.0, 0.0); o = make_double2(0.0, 0.0); rel = 0.0;
  const double a = Gc[p][p].x, b = Gc[q][q].x;
  const zc g = Gc[p][q];
  const double g2 = g.x * g.x + g.y * g.y;
  if (!(g2 > 0.0) || !(a > tiny2) || !(b > tiny2)) return;
  rel = g2 * fast_rcp(a * b);
  if (rel <= JROT_SKIP) return;
  const double ig = fast_rsqrt(g2);
  const double er = g.x * ig, ei = g.y * ig;  // e^{i phi}
  const double zeta = 0.5 * (b - a) * ig;
  // t = sign / (|zeta| + sqrt(1 + zeta^2)), cos = 1 / sqrt(1 + t^2), sin = t cos: for the small angles of the late sweeps
  // cos comes out as exactly 1 and the rotation is unitary to O(t^2).  Measured and dropped (round 3): the half-angle
  // form cos^2 = (1 + |zeta| / h) / 2, sin = sign / (2 h cos) saves a dependent root but carries a rounding error of
  // one ulp in cos at EVERY small angle; over the ~2e4 rotations a row sees that is 2.5e-13 in the singular values
  // and the reconstruction instead of 9e-14.
  const double z1 = 1.0 + zeta * zeta;
  const double t = (zeta >= 0 ? 1.0 : -1.0) * fast_rcp(fabs(zeta) + z1 * fast_rsqrt(z1));
  const double cs = fast_rsqrt(1.0 + t * t), sn = cs * t;
  // x_p' = cs x_p - sn e^{i phi} x_q ;  x_q' = sn x_p + cs e^{i phi} x_q
  if (is_p) { d = make_double2(cs, 0.0); o = make_double2(-sn * er, -sn * ei); }
  else { d = make_double2(cs * er, cs * ei); o = make_double2(sn, 0.0); }
}

// Column split (round 3): a pairing occupies nblk / 2 workgroups -- 64 of the 256 compute units at 1024 rows -- and each of
// them pulls its 16 rows (256 KB) through one compute unit's memory pipeline, which is what the Gram stage waits for
// (18 of the kernel's 31.6 us).  With gridDim.y = NS > 1 the columns of a pair are split over NS workgroups: each
// writes its partial Gram matrix with agent-scope stores, drains them, takes a ticket from the pair's counter, and the
// workgroup that draws the LAST ticket sums the partials in part order (the result does not depend on who was last)
// and runs the sweep; the others leave.  Nobody waits for anybody, so no co-residency is assumed.
template <int GU>
__global__ __launch_bounds__(JT) void k_jacobi_block_rot(const zc* __restrict__ M, int nrow, int ncol, int nblk, int round,
                                                         unsigned long long* __restrict__ offmax, double tiny2, int cd_mode,
                                                         zc* __restrict__ Vg, int* __restrict__ flags, zc* __restrict__ Gpart,
                                                         unsigned* __restrict__ tickets) {
  __shared__ double Gp[JW][2][J2][J2 + 1];
  __shared__ zc G[2][J2][J2 + 1];
  __shared__ zc V[2][J2][J2 + 1];
  __shared__ unsigned long long relmax_sh;
  __shared__ unsigned ticket_sh;
  __shared__ unsigned char ptab[J2 - 1][J2];  // pairings of the inner tournament (integer modulo chains out of the steps)
  const int NS = gridDim.y, part = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int P, Q;
  jblock_pair(blockIdx.x, nblk, round, P, Q);
  auto rowidx = [&](int i) { return i < JB ? P * JB + i : Q * JB + (i - JB); };
  if (tid == 0) relmax_sh = 0ull;
  if (tid < (J2 - 1) * J2) ptab[tid >> 4][tid & 15] = (unsigned char)jblock_pairing(tid & 15, tid >> 4);
  {
    const int li = lane & 15, lk = lane >> 4;
    const int r = rowidx(li);
    const zc* xr = M + (size_t)(r < nrow ? r : 0) * ncol;
    const bool rv = r < nrow;
    jd4 arr = {0, 0, 0, 0}, aii = {0, 0, 0, 0}, air = {0, 0, 0, 0}, ari = {0, 0, 0, 0};
    const int ngrp = (ncol + 3) / 4;
    const int gstep = JW * NS;
    for (int g0 = part * JW + wv; g0 < ngrp; g0 += gstep * GU) {
      zc x[GU];  // all loads of the pass in flight before the first MFMA
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int g = g0 + gstep * u, c = g * 4 + lk;
        x[u] = (rv && g < ngrp && c < ncol) ? xr[c] : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        arr = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].x, x[u].x, arr, 0, 0, 0);
        aii = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].y, x[u].y, aii, 0, 0, 0);
        air = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].y, x[u].x, air, 0, 0, 0);
        ari = __builtin_amdgcn_mfma_f64_16x16x4f64(x[u].x, x[u].y, ari, 0, 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = cd_mode == 0 ? (lk + 4 * q) : (4 * lk + q);
      Gp[wv][0][row][li] = arr[q] + aii[q];  // Re sum x_i conj(x_j)
      Gp[wv][1][row][li] = air[q] - ari[q];  // Im
    }
  }
  __syncthreads();
  const bool eig = tid < J2 * J2;
  const int ti = (tid >> 4) & 15, tj = tid & 15;
  double gr = 0, gi = 0;
  if (eig) {
#pragma unroll
    for (int u = 0; u < JW; ++u) { gr += Gp[u][0][ti][tj]; gi += Gp[u][1][ti][tj]; }
  }
  if (NS > 1) {
    zc* mine = Gpart + ((size_t)blockIdx.x * NS + part) * (J2 * J2);
    if (eig) gx_stz(mine + ti * J2 + tj, make_double2(gr, gi));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its agent-scope stores have landed
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)NS - 1u) __hip_atomic_store(tickets + blockIdx.x, 0u, GX_RLX);  // ready for the next launch
      ticket_sh = t;
    }
    __syncthreads();
    if (ticket_sh != (unsigned)NS - 1u) return;  // uniform: another workgroup of this pair finishes the job
    if (eig) {
      gr = 0; gi = 0;
      const zc* all = Gpart + (size_t)blockIdx.x * NS * (J2 * J2) + ti * J2 + tj;
      for (int q = 0; q < NS; ++q) {
        const zc v = gx_ldz(all + (size_t)q * (J2 * J2));
        gr += v.x; gi += v.y;
      }
    }
  }
  if (eig) {
    G[0][ti][tj] = make_double2(gr, gi);
    V[0][ti][tj] = make_double2(ti == tj ? 1.0 : 0.0, 0.0);
  }
  __syncthreads();
  int cur = 0;
  double relmax = 0.0;
  for (int rnd = 0; rnd < J2 - 1; ++rnd) {
    if (eig) {
      int pi, pj;
      zc di, oi, dj, oj;
      double ri, rj;
      jblock_rot(G[cur], ti, ptab[rnd][ti], tiny2, pi, di, oi, ri);
      jblock_rot(G[cur], tj, ptab[rnd][tj], tiny2, pj, dj, oj, rj);
      relmax = fmax(relmax, ri);
      dj = zconj(dj); oj = zconj(oj);
      // (J G)_{i l} for l = j and l = partner(j), then times J^H
      const zc a1 = zadd(zmul(di, G[cur][ti][tj]), zmul(oi, G[cur][pi][tj]));
      const zc a2 = zadd(zmul(di, G[cur][ti][pj]), zmul(oi, G[cur][pi][pj]));
      G[cur ^ 1][ti][tj] = zadd(zmul(a1, dj), zmul(a2, oj));
      V[cur ^ 1][ti][tj] = zadd(zmul(di, V[cur][ti][tj]), zmul(oi, V[cur][pi][tj]));
    }
    __syncthreads();
    cur ^= 1;
  }
  if (eig && relmax > 0.0) atomicMax(&relmax_sh, (unsigned long long)__double_as_longlong(relmax));
  __syncthreads();
  const bool any = relmax_sh > (unsigned long long)__double_as_longlong(JROT_SKIP);
  if (tid == 0) {
    flags[blockIdx.x] = any ? 1 : 0;
    if (relmax_sh) atomicMax(offmax, relmax_sh);
  }
  if (eig && any) Vg[(size_t)blockIdx.x * J2 * J2 + ti * J2 + tj] = V[cur][ti][tj];
}

// rows <- V rows, in place: grid (block pairs, JAPPLY_Y); a wave takes 16-column chunks of [M | W]
__global__ __launch_bounds__(256) void k_jacobi_block_apply(zc* __restrict__ M, zc* __restrict__ W, int nrow, int ncol,
                                                            int nblk, int round, int cd_mode, const zc* __restrict__ Vg,
                                                            const int* __restrict__ flags) {
  if (!flags[blockIdx.x]) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  int P, Q;
  jblock_pair(blockIdx.x, nblk, round, P, Q);
  auto rowidx = [&](int i) { return i < JB ? P * JB + i : Q * JB + (i - JB); };
  // A operand: V[i = li][k = 4 s + lk]
  zc v[4];
#pragma unroll
  for (int sI = 0; sI < 4; ++sI) v[sI] = Vg[(size_t)blockIdx.x * J2 * J2 + li * J2 + 4 * sI + lk];
  int rin[4], rout[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    rin[q] = rowidx(4 * q + lk);
    rout[q] = rowidx(cd_mode == 0 ? (lk + 4 * q) : (4 * lk + q));
  }
  const int ncm = (ncol + 15) / 16, ncw = W ? (nrow + 15) / 16 : 0;
  const int nwave = 4 * gridDim.y;
  constexpr int NCH = 4;  // chunks in flight per wave: loads of all of them before the first MFMA
  for (int ch0 = blockIdx.y * 4 + wv; ch0 < ncm + ncw; ch0 += nwave * NCH) {
    zc x[NCH][4];
    zc* base[NCH];
    int len[NCH], c[NCH];
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      const int ch = ch0 + u * nwave;
      const bool live = ch < ncm + ncw, isw = ch >= ncm;
      base[u] = isw ? W : M;
      len[u] = live ? (isw ? nrow : ncol) : 0;
      c[u] = (isw ? ch - ncm : ch) * 16 + li;
#pragma unroll
      for (int sI = 0; sI < 4; ++sI)
        x[u][sI] = (rin[sI] < nrow && c[u] < len[u]) ? base[u][(size_t)rin[sI] * len[u] + c[u]] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
      jd4 yr = {0, 0, 0, 0}, yi = {0, 0, 0, 0};
#pragma unroll
      for (int sI = 0; sI < 4; ++sI) {
        yr = __builtin_amdgcn_mfma_f64_16x16x4f64(v[sI].x, x[u][sI].x, yr, 0, 0, 0);
        yr = __builtin_amdgcn_mfma_f64_16x16x4f64(-v[sI].y, x[u][sI].y, yr, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f64_16x16x4f64(v[sI].x, x[u][sI].y, yi, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f64_16x16x4f64(v[sI].y, x[u][sI].x, yi, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (rout[q] < nrow && c[u] < len[u]) base[u][(size_t)rout[q] * len[u] + c[u]] = make_double2(yr[q], yi[q]);
    }
  }
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
  // blocked step from 64 rows on (below that the row-pair step has as many workgroups and less to do each)
  static const int blocked = [] { const char* e = std::getenv("MITDVP_SVD_BLOCKED"); return e ? std::atoi(e) : 1; }();
  const bool blk = blocked && nr >= 64;
  const int nblk = ((nr + JB - 1) / JB + 1) & ~1;
  const int cd_mode = blk ? zgemm_cd_mode(st) : 0;
  struct Scratch {  // per block pair: the 16 x 16 rotation and a "rotated at all" flag
    zc* p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
  } vg;
  int* flags = nullptr;
  // column split of the Gram stage: up to four workgroups per pair while the launch stays within ~two waves of the chip
  // and every part keeps at least one full pass of eight column groups per wave (MITDVP_SVD_SPLIT=1: off)
  static const int split_max = [] { const char* e = std::getenv("MITDVP_SVD_SPLIT"); return e ? std::max(1, std::atoi(e)) : 4; }();
  int ns = 1;
  while (blk && ns * 2 <= split_max && (nblk / 2) * ns * 2 <= 512 && (nc + 3) / 4 >= JW * ns * 2 * 8) ns *= 2;
  zc* gpart = nullptr;
  unsigned* tickets = nullptr;
  if (blk) {
    const size_t npair = (size_t)(nblk / 2);
    HIP_CHECK(hipMalloc((void**)&vg.p, (npair * (J2 * J2 + 1) + npair * ns * (J2 * J2) + npair) * sizeof(zc)));
    flags = reinterpret_cast<int*>(vg.p + npair * J2 * J2);
    gpart = vg.p + npair * (J2 * J2 + 1);
    tickets = reinterpret_cast<unsigned*>(gpart + npair * ns * (J2 * J2));
    HIP_CHECK(hipMemsetAsync(tickets, 0, npair * sizeof(unsigned), st));
  }
  for (; sweeps < 60; ++sweeps) {
    HIP_CHECK(hipMemsetAsync(off_dev, 0, sizeof(unsigned long long), st));
    if (blk) {
      for (int round = 0; round < nblk - 1; ++round) {
        if (ns > 1)
          hipLaunchKernelGGL(k_jacobi_block_rot<8>, dim3(nblk / 2, ns), dim3(JT), 0, st, M, nr, nc, nblk, round, off_dev, tiny2,
                             cd_mode, vg.p, flags, gpart, tickets);
        else if (nc > JW * 16 * 4)
          hipLaunchKernelGGL(k_jacobi_block_rot<32>, dim3(nblk / 2), dim3(JT), 0, st, M, nr, nc, nblk, round, off_dev, tiny2,
                             cd_mode, vg.p, flags, gpart, tickets);
        else
          hipLaunchKernelGGL(k_jacobi_block_rot<16>, dim3(nblk / 2), dim3(JT), 0, st, M, nr, nc, nblk, round, off_dev, tiny2,
                             cd_mode, vg.p, flags, gpart, tickets);
        hipLaunchKernelGGL(k_jacobi_block_apply, dim3(nblk / 2, JAPPLY_Y), dim3(256), 0, st, M, W, nr, nc, nblk, round, cd_mode,
                           vg.p, flags);
      }
    } else {
      for (int round = 0; round < np - 1; ++round)
        hipLaunchKernelGGL(k_jacobi_step, dim3(np / 2), dim3(256), 0, st, M, W, nr, nc, np, round, off_dev, tiny2);
    }
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

// QR preconditioning (Drmac / Veselic): with M^T = Q1 R1 the rows of R1 have the Gram matrix R1 R1^H, which is one
// step of the LR (Cholesky) iteration away from M M^H -- the coupling between rows belonging to singular values
// sigma_i > sigma_j has shrunk by (sigma_j / sigma_i)^2 -- so the Jacobi sweeps start from a matrix whose large and
// small parts are already decoupled: 16 -> 12 sweeps on a random 1024 x 1024 matrix, 54 -> 12 on a
// spectrum graded over twelve decades (profiles/r03_svd_precond_ab.txt; the QR is 4.3 ms of 72).  M^T rather than M^H so that no conjugation pass is needed:
//   R1 = W^H S Vq  (row Jacobi, W accumulated)   =>   M = R1^T Q1^T = Vq^T S (conj(W) Q1^T).
// MITDVP_SVD_PRECOND=0 switches it off; it is used from 128 rows on (below that the QR costs more than it saves).
// A second LR step (R1^T = Q2 R2, sweeps on R2) takes graded spectra from 12 to 8 sweeps and leaves random ones where they
// are, for a second QR worth 0.8 sweeps at 1024^2: taken when R1's diagonal is graded (default), always (=2) or never (=1).
// -1 (default): one step, and a second one when the diagonal of R1 says the spectrum is graded or rank-deficient
// (max |r_ii| > 1e3 min |r_ii|; a Gaussian matrix of 1024 rows stays below 1e2).  Bond matrices, junction matrices and
// doubled density operators are graded; the random matrices of a benchmark are not.
static int svd_precond_steps() {
  static const int v = [] { const char* e = std::getenv("MITDVP_SVD_PRECOND"); return e ? std::atoi(e) : -1; }();
  return v;
}

__global__ void k_diag_abs2(const zc* __restrict__ X, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const zc v = X[(size_t)i * n + i];
    out[i] = v.x * v.x + v.y * v.y;
  }
}
static int svd_precond_rows() { return svd_precond_steps() ? 128 : INT_MAX; }

size_t svd_work_elems(int r, int c) {
  const int nr = std::min(r, c), nc = std::max(r, c);
  size_t e = (size_t)nr * nc + (size_t)nr * nr + nr /*s*/ + nr /*idx*/ + 8 + (size_t)nr * nr + (size_t)nr * nc;
  if (nr >= svd_precond_rows()) e += (size_t)nr * nr + (size_t)nr * nc + qr_work_elems(nc, nr) + 8;
  return e;
}

// A (r x c, row-major) = U (r x k) diag(S) Vh (k x c), k = min(r, c); S descending (host array).
static void svd_jacobi_raw(hipStream_t st, const zc* A, int r, int c, zc* U, double* S_host, zc* Vh, zc* work, int* sweeps_out) {
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
  auto sorted = [&](const zc* X, int ncol, std::vector<int>& idx) {  // row norms -> S_host (descending), idx_dev
    hipLaunchKernelGGL(k_row_norms, dim3(nr), dim3(256), 0, st, X, ncol, s_dev);
    std::vector<double> s(nr);
    HIP_CHECK(hipMemcpyAsync(s.data(), s_dev, nr * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    idx.resize(nr);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return s[a] > s[b]; });
    HIP_CHECK(hipMemcpyAsync(idx_dev, idx.data(), nr * sizeof(int), hipMemcpyHostToDevice, st));
    for (int k = 0; k < nr; ++k) S_host[k] = s[idx[k]];
  };
  std::vector<int> idx;
  if (nr >= svd_precond_rows()) {
    zc* Cw = Vt + (size_t)nr * nc;        // (nr x nr) conj(W)^T with the rows of W in sorted order
    zc* G2 = Cw + (size_t)nr * nr;        // (nr x nc) conj(W) Q1^T
    zc* qrw = G2 + (size_t)nr * nc;
    zc* P = M;                            // (nc x nr) = M^T, overwritten by the reflectors
    zc* Q1 = Vt;                          // (nc x nr)
    zc* X = Ut;                           // (nr x nr) = R1, rotated in place
    if (tr) HIP_CHECK(hipMemcpyAsync(P, A, (size_t)r * c * sizeof(zc), hipMemcpyDeviceToDevice, st));
    else transpose_batched(st, A, P, r, c, c, r, 1, 0, 0);
    long nl = 0;
    qr_householder(st, P, nc, nr, Q1, X, qrw, &nl);
    set_identity(st, W, nr, nr, nr);
    bool second = svd_precond_steps() >= 2;
    if (svd_precond_steps() < 0) {
      hipLaunchKernelGGL(k_diag_abs2, dim3((nr + 255) / 256), dim3(256), 0, st, X, nr, s_dev);
      std::vector<double> dg(nr);
      HIP_CHECK(hipMemcpyAsync(dg.data(), s_dev, nr * sizeof(double), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      double mx = 0.0, mn = HUGE_VAL;
      for (double v : dg) { mx = std::max(mx, v); mn = std::min(mn, v); }
      second = !(mn > 1e-6 * mx);  // squared moduli: a ratio of 1e3 between the diagonal entries
    }
    if (second) {
      // a second LR step: R1^T = Q2 R2, sweeps on the rows of R2:  W R2 = S Vq  =>  M = R1^T Q1^T = (Q2 W^H) S (Vq Q1^T)
      zc* Q2 = Cw;
      zc* X2 = G2;
      transpose_batched(st, X, M, nr, nr, nr, nr, 1, 0, 0);
      qr_householder(st, M, nr, nr, Q2, X2, qrw, &nl);
      const int sweeps = jacobi_rows(st, X2, W, nr, nr, off_dev, s_dev);
      sorted(X2, nr, idx);
      zc* Vq = M;    // (nr x nr)
      zc* Wh = Ut;   // (nr x nr) W^H with its columns in sorted order
      hipLaunchKernelGGL(k_svd_gather, dim3(nr), dim3(256), 0, st, X2, W, idx_dev, s_dev, nr, nr, Wh, Vq);
      ZgemmDesc gu = zgemm_desc(Q2, Wh, tr ? W : U, nr, nr, nr);          // Q2 W^H
      zgemm(st, gu);
      ZgemmDesc gv = zgemm_desc(Vq, Q1, tr ? G2 : Vh, nr, nc, nr);        // Vq Q1^T
      gv.transB = 1; gv.ldb = nr;
      zgemm(st, gv);
      if (tr) {  // A = M^T = (Vq Q1^T)^T S (Q2 W^H)^T
        transpose_batched(st, G2, U, nr, nc, nc, nr, 1, 0, 0);
        transpose_batched(st, W, Vh, nr, nr, nr, nr, 1, 0, 0);
      }
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(st));
      if (sweeps_out) *sweeps_out = sweeps;
      return;
    }
    const int sweeps = jacobi_rows(st, X, W, nr, nr, off_dev, s_dev);
    sorted(X, nr, idx);
    zc* Vq = M;  // (nr x nr): the reflectors are no longer needed
    hipLaunchKernelGGL(k_svd_gather, dim3(nr), dim3(256), 0, st, X, W, idx_dev, s_dev, nr, nr, Cw, Vq);
    // conj(W_sorted) Q1^T = Cw^T Q1^T
    ZgemmDesc g = zgemm_desc(Cw, Q1, tr ? G2 : Vh, nr, nc, nr);
    g.transA = 1; g.lda = nr;
    g.transB = 1; g.ldb = nr;
    zgemm(st, g);
    if (!tr) {  // A = M = Vq^T S G2
      transpose_batched(st, Vq, U, nr, nr, nr, nr, 1, 0, 0);
    } else {    // A = M^T = G2^T S Vq
      transpose_batched(st, G2, U, nr, nc, nc, nr, 1, 0, 0);
      HIP_CHECK(hipMemcpyAsync(Vh, Vq, (size_t)nr * nr * sizeof(zc), hipMemcpyDeviceToDevice, st));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(st));
    if (sweeps_out) *sweeps_out = sweeps;
    return;
  }
  if (tr) transpose_batched(st, A, M, r, c, c, r, 1, 0, 0);
  else HIP_CHECK(hipMemcpyAsync(M, A, (size_t)r * c * sizeof(zc), hipMemcpyDeviceToDevice, st));
  set_identity(st, W, nr, nr, nr);
  const int sweeps = jacobi_rows(st, M, W, nr, nc, off_dev, s_dev);
  sorted(M, nc, idx);
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

// Column j of C (m x k, row-major) <- column j of Q times the phase of R[j][j]: the columns that were orthonormal come
// back as they were (C = Q R with R diagonal, |R_jj| = 1 there), the others as an orthonormal completion.
__global__ __launch_bounds__(256) void k_complete_cols(zc* __restrict__ C, const zc* __restrict__ Q, const zc* __restrict__ R, long m, int k) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= m * k) return;
  const int j = (int)(e % k);
  const zc d = R[(size_t)j * k + j];
  const double a = sqrt(d.x * d.x + d.y * d.y);
  const zc ph = a > 0.5 ? make_double2(d.x / a, d.y / a) : make_double2(1.0, 0.0);
  const zc q = Q[e];
  C[e] = make_double2(q.x * ph.x - q.y * ph.y, q.x * ph.y + q.y * ph.x);
}

// The one-sided Jacobi sweep leaves the singular vectors of (numerically) ZERO singular values as whatever its rotations
// cancel down to: the factor accumulated from the rotations is unitary, but the rows of the rotated matrix -- divided by
// their own norm -- are rounding residue, typically PARALLEL to the leading vectors.  LAPACK returns an orthonormal
// completion there, and callers that keep the dimension rely on it (truncate_sigvec(keepdim=True) rebuilds environment
// blocks from A U and Vh B; gauge_trf(regularize=True) lifts zero singular values ALONG those vectors).  For a
// numerically rank-deficient input both factors are therefore passed through a Householder QR: vectors that were
// orthonormal come back unchanged (to rounding), the rest as an orthonormal completion.  Rare path, own allocations.
static void complete_cols(hipStream_t st, zc* C, int m, int k) {  // C (m x k, row-major), m >= k
  zc *P = nullptr, *Q = nullptr, *R = nullptr, *wk = nullptr;
  struct Free { zc** p[4]; ~Free() { for (auto q : p) if (*q) (void)hipFree(*q); } } fr{{&P, &Q, &R, &wk}};
  HIP_CHECK(hipMalloc(&P, (size_t)m * k * sizeof(zc)));
  HIP_CHECK(hipMalloc(&Q, (size_t)m * k * sizeof(zc)));
  HIP_CHECK(hipMalloc(&R, (size_t)k * k * sizeof(zc)));
  HIP_CHECK(hipMalloc(&wk, qr_work_elems(m, k) * sizeof(zc)));
  HIP_CHECK(hipMemcpyAsync(P, C, (size_t)m * k * sizeof(zc), hipMemcpyDeviceToDevice, st));
  long nl = 0;
  qr_householder(st, P, m, k, Q, R, wk, &nl);
  hipLaunchKernelGGL(k_complete_cols, dim3((unsigned)(((long)m * k + 255) / 256)), dim3(256), 0, st, C, Q, R, (long)m, k);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(st));
}

void svd_jacobi(hipStream_t st, const zc* A, int r, int c, zc* U, double* S_host, zc* Vh, zc* work, int* sweeps_out) {
  svd_jacobi_raw(st, A, r, c, U, S_host, Vh, work, sweeps_out);
  const int k = std::min(r, c);
  if (k < 2 || S_host[k - 1] > 1e-12 * S_host[0]) return;  // every singular vector is determined
  complete_cols(st, U, r, k);
  zc* T = nullptr;  // Vh (k x c): its rows are the vectors
  struct Free { zc*& p; ~Free() { if (p) (void)hipFree(p); } } fr{T};
  HIP_CHECK(hipMalloc(&T, (size_t)k * c * sizeof(zc)));
  transpose_batched(st, Vh, T, k, c, c, k, 1, 0, 0);
  complete_cols(st, T, c, k);
  transpose_batched(st, T, Vh, c, k, k, c, 1, 0, 0);
  HIP_CHECK(hipStreamSynchronize(st));
}

}  // namespace mitdvp
