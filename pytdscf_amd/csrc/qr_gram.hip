// qr_gram.hip -- thin QR WITHOUT LAPACK's sign convention: Q orthonormal, R upper triangular with a positive diagonal.
//
// What the sweep needs from a gauge move (SiteCoef.gauge_trf, _site_cls.py:138-292) is ANY orthonormal Q with Q R = psi:
// the signs LAPACK's zgeqrf / zungqr put on diag(R) are a gauge freedom of the bond no observable sees (SURVEY appendix B
// item 6, section 8c "compare Q R and Q^H Q, not Q itself").  Reproducing them costs the panel factorisations of qr.hip /
// qr_fast.hip a chain of ~10 dependent small launches per 32 columns (77 us per panel at the C3 / C5 shapes,
// profiles/r04_qr_chain_ab.txt).  This path drops the convention and with it the chain:
//
//   blocks of up to 128 columns; per block two passes of block Gram-Schmidt in the "Pythagorean" form (BCGS-PIP, each
//   pass ONE reduction over the rows):
//       [W; G] = [Q_prev P]^H P          one GEMM over the m rows (split-K)
//       R_p    = chol(G - W^H W)         one workgroup, the block in registers (k_gq_chol), R_p^-1 beside it
//       P     <- (P - Q_prev W) R_p^-1   three GEMMs
//   after which  P_in = Q_prev (W_a + W_b R_a) + Q_k (R_b R_a).  The second pass starts from a block whose Gram matrix is
//   I + E with |E| ~ eps cond^2: below 1e-8 its Cholesky factor is I + triu(E, 1) + diag(E) / 2 to rounding and no pivot
//   chain runs at all.  ~10 launches for a 128-column matrix (C3: 4096 x 128), 67 for C5's 2048 x 512, against 55 / 188.
//
// Like CholeskyQR2 this needs cond(A)^2 eps < 1: every pivot is checked on the device against the block's largest diagonal
// element, a failed check (or a second pass that is not a small correction) raises a sticky flag and the caller redoes the
// factorisation with the Householder panels -- the input is never modified here.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "qr.h"
#include "vecops.h"

namespace mitdvp {

constexpr int GQ_NB = 128;  // widest block: 32 x 32 threads holding 4 x 4 elements each

// One workgroup: Cholesky factor R (upper, G = R^H R, positive diagonal) of the Hermitian positive definite n x n matrix
// G (n <= 4 NG) and X = R^-1.  The upper triangle is tiled in 4 x 4 blocks, one per thread (NG (NG + 1) / 2 threads: 528 at
// n = 128 -- a square thread grid would leave half of every wave idle and double the instruction issue, which is what
// bounds a step: 1.15 us per step with 1024 threads and a division in every thread, rocprofv3 round 5); a step
// broadcasts one row (and, in the inversion, one column) through LDS and costs one barrier; the pivot's reciprocal and
// inverse root are formed once, by the thread that owns the pivot, and travel with the row.
//   Cholesky, right-looking on the upper triangle: row k is used UNSCALED (g_ij -= conj(g_ki) g_kj / g_kk), its owners
//   scale it by 1 / sqrt(g_kk) afterwards.
//   Inversion in place by Gauss-Jordan steps: row k <- row k / r_kk with 1 / r_kk in the pivot's place, rows i < k <- row i
//   - r_ik row k with -r_ik / r_kk in column k.
// first_order: the caller expects G = I + E with a small E (second orthogonalisation pass): when max |E| < 1e-8 the
// factor is written down without a pivot chain; when it is not below 0.5 the flag is raised (the first pass failed).
template <int NG>
__global__ __launch_bounds__((NG * (NG + 1) / 2 + 63) / 64 * 64) void k_gq_chol(const zc* __restrict__ G, long ldg, int n,
                                                                                zc* __restrict__ Rout, zc* __restrict__ Xout,
                                                                                long ldo, int first_order, int* __restrict__ flag,
                                                                                long long* __restrict__ trace, int* __restrict__ pub,
                                                                                int pub_tag) {
  constexpr int NP = 4 * NG;
  constexpr int NTRI = NG * (NG + 1) / 2;
  constexpr int NW = (NTRI + 63) / 64;
  __shared__ zc rowbuf[2][NP + 4];  // [NP ..]: (1 / pivot, 1 / sqrt(pivot)) of the step (inversion: one per diagonal 32-block)
  __shared__ zc colbuf[2][NP];
  __shared__ double red[2][NW];
  __shared__ int badw;
  const int t = threadIdx.x;
  // thread -> block (tr, tc), tc >= tr.  First the blocks of the diagonal 32 x 32 super-blocks (36 threads each, row-major
  // inside: the only threads the inversion's pivot steps involve -- contiguous, they occupy three waves instead of a few
  // lanes of every wave), then the other blocks row-major: block rows finish (Cholesky) wave by wave.  Threads beyond the
  // triangle get tr = tc = NG: every ownership test below is false for them.
  int tr = NG, tc = NG;
  {
    constexpr int NSB = NG / 8 > 0 ? NG / 8 : 1, SB = NG < 8 ? NG : 8, PER = SB * (SB + 1) / 2;
    if (t < NSB * PER) {
      const int sb = t / PER;
      int rem = t - sb * PER, r = 0;
      while (rem >= SB - r) { rem -= SB - r; ++r; }
      tr = sb * SB + r;
      tc = tr + rem;
    } else {
      int rem = t - NSB * PER;
      for (int r = 0; r < NG; ++r) {
        const int first = (r / SB + 1) * SB, cnt = NG - first;  // columns right of the row's diagonal super-block
        if (cnt <= 0) continue;
        if (rem < cnt) { tr = r; tc = first + rem; break; }
        rem -= cnt;
      }
    }
  }
  const bool act = tr < NG;
  // the LAST factorisation of a QR also hands the sticky verdict to the host (pub: host-coherent mapped memory, [0] flag,
  // [1] sequence number the host spins on) -- one launch less per QR than a publishing kernel of its own
  auto publish = [&]() {
    if (pub && t == 0) {
      const int f = atomicOr(flag, 0);
      pub[0] = f;
      __threadfence_system();
      *reinterpret_cast<volatile int*>(pub + 1) = pub_tag;
    }
  };
  if (t == 0) badw = 0;
  zc g[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = 4 * tr + a, j = 4 * tc + b;
      zc v = make_double2(i == j ? 1.0 : 0.0, 0.0);  // rows / columns beyond n: the identity (decoupled, pivots 1)
      if (act && i < n && j < n) v = G[(long)i * ldg + j];
      if (i == j) v.y = 0.0;
      g[a][b] = v;
    }
  // largest diagonal element (scale of the pivot test) and max |G - 1| over the upper triangle
  double dmax = 0.0, emax = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = 4 * tr + a, j = 4 * tc + b;
      if (act && i == j) dmax = fmax(dmax, g[a][b].x == g[a][b].x ? g[a][b].x : 1e308);
      if (act && j >= i) {
        const double e = fmax(fabs(g[a][b].x - (i == j ? 1.0 : 0.0)), fabs(g[a][b].y));
        emax = fmax(emax, e == e ? e : 1e308);  // fmax drops a NaN: keep it visible
      }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dmax = fmax(dmax, __shfl_down(dmax, o, 64));
    emax = fmax(emax, __shfl_down(emax, o, 64));
  }
  if ((t & 63) == 0) { red[0][t >> 6] = dmax; red[1][t >> 6] = emax; }
  __syncthreads();
  dmax = 0.0; emax = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { dmax = fmax(dmax, red[0][w]); emax = fmax(emax, red[1][w]); }
  const double ptol = 1e-13 * dmax;

  auto store = [&](zc* out) {
    if (!act) return;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = 4 * tr + a, j = 4 * tc + b;
        if (i < n && j < n) {
          out[(long)i * ldo + j] = (j >= i) ? g[a][b] : make_double2(0.0, 0.0);
          if (tc > tr) out[(long)j * ldo + i] = make_double2(0.0, 0.0);  // the mirror block below the diagonal
        }
      }
  };

  if (first_order) {
    if (!(emax < 0.5)) {
      if (t == 0) atomicOr(flag, 2);
      // still write finite factors (the identity): everything downstream is thrown away by the caller
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) g[a][b] = make_double2((4 * tr + a) == (4 * tc + b) ? 1.0 : 0.0, 0.0);
      store(Rout);
      store(Xout);
      publish();
      return;
    }
    if (emax < 1e-8) {
      // R = 1 + U, X = 1 - U with U = triu(E, 1) + diag(E) / 2: both exact to O(|E|^2) < 1e-16
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int i = 4 * tr + a, j = 4 * tc + b;
          if (act && i < n && j < n) {
            zc u = g[a][b];
            if (i == j) { u.x = 0.5 * (u.x - 1.0); u.y = 0.0; }
            const double dl = i == j ? 1.0 : 0.0;
            Rout[(long)i * ldo + j] = j >= i ? make_double2(dl + u.x, u.y) : make_double2(0.0, 0.0);
            Xout[(long)i * ldo + j] = j >= i ? make_double2(dl - u.x, -u.y) : make_double2(0.0, 0.0);
            if (tc > tr) { Rout[(long)j * ldo + i] = make_double2(0.0, 0.0); Xout[(long)j * ldo + i] = make_double2(0.0, 0.0); }
          }
        }
      publish();
      return;
    }
  }

  // ---- Cholesky ----------------------------------------------------------------------------------------------
  const int nkb = (n + 3) / 4;
  int ntr = 0;
  auto stamp = [&]() { if (trace && t == 0 && ntr < 30) trace[ntr++] = (long long)__builtin_amdgcn_s_memrealtime(); };
  stamp();
  for (int kb = 0; kb < nkb; ++kb) {
    if ((kb & 7) == 0 && kb) stamp();
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
      const int k = 4 * kb + kr;
      zc* rb = rowbuf[k & 1];
      if (tr == kb) {
#pragma unroll
        for (int b = 0; b < 4; ++b) rb[4 * tc + b] = g[kr][b];
        if (tc == kb) {  // the pivot's owner: its reciprocal and inverse root, once for everybody
          const double piv = g[kr][kr].x;
          const bool ok = piv > ptol;
          if (!ok) badw = 1;
          const double ps = ok ? piv : 1.0;
          rb[NP] = make_double2(fast_rcp(ps), fast_rsqrt(ps));
        }
      }
      __syncthreads();
      if (act && tr >= kb) {
        // branch-free inside: rows that must not change (rows <= k of the pivot's own block row) get a zero multiplier,
        // the pivot row a scale factor, every other row the factor 1 (the selects are on scalars, not on the 16 elements)
        const zc pr = rb[NP];
        const bool prow = tr == kb;
        zc cj[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) cj[b] = rb[4 * tc + b];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const zc ri = rb[4 * tr + a];
          const double mm = (prow && a <= kr) ? 0.0 : pr.x;
          const zc m = make_double2(ri.x * mm, -ri.y * mm);  // conj(g_ki) / g_kk
#pragma unroll
          for (int b = 0; b < 4; ++b) {  // g -= m c, four fused multiply-adds
            g[a][b].x = __builtin_fma(-m.x, cj[b].x, g[a][b].x);
            g[a][b].x = __builtin_fma(m.y, cj[b].y, g[a][b].x);
            g[a][b].y = __builtin_fma(-m.x, cj[b].y, g[a][b].y);
            g[a][b].y = __builtin_fma(-m.y, cj[b].x, g[a][b].y);
          }
        }
        const double sc = prow ? pr.y : 1.0;  // the pivot row becomes row k of R
#pragma unroll
        for (int b = 0; b < 4; ++b) { g[kr][b].x *= sc; g[kr][b].y *= sc; }
        if (prow && tc == kb) g[kr][kr].y = 0.0;
      }
    }
  }
  __syncthreads();
  stamp();
  const bool bad = badw != 0;
  if (bad && t == 0) atomicOr(flag, 1);
  store(Rout);
  stamp();

  // ---- X = R^-1 in place -------------------------------------------------------------------------------------
  // Blocked: the diagonal 32 x 32 blocks are inverted side by side (Gauss-Jordan steps as above, 32 of them instead of n:
  // a step is latency bound -- write, barrier, read, update: 0.8 us with all of the triangle behind it in the timeline of
  // MITDVP_QR_TRACE, 105 of the kernel's 175 us at n = 128), then the off-diagonal blocks level by level from
  //   [X11 X12; 0 X22] = [R11 R12; 0 R22]^-1,  X12 = -X11 R12 X22
  // as two products per level staged through LDS (the operands live in other threads' registers).
  constexpr int NBK = NG / 8 > 0 ? NG / 8 : 1;  // diagonal 32-blocks
  const int blk_r = tr >> 3, blk_c = tc >> 3;
  const bool diag_blk = act && blk_r == blk_c;
  for (int kbl = 0; kbl < (NG < 8 ? NG : 8); ++kbl) {
    const int kb = 8 * blk_r + kbl;  // this thread's pivot block row (meaningful on diagonal blocks only)
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
      const int step = 4 * kbl + kr;
      zc* rb = rowbuf[step & 1];
      zc* cb = colbuf[step & 1];
      if (diag_blk && tr == kb) {
#pragma unroll
        for (int b = 0; b < 4; ++b) rb[4 * tc + b] = g[kr][b];
        if (tc == kb) rb[NP + blk_r] = make_double2(bad ? 1.0 : fast_rcp(g[kr][kr].x), 0.0);
      }
      if (diag_blk && tc == kb) {
#pragma unroll
        for (int a = 0; a < 4; ++a) cb[4 * tr + a] = g[a][kr];
      }
      __syncthreads();
      if (diag_blk && tr <= kb && tc >= kb) {
        const double inv = rb[NP + blk_r].x;
        const bool prow = tr == kb, pcol = tc == kb;
        zc rj[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const zc v = rb[4 * tc + b];
          const double f = (pcol && b <= kr) ? 0.0 : inv;  // columns <= k of the pivot's block column take no row update
          rj[b] = make_double2(v.x * f, v.y * f);
        }
        const zc rowk[4] = {g[kr][0], g[kr][1], g[kr][2], g[kr][3]};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          zc f = cb[4 * tr + a];
          if (prow && a >= kr) f = make_double2(0.0, 0.0);  // rows < k only
          const zc old = g[a][kr];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            g[a][b].x = __builtin_fma(-f.x, rj[b].x, g[a][b].x);
            g[a][b].x = __builtin_fma(f.y, rj[b].y, g[a][b].x);
            g[a][b].y = __builtin_fma(-f.x, rj[b].y, g[a][b].y);
            g[a][b].y = __builtin_fma(-f.y, rj[b].x, g[a][b].y);
          }
          // column k of a row above the pivot: -r_ik / r_kk (its old value IS the multiplier f)
          if (pcol && !(prow && a >= kr)) g[a][kr] = make_double2(-old.x * inv, -old.y * inv);
        }
        if (prow) {  // row k itself: scaled, 1 / r_kk in the pivot's place, nothing left of it
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (pcol && b < kr) g[kr][b] = rowk[b];
            else if (pcol && b == kr) g[kr][b] = make_double2(inv, 0.0);
            else g[kr][b] = make_double2(rowk[b].x * inv, rowk[b].y * inv);
          }
        }
      }
    }
  }
  stamp();
  if constexpr (NBK > 1) {
    extern __shared__ __attribute__((aligned(16))) char gq_dyn[];
    zc* dyn = reinterpret_cast<zc*>(gq_dyn);
    // a square h x h area of LDS <- the part of this thread's block inside [r0, r0 + h) x [c0, c0 + h); triangular operands
    // are completed with zeros below the diagonal (there are no threads there)
    // (tp: stored transposed -- the LEFT operand of a product is read down its columns: k-major storage keeps the four
    // rows a thread needs contiguous and different threads' rows in different banks; row-major it was a 16-way conflict)
    auto stage = [&](zc* area, int h, int r0, int c0, bool tri, bool tp) {
      const int i0 = 4 * tr - r0, j0 = 4 * tc - c0;
      if (!act || i0 < 0 || i0 >= h || j0 < 0 || j0 >= h) return;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bool low = tri && (4 * tc + b) < (4 * tr + a);
          const int ij = (i0 + a) * h + j0 + b, ji = (j0 + b) * h + i0 + a;
          area[tp ? ji : ij] = low ? make_double2(0.0, 0.0) : g[a][b];
          if (tri && tc > tr) area[tp ? ij : ji] = make_double2(0.0, 0.0);
        }
    };
    // g <- sgn * A B for the threads of the region [r0, r0 + h) x [c0, c0 + h); At: A stored transposed, B: row-major
    auto product = [&](const zc* At, const zc* B, int h, int r0, int c0, double sgn) {
      const int i0 = 4 * tr - r0, j0 = 4 * tc - c0;
      if (!act || i0 < 0 || i0 >= h || j0 < 0 || j0 >= h) return;
      zc acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = make_double2(0.0, 0.0);
      for (int k = 0; k < h; ++k) {
        zc av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = At[k * h + i0 + a];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = B[k * h + j0 + b];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            acc[a][b].x = __builtin_fma(av[a].x, bv[b].x, acc[a][b].x);
            acc[a][b].x = __builtin_fma(-av[a].y, bv[b].y, acc[a][b].x);
            acc[a][b].y = __builtin_fma(av[a].x, bv[b].y, acc[a][b].y);
            acc[a][b].y = __builtin_fma(av[a].y, bv[b].x, acc[a][b].y);
          }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) g[a][b] = make_double2(sgn * acc[a][b].x, sgn * acc[a][b].y);
    };
#pragma unroll
    for (int h = 32; h < 4 * NG; h *= 2) {
      const int npair = (4 * NG) / (2 * h);
      // T = R12 X22
      for (int pr = 0; pr < npair; ++pr) {
        const int p0 = pr * 2 * h;
        zc* LA = dyn + (size_t)pr * 2 * h * h;
        zc* LB = LA + (size_t)h * h;
        stage(LA, h, p0, p0 + h, false, true);
        stage(LB, h, p0 + h, p0 + h, true, false);
      }
      __syncthreads();
      for (int pr = 0; pr < npair; ++pr) {
        const int p0 = pr * 2 * h;
        const zc* LA = dyn + (size_t)pr * 2 * h * h;
        product(LA, LA + (size_t)h * h, h, p0, p0 + h, 1.0);
      }
      __syncthreads();
      // X12 = -X11 T
      for (int pr = 0; pr < npair; ++pr) {
        const int p0 = pr * 2 * h;
        zc* LA = dyn + (size_t)pr * 2 * h * h;
        zc* LB = LA + (size_t)h * h;
        stage(LA, h, p0, p0 + h, false, false);  // T
        stage(LB, h, p0, p0, true, true);        // X11
      }
      __syncthreads();
      for (int pr = 0; pr < npair; ++pr) {
        const int p0 = pr * 2 * h;
        const zc* LA = dyn + (size_t)pr * 2 * h * h;
        product(LA + (size_t)h * h, LA, h, p0, p0 + h, -1.0);
      }
      __syncthreads();
      stamp();
    }
  }
  stamp();
  store(Xout);
  stamp();
  if (trace && t == 0) trace[31] = ntr;
  publish();
}

static void gq_chol_launch(hipStream_t st, const zc* G, long ldg, int n, zc* R, zc* X, long ldo, int first_order, int* flag,
                           int* pub = nullptr, int pub_tag = 0) {
  // MITDVP_QR_TRACE=1: thread 0 stamps the 100 MHz clock every eight block rows of both chains; printed per launch
  static const bool tracing = std::getenv("MITDVP_QR_TRACE") && std::atoi(std::getenv("MITDVP_QR_TRACE")) != 0;
  static long long* tbuf = nullptr;
  if (tracing && !tbuf) HIP_CHECK(hipMalloc(&tbuf, 32 * sizeof(long long)));
  long long* trace = tracing ? tbuf : nullptr;
  if (tracing) HIP_CHECK(hipMemsetAsync(tbuf, 0, 32 * sizeof(long long), st));
  auto nthr = [](int ng) { return (ng * (ng + 1) / 2 + 63) / 64 * 64; };
  // dynamic LDS: the staging areas of the blocked inversion (two h x h operands per pair of diagonal blocks; h up to 64)
  constexpr size_t LDS16 = 2 * 32 * 32 * sizeof(zc), LDS32 = 2 * 64 * 64 * sizeof(zc);
  if (n > 64) {
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= 64 || !done[dev]) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gq_chol<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS32));
      if (dev >= 0 && dev < 64) done[dev] = true;
    }
  }
  if (n <= 32) hipLaunchKernelGGL(k_gq_chol<8>, dim3(1), dim3(nthr(8)), 0, st, G, ldg, n, R, X, ldo, first_order, flag, trace, pub, pub_tag);
  else if (n <= 64) hipLaunchKernelGGL(k_gq_chol<16>, dim3(1), dim3(nthr(16)), LDS16, st, G, ldg, n, R, X, ldo, first_order, flag, trace, pub, pub_tag);
  else hipLaunchKernelGGL(k_gq_chol<32>, dim3(1), dim3(nthr(32)), LDS32, st, G, ldg, n, R, X, ldo, first_order, flag, trace, pub, pub_tag);
  HIP_CHECK(hipGetLastError());
  if (tracing) {
    long long h[32];
    HIP_CHECK(hipMemcpyAsync(h, tbuf, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    fprintf(stderr, "[qr_trace] n=%d first_order=%d:", n, first_order);
    for (int i = 1; i < (int)h[31] && i < 30; ++i) fprintf(stderr, " %.2f", (double)(h[i] - h[0]) * 0.01);
    fprintf(stderr, " us\n");
  }
}

// workspace (complex elements): S ((n + NB) x NB: W above G), WX (n x NB), T1 (m x NB), Ra, Rb, Xa, Xb (NB x NB each), the flag
size_t qr_gram_work_elems(int m, int n) {
  const size_t nb = (size_t)std::min(n, GQ_NB);
  return ((size_t)n + nb) * nb + (size_t)n * nb + (size_t)m * nb + 4 * nb * nb + 8;
}

int* qr_gram_flag(zc* work, int m, int n) {
  return reinterpret_cast<int*>(work + qr_gram_work_elems(m, n) - 8);
}

// A (m x n, ld n, untouched) -> Q (m x n, ld n), R (n x n, ld n, upper, positive diagonal).  Returns the launches issued;
// the sticky failure flag is *qr_gram_flag(work, m, n) (device memory, cleared here); with pub the last Cholesky kernel
// copies it to pub[0] and then writes pub_tag to pub[1] (host-coherent mapped memory the caller spins on).
int qr_gram(hipStream_t st, const zc* A, int m, int n, zc* Q, zc* R, zc* work, int* pub, int pub_tag) {
  if (m < n || n < 1) throw ArgError("qr_gram: needs m >= n >= 1");
  const int NBmax = std::min(n, GQ_NB);
  zc* S = work;                              // W (j0 x nb) above G (nb x nb)
  zc* WX = S + ((size_t)n + NBmax) * NBmax;
  zc* T1 = WX + (size_t)n * NBmax;
  zc* Ra = T1 + (size_t)m * NBmax;
  zc* Rb = Ra + (size_t)NBmax * NBmax;
  zc* Xa = Rb + (size_t)NBmax * NBmax;
  zc* Xb = Xa + (size_t)NBmax * NBmax;
  int* flag = qr_gram_flag(work, m, n);
  const zc one = make_double2(1.0, 0.0), mone = make_double2(-1.0, 0.0);
  int nl = 0;
  HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), st));
  if (R && n > GQ_NB) HIP_CHECK(hipMemsetAsync(R, 0, (size_t)n * n * sizeof(zc), st));  // one block: the product below writes all of R
  // One pass: the block P (m x nb, leading dimension ldp) is orthogonalised against Q[:, :j0] and within itself,
  //   out = (P - Q_prev W) R_p^-1,  W = Q_prev^H P,  R_p = chol(P^H P - W^H W);
  // W stays in S[:j0], R_p / X_p are written.
  // Wd / ldw: where W goes (the first pass writes it straight into R's block column, where it belongs)
  auto pass = [&](const zc* P, long ldp, int j0, int nb, zc* Rp, zc* Xp, int first_order, zc* out, long ldout, bool last,
                  zc* Wd, long ldw) {
    zc* Gm = S + (size_t)j0 * nb;
    {
      ZgemmDesc g = zgemm_desc(P, P, Gm, nb, nb, m);  // G = P^H P
      g.transA = 1; g.conjA = 1; g.lda = ldp; g.ldb = ldp; g.ldc = nb;
      zgemm(st, g);
      nl += 2;
    }
    if (j0 > 0) {
      ZgemmDesc w = zgemm_desc(Q, P, Wd, j0, nb, m);  // W = Q_prev^H P
      w.transA = 1; w.conjA = 1; w.lda = n; w.ldb = ldp; w.ldc = ldw;
      zgemm(st, w);
      ZgemmDesc g = zgemm_desc(Wd, Wd, Gm, nb, nb, j0);  // G -= W^H W
      g.transA = 1; g.conjA = 1; g.lda = ldw; g.ldb = ldw; g.ldc = nb; g.alpha = mone; g.beta = one;
      zgemm(st, g);
      nl += 3;
    }
    gq_chol_launch(st, Gm, nb, nb, Rp, Xp, nb, first_order, flag, last ? pub : nullptr, pub_tag);
    nl += 1;
    {
      ZgemmDesc g = zgemm_desc(P, Xp, out, m, nb, nb);  // out = P X_p
      g.lda = ldp; g.ldb = nb; g.ldc = ldout;
      zgemm(st, g);
      nl += 1;
    }
    if (j0 > 0) {  // out -= Q_prev (W X_p)
      ZgemmDesc w = zgemm_desc(Wd, Xp, WX, j0, nb, nb);
      w.lda = ldw; w.ldb = nb; w.ldc = nb;
      zgemm(st, w);
      ZgemmDesc g = zgemm_desc(Q, WX, out, m, nb, j0);
      g.lda = n; g.ldb = nb; g.ldc = ldout; g.alpha = mone; g.beta = one;
      zgemm(st, g);
      nl += 2;
    }
  };
  for (int j0 = 0; j0 < n; j0 += GQ_NB) {
    const int nb = std::min(GQ_NB, n - j0);
    pass(A + j0, n, j0, nb, Ra, Xa, 0, T1, nb, false, R ? R + j0 : S, R ? (long)n : (long)nb);  // input block -> T1, W_a -> R[:j0, block]
    pass(T1, nb, j0, nb, Rb, Xb, 1, Q + j0, n, j0 + GQ_NB >= n, S, nb);  // T1 -> Q[:, block]; the last pass publishes the verdict
    if (R) {
      if (j0 > 0) {  // R[:j0, block] += W_b R_a
        ZgemmDesc g = zgemm_desc(S, Ra, R + j0, j0, nb, nb);
        g.lda = nb; g.ldb = nb; g.ldc = n; g.beta = one;
        zgemm(st, g);
        nl += 1;
      }
      ZgemmDesc g = zgemm_desc(Rb, Ra, R + (size_t)j0 * n + j0, nb, nb, nb);  // R[block, block] = R_b R_a
      g.lda = nb; g.ldb = nb; g.ldc = n;
      zgemm(st, g);
      nl += 1;
    }
  }
  return nl;
}

}  // namespace mitdvp
