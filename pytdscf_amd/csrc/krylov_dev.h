// krylov_dev.h -- the convergence logic of a local exponential kept on the device (multi-launch regime).
//
// The reference evaluates, on the host and after two device->host synchronisations per iteration
// (_integrator.py:553-554), the projected exponential (:590, :617-637 / :397-409) and the test
// ||psi_k - psi_{k-1}|| < thresh (:638-651).  Here both run on the device behind the vector kernels of the
// iteration; the host reads ONE published record per local exponential in the steady state (the iteration at which the
// reference's warm-up memory, _iter_info :178-186, allows the first convergence), and one more per extra iteration.
//
//   kry_ritz  (one workgroup; small_site.hip, next to the k x k exponential it shares with the small-bond kernel):
//             sums the partials of the iterations not inspected yet, detects an exhausted Krylov space (:569, :392),
//             coef = exp(scale * T_k) e_0, then either records it as the first Ritz vector, or leaves
//             dcoef = coef - coef_prev for kry_diff, or (exhausted) closes the exponential.
//   kry_diff  (NPART workgroups; vecops.hip): || sum_j dcoef_j V_j ||^2 in per-workgroup partials; the workgroup whose
//             arrival ticket comes last sums them in a fixed order, applies the threshold, rolls coef_prev and publishes
//             {state, k, coef, beta0} into host-mapped memory behind a sequence word.
#pragma once
#include "common.h"
#include "vecops.h"

namespace mitdvp {

enum { KRY_RUNNING = 0, KRY_CONVERGED = 1, KRY_EXHAUSTED = 2 };

struct KryDev {  // device memory, one per engine
  zc alpha[MAXK];
  zc hess[(MAXK + 1) * MAXK];  // row-major, ld = MAXK
  zc coef[MAXK], cprev[MAXK], dcoef[MAXK];
  double beta[MAXK];
  double beta0, err;
  double part[NPART];
  int have_prev, prev_len, k, state, need_diff;
  unsigned ticket;
  int deferred;         // the stored basis is unnormalised: v_j = u_j * invb[j]
  double invb[MAXK];
};

struct KryPub {  // host-coherent mapped memory
  zc coef[MAXK];
  double beta0, err;
  int state, k;
  unsigned seq;  // written last
};

struct KryRitzArgs {
  KryDev* st;
  const zc* alpha_p;       // [MAXK][NPART]
  const double* nrm_p;     // [MAXK][NPART]
  const zc* h_p;           // [MAXK][MAXK][NPART] (Arnoldi: column q at q * MAXK * NPART)
  const double* beta0_p;   // [NPART] partials of |x|^2, nullptr with conserve_norm
  int l, q0, n_warm, ndim;
  long nsize;
  int lanczos, first;
  zc scale;
  double eps;
  int deferred;   // Lanczos on an unnormalised basis: alpha_p holds raw dots <x | H u_q>; orthodox: x = u_q
  int orthodox;
};

void kry_ritz(hipStream_t st, const KryRitzArgs& a);
void kry_diff(hipStream_t st, KryDev* kst, const zc* V, long ldv, long n, double thresh, KryPub* pub_dev, unsigned tag);

}  // namespace mitdvp
