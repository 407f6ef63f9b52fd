/* A host program in plain C on top of the C ABI (include/mitdvp.h): synthetic Hermitian chain,
 * device-side random state, a few time steps, observables.  No Python, no C++ types in sight.
 *
 *   gcc -std=c11 -O2 examples/c_host.c -Iinclude -Lpytdscf_amd/csrc -lmitdvp -lm \
 *       -Wl,-rpath,$PWD/pytdscf_amd/csrc -o c_host && ./c_host
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "mitdvp.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    int rc_ = (call);                                                                \
    if (rc_ != MITDVP_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mitdvp_last_error(eng));   \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

/* nearest-neighbour Hermitian MPO, bond dimension 3: [[1, A, B], [0, 0, A], [0, 0, 1]] */
static void fill_core(double* w, int ml, int d, int mr, int first, int last, unsigned* seed) {
  for (long i = 0; i < 2L * ml * d * d * mr; ++i) w[i] = 0.0;
  double* A = calloc(2 * d * d, sizeof(double));
  double* B = calloc(2 * d * d, sizeof(double));
  for (int i = 0; i < d; ++i)
    for (int j = i; j < d; ++j) {
      const double ar = 0.01 * ((double)rand_r(seed) / RAND_MAX - 0.5), ai = i == j ? 0.0 : 0.01 * ((double)rand_r(seed) / RAND_MAX - 0.5);
      const double br = 0.05 * ((double)rand_r(seed) / RAND_MAX - 0.5), bi = i == j ? 0.0 : 0.05 * ((double)rand_r(seed) / RAND_MAX - 0.5);
      A[2 * (i * d + j)] = ar; A[2 * (i * d + j) + 1] = ai; A[2 * (j * d + i)] = ar; A[2 * (j * d + i) + 1] = -ai;
      B[2 * (i * d + j)] = br; B[2 * (i * d + j) + 1] = bi; B[2 * (j * d + i)] = br; B[2 * (j * d + i) + 1] = -bi;
    }
#define W(c, i, j, t) (w + 2 * ((((long)(c) * d + (i)) * d + (j)) * mr + (t)))
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) {
      const double* a = A + 2 * (i * d + j);
      const double* b = B + 2 * (i * d + j);
      const double id = i == j ? 1.0 : 0.0;
      if (first && last) { W(0, i, j, 0)[0] = b[0]; W(0, i, j, 0)[1] = b[1]; continue; }
      if (first) {  /* row 0: [1, A, B] */
        W(0, i, j, 0)[0] = id; W(0, i, j, 1)[0] = a[0]; W(0, i, j, 1)[1] = a[1]; W(0, i, j, 2)[0] = b[0]; W(0, i, j, 2)[1] = b[1];
      } else if (last) {  /* column 2: [B, A, 1]^T */
        W(0, i, j, 0)[0] = b[0]; W(0, i, j, 0)[1] = b[1]; W(1, i, j, 0)[0] = a[0]; W(1, i, j, 0)[1] = a[1]; W(2, i, j, 0)[0] = id;
      } else {
        W(0, i, j, 0)[0] = id; W(0, i, j, 1)[0] = a[0]; W(0, i, j, 1)[1] = a[1]; W(0, i, j, 2)[0] = b[0]; W(0, i, j, 2)[1] = b[1];
        W(1, i, j, 2)[0] = a[0]; W(1, i, j, 2)[1] = a[1]; W(2, i, j, 2)[0] = id;
      }
    }
#undef W
  free(A); free(B);
}

/* Two electronic states (mitdvp_ms_*): the same chain on both states, shifted by a scalar on the
 * second one and coupled by a scalar term; host-side random tensors, canonicalised on the device. */
static int two_states(void) {
  enum { L = 6, D = 8, PHYS = 3, M = 3, S = 2 };
  mitdvp_engine* eng = NULL;
  mitdvp_config cfg = {0};
  cfg.nsite = L; cfg.device = 0; cfg.integrator = MITDVP_LANCZOS; cfg.conserve_norm = 1; cfg.thresh = 1e-9; cfg.max_krylov = 20;
  CHECK(mitdvp_create(&cfg, &eng));
  CHECK(mitdvp_ms_configure(eng, S));
  unsigned seed = 11;
  for (int p = 0; p < L; ++p) {
    const int ml = p == 0 ? 1 : M, mr = p == L - 1 ? 1 : M;
    double* w = malloc(sizeof(double) * 2 * ml * PHYS * PHYS * mr);
    fill_core(w, ml, PHYS, mr, p == 0, p == L - 1, &seed);
    for (int s = 0; s < S; ++s) CHECK(mitdvp_ms_set_mpo_core(eng, 0, s, s, p, w, ml, PHYS, PHYS, mr));
    free(w);
  }
  CHECK(mitdvp_ms_set_coupleJ(eng, 0, 1, 1, 0.2, 0.0));   /* energy offset of state 1 */
  CHECK(mitdvp_ms_set_coupleJ(eng, 0, 0, 1, 0.05, 0.02)); /* Hermitian scalar coupling */
  CHECK(mitdvp_ms_set_coupleJ(eng, 0, 1, 0, 0.05, -0.02));
  int bl[L + 1];
  bl[0] = 1; bl[L] = 1;
  for (int p = 1; p < L; ++p) {  /* min(D, d^p, d^(L-p)) */
    long a = 1, b = 1;
    for (int q = 0; q < p && a <= D; ++q) a *= PHYS;
    for (int q = p; q < L && b <= D; ++q) b *= PHYS;
    bl[p] = (int)(a < b ? (a < D ? a : D) : (b < D ? b : D));
  }
  for (int s = 0; s < S; ++s)
    for (int p = 0; p < L; ++p) {
      const long n = 2L * bl[p] * PHYS * bl[p + 1];
      double* t = malloc(sizeof(double) * n);
      for (long i = 0; i < n; ++i) t[i] = (double)rand_r(&seed) / RAND_MAX - 0.5;
      CHECK(mitdvp_ms_set_site(eng, s, p, t, bl[p], PHYS, bl[p + 1], MITDVP_GAUGE_C));
      free(t);
    }
  CHECK(mitdvp_ms_canonicalize(eng, 0, sqrt(0.8)));
  CHECK(mitdvp_ms_canonicalize(eng, 1, sqrt(0.2)));
  double e0[2], e1[2], pops0[S], pops[S];
  CHECK(mitdvp_ms_expect(eng, 0, e0));
  CHECK(mitdvp_ms_pops(eng, pops0));
  for (int k = 0; k < 4; ++k) CHECK(mitdvp_ms_step(eng, 0.5));
  CHECK(mitdvp_ms_expect(eng, 0, e1));
  CHECK(mitdvp_ms_pops(eng, pops));
  printf("two states: populations %.6f %.6f -> %.6f %.6f, energy %.12f -> %.12f\n", pops0[0], pops0[1], pops[0], pops[1], e0[0], e1[0]);
  const int ok = fabs(pops[0] + pops[1] - 1.0) < 1e-12 && fabs(e1[0] - e0[0]) < 1e-8 && fabs(pops[0] - pops0[0]) > 1e-6;
  mitdvp_destroy(eng);
  return ok ? 0 : 1;
}

int main(void) {
  enum { L = 8, D = 16, PHYS = 4, M = 3 };
  mitdvp_engine* eng = NULL;
  mitdvp_config cfg = {0};
  cfg.nsite = L; cfg.device = 0; cfg.integrator = MITDVP_LANCZOS; cfg.conserve_norm = 1; cfg.relax = 0;
  cfg.thresh = 1e-9; cfg.max_krylov = 20;
  CHECK(mitdvp_create(&cfg, &eng));
  unsigned seed = 7;
  for (int p = 0; p < L; ++p) {
    const int ml = p == 0 ? 1 : M, mr = p == L - 1 ? 1 : M;
    double* w = malloc(sizeof(double) * 2 * ml * PHYS * PHYS * mr);
    fill_core(w, ml, PHYS, mr, p == 0, p == L - 1, &seed);
    CHECK(mitdvp_set_mpo_core(eng, 0, p, w, ml, PHYS, PHYS, mr));
    free(w);
  }
  int dims[L];
  for (int p = 0; p < L; ++p) dims[p] = PHYS;
  CHECK(mitdvp_init_random(eng, dims, D, 1));
  double e0[2], e1[2], ac[2], nrm;
  CHECK(mitdvp_expect(eng, 0, e0));
  for (int s = 0; s < 5; ++s) CHECK(mitdvp_step(eng, 0.5));
  CHECK(mitdvp_expect(eng, 0, e1));
  CHECK(mitdvp_autocorr(eng, ac));
  CHECK(mitdvp_norm(eng, &nrm));
  int k[L];
  CHECK(mitdvp_krylov_stats(eng, k));
  printf("%s\n", mitdvp_version());
  printf("energy before %.12f  after 5 steps %.12f (imag %.1e)\n", e0[0], e1[0], e1[1]);
  printf("norm %.15f  autocorr %.9f%+.9fi  krylov[0] %d\n", nrm, ac[0], ac[1], k[0]);
  const int ok = fabs(nrm - 1.0) < 1e-12 && fabs(e1[0] - e0[0]) < 1e-8 * fmax(1.0, fabs(e0[0])) && fabs(e1[1]) < 1e-12;
  mitdvp_destroy(eng);
  const int ok2 = two_states() == 0;
  printf("%s\n", ok && ok2 ? "C-HOST OK" : "C-HOST FAILED");
  return ok && ok2 ? 0 : 1;
}
