// small_site.hip -- the small-bond kernel family (SURVEY 8d "C2 and small-D": a site is a few hundred KB,
// every step of the reference is latency bound; north_star: "SIL local propagator fused into one kernel
// launch per site").
//
// Reference functions replaced (paths relative to /root/reference/pytdscf):
//   multiplyH_MPS_direct_MPO._op_lcr_dot / .dot     _contraction.py:1038-1243   (H_eff apply)
//   multiplyK_MPS_direct_MPO._op_lr_dot / .dot      _contraction.py:1297-1407   (K_eff apply)
//   contract_with_site_mpo                          _contraction.py:148-397     (environment update)
//   short_iterative_lanczos / short_iterative_arnoldi  _integrator.py:453-655, :287-432
//     incl. the host work the reference does per iteration: two device->host syncs (:553-554),
//     eigh_tridiagonal / eig + solve of the projected matrix (:590, :618, :402-408), vstack (:568).
//
// ONE kernel, k_small_site:
//   * a workgroup owns (slab a, chunk sc of the contracted right index s); the three stages of the
//     contraction chain run out of LDS (operands A_a, W2, R staged once per launch, X and Y never leave
//     LDS); the chunk partials of a slab are summed in a fixed order by the owners of the output range;
//   * EXP mode wraps the Krylov loop around it: per iteration TWO grid-wide exchanges of a few doubles
//     per workgroup (the projections <v_j|H v_l>, which are linear in the chunk partials and therefore
//     travel with them; then the norm of the new vector), the k x k exponential (scaling and squaring,
//     degree-20 Taylor: the algorithm of small_linalg.h::expm_col0) evaluated redundantly by every
//     workgroup, the reference's convergence test and the final linear combination -- no host involvement.
//     The basis is kept unnormalised (u_j) with the factors 1/beta_j applied on the fly: v_j = u_j * invb_j
//     is the same floating-point number wherever it is formed, so results equal the multi-launch path's.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md, "Workgroup dispatch ... visibility", valid forms, row 1):
// every byte another workgroup reads is stored with an agent-scope (sc1) store and read with an agent-scope
// (sc1) load; every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the
// ONE arrival atomic per workgroup; one lane polls the counter and the other waves proceed behind a
// workgroup barrier.  The grid never exceeds the CU count and a workgroup takes more than half a CU's LDS,
// so all workgroups are resident; every poll loop is bounded (2 s) and raises a sticky error instead of
// hanging.  Reductions are summed in a fixed order: all workgroups obtain bit-identical scalars and take
// identical branches.
#include "small_site.h"
#include "krylov_dev.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/mitdvp.h"

namespace mitdvp {
namespace {

// Threads per workgroup.  Rounds 2-4 ran 1024 (16 waves: "four per SIMD hide the LDS / L2 latencies of the short dependent
// chains") -- at a register budget of 128 per thread, which this kernel exceeds: 125 VGPRs spilled, 380 B of scratch per
// lane (hipcc -Rpass-analysis=kernel-resource-usage).  With 512 threads (8 waves, 256 registers each: 233 used, no scratch)
// the same code runs C2 at 237 instead of 208 sweeps/s and the ensembles at 468 / 806 / 882 / 1134 instead of 407 / 693 /
// 782 / 995 (2 / 4 / 8 / 16 replicas, same box: profiles/r05_ss_threads_ab.txt).  make variantf DEFS=-DMITDVP_SS_THREADS=1024
// builds the old form for A/B runs.
#ifndef MITDVP_SS_THREADS
#define MITDVP_SS_THREADS 512
#endif
constexpr int SS_THREADS = MITDVP_SS_THREADS;
static_assert(SS_THREADS == 1024 || SS_THREADS == 512, "small-site workgroups have 16 or 8 waves");
constexpr int SS_WAVES = SS_THREADS / 64;
constexpr int SS_PAYMAX = 2 * MAXK + 2;  // doubles one workgroup contributes to an exchange
constexpr int SS_MAXG = 256;
constexpr int SS_NGR = 2 * SS_PAYMAX;  // granules per workgroup and exchange
constexpr double SS_EPS = 1e-12;  // _integrator.py:22

#define SS_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ double ld_sh(const double* p) { return __hip_atomic_load(p, SS_RLX); }
__device__ __forceinline__ void st_sh(double* p, double v) { __hip_atomic_store(p, v, SS_RLX); }
__device__ __forceinline__ zc ldz_sh(const zc* p) {
  const double* q = reinterpret_cast<const double*>(p);
  return make_double2(ld_sh(q), ld_sh(q + 1));
}
__device__ __forceinline__ void stz_sh(zc* p, zc v) {
  double* q = reinterpret_cast<double*>(p);
  st_sh(q, v.x);
  st_sh(q + 1, v.y);
}

struct SsArgs {
  SmallChain c;
  int mode;
  const zc* v;  // APPLY: input vector
  zc* out;      // APPLY: output (contiguous (na, ni, nr))
  zc* x;        // EXP: in/out vector
  zc* U;        // EXP: unnormalised Krylov vectors, U[j] at j*N (j >= 1; vector 0 is x)
  zc* P;        // chunk partials [nsc][N]
  zc shift;
  int add_shift;
  SmallExp e;
  unsigned epoch0;  // tags of this launch's exchanges are epoch0 + 1, + 2, ...
  unsigned* abort_w;
  unsigned* err_w;
  unsigned long long* gran;
  long long* stats;
  int* kprev;
  long long* trace;  // debugging: workgroup 0 stamps s_memrealtime at its phase boundaries (nullptr = off)
};

struct Sync {
  unsigned long long* gran;
  unsigned* abort_w;
  int G, wg;
  unsigned epoch;
};

// fixed-order sum of the SS_WAVES per-wave partials p[0..SS_WAVES)
__device__ __forceinline__ double wtree(const double* p) {
  // written out (halving tree: i += i + 8, then 4, 2, 1): as loops over a local array hipcc kept the array in scratch
  double b0, b1, b2, b3, b4, b5, b6, b7;
  if constexpr (SS_WAVES == 16) {
    b0 = p[0] + p[8]; b1 = p[1] + p[9]; b2 = p[2] + p[10]; b3 = p[3] + p[11];
    b4 = p[4] + p[12]; b5 = p[5] + p[13]; b6 = p[6] + p[14]; b7 = p[7] + p[15];
  } else {
    b0 = p[0]; b1 = p[1]; b2 = p[2]; b3 = p[3]; b4 = p[4]; b5 = p[5]; b6 = p[6]; b7 = p[7];
  }
  const double c0 = b0 + b4, c1 = b1 + b5, c2 = b2 + b6, c3 = b3 + b7;
  const double d0 = c0 + c2, d1 = c1 + c3;
  return d0 + d1;
}

__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Grid-wide exchange: every workgroup contributes pay[0..npay) (LDS), every workgroup receives the
// element-wise sums over all workgroups in red[0..npay) (LDS), summed in the same order everywhere.
// Also orders all sc1 stores issued before it against all sc1 loads issued after it, grid-wide.
//
// The data is the flag (MI355X_MICROARCH.md price list, "allgather"; cdna_hip_programming.md Guideline 16 R2):
// every double travels as two 8-byte granules {tag = epoch, 32 payload bits}, each written by ONE
// agent-scope store; thread t reads the granules of workgroup t until all carry this epoch's tag.  No
// counter, no separate flag: one store->load round trip.  Granules are laid out [index][workgroup] so a
// wave's poll is one coalesced request; two buffers alternate (a writer can be at most one epoch ahead of
// the slowest reader).  Returns false (uniformly) when a wait timed out or the abort flag was raised.
__device__ bool ss_exchange(Sync& s, const double* pay, int npay, double* red, double* wsh /*[SS_WAVES*SS_PAYMAX]*/) {
  s.epoch += 1;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ng = npay > 0 ? 2 * npay : 1;
  unsigned long long* base = s.gran + (size_t)(s.epoch & 1u) * SS_NGR * SS_MAXG;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its sc1 stores of vector data have landed
  __syncthreads();
  if (tid < ng) {
    unsigned half = 0u;
    if (npay > 0) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(pay[tid >> 1]);
      half = (tid & 1) ? (unsigned)(b >> 32) : (unsigned)b;
    }
    __hip_atomic_store(base + (size_t)tid * SS_MAXG + s.wg, ((unsigned long long)s.epoch << 32) | half, SS_RLX);
  }
  int ok = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  for (int g0 = 0; g0 < ng; g0 += 8) {
    unsigned long long x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = 0ull;
    if (tid < s.G && ok) {
      unsigned spins = 0;
      for (;;) {
        bool ready = true;
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (g0 + q < ng) {
            x[q] = __hip_atomic_load(base + (size_t)(g0 + q) * SS_MAXG + tid, SS_RLX);
            ready = ready && (unsigned)(x[q] >> 32) == s.epoch;
          }
        if (ready) break;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u) {
          if (__hip_atomic_load(s.abort_w, SS_RLX) != 0u) { ok = 0; break; }
          if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s: a workgroup is not resident
            __hip_atomic_store(s.abort_w, 1u, SS_RLX);
            ok = 0;
            break;
          }
        }
      }
    }
    if (npay > 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (g0 + 2 * q < ng) {
          double v = 0.0;
          if (tid < s.G) v = __longlong_as_double((long long)((x[2 * q] & 0xffffffffull) | (x[2 * q + 1] << 32)));
          v = wave_sum64(v);
          if (lane == 0 && w < 4) wsh[(g0 / 2 + q) * 4 + w] = v;  // G <= 256: waves 0..3 hold the workgroups
        }
    }
  }
  if (!__syncthreads_and(ok)) return false;
  if (npay > 0) {
    if (tid < npay) red[tid] = (wsh[tid * 4 + 0] + wsh[tid * 4 + 1]) + (wsh[tid * 4 + 2] + wsh[tid * 4 + 3]);
    __syncthreads();
  }
  return true;
}

// C(M x N) = A(M x K) * B(K x N), all row-major in LDS.  The operands are tiny (a few hundred outputs,
// K of a few dozen): the time goes into the dependent chain of K multiply-adds behind LDS latency, so K is
// split over KS adjacent lanes (KS = the largest power of two that still leaves every thread an output)
// and the KS partial sums are combined with cross-lane adds in a fixed order.
// all-reduce over groups of KS (<= 16) adjacent lanes with DPP moves (a cross-lane add costs one VALU
// issue; the LDS-crossbar shuffle would cost a round trip per step); the pairing order is fixed
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ks_allreduce(double v, int KS) {
  if (KS >= 2) v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  if (KS >= 4) v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  if (KS >= 8) v = dpp_add<0x141>(v);  // row_half_mirror: the other quad of the same 8 lanes
  if (KS >= 16) v = dpp_add<0x140>(v); // row_mirror: the other half of the same 16 lanes
  return v;
}

// C = scl * A B.  (Measured and dropped, round 2: a 2 x 2 block of outputs per thread -- half the LDS reads per
// multiply-add, but four times the cross-lane reduction work: the stages got 25 % slower.  At 16 waves per CU this
// kernel is bound by VALU issue slots, ~500 instructions per wave and stage of which the multiply-adds are a quarter.
// The same products as 16 x 16 MFMA tiles (one wave per tile, zero-padded edges): equal stage times (2.8 / 3.2 / 2.5 us) --
// four or five tiles keep four or five of the sixteen waves busy on a dependent read -> MFMA chain -- and the second code
// path cost the kernel 6 % through register spills: dropped as well.)
__device__ __forceinline__ void lds_gemm(const zc* __restrict__ A, int lda, const zc* __restrict__ B, int ldb,
                                         zc* __restrict__ C, int ldc, int M, int N, int K, double scl) {
  const int mn = M * N;
  int KS = 1, lg = 0;
  while (KS < 16 && mn * KS * 2 <= SS_THREADS && KS * 2 <= K) { KS *= 2; ++lg; }
  const int per = SS_THREADS >> lg;
  const int ks = threadIdx.x & (KS - 1);
  const int npass = (mn + per - 1) / per;
  for (int ps = 0; ps < npass; ++ps) {
    const int o = ps * per + (threadIdx.x >> lg);
    const bool valid = o < mn;
    const int oo = valid ? o : mn - 1;
    const int m = oo / N, n = oo - m * N;
    const zc* ap = A + m * lda;
    const zc* bp = B + n;
    // four k-steps per batch: all eight LDS reads are issued before the first multiply-add (hipcc otherwise
    // waits for every pair), and the four products feed independent accumulators
    double r0 = 0.0, r1 = 0.0, i0 = 0.0, i1 = 0.0;
    int k = ks;
    for (; k + 3 * KS < K; k += 4 * KS) {
      zc av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { av[u] = ap[k + u * KS]; bv[u] = bp[(k + u * KS) * ldb]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        r0 = fma(av[u].x, bv[u].x, r0); r1 = fma(-av[u].y, bv[u].y, r1);
        i0 = fma(av[u].x, bv[u].y, i0); i1 = fma(av[u].y, bv[u].x, i1);
      }
    }
    for (; k < K; k += KS) {
      const zc a = ap[k], b = bp[k * ldb];
      r0 = fma(a.x, b.x, r0); r1 = fma(-a.y, b.y, r1);
      i0 = fma(a.x, b.y, i0); i1 = fma(a.y, b.x, i1);
    }
    const double re = ks_allreduce(r0 + r1, KS), im = ks_allreduce(i0 + i1, KS);
    if (valid && ks == 0) C[m * ldc + n] = make_double2(re * scl, im * scl);
  }
}

// first column of exp(T) for the k x k matrix T (LDS, row-major, ld = k), by the whole workgroup:
// scaling and squaring (|T / 2^s|_1 <= 1/2) around the degree-16 Taylor polynomial, remainder
// 0.5^17 / 17! = 2e-20 (small_linalg.h::expm_col0 is the host twin, summed to degree 20), evaluated in
// Paterson-Stockmeyer form with the powers A^2, A^3, A^4:
//   p(A) = B0 + A^4 (B1 + A^4 (B2 + A^4 (B3 + A^4 / 16!))),  Bi = sum_{r<4} A^r / (4i + r)!
// = 3 + 4 products instead of 16.  Tm is destroyed; M2, M3, M4, Pm, Qm are k x k scratch.
__device__ void ss_expm_col0(zc* Tm, zc* M2, zc* M3, zc* M4, zc* Pm, zc* Qm, int k, zc* coef, double* wsh) {
  const int tid = threadIdx.x, kk = k * k;
  if (k == 1) {
    if (tid == 0) {
      const zc z = Tm[0];
      const double e = exp(z.x);
      coef[0] = make_double2(e * cos(z.y), e * sin(z.y));
    }
    __syncthreads();
    return;
  }
  if (tid < k) {  // 1-norm: max column sum
    double s = 0.0;
    for (int i = 0; i < k; ++i) { const zc z = Tm[i * k + tid]; s += sqrt(z.x * z.x + z.y * z.y); }
    wsh[tid] = s;
  }
  __syncthreads();
  double nrm = 0.0;
  for (int j = 0; j < k; ++j) nrm = fmax(nrm, wsh[j]);
  int sq = 0;
  while (nrm > 0.5 && sq < 60) { nrm *= 0.5; ++sq; }
  const double sc = ldexp(1.0, -sq);
  for (int t = tid; t < kk; t += SS_THREADS) { zc z = Tm[t]; z.x *= sc; z.y *= sc; Tm[t] = z; }
  __syncthreads();
  lds_gemm(Tm, k, Tm, k, M2, k, k, k, k, 1.0);
  __syncthreads();
  lds_gemm(M2, k, Tm, k, M3, k, k, k, k, 1.0);
  lds_gemm(M2, k, M2, k, M4, k, k, k, k, 1.0);
  __syncthreads();
  // inverse factorials 1/n!, n = 0..16
  constexpr double F[17] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                            1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0, 1.0 / 87178291200.0,
                            1.0 / 1307674368000.0, 1.0 / 20922789888000.0};
  auto bcoef = [&](int i, int t) -> zc {  // element t of B_i
    const zc a1 = Tm[t], a2 = M2[t], a3 = M3[t];
    const double one = (t / k == t % k) ? 1.0 : 0.0;
    return make_double2(F[4 * i] * one + F[4 * i + 1] * a1.x + F[4 * i + 2] * a2.x + F[4 * i + 3] * a3.x,
                        F[4 * i + 1] * a1.y + F[4 * i + 2] * a2.y + F[4 * i + 3] * a3.y);
  };
  for (int t = tid; t < kk; t += SS_THREADS) {  // P = B3 + A^4 / 16!
    const zc b = bcoef(3, t), a4 = M4[t];
    Pm[t] = make_double2(b.x + F[16] * a4.x, b.y + F[16] * a4.y);
  }
  __syncthreads();
  for (int i = 2; i >= 0; --i) {  // P <- B_i + A^4 P
    lds_gemm(M4, k, Pm, k, Qm, k, k, k, k, 1.0);
    __syncthreads();
    for (int t = tid; t < kk; t += SS_THREADS) {
      const zc b = bcoef(i, t), q = Qm[t];
      Pm[t] = make_double2(b.x + q.x, b.y + q.y);
    }
    __syncthreads();
  }
  for (int s = 0; s < sq; ++s) {  // E <- E E
    lds_gemm(Pm, k, Pm, k, Qm, k, k, k, k, 1.0);
    __syncthreads();
    for (int t = tid; t < kk; t += SS_THREADS) Pm[t] = Qm[t];
    __syncthreads();
  }
  if (tid < k) coef[tid] = Pm[tid * k];
  __syncthreads();
}

// coef = first column of exp(scale * T_k) for the TRIDIAGONAL T of a Lanczos recurrence (diagonal alpha, off-diagonal
// beta), by ONE WAVE with the vector in registers: lane q holds row q of T / 2^s and entry q of the vector, a product
// T p is two wave shifts and three complex multiply-adds, exp(T / 2^s) e_0 is the degree-20 Taylor sum (|T / 2^s|_1 <= 1:
// remainder 1 / 21! = 2e-20) applied 2^s times.  No LDS, no barrier, no k x k products: 20 dependent steps of ~50 cycles
// where ss_expm_col0 takes seven k x k products behind workgroup barriers (measured in k_small_site at C2: 14.2 us per
// inspected iteration, two per local exponential).  Every wave of the workgroup may run it redundantly (same
// instructions, same bits): all then know s, which decides uniformly whether this form is used (s <= SS_VEC_SMAX; for
// larger norms the 2^s repetitions cost more than squaring the matrix).  Returns s; coef is written by the caller's wave 0.
constexpr int SS_VEC_SMAX = 3;
__device__ __forceinline__ double wave_shr1(double v) {  // lane q <- lane q - 1 (lane 0 <- 0)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double v) {  // lane q <- lane q + 1 (lane 63 <- 0)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
// compute: only the wave(s) that pass true run the Taylor sum (the scaling exponent is returned to every caller).  All 16
// waves of a workgroup running it side by side shared four SIMDs: ~8 us per call in the MITDVP_SS_TRACE timelines, 2.4 us
// with one wave (the others wait at the barrier that follows anyway) and 1 / n as literals of the unrolled sum.
__device__ __forceinline__ int wave_expm_tridiag(const zc* alpha, const double* beta, zc scale, int k, bool real_alpha,
                                                 zc& out, bool compute = true) {
  const int q = threadIdx.x & 63;
  zc a = make_double2(0.0, 0.0), bl = a, bu = a;
  if (q < k) {
    zc al = alpha[q];
    if (real_alpha) al.y = 0.0;
    a = make_double2(scale.x * al.x - scale.y * al.y, scale.x * al.y + scale.y * al.x);
    if (q + 1 < k) { const double b = beta[q]; bu = make_double2(scale.x * b, scale.y * b); }
    if (q > 0) { const double b = beta[q - 1]; bl = make_double2(scale.x * b, scale.y * b); }
  }
  // column q of T: T[q][q] = a, T[q-1][q] = b_{q-1}, T[q+1][q] = b_q
  double nrm = wave_max64(sqrt(a.x * a.x + a.y * a.y) + sqrt(bl.x * bl.x + bl.y * bl.y) + sqrt(bu.x * bu.x + bu.y * bu.y));
  int s = 0;
  while (nrm > 1.0 && s < 60) { nrm *= 0.5; ++s; }
  if (s > SS_VEC_SMAX || !compute) return s;
  const double sc = ldexp(1.0, -s);
  a.x *= sc; a.y *= sc; bl.x *= sc; bl.y *= sc; bu.x *= sc; bu.y *= sc;
  zc y = make_double2(q == 0 ? 1.0 : 0.0, 0.0);
  for (int rep = 0; rep < (1 << s); ++rep) {
    zc p = y, acc = y;
#pragma unroll
    for (int n = 1; n <= 20; ++n) {
      const zc pm = make_double2(wave_shr1(p.x), wave_shr1(p.y));  // p_{q-1}
      const zc pp = make_double2(wave_shl1(p.x), wave_shl1(p.y));  // p_{q+1}
      // (T p)_q = T[q][q-1] p_{q-1} + T[q][q] p_q + T[q][q+1] p_{q+1}, T[q][q-1] = b_{q-1}, T[q][q+1] = b_q
      double re = a.x * p.x - a.y * p.y, im = a.x * p.y + a.y * p.x;
      re = fma(bl.x, pm.x, re); re = fma(-bl.y, pm.y, re);
      im = fma(bl.x, pm.y, im); im = fma(bl.y, pm.x, im);
      re = fma(bu.x, pp.x, re); re = fma(-bu.y, pp.y, re);
      im = fma(bu.x, pp.y, im); im = fma(bu.y, pp.x, im);
      const double inv = 1.0 / (double)n;
      p = make_double2(re * inv, im * inv);
      acc.x += p.x; acc.y += p.y;
    }
    y = acc;
  }
  out = y;
  return s;
}

// ---------------------------------------------------------------------------
// krylov_dev.h: the Ritz step of the MULTI-launch Krylov loop on the device (one workgroup).  Reference semantics:
// _iter_info warm-up (_integrator.py:178-186) is applied by the host (it decides WHICH iterations are inspected, a
// function of k_prev only); here: the scalars of iterations [q0, l], exhaustion (:569 / :392), the projected exponential
// (:581-637 / :397-409; real alpha -> the symmetric tridiagonal T, :617-621), and the bookkeeping of the previous Ritz
// coefficients for the convergence test (:638-651) that kry_diff finishes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double kr_sum256(const double* __restrict__ p, int stride, int lane) {
  // one wave: fixed-order sum of NPART = 256 partials (stride in doubles between consecutive partials)
  const double v = (p[(size_t)lane * stride] + p[(size_t)(lane + 64) * stride]) +
                   (p[(size_t)(lane + 128) * stride] + p[(size_t)(lane + 192) * stride]);
  return wave_sum64(v);  // lane 0
}

__global__ __launch_bounds__(SS_THREADS) void k_kry_ritz(KryRitzArgs a) {
  static_assert(NPART == 256, "kr_sum256 is written for 256 partials");
  __shared__ zc mats[6 * MAXK * MAXK];
  __shared__ zc coef_s[MAXK];
  __shared__ double wsh[SS_WAVES * SS_PAYMAX];
  __shared__ int ctl[4];
  KryDev* st = a.st;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (!a.first && st->state != KRY_RUNNING) return;  // closed by an earlier check: the host has queued one iteration too many
  for (int q = a.q0; q <= a.l; ++q) {
    const int nz = a.lanczos ? 1 : q + 1;
    for (int j = w; j <= nz; j += SS_WAVES) {
      if (j < nz) {
        const double* p = reinterpret_cast<const double*>(a.lanczos ? a.alpha_p + (size_t)q * NPART
                                                                   : a.h_p + ((size_t)q * MAXK + j) * NPART);
        const double re = kr_sum256(p, 2, lane), im = kr_sum256(p + 1, 2, lane);
        if (lane == 0) {
          if (a.lanczos) st->alpha[q] = make_double2(re, im);
          else st->hess[j * MAXK + q] = make_double2(re, im);
        }
      } else {
        const double s = kr_sum256(a.nrm_p + (size_t)q * NPART, 1, lane);
        if (lane == 0) st->beta[q] = sqrt(s);
      }
    }
  }
  if (a.first && w == SS_WAVES - 1) {
    const double s = a.beta0_p ? kr_sum256(a.beta0_p, 1, lane) : 1.0;
    if (lane == 0) st->beta0 = a.beta0_p ? sqrt(s) : 1.0;
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {
    st->deferred = a.deferred;
    if (a.deferred) {  // unnormalised basis: invb_q = 1 / beta_{q-1} (1 below eps), alpha_q = invb_q (^2) * raw dot
      if (a.first) st->invb[0] = 1.0;
      for (int q = a.q0; q <= a.l; ++q) {
        const double f = st->invb[q];
        const double fa = a.orthodox ? f * f : f;
        st->alpha[q].x *= fa;
        st->alpha[q].y *= fa;
        const double b = st->beta[q];
        if (q + 1 < MAXK) st->invb[q + 1] = b >= a.eps ? 1.0 / b : 1.0;
      }
    }
    int ld = a.l, exhausted = 0;
    for (int q = a.q0; q <= a.l; ++q) {
      const double b = st->beta[q];
      if (!a.lanczos && b > a.eps) st->hess[(q + 1) * MAXK + q] = make_double2(b, 0.0);
      if (b < a.eps || q + 1 == a.nsize) { ld = q; exhausted = 1; break; }
    }
    int real_alpha = 1;
    if (a.lanczos)
      for (int q = 0; q <= ld; ++q)
        if (fabs(st->alpha[q].y) > 1e-10) real_alpha = 0;
    ctl[0] = ld + 1;
    ctl[1] = exhausted;
    ctl[2] = real_alpha;
  }
  __threadfence_block();
  __syncthreads();
  const int k = ctl[0];
  const bool exhausted = ctl[1] != 0, real_alpha = ctl[2] != 0;
  zc* Tm = mats;
  zc* M2 = Tm + MAXK * MAXK;
  zc* M3 = M2 + MAXK * MAXK;
  zc* M4 = M3 + MAXK * MAXK;
  zc* Pm = M4 + MAXK * MAXK;
  zc* Qm = Pm + MAXK * MAXK;
  bool vec_done = false;
  if (a.lanczos && k > 1) {  // tridiagonal: vector form in one wave (wave_expm_tridiag)
    zc cq;
    const int sq_ = wave_expm_tridiag(st->alpha, st->beta, a.scale, k, real_alpha, cq, tid < 64);
    if (sq_ <= SS_VEC_SMAX) {
      if (tid < k) coef_s[tid] = cq;
      __syncthreads();
      vec_done = true;
    }
  }
  if (!vec_done) {
    for (int t = tid; t < k * k; t += SS_THREADS) {
      const int i = t / k, j = t - i * k;
      zc z = make_double2(0.0, 0.0);
      if (a.lanczos) {
        if (i == j) { z = st->alpha[i]; if (real_alpha) z.y = 0.0; }
        else if (i == j + 1) z = make_double2(st->beta[j], 0.0);
        else if (j == i + 1) z = make_double2(st->beta[i], 0.0);
      } else {
        if (i <= j + 1) z = st->hess[i * MAXK + j];
      }
      Tm[t] = make_double2(a.scale.x * z.x - a.scale.y * z.y, a.scale.x * z.y + a.scale.y * z.x);
    }
    __syncthreads();
    ss_expm_col0(Tm, M2, M3, M4, Pm, Qm, k, coef_s, wsh);
  }
  const int have_prev = a.first ? 0 : st->have_prev;
  const int prev_len = a.first ? 0 : st->prev_len;
  if (tid < k) {
    const zc c = coef_s[tid];
    st->coef[tid] = c;
    if (!exhausted) {
      if (have_prev) {
        zc d = c;
        if (tid < prev_len) { const zc o = st->cprev[tid]; d.x -= o.x; d.y -= o.y; }
        if (a.deferred) { const double f = st->invb[tid]; d.x *= f; d.y *= f; }  // kry_diff combines the STORED vectors
        st->dcoef[tid] = d;
      } else {
        st->cprev[tid] = c;
      }
    }
  }
  if (tid == 0) {
    st->k = k;
    if (a.first) { st->ticket = 0u; st->err = -1.0; }
    if (exhausted) {
      st->state = KRY_EXHAUSTED;
      st->need_diff = 0;
    } else {
      st->state = KRY_RUNNING;
      st->need_diff = have_prev ? 1 : 0;
      if (!have_prev) { st->have_prev = 1; st->prev_len = k; }
    }
  }
}

// LDS carve (units: zc unless noted); the same arithmetic runs on the host in small_chain_lds
struct Carve {
  size_t As, Rs, Ws, Bs, Xs, Ys, Sg, misc, total;  // offsets in zc
};
__host__ __device__ inline Carve ss_carve(const SmallChain& c, bool exp_mode) {
  Carve k{};
  size_t o = 0;
  // A_a of the workgroup's slabs: all of them when they fit (a_resident), else one slot re-staged per slab
  k.As = o; o += (size_t)c.nc * c.nb * (c.a_resident && c.spw > 1 ? c.spw : 1);
  k.Rs = o; o += (size_t)c.nt * c.cs * c.nr;
  k.Ws = o; o += c.W2 ? (size_t)c.ni * c.nt * c.nc * c.nj : 0;
  k.Bs = o;
  size_t bs = (size_t)c.nb * c.nj * c.cs;
  if (exp_mode) bs = bs > (size_t)6 * MAXK * MAXK ? bs : (size_t)6 * MAXK * MAXK;  // also the k x k exponential's scratch
  o += bs;
  k.Xs = o; o += (size_t)c.nc * c.nj * c.cs;
  k.Ys = o; o += c.W2 ? (size_t)c.ni * c.nt * c.cs : 0;
  // this chunk's partial of the output slab; with a W stage it may reuse X's space (stage 3 reads Y and R only)
  if (c.W2 && (size_t)c.ni * c.nr <= (size_t)c.nc * c.nj * c.cs) k.Sg = k.Xs;
  else { k.Sg = o; o += (size_t)c.ni * c.nr; }
  k.misc = o;
  // misc: pay[SS_PAYMAX] red[SS_PAYMAX] wsh[SS_WAVES*SS_PAYMAX] (doubles) + alpha[MAXK] coef[MAXK] cprev[MAXK] (zc)
  //       + hess[(MAXK+1)*MAXK] (zc) + beta[MAXK] invb[MAXK+1] (doubles) + ints
  o += (size_t)((2 + SS_WAVES) * SS_PAYMAX + 1) / 2 + 3 * MAXK + (size_t)(MAXK + 1) * MAXK + (2 * MAXK + 2) / 2 + 8;
  k.total = o;
  return k;
}

__global__ __launch_bounds__(SS_THREADS) void k_small_site(SsArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  zc* sm = reinterpret_cast<zc*>(smem);
  const SmallChain& c = g.c;
  const bool exp_mode = g.mode == SS_MODE_EXP;
  const Carve cv = ss_carve(c, exp_mode);
  zc* As = sm + cv.As;
  zc* Rs = sm + cv.Rs;
  zc* Ws = sm + cv.Ws;
  zc* Bs = sm + cv.Bs;
  zc* Xs = sm + cv.Xs;
  zc* Ys = c.W2 ? sm + cv.Ys : Xs;
  zc* Sg = sm + cv.Sg;
  double* pay = reinterpret_cast<double*>(sm + cv.misc);
  double* red = pay + SS_PAYMAX;
  double* wsh = red + SS_PAYMAX;
  zc* alpha = reinterpret_cast<zc*>(wsh + SS_WAVES * SS_PAYMAX);
  zc* coef = alpha + MAXK;
  zc* cprev = coef + MAXK;
  zc* hess = cprev + MAXK;  // (MAXK+1) x MAXK, row-major, ld = MAXK
  double* beta = reinterpret_cast<double*>(hess + (MAXK + 1) * MAXK);
  double* invb = beta + MAXK;
  int* ctl = reinterpret_cast<int*>(invb + MAXK + 1);  // [0] exchange verdict, [1] action, [2] kdim, [3] next_unread

  const int tid = threadIdx.x;
  const int nsc = c.nsc, cs = c.cs;
  // a workgroup owns chunk sc of the contracted index for spw consecutive slabs (spw = 1: one slab -- the layout of a
  // full-size grid; spw > 1: an engine confined to a slice of the chip, ensemble mode): R, W and the stage-1 operand B of
  // the chunk are shared by its slabs, A_a is per slab (all resident in LDS when they fit, else re-staged per slab)
  const int spw = c.spw > 1 ? c.spw : 1;
  const int ag = blockIdx.x / nsc, sc = blockIdx.x - ag * nsc;
  const int a_first = ag * spw;
  const int nsl = min(spw, c.na - a_first);  // >= 1 by construction of the plan
  const int s0 = sc * cs;
  const int csl = min(cs, c.ns - s0);  // >= 1 by construction of the plan
  const int slab = c.ni * c.nr;
  const long N = (long)c.na * slab;
  const int rg = (slab + nsc - 1) / nsc;
  const int r_lo = min(sc * rg, slab), r_hi = min((sc + 1) * rg, slab);  // this chunk's share of an output slab
  const int nAone = c.nc * c.nb;

  Sync sy{g.gran, g.abort_w, (int)gridDim.x, (int)blockIdx.x, g.epoch0};
  int ntr = 1;
  auto stamp = [&](int label) {
    if (g.trace && blockIdx.x == 0 && tid == 0 && ntr < 250) {
      g.trace[2 * ntr] = (long long)__builtin_amdgcn_s_memrealtime();
      g.trace[2 * ntr + 1] = label;
      g.trace[512 + ntr] = (long long)__builtin_amdgcn_s_memtime();
      ntr += 1;
      g.trace[0] = ntr;
    }
  };
  stamp(0);

  // ---- operands that do not change during the launch -> LDS -------------------------------
  // Every thread's first elements of the three operands are loaded before any of them is stored: one exposed L2 latency
  // at the start of a launch instead of one per loop trip (As 1, Rs 2, Ws 4 trips at C2: ~4 of the ~10 us a launch
  // spent before its first product in the MITDVP_SS_TRACE timelines).  Longer operands finish in the plain loops below.
  {
    const int nA = c.a_resident ? nAone * nsl : 0, kk = c.nt * cs, nR = kk * c.nr, nW = c.W2 ? c.ni * c.nt * c.nc * c.nj : 0;
    auto ldA = [&](int t) __attribute__((always_inline)) -> zc {
      const int q = t / nAone, tt = t - q * nAone;
      const int cc = tt / c.nb, b = tt - cc * c.nb;
      zc z = c.A[(long)(a_first + q) * c.sAa + (long)cc * c.sAc + (long)b * c.sAb];
      if (c.conjA) z.y = -z.y;
      return z;
    };
    auto ldR = [&](int t) __attribute__((always_inline)) -> zc {
      // r fastest in LDS; pick the global order that keeps loads coalesced for the common layouts
      const int k = t / c.nr, r = t - k * c.nr;
      const int tt = k / cs, sl = k - tt * cs;
      zc z = make_double2(0.0, 0.0);
      if (sl < csl) z = c.R[(long)r * c.sRr + (long)tt * c.sRt + (long)(s0 + sl) * c.sRs];
      return z;
    };
    zc za = make_double2(0.0, 0.0), zr[2], zw[4];
    if (tid < nA) za = ldA(tid);
#pragma unroll
    for (int u = 0; u < 2; ++u) { zr[u] = make_double2(0.0, 0.0); if (tid + u * SS_THREADS < nR) zr[u] = ldR(tid + u * SS_THREADS); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { zw[u] = make_double2(0.0, 0.0); if (tid + u * SS_THREADS < nW) zw[u] = c.W2[tid + u * SS_THREADS]; }
    if (tid < nA) As[tid] = za;
#pragma unroll
    for (int u = 0; u < 2; ++u) if (tid + u * SS_THREADS < nR) Rs[tid + u * SS_THREADS] = zr[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (tid + u * SS_THREADS < nW) Ws[tid + u * SS_THREADS] = zw[u];
    for (int t = tid + SS_THREADS; t < nA; t += SS_THREADS) As[t] = ldA(t);
    for (int t = tid + 2 * SS_THREADS; t < nR; t += SS_THREADS) Rs[t] = ldR(t);
    for (int t = tid + 4 * SS_THREADS; t < nW; t += SS_THREADS) Ws[t] = c.W2[t];
  }
  stamp(3);

  // stage-1 operand of this chunk: Bs[b][(j, sl)] = scale * vec(b, j, s0 + sl)
  auto load_B = [&](const zc* vec, bool shared, double scl) {
    const int ncol = c.nj * cs, total = c.nb * ncol;
    for (int t0 = 0; t0 < total; t0 += 4 * SS_THREADS) {  // all loads of a batch in flight before the first use
      zc z[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * SS_THREADS + tid;
        z[u] = make_double2(0.0, 0.0);
        if (t < total) {
          const int b = t / ncol, q = t - b * ncol;
          const int j = q / cs, sl = q - j * cs;
          if (sl < csl) {
            const zc* p = vec + (long)b * c.sBb + (long)j * c.sBj + (long)(s0 + sl) * c.sBs;
            z[u] = shared ? ldz_sh(p) : *p;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * SS_THREADS + tid;
        if (t < total) Bs[t] = make_double2(z[u].x * scl, z[u].y * scl);
      }
    }
  };

  // the three stages; this chunk's partial of out[a][:][:] ends up in Sg (LDS)
  auto chain = [&](int q) {
    const zc* Aq = As + (c.a_resident ? (size_t)q * nAone : 0);
    if (!c.a_resident) {  // A of this slab -> LDS (the previous slab's stage 1 is long over: two barriers ago)
      // (round 5, measured and dropped: requesting slab q + 1's A into registers before this slab's stages and storing it
      // into a second slot afterwards -- the registers held across the three stages added spills to a kernel that is
      // already spill-bound: 8 x 32 CUs 812 -> 634 sweeps/s, 16 x 16 CUs 998 -> 756, profiles/r05_ensemble_probe.txt)
      for (int t = tid; t < nAone; t += SS_THREADS) {
        const int cc = t / c.nb, b = t - cc * c.nb;
        zc z = c.A[(long)(a_first + q) * c.sAa + (long)cc * c.sAc + (long)b * c.sAb];
        if (c.conjA) z.y = -z.y;
        As[t] = z;
      }
    }
    __syncthreads();
    stamp(20);
    lds_gemm(Aq, c.nb, Bs, c.nj * cs, Xs, c.nj * cs, c.nc, c.nj * cs, c.nb, 1.0);
    __syncthreads();
    stamp(21);
    if (c.W2) {
      lds_gemm(Ws, c.nc * c.nj, Xs, cs, Ys, cs, c.ni * c.nt, cs, c.nc * c.nj, 1.0);
      __syncthreads();
    }
    stamp(22);
    lds_gemm(Ys, c.nt * cs, Rs, c.nr, Sg, c.nr, c.ni, c.nr, c.nt * cs, 1.0);
    __syncthreads();
    stamp(23);
  };

  zc* Pchunk = g.P + (size_t)sc * N;  // this chunk's partials; slab a at + a * slab

  if (!exp_mode) {
    load_B(g.v, false, 1.0);
    stamp(1);
    for (int sl = 0; sl < nsl; ++sl) {
      const long a = a_first + sl;
      chain(sl);
      for (int q = tid; q < slab; q += SS_THREADS) {
        zc v = Sg[q];
        if (g.add_shift && sc == 0) {  // the scalar term (coupleJ * ovlp, _contraction.py:1200-1216) rides on chunk 0
          const zc x = g.v[a * slab + q];
          v.x += g.shift.x * x.x - g.shift.y * x.y;
          v.y += g.shift.x * x.y + g.shift.y * x.x;
        }
        stz_sh(Pchunk + a * slab + q, v);
      }
    }
    stamp(2);
    if (!ss_exchange(sy, pay, 0, red, wsh)) {
      if (blockIdx.x == 0 && tid == 0) atomicMax(g.err_w, (unsigned)SS_ETIMEOUT);
      return;
    }
    for (int sl = 0; sl < nsl; ++sl) {
      const long eb = (long)(a_first + sl) * slab;
      for (long e = eb + r_lo + tid; e < eb + r_hi; e += SS_THREADS) {
        zc s = make_double2(0.0, 0.0);
        for (int q = 0; q < nsc; ++q) {
          const zc p = ldz_sh(g.P + (size_t)q * N + e);
          s.x += p.x; s.y += p.y;
        }
        g.out[e] = s;
      }
    }
    return;
  }

  // =====================================================================================
  // EXP mode: x <- exp(scale * Op) x   (short_iterative_lanczos / _arnoldi)
  // =====================================================================================
  const SmallExp& ex = g.e;
  const bool lanczos = ex.integrator == MITDVP_LANCZOS;
  const bool cn = ex.conserve_norm != 0;
  const long nsize = N;
  const int ndim = (int)min((long)ex.max_krylov, nsize);
  const int k_prev = g.kprev[ex.site];
  const int n_warm = (int)min(nsize, (long)min(max(0, k_prev - 2), 15));  // _iter_info, _integrator.py:178-186
  const zc scale = make_double2(ex.scale_re, ex.scale_im);

  // basis vector j at element e: v_j(e) = u_j(e) * invb[j], u_0 = x (input, plain loads)
  auto basis = [&](int j, long e) -> zc {
    zc z = j == 0 ? g.x[e] : ldz_sh(g.U + (size_t)j * N + e);
    const double f = invb[j];
    z.x *= f; z.y *= f;
    return z;
  };
  auto fail = [&](int code) {
    if (blockIdx.x == 0 && tid == 0) atomicMax(g.err_w, (unsigned)code);
  };

  // the elements this workgroup owns in the assembled vectors: its chunk's share [r_lo, r_hi) of each of its slabs
  auto own = [&](auto&& body) __attribute__((always_inline)) {
    for (int sl = 0; sl < nsl; ++sl) {
      const long eb = (long)(a_first + sl) * slab;
      for (long e = eb + r_lo + tid; e < eb + r_hi; e += SS_THREADS) body(e);
    }
  };

  // ---- _normalize (_integrator.py:189-203) ---------------------------------------------
  double beta0 = 1.0;
  if (!cn) {
    double s = 0.0;
    own([&](long e) { const zc z = g.x[e]; s += z.x * z.x + z.y * z.y; });
    s = wave_sum64(s);
    if ((tid & 63) == 0) wsh[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) pay[0] = wtree(wsh);
    __syncthreads();
    if (!ss_exchange(sy, pay, 1, red, wsh)) { fail(SS_ETIMEOUT); return; }
    beta0 = sqrt(red[0]);
    if (beta0 == 0.0) { fail(SS_EZERO); return; }
  }
  stamp(4);
  if (tid == 0) {
    invb[0] = 1.0 / beta0;
    ctl[3] = 0;  // next_unread
  }
  __syncthreads();
  stamp(5);

  bool have_prev = false;
  int prev_len = 0;
  int napply = 0;
  for (int l = 0; l < ndim; ++l) {
    // ---- sigma = Op v_l : chunk partials + the projections that are linear in them ----------
    stamp(10);
    load_B(l == 0 ? g.x : g.U + (size_t)l * N, l != 0, invb[l]);
    stamp(11);
    const int jlo = lanczos ? (ex.variant == 0 ? 0 : l) : 0;
    const int jhi = lanczos ? jlo : l;
    const int nd = jhi - jlo + 1;
    {
      const int lane = tid & 63, w = tid >> 6;
      // per-wave sums of the projections accumulate over the workgroup's slabs in wsh (each slot has ONE writer: lane 0
      // of its wave, which also clears it here)
      if (lane == 0)
        for (int j = 0; j < 2 * nd; ++j) wsh[j * SS_WAVES + w] = 0.0;
      for (int sl = 0; sl < nsl; ++sl) {
        const long ab = (long)(a_first + sl) * slab;
        chain(sl);
        // store this chunk's partial (with the scalar term on chunk 0) and keep the shifted values in Sg
        for (int q = tid; q < slab; q += SS_THREADS) {
          zc v = Sg[q];
          const long e = ab + q;
          if (g.add_shift && sc == 0) {  // (H + shift) v_l: the projections below must see the scalar term too
            const zc vl = basis(l, e);
            v.x += g.shift.x * vl.x - g.shift.y * vl.y;
            v.y += g.shift.x * vl.y + g.shift.y * vl.x;
            Sg[q] = v;
          }
          stz_sh(Pchunk + e, v);
        }
        // projections <v_j | partial>, four basis vectors per pass (registers: four accumulator pairs at a time)
        for (int j0 = 0; j0 < nd; j0 += 4) {
          double dre[4] = {0.0, 0.0, 0.0, 0.0}, dim_[4] = {0.0, 0.0, 0.0, 0.0};
          for (int q = tid; q < slab; q += SS_THREADS) {
            const zc v = Sg[q];
            const long e = ab + q;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j0 + j < nd) {
                const zc b = basis(jlo + j0 + j, e);  // conj(b) * v
                dre[j] += b.x * v.x + b.y * v.y;
                dim_[j] += b.x * v.y - b.y * v.x;
              }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j0 + j < nd) {
              const double r1 = wave_sum64(dre[j]), r2 = wave_sum64(dim_[j]);
              if (lane == 0) { wsh[(2 * (j0 + j)) * SS_WAVES + w] += r1; wsh[(2 * (j0 + j) + 1) * SS_WAVES + w] += r2; }
            }
        }
      }
      stamp(12);
      napply += 1;
      __syncthreads();
      if (tid < 2 * nd) pay[tid] = wtree(wsh + tid * SS_WAVES);
      __syncthreads();
      stamp(13);
      if (!ss_exchange(sy, pay, 2 * nd, red, wsh)) { fail(SS_ETIMEOUT); return; }
      stamp(14);
      if (tid == 0) {
        if (lanczos) alpha[l] = make_double2(red[0], red[1]);
        else
          for (int j = 0; j <= l; ++j) hess[j * MAXK + l] = make_double2(red[2 * j], red[2 * j + 1]);
      }
      __syncthreads();
    }
    // ---- own range: assemble sigma, orthogonalise, norm ---------------------------------------
    {
      double s = 0.0;
      const double bprev = l > 0 ? beta[l - 1] : 0.0;
      own([&](long e) {
        // v_l and v_{l-1} are requested before the chunk partials: their latencies overlap the partial loop's
        zc vl = l == 0 ? g.x[e] : ldz_sh(g.U + (size_t)l * N + e);
        zc vm = make_double2(0.0, 0.0);
        if (lanczos && l > 0) vm = l == 1 ? g.x[e] : ldz_sh(g.U + (size_t)(l - 1) * N + e);
        zc u = make_double2(0.0, 0.0);
        for (int q = 0; q < nsc; ++q) {
          const zc p = ldz_sh(g.P + (size_t)q * N + e);
          u.x += p.x; u.y += p.y;
        }
        { const double f = invb[l]; vl.x *= f; vl.y *= f; }
        if (lanczos) {  // _integrator.py:556-562
          const zc al = alpha[l];
          u.x -= al.x * vl.x - al.y * vl.y;
          u.y -= al.x * vl.y + al.y * vl.x;
          if (l > 0) {
            const double f = invb[l - 1];
            vm.x *= f; vm.y *= f;
            u.x -= bprev * vm.x;
            u.y -= bprev * vm.y;
          }
        } else {  // classical Gram-Schmidt against all vectors, _orth_step_np (:247-260)
          for (int j = 0; j <= l; ++j) {
            const zc h = hess[j * MAXK + l];
            const zc vj = j == l ? vl : basis(j, e);
            u.x -= h.x * vj.x - h.y * vj.y;
            u.y -= h.x * vj.y + h.y * vj.x;
          }
        }
        stz_sh(g.U + (size_t)(l + 1) * N + e, u);
        s += u.x * u.x + u.y * u.y;
      });
      s = wave_sum64(s);
      __syncthreads();
      if ((tid & 63) == 0) wsh[tid >> 6] = s;
      __syncthreads();
      if (tid == 0) pay[0] = wtree(wsh);
      __syncthreads();
      stamp(15);
      if (!ss_exchange(sy, pay, 1, red, wsh)) { fail(SS_ETIMEOUT); return; }
      stamp(16);
    }
    // ---- scalars; decide what this iteration inspects (_integrator.py:569-652, :392-430) --------
    if (tid == 0) {
      const double b = sqrt(red[0]);
      beta[l] = b;
      invb[l + 1] = b >= SS_EPS ? 1.0 / b : 1.0;  // exhausted Krylov space: the vector is left as it is
      if (!lanczos && b > SS_EPS) hess[(l + 1) * MAXK + l] = make_double2(b, 0.0);
      int act = 0, kd = 0;
      const bool last_possible = (l + 1 == nsize);
      if (!(l < n_warm && !last_possible && l + 1 < ndim)) {
        int ld = l;
        bool exhausted = false;
        for (int q = ctl[3]; q <= l; ++q)
          if (beta[q] < SS_EPS || q + 1 == nsize) { ld = q; exhausted = true; break; }
        ctl[3] = l + 1;
        if (!(ld < n_warm && !exhausted)) { kd = ld + 1; act = exhausted ? 2 : 1; }
      }
      ctl[1] = act;
      ctl[2] = kd;
    }
    __syncthreads();
    int act = ctl[1];
    const int k = ctl[2];
    if (act == 0) continue;

    // ---- coef = exp(scale * T_k) e_0 ------------------------------------------------------------
    bool vec_done = false;
    stamp(30);
    if (lanczos && k > 1) {  // tridiagonal: one wave, vector form (every wave runs it: the decision is uniform)
      zc cq;
      const int sq_ = wave_expm_tridiag(alpha, beta, scale, k, false, cq, tid < 64);
      if (sq_ <= SS_VEC_SMAX) {
        if (tid < k) coef[tid] = cq;
        __syncthreads();
        vec_done = true;
      }
    }
    if (!vec_done) {
      zc* Tm = Bs;
      zc* M2 = Tm + MAXK * MAXK;
      zc* M3 = M2 + MAXK * MAXK;
      zc* M4 = M3 + MAXK * MAXK;
      zc* Pm = M4 + MAXK * MAXK;
      zc* Qm = Pm + MAXK * MAXK;
      for (int t = tid; t < k * k; t += SS_THREADS) {
        const int i = t / k, j = t - i * k;
        zc z = make_double2(0.0, 0.0);
        if (lanczos) {
          if (i == j) z = alpha[i];
          else if (i == j + 1) z = make_double2(beta[j], 0.0);
          else if (j == i + 1) z = make_double2(beta[i], 0.0);
        } else {
          if (i <= j + 1) z = hess[i * MAXK + j];
        }
        Tm[t] = make_double2(scale.x * z.x - scale.y * z.y, scale.x * z.y + scale.y * z.x);
      }
      __syncthreads();
      stamp(17);
      ss_expm_col0(Tm, M2, M3, M4, Pm, Qm, k, coef, wsh);
      stamp(18);
    }
    stamp(31);
    if (act == 1) {
      if (have_prev) {  // || psi_k - psi_{k-1} ||  (:644-652)
        // (round 4, measured and dropped: issuing the k agent-scope loads of an element in batches of eight before their
        // first use, here, in the closing combination and in the own-range assembly -- C2 204 -> 201 sweeps/s on one box,
        // the wider live ranges spill in the per-iteration loops)
        double s = 0.0;
        own([&](long e) {
          double re = 0.0, im = 0.0;
          for (int j = 0; j < k; ++j) {
            zc d = coef[j];
            if (j < prev_len) { d.x -= cprev[j].x; d.y -= cprev[j].y; }
            const zc vj = basis(j, e);
            re += d.x * vj.x - d.y * vj.y;
            im += d.x * vj.y + d.y * vj.x;
          }
          s += re * re + im * im;
        });
        s = wave_sum64(s);
        __syncthreads();
        if ((tid & 63) == 0) wsh[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) pay[0] = wtree(wsh);
        __syncthreads();
        stamp(32);
        if (!ss_exchange(sy, pay, 1, red, wsh)) { fail(SS_ETIMEOUT); return; }
        stamp(33);
        if (sqrt(red[0]) < ex.thresh) act = 2;
      }
      if (act == 1) {
        __syncthreads();
        if (tid < k) cprev[tid] = coef[tid];
        __syncthreads();
        have_prev = true;
        prev_len = k;
        continue;
      }
    }
    // ---- act == 2: psi = sum_j c_j v_j, renormalised or rescaled (_rescale, :206-213) ------------
    // x is read by OTHER workgroups only before this iteration's exchanges; from here on every
    // workgroup touches its own range only.  With conserve_norm the new vector waits in the unused
    // slot 0 of the basis buffer until its norm is known.
    {
      const double cs_ = cn ? 1.0 : beta0;
      double s = 0.0;
      own([&](long e) {
        double re = 0.0, im = 0.0;
        for (int j = 0; j < k; ++j) {
          const zc d = make_double2(coef[j].x * cs_, coef[j].y * cs_);
          const zc vj = basis(j, e);
          re += d.x * vj.x - d.y * vj.y;
          im += d.x * vj.y + d.y * vj.x;
        }
        if (cn) g.U[e] = make_double2(re, im);
        else g.x[e] = make_double2(re, im);
        s += re * re + im * im;
      });
      if (cn) {
        s = wave_sum64(s);
        __syncthreads();
        if ((tid & 63) == 0) wsh[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) pay[0] = wtree(wsh);
        __syncthreads();
        if (!ss_exchange(sy, pay, 1, red, wsh)) { fail(SS_ETIMEOUT); return; }
        const double inv = 1.0 / sqrt(red[0]);
        own([&](long e) {  // same thread wrote g.U[e] above
          zc z = g.U[e];
          z.x *= inv; z.y *= inv;
          g.x[e] = z;
        });
      }
      if (blockIdx.x == 0 && tid == 0) {
        g.kprev[ex.site] = k;
        atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + ex.stat_slot), (unsigned long long)napply);
        atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + 2 + ex.stat_slot),
                  (unsigned long long)napply * (unsigned long long)ex.flops_per_apply);
      }
      return;
    }
  }
  fail(SS_ENOTCONV);  // "... is not converged in N basis. Try shorter time interval." (:430, :653)
  if (blockIdx.x == 0 && tid == 0) {
    g.kprev[ex.site] = ndim;
    atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + ex.stat_slot), (unsigned long long)napply);
    atomicAdd(reinterpret_cast<unsigned long long*>(g.stats + 2 + ex.stat_slot),
              (unsigned long long)napply * (unsigned long long)ex.flops_per_apply);
  }
}

}  // namespace

void kry_ritz(hipStream_t st, const KryRitzArgs& a) {
  hipLaunchKernelGGL(k_kry_ritz, dim3(1), dim3(SS_THREADS), 0, st, a);
  HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
size_t small_chain_lds(const SmallChain& c, bool exp_mode) { return ss_carve(c, exp_mode).total * sizeof(zc); }

bool small_chain_plan(SmallChain& c, bool exp_mode, int n_cu) {
  if (c.na < 1 || c.ns < 1 || c.nr < 1 || c.nb < 1) return false;
  const int gmax = std::min(n_cu, SS_MAXG);
  c.spw = 1;
  c.a_resident = 1;
  // chunks over s: enough workgroups that one holds about 0.4 Mflop of the chain (beyond that the exchanges
  // between workgroups, not the arithmetic, set the pace), at least 4 columns per chunk, LDS permitting
  const double flops = 8.0 * c.na * ((double)c.nc * c.nb * c.nj * c.ns + (c.W2 ? (double)c.ni * c.nt * c.nc * c.nj * c.ns : 0.0) +
                                     (double)c.ni * c.nt * c.ns * c.nr);
  int want = 1;
  while (want < 8 && flops / ((double)c.na * want) > 0.6e6) want *= 2;
  if (c.na <= gmax) {
    for (int nsc = want; nsc <= 8; nsc *= 2) {
      if (c.na * nsc > gmax) break;
      const int cs = (c.ns + nsc - 1) / nsc;
      if (nsc > 1 && cs < 4) break;
      if ((nsc - 1) * cs >= c.ns) continue;  // every chunk must be non-empty
      c.nsc = nsc;
      c.cs = cs;
      if (small_chain_lds(c, exp_mode) <= 155 * 1024) return true;
    }
    // fewer chunks than wanted: an engine confined to part of the chip (ensemble mode) has fewer compute units than the
    // chain would like workgroups
    for (int nsc = want / 2; nsc >= 1; nsc /= 2) {
      if (c.na * nsc > gmax) continue;
      const int cs = (c.ns + nsc - 1) / nsc;
      if ((nsc - 1) * cs >= c.ns) continue;
      c.nsc = nsc;
      c.cs = cs;
      if (small_chain_lds(c, exp_mode) <= 155 * 1024) return true;
    }
  }
  // Still not placed: one slab per workgroup wants more workgroups, or more LDS per workgroup (wide chunks), than the
  // slice offers.  Several slabs per workgroup (round 5): narrow chunks keep B / R / X / Y small, the grid is
  // ceil(na / spw) * nsc <= gmax.  Fewest slabs per workgroup first, then the most chunks that fit; A resident if it fits.
  static const bool multi_on = !(std::getenv("MITDVP_SS_MULTISLAB") && std::atoi(std::getenv("MITDVP_SS_MULTISLAB")) == 0);
  // (only for slices of at most 64 compute units -- four or more replicas: on larger grids a chain that does not fit with
  // one slab per workgroup is better served by the general multi-launch kernels, which is what it got before)
  if (!multi_on || n_cu > 64) return false;
  for (int spw = 2; spw <= 64 && spw <= 2 * c.na; spw *= 2) {
    const int ngrp = (c.na + spw - 1) / spw;
    if (ngrp > gmax) continue;
    for (int nsc = 8; nsc >= 1; nsc /= 2) {
      if (ngrp * nsc > gmax) continue;
      const int cs = (c.ns + nsc - 1) / nsc;
      if (nsc > 1 && cs < 4) continue;
      if ((nsc - 1) * cs >= c.ns) continue;
      c.nsc = nsc;
      c.cs = cs;
      c.spw = spw;
      for (int res = 1; res >= 0; --res) {
        c.a_resident = res;
        if (small_chain_lds(c, exp_mode) <= 155 * 1024) return true;
      }
    }
  }
  c.spw = 1;
  c.a_resident = 1;
  return false;
}

// ---- persistent launches of several engines on one GPU ---------------------------------------------------------
// k_small_site / k_qr_panel exchange data BETWEEN their workgroups inside one launch, so all workgroups of a launch
// must be resident together.  Two such grids from two engines (an ensemble of trajectories on one GPU, host threads)
// could each get part of the chip and wait for the rest of themselves until the 2 s timeout.  Per device, process
// wide, while more than one engine uses the family: every persistent launch is entered in a list with an event and
// its grid size (one workgroup per compute unit: each takes more than half a CU's LDS), and a new launch is issued only
// once "grids of other engines that may run beside it + its own grid <= compute units" holds (the launching host thread
// waits for the oldest foreign launch otherwise).  Launches that fit beside
// each other still overlap (two C2 local exponentials use 64-128 of the 256 compute units each); any set of launches
// that can be running at the same time fits the chip as a whole, so none can starve another.
namespace {
struct PEvent {  // one recorded launch; shared so that a waiter keeps it alive while it is being retired elsewhere
  hipEvent_t ev = nullptr;
  ~PEvent() { if (ev) (void)hipEventDestroy(ev); }
};
struct PersistentChain {
  std::mutex mu;
  struct InFlight { std::shared_ptr<PEvent> e; int grid; hipStream_t st; };
  std::vector<InFlight> fl;
  int users = 0;
  int n_cu = 0;
};
PersistentChain g_pchain[64];
inline int current_device_slot() {
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  return dev >= 0 && dev < 64 ? dev : 63;
}
}  // namespace

void persistent_register(int delta) {
  PersistentChain& c = g_pchain[current_device_slot()];
  std::lock_guard<std::mutex> lk(c.mu);
  c.users += delta;
  if (delta > 0 && c.users == 2) {
    // the engine that was alone so far recorded nothing: let its launches drain once, from now on the list orders them
    HIP_CHECK(hipDeviceSynchronize());
  }
}

// Admission on the HOST: the launching thread waits (hipEventSynchronize, lock released) for the oldest foreign launch
// still in flight until its own grid fits beside the rest.  (Making the STREAM wait instead -- hipStreamWaitEvent on the
// other engine's event -- was measured first: 8 engines fell from ~370 to 43 sweeps/s in aggregate, a cross-stream
// dependency costs far more than the 20-140 us kernels it orders.)
PersistentLaunch::PersistentLaunch(hipStream_t st, int grid, bool partitioned)
    : st_(st), slot_(partitioned ? -1 : current_device_slot()), grid_(grid) {
  if (slot_ < 0) return;  // a stream confined to its own compute units: nobody to wait for
  PersistentChain& c = g_pchain[slot_];
  c.mu.lock();
  chained_ = c.users > 1;
  if (!chained_) return;
  try {
    if (!c.n_cu) {
      int dev = 0;
      HIP_CHECK(hipGetDevice(&dev));
      HIP_CHECK(hipDeviceGetAttribute(&c.n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    for (;;) {
      size_t keep = 0;  // forget what has completed
      for (size_t k = 0; k < c.fl.size(); ++k)
        if (hipEventQuery(c.fl[k].e->ev) != hipSuccess) c.fl[keep++] = c.fl[k];
      c.fl.resize(keep);
      (void)hipGetLastError();  // hipEventQuery reports "not ready" through the error state
      int beside = 0;
      std::shared_ptr<PEvent> oldest;
      for (const auto& f : c.fl)
        if (f.st != st) {  // stream order already puts this stream's own launches before the new one
          beside += f.grid;
          if (!oldest) oldest = f.e;
        }
      if (beside + grid <= c.n_cu || !oldest) break;
      c.mu.unlock();
      const hipError_t rc = hipEventSynchronize(oldest->ev);
      c.mu.lock();
      if (rc != hipSuccess) throw HipError("persistent launch admission: hipEventSynchronize failed");
    }
  } catch (...) {
    c.mu.unlock();
    throw;
  }
}

PersistentLaunch::~PersistentLaunch() {
  if (slot_ < 0) return;
  PersistentChain& c = g_pchain[slot_];
  if (chained_) {
    auto e = std::make_shared<PEvent>();
    if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(e->ev, st_) == hipSuccess)
      c.fl.push_back({e, grid_, st_});
  }
  c.mu.unlock();
}

void small_sync_alloc(SmallSync& s, int nsite, hipStream_t st) {
  if (s.words) return;
  persistent_register(+1);
  HIP_CHECK(hipMalloc(&s.words, 16 * sizeof(unsigned)));
  HIP_CHECK(hipMalloc(&s.slots, (size_t)2 * SS_MAXG * SS_NGR * sizeof(unsigned long long)));
  HIP_CHECK(hipMalloc(&s.stats, 4 * sizeof(long long)));
  HIP_CHECK(hipMalloc(&s.kprev, (size_t)std::max(nsite, 1) * sizeof(int)));
  HIP_CHECK(hipMemsetAsync(s.words, 0, 16 * sizeof(unsigned), st));
  HIP_CHECK(hipMemsetAsync(s.slots, 0, (size_t)2 * SS_MAXG * SS_NGR * sizeof(unsigned long long), st));
  HIP_CHECK(hipMemsetAsync(s.stats, 0, 4 * sizeof(long long), st));
  HIP_CHECK(hipMemsetAsync(s.kprev, 0, (size_t)std::max(nsite, 1) * sizeof(int), st));
  HIP_CHECK(hipStreamSynchronize(st));
  s.launches = 0;
}

void small_sync_free(SmallSync& s) {
  if (s.words) {
    try { persistent_register(-1); } catch (...) {}
  }
  if (s.words) (void)hipFree(s.words);
  if (s.slots) (void)hipFree(s.slots);
  if (s.stats) (void)hipFree(s.stats);
  if (s.kprev) (void)hipFree(s.kprev);
  s = SmallSync{};
}

static void ss_launch(hipStream_t st, SmallSync& sy, SsArgs& g, bool exp_mode) {
  const SmallChain& c = g.c;
  // validate what the kernel's indexing assumes before anything is launched
  if (c.nsc < 1 || c.cs < 1 || (c.nsc - 1) * c.cs >= c.ns || c.nsc * c.cs < c.ns) throw ArgError("small_site: bad chunking");
  const int spw_ = c.spw > 1 ? c.spw : 1;
  const int grid_ = ((c.na + spw_ - 1) / spw_) * c.nsc;
  if (grid_ > SS_MAXG || (sy.max_grid > 0 && grid_ > sy.max_grid)) throw ArgError("small_site: grid exceeds the resident-workgroup bound");
  if (!c.W2 && (c.ni != 1 || c.nj != 1 || c.nt != c.nc)) throw ArgError("small_site: chain without W stage needs ni = nj = 1, nt = nc");
  size_t lds = small_chain_lds(c, exp_mode);
  if (lds > 156 * 1024) throw ArgError("small_site: chain does not fit LDS");
  lds = std::max<size_t>(lds, 84 * 1024);  // > half a CU's LDS: at most one workgroup per CU
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  static bool attr_set[64] = {};  // the attribute is per device
  static std::mutex attr_mu;
  std::lock_guard<std::mutex> lk(attr_mu);
  if (dev >= 64 || !attr_set[dev]) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_site), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  156 * 1024));  // the kernel's few static LDS words come out of the same 160 KiB
    if (dev < 64) attr_set[dev] = true;
  }
  // exchange tags are unique per launch: 4096 epochs each (a launch needs at most 4 * MAXK + 2)
  sy.launches += 1;
  if ((sy.launches & 0xFFFFFu) == 0u) {  // tag space wraps: forget the old granules
    HIP_CHECK(hipMemsetAsync(sy.slots, 0, (size_t)2 * SS_MAXG * SS_NGR * sizeof(unsigned long long), st));
    sy.launches += 1;
  }
  g.epoch0 = (sy.launches & 0xFFFFFu) << 12;
  g.abort_w = sy.words + 2;
  g.err_w = sy.words + 3;
  g.gran = reinterpret_cast<unsigned long long*>(sy.slots);
  g.stats = sy.stats;
  g.kprev = sy.kprev;
  static const bool tracing = std::getenv("MITDVP_SS_TRACE") != nullptr;
  static long long* trace_buf = nullptr;
  if (tracing && !trace_buf) {
    HIP_CHECK(hipMalloc(&trace_buf, 1024 * sizeof(long long)));
  }
  g.trace = tracing ? trace_buf : nullptr;
  if (tracing) HIP_CHECK(hipMemsetAsync(trace_buf, 0, 1024 * sizeof(long long), st));
  {
    PersistentLaunch chain(st, grid_, sy.partitioned);  // admitted only when it fits beside the persistent launches in flight
    hipLaunchKernelGGL(k_small_site, dim3(grid_), dim3(SS_THREADS), lds, st, g);
  }
  HIP_CHECK(hipGetLastError());
  if (tracing) {  // debugging aid: phase timeline of workgroup 0, microseconds since its first stamp
    long long h[1024];
    HIP_CHECK(hipMemcpyAsync(h, trace_buf, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    fprintf(stderr, "[ss_trace] mode=%d G=%d (na=%d nsc=%d cs=%d spw=%d res=%d) lds=%zu:", g.mode, grid_, c.na, c.nsc, c.cs, spw_, c.a_resident, lds);
    for (int i = 1; i < (int)h[0] && i < 250; ++i) fprintf(stderr, " %lld:%.2f", h[2 * i + 1], (double)(h[2 * i] - h[2]) * 0.01);
    const int nn = (int)h[0];
    if (nn > 2)
      fprintf(stderr, " | shader clock %.0f MHz", (double)(h[512 + nn - 1] - h[512 + 1]) / ((double)(h[2 * (nn - 1)] - h[2]) * 0.01));
    fprintf(stderr, "\n");
  }
}

void small_apply(hipStream_t st, SmallSync& sy, const SmallChain& c, const zc* v, zc* out, zc* partials, zc shift,
                 bool add_shift) {
  SsArgs g{};
  g.c = c;
  g.mode = SS_MODE_APPLY;
  g.v = v;
  g.out = out;
  g.P = partials;
  g.shift = shift;
  g.add_shift = add_shift ? 1 : 0;
  ss_launch(st, sy, g, false);
}

void small_exp(hipStream_t st, SmallSync& sy, const SmallChain& c, const SmallExp& e, zc* x, zc* basis, zc* partials,
               zc shift) {
  if ((long)c.na * c.ni * c.nr != (long)c.nb * c.nj * c.ns) throw ArgError("small_exp: the operator must be square");
  SsArgs g{};
  g.c = c;
  g.mode = SS_MODE_EXP;
  g.x = x;
  g.U = basis;
  g.P = partials;
  g.shift = shift;
  g.add_shift = (shift.x != 0.0 || shift.y != 0.0) ? 1 : 0;
  g.e = e;
  ss_launch(st, sy, g, true);
}

}  // namespace mitdvp
