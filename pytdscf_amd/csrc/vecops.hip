// vecops.hip -- HBM-bound helper kernels of the TDVP engine: Krylov vector
// algebra with device-resident scalars, tensor-leg transposes, RNG.
//
// Reductions are two-stage and deterministic: a producer kernel writes one
// partial per workgroup, the CONSUMER kernel (or the host, after an async copy)
// sums the partials in a fixed order.  No atomics, no host round trip between
// the kernels of one Krylov iteration (the reference syncs twice per iteration,
// _integrator.py:553-554).
#include "vecops.h"
#include "krylov_dev.h"

#include <algorithm>

namespace mitdvp {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sum over the 256-thread block; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double* sh /*[5]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return sh[4];
}

__device__ __forceinline__ zc sum_partials_z(const zc* p, int np, double* sh) {
  double re = 0, im = 0;
  for (int i = threadIdx.x; i < np; i += 256) { re += p[i].x; im += p[i].y; }
  re = block_sum(re, sh);
  im = block_sum(im, sh);
  return make_double2(re, im);
}
__device__ __forceinline__ double sum_partials_d(const double* p, int np, double* sh) {
  double re = 0;
  for (int i = threadIdx.x; i < np; i += 256) re += p[i];
  return block_sum(re, sh);
}

// out[b] = sum_i conj(x_i) * y_i   (conj_x = 0: no conjugation)
__global__ __launch_bounds__(256) void k_dot(const zc* __restrict__ x, const zc* __restrict__ y, long n,
                                             int conj_x, zc* __restrict__ out) {
  __shared__ double sh[5];
  double re = 0, im = 0;
  const double s = conj_x ? 1.0 : -1.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const zc a = x[i], b = y[i];
    re += a.x * b.x + s * a.y * b.y;
    im += a.x * b.y - s * a.y * b.x;
  }
  re = block_sum(re, sh);
  im = block_sum(im, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = make_double2(re, im);
}

// out[b] = sum |x_i|^2
__global__ __launch_bounds__(256) void k_sumsq(const zc* __restrict__ x, long n, double* __restrict__ out) {
  __shared__ double sh[5];
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const zc a = x[i];
    s += a.x * a.x + a.y * a.y;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// Lanczos three-term update (reference form, _integrator.py:556-562):
//   alpha = sum(alpha_p);  beta_prev = sqrt(sum(betaprev_p))
//   v -= alpha * vm1 + beta_prev * vm2 ;  out[b] = sum |v|^2
__global__ __launch_bounds__(256) void k_lanczos_update(zc* __restrict__ v, const zc* __restrict__ vm1,
                                                        const zc* __restrict__ vm2, long n,
                                                        const zc* __restrict__ alpha_p,
                                                        const double* __restrict__ betaprev_p, int np,
                                                        double* __restrict__ out) {
  __shared__ double sh[5];
  const zc alpha = sum_partials_z(alpha_p, np, sh);
  double bp = 0.0;
  if (vm2) bp = sqrt(sum_partials_d(betaprev_p, np, sh));
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    zc x = v[i];
    const zc a = vm1[i];
    x.x -= alpha.x * a.x - alpha.y * a.y;
    x.y -= alpha.x * a.y + alpha.y * a.x;
    if (vm2) {
      const zc c = vm2[i];
      x.x -= bp * c.x;
      x.y -= bp * c.y;
    }
    v[i] = x;
    s += x.x * x.x + x.y * x.y;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// The same update with DEFERRED normalisation: the basis is kept unnormalised, u_j, and v_j = u_j * invb_j is formed
// where it is used (invb_0 = 1, invb_j = 1 / beta_{j-1}, or 1 when beta_{j-1} < eps: exactly the factor the separate
// normalisation kernel would have applied, so v_j has the same bits wherever it is formed).  v holds H u_l on entry
// (H is linear: H v_l = invb_l H u_l) and u_{l+1} on exit; one launch and one pass over the vector less per iteration.
//   alpha_l = invb_l <v_0 | H u_l>  (reference form; orthodox: invb_l^2 <u_l | H u_l>),  nrm_all[j][.] = partials of |u_{j+1}|^2
__global__ __launch_bounds__(256) void k_lanczos_update_def(zc* __restrict__ v, const zc* __restrict__ ul,
                                                            const zc* __restrict__ ulm1, long n,
                                                            const zc* __restrict__ alpha_raw_p,
                                                            const double* __restrict__ nrm_all, int l, int orthodox,
                                                            double eps, int np, double* __restrict__ out) {
  __shared__ double sh[5];
  double invb_l = 1.0, invb_lm1 = 1.0, beta_prev = 0.0;
  if (l > 0) {
    beta_prev = sqrt(sum_partials_d(nrm_all + (long)(l - 1) * np, np, sh));
    if (beta_prev >= eps) invb_l = 1.0 / beta_prev;
  }
  if (l > 1) {
    const double b2 = sqrt(sum_partials_d(nrm_all + (long)(l - 2) * np, np, sh));
    if (b2 >= eps) invb_lm1 = 1.0 / b2;
  }
  zc alpha = sum_partials_z(alpha_raw_p, np, sh);
  const double fa = orthodox ? invb_l * invb_l : invb_l;
  alpha.x *= fa;
  alpha.y *= fa;
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    zc x = v[i];
    x.x *= invb_l;
    x.y *= invb_l;
    zc a = ul[i];
    a.x *= invb_l;
    a.y *= invb_l;
    x.x -= alpha.x * a.x - alpha.y * a.y;
    x.y -= alpha.x * a.y + alpha.y * a.x;
    if (ulm1) {
      zc c = ulm1[i];
      c.x *= invb_lm1;
      c.y *= invb_lm1;
      x.x -= beta_prev * c.x;
      x.y -= beta_prev * c.y;
    }
    v[i] = x;
    s += x.x * x.x + x.y * x.y;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// v /= sqrt(sum(nrm_p)) unless that norm is below eps (Krylov space exhausted,
// _integrator.py:563-567, :254-256)
__global__ __launch_bounds__(256) void k_scale_inv_norm(zc* __restrict__ v, long n,
                                                        const double* __restrict__ nrm_p, int np, double eps) {
  __shared__ double sh[5];
  const double beta = sqrt(sum_partials_d(nrm_p, np, sh));
  if (!(beta >= eps)) return;
  const double inv = 1.0 / beta;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    zc x = v[i];
    x.x *= inv;
    x.y *= inv;
    v[i] = x;
  }
}

// Arnoldi: h_j = <V_j | v>, j < k   -> out[j*np + b]   (_orth_step_np, _integrator.py:250)
__global__ __launch_bounds__(256) void k_multi_dot(const zc* __restrict__ V, long ldv, int k,
                                                   const zc* __restrict__ v, long n, zc* __restrict__ out) {
  __shared__ double sh[5];
  double re[MAXK], im[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; ++j) { re[j] = 0; im[j] = 0; }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const zc b = v[i];
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
      if (j < k) {
        const zc a = V[(long)j * ldv + i];
        re[j] += a.x * b.x + a.y * b.y;
        im[j] += a.x * b.y - a.y * b.x;
      }
  }
#pragma unroll
  for (int j = 0; j < MAXK; ++j)
    if (j < k) {
      const double r = block_sum(re[j], sh);
      const double m = block_sum(im[j], sh);
      if (threadIdx.x == 0) out[(long)j * gridDim.x + blockIdx.x] = make_double2(r, m);
    }
}

// Arnoldi: v -= sum_j h_j V_j ; out[b] = sum |v|^2   (_integrator.py:251-252)
__global__ __launch_bounds__(256) void k_arnoldi_update(zc* __restrict__ v, const zc* __restrict__ V, long ldv,
                                                        int k, long n, const zc* __restrict__ h_p, int np,
                                                        double* __restrict__ out) {
  __shared__ double sh[5];
  __shared__ zc hs[MAXK];
  for (int j = 0; j < k; ++j) {
    const zc h = sum_partials_z(h_p + (long)j * np, np, sh);
    if (threadIdx.x == 0) hs[j] = h;
  }
  __syncthreads();
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    zc x = v[i];
    for (int j = 0; j < k; ++j) {
      const zc a = V[(long)j * ldv + i];
      const zc h = hs[j];
      x.x -= h.x * a.x - h.y * a.y;
      x.y -= h.x * a.y + h.y * a.x;
    }
    v[i] = x;
    s += x.x * x.x + x.y * x.y;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// out = sum_j c_j V_j (out may be null: norm only); nrm[b] = sum |.|^2 (nrm may be null)
__global__ __launch_bounds__(256) void k_lincomb(zc* __restrict__ out, const zc* __restrict__ V, long ldv, int k,
                                                 Coefs c, long n, double* __restrict__ nrm) {
  __shared__ double sh[5];
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double re = 0, im = 0;
    for (int j = 0; j < k; ++j) {
      const zc a = V[(long)j * ldv + i];
      re += c.c[j].x * a.x - c.c[j].y * a.y;
      im += c.c[j].x * a.y + c.c[j].y * a.x;
    }
    if (out) out[i] = make_double2(re, im);
    s += re * re + im * im;
  }
  if (nrm) {
    s = block_sum(s, sh);
    if (threadIdx.x == 0) nrm[blockIdx.x] = s;
  }
}

// krylov_dev.h: || sum_j dcoef_j V_j || against the threshold (_integrator.py:638-651), decided on the device.
// Every workgroup leaves its partial with an agent-scope store and takes an arrival ticket; the one whose ticket is the
// last sums the partials in a fixed order (thread i <- partial i, then the block tree: the same bits whichever
// workgroup it is), closes the exponential or rolls coef_prev, and publishes the record the host spins on.
// (MI355X_MICROARCH.md, inter-workgroup visibility, valid forms: sc1 stores drained before ONE agent-scope add per
// workgroup to one counter; the last arriver, told by the value its add returned, reads with sc1 loads.)
__device__ __forceinline__ void kry_publish(const KryDev* st, KryPub* pub, unsigned tag) {
  const int k = st->k;
  for (int j = 0; j < k; ++j) {  // what the host combines the STORED vectors with (deferred normalisation: u_j = v_j / invb_j)
    const zc c = st->coef[j];
    const double f = st->deferred ? st->invb[j] : 1.0;
    pub->coef[j] = make_double2(c.x * f, c.y * f);
  }
  pub->beta0 = st->beta0;
  pub->err = st->err;
  pub->state = st->state;
  pub->k = k;
  __threadfence_system();
  *reinterpret_cast<volatile unsigned*>(&pub->seq) = tag;
}
__global__ __launch_bounds__(256) void k_kry_diff(KryDev* st, const zc* __restrict__ V, long ldv, long n, double thresh,
                                                  KryPub* pub, unsigned tag) {
  __shared__ double sh[5];
  __shared__ zc dc[MAXK];
  __shared__ unsigned last_s;
  if (!st->need_diff) {  // first Ritz vector of this exponential, or closed already: nothing to compare
    if (blockIdx.x == 0 && threadIdx.x == 0) kry_publish(st, pub, tag);
    return;
  }
  const int k = st->k;
  if (threadIdx.x < k) dc[threadIdx.x] = st->dcoef[threadIdx.x];
  __syncthreads();
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    double re = 0, im = 0;
    for (int j = 0; j < k; ++j) {
      const zc a = V[(long)j * ldv + i];
      const zc c = dc[j];
      re += c.x * a.x - c.y * a.y;
      im += c.x * a.y + c.y * a.x;
    }
    s += re * re + im * im;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&st->part[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_s = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!last_s) return;
  double v = (int)threadIdx.x < (int)gridDim.x
                 ? __hip_atomic_load(&st->part[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                 : 0.0;
  v = block_sum(v, sh);
  if (threadIdx.x == 0) {
    const double err = sqrt(v);
    st->err = err;
    if (err < thresh) {
      st->state = KRY_CONVERGED;
    } else {
      for (int j = 0; j < k; ++j) st->cprev[j] = st->coef[j];
      st->prev_len = k;
    }
    st->need_diff = 0;
    __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    kry_publish(st, pub, tag);
  }
}

// y = a*x + b*y
__global__ __launch_bounds__(256) void k_axpby(zc* __restrict__ y, const zc* __restrict__ x, long n, zc a, zc b) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const zc xv = x[i], yv = y[i];
    y[i] = make_double2(a.x * xv.x - a.y * xv.y + b.x * yv.x - b.y * yv.y,
                        a.x * xv.y + a.y * xv.x + b.x * yv.y + b.y * yv.x);
  }
}

__global__ __launch_bounds__(256) void k_scale(zc* __restrict__ y, long n, zc a) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const zc yv = y[i];
    y[i] = make_double2(a.x * yv.x - a.y * yv.y, a.x * yv.y + a.y * yv.x);
  }
}

// out[c][r] = in[r][c], batched: in_b = in + b*in_bs (row stride ldi), out_b = out + b*out_bs (ldo)
__global__ __launch_bounds__(256) void k_transpose(const zc* __restrict__ in, zc* __restrict__ out, int rows,
                                                   int cols, long ldi, long ldo, long in_bs, long out_bs) {
  __shared__ zc tile[32][33];
  const zc* ib = in + (long)blockIdx.z * in_bs;
  zc* ob = out + (long)blockIdx.z * out_bs;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + ty + 8 * q, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * q][tx] = ib[(long)r * ldi + c];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + ty + 8 * q, r = r0 + tx;
    if (r < rows && c < cols) ob[(long)c * ldo + r] = tile[tx][ty + 8 * q];
  }
}

// Liouville space (site dimension n*n, physical index = row*n + col):
//   diag: out[b][c][e] = C[b][c*n+c][e]     trace: out[b][e] = sum_c C[b][c*n+c][e]
__global__ __launch_bounds__(256) void k_phys_diag(const zc* __restrict__ C, zc* __restrict__ out, int dl, int n, int dr,
                                                   int trace) {
  const long tot = trace ? (long)dl * dr : (long)dl * n * dr;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    if (trace) {
      const int s = e % dr;
      const long b = e / dr;
      double re = 0, im = 0;
      for (int c = 0; c < n; ++c) {
        const zc v = C[(b * n * n + (long)c * n + c) * dr + s];
        re += v.x;
        im += v.y;
      }
      out[e] = make_double2(re, im);
    } else {
      const int s = e % dr;
      const int c = (e / dr) % n;
      const long b = e / ((long)dr * n);
      out[e] = C[(b * n * n + (long)c * n + c) * dr + s];
    }
  }
}

// out[i0][i2][i1][i3] = in[i0][i1][i2][i3]
__global__ __launch_bounds__(256) void k_permute_0213(const zc* __restrict__ in, zc* __restrict__ out, long n0, int n1,
                                                      int n2, int n3) {
  const long tot = n0 * n1 * n2 * n3;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const int i3 = e % n3;
    long r = e / n3;
    const int i1 = r % n1;  // output order: i0, i2, i1, i3
    r /= n1;
    const int i2 = r % n2;
    const long i0 = r / n2;
    out[e] = in[((i0 * n1 + i1) * n2 + i2) * n3 + i3];
  }
}

// out (C order over dims d0..d4) [i0..i4] = in[sum_j i_j * stride_j]; an optional index map
// replaces i2 (row gather by sorted index)
struct Perm5 {
  int d[5];
  long s[5];
};
__global__ __launch_bounds__(256) void k_permute5(const zc* __restrict__ in, zc* __restrict__ out, Perm5 p,
                                                  const int* __restrict__ map2) {
  const long tot = (long)p.d[0] * p.d[1] * p.d[2] * p.d[3] * p.d[4];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    long r = e;
    const int i4 = r % p.d[4]; r /= p.d[4];
    const int i3 = r % p.d[3]; r /= p.d[3];
    int i2 = r % p.d[2]; r /= p.d[2];
    const int i1 = r % p.d[1];
    const long i0 = r / p.d[1];
    if (map2) i2 = map2[i2];
    out[e] = in[i0 * p.s[0] + i1 * p.s[1] + i2 * p.s[2] + i3 * p.s[3] + i4 * p.s[4]];
  }
}

// x[r][c] *= s[c]  (real scale per column)
__global__ __launch_bounds__(256) void k_scale_cols(zc* __restrict__ x, long rows, int cols, long ld,
                                                    const double* __restrict__ sc) {
  const long tot = rows * cols;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const long r = e / cols;
    const int c = (int)(e % cols);
    zc v = x[r * ld + c];
    v.x *= sc[c]; v.y *= sc[c];
    x[r * ld + c] = v;
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// complex standard normal (counter based, reproducible for a given seed)
__global__ __launch_bounds__(256) void k_randn(zc* __restrict__ out, long n, uint64_t seed) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const uint64_t a = splitmix64(seed ^ (uint64_t)(2 * i));
    const uint64_t b = splitmix64(seed ^ (uint64_t)(2 * i + 1) ^ 0xD1B54A32D192ED03ull);
    const double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0);
    const double u2 = (b >> 11) * (1.0 / 9007199254740992.0);
    const double r = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925 * u2, &sn, &cs);
    out[i] = make_double2(r * cs, r * sn);
  }
}

__global__ __launch_bounds__(256) void k_identity(zc* __restrict__ out, int rows, int cols, long ld) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int r = i / cols, c = i % cols;
    out[(long)r * ld + c] = make_double2(r == c ? 1.0 : 0.0, 0.0);
  }
}

// ---------------------------------------------------------------------------
// Small-bond regime: the whole Lanczos vector step of one Krylov iteration in ONE
// single-workgroup launch (dot, three-term update, norm, normalisation) -- the
// vector (<= 16384 elements) lives in registers between the passes.  Replaces three
// dependent launches of a few microseconds each; the partial arrays are written in
// the same layout ([0] = value, rest 0) so that every consumer stays unchanged.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_1024(double v, double* sh /*[17]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sh[i];
    sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

__global__ __launch_bounds__(1024) void k_lanczos_step_small(zc* __restrict__ w, const zc* __restrict__ x,
                                                             const zc* __restrict__ vl, const zc* __restrict__ vm2,
                                                             int n, zc* __restrict__ alpha_p,
                                                             const double* __restrict__ betaprev_p,
                                                             double* __restrict__ nrm_p, double eps) {
  __shared__ double sh[17];
  zc wv[SMALL_VEC_EPT];
  double re = 0, im = 0;
#pragma unroll
  for (int e = 0; e < SMALL_VEC_EPT; ++e) {
    const int i = threadIdx.x + e * 1024;
    if (i < n) {
      const zc b = w[i], a = x[i];
      wv[e] = b;
      re += a.x * b.x + a.y * b.y;  // conj(x) * w
      im += a.x * b.y - a.y * b.x;
    }
  }
  re = block_sum_1024(re, sh);
  im = block_sum_1024(im, sh);
  double bp = 0.0;
  if (vm2) {
    double t = threadIdx.x < NPART ? betaprev_p[threadIdx.x] : 0.0;
    bp = sqrt(block_sum_1024(t, sh));
  }
  double s = 0;
#pragma unroll
  for (int e = 0; e < SMALL_VEC_EPT; ++e) {
    const int i = threadIdx.x + e * 1024;
    if (i < n) {
      zc v = wv[e];
      const zc a = vl[i];
      v.x -= re * a.x - im * a.y;
      v.y -= re * a.y + im * a.x;
      if (vm2) {
        const zc c = vm2[i];
        v.x -= bp * c.x;
        v.y -= bp * c.y;
      }
      wv[e] = v;
      s += v.x * v.x + v.y * v.y;
    }
  }
  s = block_sum_1024(s, sh);
  const double beta = sqrt(s);
  const double inv = beta >= eps ? 1.0 / beta : 1.0;  // exhausted Krylov space: left as is
#pragma unroll
  for (int e = 0; e < SMALL_VEC_EPT; ++e) {
    const int i = threadIdx.x + e * 1024;
    if (i < n) w[i] = make_double2(wv[e].x * inv, wv[e].y * inv);
  }
  if (threadIdx.x < NPART) {
    alpha_p[threadIdx.x] = threadIdx.x == 0 ? make_double2(re, im) : make_double2(0.0, 0.0);
    nrm_p[threadIdx.x] = threadIdx.x == 0 ? s : 0.0;
  }
}

// ---------------------------------------------------------------------------
// adaptive bond dimension: strided block copies and the norm profiles of the
// rank-selection functional f(D) (_mps_cls.py:2083-2105)
// ---------------------------------------------------------------------------
// dst[r][c] = a * src[r][c] (+ dst[r][c] if acc); dst columns >= cols are zero-filled up to zcols
__global__ __launch_bounds__(256) void k_copy2d(zc* __restrict__ dst, long ldd, const zc* __restrict__ src, long lds,
                                                long rows, int cols, int zcols, zc a, int acc) {
  const int w = zcols > cols ? zcols : cols;
  const long n = rows * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / w;
    const int c = (int)(i % w);
    if (c < cols) {
      const zc v = src[r * lds + c];
      zc o = make_double2(a.x * v.x - a.y * v.y, a.x * v.y + a.y * v.x);
      if (acc) { const zc d = dst[r * ldd + c]; o.x += d.x; o.y += d.y; }
      dst[r * ldd + c] = o;
    } else {
      dst[r * ldd + c] = make_double2(0.0, 0.0);
    }
  }
}

// out[c] = sum_r |x[r][c]|^2 : one workgroup per column (fixed summation order)
__global__ __launch_bounds__(256) void k_col_sumsq(const zc* __restrict__ x, long rows, int cols,
                                                   double* __restrict__ out) {
  __shared__ double sh[5];
  const int c = blockIdx.x;
  double a = 0;
  for (long r = threadIdx.x; r < rows; r += 256) {
    const zc v = x[r * cols + c];
    a += v.x * v.x + v.y * v.y;
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) out[c] = a;
}

// out[r] = sum_c |x[r][c]|^2
__global__ __launch_bounds__(256) void k_row_sumsq(const zc* __restrict__ x, int rows, long cols,
                                                   double* __restrict__ out) {
  __shared__ double sh[5];
  const zc* xr = x + (long)blockIdx.x * cols;
  double a = 0;
  for (long c = threadIdx.x; c < cols; c += 256) {
    const zc v = xr[c];
    a += v.x * v.x + v.y * v.y;
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}

// out[t] = sum over the "shell" max(i, j) == t of |x[i][j]|^2  (square n x n):
// |x[:D, :D]|^2 = sum_{t < D} out[t]
__global__ __launch_bounds__(256) void k_shell_sumsq(const zc* __restrict__ x, int n, double* __restrict__ out) {
  __shared__ double sh[5];
  const int t = blockIdx.x;
  double a = 0;
  for (int j = threadIdx.x; j <= t; j += 256) {  // row t, columns 0..t
    const zc v = x[(long)t * n + j];
    a += v.x * v.x + v.y * v.y;
  }
  for (int i = threadIdx.x; i < t; i += 256) {  // column t, rows 0..t-1
    const zc v = x[(long)i * n + t];
    a += v.x * v.x + v.y * v.y;
  }
  a = block_sum(a, sh);
  if (threadIdx.x == 0) out[t] = a;
}

// ---------------------------------------------------------------------------
// host wrappers
// ---------------------------------------------------------------------------
int vec_blocks(long n) {
  long nb = (n + 1023) / 1024;
  if (nb < 1) nb = 1;
  if (nb > NPART) nb = NPART;
  return (int)nb;
}

#define LAUNCH(k, nb, st, ...)                               \
  do {                                                       \
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, st, __VA_ARGS__); \
    HIP_CHECK(hipGetLastError());                            \
  } while (0)

void vec_dot(hipStream_t st, const zc* x, const zc* y, long n, bool conj_x, zc* out_p) {
  LAUNCH(k_dot, NPART, st, x, y, n, conj_x ? 1 : 0, out_p);
}
void vec_sumsq(hipStream_t st, const zc* x, long n, double* out_p) { LAUNCH(k_sumsq, NPART, st, x, n, out_p); }
void vec_lanczos_update(hipStream_t st, zc* v, const zc* vm1, const zc* vm2, long n, const zc* alpha_p,
                        const double* betaprev_p, double* out_p) {
  LAUNCH(k_lanczos_update, NPART, st, v, vm1, vm2, n, alpha_p, betaprev_p, NPART, out_p);
}
void vec_lanczos_update_deferred(hipStream_t st, zc* v, const zc* ul, const zc* ulm1, long n, const zc* alpha_raw_p,
                                 const double* nrm_all, int l, bool orthodox, double eps, double* out_p) {
  LAUNCH(k_lanczos_update_def, NPART, st, v, ul, ulm1, n, alpha_raw_p, nrm_all, l, orthodox ? 1 : 0, eps, NPART, out_p);
}
void vec_lanczos_step_small(hipStream_t st, zc* w, const zc* x, const zc* vl, const zc* vm2, long n, zc* alpha_p,
                            const double* betaprev_p, double* nrm_p, double eps) {
  if (n > SMALL_VEC_N) throw ArgError("vec_lanczos_step_small: vector too long");
  hipLaunchKernelGGL(k_lanczos_step_small, dim3(1), dim3(1024), 0, st, w, x, vl, vm2, (int)n, alpha_p, betaprev_p, nrm_p,
                     eps);
  HIP_CHECK(hipGetLastError());
}
void vec_scale_inv_norm(hipStream_t st, zc* v, long n, const double* nrm_p, double eps) {
  LAUNCH(k_scale_inv_norm, vec_blocks(n), st, v, n, nrm_p, NPART, eps);
}
void vec_multi_dot(hipStream_t st, const zc* V, long ldv, int k, const zc* v, long n, zc* out_p) {
  if (k > MAXK) throw ArgError("vec_multi_dot: k > MAXK");
  LAUNCH(k_multi_dot, NPART, st, V, ldv, k, v, n, out_p);
}
void vec_arnoldi_update(hipStream_t st, zc* v, const zc* V, long ldv, int k, long n, const zc* h_p, double* out_p) {
  LAUNCH(k_arnoldi_update, NPART, st, v, V, ldv, k, n, h_p, NPART, out_p);
}
void vec_lincomb(hipStream_t st, zc* out, const zc* V, long ldv, int k, const Coefs& c, long n, double* nrm_p) {
  if (k > MAXK) throw ArgError("vec_lincomb: k > MAXK");
  LAUNCH(k_lincomb, NPART, st, out, V, ldv, k, c, n, nrm_p);
}
void kry_diff(hipStream_t st, KryDev* kst, const zc* V, long ldv, long n, double thresh, KryPub* pub_dev, unsigned tag) {
  LAUNCH(k_kry_diff, NPART, st, kst, V, ldv, n, thresh, pub_dev, tag);
}
void vec_axpby(hipStream_t st, zc* y, const zc* x, long n, zc a, zc b) { LAUNCH(k_axpby, vec_blocks(n), st, y, x, n, a, b); }
void vec_scale(hipStream_t st, zc* y, long n, zc a) { LAUNCH(k_scale, vec_blocks(n), st, y, n, a); }
void vec_randn(hipStream_t st, zc* out, long n, uint64_t seed) { LAUNCH(k_randn, vec_blocks(n), st, out, n, seed); }
void set_identity(hipStream_t st, zc* out, int rows, int cols, long ld) {
  LAUNCH(k_identity, vec_blocks((long)rows * cols), st, out, rows, cols, ld);
}

// shader clock seen by a short single-workgroup kernel: cycles of s_memtime per 100 MHz
// tick of s_memrealtime over a dependent FMA chain (what the small-bond regime runs at)
__global__ void k_clock_probe(long iters, double* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  double x = 1.0 + threadIdx.x * 1e-9;
  for (long i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-12;
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = (double)(t1 - t0);
    out[1] = (double)(r1 - r0);
    out[2] = x;
  }
}
void clock_probe(hipStream_t st, long iters, double* host_out3) {
  double* d = nullptr;
  HIP_CHECK(hipMalloc(&d, 3 * sizeof(double)));
  hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, st, iters, d);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(host_out3, d, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  (void)hipFree(d);
}

// Where the workgroups of a launch on a CU-masked stream run: out[b] = XCC_ID (0-7) | CU id << 8 of workgroup b's wave 0
// (HW_REG_XCC_ID; HW_REG_HW_ID: CU_ID bits 11:8, SH_ID bit 12, SE_ID bits 15:13).  Every workgroup takes `lds_bytes` of
// LDS and waits `spin_us`, so that a grid larger than the mask's CUs shows up as a second round, not as sharing.
__global__ void k_where(int* out, int spin_us) {
  extern __shared__ char hold[];
  if (threadIdx.x == 0) {
    unsigned xcc = 0, hw = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    hold[0] = 1;
    out[blockIdx.x] = (int)((xcc & 0xF) | (((hw >> 8) & 0xFF) << 8));
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
  }
}
void where_probe(const uint32_t* mask, int nwords, int nblocks, size_t lds_bytes, int spin_us, int* host_out) {
  hipStream_t st = nullptr;
  if (mask && nwords > 0) HIP_CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t)nwords, mask));
  else HIP_CHECK(hipStreamCreate(&st));
  int* d = nullptr;
  HIP_CHECK(hipMalloc(&d, (size_t)nblocks * sizeof(int)));
  if (lds_bytes > 65536)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_where), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k_where, dim3(nblocks), dim3(64), lds_bytes, st, d, spin_us);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(host_out, d, (size_t)nblocks * sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  (void)hipFree(d);
  (void)hipStreamDestroy(st);
}

void copy2d(hipStream_t st, zc* dst, long ldd, const zc* src, long lds, long rows, int cols, int zero_to, zc a,
            bool accumulate) {
  const long n = rows * (long)std::max(cols, zero_to);
  if (n <= 0) return;
  LAUNCH(k_copy2d, vec_blocks(n), st, dst, ldd, src, lds, rows, cols, zero_to, a, accumulate ? 1 : 0);
}
__global__ __launch_bounds__(256) void k_gather_blocks(zc* __restrict__ dst, int n_dst, const zc* __restrict__ src, int m_src,
                                                      long rows, int cols, BlockList bl) {
  const long per = (long)bl.n * cols, tot = rows * per;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const long a = e / per;
    const int r = (int)(e - a * per), k = r / cols, c = r - k * cols;
    dst[(a * n_dst + k) * cols + c] = src[(a * m_src + bl.idx[k]) * cols + c];
  }
}
__global__ __launch_bounds__(256) void k_fill_scaled_blocks(zc* __restrict__ X, long ldx, int pos0, const zc* __restrict__ sig,
                                                           long rows, int cols, BlockList bl) {
  const long per = (long)bl.n * cols, tot = rows * per;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const long a = e / per;
    const int r = (int)(e - a * per), k = r / cols, c = r - k * cols;
    const zc v = sig[a * cols + c], f = bl.f[k];
    X[a * ldx + (long)(pos0 + k) * cols + c] = make_double2(f.x * v.x - f.y * v.y, f.x * v.y + f.y * v.x);
  }
}
__global__ __launch_bounds__(256) void k_accum_scaled_blocks(zc* __restrict__ out, const zc* __restrict__ X, long ldx,
                                                            const zc* __restrict__ sig, long rows, int cols, BlockList bl, zc both) {
  const long tot = rows * cols;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const long a = e / cols;
    const int c = (int)(e - a * cols);
    zc o = out[e];
    const zc s0 = sig[e];
    o.x += both.x * s0.x - both.y * s0.y;
    o.y += both.x * s0.y + both.y * s0.x;
    for (int k = 0; k < bl.n; ++k) {  // fixed order
      const zc v = X[a * ldx + (long)bl.idx[k] * cols + c], f = bl.f[k];
      o.x += f.x * v.x - f.y * v.y;
      o.y += f.x * v.y + f.y * v.x;
    }
    out[e] = o;
  }
}
void gather_blocks(hipStream_t st, zc* dst, int n_dst, const zc* src, int m_src, long rows, int cols, const BlockList& bl) {
  if (bl.n > 0 && rows > 0) LAUNCH(k_gather_blocks, vec_blocks(rows * bl.n * cols), st, dst, n_dst, src, m_src, rows, cols, bl);
}
void fill_scaled_blocks(hipStream_t st, zc* X, long ldx, int pos0, const zc* sig, long rows, int cols, const BlockList& bl) {
  if (bl.n > 0 && rows > 0) LAUNCH(k_fill_scaled_blocks, vec_blocks(rows * bl.n * cols), st, X, ldx, pos0, sig, rows, cols, bl);
}
void accum_scaled_blocks(hipStream_t st, zc* out, const zc* X, long ldx, const zc* sig, long rows, int cols, const BlockList& bl,
                         zc both) {
  if (rows > 0) LAUNCH(k_accum_scaled_blocks, vec_blocks(rows * cols), st, out, X, ldx, sig, rows, cols, bl, both);
}

void col_sumsq(hipStream_t st, const zc* x, long rows, int cols, double* out) {
  if (cols > 0) LAUNCH(k_col_sumsq, cols, st, x, rows, cols, out);
}
void row_sumsq(hipStream_t st, const zc* x, int rows, long cols, double* out) {
  if (rows > 0) LAUNCH(k_row_sumsq, rows, st, x, rows, cols, out);
}
void shell_sumsq(hipStream_t st, const zc* x, int n, double* out) {
  if (n > 0) LAUNCH(k_shell_sumsq, n, st, x, n, out);
}

void transpose_batched(hipStream_t st, const zc* in, zc* out, int rows, int cols, long ldi, long ldo, int batch,
                       long in_bs, long out_bs) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return;
  if (batch > 65535) throw ArgError("transpose: batch too large");
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, st, in, out, rows, cols, ldi, ldo, in_bs, out_bs);
  HIP_CHECK(hipGetLastError());
}

void phys_diag(hipStream_t st, const zc* C, zc* out, int dl, int n, int dr, bool trace) {
  const long tot = trace ? (long)dl * dr : (long)dl * n * dr;
  LAUNCH(k_phys_diag, (int)std::min<long>(4096, (tot + 255) / 256), st, C, out, dl, n, dr, trace ? 1 : 0);
}

void permute_0213(hipStream_t st, const zc* in, zc* out, long n0, int n1, int n2, int n3) {
  const long tot = n0 * n1 * n2 * n3;
  if (tot <= 0) return;
  LAUNCH(k_permute_0213, (int)std::min<long>(4096, (tot + 255) / 256), st, in, out, n0, n1, n2, n3);
}

void permute5(hipStream_t st, const zc* in, zc* out, const int dims[5], const long in_strides[5], const int* map2) {
  Perm5 p;
  long tot = 1;
  for (int i = 0; i < 5; ++i) { p.d[i] = dims[i]; p.s[i] = in_strides[i]; tot *= dims[i]; }
  if (tot <= 0) return;
  LAUNCH(k_permute5, (int)std::min<long>(4096, (tot + 255) / 256), st, in, out, p, map2);
}
void scale_cols(hipStream_t st, zc* x, long rows, int cols, long ld, const double* sc_dev) {
  if (rows * cols > 0) LAUNCH(k_scale_cols, vec_blocks(rows * cols), st, x, rows, cols, ld, sc_dev);
}

// out[s][j][a] = in[a][j][s]   (in: (na, nj, ns) C order)
void transpose_rev3(hipStream_t st, const zc* in, zc* out, int na, int nj, int ns) {
  // for every j: matrix in_j[a][s] (row stride nj*ns) -> out_j[s][a] (row stride nj*na)
  transpose_batched(st, in, out, na, ns, (long)nj * ns, (long)nj * na, nj, ns, na);
}

__global__ __launch_bounds__(256) void k_copy_raw(zc* __restrict__ dst, const zc* __restrict__ src, size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) dst[e] = src[e];
}
void vec_copy_raw(hipStream_t st, zc* dst, const zc* src, size_t n) {
  if (!n) return;
  const size_t nb = std::min<size_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_copy_raw, dim3((unsigned)nb), dim3(256), 0, st, dst, src, n);
  HIP_CHECK(hipGetLastError());
}

// the workgroup's maximum -> ONE atomic per workgroup.  (One per wave, all on the same word, was the whole cost of these
// kernels: atomics on one address serialise at ~130 ns each -- 3 x 1024 of them made the masked check of the three identity
// states of a C5 site 134 us for 12.6 MB, 1.5 % of a C5 sweep in profiles/r04_c5_kernel_stats.csv.)
__device__ __forceinline__ void ident_block_max(double m, unsigned long long* out) {
  __shared__ double wmax[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
    // non-negative doubles order like their bit patterns
    if (t > 0.0) atomicMax(out, (unsigned long long)__double_as_longlong(t));
  }
}

// fmax drops a NaN: a non-finite element must make the block FAIL the identity test, not pass it with deviation 0 (the
// edge form would then replace a corrupted environment block by a clean identity multiple and hide the corruption)
__device__ __forceinline__ double finite_or_huge(zc v, double dev) {
  return (v.x - v.x == 0.0 && v.y - v.y == 0.0 && dev == dev) ? dev : 1e308;  // x - x is NaN for NaN and +-Inf
}

__global__ __launch_bounds__(256) void k_ident_dev(const zc* __restrict__ blk, long ld, int n, unsigned long long* __restrict__ out) {
  double m = 0.0;
  const long tot = (long)n * n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    const zc v = blk[(long)r * ld + c];
    const double dv = fmax(fabs(v.x - (r == c ? 1.0 : 0.0)), fabs(v.y));
    m = fmax(m, finite_or_huge(v, dv));
  }
  ident_block_max(m, out);
}
// nblk blocks at once, against MULTIPLES of the identity: block c starts at base + c * blk_stride; lam[c] = its first
// diagonal element, out[c] = max |blk - lam * 1|
__global__ __launch_bounds__(256) void k_ident_dev_multi(const zc* __restrict__ base, long blk_stride, long ld, int n,
                                                         unsigned long long* __restrict__ out, zc* __restrict__ lam,
                                                         unsigned long long mask) {
  if (!((mask >> blockIdx.y) & 1ull)) return;  // not asked for: out[c] keeps its 0, the caller ignores it
  const zc* blk = base + (long)blockIdx.y * blk_stride;
  const zc l = blk[0];
  if (blockIdx.x == 0 && threadIdx.x == 0) lam[blockIdx.y] = l;
  double m = 0.0;
  const long tot = (long)n * n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    const zc v = blk[(long)r * ld + c];
    const double dv = r == c ? fmax(fabs(v.x - l.x), fabs(v.y - l.y)) : fmax(fabs(v.x), fabs(v.y));
    m = fmax(m, finite_or_huge(v, dv));
  }
  ident_block_max(m, out + blockIdx.y);
}
void ident_deviation_multi(hipStream_t st, const zc* base, int nblk, long blk_stride, long ld, int n, double* out_dev,
                           zc* lam_dev, unsigned long long mask, bool clear) {
  if (nblk < 1) return;
  if (clear) HIP_CHECK(hipMemsetAsync(out_dev, 0, (size_t)nblk * sizeof(double), st));
  const long tot = (long)n * n;
  const int nb = (int)std::min<long>((tot + 2047) / 2048, 64);  // eight elements per thread and pass, <= 64 atomics per block
  hipLaunchKernelGGL(k_ident_dev_multi, dim3(nb, nblk), dim3(256), 0, st, base, blk_stride, ld, n,
                     reinterpret_cast<unsigned long long*>(out_dev), lam_dev, mask);
  HIP_CHECK(hipGetLastError());
}
void ident_deviation(hipStream_t st, const zc* blk, long ld, int n, double* out_dev) {
  HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(double), st));
  const long tot = (long)n * n;
  const int nb = (int)std::min<long>((tot + 2047) / 2048, 256);
  hipLaunchKernelGGL(k_ident_dev, dim3(nb), dim3(256), 0, st, blk, ld, n, reinterpret_cast<unsigned long long*>(out_dev));
  HIP_CHECK(hipGetLastError());
}

}  // namespace mitdvp
