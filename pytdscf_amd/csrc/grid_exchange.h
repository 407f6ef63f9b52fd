// grid_exchange.h -- exchange of a few doubles between ALL workgroups of a running launch (persistent
// kernels: qr.hip's panel kernel; small_site.hip carries its own specialisation of the same protocol).
//
// Protocol (MI355X_MICROARCH.md "Workgroup dispatch ... visibility", valid forms; price list "allgather";
// cdna_hip_programming.md Guideline 16, recipe R2): the data is the flag.  Every double travels as two 8-byte
// granules {tag = epoch, 32 payload bits}, each written by ONE agent-scope (sc1) store; readers re-read a
// granule with agent-scope loads until it carries this epoch's tag.  Bulk data handed over at the same
// point is stored with agent-scope stores and drained (s_waitcnt vmcnt(0) by every storing wave, then the
// workgroup barrier) BEFORE the granules are written, and read with agent-scope loads afterwards.  Two
// granule buffers alternate (a writer is at most one epoch ahead of the slowest reader); tags are unique
// per launch (epoch0 from a host-side launch counter), so nothing needs re-zeroing between launches.
// Every wait is bounded (2 s) and raises the shared abort flag instead of hanging.
#pragma once
#include <hip/hip_runtime.h>

namespace mitdvp {

#define GX_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ double gx_ld(const double* p) { return __hip_atomic_load(p, GX_RLX); }
__device__ __forceinline__ void gx_st(double* p, double v) { __hip_atomic_store(p, v, GX_RLX); }
__device__ __forceinline__ double2 gx_ldz(const double2* p) {
  const double* q = reinterpret_cast<const double*>(p);
  return make_double2(gx_ld(q), gx_ld(q + 1));
}
__device__ __forceinline__ void gx_stz(double2* p, double2 v) {
  double* q = reinterpret_cast<double*>(p);
  gx_st(q, v.x);
  gx_st(q + 1, v.y);
}

constexpr int GX_MAXG = 64;    // workgroups
constexpr int GX_MAXPAY = 64;  // doubles per workgroup and exchange
// granules needed: 2 buffers x (2 * GX_MAXPAY) x GX_MAXG
constexpr size_t GX_GRANULES = (size_t)2 * 2 * GX_MAXPAY * GX_MAXG;

struct GxSync {
  unsigned long long* gran;
  unsigned* abort_w;
  int G, wg;
  unsigned epoch;
};

// pay[0..npay) (LDS) of every workgroup -> red[c] = sum over workgroups, in workgroup order, identical
// everywhere.  val: LDS scratch of G * npay doubles.  NT = workgroup size.  Returns false on abort.
template <int NT>
__device__ bool gx_exchange(GxSync& s, const double* pay, int npay, double* red, double* val) {
  s.epoch += 1;
  const int tid = threadIdx.x;
  unsigned long long* base = s.gran + (size_t)(s.epoch & 1u) * (2 * GX_MAXPAY) * GX_MAXG;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its agent-scope stores of bulk data have landed
  __syncthreads();
  for (int t = tid; t < 2 * npay; t += NT) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(pay[t >> 1]);
    const unsigned half = (t & 1) ? (unsigned)(b >> 32) : (unsigned)b;
    __hip_atomic_store(base + (size_t)t * GX_MAXG + s.wg, ((unsigned long long)s.epoch << 32) | half, GX_RLX);
  }
  int ok = 1;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  for (int it = tid; it < s.G * npay && ok; it += NT) {
    const int comp = it / s.G, g = it - comp * s.G;  // neighbouring lanes poll neighbouring words
    const unsigned long long* p0 = base + (size_t)(2 * comp) * GX_MAXG + g;
    const unsigned long long* p1 = p0 + GX_MAXG;
    unsigned long long x0, x1;
    unsigned spins = 0;
    for (;;) {
      x0 = __hip_atomic_load(p0, GX_RLX);
      x1 = __hip_atomic_load(p1, GX_RLX);
      if ((unsigned)(x0 >> 32) == s.epoch && (unsigned)(x1 >> 32) == s.epoch) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 255u) == 0u) {
        if (__hip_atomic_load(s.abort_w, GX_RLX) != 0u) { ok = 0; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
          __hip_atomic_store(s.abort_w, 1u, GX_RLX);
          ok = 0;
          break;
        }
      }
    }
    val[g * npay + comp] = __longlong_as_double((long long)((x0 & 0xffffffffull) | (x1 << 32)));
  }
  if (!__syncthreads_and(ok)) return false;
  for (int c = tid; c < npay; c += NT) {
    double acc = 0.0;
    for (int g = 0; g < s.G; ++g) acc += val[g * npay + c];
    red[c] = acc;
  }
  __syncthreads();
  return true;
}

}  // namespace mitdvp
