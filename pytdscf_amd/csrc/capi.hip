// capi.hip -- extern "C" surface declared in include/mitdvp.h.
#include <mutex>
#include <thread>
#include <vector>

#include "capi_internal.h"
#include "engine_internal.h"
#include "engine_krylov.inc"

namespace {
thread_local std::string g_err;

template <class F>
int guard(mitdvp_engine* h, F&& f) {
  std::string* dst = h ? &h->err : &g_err;
  try {
    if (h) {
      hipError_t e = hipSetDevice(h->device);
      if (e != hipSuccess) { *dst = hipGetErrorString(e); return MITDVP_EHIP; }
    }
    f();
    return MITDVP_OK;
  } catch (const mitdvp::NotConverged& ex) {
    *dst = ex.what();
    return MITDVP_ENOTCONV;
  } catch (const mitdvp::ArgError& ex) {
    *dst = ex.what();
    return MITDVP_EINVAL;
  } catch (const mitdvp::HipError& ex) {
    *dst = ex.what();
    return MITDVP_EHIP;
  } catch (const std::bad_alloc&) {
    *dst = "out of host memory";
    return MITDVP_ENOMEM;
  } catch (const std::exception& ex) {
    *dst = ex.what();
    return MITDVP_ESTATE;
  } catch (...) {  // nothing may leave an extern "C" entry point or a worker thread of mitdvp_ensemble_step
    *dst = "unknown exception";
    return MITDVP_ESTATE;
  }
}

mitdvp_config unit_cfg(int device) {
  mitdvp_config c{};
  c.nsite = 1;
  c.device = device;
  c.integrator = MITDVP_LANCZOS;
  c.conserve_norm = 1;
  c.thresh = 1e-9;
  c.max_krylov = 20;
  return c;
}

struct Dev {
  mitdvp::DevBuf b;
  Dev(hipStream_t st, const double* host, size_t elems) {
    b.reserve(std::max<size_t>(elems, 1));
    if (host && elems) HIP_CHECK(hipMemcpyAsync(b.p, host, elems * sizeof(mitdvp::zc), hipMemcpyHostToDevice, st));
  }
  explicit Dev(size_t elems) { b.reserve(std::max<size_t>(elems, 1)); }
  mitdvp::zc* p() { return b.p; }
};

void to_host(hipStream_t st, double* dst, const mitdvp::zc* src, size_t elems) {
  HIP_CHECK(hipMemcpyAsync(dst, src, elems * sizeof(mitdvp::zc), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
}
}  // namespace

extern "C" {

const char* mitdvp_version(void) { return "mitdvp 0.2 (gfx950)"; }

int mitdvp_device_count(int* count) {
  if (!count) { g_err = "null argument"; return MITDVP_EINVAL; }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;  // no driver / no GPU: 0 devices, not an error
  (void)hipGetLastError();
  *count = n;
  return MITDVP_OK;
}

int mitdvp_device_cu_count(int device, int* count) {
  return guard(nullptr, [&] {
    if (!count) throw mitdvp::ArgError("null pointer argument");
    HIP_CHECK(hipDeviceGetAttribute(count, hipDeviceAttributeMultiprocessorCount, device));
  });
}

int mitdvp_device_sync(int device) {
  return guard(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    HIP_CHECK(hipDeviceSynchronize());
  });
}

int mitdvp_create(const mitdvp_config* cfg, mitdvp_engine** out) {
  if (!cfg || !out) { g_err = "null argument"; return MITDVP_EINVAL; }
  *out = nullptr;
  auto* h = new mitdvp_engine();
  h->device = cfg->device;
  int rc = guard(nullptr, [&] { h->e.reset(new mitdvp::Engine(*cfg)); });
  if (rc != MITDVP_OK) { delete h; return rc; }
  *out = h;
  return MITDVP_OK;
}
void mitdvp_destroy(mitdvp_engine* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  delete h;
}
const char* mitdvp_last_error(const mitdvp_engine* h) { return h ? h->err.c_str() : g_err.c_str(); }

#define ENG_CALL(h, body)                                   \
  if (!(h) || !(h)->e) { g_err = "null handle"; return MITDVP_EINVAL; } \
  return guard((h), [&] { body; (h)->e->check_device_errors(); })
// argument pointers that must not be NULL (reported as MITDVP_EINVAL instead of a crash)
#define NEED(...)                                                       \
  do {                                                                  \
    const void* ptrs_[] = {__VA_ARGS__};                                \
    for (const void* q_ : ptrs_)                                        \
      if (!q_) throw mitdvp::ArgError("null pointer argument");         \
  } while (0)

int mitdvp_set_site(mitdvp_engine* h, int isite, const double* reim, int l, int n, int r, int gauge) {
  ENG_CALL(h, { NEED(reim); h->e->set_site(isite, reim, l, n, r, gauge); });
}
int mitdvp_get_site_shape(mitdvp_engine* h, int isite, int* l, int* n, int* r, int* gauge) {
  ENG_CALL(h, { NEED(l, n, r, gauge); h->e->get_site_shape(isite, l, n, r, gauge); });
}
int mitdvp_get_site(mitdvp_engine* h, int isite, double* out) { ENG_CALL(h, { NEED(out); h->e->get_site(isite, out); }); }
int mitdvp_init_random(mitdvp_engine* h, const int* dims, int bond_dim, uint64_t seed) {
  ENG_CALL(h, { NEED(dims); h->e->init_random(dims, bond_dim, seed); });
}
int mitdvp_init_random_block(mitdvp_engine* h, const int* dims, int nsite_total, int first, int bond_dim, uint64_t seed, int balance) {
  ENG_CALL(h, { NEED(dims); h->e->init_random_block(dims, nsite_total, first, bond_dim, seed, balance != 0); });
}
int mitdvp_canonicalize(mitdvp_engine* h, double scale) { ENG_CALL(h, h->e->canonicalize(scale)); }
int mitdvp_set_mpo_core(mitdvp_engine* h, int op_id, int isite, const double* reim, int ml, int d_out, int d_in,
                        int mr) {
  ENG_CALL(h, { NEED(reim); h->e->set_mpo_core(op_id, isite, reim, ml, d_out, d_in, mr); });
}
int mitdvp_set_shift(mitdvp_engine* h, int op_id, double re, double im) { ENG_CALL(h, h->e->set_shift(op_id, re, im)); }
int mitdvp_step(mitdvp_engine* h, double dt) { ENG_CALL(h, h->e->step(dt)); }
int mitdvp_sweep(mitdvp_engine* h, double dt, int forward) { ENG_CALL(h, h->e->sweep(dt, forward != 0)); }
int mitdvp_ensemble_step(mitdvp_engine** hs, int n, double dt, int nsteps, int* statuses) {
  if (!hs || n < 1 || nsteps < 0) { g_err = "mitdvp_ensemble_step: bad arguments"; return MITDVP_EINVAL; }
  for (int i = 0; i < n; ++i)
    if (!hs[i] || !hs[i]->e) { g_err = "mitdvp_ensemble_step: null engine"; return MITDVP_EINVAL; }
  for (int i = 0; i < n; ++i)  // a handle is driven by one host thread at a time
    for (int k = 0; k < i; ++k)
      if (hs[k] == hs[i]) { g_err = "mitdvp_ensemble_step: the same engine is listed twice"; return MITDVP_EINVAL; }
  std::vector<int> rc((size_t)n, MITDVP_OK);
  std::vector<std::thread> th;
  th.reserve((size_t)n);
  auto run = [&](int i) {
    mitdvp_engine* h = hs[i];
    rc[(size_t)i] = guard(h, [&] {
      for (int s = 0; s < nsteps; ++s) h->e->step(dt);
      h->e->check_device_errors();
      HIP_CHECK(hipStreamSynchronize(h->e->stream()));
    });
  };
  for (int i = 0; i < n; ++i) {
    try {
      th.emplace_back(run, i);
    } catch (const std::exception&) {  // no more host threads: this replica runs on the caller's thread, after the others
      for (auto& t : th) t.join();
      th.clear();
      for (int k = i; k < n; ++k) run(k);
      break;
    }
  }
  for (auto& t : th) t.join();
  int first = MITDVP_OK;
  for (int i = 0; i < n; ++i) {
    if (statuses) statuses[i] = rc[(size_t)i];
    if (first == MITDVP_OK && rc[(size_t)i] != MITDVP_OK) first = rc[(size_t)i];
  }
  return first;
}
int mitdvp_invalidate_env(mitdvp_engine* h) { ENG_CALL(h, h->e->invalidate_env()); }
int mitdvp_replace_site(mitdvp_engine* h, int isite, const double* reim, int gauge) {
  ENG_CALL(h, { NEED(reim); h->e->replace_site(isite, reim, gauge); });
}
int mitdvp_set_boundary_env(mitdvp_engine* h, int side, const double* reim, int d, int m) {
  ENG_CALL(h, { NEED(reim); h->e->set_boundary_env(side, reim, d, m); });
}
int mitdvp_get_env(mitdvp_engine* h, int side, int bond, double* reim_out, int* d, int* m) {
  ENG_CALL(h, { NEED(d, m); h->e->env_shape(side, bond, d, m); if (reim_out) h->e->get_env(side, bond, reim_out); });
}
int mitdvp_build_envs(mitdvp_engine* h, int side) { ENG_CALL(h, h->e->build_envs(side)); }
int mitdvp_site_exp(mitdvp_engine* h, double dt) { ENG_CALL(h, h->e->site_exp(dt)); }
int mitdvp_split_center(mitdvp_engine* h, int forward) { ENG_CALL(h, h->e->split_center(forward != 0)); }
int mitdvp_bond_exp(mitdvp_engine* h, double dt) { ENG_CALL(h, h->e->bond_exp(dt)); }
int mitdvp_absorb_bond(mitdvp_engine* h, int forward) { ENG_CALL(h, h->e->absorb_bond(forward != 0)); }
int mitdvp_get_bond(mitdvp_engine* h, double* reim_out, int* dim) { ENG_CALL(h, { NEED(dim); h->e->get_bond(reim_out, dim); }); }
int mitdvp_set_bond(mitdvp_engine* h, int bond, const double* reim, int dim) {
  ENG_CALL(h, { NEED(reim); h->e->set_bond(bond, reim, dim); });
}
int mitdvp_fold_block_range(mitdvp_engine* h, int op_id, int conj_bra, int from_left, int first, int count, const double* reim_in,
                            int d, int m, double* reim_out) {
  ENG_CALL(h, { NEED(reim_in, reim_out); h->e->fold_block(op_id, conj_bra != 0, from_left != 0, reim_in, d, m, reim_out, first, count); });
}
int mitdvp_site_rdm_blocks(mitdvp_engine* h, int isite, const double* left, const double* right, double* reim_out) {
  ENG_CALL(h, { NEED(left, right, reim_out); h->e->site_rdm_blocks(isite, left, right, reim_out); });
}
int mitdvp_set_qr_fast(int on) { mitdvp::qr_set_fast(on); return MITDVP_OK; }
int mitdvp_get_qr_fast(void) { return mitdvp::qr_get_fast(); }
int mitdvp_heff_apply_center(mitdvp_engine* h, const double* reim_in, double* reim_out, int* flags) {
  ENG_CALL(h, { NEED(reim_out); h->e->heff_apply_center(reim_in, reim_out, flags); });
}
int mitdvp_get_krylov_memory(mitdvp_engine* h, int isite, int* k) { ENG_CALL(h, { NEED(k); *k = h->e->kprev_get(isite); }); }
int mitdvp_set_krylov_memory(mitdvp_engine* h, int isite, int k) { ENG_CALL(h, h->e->kprev_set(isite, k)); }
int mitdvp_set_small_kernels(mitdvp_engine* h, int on) { ENG_CALL(h, h->e->set_small_kernels(on != 0)); }
int mitdvp_set_pointer_mode(mitdvp_engine* h, int mode) { ENG_CALL(h, h->e->set_pointer_mode(mode)); }
int mitdvp_fold_block(mitdvp_engine* h, int op_id, int conj_bra, int from_left, const double* reim_in, int d, int m,
                      double* reim_out) {
  ENG_CALL(h, { NEED(reim_in, reim_out); h->e->fold_block(op_id, conj_bra != 0, from_left != 0, reim_in, d, m, reim_out); });
}
int mitdvp_expect(mitdvp_engine* h, int op_id, double out[2]) {
  ENG_CALL(h, { NEED(out); auto v = h->e->expect(op_id); out[0] = v.real(); out[1] = v.imag(); });
}
int mitdvp_autocorr(mitdvp_engine* h, double out[2]) {
  ENG_CALL(h, { NEED(out); auto v = h->e->autocorr(); out[0] = v.real(); out[1] = v.imag(); });
}
int mitdvp_norm(mitdvp_engine* h, double* out) { ENG_CALL(h, { NEED(out); *out = h->e->norm(); }); }
int mitdvp_site_rdm(mitdvp_engine* h, int isite, double* out) { ENG_CALL(h, { NEED(out); h->e->site_rdm(isite, out); }); }
int mitdvp_reduced_density(mitdvp_engine* h, const int* remain_nleg, int nlen, double* out, size_t* n_out) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] {
    if (!n_out || !remain_nleg) throw mitdvp::ArgError("null argument");
    if (!out) {  // size query: product of the kept physical dimensions
      size_t n = 1;
      for (int p = 0; p < nlen; ++p) {
        int l = 0, d = 0, r = 0, g = 0;
        h->e->get_site_shape(p, &l, &d, &r, &g);
        for (int q = 0; q < remain_nleg[p]; ++q) n *= (size_t)d;
      }
      *n_out = n;
      return;
    }
    std::vector<mitdvp::hzc> v;
    std::vector<int> shp;
    h->e->reduced_density(remain_nleg, nlen, v, shp);
    if (v.size() > *n_out) throw mitdvp::ArgError("output buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(mitdvp::hzc));
    *n_out = v.size();
  });
}
int mitdvp_truncate_bond(mitdvp_engine* h, double p, int max_dim, int* new_dim, double* svals_out) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] {
    std::vector<double> sv;
    const int n = h->e->truncate_bond(p, max_dim, sv);
    if (new_dim) *new_dim = n;
    if (svals_out) std::memcpy(svals_out, sv.data(), sv.size() * sizeof(double));
  });
}
int mitdvp_set_gate(mitdvp_engine* h, int isite, const double* U_reim, int d) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { h->e->set_gate(isite, U_reim, d); });
}
int mitdvp_apply_gates(mitdvp_engine* h) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { h->e->apply_gates(); });
}
int mitdvp_set_kraus(mitdvp_engine* h, int isite, int two_site, const double* B_reim, int k, int d) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { h->e->set_kraus(isite, two_site, B_reim, k, d); });
}
int mitdvp_apply_kraus(mitdvp_engine* h) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { h->e->apply_kraus(); });
}
int mitdvp_rccl_unique_id(char out[128]) {
  return guard(nullptr, [&] { NEED(out); mitdvp::Engine::rccl_unique_id(out); });
}
int mitdvp_set_parallel_rccl(mitdvp_engine* h, int nranks, int rank, const char id[128]) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { NEED(id); h->e->set_parallel_rccl(nranks, rank, id); });
}
int mitdvp_rccl_selftest(mitdvp_engine* h, int* mismatches) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { NEED(mismatches); *mismatches = h->e->rccl_selftest(); });
}
int mitdvp_save_reference(mitdvp_engine* h) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { h->e->save_reference(); });
}
int mitdvp_overlap_reference(mitdvp_engine* h, double out[2]) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] {
    NEED(out);
    const mitdvp::hzc v = h->e->overlap_reference();
    out[0] = v.real(); out[1] = v.imag();
  });
}
int mitdvp_operate(mitdvp_engine* h, int op_id, int maxstep, double conv_tol, double* norm_out, int* iters_out) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] {
    const double n = h->e->operate(op_id, maxstep, conv_tol, iters_out);
    if (norm_out) *norm_out = n;
  });
}
// ---- several electronic states (engine_multi.hip) ---------------------------------------------
int mitdvp_ms_configure(mitdvp_engine* h, int nstate) { ENG_CALL(h, { h->e->ms_configure(nstate); }); }
int mitdvp_ms_set_site(mitdvp_engine* h, int istate, int isite, const double* reim, int l, int n, int r, int gauge) {
  ENG_CALL(h, { NEED(reim); h->e->ms_set_site(istate, isite, reim, l, n, r, gauge); });
}
int mitdvp_ms_get_site_shape(mitdvp_engine* h, int istate, int isite, int* l, int* n, int* r, int* gauge) {
  ENG_CALL(h, { NEED(l, n, r, gauge); h->e->ms_get_site_shape(istate, isite, l, n, r, gauge); });
}
int mitdvp_ms_get_site(mitdvp_engine* h, int istate, int isite, double* out) {
  ENG_CALL(h, { NEED(out); h->e->ms_get_site(istate, isite, out); });
}
int mitdvp_ms_canonicalize(mitdvp_engine* h, int istate, double scale) { ENG_CALL(h, { h->e->ms_canonicalize(istate, scale); }); }
int mitdvp_ms_set_mpo_core(mitdvp_engine* h, int op_id, int ibra, int iket, int isite, const double* reim, int ml, int d_out,
                           int d_in, int mr) {
  ENG_CALL(h, { NEED(reim); h->e->ms_set_mpo_core(op_id, ibra, iket, isite, reim, ml, d_out, d_in, mr); });
}
int mitdvp_ms_set_coupleJ(mitdvp_engine* h, int op_id, int ibra, int iket, double re, double im) {
  ENG_CALL(h, { h->e->ms_set_couplej(op_id, ibra, iket, re, im); });
}
int mitdvp_ms_step(mitdvp_engine* h, double dt_au) { ENG_CALL(h, { h->e->ms_step(dt_au); }); }
int mitdvp_ms_expect(mitdvp_engine* h, int op_id, double out[2]) {
  ENG_CALL(h, { NEED(out); auto v = h->e->ms_expect(op_id); out[0] = v.real(); out[1] = v.imag(); });
}
int mitdvp_ms_autocorr(mitdvp_engine* h, double out[2]) {
  ENG_CALL(h, { NEED(out); auto v = h->e->ms_autocorr(); out[0] = v.real(); out[1] = v.imag(); });
}
int mitdvp_ms_operate(mitdvp_engine* h, int op_id, int maxstep, double conv_tol, double* norm_out, int* iters_out) {
  ENG_CALL(h, {
    const double n = h->e->ms_operate(op_id, maxstep, conv_tol, iters_out);
    if (norm_out) *norm_out = n;
  });
}
int mitdvp_ms_pops(mitdvp_engine* h, double* out) { ENG_CALL(h, { NEED(out); h->e->ms_pops(out); }); }

int mitdvp_set_adaptive(mitdvp_engine* h, int enable, int dmax, int dd, double p_proj) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] { h->e->set_adaptive(enable != 0, dmax, dd, p_proj); });
}
int mitdvp_thin_to_full(int device, int gauge, const double* site, int l, int c, int r, int extra, double* out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    if (l < 1 || c < 1 || r < 1 || extra < 0) throw ArgError("thin_to_full: bad shape");
    const long room = gauge == 0 ? (long)l * c - r : (long)c * r - l;
    if (room < 0) throw ArgError("thin_to_full: the tensor cannot be an isometry in that gauge");
    if (extra > room) throw ArgError("thin_to_full: extra exceeds the dimension of the orthogonal complement");
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    const long ns = (long)l * c * r;
    const long nf = gauge == 0 ? (long)l * c * (r + extra) : (long)(l + extra) * c * r;
    e.ensure_work(nf, 1, 1, gauge == 0 ? l * c : c * r, gauge == 0 ? r : l, extra + 1);
    Dev din(st, site, ns), dout(nf);
    if (gauge == 0) e.thin_to_full_A(din.p(), l, c, r, extra, dout.p());
    else e.thin_to_full_B(din.p(), l, c, r, extra, dout.p());
    to_host(st, out, dout.p(), nf);
  });
}
int mitdvp_svd(int device, const double* A, int r, int c, double* U, double* S, double* Vh, int* sweeps) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    HIP_CHECK(hipSetDevice(device));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    {
      const int k = std::min(r, c);
      Dev dA(st, A, (size_t)r * c), dU((size_t)r * k), dV((size_t)k * c), work(svd_work_elems(r, c));
      svd_jacobi(st, dA.p(), r, c, dU.p(), S, dV.p(), work.p(), sweeps);
      to_host(st, U, dU.p(), (size_t)r * k);
      to_host(st, Vh, dV.p(), (size_t)k * c);
    }
    HIP_CHECK(hipStreamDestroy(st));
  });
}
int mitdvp_set_trace_op_core(mitdvp_engine* h, int op_id, int isite, const double* reim, int ml, int n, int mr) {
  ENG_CALL(h, { NEED(reim); h->e->set_trace_op_core(op_id, isite, reim, ml, n, mr); });
}
int mitdvp_expect_trace(mitdvp_engine* h, int op_id, double out[2]) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] {
    NEED(out);
    const auto v = h->e->expect_trace(op_id);
    out[0] = v.real();
    out[1] = v.imag();
  });
}
int mitdvp_partial_trace(mitdvp_engine* h, const int* remain_nleg, int nlen, double* out, size_t* n_out) {
  if (!h || !h->e) { g_err = "null handle"; return MITDVP_EINVAL; }
  return guard(h, [&] {
    if (!n_out || !remain_nleg) throw mitdvp::ArgError("null argument");
    if (!out) {
      int center = -1;
      for (int p = 0; p < nlen; ++p)
        if (remain_nleg[p]) center = p;
      size_t n = 1;
      for (int p = 0; p <= center; ++p) {
        int l = 0, d = 0, r = 0, g = 0;
        h->e->get_site_shape(p, &l, &d, &r, &g);
        (void)d;
        const size_t nn = (size_t)h->e->liouville_n(p);
        const int k = p == center ? 2 : remain_nleg[p];
        for (int q = 0; q < k; ++q) n *= nn;
      }
      *n_out = n;
      return;
    }
    std::vector<mitdvp::hzc> v;
    h->e->partial_trace(remain_nleg, nlen, v);
    if (v.size() > *n_out) throw mitdvp::ArgError("output buffer too small");
    std::memcpy(out, v.data(), v.size() * sizeof(mitdvp::hzc));
    *n_out = v.size();
  });
}
int mitdvp_set_subspace(mitdvp_engine* h, int isite, int n, const int* inds, int ninds) {
  ENG_CALL(h, h->e->set_subspace(isite, n, inds, ninds));
}
int mitdvp_hermitise(mitdvp_engine* h) { ENG_CALL(h, h->e->hermitise()); }
int mitdvp_krylov_stats(mitdvp_engine* h, int* per_site) { ENG_CALL(h, { NEED(per_site); h->e->krylov_stats(per_site); }); }
int mitdvp_counters_get(mitdvp_engine* h, mitdvp_counters* out) { ENG_CALL(h, { NEED(out); h->e->counters_get(out); }); }
int mitdvp_counters_reset(mitdvp_engine* h) { ENG_CALL(h, h->e->counters_reset()); }
int mitdvp_set_profiling(mitdvp_engine* h, int on) { ENG_CALL(h, h->e->set_profiling(on != 0)); }
int mitdvp_set_parallel(mitdvp_engine* h, int nranks, int rank, mitdvp_collective_fn fn, void* user) {
  ENG_CALL(h, h->e->set_parallel(nranks, rank, fn, user));
}

// ---- unit-level seam ------------------------------------------------------
int mitdvp_heff_apply(int device, const double* L, const double* W, const double* R, const double* psi, int dl,
                      int d, int dr, int ml, int mr, double* sigma_out, int reps, double* ms_out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    e.set_mpo_core(0, 0, W, ml, d, d, mr);
    const long ns = (long)dl * d * dr, mm = std::max(ml, mr);
    e.ensure_work(ns, (long)dl * dr * d * mm, (long)dl * dr * d * mm, 1, 1);
    Dev dL(st, L, (size_t)dl * ml * dl), dR(st, R, (size_t)dr * mr * dr), dpsi(st, psi, ns), dout(ns);
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    const MpoSite& w = e.mpo(0, 0);
    e.heff_apply(dL.p(), w, dR.p(), dpsi.p(), dout.p(), dl, d, dr, hzc(0, 0));
    HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) e.heff_apply(dL.p(), w, dR.p(), dpsi.p(), dout.p(), dl, d, dr, hzc(0, 0));
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms_out) *ms_out = reps > 0 ? ms / reps : 0.0;
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
    to_host(st, sigma_out, dout.p(), ns);
  });
}

int mitdvp_keff_apply(int device, const double* L, const double* R, const double* sval, int dl, int dr, int m,
                      double* out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    e.ensure_work((long)dl * dr, (long)dl * m * dr, 1, 1, 1);
    Dev dL(st, L, (size_t)dl * m * dl), dR(st, R, (size_t)dr * m * dr), ds(st, sval, (size_t)dl * dr),
        dout((size_t)dl * dr);
    e.keff_apply(dL.p(), dR.p(), ds.p(), dout.p(), dl, dr, m, hzc(0, 0));
    to_host(st, out, dout.p(), (size_t)dl * dr);
  });
}

int mitdvp_env_update(int device, int left, const double* env, const double* site, const double* W, int dl, int d,
                      int dr, int ml, int mr, double* out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    e.set_mpo_core(0, 0, W, ml, d, d, mr);
    const long ns = (long)dl * d * dr, mm = std::max(ml, mr);
    e.ensure_work(ns, (long)dl * dr * d * mm, (long)dl * dr * d * mm, 1, 1);
    const MpoSite& w = e.mpo(0, 0);
    Dev dsite(st, site, ns);
    if (left) {
      // contract_with_site_mpo, gauge "A": env (dl, ml, dl) -> (dr, mr, dr)
      Dev denv(st, env, (size_t)dl * ml * dl), dout((size_t)dr * mr * dr);
      e.env_update(denv.p(), dsite.p(), w.w2l.p, dout.p(), dl, ml, d, dr, mr);
      to_host(st, out, dout.p(), (size_t)dr * mr * dr);
    } else {
      // gauge "B": env (dr, mr, dr) -> (dl, ml, dl), through the mirrored tensor
      Dev denv(st, env, (size_t)dr * mr * dr), dout((size_t)dl * ml * dl), dbt(ns);
      transpose_rev3(st, dsite.p(), dbt.p(), dl, d, dr);
      e.env_update(denv.p(), dbt.p(), w.w2r.p, dout.p(), dr, mr, d, dl, ml);
      to_host(st, out, dout.p(), (size_t)dl * ml * dl);
    }
  });
}

int mitdvp_gauge_trf(int device, int key, const double* psi, int dl, int d, int dr, double* site_out,
                     double* sigma_out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    const long ns = (long)dl * d * dr;
    e.ensure_work(ns, 1, 1, std::max(dl, dr) * d, std::max(dl, dr));
    e.set_qr_gauge_free(false);  // SiteCoef.gauge_trf as LAPACK returns it (signs of diag(R) included)
    Dev dpsi(st, psi, ns), dsite(ns), dbt(ns);
    if (key == 0) {
      Dev dsig((size_t)dr * dr);
      e.gauge_qr_left(dpsi.p(), dl, d, dr, dsite.p(), dsig.p());
      to_host(st, site_out, dsite.p(), ns);
      to_host(st, sigma_out, dsig.p(), (size_t)dr * dr);
    } else {
      Dev dsig((size_t)dl * dl);
      e.gauge_qr_right(dpsi.p(), dl, d, dr, dsite.p(), dbt.p(), dsig.p());
      to_host(st, site_out, dsite.p(), ns);
      to_host(st, sigma_out, dsig.p(), (size_t)dl * dl);
    }
  });
}

int mitdvp_expm_dense(int device, int integrator, int conserve_norm, int lanczos_variant, const double* mat, int n,
                      const double* x, double scale_re, double scale_im, double thresh, int k_prev, double* y_out,
                      int* k_out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    mitdvp_config c = unit_cfg(device);
    c.integrator = integrator;
    c.conserve_norm = conserve_norm;
    c.lanczos_variant = lanczos_variant;
    c.thresh = thresh;
    Engine e(c);
    hipStream_t st = e.stream();
    e.ensure_work(n, 1, 1, 1, 1);
    Dev dm(st, mat, (size_t)n * n), dx(st, x, n);
    auto mv = [&](const zc* in, zc* out) {
      ZgemmDesc g = zgemm_desc(dm.p(), in, out, n, 1, n);
      zgemm(st, g);
    };
    const int k = e.krylov_exp(hzc(scale_re, scale_im), mv, dx.p(), n, k_prev);
    if (k_out) *k_out = k;
    to_host(st, y_out, dx.p(), n);
  });
}

// ---- kernel-level hooks ---------------------------------------------------
int mitdvp_zgemm(int device, int transA, int conjA, int transB, int conjB, int m, int n, int k, const double* A,
                 const double* B, double* C, const double alpha[2], const double beta[2], int tile_cfg, int reps,
                 double* ms_out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    HIP_CHECK(hipSetDevice(device));
    hipStream_t st;
    HIP_CHECK(hipStreamCreate(&st));
    {
      Dev dA(st, A, (size_t)m * k), dB(st, B, (size_t)k * n), dC(st, C, (size_t)m * n), dC0(st, C, (size_t)m * n);
      ZgemmDesc g = zgemm_desc(dA.p(), dB.p(), dC.p(), m, n, k);
      g.transA = transA; g.conjA = conjA; g.transB = transB; g.conjB = conjB;
      g.lda = transA ? m : k;
      g.ldb = transB ? k : n;
      g.alpha = make_double2(alpha[0], alpha[1]);
      g.beta = make_double2(beta[0], beta[1]);
      g.tile_cfg = tile_cfg;
      zgemm(st, g);
      to_host(st, C, dC.p(), (size_t)m * n);
      if (reps > 0) {
        hipEvent_t a, b;
        HIP_CHECK(hipEventCreate(&a));
        HIP_CHECK(hipEventCreate(&b));
        g.C = dC0.p();
        g.beta = make_double2(0.0, 0.0);
        zgemm(st, g);
        HIP_CHECK(hipEventRecord(a, st));
        for (int i = 0; i < reps; ++i) zgemm(st, g);
        HIP_CHECK(hipEventRecord(b, st));
        HIP_CHECK(hipEventSynchronize(b));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms_out) *ms_out = ms / reps;
        HIP_CHECK(hipEventDestroy(a));
        HIP_CHECK(hipEventDestroy(b));
      }
    }
    HIP_CHECK(hipStreamDestroy(st));
  });
}

int mitdvp_bench_heff(int device, int dl, int d, int dr, int ml, int mr, int reps, int warmup, double* ms_out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    const long ns = (long)dl * d * dr, mm = std::max(ml, mr);
    e.ensure_work(ns, (long)dl * dr * d * mm, (long)dl * dr * d * mm, 1, 1);
    std::vector<double> w((size_t)ml * d * d * mr * 2);
    uint64_t s = 88172645463325252ull;
    for (auto& v : w) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5) * 0.1; }
    e.set_mpo_core(0, 0, w.data(), ml, d, d, mr);
    const MpoSite& ws = e.mpo(0, 0);
    Dev dL((size_t)dl * ml * dl), dR((size_t)dr * mr * dr), dpsi(ns), dout(ns);
    vec_randn(st, dL.p(), (long)dl * ml * dl, 11);
    vec_randn(st, dR.p(), (long)dr * mr * dr, 12);
    vec_randn(st, dpsi.p(), ns, 13);
    for (int i = 0; i < warmup; ++i) e.heff_apply(dL.p(), ws, dR.p(), dpsi.p(), dout.p(), dl, d, dr, hzc(0, 0));
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) e.heff_apply(dL.p(), ws, dR.p(), dpsi.p(), dout.p(), dl, d, dr, hzc(0, 0));
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / std::max(reps, 1);
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
  });
}

int mitdvp_qr_thin(int device, const double* a, int m, int n, int gauge_free, double* q_out, double* r_out, int reps,
                   double* ms_out, long* launches, int* path_out) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    if (m < n || n < 1 || reps < 1) throw ArgError("qr_thin: need m >= n >= 1, reps >= 1");
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    Dev A((size_t)m * n), A0(st, a, (size_t)m * n), Q((size_t)m * n), R((size_t)n * n), work(qr_work_elems(m, n));
    if (!a) vec_randn(st, A0.p(), (long)m * n, 21);
    QrHistory* hist = qr_history_new();
    long nl = 0;
    bool used = false;
    auto once = [&] {
      HIP_CHECK(hipMemcpyAsync(A.p(), A0.p(), (size_t)m * n * sizeof(zc), hipMemcpyDeviceToDevice, st));
      qr_thin(st, A.p(), m, n, Q.p(), R.p(), work.p(), &nl, nullptr, hist, gauge_free != 0, &used);
    };
    try {
      once();
      if (path_out) *path_out = used ? 1 : 0;
      nl = 0;
      hipEvent_t ea, eb;
      HIP_CHECK(hipEventCreate(&ea));
      HIP_CHECK(hipEventCreate(&eb));
      HIP_CHECK(hipEventRecord(ea, st));
      for (int i = 0; i < reps; ++i) once();
      HIP_CHECK(hipEventRecord(eb, st));
      HIP_CHECK(hipEventSynchronize(eb));
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, ea, eb));
      if (ms_out) *ms_out = ms / reps;
      if (launches) *launches = nl / reps;
      HIP_CHECK(hipEventDestroy(ea));
      HIP_CHECK(hipEventDestroy(eb));
      if (q_out) to_host(st, q_out, Q.p(), (size_t)m * n);
      if (r_out) to_host(st, r_out, R.p(), (size_t)n * n);
    } catch (...) {
      qr_history_free(hist);
      throw;
    }
    qr_history_free(hist);
  });
}

// device time of one (m x n) QR (Q and R formed), averaged over reps; launches per factorisation in *launches
int mitdvp_bench_qr(int device, int m, int n, int reps, double* ms_out, long* launches) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    NEED(ms_out);
    if (m < n || n < 1 || reps < 1) throw ArgError("bench_qr: need m >= n >= 1, reps >= 1");
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    Dev A((size_t)m * n), A0((size_t)m * n), Q((size_t)m * n), R((size_t)n * n), work(qr_work_elems(m, n));
    vec_randn(st, A0.p(), (long)m * n, 21);
    long nl = 0;
    auto once = [&] {
      HIP_CHECK(hipMemcpyAsync(A.p(), A0.p(), (size_t)m * n * sizeof(zc), hipMemcpyDeviceToDevice, st));
      qr_householder(st, A.p(), m, n, Q.p(), R.p(), work.p(), &nl, 0, nullptr);
    };
    once();
    nl = 0;
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) once();
    HIP_CHECK(hipEventRecord(b, st));
    HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / reps;
    if (launches) *launches = nl / reps;
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
  });
}

int mitdvp_heff_selfcheck(int device, int dl, int d, int dr, int ml, int mr, double out[4]) {
  return guard(nullptr, [&] {
    using namespace mitdvp;
    Engine e(unit_cfg(device));
    hipStream_t st = e.stream();
    const long ns = (long)dl * d * dr, mm = std::max(ml, mr);
    e.ensure_work(ns, (long)dl * dr * d * mm, (long)dl * dr * d * mm, 1, 1);
    std::vector<double> w((size_t)ml * d * d * mr * 2);
    uint64_t s = 0x2545F4914F6CDD1Dull;
    for (auto& v : w) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = ((double)(s >> 11) / 9007199254740992.0 - 0.5); }
    e.set_mpo_core(0, 0, w.data(), ml, d, d, mr);
    const MpoSite& ws = e.mpo(0, 0);
    Dev dL((size_t)dl * ml * dl), dR((size_t)dr * mr * dr), x(ns), y(ns), z(ns), hx(ns), hy(ns), hz(ns), h4(ns);
    vec_randn(st, dL.p(), (long)dl * ml * dl, 21);
    vec_randn(st, dR.p(), (long)dr * mr * dr, 22);
    vec_randn(st, x.p(), ns, 23);
    vec_randn(st, y.p(), ns, 24);
    const zc one = make_double2(1.0, 0.0), mone = make_double2(-1.0, 0.0), twoi = make_double2(0.0, 2.0),
             mtwoi = make_double2(0.0, -2.0);
    HIP_CHECK(hipMemcpyAsync(z.p(), x.p(), ns * sizeof(zc), hipMemcpyDeviceToDevice, st));
    vec_axpby(st, z.p(), y.p(), ns, twoi, one);  // z = x + 2i y
    const int saved = zgemm_default_mode();
    Dev red(NPART);
    double* rp = reinterpret_cast<double*>(red.p());
    auto nrm = [&](zc* v) {
      std::vector<double> h(NPART);
      vec_sumsq(st, v, ns, rp);
      HIP_CHECK(hipMemcpyAsync(h.data(), rp, NPART * sizeof(double), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      double t = 0;
      for (double q : h) t += q;
      return std::sqrt(t);
    };
    zgemm_set_default_mode(0);
    e.heff_apply(dL.p(), ws, dR.p(), x.p(), h4.p(), dl, d, dr, hzc(0, 0));
    zgemm_set_default_mode(1);
    e.heff_apply(dL.p(), ws, dR.p(), x.p(), hx.p(), dl, d, dr, hzc(0, 0));
    zgemm_set_default_mode(saved);
    const double n4 = nrm(h4.p());
    vec_axpby(st, h4.p(), hx.p(), ns, mone, one);  // h4 - hx
    out[0] = nrm(h4.p()) / n4;
    e.heff_apply(dL.p(), ws, dR.p(), y.p(), hy.p(), dl, d, dr, hzc(0, 0));
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, st));
    e.heff_apply(dL.p(), ws, dR.p(), z.p(), hz.p(), dl, d, dr, hzc(0, 0));
    HIP_CHECK(hipEventRecord(b, st));
    const double nz = nrm(hz.p());
    vec_axpby(st, hz.p(), hx.p(), ns, mone, one);
    vec_axpby(st, hz.p(), hy.p(), ns, mtwoi, one);
    out[1] = nrm(hz.p()) / nz;
    out[2] = nrm(hx.p());
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    out[3] = ms;
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
  });
}

int mitdvp_set_gemm_mode(int mode3m) {
  mitdvp::zgemm_set_default_mode(mode3m);
  return MITDVP_OK;
}
int mitdvp_get_gemm_mode(void) { return mitdvp::zgemm_default_mode(); }

int mitdvp_mfma_peak_probe(int device, double* tflops_out) {
  return guard(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    *tflops_out = mitdvp::mfma_peak_probe(nullptr);
  });
}
int mitdvp_clock_probe(int device, long iters, double* cycles_ticks_out) {
  return guard(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    double o[3];
    mitdvp::clock_probe(nullptr, iters, o);
    cycles_ticks_out[0] = o[0];
    cycles_ticks_out[1] = o[1];
  });
}
int mitdvp_cu_mask_probe(int device, const unsigned* mask, int nwords, int nblocks, size_t lds_bytes, int spin_us, int* out) {
  return guard(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    mitdvp::where_probe(mask, nwords, nblocks, lds_bytes, spin_us, out);
  });
}
int mitdvp_mfma_layout_probe(int device, int* out) {
  return guard(nullptr, [&] {
    HIP_CHECK(hipSetDevice(device));
    mitdvp::mfma_layout_probe(nullptr, out);
  });
}

}  // extern "C"
