// engine.hip -- host orchestration of the device-resident one-site TDVP sweep.
//
// Reference path being replaced (paths relative to /root/reference/pytdscf):
//   MPSCoef.propagate / propagate_along_sweep      _mps_cls.py:452-503, :798-1014
//   exp_superH/K_propagation_direct                _mps_cls.py:1016-1170
//   trans_next_psite_AsigmaB / APsiB               _mps_cls.py:1798-1850, :1172-1206
//   renormalize_op_psite / contract_with_site_mpo  _mps_mpo.py:421-696, _contraction.py:148-397
//   multiplyH/K_MPS_direct_MPO.dot                 _contraction.py:1182-1243, :1358-1407
//   short_iterative_lanczos / _arnoldi             _integrator.py:453-655, :287-432
//   SiteCoef.gauge_trf                             _site_cls.py:138-292
//
// All tensors live in HBM for the whole run; one HIP stream; the host only sees
// the O(k) Krylov scalars (k <= 20) at the points where the reference evaluates
// its convergence test.
#include "engine.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "small_linalg.h"

namespace mitdvp {

static const double KRYLOV_EPS = 1e-12;  // _integrator.py:22

// layout of the reduction scratch (units: zc)
static constexpr size_t RED_ALPHA = 0;                                   // [MAXK][NPART] zc
static constexpr size_t RED_NRM = RED_ALPHA + (size_t)MAXK * NPART;      // [MAXK][NPART] double
static constexpr size_t RED_H = RED_NRM + (size_t)MAXK * NPART / 2 + 1;  // [MAXK][MAXK][NPART] zc
static constexpr size_t RED_MISC = RED_H + (size_t)MAXK * MAXK * NPART;  // [4][NPART] zc
static constexpr size_t RED_TOTAL = RED_MISC + 4 * (size_t)NPART;

Engine::Engine(const mitdvp_config& c) : cfg(c), L_(c.nsite) {
  if (c.nsite < 1) throw ArgError("nsite must be >= 1");
  if (c.max_krylov < 1 || c.max_krylov > MAXK - 1) throw ArgError("max_krylov must be in [1, 20]");
  if (c.integrator != MITDVP_LANCZOS && c.integrator != MITDVP_ARNOLDI) throw ArgError("bad integrator");
  if (c.relax < 0 || c.relax > 2) throw ArgError("relax must be 0, 1 or 2");
  max_diag_krylov_ = c.max_diag_krylov > 0 ? c.max_diag_krylov : 64;
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  if (ndev < 1) throw HipError("no HIP device visible: the MI355X engine has no CPU fallback");
  if (c.device < 0 || c.device >= ndev) throw ArgError("bad device ordinal");
  HIP_CHECK(hipSetDevice(c.device));
  HIP_CHECK(hipStreamCreate(&st_));
  dl_.assign(L_, 0); dd_.assign(L_, 0); dr_.assign(L_, 0); gauge_.assign(L_, -1);
  site_.resize(L_);
  envL_.resize(L_ + 1); envR_.resize(L_ + 1);
  envL_ok_.assign(L_ + 1, 0); envR_ok_.assign(L_ + 1, 0);
  kprev_.assign(L_, 0);
  red_.reserve(RED_TOTAL);
  red_elems_ = RED_TOTAL;
  HIP_CHECK(hipHostMalloc((void**)&h_red_, RED_TOTAL * sizeof(zc)));
  // trivial boundary blocks, construct_op_zerosite (_mps_mpo.py:364-419)
  const zc one = make_double2(1.0, 0.0);
  envL_[0].reserve(1); envR_[L_].reserve(1);
  HIP_CHECK(hipMemcpyAsync(envL_[0].p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipMemcpyAsync(envR_[L_].p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  envL_ok_[0] = 1; envR_ok_[L_] = 1;
}

Engine::~Engine() {
  if (st_) (void)hipStreamSynchronize(st_);
  for (auto& t : pending_) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto& e : evpool_) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h_red_) (void)hipHostFree(h_red_);
  if (st_) (void)hipStreamDestroy(st_);
}

// ---------------------------------------------------------------------------
DevBuf Engine::pool_get(size_t elems) {
  for (size_t i = 0; i < pool_.size(); ++i)
    if (pool_[i].n >= elems && pool_[i].n <= elems + elems / 2 + 64) {
      DevBuf b = std::move(pool_[i]);
      pool_.erase(pool_.begin() + i);
      return b;
    }
  DevBuf b;
  b.reserve(elems);
  return b;
}
void Engine::pool_put(DevBuf&& b) {
  if (b.p) pool_.push_back(std::move(b));
}

void Engine::timer_begin(int kind) {
  if (!profiling_) return;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!evpool_.empty()) { ev = evpool_.back(); evpool_.pop_back(); }
  else { HIP_CHECK(hipEventCreate(&ev.first)); HIP_CHECK(hipEventCreate(&ev.second)); }
  HIP_CHECK(hipEventRecord(ev.first, st_));
  pending_.push_back(PhaseTimer{ev.first, ev.second, kind});
  cur_timer_ = (int)pending_.size() - 1;
}
void Engine::timer_end() {
  if (!profiling_ || cur_timer_ < 0) return;
  HIP_CHECK(hipEventRecord(pending_[cur_timer_].b, st_));
  cur_timer_ = -1;
  if (pending_.size() > 8192) resolve_timers();
}
void Engine::resolve_timers() {
  if (pending_.empty()) return;
  HIP_CHECK(hipStreamSynchronize(st_));
  for (auto& t : pending_) {
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, t.a, t.b));
    switch (t.kind) {
      case 0: cnt_.heff_ms += ms; break;
      case 1: cnt_.env_ms += ms; break;
      case 2: cnt_.keff_ms += ms; break;
      case 3: cnt_.qr_ms += ms; break;
      case 10: case 11: case 12:
        cnt_.heff_stage_ms[t.kind - 10] += ms;
        cnt_.heff_ms += ms;
        break;
      default: cnt_.krylov_vec_ms += ms; break;
    }
    evpool_.emplace_back(t.a, t.b);
  }
  pending_.clear();
}
void Engine::counters_get(mitdvp_counters* out) {
  resolve_timers();
  *out = cnt_;
}
void Engine::counters_reset() {
  resolve_timers();
  std::memset(&cnt_, 0, sizeof(cnt_));
}

void Engine::read_partials(size_t off, size_t count) {
  HIP_CHECK(hipMemcpyAsync(h_red_ + off, red_.p + off, count * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

// ---------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------
void Engine::set_site(int i, const double* reim, int l, int n, int r, int gauge) {
  if (i < 0 || i >= L_) throw ArgError("set_site: bad site index");
  if (l < 1 || n < 1 || r < 1) throw ArgError("set_site: bad shape");
  const size_t e = (size_t)l * n * r;
  site_[i].reserve(e);
  HIP_CHECK(hipMemcpyAsync(site_[i].p, reim, e * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  dl_[i] = l; dd_[i] = n; dr_[i] = r; gauge_[i] = gauge;
  if (gauge == MITDVP_GAUGE_PSI) center_ = i;
  invalidate_env();
}
void Engine::get_site_shape(int i, int* l, int* n, int* r, int* gauge) const {
  if (i < 0 || i >= L_) throw ArgError("get_site_shape: bad site index");
  *l = dl_[i]; *n = dd_[i]; *r = dr_[i]; *gauge = gauge_[i];
}
void Engine::get_site(int i, double* out) {
  if (i < 0 || i >= L_ || !site_[i].p) throw ArgError("get_site: bad or unset site");
  const size_t e = (size_t)dl_[i] * dd_[i] * dr_[i];
  HIP_CHECK(hipMemcpyAsync(out, site_[i].p, e * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

Operator& Engine::op(int id) {
  auto it = ops_.find(id);
  if (it == ops_.end()) {
    Operator o;
    o.sites.resize(L_);
    it = ops_.emplace(id, std::move(o)).first;
  }
  return it->second;
}
const MpoSite& Engine::mpo(int op_id, int isite) {
  auto it = ops_.find(op_id);
  if (it == ops_.end() || !it->second.sites[isite].set) throw ArgError("operator core not set for this site");
  return it->second.sites[isite];
}

void Engine::set_mpo_core(int op_id, int isite, const double* reim, int ml, int dout, int din, int mr) {
  if (isite < 0 || isite >= L_) throw ArgError("set_mpo_core: bad site index");
  if (ml < 1 || mr < 1 || dout < 1 || dout != din) throw ArgError("set_mpo_core: need a square 4-leg core");
  const int d = dout;
  const hzc* W = reinterpret_cast<const hzc*>(reim);
  std::vector<hzc> w2l((size_t)d * mr * ml * d), w2r((size_t)d * ml * mr * d);
  for (int c = 0; c < ml; ++c)
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j)
        for (int t = 0; t < mr; ++t) {
          const hzc v = W[(((size_t)c * d + i) * d + j) * mr + t];
          w2l[((size_t)i * mr + t) * ((size_t)ml * d) + (size_t)c * d + j] = v;
          w2r[((size_t)i * ml + c) * ((size_t)mr * d) + (size_t)t * d + j] = v;
        }
  MpoSite& s = op(op_id).sites[isite];
  s.ml = ml; s.d = d; s.mr = mr;
  s.w2l.reserve(w2l.size());
  s.w2r.reserve(w2r.size());
  HIP_CHECK(hipMemcpyAsync(s.w2l.p, w2l.data(), w2l.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipMemcpyAsync(s.w2r.p, w2r.data(), w2r.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  s.set = true;
  if (op_id == 0) invalidate_env();
}
void Engine::set_shift(int op_id, double re, double im) { op(op_id).shift = hzc(re, im); }

void Engine::invalidate_env() {
  for (int b = 1; b < L_; ++b) {
    envL_ok_[b] = 0; envR_ok_[b] = 0;
    pool_put(std::move(envL_[b]));
    pool_put(std::move(envR_[b]));
  }
}

void Engine::ensure_work(long max_site, long max_x, long max_y, int max_qr_m, int max_qr_n) {
  X_.reserve(max_x);
  Y_.reserve(max_y);
  V_.reserve((size_t)MAXK * max_site);
  tmp1_.reserve(max_site);
  tmp2_.reserve(max_site);
  const size_t dd = (size_t)max_qr_n * max_qr_n;
  sig_.reserve(std::max<size_t>(dd, 1));
  sig2_.reserve(std::max<size_t>(dd, 1));
  qrwork_.reserve(qr_work_elems(max_qr_m, max_qr_n));
}

void Engine::size_workspaces() {
  long ms = 1, mx = 1, my = 1;
  int qm = 1, qn = 1;
  for (int p = 0; p < L_; ++p) {
    const long s = (long)dl_[p] * dd_[p] * dr_[p];
    ms = std::max(ms, s);
    qm = std::max(qm, std::max(dl_[p], dr_[p]) * dd_[p]);
    qn = std::max(qn, std::max(dl_[p], dr_[p]));
    for (auto& kv : ops_) {
      const MpoSite& w = kv.second.sites[p];
      if (!w.set) continue;
      const long mm = std::max(w.ml, w.mr);
      mx = std::max(mx, (long)dl_[p] * dr_[p] * dd_[p] * mm);
      my = std::max(my, (long)dl_[p] * dr_[p] * dd_[p] * mm);
    }
  }
  ensure_work(ms, mx, my, qm, qn);
  // site buffers are exchanged with a spare of capacity max_site during the
  // sweep (QR / absorb write into the spare, then swap): give all of them that
  // capacity so that any of them can play the spare's role afterwards.
  for (int p = 0; p < L_; ++p)
    if (site_[p].p) site_[p].grow_preserve((size_t)ms, (size_t)dl_[p] * dd_[p] * dr_[p], st_);
}

void Engine::require_ready() {
  for (int p = 0; p < L_; ++p) {
    if (!site_[p].p) throw ArgError("site tensor not set");
    if (p + 1 < L_ && dr_[p] != dl_[p + 1]) throw ArgError("bond dimension mismatch between neighbouring sites");
  }
  if (dl_[0] != 1 || dr_[L_ - 1] != 1) throw ArgError("open boundary bonds must be 1");
  size_workspaces();
}

// ---------------------------------------------------------------------------
// contractions
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// bond-sharded execution over several GPUs (one process per GPU)
//
// The three contractions of an apply / environment update are independent for
// every value of the leading (bra-side) bond index of the environment block, so
// rank r computes the rows a in [r*n/N, (r+1)*n/N) from replicated operands and
// the ranks exchange results with ONE collective per contraction chain:
//   H_eff / K_eff apply : in-place all-gather of the result vector
//   environment update  : in-place all-reduce (sum over the sharded bra index)
// Everything else (Krylov vector algebra, QR, absorption) is computed
// redundantly on identical data, so all ranks take identical control-flow
// decisions without exchanging scalars.  The collective itself is a callback
// (RCCL through torch.distributed in production, gloo in the 1-GPU tests).
// ---------------------------------------------------------------------------
void Engine::set_parallel(int nranks, int rank, CollFn fn, void* user) {
  if (nranks < 1 || rank < 0 || rank >= nranks) throw ArgError("set_parallel: bad rank / nranks");
  if (nranks > 1 && !fn) throw ArgError("set_parallel: a collective callback is required for nranks > 1");
  nranks_ = nranks; rank_ = rank; coll_ = fn; coll_user_ = user;
}

bool Engine::shard_range(int n, int& a0, int& a1) const {
  a0 = 0; a1 = n;
  if (nranks_ <= 1 || n % nranks_ != 0 || n < 8 * nranks_) return false;  // small / ragged bonds stay replicated
  const int c = n / nranks_;
  a0 = rank_ * c; a1 = a0 + c;
  return true;
}

void Engine::collective(int op, zc* p, size_t elems) {
  HIP_CHECK(hipStreamSynchronize(st_));
  const int rc = coll_(coll_user_, op, p, elems * sizeof(zc));
  if (rc != 0) throw HipError("collective callback failed (rc=" + std::to_string(rc) + ")");
  cnt_.n_collectives += 1;
  cnt_.collective_bytes += (double)(elems * sizeof(zc));
}

void Engine::heff_apply(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dl, int d, int dr,
                        hzc shift) {
  const int ml = w.ml, mr = w.mr;
  int a0, a1;
  const bool sharded = shard_range(dl, a0, a1);
  const int na = a1 - a0;
  timer_begin(10);
  {  // X[(a,c)][(j,s)] = L[(a,c)][b] psi[b][(j,s)]
    ZgemmDesc g = zgemm_desc(L + (size_t)a0 * ml * dl, psi, X_.p, na * ml, d * dr, dl);
    zgemm(st_, g);
  }
  timer_end();
  timer_begin(11);
  {  // Y_a[(i,t)][s] = W2L[(i,t)][(c,j)] X_a[(c,j)][s]
    ZgemmDesc g = zgemm_desc(w.w2l.p, X_.p, Y_.p, d * mr, dr, ml * d);
    g.batch = na; g.strideA = 0; g.strideB = (long)ml * d * dr; g.strideC = (long)d * mr * dr;
    zgemm(st_, g);
  }
  timer_end();
  timer_begin(12);
  {  // out[(a,i)][r] = Y[(a,i)][(t,s)] R[r][(t,s)]
    ZgemmDesc g = zgemm_desc(Y_.p, R, out + (size_t)a0 * d * dr, na * d, dr, mr * dr);
    g.transB = 1; g.ldb = (long)mr * dr;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLGATHER, out, (size_t)dl * d * dr);
  if (shift != hzc(0.0, 0.0))
    vec_axpby(st_, out, psi, (long)dl * d * dr, make_double2(shift.real(), shift.imag()), make_double2(1.0, 0.0));
  cnt_.n_launch += 3;
  cnt_.n_heff += 1;
  cnt_.heff_flops += 8.0 * ((double)na * dl * ml * d * dr + (double)na * dr * ml * mr * d * d + (double)na * dr * dr * mr * d);
}

void Engine::keff_apply(const zc* L, const zc* R, const zc* sig, zc* out, int d1, int d2, int m, hzc shift) {
  int a0, a1;
  const bool sharded = shard_range(d1, a0, a1);
  const int na = a1 - a0;
  timer_begin(2);
  {  // X[(a,c)][s] = L[(a,c)][b] sig[b][s]
    ZgemmDesc g = zgemm_desc(L + (size_t)a0 * m * d1, sig, X_.p, na * m, d2, d1);
    zgemm(st_, g);
  }
  {  // out[a][r] = X[a][(c,s)] R[r][(c,s)]
    ZgemmDesc g = zgemm_desc(X_.p, R, out + (size_t)a0 * d2, na, d2, m * d2);
    g.transB = 1; g.ldb = (long)m * d2;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLGATHER, out, (size_t)d1 * d2);
  if (shift != hzc(0.0, 0.0))
    vec_axpby(st_, out, sig, (long)d1 * d2, make_double2(shift.real(), shift.imag()), make_double2(1.0, 0.0));
  cnt_.n_launch += 2;
  cnt_.n_keff += 1;
  cnt_.keff_flops += 8.0 * ((double)na * d1 * m * d2 + (double)na * d2 * d2 * m);
}

void Engine::env_update(const zc* env_in, const zc* T, const zc* w2, zc* env_out, int din, int min_, int d, int dout,
                        int mout) {
  int m0, m1;
  const bool sharded = shard_range(din, m0, m1);
  const int nm = m1 - m0;
  timer_begin(1);
  {  // X[(m,p)][(s,j)] = env[(m,p)][n] T[n][(s,j)]
    ZgemmDesc g = zgemm_desc(env_in + (size_t)m0 * min_ * din, T, X_.p, nm * min_, d * dout, din);
    zgemm(st_, g);
  }
  {  // Y_m[(r,q)][j] = W2[(r,q)][(p,s)] X_m[(p,s)][j]
    ZgemmDesc g = zgemm_desc(w2, X_.p, Y_.p, d * mout, dout, min_ * d);
    g.batch = nm; g.strideA = 0; g.strideB = (long)min_ * d * dout; g.strideC = (long)d * mout * dout;
    zgemm(st_, g);
  }
  {  // env'[i][(q,j)] = conj(T)[(m,r)][i] Y[(m,r)][(q,j)]   (sum over this rank's m)
    ZgemmDesc g = zgemm_desc(T + (size_t)m0 * d * dout, Y_.p, env_out, dout, mout * dout, nm * d);
    g.transA = 1; g.conjA = 1; g.lda = dout;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLREDUCE, env_out, (size_t)dout * mout * dout);
  cnt_.n_launch += 3;
  cnt_.n_env += 1;
  cnt_.env_flops += 8.0 * ((double)nm * din * min_ * d * dout + (double)nm * dout * min_ * mout * d * d +
                           (double)nm * dout * dout * mout * d);
}

// ---------------------------------------------------------------------------
// local propagator: x <- exp(scale*Op) x
// ---------------------------------------------------------------------------
template <class MV>
int Engine::krylov_exp(hzc scale, MV&& matvec, zc* x, long n, int k_prev) {
  const int ndim = (int)std::min<long>(n, cfg.max_krylov);
  // _iter_info (_integrator.py:178-186)
  const int n_warm = (int)std::min<long>(n, std::min(std::max(0, k_prev - 2), 15));
  const bool lanczos = cfg.integrator == MITDVP_LANCZOS;
  const bool cn = cfg.conserve_norm != 0;
  zc* V = V_.p;
  const long ldv = n;
  zc* alpha_p = red_.p + RED_ALPHA;
  double* nrm_p = reinterpret_cast<double*>(red_.p + RED_NRM);
  zc* h_p = red_.p + RED_H;
  double* misc_d = reinterpret_cast<double*>(red_.p + RED_MISC);

  // _normalize (_integrator.py:189-203)
  double beta0 = 1.0;
  HIP_CHECK(hipMemcpyAsync(V, x, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  if (!cn) {
    vec_sumsq(st_, x, n, misc_d);
    read_partials(RED_MISC, NPART / 2);
    const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
    double s = 0;
    for (int i = 0; i < NPART; ++i) s += hp[i];
    beta0 = std::sqrt(s);
    if (beta0 == 0.0) throw ArgError("Initial psi has zero norm.");
    vec_scale(st_, V, n, make_double2(1.0 / beta0, 0.0));
  }

  std::vector<hzc> alpha;            // Lanczos diagonal
  std::vector<double> beta;          // norms of the new vectors
  std::vector<hzc> hess((size_t)(ndim + 1) * ndim, hzc(0, 0));  // Arnoldi Hessenberg (row-major, ld = ndim)
  std::vector<hzc> coef_prev;
  int next_unread = 0;

  auto sum_z = [&](size_t off) {
    double re = 0, im = 0;
    for (int i = 0; i < NPART; ++i) { re += h_red_[off + i].x; im += h_red_[off + i].y; }
    return hzc(re, im);
  };
  auto sum_d = [&](size_t off_zc, int row) {
    const double* p = reinterpret_cast<const double*>(h_red_ + off_zc) + (size_t)row * NPART;
    double s = 0;
    for (int i = 0; i < NPART; ++i) s += p[i];
    return s;
  };

  auto finalize = [&](const std::vector<hzc>& coef, int k) {
    Coefs c{};
    for (int j = 0; j < k; ++j) {
      const hzc v = cn ? coef[j] : coef[j] * beta0;  // _rescale, :206-213
      c.c[j] = make_double2(v.real(), v.imag());
    }
    if (cn) {
      vec_lincomb(st_, x, V, ldv, k, c, n, misc_d);
      read_partials(RED_MISC, NPART / 2);
      const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
      double s = 0;
      for (int i = 0; i < NPART; ++i) s += hp[i];
      vec_scale(st_, x, n, make_double2(1.0 / std::sqrt(s), 0.0));
    } else {
      vec_lincomb(st_, x, V, ldv, k, c, n, nullptr);
    }
    cnt_.n_launch += 2;
  };

  for (int l = 0; l < ndim; ++l) {
    zc* vl = V + (size_t)l * ldv;
    zc* vn = V + (size_t)(l + 1) * ldv;
    matvec(vl, vn);
    timer_begin(4);
    if (lanczos) {
      // alpha_l = <v0 | H v_l> (reference, :556) or <v_l | H v_l> (orthodox)
      vec_dot(st_, cfg.lanczos_variant == 0 ? V : vl, vn, n, true, alpha_p + (size_t)l * NPART);
      vec_lanczos_update(st_, vn, vl, l > 0 ? V + (size_t)(l - 1) * ldv : nullptr, n, alpha_p + (size_t)l * NPART,
                         l > 0 ? nrm_p + (size_t)(l - 1) * NPART : nullptr, nrm_p + (size_t)l * NPART);
    } else {
      vec_multi_dot(st_, V, ldv, l + 1, vn, n, h_p + (size_t)l * MAXK * NPART);
      vec_arnoldi_update(st_, vn, V, ldv, l + 1, n, h_p + (size_t)l * MAXK * NPART, nrm_p + (size_t)l * NPART);
    }
    vec_scale_inv_norm(st_, vn, n, nrm_p + (size_t)l * NPART, KRYLOV_EPS);
    timer_end();
    cnt_.n_launch += 3;

    const bool last_possible = (l + 1 == n);
    if (l < n_warm && !last_possible && l + 1 < ndim) continue;  // warm-up: no host sync (:578-579)

    // ---- bring the scalars of iterations [next_unread, l] to the host -------
    if (lanczos) {
      read_partials(RED_ALPHA + (size_t)next_unread * NPART, (size_t)(l + 1 - next_unread) * NPART);
    } else {
      read_partials(RED_H + (size_t)next_unread * MAXK * NPART, (size_t)(l + 1 - next_unread) * MAXK * NPART);
    }
    read_partials(RED_NRM, (size_t)MAXK * NPART / 2 + 1);
    int ld = l;
    bool exhausted = false;
    for (int q = next_unread; q <= l; ++q) {
      const double b = std::sqrt(sum_d(RED_NRM, q));
      if ((int)beta.size() <= q) beta.resize(q + 1);
      beta[q] = b;
      if (lanczos) {
        if ((int)alpha.size() <= q) alpha.resize(q + 1);
        alpha[q] = sum_z(RED_ALPHA + (size_t)q * NPART);
      } else {
        for (int j = 0; j <= q; ++j) hess[(size_t)j * ndim + q] = sum_z(RED_H + ((size_t)q * MAXK + j) * NPART);
        if (b > KRYLOV_EPS && q + 1 < ndim + 1) hess[(size_t)(q + 1) * ndim + q] = b;
      }
      if (b < KRYLOV_EPS || q + 1 == n) {  // Krylov space exhausted (:569, :392)
        ld = q;
        exhausted = true;
        break;
      }
    }
    next_unread = l + 1;
    if (ld < n_warm && !exhausted) continue;

    // ---- Ritz propagation in the Krylov space (:581-637, :397-409) ---------
    const int k = ld + 1;
    std::vector<hzc> coef(k);
    if (ld == 0) {
      coef[0] = std::exp(scale * (lanczos ? alpha[0] : hess[0]));
    } else if (lanczos) {
      bool real_alpha = true;
      for (int q = 0; q < k; ++q)
        if (std::fabs(alpha[q].imag()) > 1e-10) real_alpha = false;
      if (real_alpha) {
        std::vector<double> a(k), b(k);
        for (int q = 0; q < k; ++q) { a[q] = alpha[q].real(); b[q] = beta[q]; }
        coef = expm_tridiag_e0(a, b, k, scale);
      } else {
        std::vector<hzc> T((size_t)k * k, hzc(0, 0));
        for (int q = 0; q < k; ++q) {
          T[(size_t)q * k + q] = scale * alpha[q];
          if (q + 1 < k) T[(size_t)q * k + q + 1] = T[(size_t)(q + 1) * k + q] = scale * beta[q];
        }
        coef = expm_col0(T, k);
      }
    } else {
      std::vector<hzc> Hk((size_t)k * k);
      for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) Hk[(size_t)i * k + j] = scale * hess[(size_t)i * ndim + j];
      coef = expm_col0(Hk, k);
    }

    if (exhausted) {
      finalize(coef, k);
      return k;
    }
    if (!coef_prev.empty()) {
      // || psi_k - psi_{k-1} ||  (:644-652) without materialising either vector
      Coefs dc{};
      for (int j = 0; j < k; ++j) {
        const hzc dlt = coef[j] - (j < (int)coef_prev.size() ? coef_prev[j] : hzc(0, 0));
        dc.c[j] = make_double2(dlt.real(), dlt.imag());
      }
      vec_lincomb(st_, nullptr, V, ldv, k, dc, n, misc_d);
      cnt_.n_launch += 1;
      read_partials(RED_MISC, NPART / 2);
      const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
      double s = 0;
      for (int i = 0; i < NPART; ++i) s += hp[i];
      if (std::sqrt(s) < cfg.thresh) {
        finalize(coef, k);
        return k;
      }
    }
    coef_prev = coef;
  }
  throw NotConverged(std::string(lanczos ? "Short Iterative Lanczos" : "Short Iterative Arnoldi") +
                     " is not converged in " + std::to_string(ndim) + " basis. Try shorter time interval.");
}

// ---------------------------------------------------------------------------
// improved relaxation: lowest eigenvector of H_eff by Lanczos
// (matrix_diagonalize_lanczos, _integrator.py:74-138): orthodox Lanczos
// (alpha_l = Re <v_l|H|v_l>), the projected tridiagonal problem is solved on the
// host after every new vector, convergence on the change of the Ritz vector.
// The Ritz vector's sign is fixed by a non-negative overlap with the start
// vector (LAPACK leaves it arbitrary; only the global phase of the state is
// affected).  The change ||psi_k - psi_{k-1}|| is evaluated in the Lanczos
// coefficient space (the basis is orthonormal to working accuracy).
// ---------------------------------------------------------------------------
template <class MV>
int Engine::krylov_diag(MV&& matvec, zc* x, long n) {
  const int kcap = (int)std::min<long>(n, max_diag_krylov_);
  Vdiag_.reserve((size_t)(kcap + 1) * n);
  zc* V = Vdiag_.p;
  zc* alpha_p = red_.p + RED_ALPHA;
  double* nrm_p = reinterpret_cast<double*>(red_.p + RED_NRM);
  HIP_CHECK(hipMemcpyAsync(V, x, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  std::vector<double> alpha, beta;  // beta[i] = norm of vector i+1 before normalisation
  std::vector<double> prev;
  auto finalize = [&](const std::vector<double>& c) {
    const int k = (int)c.size();
    double* misc_d = reinterpret_cast<double*>(red_.p + RED_MISC);
    for (int c0 = 0; c0 < k; c0 += MAXK) {  // sum_j c_j V_j in chunks of MAXK vectors
      const int kc = std::min(MAXK, k - c0);
      Coefs cf{};
      for (int j = 0; j < kc; ++j) cf.c[j] = make_double2(c[c0 + j], 0.0);
      if (c0 == 0) {
        vec_lincomb(st_, x, V, n, kc, cf, n, nullptr);
      } else {
        vec_lincomb(st_, tmp1_.p, V + (size_t)c0 * n, n, kc, cf, n, nullptr);
        vec_axpby(st_, x, tmp1_.p, n, make_double2(1.0, 0.0), make_double2(1.0, 0.0));
      }
    }
    vec_sumsq(st_, x, n, misc_d);  // renormalise (get_C_sval_states_norm, _mps_cls.py:1083-1084)
    vec_scale_inv_norm(st_, x, n, misc_d, 0.0);
  };
  for (int i = 0; i < kcap; ++i) {
    zc* vi = V + (size_t)i * n;
    zc* vn = V + (size_t)(i + 1) * n;
    matvec(vi, vn);
    const int slot = i % MAXK;  // scalar slots are recycled; they are read every iteration
    vec_dot(st_, vi, vn, n, true, alpha_p + (size_t)slot * NPART);
    const int pslot = (i + MAXK - 1) % MAXK;
    vec_lanczos_update(st_, vn, vi, i > 0 ? V + (size_t)(i - 1) * n : nullptr, n, alpha_p + (size_t)slot * NPART,
                       i > 0 ? nrm_p + (size_t)pslot * NPART : nullptr, nrm_p + (size_t)slot * NPART);
    vec_scale_inv_norm(st_, vn, n, nrm_p + (size_t)slot * NPART, KRYLOV_EPS);
    cnt_.n_launch += 3;
    read_partials(RED_ALPHA + (size_t)slot * NPART, NPART);
    read_partials(RED_NRM, (size_t)MAXK * NPART / 2 + 1);
    double a = 0, b2 = 0;
    for (int q = 0; q < NPART; ++q) a += h_red_[RED_ALPHA + (size_t)slot * NPART + q].x;
    const double* np_ = reinterpret_cast<const double*>(h_red_ + RED_NRM) + (size_t)slot * NPART;
    for (int q = 0; q < NPART; ++q) b2 += np_[q];
    alpha.push_back(a);
    beta.push_back(std::sqrt(b2));
    const int k = i + 1;
    std::vector<double> c = tridiag_eigvec(alpha, beta, k, 0, nullptr);
    if (c.empty()) throw NotConverged("tridiagonal eigen-solver did not converge");
    bool done = beta.back() < KRYLOV_EPS || k == n;
    if (!done && i > 0) {
      double err = 0;
      for (int j = 0; j < k; ++j) {
        const double dlt = c[j] - (j < (int)prev.size() ? prev[j] : 0.0);
        err += dlt * dlt;
      }
      done = std::sqrt(err) < cfg.thresh;
    }
    if (done) {
      finalize(c);
      return k;
    }
    prev = c;
  }
  throw NotConverged("Lanczos Diagonalization is not converged in " + std::to_string(kcap) + " basis");
}

// ---------------------------------------------------------------------------
// gauge moves
// ---------------------------------------------------------------------------
void Engine::gauge_qr_left(const zc* psi, int dl, int d, int dr, zc* A_out, zc* sigma_out) {
  const long n = (long)dl * d * dr;
  HIP_CHECK(hipMemcpyAsync(tmp1_.p, psi, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  timer_begin(3);
  long nl = 0;
  qr_householder(st_, tmp1_.p, dl * d, dr, A_out, sigma_out, qrwork_.p, &nl);
  timer_end();
  cnt_.n_launch += nl;
  cnt_.n_qr += 1;
  const double m = (double)dl * d, nn = dr;
  cnt_.qr_flops += 4.0 * (4.0 * m * nn * nn - 4.0 * nn * nn * nn / 3.0);
}

void Engine::gauge_qr_right(const zc* psi, int dl, int d, int dr, zc* B_out, zc* Bt_out, zc* sigma_out) {
  timer_begin(3);
  long nl = 0;
  transpose_rev3(st_, psi, tmp1_.p, dl, d, dr);  // (dr, d, dl)
  qr_householder(st_, tmp1_.p, dr * d, dl, Bt_out, sig2_.p, qrwork_.p, &nl);
  transpose_batched(st_, sig2_.p, sigma_out, dl, dl, dl, dl, 1, 0, 0);  // sigma = R^T
  if (B_out) transpose_rev3(st_, Bt_out, B_out, dr, d, dl);             // (dl, d, dr)
  timer_end();
  cnt_.n_launch += nl + 3;
  cnt_.n_qr += 1;
  const double m = (double)dr * d, nn = dl;
  cnt_.qr_flops += 4.0 * (4.0 * m * nn * nn - 4.0 * nn * nn * nn / 3.0);
}

// ---------------------------------------------------------------------------
// initial state
// ---------------------------------------------------------------------------
void Engine::init_random(const int* dims, int D, uint64_t seed) {
  if (D < 1) throw ArgError("bond_dim must be >= 1");
  // LatticeInfo.get_bond_dim (_mps_cls.py:2616-2631), products saturated at D
  auto satprod = [&](int lo, int hi) {
    double p = 1;
    for (int i = lo; i < hi; ++i) { p *= dims[i]; if (p > D) return (long)D + 1; }
    return (long)p;
  };
  for (int i = 0; i < L_; ++i) {
    if (dims[i] < 1) throw ArgError("bad physical dimension");
    const long left = i == 0 ? 1 : std::min<long>(D, satprod(0, i));
    const long right = i == L_ - 1 ? 1 : std::min<long>(D, satprod(i + 1, L_));
    const long dc = dims[i];
    dl_[i] = (int)std::min({left, dc * right, (long)D});
    dr_[i] = (int)std::min({left * dc, right, (long)D});
    dd_[i] = dims[i];
    const size_t e = (size_t)dl_[i] * dd_[i] * dr_[i];
    site_[i].reserve(e);
    vec_randn(st_, site_[i].p, (long)e, seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
    gauge_[i] = -1;
  }
  invalidate_env();
  canonicalize(1.0);
}

void Engine::canonicalize(double scale) {
  require_ready();
  DevBuf spare = pool_get(V_.n / MAXK);
  double log_scale = 0.0;
  for (int p = L_ - 1; p > 0; --p) {
    const int dl = dl_[p], d = dd_[p], dr = dr_[p];
    // C2sigmaB (_mps_cls.py:2684-2693)
    gauge_qr_right(site_[p].p, dl, d, dr, spare.p, tmp2_.p, sig_.p);
    std::swap(site_[p], spare);
    gauge_[p] = MITDVP_GAUGE_B;
    // sigma is rescaled to unit Frobenius norm on the device (unnormalised, e.g.
    // random, cores would otherwise grow geometrically along a long chain and
    // overflow); the factors are accumulated on the host for scale <= 0 (keep the
    // state's own normalisation: Liouville space, _mps_cls.py:2695-2699)
    {
      double* nrm = reinterpret_cast<double*>(red_.p + RED_MISC);
      vec_sumsq(st_, sig_.p, (long)dl * dl, nrm);
      vec_scale_inv_norm(st_, sig_.p, (long)dl * dl, nrm, 1e-300);
      if (scale <= 0.0) {
        read_partials(RED_MISC, NPART / 2);
        const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
        double t = 0;
        for (int i = 0; i < NPART; ++i) t += hp[i];
        if (t > 0) log_scale += 0.5 * std::log(t);
      }
    }
    // site[p-1] <- site[p-1] . sigma
    const int m = dl_[p - 1] * dd_[p - 1];
    ZgemmDesc g = zgemm_desc(site_[p - 1].p, sig_.p, spare.p, m, dl, dl);
    zgemm(st_, g);
    std::swap(site_[p - 1], spare);
    cnt_.n_launch += 1;
  }
  pool_put(std::move(spare));
  const long n0 = (long)dl_[0] * dd_[0] * dr_[0];
  vec_sumsq(st_, site_[0].p, n0, reinterpret_cast<double*>(red_.p + RED_MISC));
  read_partials(RED_MISC, NPART / 2);
  const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
  double s = 0;
  for (int i = 0; i < NPART; ++i) s += hp[i];
  if (s == 0.0) throw ArgError("canonicalize: zero state");
  if (scale > 0.0)
    vec_scale(st_, site_[0].p, n0, make_double2(scale / std::sqrt(s), 0.0));
  else
    vec_scale(st_, site_[0].p, n0, make_double2(std::exp(log_scale), 0.0));
  gauge_[0] = MITDVP_GAUGE_PSI;
  center_ = 0;
  invalidate_env();
}

// ---------------------------------------------------------------------------
// sweep
// ---------------------------------------------------------------------------
hzc Engine::scale_site(double dt) const { return cfg.relax ? hzc(-dt / 2, 0.0) : hzc(0.0, -dt / 2); }  // :1070 / :1088
hzc Engine::scale_bond(double dt) const { return cfg.relax ? hzc(+dt / 2, 0.0) : hzc(0.0, +dt / 2); }

void Engine::build_right_envs() {
  // construct_op_sites(begin=L-1, end=0) (_mps_cls.py:835-843, :1738-1796)
  for (int p = L_ - 1; p >= 1; --p) {
    if (envR_ok_[p]) continue;
    if (!envR_ok_[p + 1]) throw ArgError("internal: right environment chain broken");
    if (gauge_[p] != MITDVP_GAUGE_B) throw ArgError("sites right of the centre must be in gauge B");
    const MpoSite& w = mpo(0, p);
    transpose_rev3(st_, site_[p].p, tmp1_.p, dl_[p], dd_[p], dr_[p]);
    envR_[p] = pool_get((size_t)dl_[p] * w.ml * dl_[p]);
    env_update(envR_[p + 1].p, tmp1_.p, w.w2r.p, envR_[p].p, dr_[p], w.mr, dd_[p], dl_[p], w.ml);
    envR_ok_[p] = 1;
  }
}

void Engine::local_site_exp(int p, double dt) {
  const MpoSite& w = mpo(0, p);
  if (dd_[p] != w.d) throw ArgError("MPO physical dimension differs from the site tensor's");
  const int dl = dl_[p], d = dd_[p], dr = dr_[p];
  const zc* Lb = envL_[p].p;
  const zc* Rb = envR_[p + 1].p;
  const hzc shift = op(0).shift;
  auto mv = [&](const zc* in, zc* out) { heff_apply(Lb, w, Rb, in, out, dl, d, dr, shift); };
  if (cfg.relax == 2)  // improved relaxation, _mps_cls.py:1078-1084
    kprev_[p] = krylov_diag(mv, site_[p].p, (long)dl * d * dr);
  else
    kprev_[p] = krylov_exp(scale_site(dt), mv, site_[p].p, (long)dl * d * dr, kprev_[p]);
  cnt_.n_exp_site += 1;
}

void Engine::sweep(double dt, bool forward) {
  require_ready();
  if (L_ == 1) {
    if (center_ != 0) throw ArgError("no centre site");
    local_site_exp(0, dt);
    return;
  }
  const int begin = forward ? 0 : L_ - 1, end = forward ? L_ - 1 : 0;
  if (center_ != begin) throw ArgError("sweep must start at the centre (Psi) site");
  if (forward) build_right_envs();
  else
    for (int b = 1; b < L_; ++b)
      if (!envL_ok_[b]) throw ArgError("backward sweep needs the left environments of a forward sweep");
  const hzc shift = op(0).shift;
  DevBuf spare = pool_get(V_.n / MAXK);
  for (int p = begin; forward ? p <= end : p >= end; p += forward ? 1 : -1) {
    local_site_exp(p, dt);  // exp_superH_propagation_direct
    if (p == end) break;
    const MpoSite& w = mpo(0, p);
    const int dl = dl_[p], d = dd_[p], dr = dr_[p];
    if (forward) {
      // Psi2Asigma: site[p] (destroyed) -> A in spare, sigma in sig_
      timer_begin(3);
      long nl = 0;
      qr_householder(st_, site_[p].p, dl * d, dr, spare.p, sig_.p, qrwork_.p, &nl);
      timer_end();
      cnt_.n_launch += nl; cnt_.n_qr += 1;
      cnt_.qr_flops += 4.0 * (4.0 * (double)dl * d * dr * dr - 4.0 * (double)dr * dr * dr / 3.0);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_A;
      // renormalize_op_psite: L_{p+1}
      envL_[p + 1] = pool_get((size_t)dr * w.mr * dr);
      env_update(envL_[p].p, site_[p].p, w.w2l.p, envL_[p + 1].p, dl, w.ml, d, dr, w.mr);
      envL_ok_[p + 1] = 1;
      // exp(+i K dt/2) on the bond matrix
      const zc* Lb = envL_[p + 1].p;
      const zc* Rb = envR_[p + 1].p;
      const int m = w.mr;
      auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, dr, dr, m, shift); };
      if (cfg.relax != 2) {  // improved relaxation leaves the bond matrix alone (_mps_cls.py:1159-1160)
        kprev_[p] = krylov_exp(scale_bond(dt), mk, sig_.p, (long)dr * dr, kprev_[p]);
        cnt_.n_exp_bond += 1;
      }
      envR_ok_[p + 1] = 0;
      pool_put(std::move(envR_[p + 1]));
      // trans_next_psite_APsiB: Psi(p+1) = sigma . B(p+1)
      ZgemmDesc g = zgemm_desc(sig_.p, site_[p + 1].p, spare.p, dr, dd_[p + 1] * dr_[p + 1], dr);
      zgemm(st_, g);
      cnt_.n_launch += 1;
      std::swap(site_[p + 1], spare);
      gauge_[p + 1] = MITDVP_GAUGE_PSI;
      center_ = p + 1;
    } else {
      // Psi2sigmaB: B in spare, mirrored B~ (dr,d,dl) in tmp2_, sigma (dl x dl) in sig_
      gauge_qr_right(site_[p].p, dl, d, dr, spare.p, tmp2_.p, sig_.p);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_B;
      envR_[p] = pool_get((size_t)dl * w.ml * dl);
      env_update(envR_[p + 1].p, tmp2_.p, w.w2r.p, envR_[p].p, dr, w.mr, d, dl, w.ml);
      envR_ok_[p] = 1;
      const zc* Lb = envL_[p].p;
      const zc* Rb = envR_[p].p;
      const int m = w.ml;
      auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, dl, dl, m, shift); };
      if (cfg.relax != 2) {
        kprev_[p] = krylov_exp(scale_bond(dt), mk, sig_.p, (long)dl * dl, kprev_[p]);
        cnt_.n_exp_bond += 1;
      }
      envL_ok_[p] = 0;
      pool_put(std::move(envL_[p]));
      // Psi(p-1) = A(p-1) . sigma
      ZgemmDesc g = zgemm_desc(site_[p - 1].p, sig_.p, spare.p, dl_[p - 1] * dd_[p - 1], dl, dl);
      zgemm(st_, g);
      cnt_.n_launch += 1;
      std::swap(site_[p - 1], spare);
      gauge_[p - 1] = MITDVP_GAUGE_PSI;
      center_ = p - 1;
    }
  }
  pool_put(std::move(spare));
}

void Engine::step(double dt) {
  sweep(dt, true);
  sweep(dt, false);
}

// ---------------------------------------------------------------------------
// observables
// ---------------------------------------------------------------------------
double Engine::norm() {
  if (center_ < 0) throw ArgError("no centre site");
  const long n = (long)dl_[center_] * dd_[center_] * dr_[center_];
  vec_sumsq(st_, site_[center_].p, n, reinterpret_cast<double*>(red_.p + RED_MISC));
  read_partials(RED_MISC, NPART / 2);
  const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
  double s = 0;
  for (int i = 0; i < NPART; ++i) s += hp[i];
  return std::sqrt(s);
}

hzc Engine::expect(int op_id) {
  require_ready();
  if (center_ != 0) throw ArgError("expectation needs the centre at site 0 (psite = 0)");
  Operator& o = op(op_id);
  const zc* R1 = nullptr;
  DevBuf ra, rb;
  bool cached = (op_id == 0);
  for (int b = 1; b < L_ && cached; ++b) cached = envR_ok_[b];
  if (L_ == 1) {
    R1 = envR_[1].p;
  } else if (cached) {
    R1 = envR_[1].p;
  } else {
    // fresh right environments (_mps_cls.py:570-576)
    size_t mx = 1;
    for (int p = 1; p < L_; ++p) mx = std::max(mx, (size_t)dl_[p] * mpo(op_id, p).ml * dl_[p]);
    ra = pool_get(mx);
    rb = pool_get(mx);
    const zc* cur = envR_[L_].p;
    for (int p = L_ - 1; p >= 1; --p) {
      const MpoSite& w = mpo(op_id, p);
      if (gauge_[p] != MITDVP_GAUGE_B) throw ArgError("sites right of the centre must be in gauge B");
      transpose_rev3(st_, site_[p].p, tmp1_.p, dl_[p], dd_[p], dr_[p]);
      env_update(cur, tmp1_.p, w.w2r.p, ra.p, dr_[p], w.mr, dd_[p], dl_[p], w.ml);
      cur = ra.p;
      std::swap(ra, rb);  // result now lives in rb
    }
    R1 = cur;
  }
  const MpoSite& w0 = mpo(op_id, 0);
  heff_apply(envL_[0].p, w0, R1, site_[0].p, tmp2_.p, dl_[0], dd_[0], dr_[0], o.shift);
  const long n0 = (long)dl_[0] * dd_[0] * dr_[0];
  vec_dot(st_, site_[0].p, tmp2_.p, n0, true, red_.p + RED_MISC);
  read_partials(RED_MISC, NPART);
  double re = 0, im = 0;
  for (int i = 0; i < NPART; ++i) { re += h_red_[RED_MISC + i].x; im += h_red_[RED_MISC + i].y; }
  pool_put(std::move(ra));
  pool_put(std::move(rb));
  return hzc(re, im);
}

hzc Engine::autocorr() {
  require_ready();
  // <Psi^*|Psi>: block = einsum("abc,abk->ck", bra, einsum("ibk,ai->abk", ket, block))
  // with bra = ket unconjugated (wavefunction.py:226-257 with conj=False)
  const zc one = make_double2(1.0, 0.0);
  HIP_CHECK(hipMemcpyAsync(sig_.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  zc* T = sig_.p;
  zc* Tn = sig2_.p;
  for (int p = 0; p < L_; ++p) {
    const int dl = dl_[p], d = dd_[p], dr = dr_[p];
    ZgemmDesc u = zgemm_desc(T, site_[p].p, tmp1_.p, dl, d * dr, dl);  // U[m][(s,j)] = T[m][n] C[n][(s,j)]
    zgemm(st_, u);
    ZgemmDesc t = zgemm_desc(site_[p].p, tmp1_.p, Tn, dr, dr, dl * d);  // T'[i][j] = C[(m,s)][i] U[(m,s)][j]
    t.transA = 1; t.lda = dr;
    zgemm(st_, t);
    std::swap(T, Tn);
  }
  hzc out;
  HIP_CHECK(hipMemcpyAsync(&out, T, sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  return out;
}

void Engine::site_rdm(int isite, double* out) {
  require_ready();
  if (center_ != 0) throw ArgError("reduced density needs the centre at site 0");
  if (isite < 0 || isite >= L_) throw ArgError("bad site index");
  // T[a][a'] = sum over sites < isite of ket (x) conj(bra); sites > isite are right-canonical
  const zc one = make_double2(1.0, 0.0);
  HIP_CHECK(hipMemcpyAsync(sig_.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  zc* T = sig_.p;
  zc* Tn = sig2_.p;
  for (int p = 0; p <= isite; ++p) {
    const int dl = dl_[p], d = dd_[p], dr = dr_[p];
    // U[a'][(j,s)] = sum_a T[a][a'] C[a][(j,s)]
    ZgemmDesc u = zgemm_desc(T, site_[p].p, tmp1_.p, dl, d * dr, dl);
    u.transA = 1; u.lda = dl;
    zgemm(st_, u);
    if (p < isite) {
      // T'[s][s'] = sum_(a',j) U[(a',j)][s] conj(C[(a',j)][s'])
      ZgemmDesc t = zgemm_desc(tmp1_.p, site_[p].p, Tn, dr, dr, dl * d);
      t.transA = 1; t.lda = dr; t.conjB = 1;
      zgemm(st_, t);
      std::swap(T, Tn);
    } else {
      // rho_a'[j][j'] = sum_s U[a'][j][s] conj(C[a'][j'][s]); summed over a' on the host
      DevBuf rho = pool_get((size_t)dl * d * d);
      ZgemmDesc r = zgemm_desc(tmp1_.p, site_[p].p, rho.p, d, d, dr);
      r.transB = 1; r.conjB = 1; r.ldb = dr; r.ldc = d;
      r.batch = dl; r.strideA = (long)d * dr; r.strideB = (long)d * dr; r.strideC = (long)d * d;
      zgemm(st_, r);
      std::vector<hzc> h((size_t)dl * d * d);
      HIP_CHECK(hipMemcpyAsync(h.data(), rho.p, h.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
      pool_put(std::move(rho));
      hzc* o = reinterpret_cast<hzc*>(out);
      for (int e = 0; e < d * d; ++e) o[e] = hzc(0, 0);
      for (int a = 0; a < dl; ++a)
        for (int e = 0; e < d * d; ++e) o[e] += h[(size_t)a * d * d + e];
    }
  }
}

// General pure-state reduced density (_get_pure_reduced_density,
// _mps_cls.py:1208-1283): per site keep 2 legs (ket, bra), 1 leg (diagonal) or
// none.  Left-to-right transfer with the open physical legs folded into a batch
// index o: T_o[a][a'] (ket bond, bra bond); sites right of the last kept one
// are right-canonical and drop out.  Output axes: kept sites ascending, (ket,
// bra) per 2-leg site -- the reference's order.
void Engine::reduced_density(const int* legs, int nlen, std::vector<hzc>& out, std::vector<int>& shape) {
  require_ready();
  if (center_ != 0) throw ArgError("reduced density needs the centre at site 0");
  if (nlen < 1 || nlen > L_) throw ArgError("reduced_density: bad number of sites");
  int last = -1;
  for (int p = 0; p < nlen; ++p) {
    if (legs[p] < 0 || legs[p] > 2) throw ArgError("The number of legs must be less than 3.");
    if (legs[p]) last = p;
  }
  if (last < 0) throw ArgError("The number of legs must be greater than 0.");
  shape.clear();
  const zc one = make_double2(1.0, 0.0);
  long no = 1;
  DevBuf T = pool_get(1);
  HIP_CHECK(hipMemcpyAsync(T.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  for (int p = 0; p <= last; ++p) {
    const int dl = dl_[p], d = dd_[p], dr = dr_[p], n = legs[p];
    if (no > 65535) throw ArgError("reduced_density: too many open legs for one call");
    const zc* C = site_[p].p;
    DevBuf U = pool_get((size_t)no * dl * d * dr);
    {  // U_o[a'][(j,s)] = sum_a T_o[a][a'] C[a][(j,s)]
      ZgemmDesc g = zgemm_desc(T.p, C, U.p, dl, d * dr, dl);
      g.transA = 1; g.lda = dl; g.batch = (int)no;
      g.strideA = (long)dl * dl; g.strideB = 0; g.strideC = (long)dl * d * dr;
      zgemm(st_, g);
    }
    pool_put(std::move(T));
    if (p < last) {
      if (n == 0) {
        T = pool_get((size_t)no * dr * dr);
        ZgemmDesc g = zgemm_desc(U.p, C, T.p, dr, dr, dl * d);  // T'[s][s'] = U[(a',j)][s] conj(C[(a',j)][s'])
        g.transA = 1; g.lda = dr; g.conjB = 1; g.batch = (int)no;
        g.strideA = (long)dl * d * dr; g.strideB = 0; g.strideC = (long)dr * dr;
        zgemm(st_, g);
      } else if (n == 2) {
        const long ds = (long)d * dr;
        DevBuf Z = pool_get((size_t)no * ds * ds);
        ZgemmDesc g = zgemm_desc(U.p, C, Z.p, (int)ds, (int)ds, dl);  // Z[(j,s)][(j',s')]
        g.transA = 1; g.lda = ds; g.conjB = 1; g.batch = (int)no;
        g.strideA = (long)dl * ds; g.strideB = 0; g.strideC = ds * ds;
        zgemm(st_, g);
        T = pool_get((size_t)no * ds * ds);
        permute_0213(st_, Z.p, T.p, no * d, dr, d, dr);  // (o,j,s,j',s') -> (o,j,j',s,s')
        pool_put(std::move(Z));
        no *= (long)d * d;
        shape.push_back(d); shape.push_back(d);
      } else {
        T = pool_get((size_t)no * d * dr * dr);
        for (int j = 0; j < d; ++j) {  // T'_(o,j)[s][s'] = sum_a' U_o[a'][j][s] conj(C[a'][j][s'])
          ZgemmDesc g = zgemm_desc(U.p + (size_t)j * dr, C + (size_t)j * dr, T.p + (size_t)j * dr * dr, dr, dr, dl);
          g.transA = 1; g.lda = (long)d * dr; g.ldb = (long)d * dr; g.conjB = 1; g.batch = (int)no;
          g.strideA = (long)dl * d * dr; g.strideB = 0; g.strideC = (long)d * dr * dr;
          zgemm(st_, g);
        }
        no *= d;
        shape.push_back(d);
      }
    } else {
      // last kept site: the right side is the identity -> trace over s
      DevBuf Ut = pool_get((size_t)no * dl * d * dr), Ct = pool_get((size_t)dl * d * dr), rho = pool_get((size_t)no * d * d);
      permute_0213(st_, U.p, Ut.p, no, dl, d, dr);  // (o,a',j,s) -> (o,j,a',s)
      permute_0213(st_, C, Ct.p, 1, dl, d, dr);
      ZgemmDesc g = zgemm_desc(Ut.p, Ct.p, rho.p, d, d, dl * dr);  // rho_o[j][j'] = Ut_o[j][(a',s)] conj(Ct[j'][(a',s)])
      g.transB = 1; g.conjB = 1; g.ldb = (long)dl * dr; g.batch = (int)no;
      g.strideA = (long)d * dl * dr; g.strideB = 0; g.strideC = (long)d * d;
      zgemm(st_, g);
      std::vector<hzc> h((size_t)no * d * d);
      HIP_CHECK(hipMemcpyAsync(h.data(), rho.p, h.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
      if (n == 2) {
        out = std::move(h);
        shape.push_back(d); shape.push_back(d);
      } else {
        out.resize((size_t)no * d);
        for (long o = 0; o < no; ++o)
          for (int j = 0; j < d; ++j) out[(size_t)o * d + j] = h[((size_t)o * d + j) * d + j];
        shape.push_back(d);
      }
      pool_put(std::move(Ut)); pool_put(std::move(Ct)); pool_put(std::move(rho));
    }
    pool_put(std::move(U));
  }
  pool_put(std::move(T));
}

// ---------------------------------------------------------------------------
// bond truncation by SVD (truncate_sigvec, _site_cls.py:586-690) at the bond
// right of the centre site c:  Psi(c) = A sigma,  sigma = U s Vh;  keep the first
// idx singular values with cumulative weight sum_{k<idx} s_k / sum s_k >= 1 - p
// (and idx <= max_dim if max_dim > 0);  A <- A U,  B(c+1) <- Vh B(c+1),
// sigma' = diag(s / ||s||).  The result is stored as Psi(c) = A sigma', B(c+1).
// ---------------------------------------------------------------------------
int Engine::truncate_bond(double p, int max_dim, std::vector<double>& svals) {
  require_ready();
  const int c = center_;
  if (c < 0 || c >= L_ - 1) throw ArgError("truncate_bond: the centre must not be the last site");
  if (gauge_[c + 1] != MITDVP_GAUGE_B) throw ArgError("truncate_bond: the right neighbour must be in gauge B");
  const int dl = dl_[c], d = dd_[c], dr = dr_[c];
  const int dn = dd_[c + 1], drn = dr_[c + 1];
  DevBuf A = pool_get((size_t)dl * d * dr), U = pool_get((size_t)dr * dr), Vh = pool_get((size_t)dr * dr),
         work = pool_get(svd_work_elems(dr, dr));
  gauge_qr_left(site_[c].p, dl, d, dr, A.p, sig_.p);  // Psi2Asigma
  std::vector<double> s(dr);
  int sweeps = 0;
  svd_jacobi(st_, sig_.p, dr, dr, U.p, s.data(), Vh.p, work.p, &sweeps);
  double tot = 0;
  for (double v : s) tot += v;
  int idx = dr;
  double cum = 0;
  for (int k = 0; k < dr; ++k) {  // idx = argmax(cumsum / total >= 1 - p) + 1
    cum += s[k];
    if (cum / tot >= 1.0 - p) { idx = k + 1; break; }
  }
  if (max_dim > 0) idx = std::min(idx, max_dim);
  double nrm2 = 0;
  for (int k = 0; k < idx; ++k) nrm2 += s[k] * s[k];
  svals.assign(s.begin(), s.begin() + idx);
  for (auto& v : svals) v /= std::sqrt(nrm2);
  // A' sigma' = A U[:, :idx] diag(s'/||s'||): scale the kept columns of U first
  std::vector<hzc> hU((size_t)dr * dr);
  HIP_CHECK(hipMemcpyAsync(hU.data(), U.p, hU.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  std::vector<hzc> hUs((size_t)dr * idx);
  for (int r = 0; r < dr; ++r)
    for (int k = 0; k < idx; ++k) hUs[(size_t)r * idx + k] = hU[(size_t)r * dr + k] * svals[k];
  HIP_CHECK(hipMemcpyAsync(U.p, hUs.data(), hUs.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  DevBuf newc = pool_get(site_[c].n), newn = pool_get(site_[c + 1].n);
  {
    ZgemmDesc g = zgemm_desc(A.p, U.p, newc.p, dl * d, idx, dr);  // (dl d x dr) (dr x idx)
    zgemm(st_, g);
  }
  {
    ZgemmDesc g = zgemm_desc(Vh.p, site_[c + 1].p, newn.p, idx, dn * drn, dr);  // Vh[:idx] B
    zgemm(st_, g);
  }
  HIP_CHECK(hipStreamSynchronize(st_));
  std::swap(site_[c], newc);
  std::swap(site_[c + 1], newn);
  dr_[c] = idx;
  dl_[c + 1] = idx;
  invalidate_env();
  pool_put(std::move(A)); pool_put(std::move(U)); pool_put(std::move(Vh)); pool_put(std::move(work));
  pool_put(std::move(newc)); pool_put(std::move(newn));
  return idx;
}

// ---------------------------------------------------------------------------
// Liouville space: the MPS is a vectorised density matrix, site dimension n*n,
// physical index = row*n + col (reshape_mat, _mps_mpo.py:135-194)
// ---------------------------------------------------------------------------
void Engine::set_trace_op_core(int op_id, int isite, const double* reim, int ml, int n, int mr) {
  if (isite < 0 || isite >= L_) throw ArgError("set_trace_op_core: bad site index");
  if (ml < 1 || mr < 1 || n < 1) throw ArgError("set_trace_op_core: bad shape");
  const hzc* O = reinterpret_cast<const hzc*>(reim);  // O[a][d][c][f]  (bond, out, in, bond)
  std::vector<hzc> o2((size_t)mr * ml * n * n);
  for (int a = 0; a < ml; ++a)
    for (int dd = 0; dd < n; ++dd)
      for (int c = 0; c < n; ++c)
        for (int f = 0; f < mr; ++f)
          o2[(size_t)f * ml * n * n + ((size_t)a * n + c) * n + dd] = O[(((size_t)a * n + dd) * n + c) * mr + f];
  MpoSite& s = op(op_id).sites[isite];
  s.wtr.reserve(o2.size());
  HIP_CHECK(hipMemcpyAsync(s.wtr.p, o2.data(), o2.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  s.ntr = n; s.mltr = ml; s.mrtr = mr;
}

// Tr(O rho): left[f][e] = sum left[a][b] rho[b][c][d][e] O[a][d][c][f]   (_exp_liouville)
hzc Engine::expect_trace(int op_id) {
  require_ready();
  auto it = ops_.find(op_id);
  if (it == ops_.end()) throw ArgError("trace operator not set");
  const zc one = make_double2(1.0, 0.0);
  size_t mx = 1;
  for (int p = 0; p < L_; ++p) {
    const MpoSite& w = it->second.sites[p];
    if (!w.ntr) throw ArgError("trace operator core not set for this site");
    if (w.ntr * w.ntr != dd_[p]) throw ArgError("trace operator: site dimension is not n*n");
    mx = std::max(mx, (size_t)std::max(w.mltr, w.mrtr) * dd_[p] * std::max(dl_[p], dr_[p]));
  }
  DevBuf left = pool_get(mx), nxt = pool_get(mx), U = pool_get(mx);
  HIP_CHECK(hipMemcpyAsync(left.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  int ma = 1;
  for (int p = 0; p < L_; ++p) {
    const MpoSite& w = it->second.sites[p];
    if (w.mltr != ma) throw ArgError("trace operator: MPO bond mismatch");
    const int dl = dl_[p], d = dd_[p], dr = dr_[p];
    ZgemmDesc g1 = zgemm_desc(left.p, site_[p].p, U.p, ma, d * dr, dl);  // U[a][(c,d,e)]
    zgemm(st_, g1);
    ZgemmDesc g2 = zgemm_desc(w.wtr.p, U.p, nxt.p, w.mrtr, dr, ma * d);   // left'[f][e]
    zgemm(st_, g2);
    std::swap(left, nxt);
    ma = w.mrtr;
  }
  if (ma != 1) throw ArgError("trace operator: last core must close the MPO bond");
  hzc out;
  HIP_CHECK(hipMemcpyAsync(&out, left.p, sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  pool_put(std::move(left)); pool_put(std::move(nxt)); pool_put(std::move(U));
  return out;
}

// get_partial_trace (_mps_cls.py:1438-1510)
void Engine::partial_trace(const int* legs, int nlen, std::vector<hzc>& out) {
  require_ready();
  if (nlen < 1 || nlen > L_) throw ArgError("partial_trace: bad number of sites");
  int center = -1;
  for (int p = 0; p < nlen; ++p) {
    if (legs[p] < 0 || legs[p] > 2) throw ArgError("Invalid number of legs");
    if (legs[p]) center = p;
  }
  if (center < 0) throw ArgError("No site with 2 legs found in remain_nleg");
  std::vector<int> nn(L_);
  size_t maxd = 1;
  for (int p = 0; p < L_; ++p) {
    nn[p] = (int)std::lround(std::sqrt((double)dd_[p]));
    if (nn[p] * nn[p] != dd_[p]) throw ArgError("partial_trace: site dimension is not n*n");
    maxd = std::max(maxd, (size_t)std::max(dl_[p], dr_[p]));
  }
  const zc one = make_double2(1.0, 0.0);
  // right environment vector: sites right of the centre are traced out
  DevBuf right = pool_get(maxd), rnext = pool_get(maxd), tq = pool_get(maxd * maxd * 0 + (size_t)maxd * maxd);
  HIP_CHECK(hipMemcpyAsync(right.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  for (int q = L_ - 1; q > center; --q) {
    phys_diag(st_, site_[q].p, tq.p, dl_[q], nn[q], dr_[q], true);
    ZgemmDesc g = zgemm_desc(tq.p, right.p, rnext.p, dl_[q], 1, dr_[q]);
    zgemm(st_, g);
    std::swap(right, rnext);
  }
  // left environment with the open legs of the kept sites folded into its rows
  long no = 1;
  DevBuf left = pool_get(1);
  HIP_CHECK(hipMemcpyAsync(left.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  for (int q = 0; q < center; ++q) {
    const int dl = dl_[q], dr = dr_[q], n = nn[q];
    const zc* M = nullptr;
    DevBuf tmp;
    long cols;
    if (legs[q] == 2) {
      M = site_[q].p;
      cols = (long)n * n * dr;
    } else {
      tmp = pool_get((size_t)dl * n * dr);
      phys_diag(st_, site_[q].p, tmp.p, dl, n, dr, legs[q] == 0);
      M = tmp.p;
      cols = (legs[q] == 0 ? 1L : (long)n) * dr;
    }
    DevBuf nl = pool_get((size_t)no * cols);
    ZgemmDesc g = zgemm_desc(left.p, M, nl.p, (int)no, (int)cols, dl);
    zgemm(st_, g);
    pool_put(std::move(left));
    left = std::move(nl);
    no = no * cols / dr;
    pool_put(std::move(tmp));
  }
  {
    const int dl = dl_[center], dr = dr_[center], n = nn[center];
    DevBuf wv = pool_get((size_t)dl * n * n), dm = pool_get((size_t)no * n * n);
    ZgemmDesc g1 = zgemm_desc(site_[center].p, right.p, wv.p, dl * n * n, 1, dr);  // C (x) right
    zgemm(st_, g1);
    ZgemmDesc g2 = zgemm_desc(left.p, wv.p, dm.p, (int)no, n * n, dl);
    zgemm(st_, g2);
    out.resize((size_t)no * n * n);
    HIP_CHECK(hipMemcpyAsync(out.data(), dm.p, out.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    pool_put(std::move(wv)); pool_put(std::move(dm));
  }
  pool_put(std::move(left)); pool_put(std::move(right)); pool_put(std::move(rnext)); pool_put(std::move(tq));
}

void Engine::krylov_stats(int* per_site) const {
  for (int i = 0; i < L_; ++i) per_site[i] = kprev_[i];
}

}  // namespace mitdvp

#include "capi.inc"
