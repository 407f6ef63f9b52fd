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
#include <cstdlib>
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
  if (const char* e = std::getenv("MITDVP_SMALL_KERNELS")) small_kernels_ = std::atoi(e) != 0;
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
  if (pool_.size() > 48) pool_.erase(pool_.begin());  // adaptive ranks: block sizes drift, drop the oldest
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

void Engine::ensure_work(long max_site, long max_x, long max_y, int max_qr_m, int max_qr_n, int qr_next) {
  X_.reserve(max_x);
  Y_.reserve(max_y);
  V_.reserve((size_t)MAXK * max_site);
  tmp1_.reserve(max_site);
  tmp2_.reserve(max_site);
  const size_t dd = (size_t)max_qr_n * max_qr_n;
  sig_.reserve(std::max<size_t>(dd, 1));
  sig2_.reserve(std::max<size_t>(dd, 1));
  qrwork_.reserve(qr_work_elems(max_qr_m, max_qr_n, qr_next));
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

// The blocks may be rectangular (bra bond != ket bond): L (dlo, ml, dli), R (dro, mr, dri),
// psi (dli, d, dri) -> out (dlo, d, dro).  That is the adaptive-rank case
// (tensor_shapes_out, _contraction.py:455-477); the plain sweep has dlo == dli, dro == dri.
void Engine::heff_apply_rect(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dlo, int dli,
                             int d, int dro, int dri) {
  const int ml = w.ml, mr = w.mr;
  int a0, a1;
  const bool sharded = shard_range(dlo, a0, a1);
  const int na = a1 - a0;
  timer_begin(10);
  {  // X[(a,c)][(j,s)] = L[(a,c)][b] psi[b][(j,s)]
    ZgemmDesc g = zgemm_desc(L + (size_t)a0 * ml * dli, psi, X_.p, na * ml, d * dri, dli);
    zgemm(st_, g);
  }
  timer_end();
  timer_begin(11);
  {  // Y_a[(i,t)][s] = W2L[(i,t)][(c,j)] X_a[(c,j)][s]
    ZgemmDesc g = zgemm_desc(w.w2l.p, X_.p, Y_.p, d * mr, dri, ml * d);
    g.batch = na; g.strideA = 0; g.strideB = (long)ml * d * dri; g.strideC = (long)d * mr * dri;
    zgemm(st_, g);
  }
  timer_end();
  timer_begin(12);
  {  // out[(a,i)][r] = Y[(a,i)][(t,s)] R[r][(t,s)]
    ZgemmDesc g = zgemm_desc(Y_.p, R, out + (size_t)a0 * d * dro, na * d, dro, mr * dri);
    g.transB = 1; g.ldb = (long)mr * dri;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLGATHER, out, (size_t)dlo * d * dro);
  cnt_.n_launch += 3;
  cnt_.n_heff += 1;
  cnt_.heff_flops += 8.0 * ((double)na * dli * ml * d * dri + (double)na * dri * ml * mr * d * d + (double)na * dro * dri * mr * d);
}

void Engine::heff_apply(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dl, int d, int dr,
                        hzc shift) {
  heff_apply_rect(L, w, R, psi, out, dl, dl, d, dr, dr);
  if (shift != hzc(0.0, 0.0))
    vec_axpby(st_, out, psi, (long)dl * d * dr, make_double2(shift.real(), shift.imag()), make_double2(1.0, 0.0));
}

// L (dlo, m, dli), R (dro, m, dri), sig (dli, dri) -> out (dlo, dro)
void Engine::keff_apply_rect(const zc* L, const zc* R, const zc* sig, zc* out, int dlo, int dli, int dro, int dri,
                             int m) {
  int a0, a1;
  const bool sharded = shard_range(dlo, a0, a1);
  const int na = a1 - a0;
  timer_begin(2);
  {  // X[(a,c)][s] = L[(a,c)][b] sig[b][s]
    ZgemmDesc g = zgemm_desc(L + (size_t)a0 * m * dli, sig, X_.p, na * m, dri, dli);
    zgemm(st_, g);
  }
  {  // out[a][r] = X[a][(c,s)] R[r][(c,s)]
    ZgemmDesc g = zgemm_desc(X_.p, R, out + (size_t)a0 * dro, na, dro, m * dri);
    g.transB = 1; g.ldb = (long)m * dri;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLGATHER, out, (size_t)dlo * dro);
  cnt_.n_launch += 2;
  cnt_.n_keff += 1;
  cnt_.keff_flops += 8.0 * ((double)na * dli * m * dri + (double)na * dro * dri * m);
}

void Engine::keff_apply(const zc* L, const zc* R, const zc* sig, zc* out, int d1, int d2, int m, hzc shift) {
  keff_apply_rect(L, R, sig, out, d1, d1, d2, d2, m);
  if (shift != hzc(0.0, 0.0))
    vec_axpby(st_, out, sig, (long)d1 * d2, make_double2(shift.real(), shift.imag()), make_double2(1.0, 0.0));
}

// env_in (dbi, min, dki), ket tensor Tk (dki, d, dko), bra tensor Tb (dbi, d, dbo),
// W2 ((d*mout) x (min*d)) -> env_out (dbo, mout, dko).  Tb != Tk is the adaptive-rank
// "bra" block (superblock_states_bra, _mps_cls.py:1950-1963).
void Engine::env_update_rect(const zc* env_in, const zc* Tk, const zc* Tb, const zc* w2, zc* env_out, int dbi, int dki,
                             int min_, int d, int dbo, int dko, int mout) {
  int m0, m1;
  const bool sharded = shard_range(dbi, m0, m1);
  const int nm = m1 - m0;
  timer_begin(1);
  {  // X[(m,p)][(s,j)] = env[(m,p)][n] Tk[n][(s,j)]
    ZgemmDesc g = zgemm_desc(env_in + (size_t)m0 * min_ * dki, Tk, X_.p, nm * min_, d * dko, dki);
    zgemm(st_, g);
  }
  {  // Y_m[(r,q)][j] = W2[(r,q)][(p,s)] X_m[(p,s)][j]
    ZgemmDesc g = zgemm_desc(w2, X_.p, Y_.p, d * mout, dko, min_ * d);
    g.batch = nm; g.strideA = 0; g.strideB = (long)min_ * d * dko; g.strideC = (long)d * mout * dko;
    zgemm(st_, g);
  }
  {  // env'[i][(q,j)] = conj(Tb)[(m,r)][i] Y[(m,r)][(q,j)]   (sum over this rank's m)
    ZgemmDesc g = zgemm_desc(Tb + (size_t)m0 * d * dbo, Y_.p, env_out, dbo, mout * dko, nm * d);
    g.transA = 1; g.conjA = 1; g.lda = dbo;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLREDUCE, env_out, (size_t)dbo * mout * dko);
  cnt_.n_launch += 3;
  cnt_.n_env += 1;
  cnt_.env_flops += 8.0 * ((double)nm * dki * min_ * d * dko + (double)nm * dko * min_ * mout * d * d +
                           (double)nm * dbo * dko * mout * d);
}

void Engine::env_update(const zc* env_in, const zc* T, const zc* w2, zc* env_out, int din, int min_, int d, int dout,
                        int mout) {
  env_update_rect(env_in, T, T, w2, env_out, din, din, min_, d, dout, dout, mout);
}

// ---------------------------------------------------------------------------
// local propagator: x <- exp(scale*Op) x
// ---------------------------------------------------------------------------
template <class MV>
int Engine::krylov_exp(hzc scale, MV&& matvec, zc* x, long n, int k_prev, long nsize) {
  // nsize: element count of the UNPADDED input tensor (adaptive rank: x is zero-padded to
  // the output shape but _iter_info still counts psi_states, _integrator.py:178-186, :524)
  if (nsize <= 0) nsize = n;
  const int ndim = (int)std::min<long>(nsize, cfg.max_krylov);
  const int n_warm = (int)std::min<long>(nsize, std::min(std::max(0, k_prev - 2), 15));
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
    if (lanczos && n <= SMALL_VEC_N && small_kernels_) {
      // small-bond regime: dot, update, norm and normalisation in one single-workgroup launch
      vec_lanczos_step_small(st_, vn, cfg.lanczos_variant == 0 ? V : vl, vl, l > 0 ? V + (size_t)(l - 1) * ldv : nullptr, n,
                             alpha_p + (size_t)l * NPART, l > 0 ? nrm_p + (size_t)(l - 1) * NPART : nullptr,
                             nrm_p + (size_t)l * NPART, KRYLOV_EPS);
      cnt_.n_launch += 1;
    } else {
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
      cnt_.n_launch += 3;
    }
    timer_end();

    const bool last_possible = (l + 1 == nsize);
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
      if (b < KRYLOV_EPS || q + 1 == nsize) {  // Krylov space exhausted (:569, :392)
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

void Engine::build_left_envs() {
  // construct_op_sites(begin=0, end=L-1): needed when a gate / Kraus map re-orthogonalised
  // sites after the forward half-sweep (op_sys_sites = None, _mps_cls.py:2370)
  for (int p = 0; p < L_ - 1; ++p) {
    if (envL_ok_[p + 1]) continue;
    if (!envL_ok_[p]) throw ArgError("internal: left environment chain broken");
    if (gauge_[p] != MITDVP_GAUGE_A) throw ArgError("sites left of the centre must be in gauge A");
    const MpoSite& w = mpo(0, p);
    pool_put(std::move(envL_[p + 1]));
    envL_[p + 1] = pool_get((size_t)dr_[p] * w.mr * dr_[p]);
    env_update(envL_[p].p, site_[p].p, w.w2l.p, envL_[p + 1].p, dl_[p], w.ml, dd_[p], dr_[p], w.mr);
    envL_ok_[p + 1] = 1;
  }
}

// ---------------------------------------------------------------------------
// one-site gates (Model(one_gate_to_apply=...), MPSCoef.apply_one_gate,
// _mps_cls.py:2314-2373, :2420-2451): out[a, d', c] = sum_b U[d', b] site[a, b, c],
// then re-orthogonalisation towards the current centre over the touched span
// (canonicalizeB / canonicalizeA, :3539-3598); the environment blocks that saw a
// touched site are dropped and rebuilt by the next half-sweep.
// ---------------------------------------------------------------------------
void Engine::set_gate(int isite, const double* reim, int d) {
  if (isite < 0 || isite >= L_) throw ArgError("set_gate: bad site index");
  if (!reim) { gates_.erase(isite); return; }
  if (d < 1) throw ArgError("set_gate: bad dimension");
  std::vector<zc> h((size_t)d * d);
  for (size_t i = 0; i < h.size(); ++i) h[i] = make_double2(reim[2 * i], reim[2 * i + 1]);
  Gate& g = gates_[isite];
  g.d = d;
  g.u.reserve(h.size());
  HIP_CHECK(hipMemcpyAsync(g.u.p, h.data(), h.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

void Engine::apply_gates() {
  if (gates_.empty()) return;
  require_ready();
  if (center_ < 0) throw ArgError("apply_gates: the MPS has no centre (Psi) site");
  DevBuf spare = pool_get(V_.n / MAXK);
  int lo = L_, hi = -1;
  for (auto& kv : gates_) {
    const int p = kv.first;
    const Gate& g = kv.second;
    const int l = dl_[p], d = dd_[p], r = dr_[p];
    if (g.d != d) throw ArgError("gate dimension differs from the site's physical dimension");
    ZgemmDesc z = zgemm_desc(g.u.p, site_[p].p, spare.p, d, r, d);
    z.batch = l; z.strideA = 0; z.strideB = (long)d * r; z.strideC = (long)d * r;
    zgemm(st_, z);
    cnt_.n_launch += 1;
    std::swap(site_[p], spare);
    if (p != center_) {
      gauge_[p] = MITDVP_GAUGE_C;
      lo = std::min(lo, p); hi = std::max(hi, p);
    }
  }
  recanonicalize(lo, hi, spare);
  pool_put(std::move(spare));
}

// canonicalizeB(superblock[centre : hi + 1]) and canonicalizeA(superblock[lo : centre + 1])
// (_mps_cls.py:3539-3598) after sites in [lo, hi] were modified; the environment blocks that
// contain a modified site are dropped (op_sys_sites = None, :2370, :2417)
void Engine::recanonicalize(int lo, int hi, DevBuf& spare) {
  const int c0 = center_;
  if (hi > c0) {
    for (int p = hi; p > c0; --p) {
      gauge_qr_right(site_[p].p, dl_[p], dd_[p], dr_[p], spare.p, tmp2_.p, sig_.p);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_B;
      const int m = dl_[p - 1] * dd_[p - 1];
      ZgemmDesc z = zgemm_desc(site_[p - 1].p, sig_.p, spare.p, m, dl_[p], dl_[p]);
      zgemm(st_, z);
      cnt_.n_launch += 1;
      std::swap(site_[p - 1], spare);
    }
    for (int b = 1; b <= hi; ++b) { envR_ok_[b] = 0; pool_put(std::move(envR_[b])); }
  }
  if (lo < c0) {
    for (int p = lo; p < c0; ++p) {
      gauge_qr_left(site_[p].p, dl_[p], dd_[p], dr_[p], spare.p, sig_.p);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_A;
      ZgemmDesc z = zgemm_desc(sig_.p, site_[p + 1].p, spare.p, dr_[p], dd_[p + 1] * dr_[p + 1], dr_[p]);
      zgemm(st_, z);
      cnt_.n_launch += 1;
      std::swap(site_[p + 1], spare);
    }
    for (int b = lo + 1; b < L_; ++b) { envL_ok_[b] = 0; pool_put(std::move(envL_[b])); }
  }
  gauge_[c0] = MITDVP_GAUGE_PSI;
}

// ---------------------------------------------------------------------------
// Kraus maps on purified states (Model(kraus_op=...), MPSCoef.apply_kraus,
// _mps_cls.py:2375-2418; kraus.py:146-358).  theta (m, d*K, n) has the physical index
// (system d, ancilla K); C[(m,n,x),(k,K)] = sum_d B[k,x,d] theta[m,d,K,n] and the ancilla
// index (k,K) is cut back to K keeping "U S" of the leading singular values.  One-sided
// Jacobi on the k*K ROWS of C^T delivers exactly that factor (the rotated rows are
// s_i q_i), so neither U, V nor a normalisation is formed.
// ---------------------------------------------------------------------------
void Engine::set_kraus(int isite, int two_site, const double* reim, int k, int d) {
  if (isite < 0 || isite >= L_ || (two_site && isite + 1 >= L_)) throw ArgError("set_kraus: bad site index");
  if (!reim) { kraus_.erase(isite); return; }
  if (k < 1 || d < 1) throw ArgError("set_kraus: bad Kraus tensor shape");
  std::vector<zc> h((size_t)k * d * d);
  for (size_t i = 0; i < h.size(); ++i) h[i] = make_double2(reim[2 * i], reim[2 * i + 1]);
  KrausOp& o = kraus_[isite];
  o.k = k; o.d = d; o.two_site = two_site != 0;
  o.b.reserve(h.size());
  HIP_CHECK(hipMemcpyAsync(o.b.p, h.data(), h.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

void Engine::kraus_core(const zc* theta, int m, int d, int K, int n, const KrausOp& op, zc* out) {
  const int k = op.k, x = d;
  const size_t tot = (size_t)m * k * x * K * n;
  DevBuf T = pool_get(tot), M = pool_get(tot);
  {  // T[m][(k,x)][(K,n)] = B[(k,x)][d] theta[m][d][(K,n)]
    ZgemmDesc z = zgemm_desc(op.b.p, theta, T.p, k * x, K * n, d);
    z.batch = m; z.strideA = 0; z.strideB = (long)d * K * n; z.strideC = (long)k * x * K * n;
    zgemm(st_, z);
  }
  {  // M[(k,K)][(m,x,n)] = T[m][k][x][K][n]
    const int dims[5] = {k, K, m, x, n};
    const long str[5] = {(long)x * K * n, (long)n, (long)k * x * K * n, (long)K * n, 1};
    permute5(st_, T.p, M.p, dims, str, nullptr);
  }
  const int nr = k * K, nc = m * x * n;
  DevBuf wk = pool_get((size_t)nr + 8 + (size_t)(nr + 1) / 2);
  int* idx_dev = reinterpret_cast<int*>(wk.p + nr / 2 + 4);
  std::vector<double> S(nr);
  int sweeps = 0;
  svd_rows_us(st_, M.p, nr, nc, S.data(), idx_dev, wk.p, &sweeps);
  {  // out[m][x][K''][n] = M[idx[K'']][(m,x,n)]
    const int dims[5] = {m, x, K, n, 1};
    const long str[5] = {(long)x * n, (long)n, (long)nc, 1, 0};
    permute5(st_, M.p, out, dims, str, idx_dev);
  }
  HIP_CHECK(hipStreamSynchronize(st_));  // idx lives in wk
  cnt_.n_launch += 3 + (long)sweeps * (nr + (nr & 1) - 1);
  pool_put(std::move(T)); pool_put(std::move(M)); pool_put(std::move(wk));
}

void Engine::apply_kraus() {
  if (kraus_.empty()) return;
  require_ready();
  if (center_ < 0) throw ArgError("apply_kraus: the MPS has no centre (Psi) site");
  DevBuf spare = pool_get(V_.n / MAXK);
  int lo = L_, hi = -1;
  for (auto& kv : kraus_) {
    const int p = kv.first;
    const KrausOp& op = kv.second;
    if (!op.two_site) {
      const int l = dl_[p], dim = dd_[p], r = dr_[p];
      if (dim % op.d != 0) throw ArgError("Kraus contract: dK must be divisible by d");
      kraus_core(site_[p].p, l, op.d, dim / op.d, r, op, spare.p);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_C;
      lo = std::min(lo, p); hi = std::max(hi, p);
      continue;
    }
    const int q = p + 1;
    const int m = dl_[p], d = dd_[p], l = dr_[p], K = dd_[q], n = dr_[q];
    if (d != op.d) throw ArgError("two-site Kraus map: the system site's dimension differs from the Kraus operators'");
    DevBuf theta = pool_get((size_t)m * d * K * n), c2 = pool_get((size_t)m * d * K * n);
    {  // theta[m][d][(K,n)] = A1[(m,d)][l] A2[l][(K,n)]
      ZgemmDesc z = zgemm_desc(site_[p].p, site_[q].p, theta.p, m * d, K * n, l);
      zgemm(st_, z);
    }
    kraus_core(theta.p, m, d, K, n, op, c2.p);  // (m, x, K, n) = matrix (m x) x (K n)
    const int rr = m * d, cc = K * n, kk = std::min(rr, cc), lnew = std::min(l, kk);
    DevBuf U = pool_get((size_t)rr * kk), Vh = pool_get((size_t)kk * cc), wk = pool_get(svd_work_elems(rr, cc));
    std::vector<double> S(kk);
    svd_jacobi(st_, c2.p, rr, cc, U.p, S.data(), Vh.p, wk.p, nullptr);
    // A1 = U[:, :l] S[:l], A2 = Vh[:l]  (kraus.py:338-353)
    copy2d(st_, site_[p].p, lnew, U.p, kk, rr, lnew, 0, make_double2(1.0, 0.0), false);
    double* sdev = reinterpret_cast<double*>(wk.p);
    HIP_CHECK(hipMemcpyAsync(sdev, S.data(), lnew * sizeof(double), hipMemcpyHostToDevice, st_));
    scale_cols(st_, site_[p].p, rr, lnew, lnew, sdev);
    HIP_CHECK(hipMemcpyAsync(site_[q].p, Vh.p, (size_t)lnew * cc * sizeof(zc), hipMemcpyDeviceToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    dr_[p] = lnew; dl_[q] = lnew;
    gauge_[p] = MITDVP_GAUGE_C; gauge_[q] = MITDVP_GAUGE_C;
    lo = std::min(lo, p); hi = std::max(hi, q);
    cnt_.n_launch += 6;
    pool_put(std::move(theta)); pool_put(std::move(c2)); pool_put(std::move(U)); pool_put(std::move(Vh)); pool_put(std::move(wk));
  }
  recanonicalize(lo, hi, spare);
  pool_put(std::move(spare));
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
  else build_left_envs();
  const hzc shift = op(0).shift;
  if (adaptive_) {
    if (cfg.relax) throw ArgError("adaptive bond dimension is implemented for real-time propagation only");
    adaptive_prepare();
    build_superblock_full(forward);
  }
  DevBuf spare = pool_get(V_.n / MAXK);
  for (int p = begin; forward ? p <= end : p >= end; p += forward ? 1 : -1) {
    if (adaptive_ && p != end && adaptive_site(p, dt, forward, spare)) continue;
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

// ---------------------------------------------------------------------------
// adaptive bond dimension (a1TDVP): const.adaptive branches of
// propagate_along_sweep (_mps_cls.py:863-987), get_adaptive_rank_and_block
// (:2152-2286), get_rank_and_projection_error (:1985-2105), thin_to_full
// (_site_cls.py:294-405).  Every block that the reference builds twice ("bra"
// and "braket") is built once here from the widened neighbour tensor and
// sliced: the leading columns / rows of the widened tensor ARE the thin tensor.
// ---------------------------------------------------------------------------
void Engine::set_adaptive(bool on, int dmax, int dd, double p_proj) {
  if (on && (dmax < 1 || dd < 0 || !(p_proj >= 0.0))) throw ArgError("set_adaptive: need Dmax >= 1, dD >= 0, p_proj >= 0");
  adaptive_ = on; ad_dmax_ = dmax; ad_dd_ = dd; ad_p_ = p_proj;
}

// workspaces for the largest shapes the bonds can reach during this sweep
void Engine::adaptive_prepare() {
  std::vector<long> cap(L_ + 1, 1);  // cap[b]: largest possible dimension of the bond left of site b
  {
    std::vector<double> lp(L_ + 1, 1.0), rp(L_ + 1, 1.0);
    for (int b = 1; b <= L_; ++b) lp[b] = std::min(1e15, lp[b - 1] * dd_[b - 1]);
    for (int b = L_ - 1; b >= 0; --b) rp[b] = std::min(1e15, rp[b + 1] * dd_[b]);
    for (int b = 0; b <= L_; ++b) cap[b] = (long)std::min(lp[b], rp[b]);
  }
  auto bound = [&](int b) -> long {  // bond left of site b, widened tensors included
    const long cur = b == 0 ? 1 : (b == L_ ? 1 : dl_[b]);
    if (b == 0 || b == L_) return 1;
    return std::min<long>(cap[b], std::max<long>(cur, ad_dmax_) + ad_dd_);
  };
  long ms = 1, mx = 1, my = 1;
  int qm = 1, qn = 1;
  for (int p = 0; p < L_; ++p) {
    const long bl = bound(p), br = bound(p + 1);
    ms = std::max(ms, bl * dd_[p] * br);
    qm = std::max<long>(qm, std::max(bl, br) * dd_[p]);
    qn = std::max<long>(qn, std::max(bl, br));
    const MpoSite& w = mpo(0, p);
    const long mm = std::max(w.ml, w.mr);
    mx = std::max(mx, bl * br * dd_[p] * mm);
    my = mx;
  }
  ensure_work(ms, mx, my, qm, qn, ad_dd_ + 1);
  for (int p = 0; p < L_; ++p) site_[p].grow_preserve((size_t)ms, (size_t)dl_[p] * dd_[p] * dr_[p], st_);
  if (full_.size() != (size_t)L_) { full_.clear(); full_.resize(L_); fdl_.assign(L_, 0); fdr_.assign(L_, 0); }
}

// (l, c, r) isometry over (l c) x r -> (l, c, r + e): e more orthonormal columns
void Engine::thin_to_full_A(const zc* A, int l, int c, int r, int e, zc* out) {
  const size_t n = (size_t)l * c * r;
  if (e == 0) {
    HIP_CHECK(hipMemcpyAsync(out, A, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
    return;
  }
  HIP_CHECK(hipMemcpyAsync(tmp1_.p, A, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  long nl = 0;
  timer_begin(3);
  qr_householder(st_, tmp1_.p, l * c, r, out, nullptr, qrwork_.p, &nl, e);
  timer_end();
  // sign alignment (_site_cls.py:321-335): the leading columns equal the input
  copy2d(st_, out, r + e, A, r, (long)l * c, r, 0, make_double2(1.0, 0.0), false);
  cnt_.n_launch += nl + 1;
  cnt_.n_qr += 1;
}

// (l, c, r) isometry over l x (c r) -> (l + e, c, r): e more orthonormal rows
void Engine::thin_to_full_B(const zc* B, int l, int c, int r, int e, zc* out) {
  const size_t n = (size_t)l * c * r;
  if (e > 0) {
    const int m = c * r;
    transpose_batched(st_, B, tmp1_.p, l, m, m, l, 1, 0, 0);  // mat = B.reshape(l, c r).T, _site_cls.py:357
    long nl = 0;
    timer_begin(3);
    qr_householder(st_, tmp1_.p, m, l, tmp2_.p, nullptr, qrwork_.p, &nl, e);
    timer_end();
    transpose_batched(st_, tmp2_.p, out, m, l + e, l + e, m, 1, 0, 0);
    cnt_.n_launch += nl + 2;
    cnt_.n_qr += 1;
  }
  HIP_CHECK(hipMemcpyAsync(out, B, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
}

// get_superblock_full / get_actual_delta_rank (_mps_cls.py:3699-3755)
void Engine::build_superblock_full(bool forward) {
  for (int q = 0; q < L_; ++q) {
    if (q == (forward ? 0 : L_ - 1)) continue;
    const int l1 = dl_[q], c1 = dd_[q], r1 = dr_[q];
    pool_put(std::move(full_[q]));
    if (forward) {  // gauge B, neighbour q-1
      const int l2 = dl_[q - 1], c2 = dd_[q - 1], r2 = dr_[q - 1];
      const long e = std::max<long>(0, std::min<long>(ad_dd_, std::min((long)c1 * r1 - l1, (long)l2 * c2 - r2)));
      full_[q] = pool_get((size_t)(l1 + e) * c1 * r1);
      thin_to_full_B(site_[q].p, l1, c1, r1, (int)e, full_[q].p);
      fdl_[q] = l1 + (int)e; fdr_[q] = r1;
    } else {  // gauge A, neighbour q+1
      const int l2 = dl_[q + 1], c2 = dd_[q + 1], r2 = dr_[q + 1];
      const long e = std::max<long>(0, std::min<long>(ad_dd_, std::min((long)l1 * c1 - r1, (long)c2 * r2 - l2)));
      full_[q] = pool_get((size_t)l1 * c1 * (r1 + e));
      thin_to_full_A(site_[q].p, l1, c1, r1, (int)e, full_[q].p);
      fdl_[q] = l1; fdr_[q] = r1 + (int)e;
    }
  }
}

// the D loop of get_rank_and_projection_error (_mps_cls.py:2083-2105):
// f(D) = |H psi_left[..., :D]|^2 - |K sigma[:D, :D]|^2 + |H psi_right[:D, ...]|^2
int Engine::select_rank(const zc* hl, long hl_rows, const zc* ks, const zc* hr, long hr_cols, int dmin, int dmax) {
  DevBuf prof = pool_get((size_t)(3 * dmax) / 2 + 2);
  double* pd = reinterpret_cast<double*>(prof.p);
  col_sumsq(st_, hl, hl_rows, dmax, pd);
  row_sumsq(st_, hr, dmax, hr_cols, pd + dmax);
  shell_sumsq(st_, ks, dmax, pd + 2 * (size_t)dmax);
  std::vector<double> h(3 * (size_t)dmax);
  HIP_CHECK(hipMemcpyAsync(h.data(), pd, h.size() * sizeof(double), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  cnt_.n_launch += 3;
  pool_put(std::move(prof));
  double a = 0, b = 0, k = 0, prev = 0;
  for (int D = 1; D <= dmax; ++D) {
    a += h[D - 1]; b += h[(size_t)dmax + D - 1]; k += h[2 * (size_t)dmax + D - 1];
    if (D < dmin) continue;
    const double tot = a - k + b;
    if (D > dmin) {
      const double metric = (tot - prev) / tot;
      if (metric < ad_p_) return D - 1;
    }
    prev = tot;
  }
  return dmax;
}

// one site of an adaptive half-sweep; false: the bond is at maximal rank
// (is_max_rank, _mps_cls.py:3757-3766) and the caller does the plain step
bool Engine::adaptive_site(int p, double dt, bool forward, DevBuf& spare) {
  const hzc shift = op(0).shift;
  const zc one = make_double2(1.0, 0.0);
  const zc zshift = make_double2(shift.real(), shift.imag());
  const bool has_shift = shift != hzc(0.0, 0.0);
  const int l = dl_[p], c = dd_[p], r = dr_[p];
  const MpoSite& wp = mpo(0, p);
  if (c != wp.d) throw ArgError("MPO physical dimension differs from the site tensor's");
  long nl = 0;
  if (forward) {
    if ((long)l * c <= r || r >= ad_dmax_) return false;
    const int q = p + 1;
    const MpoSite& wq = mpo(0, q);
    const int cq = dd_[q], rq = dr_[q], Df = fdl_[q], M = wp.mr;
    // environment right of site p from the widened B(q): "braket", and its ket-thin slice "bra"
    DevBuf fm = pool_get((size_t)Df * cq * rq);
    transpose_rev3(st_, full_[q].p, fm.p, Df, cq, rq);
    DevBuf env_braket = pool_get((size_t)Df * M * Df);
    env_update(envR_[q + 1].p, fm.p, wq.w2r.p, env_braket.p, rq, wq.mr, cq, Df, M);
    pool_put(std::move(fm));
    DevBuf env_bra = pool_get((size_t)Df * M * r);
    copy2d(st_, env_bra.p, r, env_braket.p, Df, (long)Df * M, r, 0, one, false);
    int dmax = std::min(ad_dmax_, Df);
    // get_psi_sigvec_psi_fullblock: Psi = A sigma, Psi' = sigma B(q), widened A
    DevBuf A = pool_get((size_t)l * c * r);
    gauge_qr_left(site_[p].p, l, c, r, A.p, sig_.p);
    DevBuf psip = pool_get((size_t)r * cq * rq);
    {
      ZgemmDesc g = zgemm_desc(sig_.p, site_[q].p, psip.p, r, cq * rq, r);
      zgemm(st_, g);
    }
    const int ea = (int)std::min<long>(dmax - r, (long)l * c - r);
    DevBuf Afull = pool_get((size_t)l * c * (r + ea));
    thin_to_full_A(A.p, l, c, r, ea, Afull.p);
    DevBuf sys_bra = pool_get((size_t)(r + ea) * M * r);
    env_update_rect(envL_[p].p, A.p, Afull.p, wp.w2l.p, sys_bra.p, l, l, wp.ml, c, r + ea, r, M);
    pool_put(std::move(A));
    pool_put(std::move(Afull));
    dmax = (int)std::min<long>(dmax, std::min((long)l * c, (long)cq * rq));
    int newD = r;
    if (r != dmax) {
      DevBuf hl = pool_get((size_t)l * c * dmax), hr = pool_get((size_t)dmax * cq * rq), ks = pool_get((size_t)dmax * dmax);
      heff_apply_rect(envL_[p].p, wp, env_bra.p, site_[p].p, hl.p, l, l, c, dmax, r);
      heff_apply_rect(sys_bra.p, wq, envR_[q + 1].p, psip.p, hr.p, dmax, r, cq, rq, rq);
      keff_apply_rect(sys_bra.p, env_bra.p, sig_.p, ks.p, dmax, r, dmax, r, M);
      newD = select_rank(hl.p, (long)l * c, ks.p, hr.p, (long)cq * rq, r, dmax);
      pool_put(std::move(hl)); pool_put(std::move(hr)); pool_put(std::move(ks));
    }
    pool_put(std::move(psip));
    pool_put(std::move(sys_bra));
    // blocks at the chosen rank: bra = leading newD*M rows of env_bra, braket = [:newD, :, :newD]
    DevBuf envD_braket = pool_get((size_t)newD * M * newD);
    copy2d(st_, envD_braket.p, newD, env_braket.p, Df, (long)newD * M, newD, 0, one, false);
    pool_put(std::move(env_braket));
    // B(q) <- widened B(q)[:newD]
    HIP_CHECK(hipMemcpyAsync(site_[q].p, full_[q].p, (size_t)newD * cq * rq * sizeof(zc), hipMemcpyDeviceToDevice, st_));
    dl_[q] = newD;
    // exp(-i H dt/2) on the zero-padded centre tensor; every apply sees the vector cut
    // back to the old shape (SplitStack.split(truncate=True), _contraction.py:593-610)
    copy2d(st_, spare.p, newD, site_[p].p, r, (long)l * c, r, newD, one, false);
    std::swap(site_[p], spare);
    dr_[p] = newD;
    {
      const zc* Lb = envL_[p].p;
      const zc* Rb = env_bra.p;
      auto mv = [&](const zc* in, zc* out) {
        copy2d(st_, tmp2_.p, r, in, newD, (long)l * c, r, 0, one, false);
        heff_apply_rect(Lb, wp, Rb, tmp2_.p, out, l, l, c, newD, r);
        if (has_shift) copy2d(st_, out, newD, tmp2_.p, r, (long)l * c, r, 0, zshift, true);
        cnt_.n_launch += has_shift ? 2 : 1;
      };
      kprev_[p] = krylov_exp(scale_site(dt), mv, site_[p].p, (long)l * c * newD, kprev_[p], (long)l * c * r);
      cnt_.n_exp_site += 1;
    }
    pool_put(std::move(env_bra));
    // from here on the plain step at the new rank
    timer_begin(3);
    qr_householder(st_, site_[p].p, l * c, newD, spare.p, sig_.p, qrwork_.p, &nl);
    timer_end();
    cnt_.n_launch += nl; cnt_.n_qr += 1;
    cnt_.qr_flops += 4.0 * (4.0 * (double)l * c * newD * newD - 4.0 * (double)newD * newD * newD / 3.0);
    std::swap(site_[p], spare);
    gauge_[p] = MITDVP_GAUGE_A;
    pool_put(std::move(envL_[q]));
    envL_[q] = pool_get((size_t)newD * M * newD);
    env_update(envL_[p].p, site_[p].p, wp.w2l.p, envL_[q].p, l, wp.ml, c, newD, M);
    envL_ok_[q] = 1;
    {
      const zc* Lb = envL_[q].p;
      const zc* Rb = envD_braket.p;
      auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, newD, newD, M, shift); };
      kprev_[p] = krylov_exp(scale_bond(dt), mk, sig_.p, (long)newD * newD, kprev_[p]);
      cnt_.n_exp_bond += 1;
    }
    pool_put(std::move(envD_braket));
    envR_ok_[q] = 0;
    pool_put(std::move(envR_[q]));
    ZgemmDesc g = zgemm_desc(sig_.p, site_[q].p, spare.p, newD, cq * rq, newD);
    zgemm(st_, g);
    cnt_.n_launch += 1;
    std::swap(site_[q], spare);
    gauge_[q] = MITDVP_GAUGE_PSI;
    center_ = q;
    return true;
  }
  // ---- backward: the mirror image ------------------------------------------
  if (l >= (long)c * r || l >= ad_dmax_) return false;
  const int q = p - 1;
  const MpoSite& wq = mpo(0, q);
  const int lq = dl_[q], cq = dd_[q], Df = fdr_[q], M = wp.ml;
  DevBuf env_braket = pool_get((size_t)Df * M * Df);
  env_update(envL_[q].p, full_[q].p, wq.w2l.p, env_braket.p, lq, wq.ml, cq, Df, M);
  DevBuf env_bra = pool_get((size_t)Df * M * l);
  copy2d(st_, env_bra.p, l, env_braket.p, Df, (long)Df * M, l, 0, one, false);
  int dmax = std::min(ad_dmax_, Df);
  DevBuf B = pool_get((size_t)l * c * r), Bt = pool_get((size_t)l * c * r);
  gauge_qr_right(site_[p].p, l, c, r, B.p, Bt.p, sig_.p);
  DevBuf psip = pool_get((size_t)lq * cq * l);
  {
    ZgemmDesc g = zgemm_desc(site_[q].p, sig_.p, psip.p, lq * cq, l, l);
    zgemm(st_, g);
  }
  const int eb = (int)std::min<long>(dmax - l, (long)c * r - l);
  DevBuf Bfull = pool_get((size_t)(l + eb) * c * r), Bfull_t = pool_get((size_t)(l + eb) * c * r);
  thin_to_full_B(B.p, l, c, r, eb, Bfull.p);
  transpose_rev3(st_, Bfull.p, Bfull_t.p, l + eb, c, r);
  DevBuf sys_bra = pool_get((size_t)(l + eb) * M * l);
  env_update_rect(envR_[p + 1].p, Bt.p, Bfull_t.p, wp.w2r.p, sys_bra.p, r, r, wp.mr, c, l + eb, l, M);
  pool_put(std::move(B)); pool_put(std::move(Bt)); pool_put(std::move(Bfull)); pool_put(std::move(Bfull_t));
  dmax = (int)std::min<long>(dmax, std::min((long)lq * cq, (long)c * r));
  int newD = l;
  if (l != dmax) {
    DevBuf hl = pool_get((size_t)lq * cq * dmax), hr = pool_get((size_t)dmax * c * r), ks = pool_get((size_t)dmax * dmax);
    heff_apply_rect(envL_[q].p, wq, sys_bra.p, psip.p, hl.p, lq, lq, cq, dmax, l);
    heff_apply_rect(env_bra.p, wp, envR_[p + 1].p, site_[p].p, hr.p, dmax, l, c, r, r);
    keff_apply_rect(env_bra.p, sys_bra.p, sig_.p, ks.p, dmax, l, dmax, l, M);
    newD = select_rank(hl.p, (long)lq * cq, ks.p, hr.p, (long)c * r, l, dmax);
    pool_put(std::move(hl)); pool_put(std::move(hr)); pool_put(std::move(ks));
  }
  pool_put(std::move(psip));
  pool_put(std::move(sys_bra));
  DevBuf envD_braket = pool_get((size_t)newD * M * newD);
  copy2d(st_, envD_braket.p, newD, env_braket.p, Df, (long)newD * M, newD, 0, one, false);
  pool_put(std::move(env_braket));
  // A(q) <- widened A(q)[:, :, :newD]
  copy2d(st_, site_[q].p, newD, full_[q].p, Df, (long)lq * cq, newD, 0, one, false);
  dr_[q] = newD;
  // zero-padded centre tensor (newD, c, r): the old tensor is the leading block
  HIP_CHECK(hipMemcpyAsync(spare.p, site_[p].p, (size_t)l * c * r * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  if (newD > l) HIP_CHECK(hipMemsetAsync(spare.p + (size_t)l * c * r, 0, (size_t)(newD - l) * c * r * sizeof(zc), st_));
  std::swap(site_[p], spare);
  dl_[p] = newD;
  {
    const zc* Lb = env_bra.p;
    const zc* Rb = envR_[p + 1].p;
    auto mv = [&](const zc* in, zc* out) {
      heff_apply_rect(Lb, wp, Rb, in, out, newD, l, c, r, r);
      if (has_shift) vec_axpby(st_, out, in, (long)l * c * r, zshift, one);
    };
    kprev_[p] = krylov_exp(scale_site(dt), mv, site_[p].p, (long)newD * c * r, kprev_[p], (long)l * c * r);
    cnt_.n_exp_site += 1;
  }
  pool_put(std::move(env_bra));
  gauge_qr_right(site_[p].p, newD, c, r, spare.p, tmp2_.p, sig_.p);
  std::swap(site_[p], spare);
  gauge_[p] = MITDVP_GAUGE_B;
  pool_put(std::move(envR_[p]));
  envR_[p] = pool_get((size_t)newD * M * newD);
  env_update(envR_[p + 1].p, tmp2_.p, wp.w2r.p, envR_[p].p, r, wp.mr, c, newD, M);
  envR_ok_[p] = 1;
  {
    const zc* Lb = envD_braket.p;
    const zc* Rb = envR_[p].p;
    auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, newD, newD, M, shift); };
    kprev_[p] = krylov_exp(scale_bond(dt), mk, sig_.p, (long)newD * newD, kprev_[p]);
    cnt_.n_exp_bond += 1;
  }
  pool_put(std::move(envD_braket));
  envL_ok_[p] = 0;
  pool_put(std::move(envL_[p]));
  ZgemmDesc g = zgemm_desc(site_[q].p, sig_.p, spare.p, lq * cq, newD, newD);
  zgemm(st_, g);
  cnt_.n_launch += 1;
  std::swap(site_[q], spare);
  gauge_[q] = MITDVP_GAUGE_PSI;
  center_ = q;
  return true;
}

void Engine::step(double dt) {
  sweep(dt, true);
  apply_gates();  // Model(one_gate_to_apply=...), _mps_cls.py:489-490 (reorth_center = nsite - 1)
  apply_kraus();  // Model(kraus_op=...), :491-492
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
