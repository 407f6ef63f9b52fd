// engine_small.hip -- how the engine uses the small-bond kernel family (small_site.hip): which sites
// qualify, the operand views of the three contraction chains, and the error / statistics hand-back.
//
// A site p is "small" when its H_eff chain, the K_eff chains of both its bonds and nothing else about
// the run (bond sharding, adaptive ranks, several electronic states, improved relaxation) rule it out;
// then exp(-i H_eff dt/2) and exp(+i K_eff dt/2) at p are ONE launch each and the host learns nothing
// about them until the sweep ends (Krylov counts stay on the device, keyed by site like
// _Debug.niter_krylov, _helper.py:29).  Single applies and environment updates take the one-launch form
// whenever their chain fits, whatever the mode.
#include "engine_internal.h"

#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace mitdvp {

// Compute-unit ranges claimed by CU-masked engines, per device.  Persistent grids (k_small_site, k_qr_panel) need all
// their workgroups resident at once: two masked engines whose ranges overlap, or a masked engine beside a full-device one,
// could each hold compute units the other's waiting workgroups need.  Overlapping claims are refused, and while any range
// is claimed on a device the full-device engines there run the multi-launch kernels (small_ok()).
namespace {
struct CuClaims {
  std::mutex mu;
  std::vector<std::pair<int, int>> r[64];
};
CuClaims& cu_claims() { static CuClaims c; return c; }
}  // namespace
void cu_range_claim(int device, int first, int count) {
  if (device < 0 || device >= 64) throw ArgError("cu_first / cu_count: device ordinal outside the claim table");
  CuClaims& c = cu_claims();
  std::lock_guard<std::mutex> lk(c.mu);
  for (const auto& q : c.r[device])
    if (first < q.first + q.second && q.first < first + count)
      throw ArgError("cu_first / cu_count: [" + std::to_string(first) + ", " + std::to_string(first + count) +
                     ") overlaps the range [" + std::to_string(q.first) + ", " + std::to_string(q.first + q.second) +
                     ") of another engine on this device");
  c.r[device].push_back({first, count});
}
void cu_range_release(int device, int first, int count) {
  if (device < 0 || device >= 64) return;
  CuClaims& c = cu_claims();
  std::lock_guard<std::mutex> lk(c.mu);
  for (size_t i = 0; i < c.r[device].size(); ++i)
    if (c.r[device][i].first == first && c.r[device][i].second == count) { c.r[device].erase(c.r[device].begin() + i); return; }
}
int cu_ranges_claimed(int device) {
  if (device < 0 || device >= 64) return 0;
  CuClaims& c = cu_claims();
  std::lock_guard<std::mutex> lk(c.mu);
  return (int)c.r[device].size();
}

bool Engine::small_ok() const {
  return small_kernels_ && nranks_ == 1 && n_cu_ > 0 && (ss_.partitioned || cu_ranges_claimed(cfg.device) == 0);
}

// sigma[a,i,r] = sum L[a,c,b] W[c,i,j,t] R[r,t,s] psi[b,j,s]   (_contraction.py:1182-1243)
bool Engine::chain_heff(SmallChain& c, const zc* L, const MpoSite& w, const zc* R, int dl, int d, int dr, bool exp_mode) const {
  c = SmallChain{};
  c.A = L; c.sAa = (long)w.ml * dl; c.sAc = dl; c.sAb = 1; c.conjA = 0;
  c.R = R; c.sRr = (long)w.mr * dr; c.sRt = dr; c.sRs = 1;
  c.W2 = w.w2l.p;
  c.sBb = (long)d * dr; c.sBj = dr; c.sBs = 1;
  c.na = dl; c.nb = dl; c.nc = w.ml; c.nj = d; c.ni = d; c.nt = w.mr; c.ns = dr; c.nr = dr;
  return small_chain_plan(c, exp_mode, n_cu_);
}

// sigma'[a,r] = sum L[a,c,b] sigma[b,s] R[r,c,s]   (_contraction.py:1339-1352)
bool Engine::chain_keff(SmallChain& c, const zc* L, const zc* R, int d1, int d2, int m, bool exp_mode) const {
  c = SmallChain{};
  c.A = L; c.sAa = (long)m * d1; c.sAc = d1; c.sAb = 1; c.conjA = 0;
  c.R = R; c.sRr = (long)m * d2; c.sRt = d2; c.sRs = 1;
  c.W2 = nullptr;
  c.sBb = d2; c.sBj = 0; c.sBs = 1;
  c.na = d1; c.nb = d1; c.nc = m; c.nj = 1; c.ni = 1; c.nt = m; c.ns = d2; c.nr = d2;
  return small_chain_plan(c, exp_mode, n_cu_);
}

// env'[i,q,j] = sum conj(T)[m,r,i] env[m,p,n] W2[(r,q),(p,s)] T[n,s,j]   (_contraction.py:148-397):
// slab = new bra index i; stage 1 contracts the old bra index m, stage 3 the old ket index n.
bool Engine::chain_env(SmallChain& c, const zc* T, const zc* w2e, int din, int min_, int d, int dout, int mout) const {
  c = SmallChain{};
  c.A = T; c.sAa = 1; c.sAc = dout; c.sAb = (long)d * dout; c.conjA = 1;
  c.R = T; c.sRr = 1; c.sRt = dout; c.sRs = (long)d * dout;
  c.W2 = w2e;
  c.sBb = (long)min_ * din; c.sBj = din; c.sBs = 1;
  c.na = dout; c.nb = din; c.nc = d; c.nj = min_; c.ni = mout; c.nt = d; c.ns = din; c.nr = dout;
  return small_chain_plan(c, false, n_cu_);
}

zc* Engine::ss_partials(const SmallChain& c) {
  ss_part_.reserve((size_t)c.nsc * c.na * c.ni * c.nr);
  small_sync_alloc(ss_, L_, st_);
  ss_dirty_ = true;
  return ss_part_.p;
}

SmallSync* Engine::qr_sync() {
  if (!small_kernels_ || n_cu_ <= 0) return nullptr;
  small_sync_alloc(ss_, L_, st_);
  ss_dirty_ = true;
  return &ss_;
}

// which sites run their local exponentials in one launch; device <-> host Krylov memories are
// reconciled whenever the answer changes (shapes change rarely: adaptive ranks, new tensors)
void Engine::ss_refresh_plan() {
  std::vector<int> key;
  key.reserve(3 * (size_t)L_ + 4);
  for (int p = 0; p < L_; ++p) { key.push_back(dl_[p]); key.push_back(dd_[p]); key.push_back(dr_[p]); }
  key.push_back(small_ok() ? 1 : 0);
  key.push_back(adaptive_ ? 1 : 0);
  key.push_back(ms_ ? 1 : 0);
  key.push_back(cfg.relax);
  auto it = ops_.find(0);
  for (int p = 0; p < L_; ++p) {
    const bool set = it != ops_.end() && it->second.sites[p].set;
    key.push_back(set ? it->second.sites[p].ml : -1);
    key.push_back(set ? it->second.sites[p].mr : -1);
  }
  if (key == ss_shape_key_ && (int)exp_small_.size() == L_) return;
  ss_pull_kprev();  // whatever the device knows goes to the host copy first
  ss_shape_key_ = key;
  exp_small_.assign(L_, 0);
  if (small_ok() && !adaptive_ && !ms_ && cfg.relax != 2 && it != ops_.end()) {
    for (int p = 0; p < L_; ++p) {
      const MpoSite& w = it->second.sites[p];
      if (!w.set || w.d != dd_[p]) continue;
      SmallChain c;
      bool ok = chain_heff(c, nullptr, w, nullptr, dl_[p], dd_[p], dr_[p], true);
      if (ok && p + 1 < L_) ok = chain_keff(c, nullptr, nullptr, dr_[p], dr_[p], w.mr, true);  // -> sweep: bond right of p
      if (ok && p > 0) ok = chain_keff(c, nullptr, nullptr, dl_[p], dl_[p], w.ml, true);        // <- sweep: bond left of p
      exp_small_[p] = ok ? 1 : 0;
    }
  }
  bool any = false;
  for (char f : exp_small_) any = any || f;
  if (any) {
    small_sync_alloc(ss_, L_, st_);
    HIP_CHECK(hipMemcpyAsync(ss_.kprev, kprev_.data(), (size_t)L_ * sizeof(int), hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
  }
}

void Engine::ss_pull_kprev() {
  if (!ss_.kprev || exp_small_.empty()) return;
  std::vector<int> h(L_);
  HIP_CHECK(hipMemcpyAsync(h.data(), ss_.kprev, (size_t)L_ * sizeof(int), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  for (int p = 0; p < L_; ++p)
    if (exp_small_[p]) kprev_[p] = h[p];
}

int Engine::kprev_get(int p) {
  if (p < 0 || p >= L_) throw ArgError("kprev_get: bad site index");
  if (!exp_small_.empty() && exp_small_[p]) ss_pull_kprev();
  return kprev_[p];
}

void Engine::kprev_set(int p, int k) {
  if (p < 0 || p >= L_) throw ArgError("kprev_set: bad site index");
  kprev_[p] = k;
  if (ss_.kprev && !exp_small_.empty() && exp_small_[p]) {
    HIP_CHECK(hipMemcpyAsync(ss_.kprev + p, &kprev_[p], sizeof(int), hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
  }
}

void Engine::set_small_kernels(bool on) {
  if (on == small_kernels_) return;
  ss_check();
  ss_pull_kprev();
  small_kernels_ = on;
  ss_shape_key_.clear();  // the plan is rebuilt at the next require_ready
}

void Engine::ss_check() {
  if (!ss_dirty_ || !ss_.words) return;
  ss_dirty_ = false;
  unsigned w[4] = {0, 0, 0, 0};
  long long stats[4] = {0, 0, 0, 0};
  HIP_CHECK(hipMemcpyAsync(w, ss_.words, sizeof(w), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipMemcpyAsync(stats, ss_.stats, sizeof(stats), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  // applies counted on the device (their number is decided there)
  cnt_.n_heff += stats[0];
  cnt_.n_keff += stats[1];
  cnt_.heff_flops += (double)stats[2];
  cnt_.keff_flops += (double)stats[3];
  if (stats[0] | stats[1] | stats[2] | stats[3]) {
    HIP_CHECK(hipMemsetAsync(ss_.stats, 0, sizeof(stats), st_));
  }
  if (w[3] != 0 || w[2] != 0) {
    HIP_CHECK(hipMemsetAsync(ss_.words, 0, 4 * sizeof(unsigned), st_));  // counters, abort flag, error code
    HIP_CHECK(hipStreamSynchronize(st_));
    if (w[3] == SS_ENOTCONV)
      throw NotConverged(std::string(cfg.integrator == MITDVP_LANCZOS ? "Short Iterative Lanczos" : "Short Iterative Arnoldi") +
                         " is not converged in " + std::to_string(cfg.max_krylov) + " basis. Try shorter time interval.");
    if (w[3] == SS_EZERO) throw ArgError("Initial psi has zero norm.");
    throw HipError("small-site kernel: a grid-wide exchange timed out (its workgroups were not all resident: is another "
                   "process or engine using this GPU?  MITDVP_SMALL_KERNELS=0 selects the multi-launch kernels)");
  }
}

// exp(scale * H_eff) on the centre tensor of site p in one launch
bool Engine::small_site_exp(int p, double dt) {
  if (exp_small_.empty() || !exp_small_[p]) return false;
  const MpoSite& w = mpo(0, p);
  const int dl = dl_[p], d = dd_[p], dr = dr_[p];
  SmallChain c;
  if (!chain_heff(c, envL_[p].p, w, envR_[p + 1].p, dl, d, dr, true)) return false;
  SmallExp e{};
  e.integrator = cfg.integrator; e.variant = cfg.lanczos_variant; e.conserve_norm = cfg.conserve_norm;
  e.max_krylov = cfg.max_krylov; e.thresh = cfg.thresh;
  const hzc s = scale_site(dt);
  e.scale_re = s.real(); e.scale_im = s.imag();
  e.site = p; e.stat_slot = 0;
  e.flops_per_apply = (long long)(8.0 * ((double)dl * dl * w.ml * d * dr + (double)dl * dr * w.ml * w.mr * d * d +
                                         (double)dl * dr * dr * w.mr * d));
  const hzc sh = op(0).shift;
  zc* part = ss_partials(c);
  timer_begin(0);
  small_exp(st_, ss_, c, e, site_[p].p, V_.p, part, make_double2(sh.real(), sh.imag()));
  timer_end();
  cnt_.n_launch += 1;
  cnt_.n_exp_site += 1;
  return true;
}

// exp(scale * K_eff) on the bond matrix sig_ (dim x dim) in one launch; Krylov memory of site p
bool Engine::small_bond_exp(int p, const zc* Lb, const zc* Rb, int dim, int m, double dt) {
  if (exp_small_.empty() || !exp_small_[p]) return false;
  SmallChain c;
  if (!chain_keff(c, Lb, Rb, dim, dim, m, true)) return false;
  SmallExp e{};
  e.integrator = cfg.integrator; e.variant = cfg.lanczos_variant; e.conserve_norm = cfg.conserve_norm;
  e.max_krylov = cfg.max_krylov; e.thresh = cfg.thresh;
  const hzc s = scale_bond(dt);
  e.scale_re = s.real(); e.scale_im = s.imag();
  e.site = p; e.stat_slot = 1;
  e.flops_per_apply = (long long)(16.0 * (double)m * dim * dim * dim);
  const hzc sh = op(0).shift;
  zc* part = ss_partials(c);
  timer_begin(2);
  small_exp(st_, ss_, c, e, sig_.p, V_.p, part, make_double2(sh.real(), sh.imag()));
  timer_end();
  cnt_.n_launch += 1;
  cnt_.n_exp_bond += 1;
  return true;
}

}  // namespace mitdvp
