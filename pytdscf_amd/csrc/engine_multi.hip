// engine_multi.hip -- several electronic states (MPS-SM with nstate > 1).
//
// The reference keeps one MPS per electronic state (superblock_states[istate][isite],
// _mps_cls.py:84-118) and a Hamiltonian with one MPO block per (bra, ket) state pair
// (TensorHamiltonian.mpo[i][j] and the scalar coupleJ[i][j], hamiltonian_cls.py:618-752).
// At every site the centre tensors of ALL states form one stacked Krylov vector
// (SplitStack.stack/split, _contraction.py:479-608) and
//     sigma_i = sum_j  L_ij . W_ij . R_ij  psi_j  +  coupleJ_ij  Lov_ij psi_j Rov_ij
// (multiplyH_MPS_direct_MPO.dot, _contraction.py:1182-1243), with environment blocks per state
// pair built from bra tensors of state i and ket tensors of state j (renormalize_op_psite,
// _mps_mpo.py:421-696).  For i == j the overlap blocks are the identity; for i != j they are
// carried as a chain of identity cores (M = 1), exactly like the scalar term of `operate`.
// The gauge move is a separate QR per state, the bond matrices are again stacked for the
// K_eff solve, and one Krylov count per site is shared by everything (_helper.py:29).
//
// Everything runs through the same kernels as the single-state path: the rectangular
// applies / environment update (bra bond != ket bond), the Householder QR and the Krylov
// drivers of engine_krylov.inc.
#include "engine_krylov.inc"

namespace mitdvp {

struct Engine::Multi {
  struct OpMs {
    std::vector<std::vector<MpoSite>> blk;  // [i*S+j][site]
    std::vector<char> has;                  // block (i,j) has an MPO
    std::vector<hzc> cj;                    // scalar terms
  };
  struct Chain {
    int i = 0, j = 0;
    hzc f{1.0, 0.0};
    const std::vector<MpoSite>* w = nullptr;
    std::vector<DevBuf> L, R;  // by bond b = 0..nsite (bond b is left of site b)
    std::vector<char> Lok, Rok;
  };
  int S = 0;
  std::vector<std::vector<DevBuf>> site;       // [state][site]
  std::vector<std::vector<int>> dl, dr, gauge;  // [state][site]
  std::vector<int> d;                           // physical dimensions (shared by the states)
  std::map<int, OpMs> ops;
  std::vector<MpoSite> ident;  // identity cores (M = 1) for the overlap chains
  std::vector<Chain> chains;   // of operator 0
  bool chains_ok = false;
  int center = -1;
  DevBuf stack, acc, sigstack, spare, one;
  std::vector<DevBuf> bt;  // mirrored (dr, d, dl) tensors of the current site, per state
  std::vector<long> off;   // offsets of the states inside the stacked vector
};

Engine::Multi& Engine::ms() {
  if (!ms_) throw ArgError("the handle is not in multi-state mode (mitdvp_ms_configure)");
  return *ms_;
}
int Engine::ms_nstate() const { return ms_ ? ms_->S : 0; }

void Engine::ms_configure(int nstate) {
  if (nstate < 1 || nstate > 64) throw ArgError("ms_configure: nstate must be in [1, 64]");
  ms_ = std::make_shared<Multi>();
  Multi& m = *ms_;
  m.S = nstate;
  m.site.resize(nstate);
  for (auto& v : m.site) v.resize(L_);
  m.dl.assign(nstate, std::vector<int>(L_, 0));
  m.dr.assign(nstate, std::vector<int>(L_, 0));
  m.gauge.assign(nstate, std::vector<int>(L_, -1));
  m.d.assign(L_, 0);
  m.ident.resize(L_);
  m.bt.resize(nstate);
  m.off.assign(nstate + 1, 0);
  m.one.reserve(1);
  const zc one = make_double2(1.0, 0.0);
  HIP_CHECK(hipMemcpyAsync(m.one.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

void Engine::ms_set_site(int s, int p, const double* reim, int l, int n, int r, int gauge) {
  Multi& m = ms();
  if (s < 0 || s >= m.S) throw ArgError("ms_set_site: bad state index");
  if (p < 0 || p >= L_) throw ArgError("ms_set_site: bad site index");
  if (l < 1 || n < 1 || r < 1) throw ArgError("ms_set_site: bad shape");
  if (m.d[p] != 0 && m.d[p] != n) {
    bool other = false;
    for (int t = 0; t < m.S; ++t) other |= (t != s && m.site[t][p].p != nullptr);
    if (other) throw ArgError("ms_set_site: all states must share the physical dimension of a site");
  }
  const size_t e = (size_t)l * n * r;
  m.site[s][p].reserve(e);
  HIP_CHECK(hipMemcpyAsync(m.site[s][p].p, reim, e * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  m.dl[s][p] = l; m.dr[s][p] = r; m.gauge[s][p] = gauge; m.d[p] = n;
  if (gauge == MITDVP_GAUGE_PSI) m.center = p;
  m.chains_ok = false;
}

void Engine::ms_get_site_shape(int s, int p, int* l, int* n, int* r, int* gauge) {
  Multi& m = ms();
  if (s < 0 || s >= m.S || p < 0 || p >= L_) throw ArgError("ms_get_site_shape: bad index");
  *l = m.dl[s][p]; *n = m.d[p]; *r = m.dr[s][p]; *gauge = m.gauge[s][p];
}

void Engine::ms_get_site(int s, int p, double* out) {
  Multi& m = ms();
  if (s < 0 || s >= m.S || p < 0 || p >= L_ || !m.site[s][p].p) throw ArgError("ms_get_site: bad or unset site");
  const size_t e = (size_t)m.dl[s][p] * m.d[p] * m.dr[s][p];
  HIP_CHECK(hipMemcpyAsync(out, m.site[s][p].p, e * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

void Engine::ms_set_mpo_core(int op_id, int i, int j, int p, const double* reim, int ml, int dout, int din, int mr) {
  Multi& m = ms();
  if (i < 0 || i >= m.S || j < 0 || j >= m.S) throw ArgError("ms_set_mpo_core: bad state index");
  if (p < 0 || p >= L_) throw ArgError("ms_set_mpo_core: bad site index");
  Multi::OpMs& o = m.ops[op_id];
  if (o.blk.empty()) {
    o.blk.resize((size_t)m.S * m.S);
    for (auto& b : o.blk) b.resize(L_);
    o.has.assign((size_t)m.S * m.S, 0);
    o.cj.assign((size_t)m.S * m.S, hzc(0, 0));
  }
  upload_mpo_core(o.blk[(size_t)i * m.S + j][p], reim, ml, dout, din, mr);
  o.has[(size_t)i * m.S + j] = 1;
  if (op_id == 0) m.chains_ok = false;
}

void Engine::ms_set_couplej(int op_id, int i, int j, double re, double im) {
  Multi& m = ms();
  if (i < 0 || i >= m.S || j < 0 || j >= m.S) throw ArgError("ms_set_couplej: bad state index");
  Multi::OpMs& o = m.ops[op_id];
  if (o.blk.empty()) {
    o.blk.resize((size_t)m.S * m.S);
    for (auto& b : o.blk) b.resize(L_);
    o.has.assign((size_t)m.S * m.S, 0);
    o.cj.assign((size_t)m.S * m.S, hzc(0, 0));
  }
  o.cj[(size_t)i * m.S + j] = hzc(re, im);
  if (op_id == 0) m.chains_ok = false;
}

// right-to-left QR of one state's tensors, site 0 scaled to `scale` (alloc_superblock_random,
// _mps_cls.py:2684-2699; scale = sqrt(weight of the state), _mps_mpo.py:88-94; 0 is allowed)
void Engine::ms_canonicalize(int s, double scale) {
  Multi& m = ms();
  if (s < 0 || s >= m.S) throw ArgError("ms_canonicalize: bad state index");
  if (scale < 0.0) throw ArgError("ms_canonicalize: scale must be >= 0");
  ms_require_ready();
  for (int p = L_ - 1; p > 0; --p) {
    const int dl = m.dl[s][p], d = m.d[p], dr = m.dr[s][p];
    gauge_qr_right(m.site[s][p].p, dl, d, dr, m.spare.p, tmp2_.p, sig_.p);
    std::swap(m.site[s][p], m.spare);
    m.gauge[s][p] = MITDVP_GAUGE_B;
    double* nrm = reinterpret_cast<double*>(red_.p + RED_MISC);
    vec_sumsq(st_, sig_.p, (long)dl * dl, nrm);
    vec_scale_inv_norm(st_, sig_.p, (long)dl * dl, nrm, 1e-300);
    const int mm = m.dl[s][p - 1] * m.d[p - 1];
    ZgemmDesc g = zgemm_desc(m.site[s][p - 1].p, sig_.p, m.spare.p, mm, dl, dl);
    zgemm(st_, g);
    std::swap(m.site[s][p - 1], m.spare);
  }
  const long n0 = (long)m.dl[s][0] * m.d[0] * m.dr[s][0];
  vec_sumsq(st_, m.site[s][0].p, n0, reinterpret_cast<double*>(red_.p + RED_MISC));
  read_partials(RED_MISC, NPART / 2);
  const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
  double t = 0;
  for (int i = 0; i < NPART; ++i) t += hp[i];
  if (t == 0.0) throw ArgError("ms_canonicalize: zero state");
  vec_scale(st_, m.site[s][0].p, n0, make_double2(scale / std::sqrt(t), 0.0));
  m.gauge[s][0] = MITDVP_GAUGE_PSI;
  m.center = 0;
  m.chains_ok = false;
}

void Engine::ms_require_ready() {
  Multi& m = ms();
  long stack_max = 1, single_max = 1, sig_max = 1, mx = 1;
  int qm = 1, qn = 1, mmax = 1;
  for (auto& kv : m.ops)
    for (auto& b : kv.second.blk)
      for (auto& w : b)
        if (w.set) mmax = std::max(mmax, std::max(w.ml, w.mr));
  for (int p = 0; p < L_; ++p) {
    long tot = 0, sg = 0;
    int dmax = 1;
    for (int s = 0; s < m.S; ++s) {
      if (!m.site[s][p].p) throw ArgError("site tensor not set");
      if (p + 1 < L_ && m.dr[s][p] != m.dl[s][p + 1]) throw ArgError("bond dimension mismatch between neighbouring sites");
      const long e = (long)m.dl[s][p] * m.d[p] * m.dr[s][p];
      tot += e;
      single_max = std::max(single_max, e);
      const int dd = std::max(m.dl[s][p], m.dr[s][p]);
      dmax = std::max(dmax, dd);
      sg += (long)dd * dd;
      qm = std::max(qm, dd * m.d[p]);
      qn = std::max(qn, dd);
    }
    stack_max = std::max(stack_max, tot);
    sig_max = std::max(sig_max, sg);
    mx = std::max(mx, (long)dmax * dmax * m.d[p] * mmax);
  }
  for (int s = 0; s < m.S; ++s)
    if (m.dl[s][0] != 1 || m.dr[s][L_ - 1] != 1) throw ArgError("open boundary bonds must be 1");
  ensure_work(stack_max, mx, mx, qm, qn);
  m.stack.reserve(stack_max);
  m.acc.reserve(single_max);
  m.sigstack.reserve(sig_max);
  m.spare.reserve(single_max);
  for (int s = 0; s < m.S; ++s) {
    m.bt[s].reserve(single_max);
    for (int p = 0; p < L_; ++p)
      m.site[s][p].grow_preserve((size_t)single_max, (size_t)m.dl[s][p] * m.d[p] * m.dr[s][p], st_);
  }
}

// identity cores (M = 1) of the overlap chains that carry the scalar terms
void Engine::ms_ident_cores() {
  Multi& m = ms();
  for (int p = 0; p < L_; ++p) {
    if (m.ident[p].set && m.ident[p].d == m.d[p]) continue;
    const int d = m.d[p];
    std::vector<hzc> eye((size_t)d * d, hzc(0, 0));
    for (int a = 0; a < d; ++a) eye[(size_t)a * d + a] = hzc(1, 0);
    upload_mpo_core(m.ident[p], reinterpret_cast<const double*>(eye.data()), 1, d, d, 1);
  }
}

// the chains that enter operator 0: MPO blocks, and overlap chains for i != j with a scalar term
void Engine::ms_build_chains() {
  Multi& m = ms();
  if (m.chains_ok) return;
  m.chains.clear();
  auto it = m.ops.find(0);
  if (it == m.ops.end()) throw ArgError("operator 0 (the Hamiltonian) is not set");
  Multi::OpMs& o = it->second;
  auto add = [&](int i, int j, hzc f, const std::vector<MpoSite>* w) {
    Multi::Chain c;
    c.i = i; c.j = j; c.f = f; c.w = w;
    c.L.resize(L_ + 1); c.R.resize(L_ + 1);
    c.Lok.assign(L_ + 1, 0); c.Rok.assign(L_ + 1, 0);
    m.chains.push_back(std::move(c));
  };
  for (int i = 0; i < m.S; ++i)
    for (int j = 0; j < m.S; ++j) {
      const size_t ij = (size_t)i * m.S + j;
      if (o.has[ij]) {
        for (int p = 0; p < L_; ++p) {
          const MpoSite& w = o.blk[ij][p];
          if (!w.set) throw ArgError("an MPO block is missing cores (every block needs all sites)");
          if (w.d != m.d[p]) throw ArgError("MPO physical dimension differs from the site tensor's");
          if (p + 1 < L_ && w.mr != o.blk[ij][p + 1].ml) throw ArgError("MPO bond mismatch");
        }
        if (o.blk[ij][0].ml != 1 || o.blk[ij][L_ - 1].mr != 1) throw ArgError("MPO boundary bonds must be 1");
        add(i, j, hzc(1.0, 0.0), &o.blk[ij]);
      }
      if (i != j && o.cj[ij] != hzc(0, 0)) {
        ms_ident_cores();
        add(i, j, o.cj[ij], &m.ident);
      }
    }
  // every chain parks one block per bond in the pool between the half-sweeps
  pool_cap_ = std::max<size_t>(pool_cap_, 2 * m.chains.size() * (size_t)(L_ + 1) + 16);
  m.chains_ok = true;
}

void Engine::ms_build_right_envs() {
  Multi& m = ms();
  for (int p = L_ - 1; p >= 1; --p) {
    bool need = false;
    for (auto& c : m.chains) need |= !c.Rok[p];
    if (!need) continue;
    for (int s = 0; s < m.S; ++s) {
      if (m.gauge[s][p] != MITDVP_GAUGE_B) throw ArgError("sites right of the centre must be in gauge B");
      transpose_rev3(st_, m.site[s][p].p, m.bt[s].p, m.dl[s][p], m.d[p], m.dr[s][p]);
    }
    for (auto& c : m.chains) {
      if (c.Rok[p]) continue;
      const MpoSite& w = (*c.w)[p];
      const zc* in = (p + 1 == L_) ? m.one.p : c.R[p + 1].p;
      if (p + 1 < L_ && !c.Rok[p + 1]) throw ArgError("internal: right environment chain broken");
      c.R[p] = pool_get((size_t)m.dl[c.i][p] * w.ml * m.dl[c.j][p]);
      env_update_rect(in, m.bt[c.j].p, m.bt[c.i].p, w.w2r.p, c.R[p].p, m.dr[c.i][p], m.dr[c.j][p], w.mr, m.d[p],
                      m.dl[c.i][p], m.dl[c.j][p], w.ml);
      c.Rok[p] = 1;
    }
  }
}

// exp_superH_propagation_direct on the stacked centre tensors (_mps_cls.py:1016-1100)
void Engine::ms_site_exp(int p, double dt) {
  Multi& m = ms();
  const int d = m.d[p];
  long nmax = 0;
  for (int s = 0; s < m.S; ++s) {
    const long e = (long)m.dl[s][p] * d * m.dr[s][p];
    m.off[s + 1] = m.off[s] + e;
    nmax = std::max(nmax, e);
    HIP_CHECK(hipMemcpyAsync(m.stack.p + m.off[s], m.site[s][p].p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  }
  const Multi::OpMs& o = m.ops.at(0);
  auto mv = [&](const zc* in, zc* out) {
    std::vector<char> started(m.S, 0);
    for (int i = 0; i < m.S; ++i) {  // diagonal scalar terms: the overlap blocks are the identity
      const hzc c = o.cj[(size_t)i * m.S + i];
      if (c == hzc(0, 0)) continue;
      const long e = m.off[i + 1] - m.off[i];
      HIP_CHECK(hipMemcpyAsync(out + m.off[i], in + m.off[i], e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
      vec_scale(st_, out + m.off[i], e, make_double2(c.real(), c.imag()));
      started[i] = 1;
    }
    for (auto& c : m.chains) {
      const MpoSite& w = (*c.w)[p];
      const zc* Lb = p == 0 ? m.one.p : c.L[p].p;
      const zc* Rb = p + 1 == L_ ? m.one.p : c.R[p + 1].p;
      const long e = m.off[c.i + 1] - m.off[c.i];
      const bool direct = !started[c.i] && c.f == hzc(1.0, 0.0);
      zc* dst = direct ? out + m.off[c.i] : m.acc.p;
      heff_apply_rect(Lb, w, Rb, in + m.off[c.j], dst, m.dl[c.i][p], m.dl[c.j][p], d, m.dr[c.i][p], m.dr[c.j][p]);
      if (!direct) {
        if (!started[c.i]) {
          HIP_CHECK(hipMemcpyAsync(out + m.off[c.i], m.acc.p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
          vec_scale(st_, out + m.off[c.i], e, make_double2(c.f.real(), c.f.imag()));
        } else {
          vec_axpby(st_, out + m.off[c.i], m.acc.p, e, make_double2(c.f.real(), c.f.imag()), make_double2(1.0, 0.0));
        }
      }
      started[c.i] = 1;
    }
    for (int i = 0; i < m.S; ++i)
      if (!started[i]) HIP_CHECK(hipMemsetAsync(out + m.off[i], 0, (m.off[i + 1] - m.off[i]) * sizeof(zc), st_));
  };
  // _iter_info sizes the Krylov space by the largest state tensor, not by the stack (_integrator.py:178-186)
  if (cfg.relax == 2)  // improved relaxation: lowest eigenvector of the stacked H_eff (_mps_cls.py:1078-1084;
    kprev_[p] = krylov_diag(mv, m.stack.p, m.off[m.S]);  // ndim = size of the stack, _integrator.py:98)
  else
    kprev_[p] = krylov_exp(scale_site(dt), mv, m.stack.p, m.off[m.S], kprev_[p], nmax);
  for (int s = 0; s < m.S; ++s)
    HIP_CHECK(hipMemcpyAsync(m.site[s][p].p, m.stack.p + m.off[s], (m.off[s + 1] - m.off[s]) * sizeof(zc),
                             hipMemcpyDeviceToDevice, st_));
  cnt_.n_exp_site += 1;
}

void Engine::ms_sweep(double dt, bool forward) {
  Multi& m = ms();
  const int begin = forward ? 0 : L_ - 1, end = forward ? L_ - 1 : 0;
  if (m.center != begin) throw ArgError("sweep must start at the centre (Psi) site");
  if (forward) ms_build_right_envs();
  const Multi::OpMs& o = m.ops.at(0);
  for (int p = begin; forward ? p <= end : p >= end; p += forward ? 1 : -1) {
    ms_site_exp(p, dt);
    if (p == end) break;
    const int d = m.d[p];
    // gauge move, one QR per state (trans_next_psite_AsigmaB, _mps_cls.py:1798-1850)
    std::vector<long> so(m.S + 1, 0);
    std::vector<int> bd(m.S);
    long smax = 0;
    for (int s = 0; s < m.S; ++s) {
      bd[s] = forward ? m.dr[s][p] : m.dl[s][p];
      so[s + 1] = so[s] + (long)bd[s] * bd[s];
      smax = std::max(smax, (long)bd[s] * bd[s]);
    }
    for (int s = 0; s < m.S; ++s) {
      const int dl = m.dl[s][p], dr = m.dr[s][p];
      if (forward) {
        timer_begin(3);
        long nl = 0;
        qr_thin(st_, m.site[s][p].p, dl * d, dr, m.spare.p, m.sigstack.p + so[s], qrwork_.p, &nl, qr_sync(), qr_hist_, qr_gauge_free_);
        timer_end();
        cnt_.n_launch += nl; cnt_.n_qr += 1;
        cnt_.qr_flops += 4.0 * (4.0 * (double)dl * d * dr * dr - 4.0 * (double)dr * dr * dr / 3.0);
        std::swap(m.site[s][p], m.spare);
        m.gauge[s][p] = MITDVP_GAUGE_A;
      } else {
        gauge_qr_right(m.site[s][p].p, dl, d, dr, m.spare.p, m.bt[s].p, m.sigstack.p + so[s]);
        std::swap(m.site[s][p], m.spare);
        m.gauge[s][p] = MITDVP_GAUGE_B;
      }
    }
    // renormalize_op_psite for every chain (_mps_mpo.py:421-696)
    for (auto& c : m.chains) {
      const MpoSite& w = (*c.w)[p];
      if (forward) {
        const zc* in = p == 0 ? m.one.p : c.L[p].p;
        c.L[p + 1] = pool_get((size_t)m.dr[c.i][p] * w.mr * m.dr[c.j][p]);
        env_update_rect(in, m.site[c.j][p].p, m.site[c.i][p].p, w.w2l.p, c.L[p + 1].p, m.dl[c.i][p], m.dl[c.j][p], w.ml,
                        d, m.dr[c.i][p], m.dr[c.j][p], w.mr);
        c.Lok[p + 1] = 1;
      } else {
        const zc* in = p + 1 == L_ ? m.one.p : c.R[p + 1].p;
        c.R[p] = pool_get((size_t)m.dl[c.i][p] * w.ml * m.dl[c.j][p]);
        env_update_rect(in, m.bt[c.j].p, m.bt[c.i].p, w.w2r.p, c.R[p].p, m.dr[c.i][p], m.dr[c.j][p], w.mr, d,
                        m.dl[c.i][p], m.dl[c.j][p], w.ml);
        c.Rok[p] = 1;
      }
    }
    // exp_superK_propagation_direct on the stacked bond matrices (_mps_cls.py:1102-1170)
    const int b = forward ? p + 1 : p;  // the bond the matrices live on
    auto mk = [&](const zc* in, zc* out) {
      std::vector<char> started(m.S, 0);
      for (int i = 0; i < m.S; ++i) {
        const hzc c = o.cj[(size_t)i * m.S + i];
        if (c == hzc(0, 0)) continue;
        const long e = so[i + 1] - so[i];
        HIP_CHECK(hipMemcpyAsync(out + so[i], in + so[i], e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
        vec_scale(st_, out + so[i], e, make_double2(c.real(), c.imag()));
        started[i] = 1;
      }
      for (auto& c : m.chains) {
        const int mb = forward ? (*c.w)[p].mr : (*c.w)[p].ml;
        const long e = so[c.i + 1] - so[c.i];
        const bool direct = !started[c.i] && c.f == hzc(1.0, 0.0);
        zc* dst = direct ? out + so[c.i] : m.acc.p;
        keff_apply_rect(c.L[b].p, c.R[b].p, in + so[c.j], dst, bd[c.i], bd[c.j], bd[c.i], bd[c.j], mb);
        if (!direct) {
          if (!started[c.i]) {
            HIP_CHECK(hipMemcpyAsync(out + so[c.i], m.acc.p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
            vec_scale(st_, out + so[c.i], e, make_double2(c.f.real(), c.f.imag()));
          } else {
            vec_axpby(st_, out + so[c.i], m.acc.p, e, make_double2(c.f.real(), c.f.imag()), make_double2(1.0, 0.0));
          }
        }
        started[c.i] = 1;
      }
      for (int i = 0; i < m.S; ++i)
        if (!started[i]) HIP_CHECK(hipMemsetAsync(out + so[i], 0, (so[i + 1] - so[i]) * sizeof(zc), st_));
    };
    if (cfg.relax != 2) {  // improved relaxation leaves the bond matrices alone (_mps_cls.py:1159-1160)
      kprev_[p] = krylov_exp(scale_bond(dt), mk, m.sigstack.p, so[m.S], kprev_[p], smax);
      cnt_.n_exp_bond += 1;
    }
    // the block on the other side of the bond is stale now; absorb the bond matrices
    for (auto& c : m.chains) {
      if (forward) { c.Rok[p + 1] = 0; pool_put(std::move(c.R[p + 1])); }
      else { c.Lok[p] = 0; pool_put(std::move(c.L[p])); }
    }
    const int q = forward ? p + 1 : p - 1;
    for (int s = 0; s < m.S; ++s) {
      if (forward) {  // Psi(p+1) = sigma . B(p+1)
        ZgemmDesc g = zgemm_desc(m.sigstack.p + so[s], m.site[s][q].p, m.spare.p, bd[s], m.d[q] * m.dr[s][q], bd[s]);
        zgemm(st_, g);
      } else {  // Psi(p-1) = A(p-1) . sigma
        ZgemmDesc g = zgemm_desc(m.site[s][q].p, m.sigstack.p + so[s], m.spare.p, m.dl[s][q] * m.d[q], bd[s], bd[s]);
        zgemm(st_, g);
      }
      cnt_.n_launch += 1;
      std::swap(m.site[s][q], m.spare);
      m.gauge[s][q] = MITDVP_GAUGE_PSI;
    }
    m.center = q;
  }
}

void Engine::ms_step(double dt) {
  Multi& m = ms();
  if (adaptive_) throw ArgError("adaptive bond dimension is not implemented for several electronic states");
  ms_require_ready();
  ms_build_chains();
  if (L_ == 1) {
    if (m.center != 0) throw ArgError("no centre site");
    ms_site_exp(0, dt);  // the forward and the backward half-sweep each propagate the only site
    ms_site_exp(0, dt);
    return;
  }
  ms_sweep(dt, true);
  ms_sweep(dt, false);
}

// Simulator.operate for several states (WFunc.apply_dipole, wavefunction.py:303-351;
// apply_dipole_along_sweep, _mps_cls.py:718-796; apply_superOp_direct, :2733-2778): fit
// phi ~ O|psi_0> / ||O|psi_0>|| in the bond dimensions of psi_0.  Every sweep replaces all states'
// site tensors by sigma_i = sum_j O_ij psi0_j with blocks of the pair (bra = phi_i, ket = psi0_j) and
// normalises by the norm of the stack; scalar terms run through overlap chains for every pair,
// i == j included (phi_i != psi0_i).
double Engine::ms_operate(int op_id, int maxstep, double conv_tol, int* iters_out) {
  Multi& m = ms();
  ms_require_ready();
  if (m.center != 0) throw ArgError("operate needs the centre at site 0");
  if (maxstep < 1) throw ArgError("operate: maxstep must be >= 1");
  auto it_op = m.ops.find(op_id);
  if (it_op == m.ops.end()) throw ArgError("operator not set");
  Multi::OpMs& o = it_op->second;
  const int S = m.S;
  struct Ch { int i, j; hzc f; const std::vector<MpoSite>* w; std::vector<DevBuf> L, R; };
  std::vector<Ch> ch;
  std::vector<char> fed(S, 0);
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < S; ++j) {
      const size_t ij = (size_t)i * S + j;
      if (o.has[ij]) {
        for (int p = 0; p < L_; ++p)
          if (!o.blk[ij][p].set || o.blk[ij][p].d != m.d[p]) throw ArgError("an MPO block is missing cores or has the wrong physical dimension");
        ch.push_back(Ch{i, j, hzc(1, 0), &o.blk[ij], {}, {}});
        fed[i] = 1;
      }
      if (o.cj[ij] != hzc(0, 0)) {
        ms_ident_cores();
        ch.push_back(Ch{i, j, o.cj[ij], &m.ident, {}, {}});
        fed[i] = 1;
      }
    }
  for (int i = 0; i < S; ++i)
    if (!fed[i]) throw ArgError("operate: every state needs at least one operator block or scalar term acting into it");
  const size_t cap = m.spare.n;
  std::vector<std::vector<DevBuf>> ket(S), prev(S);
  std::vector<DevBuf> kt(S);
  for (int s = 0; s < S; ++s) {
    ket[s].resize(L_); prev[s].resize(L_);
    kt[s] = pool_get(cap);
    for (int p = 0; p < L_; ++p) {
      const size_t e = (size_t)m.dl[s][p] * m.d[p] * m.dr[s][p];
      ket[s][p] = pool_get(cap);
      prev[s][p] = pool_get(e);
      HIP_CHECK(hipMemcpyAsync(ket[s][p].p, m.site[s][p].p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
    }
  }
  auto envsz = [&](const Ch& c, int b) {  // block on bond b (left of site b)
    const int mb = b == 0 ? 1 : (*c.w)[b - 1].mr;
    const int di = b == 0 ? 1 : m.dr[c.i][b - 1], dj = b == 0 ? 1 : m.dr[c.j][b - 1];
    return (size_t)di * mb * dj;
  };
  // right blocks of the initial bra / ket pair (bra = ket = psi_0 at this point)
  for (auto& c : ch) { c.L.resize(L_ + 1); c.R.resize(L_ + 1); }
  for (int p = L_ - 1; p >= 1; --p) {
    for (int s = 0; s < S; ++s) transpose_rev3(st_, ket[s][p].p, kt[s].p, m.dl[s][p], m.d[p], m.dr[s][p]);
    for (auto& c : ch) {
      const MpoSite& w = (*c.w)[p];
      const zc* in = p + 1 == L_ ? m.one.p : c.R[p + 1].p;
      c.R[p] = pool_get(envsz(c, p));
      env_update_rect(in, kt[c.j].p, kt[c.i].p, w.w2r.p, c.R[p].p, m.dr[c.i][p], m.dr[c.j][p], w.mr, m.d[p], m.dl[c.i][p],
                      m.dl[c.j][p], w.ml);
    }
  }
  double nrm = 0.0;
  auto apply_site = [&](int p) {  // apply_superOp_direct
    std::vector<char> started(S, 0);
    for (auto& c : ch) {
      const MpoSite& w = (*c.w)[p];
      const zc* Lb = p == 0 ? m.one.p : c.L[p].p;
      const zc* Rb = p + 1 == L_ ? m.one.p : c.R[p + 1].p;
      const long e = (long)m.dl[c.i][p] * m.d[p] * m.dr[c.i][p];
      const bool direct = !started[c.i] && c.f == hzc(1.0, 0.0);
      zc* dst = direct ? m.site[c.i][p].p : m.acc.p;
      heff_apply_rect(Lb, w, Rb, ket[c.j][p].p, dst, m.dl[c.i][p], m.dl[c.j][p], m.d[p], m.dr[c.i][p], m.dr[c.j][p]);
      if (!direct) {
        if (!started[c.i]) {
          HIP_CHECK(hipMemcpyAsync(m.site[c.i][p].p, m.acc.p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
          vec_scale(st_, m.site[c.i][p].p, e, make_double2(c.f.real(), c.f.imag()));
        } else {
          vec_axpby(st_, m.site[c.i][p].p, m.acc.p, e, make_double2(c.f.real(), c.f.imag()), make_double2(1.0, 0.0));
        }
      }
      started[c.i] = 1;
    }
    double s2 = 0;
    for (int s = 0; s < S; ++s) {
      const long e = (long)m.dl[s][p] * m.d[p] * m.dr[s][p];
      vec_sumsq(st_, m.site[s][p].p, e, reinterpret_cast<double*>(red_.p + RED_MISC));
      read_partials(RED_MISC, NPART / 2);
      const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
      for (int k = 0; k < NPART; ++k) s2 += hp[k];
    }
    nrm = std::sqrt(s2);
    if (!(nrm > 0.0)) throw ArgError("operate: the operator annihilates the state");
    for (int s = 0; s < S; ++s) {
      vec_scale(st_, m.site[s][p].p, (long)m.dl[s][p] * m.d[p] * m.dr[s][p], make_double2(1.0 / nrm, 0.0));
      m.gauge[s][p] = MITDVP_GAUGE_PSI;
    }
  };
  const zc one = make_double2(1.0, 0.0);
  int it = 0;
  for (it = 1; it <= maxstep; ++it) {
    for (int s = 0; s < S; ++s)
      for (int p = 0; p < L_; ++p)
        HIP_CHECK(hipMemcpyAsync(prev[s][p].p, m.site[s][p].p, (size_t)m.dl[s][p] * m.d[p] * m.dr[s][p] * sizeof(zc),
                                 hipMemcpyDeviceToDevice, st_));
    for (int p = 0; p < L_; ++p) {  // ->
      apply_site(p);
      if (p == L_ - 1) break;
      const int d = m.d[p];
      for (int s = 0; s < S; ++s) {
        const int l = m.dl[s][p], r = m.dr[s][p];
        gauge_qr_left(m.site[s][p].p, l, d, r, m.spare.p, sig_.p);  // phi: Psi -> A (sigma goes into a tensor replaced next)
        std::swap(m.site[s][p], m.spare);
        m.gauge[s][p] = MITDVP_GAUGE_A;
        gauge_qr_left(ket[s][p].p, l, d, r, m.spare.p, sig_.p);  // psi_0: Psi -> A sigma, sigma into the next site
        std::swap(ket[s][p], m.spare);
        ZgemmDesc g = zgemm_desc(sig_.p, ket[s][p + 1].p, m.spare.p, r, m.d[p + 1] * m.dr[s][p + 1], r);
        zgemm(st_, g);
        std::swap(ket[s][p + 1], m.spare);
      }
      for (auto& c : ch) {
        const MpoSite& w = (*c.w)[p];
        const zc* in = p == 0 ? m.one.p : c.L[p].p;
        pool_put(std::move(c.L[p + 1]));
        c.L[p + 1] = pool_get(envsz(c, p + 1));
        env_update_rect(in, ket[c.j][p].p, m.site[c.i][p].p, w.w2l.p, c.L[p + 1].p, m.dl[c.i][p], m.dl[c.j][p], w.ml, d,
                        m.dr[c.i][p], m.dr[c.j][p], w.mr);
      }
    }
    for (int p = L_ - 1; p >= 0; --p) {  // <-
      apply_site(p);
      if (p == 0) break;
      const int d = m.d[p];
      for (int s = 0; s < S; ++s) {
        const int l = m.dl[s][p], r = m.dr[s][p];
        gauge_qr_right(m.site[s][p].p, l, d, r, m.spare.p, m.bt[s].p, sig_.p);
        std::swap(m.site[s][p], m.spare);
        m.gauge[s][p] = MITDVP_GAUGE_B;
        gauge_qr_right(ket[s][p].p, l, d, r, m.spare.p, kt[s].p, sig_.p);
        std::swap(ket[s][p], m.spare);
        ZgemmDesc g = zgemm_desc(ket[s][p - 1].p, sig_.p, m.spare.p, m.dl[s][p - 1] * m.d[p - 1], l, l);
        zgemm(st_, g);
        std::swap(ket[s][p - 1], m.spare);
      }
      for (auto& c : ch) {
        const MpoSite& w = (*c.w)[p];
        const zc* in = p + 1 == L_ ? m.one.p : c.R[p + 1].p;
        pool_put(std::move(c.R[p]));
        c.R[p] = pool_get(envsz(c, p));
        env_update_rect(in, kt[c.j].p, m.bt[c.i].p, w.w2r.p, c.R[p].p, m.dr[c.i][p], m.dr[c.j][p], w.mr, d, m.dl[c.i][p],
                        m.dl[c.j][p], w.ml);
      }
    }
    // _is_converged (wavefunction.py:285-301): |1 - |sum_s <phi_s(i) | phi_s(i-1)>|| < conv_tol
    hzc ov(0, 0);
    for (int s = 0; s < S; ++s) {
      HIP_CHECK(hipMemcpyAsync(sig_.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
      zc* T = sig_.p;
      zc* Tn = sig2_.p;
      for (int p = 0; p < L_; ++p) {
        const int dl = m.dl[s][p], d = m.d[p], dr = m.dr[s][p];
        ZgemmDesc u = zgemm_desc(T, prev[s][p].p, tmp1_.p, dl, d * dr, dl);
        zgemm(st_, u);
        ZgemmDesc t = zgemm_desc(m.site[s][p].p, tmp1_.p, Tn, dr, dr, dl * d);
        t.transA = 1; t.conjA = 1; t.lda = dr;
        zgemm(st_, t);
        std::swap(T, Tn);
      }
      hzc v;
      HIP_CHECK(hipMemcpyAsync(&v, T, sizeof(zc), hipMemcpyDeviceToHost, st_));
      HIP_CHECK(hipStreamSynchronize(st_));
      ov += v;
    }
    if (std::fabs(1.0 - std::abs(ov)) < conv_tol) break;
    if (it == maxstep) break;
  }
  m.center = 0;
  m.chains_ok = false;  // the state changed: the Hamiltonian's blocks are rebuilt by the next step
  for (auto& c : ch) {
    for (auto& b : c.L) pool_put(std::move(b));
    for (auto& b : c.R) pool_put(std::move(b));
  }
  for (int s = 0; s < S; ++s) {
    for (auto& b : ket[s]) pool_put(std::move(b));
    for (auto& b : prev[s]) pool_put(std::move(b));
    pool_put(std::move(kt[s]));
  }
  if (iters_out) *iters_out = std::min(it, maxstep);
  return nrm;
}

// pop_states (_mps_cls.py:682-703): ||Psi_i(site 0)||^2
void Engine::ms_pops(double* out) {
  Multi& m = ms();
  ms_require_ready();
  if (m.center != 0) throw ArgError("populations need the centre at site 0 (psite = 0)");
  for (int s = 0; s < m.S; ++s) {
    const long n0 = (long)m.dl[s][0] * m.d[0] * m.dr[s][0];
    vec_sumsq(st_, m.site[s][0].p, n0, reinterpret_cast<double*>(red_.p + RED_MISC));
    read_partials(RED_MISC, NPART / 2);
    const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
    double t = 0;
    for (int i = 0; i < NPART; ++i) t += hp[i];
    out[s] = t;
  }
}

// sum_ij <Psi_i|O_ij|Psi_j> at site 0 with fresh right blocks (MPSCoef.expectation, _mps_cls.py:540-612)
hzc Engine::ms_expect(int op_id) {
  Multi& m = ms();
  ms_require_ready();
  if (m.center != 0) throw ArgError("expectation needs the centre at site 0 (psite = 0)");
  auto it = m.ops.find(op_id);
  if (it == m.ops.end()) throw ArgError("operator not set");
  Multi::OpMs& o = it->second;
  auto dot0 = [&](int i, const zc* y) {
    const long n0 = (long)m.dl[i][0] * m.d[0] * m.dr[i][0];
    vec_dot(st_, m.site[i][0].p, y, n0, true, red_.p + RED_MISC);
    read_partials(RED_MISC, NPART);
    double re = 0, im = 0;
    for (int k = 0; k < NPART; ++k) { re += h_red_[RED_MISC + k].x; im += h_red_[RED_MISC + k].y; }
    return hzc(re, im);
  };
  hzc tot(0, 0);
  for (int i = 0; i < m.S; ++i)
    if (o.cj[(size_t)i * m.S + i] != hzc(0, 0)) tot += o.cj[(size_t)i * m.S + i] * dot0(i, m.site[i][0].p);
  if (L_ > 1)
    for (int s = 0; s < m.S; ++s)
      for (int p = 1; p < L_; ++p)
        if (m.gauge[s][p] != MITDVP_GAUGE_B) throw ArgError("sites right of the centre must be in gauge B");
  size_t mx = 1;
  int mmax = 1;
  for (auto& b : o.blk)
    for (auto& w : b)
      if (w.set) mmax = std::max(mmax, std::max(w.ml, w.mr));
  for (int p = 0; p < L_; ++p) {
    int dm = 1;
    for (int s = 0; s < m.S; ++s) dm = std::max(dm, m.dl[s][p]);
    mx = std::max(mx, (size_t)dm * dm * mmax);
  }
  DevBuf ra = pool_get(mx), rb = pool_get(mx), bi = pool_get(m.spare.n), bj = pool_get(m.spare.n);
  auto chain = [&](int i, int j, hzc f, const std::vector<MpoSite>& w) {
    const zc* cur = m.one.p;
    for (int p = L_ - 1; p >= 1; --p) {
      transpose_rev3(st_, m.site[i][p].p, bi.p, m.dl[i][p], m.d[p], m.dr[i][p]);
      transpose_rev3(st_, m.site[j][p].p, bj.p, m.dl[j][p], m.d[p], m.dr[j][p]);
      env_update_rect(cur, bj.p, bi.p, w[p].w2r.p, ra.p, m.dr[i][p], m.dr[j][p], w[p].mr, m.d[p], m.dl[i][p], m.dl[j][p],
                      w[p].ml);
      cur = ra.p;
      std::swap(ra, rb);
    }
    heff_apply_rect(m.one.p, w[0], cur, m.site[j][0].p, m.acc.p, m.dl[i][0], m.dl[j][0], m.d[0], m.dr[i][0], m.dr[j][0]);
    tot += f * dot0(i, m.acc.p);
  };
  for (int i = 0; i < m.S; ++i)
    for (int j = 0; j < m.S; ++j) {
      const size_t ij = (size_t)i * m.S + j;
      if (o.has[ij]) {
        for (int p = 0; p < L_; ++p)
          if (!o.blk[ij][p].set || o.blk[ij][p].d != m.d[p]) throw ArgError("an MPO block is missing cores or has the wrong physical dimension");
        chain(i, j, hzc(1, 0), o.blk[ij]);
      }
      if (i != j && o.cj[ij] != hzc(0, 0)) {
        ms_ident_cores();
        chain(i, j, o.cj[ij], m.ident);
      }
    }
  pool_put(std::move(ra)); pool_put(std::move(rb)); pool_put(std::move(bi)); pool_put(std::move(bj));
  return tot;
}

// sum_i <Psi_i^*|Psi_i>: only the diagonal pairs carry an "auto" block (_mps_mpo.py:386-395)
hzc Engine::ms_autocorr() {
  Multi& m = ms();
  ms_require_ready();
  hzc tot(0, 0);
  const zc one = make_double2(1.0, 0.0);
  for (int s = 0; s < m.S; ++s) {
    HIP_CHECK(hipMemcpyAsync(sig_.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
    zc* T = sig_.p;
    zc* Tn = sig2_.p;
    for (int p = 0; p < L_; ++p) {
      const int dl = m.dl[s][p], d = m.d[p], dr = m.dr[s][p];
      ZgemmDesc u = zgemm_desc(T, m.site[s][p].p, tmp1_.p, dl, d * dr, dl);
      zgemm(st_, u);
      ZgemmDesc t = zgemm_desc(m.site[s][p].p, tmp1_.p, Tn, dr, dr, dl * d);
      t.transA = 1; t.lda = dr;
      zgemm(st_, t);
      std::swap(T, Tn);
    }
    hzc out;
    HIP_CHECK(hipMemcpyAsync(&out, T, sizeof(zc), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    tot += out;
  }
  return tot;
}

}  // namespace mitdvp
