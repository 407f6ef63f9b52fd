// engine_segment.hip -- the engine as ONE BLOCK of a site-range sharded chain (row (e) of SURVEY 8:
// real-space parallel one-site TDVP, /root/reference/pytdscf/_mps_parallel.py).
//
// A rank's block is an ordinary engine whose outer bonds are not 1: the environment blocks at its two
// ends come from the neighbouring ranks (set_boundary_env), and the half-sweep is driven step by step
// from the host side (pytdscf_amd/parallel_sites.py) through four primitives that are the pieces of
// propagate_along_sweep (_mps_cls.py:798-1014):
//   site_exp      exp(-i H_eff dt/2) on the centre tensor            exp_superH_propagation_direct :1016-1100
//   split_center  Psi -> A sigma | sigma B (QR) + environment update trans_next_psite_AsigmaB :1798-1850
//   bond_exp      exp(+i K_eff dt/2) on the bond matrix              exp_superK_propagation_direct :1102-1170
//   absorb_bond   sigma into the neighbouring site                   trans_next_psite_APsiB :1172-1206
// The joint update of two blocks' facing sites (propagate_joint_two_sites, _mps_parallel.py:270-470) is
// the same four on a two-site engine whose boundary blocks are the two ranks' environments.
#include "engine_internal.h"
#include "engine_krylov.inc"

namespace mitdvp {

// new values for a site tensor of unchanged shape; unlike set_site the environment cache survives (the
// caller knows which blocks the new tensor invalidates: none, when a junction site is replaced by its
// re-gauged twin and the blocks through it are replaced as well)
void Engine::replace_site(int i, const double* reim, int gauge) {
  if (i < 0 || i >= L_ || !site_[i].p) throw ArgError("replace_site: bad or unset site");
  const size_t e = (size_t)dl_[i] * dd_[i] * dr_[i];
  copy_in(site_[i].p, reim, e);
  gauge_[i] = gauge;
  if (gauge == MITDVP_GAUGE_PSI) center_ = i;
  else if (center_ == i) center_ = -1;
}

// new values AND a new shape for a site tensor while the environment cache survives: the junction site of a block
// whose junction bond has grown (adaptive ranks across junctions); the caller replaces the boundary block of that
// side as well, every other cached block is built from tensors this one does not enter
void Engine::reshape_site(int i, const double* reim, int l, int n, int r, int gauge) {
  if (i < 0 || i >= L_ || !site_[i].p) throw ArgError("reshape_site: bad or unset site");
  if (l < 1 || r < 1 || n != dd_[i]) throw ArgError("reshape_site: bad shape");
  if (i != 0 && i != L_ - 1) throw ArgError("reshape_site: only a block's first or last site faces a junction");
  if ((i > 0 && l != dl_[i]) || (i < L_ - 1 && r != dr_[i])) throw ArgError("reshape_site: only the outer bond may change");
  const size_t e = (size_t)l * n * r;
  site_[i].reserve(e);
  copy_in(site_[i].p, reim, e);
  dl_[i] = l; dr_[i] = r; gauge_[i] = gauge;
  if (gauge == MITDVP_GAUGE_PSI) center_ = i;
  else if (center_ == i) center_ = -1;
}

// side 0: block left of site 0, L[a][c][b] (d, m, d); side 1: block right of the last site, R[r][t][s]
void Engine::set_boundary_env(int side, const double* reim, int d, int m) {
  if (side != 0 && side != 1) throw ArgError("set_boundary_env: side must be 0 (left) or 1 (right)");
  if (d < 1 || m < 1) throw ArgError("set_boundary_env: bad shape");
  DevBuf& b = side == 0 ? envL_[0] : envR_[L_];
  const size_t e = (size_t)d * m * d;
  b.reserve(e);
  copy_in(b.p, reim, e);
  (side == 0 ? bnd_dl_ : bnd_dr_) = d;
  (side == 0 ? bnd_ml_ : bnd_mr_) = m;
  (side == 0 ? envL_ok_[0] : envR_ok_[L_]) = 1;
  segment_ = true;
}

void Engine::env_shape(int side, int bond, int* d, int* m) {
  if (bond < 0 || bond > L_) throw ArgError("env_shape: bad bond");
  if (side == 0) {
    if (bond == 0) { *d = segment_ ? bnd_dl_ : 1; *m = segment_ ? bnd_ml_ : 1; }
    else { *d = dr_[bond - 1]; *m = mpo(0, bond - 1).mr; }
  } else {
    if (bond == L_) { *d = segment_ ? bnd_dr_ : 1; *m = segment_ ? bnd_mr_ : 1; }
    else { *d = dl_[bond]; *m = mpo(0, bond).ml; }
  }
}

void Engine::get_env(int side, int bond, double* out) {
  int d = 0, m = 0;
  env_shape(side, bond, &d, &m);
  const bool ok = side == 0 ? envL_ok_[bond] : envR_ok_[bond];
  const DevBuf& b = side == 0 ? envL_[bond] : envR_[bond];
  if (!ok || !b.p) throw ArgError("get_env: this environment block is not built");
  copy_out(out, b.p, (size_t)d * m * d);
}

// environment blocks of all sites on one side of the centre (construct_op_sites, _mps_cls.py:1738-1796)
void Engine::build_envs(int side) {
  require_ready();
  if (side == 0) build_left_envs();
  else build_right_envs();
}

void Engine::site_exp(double dt) {
  require_ready();
  if (center_ < 0) throw ArgError("site_exp: no centre site");
  const int p = center_;
  if (!envL_ok_[p] || !envR_ok_[p + 1]) throw ArgError("site_exp: the environment blocks around the centre are not built");
  local_site_exp(p, dt);
  ss_check();
}

// One H_eff apply at the centre site EXACTLY as a local exponential issues it (local_site_exp): the identity checks of
// both environment blocks decide the trimmed S1 / S3 forms, the MPO's zero blocks the list kernel of the W stage.
// in == nullptr: the centre tensor itself.  flags: bit 0 S1 trimmed, bit 1 S3 trimmed, bit 2 block-sparse W stage,
// bit 3 the one-launch small-bond kernel took the apply.
void Engine::heff_apply_center(const double* in, double* out, int* flags) {
  require_ready();
  if (center_ < 0) throw ArgError("heff_apply_center: no centre site");
  const int p = center_;
  if (!envL_ok_[p] || !envR_ok_[p + 1]) throw ArgError("heff_apply_center: the environment blocks around the centre are not built");
  const MpoSite& w = mpo(0, p);
  if (dd_[p] != w.d) throw ArgError("MPO physical dimension differs from the site tensor's");
  const int dl = dl_[p], d = dd_[p], dr = dr_[p];
  const size_t n = (size_t)dl * d * dr;
  const zc* Lb = envL_[p].p;
  const zc* Rb = envR_[p + 1].p;
  DevBuf x = pool_get(n), y = pool_get(n);
  if (in) copy_in(x.p, in, n);
  else HIP_CHECK(hipMemcpyAsync(x.p, site_[p].p, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  choose_apply_forms(Lb, w, Rb, dl, d, dr);
  struct Reset { bool& f; bool& g; bool& e; ~Reset() { f = false; g = false; e = false; } } reset{trim_r_, trim_l_, edge_};
  SmallChain sc;
  const bool small = small_ok() && chain_heff(sc, Lb, w, Rb, dl, d, dr, false);
  const bool edge = edge_ && !small;
  const bool sparse = !small && !edge && sparse_w_ && dr >= 64 && w.kl_l.p && w.sp_frac_l <= 0.6;
  if (flags)
    *flags = (trim_l_ && !small && !edge ? 1 : 0) | (trim_r_ && !small && !edge ? 2 : 0) | (sparse ? 4 : 0) | (small ? 8 : 0) |
             (edge ? 16 : 0);
  heff_apply(Lb, w, Rb, x.p, y.p, dl, d, dr, op(0).shift);
  copy_out(out, y.p, n);
  pool_put(std::move(x)); pool_put(std::move(y));
}

// forward: site p <- A, sigma (dr x dr) kept, L_{p+1} built; backward: site p <- B, sigma (dl x dl), R_p built
void Engine::split_center(bool forward) {
  require_ready();
  if (center_ < 0) throw ArgError("split_center: no centre site");
  const int p = center_;
  const MpoSite& w = mpo(0, p);
  const int dl = dl_[p], d = dd_[p], dr = dr_[p];
  DevBuf spare = pool_get(V_.n / MAXK);
  if (forward) {
    if (!envL_ok_[p]) throw ArgError("split_center: left environment missing");
    timer_begin(3);
    long nl = 0;
    qr_thin(st_, site_[p].p, dl * d, dr, spare.p, sig_.p, qrwork_.p, &nl, qr_sync(), qr_hist_, qr_gauge_free_);
    timer_end();
    cnt_.n_launch += nl; cnt_.n_qr += 1;
    std::swap(site_[p], spare);
    gauge_[p] = MITDVP_GAUGE_A;
    pool_put(std::move(envL_[p + 1]));
    envL_[p + 1] = pool_get((size_t)dr * w.mr * dr);
    env_update(envL_[p].p, site_[p].p, w.w2l.p, envL_[p + 1].p, dl, w.ml, d, dr, w.mr, w.w2el.p, &w, 0);
    envL_ok_[p + 1] = 1;
    bond_ = p + 1; bond_dim_ = dr;
  } else {
    if (!envR_ok_[p + 1]) throw ArgError("split_center: right environment missing");
    gauge_qr_right(site_[p].p, dl, d, dr, spare.p, tmp2_.p, sig_.p);
    std::swap(site_[p], spare);
    gauge_[p] = MITDVP_GAUGE_B;
    pool_put(std::move(envR_[p]));
    envR_[p] = pool_get((size_t)dl * w.ml * dl);
    env_update(envR_[p + 1].p, tmp2_.p, w.w2r.p, envR_[p].p, dr, w.mr, d, dl, w.ml, w.w2er.p, &w, 1);
    envR_ok_[p] = 1;
    bond_ = p; bond_dim_ = dl;
  }
  pool_put(std::move(spare));
  center_ = -1;
  bond_site_ = p;
}

// exp(+i K_eff dt/2) on the pending bond matrix; environments on both sides of that bond must exist
void Engine::bond_exp(double dt) {
  require_ready();
  if (bond_ < 0) throw ArgError("bond_exp: no pending bond matrix (call split_center first)");
  const int b = bond_;
  if (!envL_ok_[b] || !envR_ok_[b]) throw ArgError("bond_exp: the environment blocks around the bond are not built");
  int d = 0, m = 0;
  env_shape(0, b, &d, &m);
  if (d != bond_dim_) throw ArgError("bond_exp: environment and bond matrix disagree");
  const zc* Lb = envL_[b].p;
  const zc* Rb = envR_[b].p;
  const hzc shift = op(0).shift;
  const int p = bond_site_;  // Krylov memory of the site the bond matrix came from (_Debug.niter_krylov[site_now])
  if (!small_bond_exp(p, Lb, Rb, d, m, dt)) {
    auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, d, d, m, shift); };
    kprev_[p] = krylov_exp(scale_bond(dt), mk, sig_.p, (long)d * d, kprev_[p]);
    cnt_.n_exp_bond += 1;
  }
  ss_check();
}

// the pending bond matrix goes into the site right (forward) or left (backward) of the bond
void Engine::absorb_bond(bool forward) {
  require_ready();  // sizes the workspaces the spare tensor comes from
  if (bond_ < 0) throw ArgError("absorb_bond: no pending bond matrix");
  const int b = bond_, D = bond_dim_;
  DevBuf spare = pool_get(V_.n / MAXK);
  if (forward) {
    if (b >= L_) throw ArgError("absorb_bond: no site right of the last bond (take the matrix with get_bond)");
    if (dl_[b] != D) throw ArgError("absorb_bond: bond dimension mismatch");
    ZgemmDesc g = zgemm_desc(sig_.p, site_[b].p, spare.p, D, dd_[b] * dr_[b], D);
    zgemm(st_, g);
    std::swap(site_[b], spare);
    gauge_[b] = MITDVP_GAUGE_PSI;
    center_ = b;
  } else {
    if (b < 1) throw ArgError("absorb_bond: no site left of the first bond (take the matrix with get_bond)");
    if (dr_[b - 1] != D) throw ArgError("absorb_bond: bond dimension mismatch");
    ZgemmDesc g = zgemm_desc(site_[b - 1].p, sig_.p, spare.p, dl_[b - 1] * dd_[b - 1], D, D);
    zgemm(st_, g);
    std::swap(site_[b - 1], spare);
    gauge_[b - 1] = MITDVP_GAUGE_PSI;
    center_ = b - 1;
  }
  cnt_.n_launch += 1;
  pool_put(std::move(spare));
  bond_ = -1;
}

void Engine::get_bond(double* out, int* dim) {
  if (bond_ < 0) throw ArgError("get_bond: no pending bond matrix");
  *dim = bond_dim_;
  if (out) {
    copy_out(out, sig_.p, (size_t)bond_dim_ * bond_dim_);
  }
}

// x (dim x dim) becomes the pending bond matrix of bond b (the bond left of site b)
void Engine::set_bond(int b, const double* reim, int dim) {
  if (b < 0 || b > L_) throw ArgError("set_bond: bad bond");
  if (dim < 1) throw ArgError("set_bond: bad dimension");
  require_ready();  // workspaces first: growing sig_ afterwards would drop the matrix
  if (!((b < L_ && dl_[b] == dim) || (b > 0 && dr_[b - 1] == dim))) throw ArgError("set_bond: dimension matches neither neighbouring site");
  sig_.reserve((size_t)dim * dim);
  copy_in(sig_.p, reim, (size_t)dim * dim);
  bond_ = b; bond_dim_ = dim;
  bond_site_ = std::min(std::max(b - 1, 0), L_ - 1);
  center_ = -1;
}

// A boundary block carried through ALL sites of this engine: what the ranks of a site-sharded state pass along
// to evaluate <Psi|Psi>, <Psi*|Psi> and <Psi|O|Psi> without gathering the tensors (MPSCoefParallel.ovlp /
// expectation, _mps_parallel.py:855-983, :1210-1302; _ovlp_single_state_np_from_left / _from_right, :1473-1518).
//   op_id >= 0: the block is an environment block (d, m, d) of that operator, the sites enter as conj(bra) | ket
//               (contract_with_site_mpo, _contraction.py:148-397); the gauge tags are not consulted;
//   op_id <  0: plain transfer block T[bra][ket] (m = 1); conj_bra = false gives <Psi*|Psi> (autocorrelation).
// from_left: `in` sits left of the first site of the range and `out` right of its last site; otherwise the other way
// round.  The range is [first0, first0 + count) (default: all sites; count = 0 copies the block).
void Engine::fold_block(int op_id, bool conj_bra, bool from_left, const double* in, int d, int m, double* out, int first0,
                        int count) {
  require_ready(true);
  if (count < 0) count = L_ - first0;
  if (first0 < 0 || count < 0 || first0 + count > L_) throw ArgError("fold_block: bad site range");
  if (!in || !out) throw ArgError("fold_block: null block");
  const int first = count > 0 ? (from_left ? first0 : first0 + count - 1) : std::min(first0, L_ - 1);
  const int d0 = count > 0 ? (from_left ? dl_[first] : dr_[first]) : d;
  if (d != d0) throw ArgError("fold_block: block and site bond dimension differ");
  if (op_id < 0 && m != 1) throw ArgError("fold_block: a plain transfer block has m = 1");
  if (op_id >= 0) {
    if (!conj_bra) throw ArgError("fold_block: operator blocks are defined with the conjugated bra only");
    if (count > 0) {
      const MpoSite& w = mpo(op_id, first);
      if (m != (from_left ? w.ml : w.mr)) throw ArgError("fold_block: block and MPO bond dimension differ");
    }
  }
  size_t mx = (size_t)d * m * d;
  for (int p = first0; p < first0 + count; ++p) {
    const int mm = op_id >= 0 ? std::max(mpo(op_id, p).ml, mpo(op_id, p).mr) : 1;
    const size_t dd = std::max(dl_[p], dr_[p]);
    mx = std::max(mx, dd * mm * dd);
  }
  DevBuf cur = pool_get(mx), nxt = pool_get(mx);
  copy_in(cur.p, in, (size_t)d * m * d);
  int dout = d, mout = m;
  for (int k = 0; k < count; ++k) {
    const int p = from_left ? first0 + k : first0 + count - 1 - k;
    const int dl = dl_[p], dd = dd_[p], dr = dr_[p];
    if (op_id >= 0) {
      const MpoSite& w = mpo(op_id, p);
      if (w.d != dd) throw ArgError("fold_block: MPO physical dimension differs from the site tensor's");
      if (from_left) {
        env_update(cur.p, site_[p].p, w.w2l.p, nxt.p, dl, w.ml, dd, dr, w.mr, w.w2el.p, &w, 0);
        dout = dr; mout = w.mr;
      } else {
        transpose_rev3(st_, site_[p].p, tmp1_.p, dl, dd, dr);
        env_update(cur.p, tmp1_.p, w.w2r.p, nxt.p, dr, w.mr, dd, dl, w.ml, w.w2er.p, &w, 1);
        dout = dl; mout = w.ml;
      }
      cnt_.n_launch += 3;
    } else if (from_left) {
      // U[a][(j,s)] = T[a][b] C[b][(j,s)];  T'[i][s] = sum_(a,j) op(C)[(a,j)][i] U[(a,j)][s]
      ZgemmDesc u = zgemm_desc(cur.p, site_[p].p, tmp1_.p, dl, dd * dr, dl);
      zgemm(st_, u);
      ZgemmDesc t = zgemm_desc(site_[p].p, tmp1_.p, nxt.p, dr, dr, dl * dd);
      t.transA = 1; t.conjA = conj_bra ? 1 : 0; t.lda = dr;
      zgemm(st_, t);
      dout = dr; cnt_.n_launch += 2;
    } else {
      // U[(b,j)][r] = C[(b,j)][s] T[r][s];  T'[a][b] = sum_(j,r) op(C)[a][(j,r)] U[b][(j,r)]
      ZgemmDesc u = zgemm_desc(site_[p].p, cur.p, tmp1_.p, dl * dd, dr, dr);
      u.transB = 1; u.ldb = dr;
      zgemm(st_, u);
      ZgemmDesc t = zgemm_desc(site_[p].p, tmp1_.p, nxt.p, dl, dl, dd * dr);
      t.conjA = conj_bra ? 1 : 0; t.transB = 1; t.ldb = dd * dr;
      zgemm(st_, t);
      dout = dl; cnt_.n_launch += 2;
    }
    std::swap(cur, nxt);
  }
  copy_out(out, cur.p, (size_t)dout * mout * dout);
  pool_put(std::move(cur));
  pool_put(std::move(nxt));
}

// rho[j][j'] = <j| Tr_rest |Psi><Psi| |j'> at one site of a state whose parts left and right of the site are given as
// transfer blocks in the fold_block convention: TL[bra][ket] (dl x dl), TR[bra][ket] (dr x dr).  Reduced densities of
// a site-sharded state (MPSCoefParallel.get_reduced_densities, _mps_parallel.py:1035-1208): the blocks arrive from the
// neighbouring ranks.  `out` (d x d) is host memory.
void Engine::site_rdm_blocks(int isite, const double* TL, const double* TR, double* out) {
  require_ready(true);
  if (isite < 0 || isite >= L_) throw ArgError("site_rdm_blocks: bad site index");
  if (!TL || !TR || !out) throw ArgError("site_rdm_blocks: null argument");
  const int dl = dl_[isite], d = dd_[isite], dr = dr_[isite];
  DevBuf tl = pool_get((size_t)dl * dl), tr = pool_get((size_t)dr * dr), rho = pool_get((size_t)dl * d * d);
  copy_in(tl.p, TL, (size_t)dl * dl);
  copy_in(tr.p, TR, (size_t)dr * dr);
  // U[a'][(j,s)] = sum_a TL[a'][a] C[a][(j,s)]
  ZgemmDesc u = zgemm_desc(tl.p, site_[isite].p, tmp1_.p, dl, d * dr, dl);
  zgemm(st_, u);
  // V[(a',j)][s'] = sum_s U[(a',j)][s] TR[s'][s]
  ZgemmDesc v = zgemm_desc(tmp1_.p, tr.p, tmp2_.p, dl * d, dr, dr);
  v.transB = 1; v.ldb = dr;
  zgemm(st_, v);
  // rho_a'[j][j'] = sum_s' V[a'][j][s'] conj(C[a'][j'][s']); summed over a' on the host (d x d numbers per a')
  ZgemmDesc r = zgemm_desc(tmp2_.p, site_[isite].p, rho.p, d, d, dr);
  r.transB = 1; r.conjB = 1; r.ldb = dr; r.ldc = d;
  r.batch = dl; r.strideA = (long)d * dr; r.strideB = (long)d * dr; r.strideC = (long)d * d;
  if (dl > 65535) throw ArgError("site_rdm_blocks: bond dimension above 65535");
  zgemm(st_, r);
  cnt_.n_launch += 3;
  std::vector<hzc> h((size_t)dl * d * d);
  HIP_CHECK(hipMemcpyAsync(h.data(), rho.p, h.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  hzc* o = reinterpret_cast<hzc*>(out);
  for (int e = 0; e < d * d; ++e) o[e] = hzc(0, 0);
  for (int a = 0; a < dl; ++a)
    for (int e = 0; e < d * d; ++e) o[e] += h[(size_t)a * d * d + e];
  pool_put(std::move(tl)); pool_put(std::move(tr)); pool_put(std::move(rho));
}

}  // namespace mitdvp
