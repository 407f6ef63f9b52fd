// engine_ops.hip -- operations on the device-resident MPS beside the plain sweep:
// one-site gates and Kraus maps between the half-sweeps, adaptive bond dimension (a1TDVP).
#include "engine_internal.h"
#include "engine_krylov.inc"

namespace mitdvp {

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
  for (auto& kv : gates_)  // validate everything before the state is touched
    if (kv.second.d != dd_[kv.first]) throw ArgError("gate dimension differs from the site's physical dimension");
  DevBuf spare = pool_get(V_.n / MAXK);
  int lo = L_, hi = -1;
  for (auto& kv : gates_) {
    const int p = kv.first;
    const Gate& g = kv.second;
    const int l = dl_[p], d = dd_[p], r = dr_[p];
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
  for (auto& kv : kraus_) {  // validate everything before the state is touched
    const KrausOp& op = kv.second;
    if (!op.two_site && dd_[kv.first] % op.d != 0) throw ArgError("Kraus contract: dK must be divisible by d");
    if (op.two_site && dd_[kv.first] != op.d)
      throw ArgError("two-site Kraus map: the system site's dimension differs from the Kraus operators'");
  }
  DevBuf spare = pool_get(V_.n / MAXK);
  int lo = L_, hi = -1;
  for (auto& kv : kraus_) {
    const int p = kv.first;
    const KrausOp& op = kv.second;
    if (!op.two_site) {
      const int l = dl_[p], dim = dd_[p], r = dr_[p];
      kraus_core(site_[p].p, l, op.d, dim / op.d, r, op, spare.p);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_C;
      lo = std::min(lo, p); hi = std::max(hi, p);
      continue;
    }
    const int q = p + 1;
    const int m = dl_[p], d = dd_[p], l = dr_[p], K = dd_[q], n = dr_[q];
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
    lp[0] = dl_[0]; rp[L_] = dr_[L_ - 1];  // 1 for a whole chain; a segment's outer bonds are its neighbours' (fixed during a sweep)
    for (int b = 1; b <= L_; ++b) lp[b] = std::min(1e15, lp[b - 1] * dd_[b - 1]);
    for (int b = L_ - 1; b >= 0; --b) rp[b] = std::min(1e15, rp[b + 1] * dd_[b]);
    for (int b = 0; b <= L_; ++b) cap[b] = (long)std::min(lp[b], rp[b]);
  }
  auto bound = [&](int b) -> long {  // bond left of site b, widened tensors included
    if (b == 0) return dl_[0];
    if (b == L_) return dr_[L_ - 1];
    return std::min<long>(cap[b], std::max<long>(dl_[b], ad_dmax_) + ad_dd_);
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
  qr_householder(st_, tmp1_.p, l * c, r, out, nullptr, qrwork_.p, &nl, e, qr_sync(), qr_hist_);
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
    qr_householder(st_, tmp1_.p, m, l, tmp2_.p, nullptr, qrwork_.p, &nl, e, qr_sync(), qr_hist_);
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
    // A bond already at Dmax cannot grow (is_max_rank at the neighbouring site, whose own bond
    // towards q does not change before it is processed): its widened tensor is never read.
    // Once every bond is saturated an adaptive sweep costs what a plain one does.
    if ((forward ? l1 : r1) >= ad_dmax_) { fdl_[q] = l1; fdr_[q] = r1; continue; }
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
      if (has_shift) {  // coupleJ * ovlp: the overlap blocks <widened|thin> are [1; 0] embeddings
        copy2d(st_, hl.p, dmax, site_[p].p, r, (long)l * c, r, 0, zshift, true);
        vec_axpby(st_, hr.p, psip.p, (long)r * cq * rq, zshift, one);
        copy2d(st_, ks.p, dmax, sig_.p, r, r, r, 0, zshift, true);
      }
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
    if (ad_site_hook_) ad_site_hook_();
    // from here on the plain step at the new rank
    timer_begin(3);
    qr_householder(st_, site_[p].p, l * c, newD, spare.p, sig_.p, qrwork_.p, &nl, 0, qr_sync(), qr_hist_);
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
    if (has_shift) {
      copy2d(st_, hl.p, dmax, psip.p, l, (long)lq * cq, l, 0, zshift, true);
      vec_axpby(st_, hr.p, site_[p].p, (long)l * c * r, zshift, one);
      copy2d(st_, ks.p, dmax, sig_.p, l, l, l, 0, zshift, true);
    }
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
  if (ad_site_hook_) ad_site_hook_();
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

// ---------------------------------------------------------------------------
// Simulator.operate: fit phi ~ O|psi_0> / ||O|psi_0>|| in the bond dimensions of psi_0
// (WFunc.apply_dipole, wavefunction.py:303-351; MPSCoef.apply_dipole /
// apply_dipole_along_sweep / apply_superOp_direct, _mps_cls.py:421-450, :718-796,
// :2733-2778).  Every site tensor of phi is replaced by the mixed-environment apply
// (bra = phi, ket = psi_0); both states move their centre together; the mixed blocks are
// environment updates with different bra and ket tensors.  On return the engine's state
// is phi (site-0 centred, normalised); the norm of the last apply is returned.
// ---------------------------------------------------------------------------
double Engine::operate(int op_id, int maxstep, double conv_tol, int* iters_out) {
  require_ready();
  if (center_ != 0) throw ArgError("operate needs the centre at site 0");
  if (maxstep < 1) throw ArgError("operate: maxstep must be >= 1");
  Operator& o = op(op_id);
  const hzc shift = o.shift;
  const zc one = make_double2(1.0, 0.0);
  const size_t cap = V_.n / MAXK;
  std::vector<DevBuf> ket(L_), prev(L_), mixL(L_ + 1), mixR(L_ + 1), ovL(L_ + 1), ovR(L_ + 1);
  // The scalar term coupleJ * ovlp (_contraction.py:1200-1216) goes through the OVERLAP blocks of
  // the bra / ket pair, which are not identities here (phi != psi_0): it is carried as a second
  // operator chain whose cores are identities (bond dimension 1).
  const bool has_shift = shift != hzc(0.0, 0.0);
  std::map<int, MpoSite> idw;
  auto ident = [&](int d) -> const MpoSite& {
    MpoSite& w = idw[d];
    if (!w.set) {
      w.ml = 1; w.mr = 1; w.d = d;
      w.w2l.reserve((size_t)d * d); w.w2r.reserve((size_t)d * d);
      set_identity(st_, w.w2l.p, d, d, d);
      set_identity(st_, w.w2r.p, d, d, d);
      w.set = true;
    }
    return w;
  };
  for (int p = 0; p < L_; ++p) {
    const size_t e = (size_t)dl_[p] * dd_[p] * dr_[p];
    ket[p] = pool_get(cap);
    prev[p] = pool_get(e);
    HIP_CHECK(hipMemcpyAsync(ket[p].p, site_[p].p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  }
  mixL[0] = pool_get(1); mixR[L_] = pool_get(1); ovL[0] = pool_get(1); ovR[L_] = pool_get(1);
  for (DevBuf* b : {&mixL[0], &mixR[L_], &ovL[0], &ovR[L_]})
    HIP_CHECK(hipMemcpyAsync(b->p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  DevBuf spare = pool_get(cap), kt = pool_get(cap), bt = pool_get(cap);
  // right blocks of the initial bra / ket pair (construct_op_sites with superblock_states_ket)
  for (int p = L_ - 1; p >= 1; --p) {
    const MpoSite& w = mpo(op_id, p);
    transpose_rev3(st_, ket[p].p, kt.p, dl_[p], dd_[p], dr_[p]);
    transpose_rev3(st_, site_[p].p, bt.p, dl_[p], dd_[p], dr_[p]);
    mixR[p] = pool_get((size_t)dl_[p] * w.ml * dl_[p]);
    env_update_rect(mixR[p + 1].p, kt.p, bt.p, w.w2r.p, mixR[p].p, dr_[p], dr_[p], w.mr, dd_[p], dl_[p], dl_[p], w.ml);
    if (has_shift) {
      ovR[p] = pool_get((size_t)dl_[p] * dl_[p]);
      env_update_rect(ovR[p + 1].p, kt.p, bt.p, ident(dd_[p]).w2r.p, ovR[p].p, dr_[p], dr_[p], 1, dd_[p], dl_[p], dl_[p], 1);
    }
  }
  double nrm = 0.0;
  auto apply_site = [&](int p) {  // apply_superOp_direct
    const MpoSite& w = mpo(op_id, p);
    if (dd_[p] != w.d) throw ArgError("MPO physical dimension differs from the site tensor's");
    heff_apply(mixL[p].p, w, mixR[p + 1].p, ket[p].p, site_[p].p, dl_[p], dd_[p], dr_[p], hzc(0.0, 0.0));
    const long n = (long)dl_[p] * dd_[p] * dr_[p];
    if (has_shift) {
      heff_apply_rect(ovL[p].p, ident(dd_[p]), ovR[p + 1].p, ket[p].p, spare.p, dl_[p], dl_[p], dd_[p], dr_[p], dr_[p]);
      vec_axpby(st_, site_[p].p, spare.p, n, make_double2(shift.real(), shift.imag()), make_double2(1.0, 0.0));
    }
    vec_sumsq(st_, site_[p].p, n, reinterpret_cast<double*>(red_.p + RED_MISC));
    read_partials(RED_MISC, NPART / 2);
    const double* hp = reinterpret_cast<const double*>(h_red_ + RED_MISC);
    double s2 = 0;
    for (int i = 0; i < NPART; ++i) s2 += hp[i];
    nrm = std::sqrt(s2);
    if (!(nrm > 0.0)) throw ArgError("operate: the operator annihilates the state");
    vec_scale(st_, site_[p].p, n, make_double2(1.0 / nrm, 0.0));
    gauge_[p] = MITDVP_GAUGE_PSI;
  };
  int it = 0;
  for (it = 1; it <= maxstep; ++it) {
    for (int p = 0; p < L_; ++p)
      HIP_CHECK(hipMemcpyAsync(prev[p].p, site_[p].p, (size_t)dl_[p] * dd_[p] * dr_[p] * sizeof(zc), hipMemcpyDeviceToDevice, st_));
    for (int p = 0; p < L_; ++p) {  // ->
      apply_site(p);
      if (p == L_ - 1) break;
      const MpoSite& w = mpo(op_id, p);
      const int l = dl_[p], d = dd_[p], r = dr_[p];
      gauge_qr_left(site_[p].p, l, d, r, spare.p, sig_.p);  // phi: Psi -> A (its sigma goes into a tensor that is replaced next)
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_A;
      gauge_qr_left(ket[p].p, l, d, r, spare.p, sig_.p);    // psi_0: Psi -> A sigma, sigma into the next site
      std::swap(ket[p], spare);
      ZgemmDesc g = zgemm_desc(sig_.p, ket[p + 1].p, spare.p, r, dd_[p + 1] * dr_[p + 1], r);
      zgemm(st_, g);
      std::swap(ket[p + 1], spare);
      pool_put(std::move(mixL[p + 1]));
      mixL[p + 1] = pool_get((size_t)r * w.mr * r);
      env_update_rect(mixL[p].p, ket[p].p, site_[p].p, w.w2l.p, mixL[p + 1].p, l, l, w.ml, d, r, r, w.mr);
      if (has_shift) {
        pool_put(std::move(ovL[p + 1]));
        ovL[p + 1] = pool_get((size_t)r * r);
        env_update_rect(ovL[p].p, ket[p].p, site_[p].p, ident(d).w2l.p, ovL[p + 1].p, l, l, 1, d, r, r, 1);
      }
    }
    for (int p = L_ - 1; p >= 0; --p) {  // <-
      apply_site(p);
      if (p == 0) break;
      const MpoSite& w = mpo(op_id, p);
      const int l = dl_[p], d = dd_[p], r = dr_[p];
      gauge_qr_right(site_[p].p, l, d, r, spare.p, bt.p, sig_.p);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_B;
      gauge_qr_right(ket[p].p, l, d, r, spare.p, kt.p, sig_.p);
      std::swap(ket[p], spare);
      ZgemmDesc g = zgemm_desc(ket[p - 1].p, sig_.p, spare.p, dl_[p - 1] * dd_[p - 1], l, l);
      zgemm(st_, g);
      std::swap(ket[p - 1], spare);
      pool_put(std::move(mixR[p]));
      mixR[p] = pool_get((size_t)l * w.ml * l);
      env_update_rect(mixR[p + 1].p, kt.p, bt.p, w.w2r.p, mixR[p].p, r, r, w.mr, d, l, l, w.ml);
      if (has_shift) {
        pool_put(std::move(ovR[p]));
        ovR[p] = pool_get((size_t)l * l);
        env_update_rect(ovR[p + 1].p, kt.p, bt.p, ident(d).w2r.p, ovR[p].p, r, r, 1, d, l, l, 1);
      }
    }
    // _is_converged (wavefunction.py:285-301): |1 - |<phi_i | phi_{i-1}>|| < conv_tol
    HIP_CHECK(hipMemcpyAsync(sig_.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
    zc* T = sig_.p;
    zc* Tn = sig2_.p;
    for (int p = 0; p < L_; ++p) {
      const int dl = dl_[p], d = dd_[p], dr = dr_[p];
      ZgemmDesc u = zgemm_desc(T, prev[p].p, tmp1_.p, dl, d * dr, dl);
      zgemm(st_, u);
      ZgemmDesc t = zgemm_desc(site_[p].p, tmp1_.p, Tn, dr, dr, dl * d);
      t.transA = 1; t.conjA = 1; t.lda = dr;
      zgemm(st_, t);
      std::swap(T, Tn);
    }
    hzc ov;
    HIP_CHECK(hipMemcpyAsync(&ov, T, sizeof(zc), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    if (std::fabs(1.0 - std::abs(ov)) < conv_tol) break;
    if (it == maxstep) break;
  }
  center_ = 0;
  invalidate_env();
  for (auto* v : {&ket, &prev, &mixL, &mixR, &ovL, &ovR})
    for (auto& b : *v) pool_put(std::move(b));
  pool_put(std::move(spare)); pool_put(std::move(kt)); pool_put(std::move(bt));
  if (iters_out) *iters_out = std::min(it, maxstep);
  return nrm;
}

}  // namespace mitdvp
