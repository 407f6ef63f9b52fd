// engine_obs.hip -- observables evaluated on the device-resident MPS: norm, expectation
// values, autocorrelation, reduced densities, SVD bond truncation, Liouville-space traces.
#include "engine_internal.h"

#include <climits>

namespace mitdvp {

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

// <Psi(0)|Psi(t)> for runs without the t/2 trick (Properties._get_autocorr, properties.py:222-232,
// wf_zero._ints_wf_ovlp_mpo): the state is copied once, later states are overlapped with it
void Engine::save_reference() {
  require_ready();
  ref_.clear(); ref_.resize(L_);
  ref_l_ = dl_; ref_r_ = dr_;
  for (int p = 0; p < L_; ++p) {
    const size_t e = (size_t)dl_[p] * dd_[p] * dr_[p];
    ref_[p].reserve(e);
    HIP_CHECK(hipMemcpyAsync(ref_[p].p, site_[p].p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  }
  HIP_CHECK(hipStreamSynchronize(st_));
}

hzc Engine::overlap_reference() {
  require_ready();
  if ((int)ref_.size() != L_) throw ArgError("overlap_reference: no reference state saved");
  // T'[i][j] = sum conj(ref)[(m,s)][i] (T[m][n] C[n][(s,j)]);  bond dimensions may differ (adaptive runs)
  const zc one = make_double2(1.0, 0.0);
  size_t mx = 1;
  for (int p = 0; p < L_; ++p) mx = std::max(mx, (size_t)std::max(ref_l_[p], ref_r_[p]) * dd_[p] * std::max(dl_[p], dr_[p]));
  DevBuf T = pool_get(mx), Tn = pool_get(mx), U = pool_get(mx);
  HIP_CHECK(hipMemcpyAsync(T.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  for (int p = 0; p < L_; ++p) {
    const int dl = dl_[p], d = dd_[p], dr = dr_[p], rl = ref_l_[p], rr = ref_r_[p];
    ZgemmDesc u = zgemm_desc(T.p, site_[p].p, U.p, rl, d * dr, dl);
    zgemm(st_, u);
    ZgemmDesc t = zgemm_desc(ref_[p].p, U.p, Tn.p, rr, dr, rl * d);
    t.transA = 1; t.conjA = 1; t.lda = rr;
    zgemm(st_, t);
    std::swap(T, Tn);
  }
  hzc out;
  HIP_CHECK(hipMemcpyAsync(&out, T.p, sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  pool_put(std::move(T)); pool_put(std::move(Tn)); pool_put(std::move(U));
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
  const std::vector<int>& sub = sub_[isite];
  if (!sub.empty() && subn_[isite] != n) throw ArgError("set_trace_op_core: n differs from the site's subspace definition");
  const hzc* O = reinterpret_cast<const hzc*>(reim);  // O[a][d][c][f]  (bond, out, in, bond)
  // physical entries kept per (a, f): all n*n (index c*n + d), or the site's subspace (set_subspace comes first)
  const int dk = sub.empty() ? n * n : (int)sub.size();
  std::vector<hzc> o2((size_t)mr * ml * dk);
  for (int a = 0; a < ml; ++a)
    for (int k = 0; k < dk; ++k) {
      const int j = sub.empty() ? k : sub[k];
      const int c = j / n, dd = j % n;
      for (int f = 0; f < mr; ++f) o2[(size_t)f * ml * dk + (size_t)a * dk + k] = O[(((size_t)a * n + dd) * n + c) * mr + f];
    }
  MpoSite& s = op(op_id).sites[isite];
  s.wtr.reserve(o2.size());
  HIP_CHECK(hipMemcpyAsync(s.wtr.p, o2.data(), o2.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  s.ntr = n; s.mltr = ml; s.mrtr = mr; s.dtr = dk;
}

// ---------------------------------------------------------------------------
// subspace projection (Model(space="liouville", subspace_inds={site: P_inds}), model_cls.py:110-118): the site's
// physical index keeps the entries P_inds of the n*n vectorised density matrix.  The sweep never looks at this (the
// MPO cores and the tensors simply arrive with the shorter leg, hamiltonian_cls.py:852-880, _mps_mpo.py:196-220);
// the trace observables do: they embed the kept entries back into n x n (reshape_mat, _mps_mpo.py:135-194).
// ---------------------------------------------------------------------------
void Engine::set_subspace(int isite, int n, const int* inds, int ninds) {
  if (isite < 0 || isite >= L_) throw ArgError("set_subspace: bad site index");
  if (ninds == 0) { sub_[isite].clear(); subn_[isite] = 0; return; }
  if (n < 1 || ninds < 0 || ninds > n * n || !inds) throw ArgError("set_subspace: bad arguments");
  std::vector<char> seen((size_t)n * n, 0);
  for (int k = 0; k < ninds; ++k) {
    if (inds[k] < 0 || inds[k] >= n * n) throw ArgError("set_subspace: index outside the n*n physical leg");
    if (seen[inds[k]]) throw ArgError("set_subspace: repeated index");
    seen[inds[k]] = 1;
  }
  sub_[isite].assign(inds, inds + ninds);
  subn_[isite] = n;
  for (auto& kv : ops_) kv.second.sites[isite].ntr = 0;  // trace-operator cores of this site must be set again
}

int Engine::liouville_n(int p) const {
  if (p < 0 || p >= L_) throw ArgError("bad site index");
  if (!sub_[p].empty()) {
    if ((int)sub_[p].size() != dd_[p]) throw ArgError("Liouville space: site dimension differs from the size of its subspace");
    return subn_[p];
  }
  const int n = (int)std::lround(std::sqrt((double)dd_[p]));
  if (n * n != dd_[p]) throw ArgError("Liouville space: site dimension is not n*n");
  return n;
}

// out[l][j][r] = in[l][inv[j]][r] for the kept entries j of the full physical leg, 0 elsewhere
__global__ __launch_bounds__(256) void k_embed_phys(const zc* __restrict__ in, zc* __restrict__ out, int dl, int dsub, int dfull,
                                                   int dr, const int* __restrict__ inv) {
  const long total = (long)dl * dfull * dr;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx % dr);
    const long t = idx / dr;
    const int j = (int)(t % dfull), l = (int)(t / dfull);
    const int k = inv[j];
    out[idx] = k >= 0 ? in[((size_t)l * dsub + k) * dr + r] : make_double2(0.0, 0.0);
  }
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
    if (w.dtr != dd_[p] || (int)(sub_[p].empty() ? w.ntr * w.ntr : sub_[p].size()) != dd_[p])
      throw ArgError("trace operator: site dimension is not n*n (or the size of the site's subspace)");
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
  std::vector<DevBuf> emb(L_);     // projected sites embedded back into the full n*n leg
  std::vector<const zc*> core(L_);
  for (int p = 0; p < L_; ++p) {
    nn[p] = liouville_n(p);
    maxd = std::max(maxd, (size_t)std::max(dl_[p], dr_[p]));
    core[p] = site_[p].p;
    if (!sub_[p].empty()) {
      const int dfull = nn[p] * nn[p];
      std::vector<int> inv(dfull, -1);
      for (size_t k = 0; k < sub_[p].size(); ++k) inv[sub_[p][k]] = (int)k;
      const size_t e = (size_t)dl_[p] * dfull * dr_[p];
      emb[p] = pool_get(e + (dfull + 1) / 2 + 1);  // the index map rides behind the tensor
      int* inv_dev = reinterpret_cast<int*>(emb[p].p + e);
      HIP_CHECK(hipMemcpyAsync(inv_dev, inv.data(), dfull * sizeof(int), hipMemcpyHostToDevice, st_));
      HIP_CHECK(hipStreamSynchronize(st_));  // inv goes out of scope
      const int nb = (int)std::min<size_t>(1024, (e + 255) / 256);
      hipLaunchKernelGGL(k_embed_phys, dim3(nb), dim3(256), 0, st_, site_[p].p, emb[p].p, dl_[p], dd_[p], dfull, dr_[p], inv_dev);
      core[p] = emb[p].p;
    }
  }
  const zc one = make_double2(1.0, 0.0);
  // right environment vector: sites right of the centre are traced out
  DevBuf right = pool_get(maxd), rnext = pool_get(maxd), tq = pool_get(maxd * maxd * 0 + (size_t)maxd * maxd);
  HIP_CHECK(hipMemcpyAsync(right.p, &one, sizeof(zc), hipMemcpyHostToDevice, st_));
  for (int q = L_ - 1; q > center; --q) {
    phys_diag(st_, core[q], tq.p, dl_[q], nn[q], dr_[q], true);
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
      M = core[q];
      cols = (long)n * n * dr;
    } else {
      tmp = pool_get((size_t)dl * n * dr);
      phys_diag(st_, core[q], tmp.p, dl, n, dr, legs[q] == 0);
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
    ZgemmDesc g1 = zgemm_desc(core[center], right.p, wv.p, dl * n * n, 1, dr);  // C (x) right
    zgemm(st_, g1);
    ZgemmDesc g2 = zgemm_desc(left.p, wv.p, dm.p, (int)no, n * n, dl);
    zgemm(st_, g2);
    out.resize((size_t)no * n * n);
    HIP_CHECK(hipMemcpyAsync(out.data(), dm.p, out.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    pool_put(std::move(wv)); pool_put(std::move(dm));
  }
  pool_put(std::move(left)); pool_put(std::move(right)); pool_put(std::move(rnext)); pool_put(std::move(tq));
  for (auto& b : emb)
    if (b.p) pool_put(std::move(b));
}

// ---------------------------------------------------------------------------
// MPSCoef.hermitise (_mps_cls.py:2289-2312) = svd_conj_mpdo (:2516-2562): rho <- (rho + rho^dagger) / 2.
// Every core is doubled into the direct sum of itself and its conjugate with the two physical legs swapped
// (first site: both blocks side by side along the right bond with the factor 1/2, last site: stacked along
// the left bond), then the chain is swept left to right: two-site matrix B = rho_L rho_R, SVD, the leading
// chi = old bond dimension singular vectors stay (rho_L <- U[:, :chi], rho_R <- (S Vh)[:chi] = rho_L^H B), and
// finally the chain is brought to the site-0-centred canonical form without touching its normalisation.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hermit_double(const zc* __restrict__ in, zc* __restrict__ out, int dl, int n, int dr,
                                                      int dbl_row, int dbl_col, double scale) {
  const int l2 = dbl_row ? 2 * dl : dl, r2 = dbl_col ? 2 * dr : dr;
  const long total = (long)l2 * n * n * r2;
  const bool sym = !dbl_row && !dbl_col;  // one-site chain: the plain average
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int d2 = (int)(idx % r2);
    long t = idx / r2;
    const int c = (int)(t % n);
    t /= n;
    const int b = (int)(t % n), a2 = (int)(t / n);
    const bool lowa = dbl_row && a2 >= dl, lowd = dbl_col && d2 >= dr;
    const int a = lowa ? a2 - dl : a2, d = lowd ? d2 - dr : d2;
    const zc v = in[(((size_t)a * n + b) * n + c) * dr + d];
    const zc w = in[(((size_t)a * n + c) * n + b) * dr + d];  // dagger block: conj of the swapped legs
    zc o;
    if (sym) o = make_double2(v.x + w.x, v.y - w.y);
    else if (dbl_row && dbl_col && lowa != lowd) o = make_double2(0.0, 0.0);
    else if (lowa || lowd) o = make_double2(w.x, -w.y);
    else o = v;
    out[idx] = make_double2(scale * o.x, scale * o.y);
  }
}

void Engine::hermitise() {
  require_ready();
  std::vector<int> nn(L_);
  for (int p = 0; p < L_; ++p) {
    if (!sub_[p].empty()) throw ArgError("hermitise: sites with a subspace projection are not supported");
    nn[p] = liouville_n(p);
  }
  auto dbl = [&](int p, zc* out, bool row, bool col, double scale) {
    const size_t e = (size_t)(row ? 2 : 1) * dl_[p] * dd_[p] * (col ? 2 : 1) * dr_[p];
    const int nb = (int)std::min<size_t>(1024, (e + 255) / 256);
    hipLaunchKernelGGL(k_hermit_double, dim3(nb), dim3(256), 0, st_, site_[p].p, out, dl_[p], nn[p], dr_[p], row ? 1 : 0,
                       col ? 1 : 0, scale);
  };
  if (L_ == 1) {
    DevBuf t = pool_get(site_[0].n);
    dbl(0, t.p, false, false, 0.5);
    std::swap(site_[0], t);
    pool_put(std::move(t));
    invalidate_env();
    HIP_CHECK(hipGetLastError());
    return;
  }
  // the new tensors are collected aside and installed together at the end: a factorisation that throws half-way
  // (no convergence, out of memory) leaves the chain as it was
  std::vector<DevBuf> fresh(L_);
  std::vector<int> nbond(L_ - 1);
  // cur = the left factor of the next two-site matrix: (rows x k)
  long rows = (long)dl_[0] * dd_[0];
  int k = 2 * dr_[0];
  DevBuf cur = pool_get((size_t)rows * k);
  dbl(0, cur.p, false, true, 0.5);
  for (int p = 0; p + 1 < L_; ++p) {
    const bool last = p + 1 == L_ - 1;
    const int chi_old = dr_[p];
    const int dn = dd_[p + 1], rn = last ? dr_[p + 1] : 2 * dr_[p + 1];
    const long cols = (long)dn * rn;
    if (rows > INT_MAX / 2 || cols > INT_MAX / 2) throw ArgError("hermitise: two-site matrix too large");
    DevBuf R = pool_get((size_t)k * cols);
    dbl(p + 1, R.p, true, !last, 1.0);
    const int kk = (int)std::min(rows, cols), chi = std::min(chi_old, kk);
    DevBuf newl = pool_get(std::max(site_[p].n, (size_t)rows * chi));
    DevBuf nxt = pool_get((size_t)chi * cols);
    std::vector<double> sv;
    int sweeps = 0;
    if (rows >= k && cols >= k) {
      // B = cur R is never formed: cur = Qc Rc, R^T = Q2 R2  =>  B = Qc (Rc R2^T) Q2^T, and the SVD is that of the
      // k x k matrix in the middle (k = the doubled bond) instead of the (rows x cols) two-site matrix
      DevBuf Qc = pool_get((size_t)rows * k), Rc = pool_get((size_t)k * k), Rt = pool_get((size_t)cols * k),
             Q2 = pool_get((size_t)cols * k), R2 = pool_get((size_t)k * k), S = pool_get((size_t)k * k),
             Us = pool_get((size_t)k * k), Vsh = pool_get((size_t)k * k), T = pool_get((size_t)chi * k),
             qw = pool_get(qr_work_elems((int)std::max(rows, cols), k)), work = pool_get(svd_work_elems(k, k));
      long nl = 0;
      qr_householder(st_, cur.p, (int)rows, k, Qc.p, Rc.p, qw.p, &nl);
      transpose_batched(st_, R.p, Rt.p, k, (int)cols, cols, k, 1, 0, 0);
      qr_householder(st_, Rt.p, (int)cols, k, Q2.p, R2.p, qw.p, &nl);
      {
        ZgemmDesc g = zgemm_desc(Rc.p, R2.p, S.p, k, k, k);  // Rc R2^T
        g.transB = 1; g.ldb = k;
        zgemm(st_, g);
      }
      sv.resize(k);
      svd_jacobi(st_, S.p, k, k, Us.p, sv.data(), Vsh.p, work.p, &sweeps);
      {
        ZgemmDesc g = zgemm_desc(Qc.p, Us.p, newl.p, (int)rows, chi, k);  // rho_L = Qc Us[:, :chi]
        g.ldb = k;
        zgemm(st_, g);
      }
      {
        ZgemmDesc g = zgemm_desc(Us.p, S.p, T.p, chi, k, k);  // (Sigma Vs^H)[:chi] = Us[:, :chi]^H (Rc R2^T)
        g.transA = 1; g.conjA = 1; g.lda = k;
        zgemm(st_, g);
      }
      {
        ZgemmDesc g = zgemm_desc(T.p, Q2.p, nxt.p, chi, (int)cols, k);  // rho_R = that times Q2^T
        g.transB = 1; g.ldb = k;
        zgemm(st_, g);
      }
      HIP_CHECK(hipStreamSynchronize(st_));
      pool_put(std::move(Qc)); pool_put(std::move(Rc)); pool_put(std::move(Rt)); pool_put(std::move(Q2)); pool_put(std::move(R2));
      pool_put(std::move(S)); pool_put(std::move(Us)); pool_put(std::move(Vsh)); pool_put(std::move(T)); pool_put(std::move(qw));
      pool_put(std::move(work));
    } else {
      // a short side (first and last bond: rows or cols = the physical dimension): the two-site matrix itself is small
      DevBuf B = pool_get((size_t)rows * cols);
      {
        ZgemmDesc g = zgemm_desc(cur.p, R.p, B.p, (int)rows, (int)cols, k);
        zgemm(st_, g);
      }
      DevBuf U = pool_get((size_t)rows * kk), Vh = pool_get((size_t)kk * cols), work = pool_get(svd_work_elems((int)rows, (int)cols));
      sv.resize(kk);
      svd_jacobi(st_, B.p, (int)rows, (int)cols, U.p, sv.data(), Vh.p, work.p, &sweeps);
      copy2d(st_, newl.p, chi, U.p, kk, rows, chi, chi, make_double2(1.0, 0.0), false);  // U[:, :chi]
      {
        ZgemmDesc g = zgemm_desc(newl.p, B.p, nxt.p, chi, (int)cols, (int)rows);  // (S Vh)[:chi] = U[:, :chi]^H B
        g.transA = 1; g.conjA = 1; g.lda = chi;
        zgemm(st_, g);
      }
      HIP_CHECK(hipStreamSynchronize(st_));
      pool_put(std::move(B)); pool_put(std::move(U)); pool_put(std::move(Vh)); pool_put(std::move(work));
    }
    fresh[p] = std::move(newl);
    nbond[p] = chi;
    pool_put(std::move(R)); pool_put(std::move(cur));
    cur = std::move(nxt);
    rows = (long)chi * dn;
    k = rn;
  }
  HIP_CHECK(hipGetLastError());
  {
    const size_t e = (size_t)nbond[L_ - 2] * dd_[L_ - 1] * dr_[L_ - 1];  // <= the old tensor: the bond did not grow
    HIP_CHECK(hipMemcpyAsync(site_[L_ - 1].p, cur.p, e * sizeof(zc), hipMemcpyDeviceToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));
    pool_put(std::move(cur));
  }
  for (int p = 0; p + 1 < L_; ++p) {
    std::swap(site_[p], fresh[p]);
    pool_put(std::move(fresh[p]));
    dr_[p] = nbond[p];
    dl_[p + 1] = nbond[p];
  }
  for (int p = 0; p < L_; ++p) gauge_[p] = MITDVP_GAUGE_C;
  center_ = -1;
  invalidate_env();
  canonicalize(-1.0);  // canonicalize(superblock, orthogonal_center=0, incremental=False): the norm is kept
}

void Engine::krylov_stats(int* per_site) {
  ss_pull_kprev();
  for (int i = 0; i < L_; ++i) per_site[i] = kprev_[i];
}

}  // namespace mitdvp
