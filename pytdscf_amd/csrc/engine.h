// engine.h -- device-resident one-site TDVP engine (host orchestration).
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mitdvp.h"
#include "common.h"
#include "krylov_dev.h"
#include "qr.h"
#include "small_site.h"
#include "svd.h"
#include "vecops.h"

namespace mitdvp {

// growable device buffer (complex elements)
struct DevBuf {
  zc* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void reserve(size_t elems) {  // contents are NOT preserved
    if (elems <= n) return;
    release();
    if (elems == 0) return;
    HIP_CHECK(hipMalloc(&p, elems * sizeof(zc)));
    n = elems;
  }
  void grow_preserve(size_t elems, size_t used, hipStream_t st) {
    if (elems <= n) return;
    zc* q = nullptr;
    HIP_CHECK(hipMalloc(&q, elems * sizeof(zc)));
    if (p && used) {
      HIP_CHECK(hipMemcpyAsync(q, p, used * sizeof(zc), hipMemcpyDeviceToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    release();
    p = q;
    n = elems;
  }
};

struct MpoSite {
  int ml = 0, d = 0, mr = 0;
  DevBuf w2l;  // W2L[(i,t)][(c,j)] = W[c,i,j,t]   (d*mr) x (ml*d)
  DevBuf w2r;  // W2R[(i,c)][(t,j)] = W[c,i,j,t]   (d*ml) x (mr*d)
  // block-sparse W stage (finite-state-machine MPOs: most (c, t) blocks of W are zero): the same matrices with their
  // rows ordered (t, i) / (c, i) so that zero blocks line up with the GEMM's 64 x 16 tile grid, and per row tile the
  // list of K tiles that hold a non-zero (ZgemmDesc::klist); sp_frac = visited / all tiles (1 = dense, not used)
  DevBuf w2lt, w2rt, kl_l, kl_r;
  int kl_stride_l = 0, kl_stride_r = 0;
  double sp_frac_l = 1.0, sp_frac_r = 1.0;
  // row ranges [r0, r1) of the permuted matrix, whole bond states (d rows each), each either dense (most K tiles needed:
  // the plain kernel on exactly those rows) or sparse (the list kernel; tile0 = index of the range's first 64-row tile in
  // the list array: every sparse range has its own tile grid) -- mixing both kinds in one launch lets the few heavy tiles
  // crawl among the many light ones (23 ms instead of 6 at the C4 shape), and a dense state that shares a 64-row tile
  // with light ones would drag them through all K tiles (round 4: the "all applied" state of a finite-state-machine core
  // is d rows, a quarter or half of a tile)
  struct SpSeg { int r0, r1; bool dense; int tile0; };
  std::vector<SpSeg> seg_l, seg_r;
  DevBuf w2el;  // small-site environment update, -> direction: [(t,j)][(i,c)] = W[c,i,j,t]
  DevBuf w2er;  // small-site environment update, <- direction: [(c,j)][(i,t)] = W[c,i,j,t]
  // "edge" form of an apply (Engine::heff_apply_edge): with S = the MPO-bond states c whose left block L[:, c, :] is a
  // multiple lam_c of the identity and E = the states t whose right block R[:, t, :] is a multiple mu_t of the identity
  // (a canonical chain under a finite-state-machine MPO, or a direct sum of such: the "nothing applied yet" / "all
  // applied" states; a sign or weight of a summand may ride on them), and every non-zero (c, t) block of W in a row of S
  // or a column of E, the apply is two products with a reducing epilogue and no intermediate in memory.  The reduced
  // cores depend on (S, E, lam, mu), found numerically per site, and are cached:
  //   w_edge_r[i][(j, t)] = sum_{c in S} lam_c W[c, i, j, t]
  //   w_edge_l[i][(c, j)] = sum_{t in E} mu_t W[c, i, j, t] for c not in S, else 0
  std::vector<hzc> whost;           // the core as uploaded (ml, d, d, mr); empty when a bond exceeds 64 states
  std::vector<char> nzblk;          // [c * mr + t]: block (c, t) holds a non-zero
  mutable bool edge_valid = false;  // w_edge_l / w_edge_r correspond to (edge_s, edge_e)
  mutable unsigned long long edge_s = 0, edge_e = 0;
  mutable std::vector<hzc> edge_lam, edge_mu;
  mutable bool edge_has_l = false, edge_has_r = false;
  mutable int edge_skip = 0;        // local solves for which the (failed) structure check is not repeated
  mutable DevBuf w_edge_l, w_edge_r;
  mutable DevBuf w_edge_lf, w_edge_rf;  // the same cores in the epilogue's fragment order (zgemm_reduce_pack_core)
  mutable bool edge_lf_ok = false, edge_rf_ok = false;
  DevBuf wtr;  // Liouville trace operator: O2[f][(a,c,d)] = O[a,d,c,f], n = sqrt(site dim)
  int ntr = 0, mltr = 0, mrtr = 0;
  int dtr = 0;  // physical entries per (a, f) of wtr: n*n, or the size of the site's subspace when it was set
  bool set = false;
};
struct Operator {
  std::vector<MpoSite> sites;
  hzc shift{0.0, 0.0};
};

// collective callback: op on a device buffer of nbytes (complex128 elements)
//   COLL_ALLGATHER: rank r's shard is bytes [r*nbytes/N, (r+1)*nbytes/N); fill the rest
//   COLL_ALLREDUCE: element-wise sum over ranks (as float64)
typedef int (*CollFn)(void* user, int op, void* dev_ptr, size_t nbytes);
enum { COLL_ALLGATHER = 0, COLL_ALLREDUCE = 1 };

struct PhaseTimer {
  hipEvent_t a, b;
  int kind;
  bool closed;  // the end event was recorded (an exception between begin and end leaves it open)
};

class SiteShard;
// compute-unit ranges of CU-masked engines (engine_small.hip): overlapping claims throw ArgError
void cu_range_claim(int device, int first, int count);
void cu_range_release(int device, int first, int count);
int cu_ranges_claimed(int device);

class Engine {
  friend class SiteShard;  // shard.hip: the junction update works on the tensors and blocks of two engines in place

 public:
  explicit Engine(const mitdvp_config& cfg);
  ~Engine();

  // state
  void set_site(int isite, const double* reim, int l, int n, int r, int gauge);
  void get_site_shape(int isite, int* l, int* n, int* r, int* gauge) const;
  void get_site(int isite, double* out);
  void init_random(const int* dims, int bond_dim, uint64_t seed);
  void init_random_block(const int* dims, int ntot, int first, int D, uint64_t seed, bool balance);  // raw tensors of a block
  void canonicalize(double scale);
  void set_mpo_core(int op_id, int isite, const double* reim, int ml, int dout, int din, int mr);
  void set_shift(int op_id, double re, double im);

  // hot path
  void step(double dt);
  void sweep(double dt, bool forward);
  void invalidate_env();

  // observables
  hzc expect(int op_id);
  hzc autocorr();
  void save_reference();     // keep a copy of the current state ...
  hzc overlap_reference();   // ... and <copy|current state> later (autocorrelation without the t/2 trick)
  double norm();
  void site_rdm(int isite, double* out);
  void reduced_density(const int* legs, int nlen, std::vector<hzc>& out, std::vector<int>& shape);
  // SVD truncation of the bond right of the centre site (truncate_sigvec, _site_cls.py:586-690)
  int truncate_bond(double p, int max_dim, std::vector<double>& svals);
  // Liouville space (vectorised density matrices)
  void set_trace_op_core(int op_id, int isite, const double* reim, int ml, int n, int mr);
  hzc expect_trace(int op_id);
  void partial_trace(const int* legs, int nlen, std::vector<hzc>& out);
  // subspace projection of a Liouville-space site (Model(subspace_inds=...), _mps_mpo.py:135-220): the site's
  // physical index runs over the listed entries of the n*n vectorised density matrix only
  void set_subspace(int isite, int n, const int* inds, int ninds);
  int liouville_n(int isite) const;  // Hilbert-space dimension n of a Liouville-space site
  // rho <- (rho + rho^dagger) / 2 as a direct sum re-truncated to the old bonds (MPSCoef.hermitise, _mps_cls.py:2289-2312)
  void hermitise();
  void krylov_stats(int* per_site);

  void counters_get(mitdvp_counters* out);
  void counters_reset();
  void set_profiling(bool on) { profiling_ = on; }
  void set_qr_gauge_free(bool on) { qr_gauge_free_ = on; }
  void set_parallel(int nranks, int rank, CollFn fn, void* user);
  // native RCCL collectives on the engine's stream (librccl resolved with dlopen)
  static void rccl_unique_id(char out[128]);
  void set_parallel_rccl(int nranks, int rank, const char id_bytes[128]);
  int rccl_selftest();

  std::string last_error;

  // ---- building blocks (also used by the unit-level C entry points) -------
  void heff_apply(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dl, int d, int dr,
                  hzc shift);
  void keff_apply(const zc* L, const zc* R, const zc* sig, zc* out, int d1, int d2, int m, hzc shift);
  // rectangular blocks (bra bond != ket bond), no shift term: the adaptive-rank applies
  void heff_apply_rect(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dlo, int dli, int d,
                       int dro, int dri);
  // edge-structured MPO core between canonical environments: sigma = sum_t W[0,:,:,t] (psi R_t^T) + sum_{c>=1} W[c,:,:,mr-1] (L_c psi),
  // each a GEMM whose 64 x 64 tiles are contracted with W in the epilogue (zgemm_reduce): X / Y never exist
  void heff_apply_edge(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dl, int d, int dr);
  // which forms the H_eff applies of the site between these blocks take (sets trim_l_, trim_r_, edge_; one host
  // synchronisation for the numerical identity checks); the caller resets them when the local solve is over
  void choose_apply_forms(const zc* Lb, const MpoSite& w, const zc* Rb, int dl, int d, int dr);
  void keff_apply_rect(const zc* L, const zc* R, const zc* sig, zc* out, int dlo, int dli, int dro, int dri, int m);
  void env_update_rect(const zc* env_in, const zc* Tk, const zc* Tb, const zc* w2, zc* env_out, int dbi, int dki,
                       int min_, int d, int dbo, int dko, int mout, const MpoSite* sp = nullptr, int sp_side = 0);
  // generic environment update: env_in (din, min, din), T (din, d, dout),
  // W2 ((d*mout) x (min*d)) -> env_out (dout, mout, dout)
  // w2e: the small-site form of the same core (MpoSite::w2el / w2er), nullptr = general path only
  void env_update(const zc* env_in, const zc* T, const zc* w2, zc* env_out, int din, int min_, int d, int dout,
                  int mout, const zc* w2e = nullptr, const MpoSite* sp = nullptr, int sp_side = 0);
  // x <- exp(scale * Op) x ; returns Krylov dimension used
  template <class MV>
  int krylov_exp(hzc scale, MV&& matvec, zc* x, long n, int k_prev, long nsize = -1);
  // the same with the Ritz step and the convergence test on the device (krylov_dev.h): what krylov_exp runs unless
  // MITDVP_DEVICE_RITZ=0
  template <class MV>
  int krylov_exp_dev(hzc scale, MV&& matvec, zc* x, long n, int k_prev, long nsize);
  // x <- lowest eigenvector of Op (improved relaxation); returns Krylov dimension used
  template <class MV>
  int krylov_diag(MV&& matvec, zc* x, long n);
  void gauge_qr_left(const zc* psi, int dl, int d, int dr, zc* A_out, zc* sigma_out);   // Psi2Asigma
  void gauge_qr_right(const zc* psi, int dl, int d, int dr, zc* B_out, zc* Bt_out, zc* sigma_out);  // Psi2sigmaB
  void ensure_work(long max_site, long max_x, long max_y, int max_qr_m, int max_qr_n, int qr_next = 0);
  // one-site gates applied between the half-sweeps of every step / on demand
  void set_gate(int isite, const double* reim, int d);
  void apply_gates();
  // Kraus maps on purified states (single-site: ancilla inside the physical index; two-site:
  // ancilla on the next site), applied after the gates between the half-sweeps / on demand
  void set_kraus(int isite, int two_site, const double* reim, int k, int d);
  void apply_kraus();
  // Simulator.operate: variational application of operator op_id to the state (returns the norm)
  double operate(int op_id, int maxstep, double conv_tol, int* iters_out);
  // adaptive bond dimension (a1TDVP, const.adaptive / Dmax / dD / p_proj, _const_cls.py:120-124)
  void set_adaptive(bool on, int dmax, int dd, double p_proj);
  void thin_to_full_A(const zc* A, int l, int c, int r, int e, zc* out);
  void thin_to_full_B(const zc* B, int l, int c, int r, int e, zc* out);
  // several electronic states: one MPS per state, Hamiltonian blocks per (bra, ket) state pair,
  // the states' centre tensors stacked for the local solves (SplitStack, _contraction.py:479-608)
  void ms_configure(int nstate);
  int ms_nstate() const;
  void ms_set_site(int istate, int isite, const double* reim, int l, int n, int r, int gauge);
  void ms_get_site_shape(int istate, int isite, int* l, int* n, int* r, int* gauge);
  void ms_get_site(int istate, int isite, double* out);
  void ms_canonicalize(int istate, double scale);
  void ms_set_mpo_core(int op_id, int ibra, int iket, int isite, const double* reim, int ml, int dout, int din, int mr);
  void ms_set_couplej(int op_id, int ibra, int iket, double re, double im);
  void ms_step(double dt);
  hzc ms_expect(int op_id);
  double ms_operate(int op_id, int maxstep, double conv_tol, int* iters_out);
  hzc ms_autocorr();
  void ms_pops(double* out);
  // one block of a site-range sharded chain (engine_segment.hip)
  void replace_site(int isite, const double* reim, int gauge);  // same shape, environment cache kept
  void reshape_site(int isite, const double* reim, int l, int n, int r, int gauge);  // a junction site whose outer bond changed, cache kept
  void set_boundary_env(int side, const double* reim, int d, int m);
  void env_shape(int side, int bond, int* d, int* m);
  void get_env(int side, int bond, double* out);
  void build_envs(int side);
  void site_exp(double dt);
  void heff_apply_center(const double* in, double* out, int* flags);
  void split_center(bool forward);
  void bond_exp(double dt);
  void absorb_bond(bool forward);
  void get_bond(double* out, int* dim);
  void set_bond(int b, const double* reim, int dim);
  // tensor arguments of the setters / getters: host memory (0, default) or device memory on this GPU (1)
  void set_pointer_mode(int mode);
  void copy_in(zc* dst, const double* src, size_t elems);   // + stream synchronise
  void copy_out(double* dst, const zc* src, size_t elems);  // + stream synchronise
  void fold_block(int op_id, bool conj_bra, bool from_left, const double* in, int d, int m, double* out, int first = 0,
                  int count = -1);
  void site_rdm_blocks(int isite, const double* TL, const double* TR, double* out);
  hipStream_t stream() const { return st_; }
  // warm-up memory of the local solves at site p (_Debug.niter_krylov[p]); the device copy of the one-launch
  // exponentials is reconciled
  int kprev_get(int p);
  void kprev_set(int p, int k);
  // MITDVP_SMALL_KERNELS for this engine only (several ranks / engines sharing a GPU cannot rely on the persistent
  // kernels' co-residency)
  void set_small_kernels(bool on);
  // what the one-launch kernels recorded on the device since the last check (not converged / an exchange timed out):
  // raised here.  The C surface calls it at the end of EVERY entry point, so no call returns tensors, blocks or
  // observables computed by a launch that gave up (fold_block, get_env, expect, ... launch them without a sweep).
  void check_device_errors() { ss_check(); }
  const MpoSite& mpo(int op_id, int isite);
  mitdvp_config cfg;

 private:
  int ptr_mode_ = 0;
  // this engine's compute-unit range in the per-device claim table (engine_small.hip); a member, so that a constructor
  // that throws later on gives the range back
  struct CuClaim {
    int dev = -1, first = 0, count = 0;
    ~CuClaim() { if (dev >= 0) cu_range_release(dev, first, count); }
  } cu_claim_;
  // H_eff applies of the current local exponential: the last MPO-bond block of the right environment is the identity
  // (sites right of the centre are right-canonical and the MPO passes "nothing applied yet" through: the reference
  // short-circuits such blocks as well, _mps_mpo.py:510-523) -- verified numerically per site, see local_site_exp
  bool trim_r_ = false;
  bool trim_l_ = false;  // the same for the FIRST MPO-bond block of the left environment (stage S1: rows (a, c = 0) of X = psi)
  bool left_block_is_identity(const zc* L, int dl, int m);
  // gauge moves of the sweep without LAPACK's sign convention on diag(R) (qr_thin; MITDVP_QR_GAUGE_FREE=0: always the
  // Householder panels).  The unit-level entry point mitdvp_gauge_trf keeps the convention.
  bool qr_gauge_free_ = true;
  bool trim_identity_ = true;  // MITDVP_TRIM_IDENTITY=0 switches the shortcut off
  bool edge_ = false;          // the current local exponential's applies take heff_apply_edge
  int edge_mode_ = -1;         // MITDVP_EDGE_APPLY: 0 never, 1 wherever valid, -1 (default) the size rule of choose_apply_forms
  bool right_block_is_identity(const zc* R, int dr, int m);
  void identity_blocks(const zc* L, int dl, int ml, const zc* R, int dr, int mr, bool* left, bool* right);
  int L_;
  hipStream_t st_ = nullptr;
  std::vector<int> dl_, dd_, dr_, gauge_;
  std::vector<DevBuf> site_;
  std::vector<std::vector<int>> sub_;  // per site: kept entries of the n*n physical index (empty: all)
  std::vector<int> subn_;
  std::map<int, Operator> ops_;
  int center_ = -1;
  // segment mode: outer bonds wider than 1, boundary blocks supplied by the neighbours; pending bond matrix in sig_
  bool segment_ = false;
  int bnd_dl_ = 1, bnd_ml_ = 1, bnd_dr_ = 1, bnd_mr_ = 1;
  int bond_ = -1, bond_dim_ = 0, bond_site_ = 0;

  // environment cache: bond b is left of site b, b = 0..L
  std::vector<DevBuf> envL_, envR_;
  std::vector<char> envL_ok_, envR_ok_;
  std::vector<DevBuf> pool_;
  size_t pool_cap_ = 48;  // buffers kept for reuse (raised when many environment chains are alive)
  DevBuf pool_get(size_t elems);
  void pool_put(DevBuf&& b);

  // workspaces
  DevBuf X_, Y_, V_, Vdiag_, tmp1_, tmp2_, sig_, sig2_, qrwork_, red_;
  int max_diag_krylov_ = 64;
  zc* h_red_ = nullptr;  // pinned host mirror of red_ (host-coherent, mapped into the device: h_red_dev_)
  zc* h_red_dev_ = nullptr;
  unsigned* h_seq_ = nullptr;      // sequence word the publish kernel bumps and read_partials spins on
  unsigned* h_seq_dev_ = nullptr;
  unsigned seq_tag_ = 0;
  // device-resident convergence logic of the multi-launch Krylov loop (krylov_dev.h)
  bool device_ritz_ = true;      // MITDVP_DEVICE_RITZ=0: Ritz step and test on the host (two round trips per checked iteration)
  bool defer_norm_ = true;       // MITDVP_DEFER_NORM=0: a normalisation launch per Krylov vector (device path only)
  KryDev* kst_ = nullptr;
  KryPub* h_kpub_ = nullptr;     // host-coherent, mapped into the device: h_kpub_dev_
  KryPub* h_kpub_dev_ = nullptr;
  unsigned kry_tag_ = 0;
  void wait_pub(unsigned tag);   // spins until the record with this tag has been published
  size_t red_elems_ = 0;
  std::vector<int> kprev_;

  // bond-sharded multi-GPU execution
  int nranks_ = 1, rank_ = 0;
  CollFn coll_ = nullptr;
  void* rccl_comm_ = nullptr;  // ncclComm_t when the native path is active
  void* coll_user_ = nullptr;
  bool shard_range(int n, int& a0, int& a1) const;
  void collective(int op, zc* p, size_t elems);

  // K_eff applies of the current bond exponential with the identity states of the two blocks skipped (round 5): with
  // S = {c : L[:, c, :] = lam_c 1} and E = {c : R[:, c, :] = mu_c 1} (found numerically per bond, keff_prepare),
  //   sigma' = sum_{c not in S u E} L_c sigma R_c^T + sum_{c in S \ E} lam_c sigma R_c^T + sum_{c in E \ S} mu_c L_c sigma
  //            + (sum_{c in S n E} lam_c mu_c) sigma:
  // the first GEMM runs over the blocks not in S, the second over those not in E (compact copies of the kept blocks, made
  // once per exponential), the rest are scaled copies.  The reference skips such blocks outright (_mps_mpo.py:489-523).
  struct KeffCompact {
    bool on = false;
    int n1 = 0, nE = 0, nG = 0, nS = 0;  // X row layout per slab: [E \ S | general | S \ E]; n1 = nE + nG rows come from GEMM 1
    BlockList fillS{}, accE{};           // scaled copies lam_c sigma -> X; mu_c X_c -> out
    zc both = make_double2(0.0, 0.0);
    DevBuf Lc, Rc;
  } kc_;
  int kc_m_ = 0;               // MPO bond of the blocks kc_ was made from
  bool keff_ident_ = true;     // MITDVP_KEFF_IDENT=0: off (A/B testing)
  void keff_prepare(const zc* L, const zc* R, int d1, int d2, int m);
  void keff_apply_compact(const zc* sig, zc* out, int d1, int d2, hzc shift);
  bool sparse_w_ = true;       // MITDVP_SPARSE_W=0: always the dense W stage (A/B testing)
  // the W stage of an apply / environment update: dense GEMM, or row ranges of dense / list kernels (returns the
  // executed share of the dense flop count)
  double w_stage(const MpoSite* sp, int side, const zc* w2, int d, int mout, int min_, int ncol, int nbatch);
  bool small_kernels_ = true;  // MITDVP_SMALL_KERNELS=0: always the general multi-launch kernels (A/B testing)
  // small-bond kernel family (small_site.hip): one launch per apply / environment update / local exponential
  SmallSync ss_;
  DevBuf ss_part_;               // chunk partials
  int n_cu_ = 0;
  bool ss_dirty_ = false;        // small-site launches issued since the last error check
  std::vector<char> exp_small_;  // per site: its local exponentials (site and bonds) run in one launch each
  std::vector<int> ss_shape_key_;
  bool small_ok() const;
  bool chain_heff(SmallChain& c, const zc* L, const MpoSite& w, const zc* R, int dl, int d, int dr, bool exp_mode) const;
  bool chain_keff(SmallChain& c, const zc* L, const zc* R, int d1, int d2, int m, bool exp_mode) const;
  bool chain_env(SmallChain& c, const zc* T, const zc* w2e, int din, int min_, int d, int dout, int mout) const;
  zc* ss_partials(const SmallChain& c);
  SmallSync* qr_sync();          // exchange state for the persistent QR panel kernel (nullptr: per-column launches)
  QrHistory* qr_hist_ = nullptr; // this engine's memory of the shapes whose fast panels keep failing (qr.h)
  void ss_refresh_plan();
  void ss_check();               // raises what the small-site kernels recorded (not converged / timed out)
  void ss_pull_kprev();
  bool small_site_exp(int p, double dt);
  bool small_bond_exp(int p, const zc* Lb, const zc* Rb, int dim, int m, double dt);
  struct Gate { DevBuf u; int d = 0; };
  std::map<int, Gate> gates_;
  std::vector<DevBuf> ref_;
  std::vector<int> ref_l_, ref_r_;
  struct KrausOp { DevBuf b; int k = 0, d = 0; bool two_site = false; };
  std::map<int, KrausOp> kraus_;
  void kraus_core(const zc* theta, int m, int d, int K, int n, const KrausOp& op, zc* out);
  void recanonicalize(int lo, int hi, DevBuf& spare);
  void build_left_envs();

  // adaptive bond dimension
  bool adaptive_ = false;
  int ad_dmax_ = 100, ad_dd_ = 10;
  double ad_p_ = 1e-4;
  std::vector<DevBuf> full_;  // widened neighbour tensors of the current half-sweep (get_superblock_full)
  std::vector<int> fdl_, fdr_;
  void adaptive_prepare();
  void build_superblock_full(bool forward);
  int select_rank(const zc* hl, long hl_rows, const zc* ks, const zc* hr, long hr_cols, int dmin, int dmax);
  bool adaptive_site(int p, double dt, bool forward, DevBuf& spare);
  // runs between the site exponential and the gauge move of an adaptive step (the junction update's regularisation
  // of the centre tensor, _mps_parallel.py:362-370); the centre is still on site p, at its new shape
  std::function<void()> ad_site_hook_;

  // counters
  mitdvp_counters cnt_{};
  bool profiling_ = false;
  std::vector<PhaseTimer> pending_;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evpool_;
  void timer_begin(int kind);
  void timer_end();
  void resolve_timers();
  int cur_timer_ = -1;

  Operator& op(int id);
  void upload_mpo_core(MpoSite& s, const double* reim, int ml, int dout, int din, int mr);

  // several electronic states (MPS-SM, nstate > 1): engine_multi.hip
  struct Multi;
  std::shared_ptr<Multi> ms_;
  Multi& ms();
  void ms_require_ready();
  void ms_build_chains();
  void ms_ident_cores();
  void ms_build_right_envs();
  void ms_sweep(double dt, bool forward);
  void ms_site_exp(int p, double dt);
  void size_workspaces();
  void build_right_envs();
  void local_site_exp(int p, double dt);
  void require_ready(bool open_ends = false);  // open_ends: outer bonds wider than 1 without boundary blocks (fold_block)
  hzc scale_site(double dt) const;
  hzc scale_bond(double dt) const;
  void read_partials(size_t off_elems, size_t count);
};

}  // namespace mitdvp
