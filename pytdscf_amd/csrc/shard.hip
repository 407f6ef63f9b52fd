// shard.hip -- ONE RANK of the site-range sharded (real-space parallel) one-site TDVP, driven natively:
// the block's half-sweeps, the joint update of the two sites facing each other across a rank boundary and
// the neighbour messages around it are one C-ABI call per time step (mitdvp_shard_step).
//
// Reference: /root/reference/pytdscf/_mps_parallel.py -- MPSCoefParallel.propagate (:106-268),
// propagate_joint_two_sites (:270-470), the mpi4py messages send_Psi_to_left (:698-707), send_op_sys_to_left
// (:761-807), send_B_to_right (:728-740), send_joint_sigvec_to_right (:541-597), send_op_sys_to_right (:612-628);
// regularisation of small singular values SiteCoef.gauge_trf(regularize=True) (_site_cls.py:207-246) and
// truncate_sigvec(p, regularize=True, keepdim=True) (:586-690); pseudo-inverse multiply_sigvec_pinv (:709-754).
//
// Transport: chain neighbours exchange device buffers with grouped ncclSend / ncclRecv on the block engine's stream
// (RCCL over xGMI: one link per neighbour pair, no collective anywhere on the data path).  Ranks that share a GPU
// (the one-GPU test box: RCCL refuses two ranks on one device) plug in a host callback instead
// (mitdvp_shard_set_transport); the junction code is the same.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <mutex>

#include "capi_internal.h"
#include "engine_internal.h"
#include "engine_krylov.inc"
#include "rccl_dyn.h"

namespace mitdvp {

namespace {
constexpr double RCOND = 1e-13;        // _site_cls.py:24
constexpr double SQRT_EPSRHO = 1e-4;   // _site_cls.py:22
inline double lift(double s) { return s > SQRT_EPSRHO ? s : s + SQRT_EPSRHO * std::exp(-s / SQRT_EPSRHO); }
}  // namespace

typedef mitdvp_p2p_fn P2PFn;  // op 0: send, 1: receive (include/mitdvp.h)

class SiteShard {
 public:
  // the engines are owned by the C handle (mitdvp_shard): block = this rank's sites, joint = the two-site engine of
  // the junction to the right (nullptr on the last rank); dr_next = right bond of the right neighbour's first site
  SiteShard(Engine* block, Engine* joint, int rank, int world, int nsite_block, int dr_next)
      : rank_(rank), world_(world), n_(nsite_block), block_(block), joint_(joint), dr_next_(dr_next) {}
  // Pair mode: BOTH ranks of a junction run its update, bond-sharded over the pair (the engine's exact tensor
  // parallelism, DESIGN 7.2, with the two-rank all-gather / all-reduce carried by the shard's own point-to-point
  // transport): the partner rank works instead of waiting.  jleft = this rank's copy of the two-site engine of the
  // junction to its LEFT (nullptr on rank 0); dl_prev = left bond of the left neighbour's last site.
  void enable_pair(Engine* jleft, int dl_prev) {
    if (rank_ > 0 && (!jleft || dl_prev < 1)) throw ArgError("shard: pair mode needs the left junction engine and dl_prev");
    jleft_ = jleft;
    dl_prev_ = dl_prev;
    pair_ = true;
  }
  bool pair_mode() const { return pair_; }
  int rank() const { return rank_; }
  ~SiteShard() {
    if (comm_.load()) (void)RcclApi::get().comm_destroy(static_cast<ncclComm_t>(comm_.load()));
  }

  void set_options(int regularize, double p_svd) { regularize_ = regularize != 0; p_svd_ = p_svd; }
  void set_transport(P2PFn fn, void* user) { fn_ = fn; user_ = user; }
  void attach_rccl(const char id_bytes[128]) {
    const RcclApi& r = RcclApi::get();
    if (void* old = comm_.exchange(nullptr)) (void)r.comm_destroy(static_cast<ncclComm_t>(old));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id) < 128 ? sizeof(id) : 128);
    ncclComm_t comm = nullptr;
    rccl_check(r.comm_init_rank(&comm, world_, id, rank_), "ncclCommInitRank");
    comm_.store(comm);
  }

  // the joint matrix of the junction to the right (joint_sigvec_not_pinv of the left rank)
  void set_joint(const double* reim, int dim) {
    if (!joint_) throw ArgError("shard: the last rank holds no joint matrix");
    if (dim < 1 || !reim) throw ArgError("shard: bad joint matrix");
    X_.reserve((size_t)dim * dim);
    xdim_ = dim;
    block_->copy_in(X_.p, reim, (size_t)dim * dim);
  }
  void get_joint(double* out, int* dim) {
    if (!joint_ || xdim_ < 1) throw ArgError("shard: no joint matrix");
    *dim = xdim_;
    if (out) block_->copy_out(out, X_.p, (size_t)xdim_ * xdim_);
  }

  // propagate_along_sweep over the block (_mps_cls.py:798-1014); with skip_end the end site keeps the centre but is
  // not propagated (:876-877)
  void sweep_block(double dt, bool forward, bool skip_end) {
    Engine& b = *block_;
    const int end = forward ? n_ - 1 : 0;
    // const.adaptive (_mps_cls.py:863-987): every site but the centre widened at the start of the half-sweep
    // (get_superblock_full over the block; its end site is an orthonormal tensor then, the bond beyond it is not
    // touched), the serial adaptive step at every site but the end site, against the frozen boundary blocks
    const bool ad = b.adaptive_ && n_ > 1;
    DevBuf spare;
    if (ad) {
      b.require_ready();
      if (b.center_ != (forward ? 0 : n_ - 1)) throw ArgError("shard: an adaptive half-sweep starts at the block's centre site");
      if (forward) b.build_right_envs();
      else b.build_left_envs();
      b.adaptive_prepare();
      b.build_superblock_full(forward);
      spare = b.pool_get(b.V_.n / MAXK);
    }
    for (int p = forward ? 0 : n_ - 1; forward ? p < n_ : p >= 0; p += forward ? 1 : -1) {
      if (skip_end && p == end) break;
      if (ad && p != end && b.adaptive_site(p, dt, forward, spare)) continue;
      b.site_exp(dt);
      if (p == end) break;
      b.split_center(forward);
      b.bond_exp(dt);
      b.absorb_bond(forward);
    }
    if (ad) {
      b.pool_put(std::move(spare));
      b.ss_check();
    }
  }

  // MPSCoefParallel.propagate: one time step of this rank
  void step(double dt) {
    if (world_ == 1) {
      sweep_block(dt, true, false);
      sweep_block(dt, false, false);
      return;
    }
    // host wall clock per phase (the block's stream is drained at the phase boundaries by the message groups anyway):
    // what a rank spends sweeping its block, and what it spends in / waiting for the junction updates
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    bool fwd = rank_ % 2 == 0;
    auto t0 = now();
    sweep_block(dt, fwd, !((fwd && rank_ == world_ - 1) || (!fwd && rank_ == 0)));
    HIP_CHECK(hipStreamSynchronize(block_->st_));
    auto t1 = now();
    junctions(dt, 0);
    auto t2 = now();
    fwd = !fwd;
    sweep_block(dt, fwd, !((fwd && rank_ == world_ - 1) || (!fwd && rank_ == 0)));
    HIP_CHECK(hipStreamSynchronize(block_->st_));
    auto t3 = now();
    junctions(dt, 1);
    auto t4 = now();
    ms_block_ += ms(t0, t1) + ms(t2, t3);
    ms_junction_ += ms(t1, t2) + ms(t3, t4);
    n_steps_ += 1;
  }
  void phase_times(double* block_ms, double* junction_ms, long* steps) const {
    *block_ms = ms_block_; *junction_ms = ms_junction_; *steps = n_steps_;
  }

  void junctions(double dt, int parity) {
    const bool left = rank_ % 2 == parity && rank_ < world_ - 1, right = rank_ % 2 != parity && rank_ > 0;
    if (pair_) {
      if (left) junction_pair(dt, true);
      else if (right) junction_pair(dt, false);
    } else if (left) {
      junction_left(dt);
    } else if (right) {
      junction_right();
    }
  }

  // neighbour ping over every junction (both directions); returns the number of mismatching values
  int selftest() {
    if (world_ == 1) return 0;
    int bad = 0;
    const int n = 6;
    DevBuf buf;
    buf.reserve(n);
    std::vector<hzc> h(n);
    hipStream_t st = block_->st_;
    auto fill = [&](int r, double add) {
      for (int i = 0; i < n; ++i) h[i] = hzc(i + 10.0 * r + add, -(double)i);
      HIP_CHECK(hipMemcpyAsync(buf.p, h.data(), n * sizeof(zc), hipMemcpyHostToDevice, st));
      HIP_CHECK(hipStreamSynchronize(st));
    };
    auto check = [&](int r, double add) {
      std::vector<hzc> g(n);
      HIP_CHECK(hipMemcpyAsync(g.data(), buf.p, n * sizeof(zc), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      for (int i = 0; i < n; ++i)
        if (g[i] != hzc(i + 10.0 * r + add, -(double)i)) ++bad;
    };
    for (int parity = 0; parity < 2; ++parity) {
      if (rank_ % 2 == parity && rank_ < world_ - 1) {
        fill(rank_, 0.0);
        xfer_begin(); send_dev(buf.p, n, rank_ + 1); xfer_end();
        xfer_begin(); recv_dev(buf.p, n, rank_ + 1); xfer_end();
        check(rank_, 1.0);
      } else if (rank_ % 2 != parity && rank_ > 0) {
        xfer_begin(); recv_dev(buf.p, n, rank_ - 1); xfer_end();
        check(rank_ - 1, 0.0);
        fill(rank_ - 1, 1.0);
        xfer_begin(); send_dev(buf.p, n, rank_ - 1); xfer_end();
      }
    }
    return bad;
  }

  // a grouped ncclSend / ncclRecv of `elems` complex numbers from this rank to itself: the RCCL point-to-point path
  // on a one-rank communicator (what the one-GPU test box can exercise)
  int self_sendrecv(size_t elems) {
    if (!comm_.load()) throw ArgError("shard: no RCCL communicator (mitdvp_shard_attach_rccl)");
    DevBuf a, b;
    a.reserve(elems); b.reserve(elems);
    hipStream_t st = block_->st_;
    vec_randn(st, a.p, (long)elems, 4242);
    HIP_CHECK(hipMemsetAsync(b.p, 0, elems * sizeof(zc), st));
    xfer_begin(); send_dev(a.p, elems, rank_); recv_dev(b.p, elems, rank_); xfer_end();
    std::vector<hzc> ha(elems), hb(elems);
    HIP_CHECK(hipMemcpyAsync(ha.data(), a.p, elems * sizeof(zc), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipMemcpyAsync(hb.data(), b.p, elems * sizeof(zc), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    int bad = 0;
    for (size_t i = 0; i < elems; ++i) bad += ha[i] != hb[i];
    return bad;
  }

  void traffic(double* bytes, long* messages) const { *bytes = bytes_; *messages = messages_; }
  bool native_transport() const { return comm_.load() != nullptr && fn_ == nullptr; }

 private:
  int rank_, world_, n_;
  Engine* block_;
  Engine* joint_;
  Engine* jleft_ = nullptr;  // pair mode: two-site engine of the junction to the LEFT
  int dl_prev_ = 0;          // pair mode: left bond dimension of the left neighbour's last site
  bool pair_ = false;
  int pair_peer_ = -1;       // the rank at the other end of the junction being updated (collective callback)
  DevBuf psi_l_, env_l_, xl_, coll_tmp_, kbuf_;  // (xl_: the LEFT junction's joint matrix on its right rank; X_ is the right junction's)
  int dr_next_ = 0;  // right bond dimension of the right neighbour's first site
  DevBuf X_, psi_r_, env_r_, xin_, tmpa_, tmpb_;
  int xdim_ = 0;
  bool regularize_ = false;
  double p_svd_ = -1.0;  // < 0: the joint matrix is not truncated
  P2PFn fn_ = nullptr;
  void* user_ = nullptr;
  // written by the attach (which the host side runs on a helper thread with a deadline) and read by the sweep
  std::atomic<void*> comm_{nullptr};
  bool in_group_ = false;
  std::vector<char> stage_;
  double bytes_ = 0;
  long messages_ = 0;
  double ms_block_ = 0, ms_junction_ = 0;
  long n_steps_ = 0;

  // ---- transport ------------------------------------------------------------------------------------------
  // A message group = everything between xfer_begin and xfer_end, all with ONE neighbour.  Library RCCL: a grouped
  // ncclSend / ncclRecv on the block engine's stream.  Callback transport (torch.distributed carries host-staged
  // buffers; what the one-GPU multi-rank tests run): the SAME contract -- operations are only POSTED by send_dev /
  // recv_dev and complete at xfer_end, so that code which touched a receive buffer before the group closed, or relied
  // on the order of operations inside a group, fails in the tests exactly as it would over RCCL.  The blocking
  // callbacks are then issued in an order that cannot deadlock: the lower rank of the pair sends all it posted, then
  // receives; the higher rank receives first (per peer, RCCL too matches sends and receives in posting order).
  struct PostedOp { bool send; const zc* src; zc* dst; size_t elems; int peer; };
  std::vector<PostedOp> posted_;
  bool open_ = false;
  // Library RCCL: the group runs on the block engine's stream, operands may have been written on the junction engines'
  // streams and are read there afterwards -- ordered by EVENTS (the junction streams' work before the group, the group
  // before their next work), the host does not wait.  (Round 4 drained all three streams before and the block stream
  // after every group: four host synchronisations per message group.)  The callback transport stages through the host
  // and keeps the drains.
  hipEvent_t ev_[3] = {nullptr, nullptr, nullptr};
  hipEvent_t event(int i) {
    if (!ev_[i]) HIP_CHECK(hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming));
    return ev_[i];
  }
  void order_before_group() {
    Engine* js[2] = {joint_, jleft_};
    for (int i = 0; i < 2; ++i)
      if (js[i] && js[i]->st_ != block_->st_) {
        HIP_CHECK(hipEventRecord(event(i), js[i]->st_));
        HIP_CHECK(hipStreamWaitEvent(block_->st_, event(i), 0));
      }
  }
  void order_after_group() {
    HIP_CHECK(hipEventRecord(event(2), block_->st_));
    Engine* js[2] = {joint_, jleft_};
    for (int i = 0; i < 2; ++i)
      if (js[i] && js[i]->st_ != block_->st_) HIP_CHECK(hipStreamWaitEvent(js[i]->st_, event(2), 0));
  }
  void xfer_begin() {
    if (open_) throw ArgError("shard: message groups do not nest");
    if (!fn_) {
      if (!comm_.load()) throw ArgError("shard: no transport (mitdvp_shard_attach_rccl or mitdvp_shard_set_transport)");
      order_before_group();
      open_ = true;
      posted_.clear();
      const ncclResult_t r = RcclApi::get().group_start();
      if (r != ncclSuccess) { open_ = false; rccl_check(r, "ncclGroupStart"); }
      in_group_ = true;
      return;
    }
    HIP_CHECK(hipStreamSynchronize(block_->st_));
    if (joint_) HIP_CHECK(hipStreamSynchronize(joint_->st_));  // operands may come from any of the engines
    if (jleft_) HIP_CHECK(hipStreamSynchronize(jleft_->st_));
    open_ = true;
    posted_.clear();
  }
  void send_dev(const zc* p, size_t elems, int peer) {
    if (!open_) throw ArgError("shard: send outside a message group");
    bytes_ += 16.0 * (double)elems;
    messages_ += 1;
    if (fn_) { posted_.push_back({true, p, nullptr, elems, peer}); return; }
    group_check(RcclApi::get().send(p, 2 * elems, ncclDouble, peer, static_cast<ncclComm_t>(comm_.load()), block_->st_), "ncclSend");
  }
  void recv_dev(zc* p, size_t elems, int peer) {
    if (!open_) throw ArgError("shard: receive outside a message group");
    if (fn_) { posted_.push_back({false, nullptr, p, elems, peer}); return; }
    group_check(RcclApi::get().recv(p, 2 * elems, ncclDouble, peer, static_cast<ncclComm_t>(comm_.load()), block_->st_), "ncclRecv");
  }
  // an error inside an open group closes the group before it is raised (a group left open would swallow every later call)
  void group_check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return;
    if (in_group_) { in_group_ = false; (void)RcclApi::get().group_end(); }
    open_ = false;
    rccl_check(r, what);
  }
  void xfer_end() {
    if (!open_) throw ArgError("shard: xfer_end without xfer_begin");
    open_ = false;
    if (in_group_) {
      in_group_ = false;
      rccl_check(RcclApi::get().group_end(), "ncclGroupEnd");
      order_after_group();  // the junction engines read the received buffers from THEIR streams next
      return;
    }
    if (posted_.empty()) return;
    const int peer = posted_[0].peer;
    for (const auto& o : posted_)
      if (o.peer != peer) throw ArgError("shard: a message group talks to one neighbour");
    for (int pass = 0; pass < 2; ++pass) {
      const bool sends = (pass == 0) == (rank_ < peer);  // lower rank: sends, then receives; higher rank: the reverse
      for (const auto& o : posted_) {
        if (o.send != sends) continue;
        stage_.resize(o.elems * sizeof(zc));
        if (o.send) {
          HIP_CHECK(hipMemcpy(stage_.data(), o.src, o.elems * sizeof(zc), hipMemcpyDeviceToHost));
          if (fn_(user_, 0, o.peer, stage_.data(), o.elems * sizeof(zc)) != 0) throw HipError("shard: the send callback failed");
        } else {
          if (fn_(user_, 1, o.peer, stage_.data(), o.elems * sizeof(zc)) != 0) throw HipError("shard: the receive callback failed");
          HIP_CHECK(hipMemcpy(o.dst, stage_.data(), o.elems * sizeof(zc), hipMemcpyHostToDevice));
        }
      }
    }
    posted_.clear();
  }

  // adaptive ranks: the shapes travel ahead of the tensors (a receive is posted with its exact size), one complex
  // number per message group of its own
  DevBuf hdr_;
  void send_dims(int peer, int x, int y) {
    hdr_.reserve(1);
    const hzc h((double)x, (double)y);
    HIP_CHECK(hipMemcpy(hdr_.p, &h, sizeof(zc), hipMemcpyHostToDevice));
    xfer_begin(); send_dev(hdr_.p, 1, peer); xfer_end();
    HIP_CHECK(hipStreamSynchronize(block_->st_));  // hdr_ is written again by the next header
  }
  void recv_dims(int peer, int* x, int* y) {
    hdr_.reserve(1);
    xfer_begin(); recv_dev(hdr_.p, 1, peer); xfer_end();
    HIP_CHECK(hipStreamSynchronize(block_->st_));
    hzc h(0.0, 0.0);
    HIP_CHECK(hipMemcpy(&h, hdr_.p, sizeof(zc), hipMemcpyDeviceToHost));
    *x = (int)std::lround(h.real()); *y = (int)std::lround(h.imag());
  }

  // both directions in one message group (the pair mode's two ranks tell each other the shapes of what they are about to send)
  DevBuf hdr_in_;
  void exchange_dims(int peer, int x, int y, int* ox, int* oy) {
    hdr_.reserve(1);
    hdr_in_.reserve(1);
    const hzc h((double)x, (double)y);
    HIP_CHECK(hipMemcpy(hdr_.p, &h, sizeof(zc), hipMemcpyHostToDevice));
    exchange(hdr_.p, 1, hdr_in_.p, 1, peer);
    HIP_CHECK(hipStreamSynchronize(block_->st_));
    hzc g(0.0, 0.0);
    HIP_CHECK(hipMemcpy(&g, hdr_in_.p, sizeof(zc), hipMemcpyDeviceToHost));
    *ox = (int)std::lround(g.real()); *oy = (int)std::lround(g.imag());
  }

  struct DeviceMode {  // the engines' tensor arguments are device pointers while the shard drives them
    Engine& e; int old;
    explicit DeviceMode(Engine& en) : e(en), old(en.ptr_mode_) { e.ptr_mode_ = 1; }
    ~DeviceMode() { e.ptr_mode_ = old; }
  };
  static const double* dp(const zc* p) { return reinterpret_cast<const double*>(p); }

  static double fro(const zc* p, size_t n) {  // debugging aid (MITDVP_SHARD_DEBUG): Frobenius norm through the host
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<hzc> h(n);
    HIP_CHECK(hipMemcpy(h.data(), p, n * sizeof(zc), hipMemcpyDeviceToHost));
    double a = 0;
    for (auto& v : h) a += std::norm(v);
    return std::sqrt(a);
  }
  static void peek(const char* what, int rank, const zc* p, int D, int M) {  // a few entries of an environment block (d, m, d)
    HIP_CHECK(hipDeviceSynchronize());
    std::vector<hzc> h((size_t)D * M * D);
    HIP_CHECK(hipMemcpy(h.data(), p, h.size() * sizeof(zc), hipMemcpyDeviceToHost));
    double tr = 0, a = 0;
    for (int i = 0; i < D; ++i) tr += h[((size_t)i * M) * D + i].real();
    for (auto& v : h) a += std::norm(v);
    std::fprintf(stderr, "[shard %d] %s: D %d M %d trace(m=0) %.6g fro %.6g  [0,0]=%.4g [1,1]=%.4g [0,1]=%.4g%+.4gi\n", rank, what, D, M, tr, std::sqrt(a),
                 h[0].real(), D > 1 ? h[((size_t)1 * M) * D + 1].real() : 0.0, D > 1 ? h[1].real() : 0.0, D > 1 ? h[1].imag() : 0.0);
  }
  static bool dbg() { static const bool d = std::getenv("MITDVP_SHARD_DEBUG") != nullptr; return d; }

  // x (D x D) -> pinv(x, rcond) = V diag(1/s) U^H on the engine's stream (multiply_sigvec_pinv, _site_cls.py:734)
  void pinv_dev(Engine& J, const zc* x, int D, zc* out) {
    DevBuf U = J.pool_get((size_t)D * D), Vh = J.pool_get((size_t)D * D), work = J.pool_get(svd_work_elems(D, D)),
           sc = J.pool_get((size_t)D / 2 + 1);
    std::vector<double> s(D);
    int sweeps = 0;
    svd_jacobi(J.st_, x, D, D, U.p, s.data(), Vh.p, work.p, &sweeps);
    for (int k = 0; k < D; ++k) s[k] = s[k] > RCOND * s[0] ? 1.0 / s[k] : 0.0;
    HIP_CHECK(hipMemcpyAsync(sc.p, s.data(), D * sizeof(double), hipMemcpyHostToDevice, J.st_));
    scale_cols(J.st_, U.p, D, D, D, reinterpret_cast<const double*>(sc.p));
    // out[i][j] = sum_k conj(Vh[k][i]) (inv_k conj(U[j][k]))
    ZgemmDesc g = zgemm_desc(Vh.p, U.p, out, D, D, D);
    g.transA = 1; g.conjA = 1; g.lda = D;
    g.transB = 1; g.conjB = 1; g.ldb = D;
    zgemm(J.st_, g);
    HIP_CHECK(hipStreamSynchronize(J.st_));  // s (host) was read by the async copy
    J.pool_put(std::move(U)); J.pool_put(std::move(Vh)); J.pool_put(std::move(work)); J.pool_put(std::move(sc));
  }

  // SiteCoef.gauge_trf(regularize=True) on the centre tensor (_site_cls.py:207-246): SVD of the (D_l D_r x d)
  // unfolding, small singular values lifted, tensor rebuilt.  Singular values that are exactly zero are lifted along an
  // orthonormal completion of the singular vectors, as LAPACK's are in the reference (svd_jacobi completes the vectors
  // of a numerically rank-deficient input; round 5 -- before, they were rounding residue parallel to the leading vectors,
  // and truncate_joint built environment blocks from them: 4 % of <Psi|Psi> lost per step on the zero-padded product
  // start of the reference's tests/test_mpi.py).
  void regularize_center(Engine& J) {
    const int p = J.center_;
    if (p < 0) throw ArgError("shard: no centre site to regularise");
    const int dl = J.dl_[p], d = J.dd_[p], dr = J.dr_[p];
    const int r = dl * dr, k = std::min(r, d);
    DevBuf M = J.pool_get((size_t)r * d), U = J.pool_get((size_t)r * k), Vh = J.pool_get((size_t)k * d),
           work = J.pool_get(svd_work_elems(r, d)), sc = J.pool_get((size_t)k / 2 + 1);
    permute_0213(J.st_, J.site_[p].p, M.p, dl, d, dr, 1);  // (dl, d, dr) -> (dl, dr, d)
    std::vector<double> s(k);
    int sweeps = 0;
    svd_jacobi(J.st_, M.p, r, d, U.p, s.data(), Vh.p, work.p, &sweeps);
    if (dbg()) {
      std::fprintf(stderr, "[shard %d] regularize_center (%d x %d): |theta| %.15g |U| %.15g |Vh| %.15g s =", rank_, r, d,
                   fro(J.site_[p].p, (size_t)r * d), fro(U.p, (size_t)r * k), fro(Vh.p, (size_t)k * d));
      for (int i = 0; i < std::min(k, 6); ++i) std::fprintf(stderr, " %.6e", s[i]);
      std::fprintf(stderr, "\n");
    }
    for (int i = 0; i < k; ++i) s[i] = lift(s[i]);
    HIP_CHECK(hipMemcpyAsync(sc.p, s.data(), k * sizeof(double), hipMemcpyHostToDevice, J.st_));
    scale_cols(J.st_, U.p, r, k, k, reinterpret_cast<const double*>(sc.p));
    ZgemmDesc g = zgemm_desc(U.p, Vh.p, M.p, r, d, k);
    zgemm(J.st_, g);
    permute_0213(J.st_, M.p, J.site_[p].p, dl, dr, d, 1);  // back to (dl, d, dr)
    HIP_CHECK(hipStreamSynchronize(J.st_));
    if (dbg()) std::fprintf(stderr, "[shard %d] regularize_center out: |theta| %.15g\n", rank_, fro(J.site_[p].p, (size_t)r * d));
    J.pool_put(std::move(M)); J.pool_put(std::move(U)); J.pool_put(std::move(Vh)); J.pool_put(std::move(work));
    J.pool_put(std::move(sc));
  }

  // truncate_sigvec(Asite, sigvec, Bsite, p, regularize, keepdim=True) (_site_cls.py:586-690) on the two-site engine
  // (site 0 = A, pending bond matrix, site 1 = B): A <- A U, B <- Vh B, sigma <- diag(s' / |s'|) with zeros on the cut
  // values; both blocks through the new tensors are rebuilt (_mps_parallel.py:447-464)
  void truncate_joint(Engine& J) {
    const int D = J.bond_dim_;
    const int dl = J.dl_[0], d0 = J.dd_[0], d1 = J.dd_[1], dr = J.dr_[1];
    DevBuf U = J.pool_get((size_t)D * D), Vh = J.pool_get((size_t)D * D), work = J.pool_get(svd_work_elems(D, D));
    std::vector<double> s(D);
    int sweeps = 0;
    if (dbg()) {
      HIP_CHECK(hipStreamSynchronize(J.st_));
      std::fprintf(stderr, "[shard %d] truncate_joint in: |sigma| %.15g |A| %.15g |B| %.15g D %d\n", rank_, fro(J.sig_.p, (size_t)D * D),
                   fro(J.site_[0].p, (size_t)dl * d0 * D), fro(J.site_[1].p, (size_t)D * d1 * dr), D);
    }
    svd_jacobi(J.st_, J.sig_.p, D, D, U.p, s.data(), Vh.p, work.p, &sweeps);
    if (dbg()) {
      std::fprintf(stderr, "[shard %d] svd: sweeps %d s =", rank_, sweeps);
      for (int k = 0; k < std::min(D, 6); ++k) std::fprintf(stderr, " %.6e", s[k]);
      std::fprintf(stderr, " |U| %.15g |Vh| %.15g\n", fro(U.p, (size_t)D * D), fro(Vh.p, (size_t)D * D));
    }
    double tot = 0, cum = 0;
    for (double v : s) tot += v;
    // a zero or non-finite spectrum: refuse before any tensor is overwritten (the reference would divide by zero and
    // carry NaN into both neighbours)
    if (!(tot > 0.0) || !std::isfinite(tot)) throw NotConverged("shard: the joint bond matrix has a zero or non-finite spectrum");
    int idx = 1;  // argmax of an all-False array is 0 (numpy), hence idx = 1 when no cumulative weight reaches 1 - p
    for (int k = 0; k < D; ++k) {  // idx = argmax(cumsum / total >= 1 - p) + 1
      cum += s[k];
      if (cum / tot >= 1.0 - p_svd_) { idx = k + 1; break; }
    }
    double nrm2 = 0;
    std::vector<double> thin(idx);
    for (int k = 0; k < idx; ++k) {
      thin[k] = (regularize_ && D > 1) ? lift(s[k]) : s[k];
      nrm2 += thin[k] * thin[k];
    }
    if (!(nrm2 > 0.0) || !std::isfinite(nrm2)) throw NotConverged("shard: the kept singular values of the joint bond matrix have zero norm");
    std::vector<hzc> xn((size_t)D * D, hzc(0, 0));
    for (int k = 0; k < idx; ++k) xn[(size_t)k * D + k] = hzc(thin[k] / std::sqrt(nrm2), 0);
    DevBuf a2 = J.pool_get(J.site_[0].n), b2 = J.pool_get(J.site_[1].n);
    {
      ZgemmDesc g = zgemm_desc(J.site_[0].p, U.p, a2.p, dl * d0, D, D);
      zgemm(J.st_, g);
    }
    {
      ZgemmDesc g = zgemm_desc(Vh.p, J.site_[1].p, b2.p, D, d1 * dr, D);
      zgemm(J.st_, g);
    }
    HIP_CHECK(hipMemcpyAsync(J.sig_.p, xn.data(), xn.size() * sizeof(zc), hipMemcpyHostToDevice, J.st_));
    HIP_CHECK(hipStreamSynchronize(J.st_));
    std::swap(J.site_[0], a2);
    std::swap(J.site_[1], b2);
    const MpoSite& w0 = J.mpo(0, 0);
    const MpoSite& w1 = J.mpo(0, 1);
    J.env_update(J.envL_[0].p, J.site_[0].p, w0.w2l.p, J.envL_[1].p, dl, w0.ml, d0, D, w0.mr, w0.w2el.p, &w0, 0);
    transpose_rev3(J.st_, J.site_[1].p, J.tmp2_.p, D, d1, dr);
    J.env_update(J.envR_[2].p, J.tmp2_.p, w1.w2r.p, J.envR_[1].p, dr, w1.mr, d1, D, w1.ml, w1.w2er.p, &w1, 1);
    J.ss_check();
    if (dbg()) { peek("truncate out: envL[1]", rank_, J.envL_[1].p, D, w0.mr); peek("truncate out: envR[1]", rank_, J.envR_[1].p, D, w1.ml); }
    if (dbg())
      std::fprintf(stderr, "[shard %d] truncate_joint out: idx %d |X'| %.15g |A U| %.15g |Vh B| %.15g\n", rank_, idx, fro(J.sig_.p, (size_t)D * D),
                   fro(J.site_[0].p, (size_t)dl * d0 * D), fro(J.site_[1].p, (size_t)D * d1 * dr));
    J.pool_put(std::move(U)); J.pool_put(std::move(Vh)); J.pool_put(std::move(work));
    J.pool_put(std::move(a2)); J.pool_put(std::move(b2));
  }

  // Site L +dt/2 and the bond -dt/2 of a junction update on the two-site engine J (centre on site 0 on entry, on site 1
  // on return).  With const.adaptive (_mps_parallel.py:319-345, :371-374) the two-site superblock [Psi, B] is widened
  // (get_superblock_full), the junction's rank chosen by get_adaptive_rank_and_block, the left site propagated into the
  // widened bond, regularised, split, and the bond matrix propagated in the blocks at the new rank: the serial adaptive
  // step on two sites whose outer blocks are the two ranks' environments.  A junction already at its maximal rank takes
  // the plain step.
  void junction_first_half(Engine& J, Engine& b, double dt) {
    bool grown = false;
    if (b.adaptive_) {
      J.set_adaptive(true, b.ad_dmax_, b.ad_dd_, b.ad_p_);
      J.adaptive_prepare();
      J.build_superblock_full(true);
      DevBuf spare = J.pool_get(J.V_.n / MAXK);
      if (regularize_) J.ad_site_hook_ = [this, &J] { regularize_center(J); };
      struct Unhook { Engine& e; ~Unhook() { e.ad_site_hook_ = nullptr; } } unhook{J};
      grown = J.adaptive_site(0, dt, true, spare);
      J.pool_put(std::move(spare));
      J.ss_check();
    }
    if (!grown) {
      J.site_exp(dt);
      if (regularize_) regularize_center(J);  // trans_next_psite_AsigmaB(regularize=True), :362-370
      J.split_center(true);
      J.bond_exp(dt);
      J.kprev_set(1, J.kprev_get(0));
      J.absorb_bond(true);
    } else {
      J.kprev_set(1, J.kprev_get(0));
    }
  }

  // The left rank of a junction (propagate_joint_two_sites, _mps_parallel.py:270-470): receives psi_R and the block
  // right of it, updates both sites (site L +dt/2, bond -dt/2, site R +dt/2, bond -dt/2), returns B, X' and the block
  // left of B; its own last site becomes A X' again (send_joint_sigvec_to_right, :541-597).
  void junction_left(double dt) {
    Engine& b = *block_;
    Engine& J = *joint_;
    const bool ad = b.adaptive_;
    const int nb = rank_ + 1, pl = n_ - 1;
    const int dl = b.dl_[pl], d0 = b.dd_[pl], D = b.dr_[pl];
    if (xdim_ != D) throw ArgError("shard: joint matrix and block bond dimension differ");
    const MpoSite& w0 = J.mpo(0, 0);
    const MpoSite& w1 = J.mpo(0, 1);
    const int d1 = w1.d, Mr = w1.mr;
    int Dr = dr_next_;
    if (ad) {  // the neighbour's bonds change from step to step: (its junction bond, the right bond of its first site)
      int Dchk = 0;
      recv_dims(nb, &Dchk, &Dr);
      if (Dchk != D) throw ArgError("shard: the neighbour's junction bond differs from this rank's");
    }
    if (Dr < 1) throw ArgError("shard: the right neighbour's bond dimension was not given at creation");
    psi_r_.reserve((size_t)D * d1 * Dr);
    env_r_.reserve((size_t)Dr * Mr * Dr);
    xfer_begin();
    recv_dev(psi_r_.p, (size_t)D * d1 * Dr, nb);
    recv_dev(env_r_.p, (size_t)Dr * Mr * Dr, nb);
    xfer_end();
    if (!b.envL_ok_[pl]) throw ArgError("shard: the block's left environment at its last site is missing");
    DeviceMode mj(J), mb(b);
    J.set_site(0, dp(b.site_[pl].p), dl, d0, D, MITDVP_GAUGE_C);
    J.set_site(1, dp(psi_r_.p), D, d1, Dr, MITDVP_GAUGE_C);
    J.set_boundary_env(0, dp(b.envL_[pl].p), dl, w0.ml);
    J.set_boundary_env(1, dp(env_r_.p), Dr, Mr);
    // all four local solves share the warm-up memory of the last site the block's sweep propagated
    // (_Debug.site_now is set by the sweep only, _mps_cls.py:880)
    const int mem = n_ - 2;
    J.require_ready();
    J.kprev_set(0, b.kprev_get(mem));
    xin_.reserve((size_t)D * D);
    pinv_dev(J, X_.p, D, xin_.p);
    J.set_bond(1, dp(xin_.p), D);  // psi_L X^+
    J.absorb_bond(false);
    J.replace_site(1, dp(psi_r_.p), MITDVP_GAUGE_PSI);
    J.split_center(false);  // psi_R = sigma B, block through B
    J.absorb_bond(false);
    junction_first_half(J, b, dt);
    J.site_exp(dt);
    J.split_center(false);
    J.bond_exp(dt);
    b.kprev_set(mem, J.kprev_get(1));
    if (p_svd_ >= 0.0) truncate_joint(J);  // truncate=True, :437-466
    const int Dn = J.dr_[0];  // the junction's rank after the update (= D without adaptive ranks)
    HIP_CHECK(hipStreamSynchronize(J.st_));
    if (ad) send_dims(nb, Dn, 0);
    // B, X', the block left of B
    xfer_begin();
    send_dev(J.site_[1].p, (size_t)Dn * d1 * Dr, nb);
    send_dev(J.sig_.p, (size_t)Dn * Dn, nb);
    send_dev(J.envL_[1].p, (size_t)Dn * w0.mr * Dn, nb);
    xfer_end();
    HIP_CHECK(hipStreamSynchronize(b.st_));  // the group read J's buffers on the block's stream (callback transport: done)
    X_.reserve((size_t)Dn * Dn);
    xdim_ = Dn;
    HIP_CHECK(hipMemcpyAsync(X_.p, J.sig_.p, (size_t)Dn * Dn * sizeof(zc), hipMemcpyDeviceToDevice, b.st_));
    HIP_CHECK(hipStreamSynchronize(b.st_));
    if (Dn != D) b.reshape_site(pl, dp(J.site_[0].p), dl, d0, Dn, MITDVP_GAUGE_A);
    else b.replace_site(pl, dp(J.site_[0].p), MITDVP_GAUGE_A);
    b.set_boundary_env(1, dp(J.envR_[1].p), Dn, w0.mr);
    b.set_bond(n_, dp(X_.p), Dn);
    b.absorb_bond(false);
  }

  // ---- pair mode ------------------------------------------------------------------------------------------
  // two-rank all-gather / all-reduce of the bond-sharded junction engine over the point-to-point transport
  static int pair_collective(void* user, int op, void* dev_ptr, size_t nbytes) {
    try {
      static_cast<SiteShard*>(user)->pair_coll(op, static_cast<zc*>(dev_ptr), nbytes / sizeof(zc));
      return 0;
    } catch (...) {
      return 1;
    }
  }
  void pair_coll(int op, zc* p, size_t elems) {
    const int peer = pair_peer_, me = rank_ < peer ? 0 : 1;
    if (op == COLL_ALLGATHER) {  // my half is in place; the halves swap
      const size_t half = elems / 2;
      exchange(p + (size_t)me * half, half, p + (size_t)(1 - me) * half, half, peer);
    } else {  // sum: each rank sends its partial block whole and adds the partner's
      coll_tmp_.reserve(elems);
      exchange(p, elems, coll_tmp_.p, elems, peer);
      Engine& J = me == 0 ? *joint_ : *jleft_;
      vec_axpby(J.st_, p, coll_tmp_.p, (long)elems, make_double2(1.0, 0.0), make_double2(1.0, 0.0));
      HIP_CHECK(hipStreamSynchronize(J.st_));
    }
  }
  // send a buffer to `peer` and receive one from it: one group with RCCL; with the blocking callback transport the
  // lower rank sends first
  void exchange(const zc* snd, size_t ns, zc* rcv, size_t nr, int peer) {
    xfer_begin();
    send_dev(snd, ns, peer);
    recv_dev(rcv, nr, peer);
    xfer_end();
  }

  // propagate_joint_two_sites (_mps_parallel.py:270-470) run by BOTH ranks of the junction on identical copies of the
  // two facing sites and the two boundary blocks: every apply and environment update of the two-site engine contracts
  // half of the bra-side bond on each GPU, the Krylov algebra, the QRs and the small SVDs are replicated (identical
  // data, identical decisions).  Each rank keeps what it needs at the end (left: A, X', the block right of A; right: B,
  // X', the block left of B): nothing is sent back.
  void junction_pair(double dt, bool is_left) {
    Engine& b = *block_;
    const bool ad = b.adaptive_;
    Engine& J = is_left ? *joint_ : *jleft_;
    const int peer = is_left ? rank_ + 1 : rank_ - 1;
    const MpoSite& w0 = J.mpo(0, 0);
    const MpoSite& w1 = J.mpo(0, 1);
    int dl, d0, D, d1, Dr;
    DevBuf& kbuf = kbuf_;
    kbuf.reserve(1);
    hzc kmsg(0.0, 0.0);
    if (is_left) {
      const int pl = n_ - 1;
      dl = b.dl_[pl]; d0 = b.dd_[pl]; D = b.dr_[pl]; d1 = w1.d; Dr = dr_next_;
      if (ad) {  // adaptive ranks: the neighbour's bonds change from step to step
        int Dchk = 0;
        exchange_dims(peer, dl, D, &Dchk, &Dr);
        if (Dchk != D) throw ArgError("shard: the neighbour's junction bond differs from this rank's");
      }
      if (xdim_ != D) throw ArgError("shard: joint matrix and block bond dimension differ");
      if (!b.envL_ok_[pl]) throw ArgError("shard: the block's left environment at its last site is missing");
      psi_r_.reserve((size_t)D * d1 * Dr);
      env_r_.reserve((size_t)Dr * w1.mr * Dr);
      kmsg = hzc((double)b.kprev_get(n_ - 2), 0.0);
      HIP_CHECK(hipMemcpy(kbuf.p, &kmsg, sizeof(zc), hipMemcpyHostToDevice));
      xfer_begin();
      send_dev(b.site_[pl].p, (size_t)dl * d0 * D, peer);
      send_dev(b.envL_[pl].p, (size_t)dl * w0.ml * dl, peer);
      send_dev(X_.p, (size_t)D * D, peer);
      send_dev(kbuf.p, 1, peer);
      recv_dev(psi_r_.p, (size_t)D * d1 * Dr, peer);
      recv_dev(env_r_.p, (size_t)Dr * w1.mr * Dr, peer);
      xfer_end();
    } else {
      D = b.dl_[0]; d1 = b.dd_[0]; Dr = b.dr_[0]; d0 = w0.d; dl = dl_prev_;
      if (ad) {
        int Dchk = 0;
        exchange_dims(peer, D, Dr, &dl, &Dchk);
        if (Dchk != D) throw ArgError("shard: the neighbour's junction bond differs from this rank's");
      }
      if (b.center_ != 0) throw ArgError("shard: the block's first site must be the centre before a junction update");
      if (!b.envR_ok_[1]) throw ArgError("shard: the block's right environment at its first site is missing");
      psi_l_.reserve((size_t)dl * d0 * D);
      env_l_.reserve((size_t)dl * w0.ml * dl);
      xl_.reserve((size_t)D * D);
      xfer_begin();
      recv_dev(psi_l_.p, (size_t)dl * d0 * D, peer);
      recv_dev(env_l_.p, (size_t)dl * w0.ml * dl, peer);
      recv_dev(xl_.p, (size_t)D * D, peer);
      recv_dev(kbuf.p, 1, peer);
      send_dev(b.site_[0].p, (size_t)D * d1 * Dr, peer);
      send_dev(b.envR_[1].p, (size_t)Dr * w1.mr * Dr, peer);
      xfer_end();
      HIP_CHECK(hipMemcpy(&kmsg, kbuf.p, sizeof(zc), hipMemcpyDeviceToHost));
    }
    const zc* psi_l = is_left ? b.site_[n_ - 1].p : psi_l_.p;
    const zc* env_l = is_left ? b.envL_[n_ - 1].p : env_l_.p;
    const zc* psi_r = is_left ? psi_r_.p : b.site_[0].p;
    const zc* env_r = is_left ? env_r_.p : b.envR_[1].p;
    DeviceMode mj(J), mb(b);
    J.set_parallel(1, 0, nullptr, nullptr);  // set-up copies are local
    J.set_site(0, dp(psi_l), dl, d0, D, MITDVP_GAUGE_C);
    J.set_site(1, dp(psi_r), D, d1, Dr, MITDVP_GAUGE_C);
    J.set_boundary_env(0, dp(env_l), dl, w0.ml);
    J.set_boundary_env(1, dp(env_r), Dr, w1.mr);
    J.require_ready();
    J.kprev_set(0, (int)kmsg.real());
    xin_.reserve((size_t)D * D);
    zc* Xj = is_left ? X_.p : xl_.p;  // this junction's joint matrix
    pinv_dev(J, Xj, D, xin_.p);
    pair_peer_ = peer;
    J.set_parallel(2, is_left ? 0 : 1, &SiteShard::pair_collective, this);
    struct Unshare { Engine& e; ~Unshare() { e.set_parallel(1, 0, nullptr, nullptr); } } unshare{J};
    J.set_bond(1, dp(xin_.p), D);  // psi_L X^+
    J.absorb_bond(false);
    J.replace_site(1, dp(psi_r), MITDVP_GAUGE_PSI);
    J.split_center(false);
    J.absorb_bond(false);
    junction_first_half(J, b, dt);  // (pair mode: both ranks take the same decisions on identical data -- every sharded
                                    // contraction ends in a collective)
    J.site_exp(dt);
    J.split_center(false);
    J.bond_exp(dt);
    if (p_svd_ >= 0.0) truncate_joint(J);
    const int Dn = J.dr_[0];  // the junction's rank after the update (= D without adaptive ranks)
    HIP_CHECK(hipStreamSynchronize(J.st_));
    if (is_left) { X_.reserve((size_t)Dn * Dn); xdim_ = Dn; }
    else xl_.reserve((size_t)Dn * Dn);
    Xj = is_left ? X_.p : xl_.p;
    HIP_CHECK(hipMemcpyAsync(Xj, J.sig_.p, (size_t)Dn * Dn * sizeof(zc), hipMemcpyDeviceToDevice, b.st_));
    HIP_CHECK(hipStreamSynchronize(b.st_));
    if (is_left) {
      b.kprev_set(n_ - 2, J.kprev_get(1));
      if (Dn != D) b.reshape_site(n_ - 1, dp(J.site_[0].p), dl, d0, Dn, MITDVP_GAUGE_A);
      else b.replace_site(n_ - 1, dp(J.site_[0].p), MITDVP_GAUGE_A);
      b.set_boundary_env(1, dp(J.envR_[1].p), Dn, w0.mr);
      b.set_bond(n_, dp(X_.p), Dn);
      b.absorb_bond(false);
    } else {
      if (Dn != D) b.reshape_site(0, dp(J.site_[1].p), Dn, d1, Dr, MITDVP_GAUGE_B);
      else b.replace_site(0, dp(J.site_[1].p), MITDVP_GAUGE_B);
      b.set_boundary_env(0, dp(J.envL_[1].p), Dn, w1.ml);
      b.set_bond(0, dp(xl_.p), Dn);
      b.absorb_bond(true);
    }
  }

  // The right rank: sends its centre tensor and the block right of it, takes B, X' and the block left of B.
  void junction_right() {
    Engine& b = *block_;
    const bool ad = b.adaptive_;
    const int nb = rank_ - 1;
    const int D = b.dl_[0], d = b.dd_[0], dr = b.dr_[0];
    const MpoSite& w = b.mpo(0, 0);
    if (b.center_ != 0) throw ArgError("shard: the block's first site must be the centre before a junction update");
    if (!b.envR_ok_[1]) throw ArgError("shard: the block's right environment at its first site is missing");
    if (ad) send_dims(nb, D, dr);
    xfer_begin();
    send_dev(b.site_[0].p, (size_t)D * d * dr, nb);
    send_dev(b.envR_[1].p, (size_t)dr * w.mr * dr, nb);
    xfer_end();
    int Dn = D, unused = 0;
    if (ad) {
      recv_dims(nb, &Dn, &unused);
      if (Dn < D) throw ArgError("shard: the junction's rank came back smaller");
    }
    tmpa_.reserve((size_t)Dn * d * dr);
    tmpb_.reserve((size_t)Dn * w.ml * Dn);
    xin_.reserve((size_t)Dn * Dn);
    xfer_begin();
    recv_dev(tmpa_.p, (size_t)Dn * d * dr, nb);
    recv_dev(xin_.p, (size_t)Dn * Dn, nb);
    recv_dev(tmpb_.p, (size_t)Dn * w.ml * Dn, nb);
    xfer_end();
    DeviceMode mb(b);
    if (Dn != D) b.reshape_site(0, dp(tmpa_.p), Dn, d, dr, MITDVP_GAUGE_B);
    else b.replace_site(0, dp(tmpa_.p), MITDVP_GAUGE_B);
    b.set_boundary_env(0, dp(tmpb_.p), Dn, w.ml);
    b.set_bond(0, dp(xin_.p), Dn);
    b.absorb_bond(true);
  }
};

}  // namespace mitdvp

// --------------------------------------------------------------------------------------------------- C ABI
struct mitdvp_shard {
  mitdvp_engine block, joint, jleft;  // handed out by mitdvp_shard_engine; they live exactly as long as the shard
  std::unique_ptr<mitdvp::SiteShard> s;
  std::string err;
  int device = 0;
  ~mitdvp_shard() { s.reset(); }  // the shard (its RCCL communicator, its buffers) goes before the engines
};

namespace {
thread_local std::string g_serr;

template <class F>
int sguard(mitdvp_shard* h, F&& f) {
  std::string* dst = h ? &h->err : &g_serr;
  try {
    if (h) {
      hipError_t e = hipSetDevice(h->device);
      if (e != hipSuccess) { *dst = hipGetErrorString(e); return MITDVP_EHIP; }
    }
    f();
    return MITDVP_OK;
  } catch (const mitdvp::NotConverged& ex) { *dst = ex.what(); return MITDVP_ENOTCONV;
  } catch (const mitdvp::ArgError& ex) { *dst = ex.what(); return MITDVP_EINVAL;
  } catch (const mitdvp::HipError& ex) { *dst = ex.what(); return MITDVP_EHIP;
  } catch (const std::bad_alloc&) { *dst = "out of host memory"; return MITDVP_ENOMEM;
  } catch (const std::exception& ex) { *dst = ex.what(); return MITDVP_ESTATE; }
}
}  // namespace

#define SH_CALL(h, body)                                                \
  if (!(h) || !(h)->s) { g_serr = "null handle"; return MITDVP_EINVAL; } \
  return sguard((h), [&] { body; })
#define SH_NEED(p) \
  if (!(p)) throw mitdvp::ArgError("null pointer argument")

extern "C" {

int mitdvp_shard_create(const mitdvp_config* cfg, int rank, int world, int nsite_block, int dr_next, mitdvp_shard** out) {
  if (!cfg || !out) { g_serr = "null argument"; return MITDVP_EINVAL; }
  *out = nullptr;
  auto* h = new mitdvp_shard();
  h->device = h->block.device = h->joint.device = cfg->device;
  int rc = sguard(nullptr, [&] {
    if (world < 1 || rank < 0 || rank >= world) throw mitdvp::ArgError("shard: bad rank / world");
    if (nsite_block < (world > 1 ? 2 : 1)) throw mitdvp::ArgError("shard: a block needs at least two sites");
    if (rank < world - 1 && dr_next < 1) throw mitdvp::ArgError("shard: dr_next (right bond of the next rank's first site) must be given");
    mitdvp_config c = *cfg;
    c.nsite = nsite_block;
    h->block.e.reset(new mitdvp::Engine(c));
    if (rank < world - 1) {
      c.nsite = 2;
      h->joint.e.reset(new mitdvp::Engine(c));
    }
    h->s.reset(new mitdvp::SiteShard(h->block.e.get(), h->joint.e.get(), rank, world, nsite_block, dr_next));
  });
  if (rc != MITDVP_OK) { delete h; return rc; }
  *out = h;
  return MITDVP_OK;
}
void mitdvp_shard_destroy(mitdvp_shard* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  delete h;
}
const char* mitdvp_shard_last_error(const mitdvp_shard* h) { return h ? h->err.c_str() : g_serr.c_str(); }

int mitdvp_shard_engine(mitdvp_shard* h, int which, mitdvp_engine** out) {
  SH_CALL(h, {
    SH_NEED(out);
    if (which < 0 || which > 2) throw mitdvp::ArgError("shard_engine: 0 = block, 1 = junction engine, 2 = left junction engine (pair mode)");
    if (which == 1 && !h->joint.e) throw mitdvp::ArgError("shard_engine: the last rank has no junction engine");
    if (which == 2 && !h->jleft.e) throw mitdvp::ArgError("shard_engine: no left junction engine (mitdvp_shard_enable_pair on a rank > 0)");
    *out = which == 0 ? &h->block : (which == 1 ? &h->joint : &h->jleft);
  });
}
int mitdvp_shard_enable_pair(mitdvp_shard* h, int dl_prev) {
  SH_CALL(h, {
    if (h->s->rank() > 0 && !h->jleft.e) {
      mitdvp_config c = h->block.e->cfg;
      c.nsite = 2;
      h->jleft.device = h->device;
      h->jleft.e.reset(new mitdvp::Engine(c));
    }
    h->s->enable_pair(h->jleft.e.get(), dl_prev);
  });
}
int mitdvp_shard_set_options(mitdvp_shard* h, int regularize, double p_svd) { SH_CALL(h, h->s->set_options(regularize, p_svd)); }
int mitdvp_shard_set_joint(mitdvp_shard* h, const double* reim, int dim) { SH_CALL(h, { SH_NEED(reim); h->s->set_joint(reim, dim); }); }
int mitdvp_shard_get_joint(mitdvp_shard* h, double* reim_out, int* dim) { SH_CALL(h, { SH_NEED(dim); h->s->get_joint(reim_out, dim); }); }
int mitdvp_shard_set_transport(mitdvp_shard* h, mitdvp_p2p_fn fn, void* user) { SH_CALL(h, h->s->set_transport(fn, user)); }
int mitdvp_shard_attach_rccl(mitdvp_shard* h, const char id[128]) { SH_CALL(h, { SH_NEED(id); h->s->attach_rccl(id); }); }
int mitdvp_shard_selftest(mitdvp_shard* h, int* mismatches) { SH_CALL(h, { SH_NEED(mismatches); *mismatches = h->s->selftest(); }); }
int mitdvp_shard_self_sendrecv(mitdvp_shard* h, size_t elems, int* mismatches) {
  SH_CALL(h, { SH_NEED(mismatches); *mismatches = h->s->self_sendrecv(elems); });
}
int mitdvp_shard_sweep(mitdvp_shard* h, double dt_au, int forward, int skip_end) { SH_CALL(h, h->s->sweep_block(dt_au, forward != 0, skip_end != 0)); }
int mitdvp_shard_junctions(mitdvp_shard* h, double dt_au, int parity) { SH_CALL(h, h->s->junctions(dt_au, parity)); }
int mitdvp_shard_step(mitdvp_shard* h, double dt_au) { SH_CALL(h, h->s->step(dt_au)); }
int mitdvp_shard_phase_times(mitdvp_shard* h, double* block_ms, double* junction_ms, long* steps) {
  SH_CALL(h, { SH_NEED(block_ms); SH_NEED(junction_ms); SH_NEED(steps); h->s->phase_times(block_ms, junction_ms, steps); });
}
int mitdvp_shard_traffic(mitdvp_shard* h, double* bytes, long* messages) {
  SH_CALL(h, { SH_NEED(bytes); SH_NEED(messages); h->s->traffic(bytes, messages); });
}

}  // extern "C"
