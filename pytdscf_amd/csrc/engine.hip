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
#include "engine_internal.h"
#include "engine_krylov.inc"
#include "rccl_dyn.h"

namespace mitdvp {

Engine::Engine(const mitdvp_config& c) : cfg(c), L_(c.nsite) {
  if (c.nsite < 1) throw ArgError("nsite must be >= 1");
  if (c.max_krylov < 1 || c.max_krylov > MAXK - 1) throw ArgError("max_krylov must be in [1, 20]");
  if (c.integrator != MITDVP_LANCZOS && c.integrator != MITDVP_ARNOLDI) throw ArgError("bad integrator");
  if (c.relax < 0 || c.relax > 2) throw ArgError("relax must be 0, 1 or 2");
  max_diag_krylov_ = c.max_diag_krylov > 0 ? c.max_diag_krylov : 64;
  if (const char* e = std::getenv("MITDVP_SMALL_KERNELS")) small_kernels_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MITDVP_SPARSE_W")) sparse_w_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MITDVP_TRIM_IDENTITY")) trim_identity_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MITDVP_DEVICE_RITZ")) device_ritz_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MITDVP_EDGE_APPLY")) edge_mode_ = std::atoi(e);
  if (const char* e = std::getenv("MITDVP_DEFER_NORM")) defer_norm_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MITDVP_QR_GAUGE_FREE")) qr_gauge_free_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("MITDVP_KEFF_IDENT")) keff_ident_ = std::atoi(e) != 0;
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  if (ndev < 1) throw HipError("no HIP device visible: the MI355X engine has no CPU fallback");
  if (c.device < 0 || c.device >= ndev) throw ArgError("bad device ordinal");
  HIP_CHECK(hipSetDevice(c.device));
  HIP_CHECK(hipDeviceGetAttribute(&n_cu_, hipDeviceAttributeMultiprocessorCount, c.device));
  if (c.cu_count > 0) {
    // a stream confined to the compute units [cu_first, cu_first + cu_count): mask bit u = CU u / 8 of XCD u % 8 (measured,
    // tools/cu_mask_probe.py); every XCD must keep at least one unit or the hardware falls back to units of its own choice
    if (c.cu_first < 0 || c.cu_first + c.cu_count > n_cu_ || c.cu_count % 8 != 0 || c.cu_first % 8 != 0)
      throw ArgError("cu_first / cu_count: a multiple of 8 compute units inside the device (8 k units = k CUs on every XCD)");
    std::vector<uint32_t> mask((size_t)(n_cu_ + 31) / 32, 0u);
    for (int u = c.cu_first; u < c.cu_first + c.cu_count; ++u) mask[(size_t)u / 32] |= 1u << (u % 32);
    cu_range_claim(c.device, c.cu_first, c.cu_count);  // refuses a range that overlaps another engine's
    cu_claim_.dev = c.device; cu_claim_.first = c.cu_first; cu_claim_.count = c.cu_count;
    HIP_CHECK(hipExtStreamCreateWithCUMask(&st_, (uint32_t)mask.size(), mask.data()));
    n_cu_ = c.cu_count;
    ss_.max_grid = c.cu_count;
    ss_.partitioned = true;
  } else {
    HIP_CHECK(hipStreamCreate(&st_));
  }
  qr_hist_ = qr_history_new();
  dl_.assign(L_, 0); dd_.assign(L_, 0); dr_.assign(L_, 0); gauge_.assign(L_, -1);
  site_.resize(L_);
  sub_.assign(L_, {}); subn_.assign(L_, 0);
  envL_.resize(L_ + 1); envR_.resize(L_ + 1);
  envL_ok_.assign(L_ + 1, 0); envR_ok_.assign(L_ + 1, 0);
  kprev_.assign(L_, 0);
  red_.reserve(RED_TOTAL);
  red_elems_ = RED_TOTAL;
  HIP_CHECK(hipHostMalloc((void**)&h_red_, RED_TOTAL * sizeof(zc), hipHostMallocMapped | hipHostMallocCoherent));
  HIP_CHECK(hipHostMalloc((void**)&h_seq_, 64, hipHostMallocMapped | hipHostMallocCoherent));
  *h_seq_ = 0;
  {
    void* dp = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dp, h_red_, 0));
    h_red_dev_ = static_cast<zc*>(dp);
    HIP_CHECK(hipHostGetDevicePointer(&dp, h_seq_, 0));
    h_seq_dev_ = static_cast<unsigned*>(dp);
  }
  HIP_CHECK(hipMalloc((void**)&kst_, sizeof(KryDev)));
  HIP_CHECK(hipMemsetAsync(kst_, 0, sizeof(KryDev), st_));
  HIP_CHECK(hipHostMalloc((void**)&h_kpub_, sizeof(KryPub), hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(h_kpub_, 0, sizeof(KryPub));
  {
    void* dp = nullptr;
    HIP_CHECK(hipHostGetDevicePointer(&dp, h_kpub_, 0));
    h_kpub_dev_ = static_cast<KryPub*>(dp);
  }
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
  small_sync_free(ss_);
  qr_history_free(qr_hist_);
  if (rccl_comm_) (void)RcclApi::get().comm_destroy(static_cast<ncclComm_t>(rccl_comm_));
  for (auto& t : pending_) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto& e : evpool_) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (h_red_) (void)hipHostFree(h_red_);
  if (h_seq_) (void)hipHostFree(h_seq_);
  if (h_kpub_) (void)hipHostFree(h_kpub_);
  if (kst_) (void)hipFree(kst_);
  if (st_) { zgemm_release_stream(st_); (void)hipStreamDestroy(st_); }
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
  if (pool_.size() > pool_cap_) pool_.erase(pool_.begin());  // adaptive ranks: block sizes drift, drop the oldest
}

void Engine::timer_begin(int kind) {
  if (!profiling_) return;
  std::pair<hipEvent_t, hipEvent_t> ev;
  if (!evpool_.empty()) { ev = evpool_.back(); evpool_.pop_back(); }
  else { HIP_CHECK(hipEventCreate(&ev.first)); HIP_CHECK(hipEventCreate(&ev.second)); }
  HIP_CHECK(hipEventRecord(ev.first, st_));
  pending_.push_back(PhaseTimer{ev.first, ev.second, kind, false});
  cur_timer_ = (int)pending_.size() - 1;
}
void Engine::timer_end() {
  if (!profiling_ || cur_timer_ < 0) return;
  HIP_CHECK(hipEventRecord(pending_[cur_timer_].b, st_));
  pending_[cur_timer_].closed = true;
  cur_timer_ = -1;
  if (pending_.size() > 8192) resolve_timers();
}
void Engine::resolve_timers() {
  if (pending_.empty()) return;
  HIP_CHECK(hipStreamSynchronize(st_));
  for (auto& t : pending_) {
    float ms = 0;
    if (!t.closed) {  // unwound between begin and end: nothing to read, the events go back to the pool
      evpool_.emplace_back(t.a, t.b);
      continue;
    }
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
  cur_timer_ = -1;
}
void Engine::counters_get(mitdvp_counters* out) {
  ss_dirty_ = ss_dirty_ || ss_.words != nullptr;
  ss_check();  // merges the apply counts kept on the device
  resolve_timers();
  *out = cnt_;
}
void Engine::counters_reset() {
  ss_dirty_ = ss_dirty_ || ss_.words != nullptr;
  ss_check();
  resolve_timers();
  std::memset(&cnt_, 0, sizeof(cnt_));
}

// A Krylov iteration's scalars on their way to the host.  hipMemcpyAsync + hipStreamSynchronize costs ~15 us per round
// trip on this stack; a one-workgroup kernel that copies the values into the (host-coherent, device-mapped) pinned buffer
// and then bumps a sequence word the host spins on costs ~7 us (tools/probes/sync_latency.hip).  Local exponentials of the
// mid-size regime wait for three or four such round trips each.  MITDVP_SPIN_SYNC=0: the copy + synchronise form.
__global__ __launch_bounds__(256) void k_publish(const zc* __restrict__ src, zc* __restrict__ dst, size_t count,
                                                 volatile unsigned* seq, unsigned tag) {
  for (size_t e = threadIdx.x; e < count; e += 256) dst[e] = src[e];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *seq = tag;
}

void Engine::read_partials(size_t off, size_t count) {
  cnt_.n_host_waits += 1;
  static const bool spin = !(std::getenv("MITDVP_SPIN_SYNC") && std::atoi(std::getenv("MITDVP_SPIN_SYNC")) == 0);
  if (spin && h_seq_ && count <= 16384) {
    const unsigned tag = ++seq_tag_;
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, st_, red_.p + off, h_red_dev_ + off, count, h_seq_dev_, tag);
    HIP_CHECK(hipGetLastError());
    volatile unsigned* w = h_seq_;
    for (long spins = 0; *w != tag; ++spins) {
      if ((spins & 0xFFFF) == 0xFFFF && hipStreamQuery(st_) != hipErrorNotReady) {
        // the stream is idle (or failed) and the word never arrived: surface the error / fall through after a sync
        HIP_CHECK(hipStreamSynchronize(st_));
        if (*w != tag) throw HipError("read_partials: the publish kernel did not deliver");
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the values behind the word are read with plain loads
    return;
  }
  HIP_CHECK(hipMemcpyAsync(h_red_ + off, red_.p + off, count * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
}

void Engine::wait_pub(unsigned tag) {
  volatile unsigned* w = &h_kpub_->seq;
  for (long spins = 0; *w != tag; ++spins) {
    if ((spins & 0xFFFF) == 0xFFFF && hipStreamQuery(st_) != hipErrorNotReady) {
      HIP_CHECK(hipStreamSynchronize(st_));
      if (*w != tag) throw HipError("wait_pub: the Krylov record was not published");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

// ---------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------
void Engine::set_site(int i, const double* reim, int l, int n, int r, int gauge) {
  if (i < 0 || i >= L_) throw ArgError("set_site: bad site index");
  if (l < 1 || n < 1 || r < 1) throw ArgError("set_site: bad shape");
  const size_t e = (size_t)l * n * r;
  site_[i].reserve(e);
  copy_in(site_[i].p, reim, e);
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
  copy_out(out, site_[i].p, e);
}

void Engine::set_pointer_mode(int mode) {
  if (mode != 0 && mode != 1) throw ArgError("set_pointer_mode: 0 (host) or 1 (device)");
  ptr_mode_ = mode;
}
void Engine::copy_in(zc* dst, const double* src, size_t elems) {
  if (ptr_mode_ == 0) HIP_CHECK(hipMemcpyAsync(dst, src, elems * sizeof(zc), hipMemcpyHostToDevice, st_));
  else vec_copy_raw(st_, dst, reinterpret_cast<const zc*>(src), elems);
  HIP_CHECK(hipStreamSynchronize(st_));
}
void Engine::copy_out(double* dst, const zc* src, size_t elems) {
  if (ptr_mode_ == 0) HIP_CHECK(hipMemcpyAsync(dst, src, elems * sizeof(zc), hipMemcpyDeviceToHost, st_));
  else vec_copy_raw(st_, reinterpret_cast<zc*>(dst), src, elems);
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

void Engine::upload_mpo_core(MpoSite& s, const double* reim, int ml, int dout, int din, int mr) {
  if (ml < 1 || mr < 1 || dout < 1 || dout != din) throw ArgError("set_mpo_core: need a square 4-leg core");
  const int d = dout;
  const hzc* W = reinterpret_cast<const hzc*>(reim);
  std::vector<hzc> w2l((size_t)d * mr * ml * d), w2r((size_t)d * ml * mr * d);
  std::vector<hzc> w2el(w2l.size()), w2er(w2r.size());
  for (int c = 0; c < ml; ++c)
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j)
        for (int t = 0; t < mr; ++t) {
          const hzc v = W[(((size_t)c * d + i) * d + j) * mr + t];
          w2l[((size_t)i * mr + t) * ((size_t)ml * d) + (size_t)c * d + j] = v;
          w2r[((size_t)i * ml + c) * ((size_t)mr * d) + (size_t)t * d + j] = v;
          w2el[((size_t)t * d + j) * ((size_t)d * ml) + (size_t)i * ml + c] = v;
          w2er[((size_t)c * d + j) * ((size_t)d * mr) + (size_t)i * mr + t] = v;
        }
  s.ml = ml; s.d = d; s.mr = mr;
  s.w2l.reserve(w2l.size());
  s.w2r.reserve(w2r.size());
  HIP_CHECK(hipMemcpyAsync(s.w2l.p, w2l.data(), w2l.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipMemcpyAsync(s.w2r.p, w2r.data(), w2r.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  // block-sparse forms: rows (t, i) for W2L, (c, i) for W2R; K-tile lists on the 64 x 16 tile grid
  auto sparse_form = [&](const std::vector<hzc>& w2, int mo, int mi, DevBuf& wt, DevBuf& kl, int& stride, double& frac,
                         std::vector<MpoSite::SpSeg>& segs) {
    // w2: rows (i, q) with q in [0, mo), cols (p, j) with p in [0, mi); permuted rows (q, i)
    const int M = d * mo, K = mi * d;
    frac = 1.0; stride = 0; segs.clear();
    if (K % 16 != 0) return;
    std::vector<hzc> p((size_t)M * K);
    for (int i = 0; i < d; ++i)
      for (int q = 0; q < mo; ++q)
        std::memcpy(&p[((size_t)q * d + i) * K], &w2[((size_t)i * mo + q) * K], (size_t)K * sizeof(hzc));
    const int nkt = K / 16;
    stride = nkt + 1;
    // per bond state q (rows [q d, (q + 1) d)): the K tiles that hold a non-zero
    std::vector<std::vector<char>> need(mo, std::vector<char>(nkt, 0));
    std::vector<int> cntq(mo, 0);
    for (int q = 0; q < mo; ++q) {
      for (int kt = 0; kt < nkt; ++kt) {
        bool nz = false;
        for (int r = q * d; r < (q + 1) * d && !nz; ++r)
          for (int k = kt * 16; k < kt * 16 + 16; ++k)
            if (p[(size_t)r * K + k] != hzc(0.0, 0.0)) { nz = true; break; }
        need[q][kt] = nz;
        cntq[q] += nz;
      }
    }
    std::vector<int> list;
    long executed = 0;  // in units of (row, K tile), padding rows of the tiles included: the flop the hardware runs
    for (int q0 = 0; q0 < mo;) {
      const bool dense = 2 * cntq[q0] > nkt;
      int q1 = q0 + 1;
      while (q1 < mo && (2 * cntq[q1] > nkt) == dense) ++q1;
      MpoSite::SpSeg sg{q0 * d, q1 * d, dense, 0};
      if (dense) {
        // what the matrix cores execute: whole tiles (32 rows for a range of <= 32 rows, else 64: w_stage's choice)
        const int rows = sg.r1 - sg.r0, tile = rows <= 32 ? 32 : 64;
        executed += (long)((rows + tile - 1) / tile) * tile * nkt;
      } else {
        sg.tile0 = (int)(list.size() / stride);
        for (int r = sg.r0; r < sg.r1; r += 64) {  // this range's own grid of 64-row tiles
          const size_t o = list.size();
          list.resize(o + stride, 0);
          int cnt = 0;
          for (int kt = 0; kt < nkt; ++kt) {
            bool nz = false;
            for (int q = r / d; q <= (std::min(r + 64, sg.r1) - 1) / d && !nz; ++q) nz = need[q][kt];
            if (nz) list[o + 1 + cnt++] = kt;
          }
          list[o] = cnt;
          executed += 64L * cnt;  // a tile runs all its 64 rows through every listed K tile
        }
      }
      segs.push_back(sg);
      q0 = q1;
    }
    if (list.empty()) list.assign(stride, 0);
    frac = (double)executed / ((double)M * nkt);
    wt.reserve(p.size());
    kl.reserve((list.size() * sizeof(int) + sizeof(zc) - 1) / sizeof(zc));
    HIP_CHECK(hipMemcpyAsync(wt.p, p.data(), p.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipMemcpyAsync(kl.p, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipStreamSynchronize(st_));  // the host vectors go out of scope
  };
  sparse_form(w2l, mr, ml, s.w2lt, s.kl_l, s.kl_stride_l, s.sp_frac_l, s.seg_l);
  sparse_form(w2r, ml, mr, s.w2rt, s.kl_r, s.kl_stride_r, s.sp_frac_r, s.seg_r);
  // what the edge form of an apply needs (heff_apply_edge): the core itself and the map of its non-zero blocks
  s.whost.clear(); s.nzblk.clear(); s.edge_valid = false; s.edge_skip = 0;
  if (ml <= 64 && mr <= 64) {
    s.whost.assign(W, W + (size_t)ml * d * d * mr);
    s.nzblk.assign((size_t)ml * mr, 0);
    for (int c = 0; c < ml; ++c)
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j)
          for (int t = 0; t < mr; ++t)
            if (W[(((size_t)c * d + i) * d + j) * mr + t] != hzc(0.0, 0.0)) s.nzblk[(size_t)c * mr + t] = 1;
  }
  s.w2el.reserve(w2el.size());
  s.w2er.reserve(w2er.size());
  HIP_CHECK(hipMemcpyAsync(s.w2el.p, w2el.data(), w2el.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipMemcpyAsync(s.w2er.p, w2er.data(), w2er.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  s.set = true;
}

void Engine::set_mpo_core(int op_id, int isite, const double* reim, int ml, int dout, int din, int mr) {
  if (isite < 0 || isite >= L_) throw ArgError("set_mpo_core: bad site index");
  upload_mpo_core(op(op_id).sites[isite], reim, ml, dout, din, mr);
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

void Engine::require_ready(bool open_ends) {
  for (int p = 0; p < L_; ++p) {
    if (!site_[p].p) throw ArgError("site tensor not set");
    if (p + 1 < L_ && dr_[p] != dl_[p + 1]) throw ArgError("bond dimension mismatch between neighbouring sites");
  }
  if (!segment_) {
    if (!open_ends && (dl_[0] != 1 || dr_[L_ - 1] != 1)) throw ArgError("open boundary bonds must be 1");
  } else if ((envL_ok_[0] && bnd_dl_ != dl_[0]) || (envR_ok_[L_] && bnd_dr_ != dr_[L_ - 1])) {
    throw ArgError("segment: the outer bonds differ from the boundary blocks' dimension");
  }
  size_workspaces();
  ss_refresh_plan();
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
  if (rccl_comm_) { (void)RcclApi::get().comm_destroy(static_cast<ncclComm_t>(rccl_comm_)); rccl_comm_ = nullptr; }
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
  if (rccl_comm_) {
    // native path: stream-ordered RCCL collectives on the engine's own stream -- no host
    // synchronisation, the Krylov loop stays asynchronous between its convergence checks
    const RcclApi& r = RcclApi::get();
    ncclComm_t comm = static_cast<ncclComm_t>(rccl_comm_);
    const size_t n = elems * 2;  // float64 values
    if (op == COLL_ALLGATHER) {
      const size_t chunk = n / nranks_;
      double* base = reinterpret_cast<double*>(p);
      rccl_check(r.all_gather(base + (size_t)rank_ * chunk, base, chunk, ncclDouble, comm, st_), "ncclAllGather");
    } else {
      rccl_check(r.all_reduce(p, p, n, ncclDouble, ncclSum, comm, st_), "ncclAllReduce");
    }
  } else {
    HIP_CHECK(hipStreamSynchronize(st_));
    const int rc = coll_(coll_user_, op, p, elems * sizeof(zc));
    if (rc != 0) throw HipError("collective callback failed (rc=" + std::to_string(rc) + ")");
  }
  cnt_.n_collectives += 1;
  cnt_.collective_bytes += (double)(elems * sizeof(zc));
}

// RCCL communicator owned by the engine (one process per GPU; `id` = the 128 bytes of an
// ncclUniqueId created on one rank by rccl_unique_id() and distributed out of band)
void Engine::rccl_unique_id(char out[128]) {
  ncclUniqueId id;
  rccl_check(RcclApi::get().get_unique_id(&id), "ncclGetUniqueId");
  static_assert(sizeof(id.internal) == 128, "ncclUniqueId size");
  std::memcpy(out, id.internal, 128);
}

void Engine::set_parallel_rccl(int nranks, int rank, const char id_bytes[128]) {
  if (nranks < 1 || rank < 0 || rank >= nranks) throw ArgError("set_parallel_rccl: bad rank / nranks");
  const RcclApi& r = RcclApi::get();
  if (rccl_comm_) { (void)r.comm_destroy(static_cast<ncclComm_t>(rccl_comm_)); rccl_comm_ = nullptr; }
  ncclUniqueId id;
  std::memcpy(id.internal, id_bytes, 128);
  ncclComm_t comm = nullptr;
  rccl_check(r.comm_init_rank(&comm, nranks, id, rank), "ncclCommInitRank");
  rccl_comm_ = comm;
  nranks_ = nranks; rank_ = rank; coll_ = nullptr; coll_user_ = nullptr;
}

// both collectives on a small device buffer: returns 0 when the gathered / reduced values are right
int Engine::rccl_selftest() {
  if (!rccl_comm_) throw ArgError("rccl_selftest: no RCCL communicator (call set_parallel_rccl)");
  const int per = 8;
  std::vector<hzc> h((size_t)nranks_ * per, hzc(-1.0, -1.0));
  for (int i = 0; i < per; ++i) h[(size_t)rank_ * per + i] = hzc(rank_ + 1.0, 0.5);
  DevBuf b = pool_get(h.size());
  HIP_CHECK(hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  collective(COLL_ALLGATHER, b.p, h.size());
  HIP_CHECK(hipMemcpyAsync(h.data(), b.p, h.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  int bad = 0;
  for (int r = 0; r < nranks_; ++r)
    for (int i = 0; i < per; ++i) bad += h[(size_t)r * per + i] != hzc(r + 1.0, 0.5);
  for (auto& x : h) x = hzc(rank_ + 1.0, 1.0);
  HIP_CHECK(hipMemcpyAsync(b.p, h.data(), h.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
  collective(COLL_ALLREDUCE, b.p, h.size());
  HIP_CHECK(hipMemcpyAsync(h.data(), b.p, h.size() * sizeof(zc), hipMemcpyDeviceToHost, st_));
  HIP_CHECK(hipStreamSynchronize(st_));
  const hzc want(nranks_ * (nranks_ + 1) / 2.0, (double)nranks_);
  for (auto& x : h) bad += x != want;
  pool_put(std::move(b));
  return bad;
}

// W stage: Y_b[(i,q)][n] = W2[(i,q)][(p,j)] X_b[(p,j)][n] for nbatch slabs b (X_, Y_ workspaces).  With a
// finite-state-machine MPO most (p, q) blocks of W are zero: rows of W2 ordered (q, i), per 64-row tile the list of
// 16-wide K tiles that hold a non-zero; row ranges that need most K tiles go through the plain kernel, the others
// through the list kernel, Y's rows are mapped back to (i, q).  Skipping exact zeros leaves Y bit-identical.
double Engine::w_stage(const MpoSite* sp, int side, const zc* w2, int d, int mout, int min_, int ncol, int nbatch) {
  ZgemmDesc g = zgemm_desc(w2, X_.p, Y_.p, d * mout, ncol, min_ * d);
  g.batch = nbatch; g.strideA = 0; g.strideB = (long)min_ * d * ncol; g.strideC = (long)d * mout * ncol;
  const bool use = sp && sparse_w_ && ncol >= 64 && (side == 0 ? sp->kl_l.p : sp->kl_r.p) &&
                   (side == 0 ? sp->sp_frac_l : sp->sp_frac_r) <= 0.6;
  if (!use) {
    zgemm(st_, g);
    return 1.0;
  }
  const zc* wt = side == 0 ? sp->w2lt.p : sp->w2rt.p;
  const int* kl = reinterpret_cast<const int*>(side == 0 ? sp->kl_l.p : sp->kl_r.p);
  const int stride = side == 0 ? sp->kl_stride_l : sp->kl_stride_r;
  const auto& segs = side == 0 ? sp->seg_l : sp->seg_r;
  const int K = min_ * d;
  // heavy ranges first: they are the long-running workgroups
  for (int pass = 0; pass < 2; ++pass)
    for (const auto& sgm : segs) {
      if (sgm.dense != (pass == 0)) continue;
      ZgemmDesc h = g;
      h.A = wt + (size_t)sgm.r0 * K;
      h.M = sgm.r1 - sgm.r0;
      h.rowmap_p = d; h.rowmap_s1 = (long)mout * ncol; h.rowmap_s2 = ncol; h.rowmap_r0 = sgm.r0;
      if (!sgm.dense) { h.klist = kl + (size_t)sgm.tile0 * stride; h.klist_stride = stride; }
      h.tile_cfg = (sgm.dense && h.M <= 32) ? 2 : 1;  // a dense state of <= 32 rows: the 32 x 32 tile, no rows wasted
      zgemm(st_, h);
      cnt_.n_launch += 1;
    }
  cnt_.n_launch -= 1;  // the caller counts one launch for this stage
  return side == 0 ? sp->sp_frac_l : sp->sp_frac_r;
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
  const bool triml = trim_l_ && !sharded && dlo == dli && ml > 1;
  if (triml) {
    // L[:, 0, :] is the identity: rows (a, c = 0) of X are psi itself, the GEMM runs over the other ml - 1 rows of
    // every slab (A rows gathered, C rows scattered with the same map)
    const zc one = make_double2(1.0, 0.0);
    const long row = (long)d * dri;
    copy2d(st_, X_.p, (long)ml * row, psi, row, na, (int)row, 0, one, false);
    ZgemmDesc g = zgemm_desc(L, psi, X_.p + row, na * (ml - 1), d * dri, dli);
    g.arow_skip = ml;
    g.rowmap_p = ml - 1; g.rowmap_s1 = row; g.rowmap_s2 = (long)ml * row; g.rowmap_r0 = 0;
    g.tile_cfg = 1;
    zgemm(st_, g);
    cnt_.n_launch += 1;
    cnt_.heff_flops_skipped += 8.0 * (double)na * dli * d * dri;
  } else {  // X[(a,c)][(j,s)] = L[(a,c)][b] psi[b][(j,s)]
    ZgemmDesc g = zgemm_desc(L + (size_t)a0 * ml * dli, psi, X_.p, na * ml, d * dri, dli);
    zgemm(st_, g);
  }
  timer_end();
  timer_begin(11);
  // Y_a[(i,t)][s] = W2L[(i,t)][(c,j)] X_a[(c,j)][s]
  const double s2_frac = w_stage(&w, 0, w.w2l.p, d, mr, ml, dri, na);
  timer_end();
  timer_begin(12);
  const bool trim = trim_r_ && !sharded && dro == dri && mr > 1;
  if (trim) {
    // R[:, mr-1, :] is the identity: its K block of the contraction is a strided copy of Y, the GEMM runs over the
    // other mr - 1 blocks and adds to it
    const zc one = make_double2(1.0, 0.0);
    copy2d(st_, out + (size_t)a0 * d * dro, dro, Y_.p + (size_t)(mr - 1) * dri, (long)mr * dri, (long)na * d, dro, 0, one, false);
    ZgemmDesc g = zgemm_desc(Y_.p, R, out + (size_t)a0 * d * dro, na * d, dro, (mr - 1) * dri);
    g.lda = (long)mr * dri; g.transB = 1; g.ldb = (long)mr * dri; g.beta = one;
    zgemm(st_, g);
    cnt_.n_launch += 1;
    cnt_.heff_flops_skipped += 8.0 * (double)na * d * dro * dri;
  } else {  // out[(a,i)][r] = Y[(a,i)][(t,s)] R[r][(t,s)]
    ZgemmDesc g = zgemm_desc(Y_.p, R, out + (size_t)a0 * d * dro, na * d, dro, mr * dri);
    g.transB = 1; g.ldb = (long)mr * dri;
    zgemm(st_, g);
  }
  timer_end();
  if (sharded) collective(COLL_ALLGATHER, out, (size_t)dlo * d * dro);
  cnt_.n_launch += 3;
  cnt_.n_heff += 1;
  cnt_.heff_flops += 8.0 * ((double)na * dli * ml * d * dri + (double)na * dri * ml * mr * d * d + (double)na * dro * dri * mr * d);
  cnt_.heff_flops_skipped += 8.0 * (1.0 - s2_frac) * ((double)na * dri * ml * mr * d * d);
  cnt_.heff_stage_flops[0] += 8.0 * (double)na * (triml ? ml - 1 : ml) * dli * d * dri;
  cnt_.heff_stage_flops[1] += 8.0 * s2_frac * ((double)na * dri * ml * mr * d * d);
  cnt_.heff_stage_flops[2] += 8.0 * (double)na * d * dro * (trim ? mr - 1 : mr) * dri;
}

// The apply for an edge-structured core between canonical environments (MpoSite::edge; L[:, 0, :] = R[:, mr-1, :] = 1,
// verified numerically by choose_apply_forms).  All terms with c = 0 see X_0 = psi, all terms with t = mr - 1 see the
// identity on the right, and there are no others:
//   sigma[a,i,r] = sum_{j,t} W[0,i,j,t] T[(a,j)][(r,t)],        T = psi[(a,j)][s] R[(r,t)][s]^T        ("R side")
//                + sum_{c>=1,j} W[c,i,j,mr-1] X[(a,c)][(r,j)],  X = L[(a,c)][b] psiT[b][(r,j)]        ("L side")
// Both are GEMMs of the size of stages S1 / S3 whose 64 x 64 tiles hold whole (j, t) / (c, j) groups and are contracted
// with the d x (d M) reduced core in the epilogue (zgemm_reduce): the M-fold intermediates X and Y of the three-stage
// chain ("must be tiled / fused") are never written.  psiT = psi with its last two indices swapped.
void Engine::heff_apply_edge(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dl, int d, int dr) {
  const int ml = w.ml, mr = w.mr;
  bool first = true;
  double exe = 0.0;
  if (w.edge_has_r) {  // R side: rows (a, j), columns (r, t)
    timer_begin(12);
    ZgemmDesc g = zgemm_desc(psi, R, out, dl * d, dr * mr, dr);
    g.transB = 1; g.ldb = dr;
    g.epi_w = w.w_edge_r.p; g.epi_ldw = (long)d * mr; g.epi_xm = d; g.epi_yn = mr; g.epi_di = d;
    g.epi_wf = w.edge_rf_ok ? w.w_edge_rf.p : nullptr;
    g.epi_su = (long)d * dr; g.epi_sv = 1; g.epi_si = dr; g.epi_acc = 0;
    zgemm_reduce(st_, g);
    timer_end();
    first = false;
    cnt_.n_launch += 1;
    exe += 8.0 * ((double)dl * d * dr * mr * dr + (double)dl * dr * d * d * mr);
    cnt_.heff_stage_flops[2] += 8.0 * ((double)dl * d * dr * mr * dr + (double)dl * dr * d * d * mr);
  }
  if (w.edge_has_l) {
    timer_begin(11);
    transpose_batched(st_, psi, X_.p, d, dr, dr, d, dl, (long)d * dr, (long)d * dr);  // psiT[b][s][j]
    timer_end();
    timer_begin(10);
    // L side: rows (a, c), columns (s, j)
    ZgemmDesc g = zgemm_desc(L, X_.p, out, dl * ml, dr * d, dl);
    g.epi_w = w.w_edge_l.p; g.epi_ldw = (long)ml * d; g.epi_xm = ml; g.epi_yn = d; g.epi_di = d;
    g.epi_wf = w.edge_lf_ok ? w.w_edge_lf.p : nullptr;
    g.epi_su = (long)d * dr; g.epi_sv = 1; g.epi_si = dr; g.epi_acc = first ? 0 : 1;
    zgemm_reduce(st_, g);
    timer_end();
    first = false;
    cnt_.n_launch += 2;
    exe += 8.0 * ((double)dl * ml * dl * d * dr + (double)dl * dr * d * ml * d);
    cnt_.heff_stage_flops[0] += 8.0 * ((double)dl * ml * dl * d * dr + (double)dl * dr * d * ml * d);
  }
  if (first) HIP_CHECK(hipMemsetAsync(out, 0, (size_t)dl * d * dr * sizeof(zc), st_));  // a zero core
  cnt_.n_heff += 1;
  cnt_.n_heff_edge += 1;
  const double alg = 8.0 * ((double)dl * dl * ml * d * dr + (double)dl * dr * ml * mr * d * d + (double)dl * dr * dr * mr * d);
  cnt_.heff_flops += alg;
  cnt_.heff_flops_skipped += alg - exe;
}

// Which forms the applies of the local solve between these blocks take.  The three-stage chain may trim the identity
// blocks L[:, 0, :] and R[:, mr-1, :] (checked from D = 256 on, where one check per site buys 1 / M of stages S1 / S3 in
// every apply); the edge form needs the identity states of both bonds (all blocks checked: two launches, one copy).
void Engine::choose_apply_forms(const zc* Lb, const MpoSite& w, const zc* Rb, int dl, int d, int dr) {
  trim_l_ = trim_r_ = edge_ = false;
  int a0, a1;
  const bool sharded = shard_range(dl, a0, a1);
  const int ml = w.ml, mr = w.mr;
  const bool edge_cand = edge_mode_ != 0 && trim_identity_ && !w.whost.empty() && !sharded && dl >= 32 && dr >= 32 &&
                         zgemm_reduce_ok(d, mr, d) && zgemm_reduce_ok(ml, d, d) && (long)d * dr < (1L << 20) &&
                         // the size rule: the epilogue streams the d x (d M) reduced core once per tile -- cheap beside a
                         // tile's K loop only while d M is small (measured: profiles/r04_edge_apply_ab.txt)
                         // round 5: with the 4 x 4 x 4 epilogue (no padded products at d M = 512) the form also wins where the
                         // W stage is a large share of the chain, i.e. at short bonds: C3 (D = 128) heff -9 %, C4 (D = 1024) +2 %
                         (edge_mode_ > 0 || ((long)d * std::max(ml, mr) <= 64 && (long)dl * dr <= 512L * 512L) ||
                          // (later in round 5: with the cores in fragment order and the unguarded epilogue the form is level with
                          // the chain at C4 too -- 0.02064 against 0.02058 sweeps/s, H_eff frac 0.899 against 0.875, a ninth of the
                          // chain's intermediate traffic -- so shapes that run that variant take it at any bond)
                          ((long)d * std::max(ml, mr) <= 512 && zgemm_reduce_b4_available(st_) != 0 &&
                           ((long)dl * dr <= 256L * 256L || (zgemm_reduce_full_ok(st_, d, mr, d) && zgemm_reduce_full_ok(st_, ml, d, d)))));
  if (edge_cand && w.edge_skip > 0) {  // a core that failed the structure check recently: the plain checks, no look at all blocks
    w.edge_skip -= 1;
    identity_blocks(trim_identity_ && dl >= 256 && ml > 1 ? Lb : nullptr, dl, ml,
                    trim_identity_ && dr >= 256 && mr > 1 ? Rb : nullptr, dr, mr, &trim_l_, &trim_r_);
    return;
  }
  if (!edge_cand) {
    identity_blocks(trim_identity_ && dl >= 256 && ml > 1 ? Lb : nullptr, dl, ml,
                    trim_identity_ && dr >= 256 && mr > 1 ? Rb : nullptr, dr, mr, &trim_l_, &trim_r_);
    return;
  }
  double* dev = reinterpret_cast<double*>(red_.p + RED_MISC);
  zc* lam_dev = red_.p + RED_MISC + 64;  // behind the 128 deviations
  struct IdentRecord { double dev[128]; hzc lam[128]; };
  static_assert(sizeof(IdentRecord) == 128 * 8 + 128 * 16 && sizeof(IdentRecord) <= 4 * NPART * sizeof(zc), "layout of the identity-check record");
  // (the record comes back through the pinned mirror of the reduction area -- read_partials: ~10 us; a copy into pageable
  // host memory followed by a stream synchronisation measured ~120 us of idle GPU per site, 9 % of a C3 sweep)
  const IdentRecord& h = *reinterpret_cast<const IdentRecord*>(h_red_ + RED_MISC);
  // The identity states of a site do not change from sweep to sweep (they follow from the MPO's structure and the
  // canonical form): first only the blocks that were identity multiples last time are looked at (3 of 16 at C5); all of
  // them again when one of those has stopped being one, or when there is no previous answer.
  unsigned long long S = 0, E = 0;
  std::vector<hzc> lam(ml), mu(mr);
  for (int attempt = (w.edge_valid ? 0 : 1); attempt < 2; ++attempt) {
    const unsigned long long ms = attempt == 0 ? w.edge_s : ~0ull, me = attempt == 0 ? w.edge_e : ~0ull;
    HIP_CHECK(hipMemsetAsync(dev, 0, 128 * sizeof(double), st_));  // both sides' deviations: one clear
    ident_deviation_multi(st_, Lb, ml, dl, (long)ml * dl, dl, dev, lam_dev, ms, false);
    ident_deviation_multi(st_, Rb, mr, dr, (long)mr * dr, dr, dev + 64, lam_dev + 64, me, false);
    read_partials(RED_MISC, sizeof(IdentRecord) / sizeof(zc));
    cnt_.n_launch += 2;
    S = E = 0;
    for (int c = 0; c < ml; ++c) { lam[c] = h.lam[c]; if (((ms >> c) & 1ull) && h.dev[c] < 1e-13) S |= 1ull << c; }
    for (int t = 0; t < mr; ++t) { mu[t] = h.lam[64 + t]; if (((me >> t) & 1ull) && h.dev[64 + t] < 1e-13) E |= 1ull << t; }
    if (attempt == 0 && (S != w.edge_s || E != w.edge_e)) continue;  // something changed: look at every block
    break;
  }
  // the trimmed three-stage chain wants the plain identity in state 0 / mr - 1
  trim_l_ = ml > 1 && (S & 1ull) && std::abs(lam[0] - 1.0) < 1e-13;
  trim_r_ = mr > 1 && ((E >> (mr - 1)) & 1ull) && std::abs(mu[mr - 1] - 1.0) < 1e-13;
  for (int c = 0; c < ml; ++c)
    for (int t = 0; t < mr; ++t)
      if (w.nzblk[(size_t)c * mr + t] && !((S >> c) & 1ull) && !((E >> t) & 1ull)) {  // a block between general states
        w.edge_skip = 16;  // not an edge-structured core (or not between canonical blocks): ask again in a while
        return;
      }
  bool same = w.edge_valid && w.edge_s == S && w.edge_e == E;
  if (same) {  // the multiples are +-1 or weights that do not change along a run -- up to the rounding of the block's first
    // diagonal element (1 +- 2e-16 from sweep to sweep): compared to the tolerance of the identity test itself, so that the
    // reduced cores are not rebuilt on the host and uploaded again for every site of every sweep
    for (int c = 0; c < ml && same; ++c) if (((S >> c) & 1ull) && std::abs(w.edge_lam[c] - lam[c]) > 1e-13) same = false;
    for (int t = 0; t < mr && same; ++t) if (((E >> t) & 1ull) && std::abs(w.edge_mu[t] - mu[t]) > 1e-13) same = false;
  }
  if (std::getenv("MITDVP_EDGE_TRACE")) fprintf(stderr, "[mitdvp] edge cores of a site: %s\n", same ? "kept" : "rebuilt");
  if (!same) {
    const hzc* W = w.whost.data();
    std::vector<hzc> wl((size_t)d * ml * d, hzc(0, 0)), wr((size_t)d * d * mr, hzc(0, 0));
    bool has_l = false, has_r = false;
    for (int c = 0; c < ml; ++c)
      for (int t = 0; t < mr; ++t) {
        if (!w.nzblk[(size_t)c * mr + t]) continue;
        const bool in_s = (S >> c) & 1ull;
        const hzc f = in_s ? lam[c] : mu[t];
        if (f == hzc(0.0, 0.0)) continue;  // a zero block of the environment: the term vanishes
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j) {
            const hzc v = f * W[(((size_t)c * d + i) * d + j) * mr + t];
            if (in_s) wr[(size_t)i * d * mr + (size_t)j * mr + t] += v;
            else wl[(size_t)i * ml * d + (size_t)c * d + j] += v;
          }
        (in_s ? has_r : has_l) = true;
      }
    w.w_edge_l.reserve(wl.size());
    w.w_edge_r.reserve(wr.size());
    HIP_CHECK(hipMemcpyAsync(w.w_edge_l.p, wl.data(), wl.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
    HIP_CHECK(hipMemcpyAsync(w.w_edge_r.p, wr.data(), wr.size() * sizeof(zc), hipMemcpyHostToDevice, st_));
    w.w_edge_lf.reserve(wl.size());
    w.w_edge_rf.reserve(wr.size());
    w.edge_lf_ok = zgemm_reduce_pack_core(st_, w.w_edge_l.p, (long)ml * d, d, ml * d, w.w_edge_lf.p);
    w.edge_rf_ok = zgemm_reduce_pack_core(st_, w.w_edge_r.p, (long)d * mr, d, d * mr, w.w_edge_rf.p);
    HIP_CHECK(hipStreamSynchronize(st_));
    w.edge_s = S; w.edge_e = E; w.edge_lam = lam; w.edge_mu = mu;
    w.edge_has_l = has_l; w.edge_has_r = has_r; w.edge_valid = true;
  }
  edge_ = true;
}

void Engine::heff_apply(const zc* L, const MpoSite& w, const zc* R, const zc* psi, zc* out, int dl, int d, int dr,
                        hzc shift) {
  SmallChain sc;
  if (small_ok() && chain_heff(sc, L, w, R, dl, d, dr, false)) {  // one launch, X / Y in LDS
    timer_begin(10);
    small_apply(st_, ss_, sc, psi, out, ss_partials(sc), make_double2(shift.real(), shift.imag()), shift != hzc(0.0, 0.0));
    timer_end();
    cnt_.n_launch += 1;
    cnt_.n_heff += 1;
    cnt_.heff_flops += 8.0 * ((double)dl * dl * w.ml * d * dr + (double)dl * dr * w.ml * w.mr * d * d + (double)dl * dr * dr * w.mr * d);
    return;
  }
  if (edge_) heff_apply_edge(L, w, R, psi, out, dl, d, dr);
  else heff_apply_rect(L, w, R, psi, out, dl, dl, d, dr, dr);
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

// Which MPO-bond states of the two blocks of a bond are multiples of the identity (one look at all blocks: two launches,
// one copy, one synchronisation per bond exponential), the compact copies of the blocks that are not, and the lists of
// scaled copies.  Off (kc_.on = false) when nothing, or too little, can be skipped.
void Engine::keff_prepare(const zc* L, const zc* R, int d1, int d2, int m) {
  kc_.on = false;
  int a0, a1;
  if (!keff_ident_ || !trim_identity_ || d1 != d2 || d1 < 256 || m < 2 || m > 64 || shard_range(d1, a0, a1)) return;
  double* dev = reinterpret_cast<double*>(red_.p + RED_MISC);
  zc* lam_dev = red_.p + RED_MISC + 64;
  struct IdentRecord { double dev[128]; hzc lam[128]; };
  const IdentRecord& h = *reinterpret_cast<const IdentRecord*>(h_red_ + RED_MISC);
  HIP_CHECK(hipMemsetAsync(dev, 0, 128 * sizeof(double), st_));
  ident_deviation_multi(st_, L, m, (long)d1, (long)m * d1, d1, dev, lam_dev, ~0ull, false);
  ident_deviation_multi(st_, R, m, (long)d2, (long)m * d2, d2, dev + 64, lam_dev + 64, ~0ull, false);
  read_partials(RED_MISC, sizeof(IdentRecord) / sizeof(zc));
  cnt_.n_launch += 2;
  BlockList gl{}, gr{};  // blocks gathered into Lc ([E \ S | general]) and Rc ([general | S \ E])
  KeffCompact& k = kc_;
  k.fillS.n = k.accE.n = 0;
  hzc both(0.0, 0.0);
  std::vector<int> onlyE, gen, onlyS;
  for (int c = 0; c < m; ++c) {
    const bool inS = h.dev[c] < 1e-13, inE = h.dev[64 + c] < 1e-13;
    if (inS && inE) both += h.lam[c] * h.lam[64 + c];
    else if (inS) onlyS.push_back(c);
    else if (inE) onlyE.push_back(c);
    else gen.push_back(c);
  }
  const int skipped = 2 * (m - (int)gen.size()) - (int)onlyS.size() - (int)onlyE.size();  // block products saved, of 2 m
  if (skipped * 16 < 2 * m) return;  // less than 1 / 16 of the apply: not worth the extra launches
  k.nE = (int)onlyE.size(); k.nG = (int)gen.size(); k.nS = (int)onlyS.size();
  k.n1 = k.nE + k.nG;
  for (int c : onlyE) gl.idx[gl.n++] = c;
  for (int c : gen) { gl.idx[gl.n++] = c; gr.idx[gr.n++] = c; }
  for (int c : onlyS) gr.idx[gr.n++] = c;
  for (int q = 0; q < k.nS; ++q) { const hzc l = h.lam[onlyS[q]]; k.fillS.f[k.fillS.n++] = make_double2(l.real(), l.imag()); }
  for (int q = 0; q < k.nE; ++q) {
    const hzc mu = h.lam[64 + onlyE[q]];
    k.accE.idx[k.accE.n] = q;  // position of the block in X's row layout
    k.accE.f[k.accE.n++] = make_double2(mu.real(), mu.imag());
  }
  k.both = make_double2(both.real(), both.imag());
  kc_m_ = m;
  k.Lc.reserve((size_t)d1 * std::max(k.n1, 1) * d1);
  k.Rc.reserve((size_t)d2 * std::max(k.nG + k.nS, 1) * d2);
  gather_blocks(st_, k.Lc.p, k.n1, L, m, d1, d1, gl);
  gather_blocks(st_, k.Rc.p, k.nG + k.nS, R, m, d2, d2, gr);
  cnt_.n_launch += 2;
  k.on = true;
}

void Engine::keff_apply_compact(const zc* sig, zc* out, int d1, int d2, hzc shift) {
  const KeffCompact& k = kc_;
  const int nx = k.nE + k.nG + k.nS;
  const long ldx = (long)nx * d2;
  timer_begin(2);
  if (k.n1 > 0) {  // X[a][ci][s] = Lc[(a, ci)][b] sig[b][s], ci over [E \ S | general]
    ZgemmDesc g = zgemm_desc(k.Lc.p, sig, X_.p, d1 * k.n1, d2, d1);
    g.rowmap_p = k.n1; g.rowmap_s1 = d2; g.rowmap_s2 = ldx; g.rowmap_r0 = 0;
    zgemm(st_, g);
    cnt_.n_launch += 1;
  }
  fill_scaled_blocks(st_, X_.p, ldx, k.n1, sig, d1, d2, k.fillS);  // X[a][n1 + q][:] = lam_q sig[a][:]
  if (k.fillS.n) cnt_.n_launch += 1;
  const int n2 = k.nG + k.nS;
  if (n2 > 0) {  // out[a][r] = X[a][(cj, s)] Rc[r][(cj, s)], cj over [general | S \ E]
    ZgemmDesc g = zgemm_desc(X_.p + (size_t)k.nE * d2, k.Rc.p, out, d1, d2, n2 * d2);
    g.lda = ldx; g.transB = 1; g.ldb = (long)n2 * d2;
    zgemm(st_, g);
    cnt_.n_launch += 1;
  } else {
    HIP_CHECK(hipMemsetAsync(out, 0, (size_t)d1 * d2 * sizeof(zc), st_));
  }
  const hzc tot = hzc(k.both.x, k.both.y) + shift;  // the scalar term of the operator rides on the same pass
  accum_scaled_blocks(st_, out, X_.p, ldx, sig, d1, d2, k.accE, make_double2(tot.real(), tot.imag()));
  cnt_.n_launch += 1;
  timer_end();
  cnt_.n_keff += 1;
  // (algorithmic count: all m blocks, as SURVEY 8d F_K)
  cnt_.keff_flops += 8.0 * ((double)d1 * d1 * kc_m_ * d2 + (double)d1 * d2 * d2 * kc_m_);
}

void Engine::keff_apply(const zc* L, const zc* R, const zc* sig, zc* out, int d1, int d2, int m, hzc shift) {
  if (kc_.on) { keff_apply_compact(sig, out, d1, d2, shift); return; }
  SmallChain sc;
  if (small_ok() && chain_keff(sc, L, R, d1, d2, m, false)) {
    timer_begin(2);
    small_apply(st_, ss_, sc, sig, out, ss_partials(sc), make_double2(shift.real(), shift.imag()), shift != hzc(0.0, 0.0));
    timer_end();
    cnt_.n_launch += 1;
    cnt_.n_keff += 1;
    cnt_.keff_flops += 8.0 * ((double)d1 * d1 * m * d2 + (double)d1 * d2 * d2 * m);
    return;
  }
  keff_apply_rect(L, R, sig, out, d1, d1, d2, d2, m);
  if (shift != hzc(0.0, 0.0))
    vec_axpby(st_, out, sig, (long)d1 * d2, make_double2(shift.real(), shift.imag()), make_double2(1.0, 0.0));
}

// env_in (dbi, min, dki), ket tensor Tk (dki, d, dko), bra tensor Tb (dbi, d, dbo),
// W2 ((d*mout) x (min*d)) -> env_out (dbo, mout, dko).  Tb != Tk is the adaptive-rank
// "bra" block (superblock_states_bra, _mps_cls.py:1950-1963).
void Engine::env_update_rect(const zc* env_in, const zc* Tk, const zc* Tb, const zc* w2, zc* env_out, int dbi, int dki,
                             int min_, int d, int dbo, int dko, int mout, const MpoSite* sp, int sp_side) {
  int m0, m1;
  const bool sharded = shard_range(dbi, m0, m1);
  const int nm = m1 - m0;
  timer_begin(1);
  {  // X[(m,p)][(s,j)] = env[(m,p)][n] Tk[n][(s,j)]
    ZgemmDesc g = zgemm_desc(env_in + (size_t)m0 * min_ * dki, Tk, X_.p, nm * min_, d * dko, dki);
    zgemm(st_, g);
  }
  // Y_m[(r,q)][j] = W2[(r,q)][(p,s)] X_m[(p,s)][j]
  (void)w_stage(sp, sp_side, w2, d, mout, min_, dko, nm);
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
                        int mout, const zc* w2e, const MpoSite* sp, int sp_side) {
  SmallChain sc;
  if (w2e && small_ok() && chain_env(sc, T, w2e, din, min_, d, dout, mout)) {
    timer_begin(1);
    small_apply(st_, ss_, sc, env_in, env_out, ss_partials(sc), make_double2(0.0, 0.0), false);
    timer_end();
    cnt_.n_launch += 1;
    cnt_.n_env += 1;
    cnt_.env_flops += 8.0 * ((double)din * din * min_ * d * dout + (double)din * dout * min_ * mout * d * d +
                             (double)din * dout * dout * mout * d);
    return;
  }
  env_update_rect(env_in, T, T, w2, env_out, din, din, min_, d, dout, dout, mout, sp, sp_side);
}

// ---------------------------------------------------------------------------
// gauge moves
// ---------------------------------------------------------------------------
void Engine::gauge_qr_left(const zc* psi, int dl, int d, int dr, zc* A_out, zc* sigma_out) {
  const long n = (long)dl * d * dr;
  HIP_CHECK(hipMemcpyAsync(tmp1_.p, psi, n * sizeof(zc), hipMemcpyDeviceToDevice, st_));
  timer_begin(3);
  long nl = 0;
  qr_thin(st_, tmp1_.p, dl * d, dr, A_out, sigma_out, qrwork_.p, &nl, qr_sync(), qr_hist_, qr_gauge_free_);
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
  qr_thin(st_, tmp1_.p, dr * d, dl, Bt_out, sig2_.p, qrwork_.p, &nl, qr_sync(), qr_hist_, qr_gauge_free_);
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

// The raw (not canonicalised) random tensors of sites [first, first + L_) of an ntot-site chain: shapes and seeds by the
// GLOBAL site index, i.e. what init_random draws for those sites before it canonicalises -- a rank of a site-sharded
// state fills only its own block and the ranks canonicalise in a pipeline (parallel_sites.py).  balance: every tensor is
// multiplied by the largest power of two <= 1 / sqrt(d_l d) (exact in floating point; the right-canonical factors do not
// change, the weight passed on along the chain stays O(1) instead of growing by ~sqrt(D d D) per site).
void Engine::init_random_block(const int* dims, int ntot, int first, int D, uint64_t seed, bool balance) {
  if (D < 1) throw ArgError("bond_dim must be >= 1");
  if (first < 0 || first + L_ > ntot) throw ArgError("init_random_block: the block lies outside the chain");
  auto satprod = [&](int lo, int hi) {
    double p = 1;
    for (int i = lo; i < hi; ++i) { p *= dims[i]; if (p > D) return (long)D + 1; }
    return (long)p;
  };
  for (int q = 0; q < L_; ++q) {
    const int i = first + q;
    if (dims[i] < 1) throw ArgError("bad physical dimension");
    const long left = i == 0 ? 1 : std::min<long>(D, satprod(0, i));
    const long right = i == ntot - 1 ? 1 : std::min<long>(D, satprod(i + 1, ntot));
    const long dc = dims[i];
    dl_[q] = (int)std::min({left, dc * right, (long)D});
    dr_[q] = (int)std::min({left * dc, right, (long)D});
    dd_[q] = dims[i];
    const size_t e = (size_t)dl_[q] * dd_[q] * dr_[q];
    site_[q].reserve(e);
    vec_randn(st_, site_[q].p, (long)e, seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
    if (balance) {
      int ex = 0;
      (void)std::frexp(1.0 / std::sqrt((double)dl_[q] * dd_[q]), &ex);  // 1 / sqrt = m 2^ex, m in [0.5, 1)
      vec_scale(st_, site_[q].p, (long)e, make_double2(std::ldexp(1.0, ex - 1), 0.0));
    }
    gauge_[q] = -1;
  }
  center_ = -1;
  bond_ = -1;
  invalidate_env();
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
    env_update(envR_[p + 1].p, tmp1_.p, w.w2r.p, envR_[p].p, dr_[p], w.mr, dd_[p], dl_[p], w.ml, w.w2er.p, &w, 1);
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
    env_update(envL_[p].p, site_[p].p, w.w2l.p, envL_[p + 1].p, dl_[p], w.ml, dd_[p], dr_[p], w.mr, w.w2el.p, &w, 0);
    envL_ok_[p + 1] = 1;
  }
}

// R[r][m-1][s] == delta_rs to 1e-13 (orthonormality of the tensors right of the site, to rounding)
bool Engine::right_block_is_identity(const zc* R, int dr, int m) {
  bool l = false, r = false;
  identity_blocks(nullptr, 0, 0, R, dr, m, &l, &r);
  return r;
}

// L[a][0][b] == delta_ab to 1e-13 (the tensors left of the site are left-canonical)
bool Engine::left_block_is_identity(const zc* L, int dl, int m) {
  bool l = false, r = false;
  identity_blocks(L, dl, m, nullptr, 0, 0, &l, &r);
  return l;
}

// both checks of a site with ONE host synchronisation (two small reduction launches, one 16-byte copy); a null block
// is not checked
void Engine::identity_blocks(const zc* L, int dl, int ml, const zc* R, int dr, int mr, bool* left, bool* right) {
  double* dev = reinterpret_cast<double*>(red_.p + RED_MISC);
  const double* h = reinterpret_cast<const double*>(h_red_ + RED_MISC);
  if (L) ident_deviation(st_, L, (long)ml * dl, dl, dev);
  if (R) ident_deviation(st_, R + (size_t)(mr - 1) * dr, (long)mr * dr, dr, dev + 1);
  if (L || R) read_partials(RED_MISC, 1);
  *left = L && h[0] < 1e-13;
  *right = R && h[1] < 1e-13;
}

void Engine::local_site_exp(int p, double dt) {
  const MpoSite& w = mpo(0, p);
  if (dd_[p] != w.d) throw ArgError("MPO physical dimension differs from the site tensor's");
  if (small_site_exp(p, dt)) return;  // one launch, no host round trip
  const int dl = dl_[p], d = dd_[p], dr = dr_[p];
  const zc* Lb = envL_[p].p;
  const zc* Rb = envR_[p + 1].p;
  const hzc shift = op(0).shift;
  auto mv = [&](const zc* in, zc* out) { heff_apply(Lb, w, Rb, in, out, dl, d, dr, shift); };
  // large bonds: one check per site (two tiny launches and a synchronisation) buys 1 / M_r of stage S3 in every apply
  choose_apply_forms(Lb, w, Rb, dl, d, dr);
  struct Reset { bool& f; bool& g; bool& e; ~Reset() { f = false; g = false; e = false; } } reset{trim_r_, trim_l_, edge_};
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
    ss_check();
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
      qr_thin(st_, site_[p].p, dl * d, dr, spare.p, sig_.p, qrwork_.p, &nl, qr_sync(), qr_hist_, qr_gauge_free_);
      timer_end();
      cnt_.n_launch += nl; cnt_.n_qr += 1;
      cnt_.qr_flops += 4.0 * (4.0 * (double)dl * d * dr * dr - 4.0 * (double)dr * dr * dr / 3.0);
      std::swap(site_[p], spare);
      gauge_[p] = MITDVP_GAUGE_A;
      // renormalize_op_psite: L_{p+1}
      envL_[p + 1] = pool_get((size_t)dr * w.mr * dr);
      env_update(envL_[p].p, site_[p].p, w.w2l.p, envL_[p + 1].p, dl, w.ml, d, dr, w.mr, w.w2el.p, &w, 0);
      envL_ok_[p + 1] = 1;
      // exp(+i K dt/2) on the bond matrix
      const zc* Lb = envL_[p + 1].p;
      const zc* Rb = envR_[p + 1].p;
      const int m = w.mr;
      auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, dr, dr, m, shift); };
      if (cfg.relax != 2 && !small_bond_exp(p, Lb, Rb, dr, m, dt)) {  // improved relaxation leaves the bond matrix alone (_mps_cls.py:1159-1160)
        keff_prepare(Lb, Rb, dr, dr, m);  // identity states of the two blocks: skipped in every apply of this solve
        struct Off { bool& f; ~Off() { f = false; } } off{kc_.on};
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
      env_update(envR_[p + 1].p, tmp2_.p, w.w2r.p, envR_[p].p, dr, w.mr, d, dl, w.ml, w.w2er.p, &w, 1);
      envR_ok_[p] = 1;
      const zc* Lb = envL_[p].p;
      const zc* Rb = envR_[p].p;
      const int m = w.ml;
      auto mk = [&](const zc* in, zc* out) { keff_apply(Lb, Rb, in, out, dl, dl, m, shift); };
      if (cfg.relax != 2 && !small_bond_exp(p, Lb, Rb, dl, m, dt)) {
        keff_prepare(Lb, Rb, dl, dl, m);
        struct Off { bool& f; ~Off() { f = false; } } off{kc_.on};
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
  ss_check();  // the one host synchronisation of a small-bond sweep: errors raised on the device
}

void Engine::step(double dt) {
  sweep(dt, true);
  apply_gates();  // Model(one_gate_to_apply=...), _mps_cls.py:489-490 (reorth_center = nsite - 1)
  apply_kraus();  // Model(kraus_op=...), :491-492
  sweep(dt, false);
}

}  // namespace mitdvp
