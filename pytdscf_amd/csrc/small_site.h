// small_site.h -- the small-bond kernel family (see small_site.hip): one launch per H_eff / K_eff
// apply, per environment update and per LOCAL EXPONENTIAL (short-iterative Lanczos / Arnoldi with the
// k x k projected exponential and the convergence test on the device, no host round trip).
#pragma once
#include "common.h"
#include "vecops.h"

namespace mitdvp {

// One slab-contraction chain
//   out[a][i][r] = sum_{b,c,j,t,s} A(a;c,b) * B(b,j,s) * W2[(i,t)][(c,j)] * R(r,t,s)
// with strided operand views, so that the H_eff apply, the K_eff apply (no W stage) and the
// environment update (roles of the operands permuted) are the same kernel body:
//   stage 1  X[(c,j)][s] = sum_b A[c][b] B[b][(j,s)]         (nc x nj*ns, K = nb)
//   stage 2  Y[(i,t)][s] = sum_(c,j) W2[(i,t)][(c,j)] X[(c,j)][s]
//   stage 3  out[i][r]   = sum_(t,s) Y[i][(t,s)] R[r][(t,s)]
// One workgroup per (slab a, chunk of s); X and Y live in LDS only.
struct SmallChain {
  const zc* A; long sAa, sAc, sAb; int conjA;  // A(a;c,b) = A[a*sAa + c*sAc + b*sAb]
  const zc* R; long sRr, sRt, sRs;             // R(r,t,s) = R[r*sRr + t*sRt + s*sRs]
  const zc* W2;                                // (ni*nt) x (nc*nj) row-major; nullptr: no W stage (ni = nj = 1, nt = nc)
  long sBb, sBj, sBs;                          // vector operand B(b,j,s) = v[b*sBb + j*sBj + s*sBs]
  int na, nb, nc, nj, ni, nt, ns, nr;
  int nsc, cs;                                 // chunks over s and their width (nsc * cs >= ns)
  // slabs per workgroup (0 / 1: one): an engine confined to a slice of the chip (ensemble mode) has fewer compute units
  // than the chain has slabs; a workgroup then walks over spw consecutive slabs of its chunk (R, W and the stage-1 operand
  // are shared by them).  a_resident: all spw slabs' A fit in LDS beside the rest (else A is re-staged per slab).
  int spw, a_resident;
};

enum { SS_MODE_APPLY = 0, SS_MODE_EXP = 1 };
enum { SS_OK = 0, SS_ENOTCONV = 1, SS_ETIMEOUT = 2, SS_EZERO = 3 };

// device-resident state shared by all small-site launches of one engine
struct SmallSync {
  unsigned* words = nullptr;  // [2]: abort flag; [3]: sticky error code
  double* slots = nullptr;    // granule exchange area (two buffers)
  long long* stats = nullptr; // [0] applies inside site exponentials, [1] inside bond exponentials, [2], [3] their flops
  int* kprev = nullptr;       // per-site Krylov iteration memory (device copy)
  unsigned launches = 0;      // launch sequence number (exchange tags are unique per launch)
  int max_grid = 0;           // compute units this engine may fill with one persistent launch (0: the device's)
  bool partitioned = false;   // the engine's stream owns its compute units (CU mask): no admission control needed
};

struct SmallExp {
  int integrator;       // MITDVP_LANCZOS / MITDVP_ARNOLDI
  int variant;          // Lanczos alpha: 0 reference <v0|H v_l>, 1 orthodox <v_l|H v_l>
  int conserve_norm;
  int max_krylov;
  double thresh;
  double scale_re, scale_im;
  int site;             // index into kprev
  int stat_slot;        // 0: site exponential, 1: bond exponential
  long long flops_per_apply;  // algorithmic flops of one apply, summed on the device into stats[2 + stat_slot]
};

// LDS bytes the chain needs (0 when it does not fit the small family)
size_t small_chain_lds(const SmallChain& c, bool exp_mode);
// choose nsc / cs for a chain whose other fields are set; false when the chain does not qualify
bool small_chain_plan(SmallChain& c, bool exp_mode, int n_cu);

// Persistent launches (kernels whose workgroups exchange data inside the launch and must all be resident, one per
// compute unit) of several engines on one GPU are admitted against a per-device budget of compute units while more
// than one engine uses them (small_site.hip): construct around the launch with the launch's grid size.
// persistent_register(+1 / -1): an engine starts / stops using the family on the current device.
void persistent_register(int delta);
class PersistentLaunch {
 public:
  PersistentLaunch(hipStream_t st, int grid, bool partitioned = false);
  ~PersistentLaunch();
  PersistentLaunch(const PersistentLaunch&) = delete;
  PersistentLaunch& operator=(const PersistentLaunch&) = delete;

 private:
  hipStream_t st_;
  int slot_;
  int grid_;
  bool chained_ = false;
};

void small_sync_alloc(SmallSync& s, int nsite, hipStream_t st);
void small_sync_free(SmallSync& s);

// out = chain(v) (+ shift * v when v and out have the same length)
void small_apply(hipStream_t st, SmallSync& sy, const SmallChain& c, const zc* v, zc* out, zc* partials, zc shift,
                 bool add_shift);
// x <- exp(scale * chain) x, Krylov basis in `basis` ((MAXK) x N), partial buffers in `partials` (nsc x N)
void small_exp(hipStream_t st, SmallSync& sy, const SmallChain& c, const SmallExp& e, zc* x, zc* basis, zc* partials,
               zc shift);

}  // namespace mitdvp
