/*
 * mitdvp.h -- C ABI of the MI355X-native one-site TDVP sweep engine.
 *
 * This is the drop-in boundary for PyTDSCF's hot path.  Every entry point
 * names the reference interface it replaces (paths relative to
 * /root/reference/pytdscf).  The coarse seam is
 *     MPSCoef.propagate(stepsize_au, ints_spf, matH)      _mps_cls.py:452-503
 * called once per time step from WFunc.propagate_SM        wavefunction.py:408-412
 * plus the observables MPSCoef.expectation / autocorr / norm / pop_states
 * (_mps_cls.py:540-716, wavefunction.py:226-257).
 *
 * Conventions
 *   - all tensors are complex128, C order, passed as interleaved (re, im)
 *     doubles; host buffers are copied in/out, the handle owns device memory;
 *   - site tensor psi[b][j][s]      shape (D_l, d, D_r)   _site_cls.py:27-60
 *   - MPO core    W[c][i][j][t]     shape (M_l, d, d, M_r) _mpo_cls.py:166-198
 *   - every function returns 0 on success, a negative MITDVP_E* code on error;
 *     mitdvp_last_error() gives the message.  The Python shell maps
 *     MITDVP_ENOTCONV to ValueError like _integrator.py:430,653.
 *   - a handle is bound to one GPU and one HIP stream and is not thread safe.
 */
#ifndef MITDVP_H
#define MITDVP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mitdvp_engine mitdvp_engine;

enum {
  MITDVP_OK = 0,
  MITDVP_EINVAL = -1,   /* bad argument / shape                        */
  MITDVP_EHIP = -2,     /* HIP runtime error                           */
  MITDVP_ENOTCONV = -3, /* Krylov not converged in 20 vectors          */
  MITDVP_ESTATE = -4,   /* call sequence error (e.g. missing tensors)  */
  MITDVP_ENOMEM = -5
};

enum { MITDVP_LANCZOS = 0, MITDVP_ARNOLDI = 1 };
enum { MITDVP_GAUGE_C = -1 /* no gauge */, MITDVP_GAUGE_PSI = 0, MITDVP_GAUGE_A = 1, MITDVP_GAUGE_B = 2 };

/* const.set_runtype(...) flags that reach the sweep (_const_cls.py:102-252). */
typedef struct {
  int nsite;            /* chain length L                                   */
  int device;           /* HIP device ordinal                               */
  int integrator;       /* MITDVP_LANCZOS | MITDVP_ARNOLDI  (:integrator)   */
  int conserve_norm;    /* const.conserve_norm, _integrator.py:189-213      */
  int relax;            /* 0 real time; 1 imaginary time (const.doRelax = True);
                           2 improved relaxation (doRelax = "improved": Lanczos
                           ground state of H_eff per site, _integrator.py:74-138) */
  double thresh;        /* const.thresh_exp, default 1e-9                   */
  int max_krylov;       /* 20, _integrator.py:182                           */
  int lanczos_variant;  /* 0: reference alpha_l=<v0|H|v_l> (_integrator.py:556)
                           1: orthodox alpha_l=<v_l|H|v_l>                  */
  int max_diag_krylov;  /* Krylov vectors kept by the improved-relaxation solver
                           (0 = default 64; the reference allows up to 3000)   */
  int cu_first, cu_count; /* > 0: the engine's stream is confined to cu_count compute units starting at cu_first
                             (hipExtStreamCreateWithCUMask; unit u = CU u / 8 of XCD u % 8, so a range of 8 k units is
                             k CUs on every XCD): several engines of one process with DISJOINT ranges run side by side
                             without ever competing for a compute unit -- an ensemble of trajectories in the small-bond
                             regime (mitdvp_ensemble_step).  0 = the whole device.                                     */
  int reserved[5];
} mitdvp_config;

/* -- lifetime ----------------------------------------------------------
 * Threading: a handle owns one HIP stream and all its device memory and must be driven by one host
 * thread at a time; different handles are independent and may be driven concurrently from different
 * threads (an ensemble of trajectories on one GPU). */
int mitdvp_create(const mitdvp_config* cfg, mitdvp_engine** out);
void mitdvp_destroy(mitdvp_engine* h);
const char* mitdvp_last_error(const mitdvp_engine* h); /* h may be NULL */
const char* mitdvp_version(void);
/* Number of HIP devices visible to this process (0 without a GPU; no engine needed).  Lets a launcher map
 * ranks to devices (rank % count) without importing a second GPU runtime. */
int mitdvp_device_count(int* count);
int mitdvp_device_cu_count(int device, int* count); /* compute units of a device (256 on MI355X): what cu_first / cu_count divide */
/* hipDeviceSynchronize on `device`: the fence bench.py puts on both sides of its timed region (the engine's
 * stream is private, so a foreign runtime's "current stream" synchronise would not see its work). */
int mitdvp_device_sync(int device);

/* -- state: superblock_states[0][isite] (SiteCoef), _site_cls.py:27-60 --- */
int mitdvp_set_site(mitdvp_engine* h, int isite, const double* reim, int l, int n, int r, int gauge);
int mitdvp_get_site_shape(mitdvp_engine* h, int isite, int* l, int* n, int* r, int* gauge);
int mitdvp_get_site(mitdvp_engine* h, int isite, double* reim_out);
/* Device-side full-rank random MPS with LatticeInfo.get_bond_dim bond
 * dimensions (_mps_cls.py:2616-2631), canonicalised right->left like
 * alloc_superblock_random (:2684-2699).  For workloads too large to stage
 * through the host. */
int mitdvp_init_random(mitdvp_engine* h, const int* dims, int bond_dim, uint64_t seed);
/* The raw (not canonicalised) tensors mitdvp_init_random draws for sites [first, first + nsite) of an nsite_total-site
 * chain (shapes and seeds follow the GLOBAL site index; h holds nsite sites): one block of a site-range sharded state
 * (MPSCoefParallel keeps one block per rank, _mps_parallel.py:64-118) without the whole chain on every rank; the ranks
 * then canonicalise in a pipeline with mitdvp_split_center / mitdvp_absorb_bond.  balance != 0: each tensor times the
 * largest power of two <= 1 / sqrt(d_l d), so the weight handed along the chain stays O(1). */
int mitdvp_init_random_block(mitdvp_engine* h, const int* dims, int nsite_total, int first, int bond_dim, uint64_t seed, int balance);
/* alloc_superblock_random's C2sigmaB sweep for tensors given by set_site
 * with gauge "C": site 0 becomes "Psi", scaled to norm `scale`; scale <= 0 keeps
 * the state's own normalisation (Liouville space: trace-normalised start,
 * _mps_cls.py:2695-2699). */
int mitdvp_canonicalize(mitdvp_engine* h, double scale);

/* -- operators: TensorHamiltonian.mpo[0][0] after reduction to one
 *    full-chain 4-leg MPO (hamiltonian_cls.py:618-752, _mpo_cls.py:44-234).
 *    op_id 0 is the Hamiltonian used by mitdvp_step; others are observables. */
int mitdvp_set_mpo_core(mitdvp_engine* h, int op_id, int isite, const double* reim,
                        int ml, int d_out, int d_in, int mr);
/* coupleJ[0][0] * ovlp term (_contraction.py:1200-1216): adds shift*psi. */
int mitdvp_set_shift(mitdvp_engine* h, int op_id, double re, double im);

/* -- the hot path ------------------------------------------------------- */
/* MPSCoef.propagate: forward + backward half-sweep with dt/2 each
 * (_mps_cls.py:482-500).  First call builds all right environments
 * (:835-843); the cache is reused afterwards (:848-861). */
int mitdvp_step(mitdvp_engine* h, double dt_au);
/* nsteps time steps of n independent engines at once (independent trajectories / replicas of one model: SURVEY 7 step 6,
 * 8e "fallback"; the reference averages trajectories in a Python loop, tests/test_mixedstate.py:269-308): every engine is
 * driven by its own host thread inside this call.  Meant for engines created with disjoint cu_first / cu_count ranges (see
 * mitdvp_config): their launches then overlap on the chip without any admission control.  Returns the first non-zero
 * status (that engine's mitdvp_last_error holds the message); statuses[i] (may be NULL) receives each engine's own. */
int mitdvp_ensemble_step(mitdvp_engine** hs, int n, double dt_au, int nsteps, int* statuses);
/* propagate_along_sweep (_mps_cls.py:798-1014), one direction only. */
int mitdvp_sweep(mitdvp_engine* h, double dt_au, int forward);
/* op_sys_sites = None (_mps_cls.py:2311,2371,2417, wavefunction.py:64-65). */
int mitdvp_invalidate_env(mitdvp_engine* h);

/* -- multi-GPU: bond-sharded (tensor-parallel) applies ------------------- */
/* One process per GPU, every rank holds the replicated state and calls the
 * same sequence of entry points.  H_eff / K_eff applies and environment
 * updates are computed for this rank's slice of the bra-side bond index and
 * combined by ONE collective each, issued through `fn` (RCCL via
 * torch.distributed in pytdscf_amd/dist.py; gloo in the tests):
 *   op 0: in-place all-gather  (rank r owns bytes [r*nbytes/N, (r+1)*nbytes/N))
 *   op 1: in-place all-reduce  (sum, float64)
 * `fn` is called with the engine's stream drained and must return after the
 * collective has completed on the device.  The reference's only parallel path
 * is the approximate mpi4py real-space scheme (_mps_parallel.py:106-268); this
 * one is exact (same results as 1 GPU up to summation order). */
typedef int (*mitdvp_collective_fn)(void* user, int op, void* dev_ptr, size_t nbytes);
int mitdvp_set_parallel(mitdvp_engine* h, int nranks, int rank, mitdvp_collective_fn fn, void* user);

/* The same bond-sharded mode with the collectives issued by the library itself: RCCL
 * (librccl, resolved with dlopen on first use) on the engine's own HIP stream, stream-ordered,
 * no host synchronisation per collective.  One rank creates the 128-byte ncclUniqueId with
 * mitdvp_rccl_unique_id and distributes it out of band (torch.distributed, MPI, a file);
 * every rank then calls mitdvp_set_parallel_rccl.  mitdvp_rccl_selftest runs one all-gather and
 * one all-reduce on a small device buffer and returns the number of wrong values. */
int mitdvp_rccl_unique_id(char out[128]);
int mitdvp_set_parallel_rccl(mitdvp_engine* h, int nranks, int rank, const char id[128]);
int mitdvp_rccl_selftest(mitdvp_engine* h, int* mismatches);

/* -- several electronic states (MPS-SM, nstate > 1) -------------------------
 * The reference keeps one MPS per electronic state (superblock_states[istate][isite],
 * _mps_cls.py:84-118; the SURVEY's istate / ibra / iket arguments) and one MPO block plus one
 * scalar per (bra, ket) state pair (TensorHamiltonian.mpo[i][j], .coupleJ[i][j],
 * hamiltonian_cls.py:618-752).  The local solves act on all states' centre tensors stacked into
 * one Krylov vector (SplitStack, _contraction.py:479-608; multiplyH_MPS_direct_MPO.dot, :1182-1243),
 * environment blocks are kept per state pair (renormalize_op_psite, _mps_mpo.py:421-696) and the
 * gauge move is one QR per state (_mps_cls.py:1798-1850).  mitdvp_ms_configure switches a handle
 * to this mode; states share the physical dimensions, bond dimensions may differ per state;
 * every MPO block spans all sites (4-leg cores).  mitdvp_krylov_stats and mitdvp_counters apply
 * unchanged.  Not available in this mode: adaptive bonds, gates, Kraus maps (single-state only in
 * the reference too).  mitdvp_set_parallel* shards this mode like the single-state one. */
int mitdvp_ms_configure(mitdvp_engine* h, int nstate);
int mitdvp_ms_set_site(mitdvp_engine* h, int istate, int isite, const double* reim, int l, int n, int r, int gauge);
int mitdvp_ms_get_site_shape(mitdvp_engine* h, int istate, int isite, int* l, int* n, int* r, int* gauge);
int mitdvp_ms_get_site(mitdvp_engine* h, int istate, int isite, double* out);
/* right-to-left QR of one state, site 0 scaled to `scale` = sqrt(weight of the state) >= 0
 * (alloc_superblock_random, _mps_cls.py:2684-2699; _mps_mpo.py:88-94) */
int mitdvp_ms_canonicalize(mitdvp_engine* h, int istate, double scale);
int mitdvp_ms_set_mpo_core(mitdvp_engine* h, int op_id, int ibra, int iket, int isite, const double* reim, int ml, int d_out,
                           int d_in, int mr);
int mitdvp_ms_set_coupleJ(mitdvp_engine* h, int op_id, int ibra, int iket, double re, double im);
int mitdvp_ms_step(mitdvp_engine* h, double dt_au);                   /* MPSCoef.propagate, _mps_cls.py:452-503 */
int mitdvp_ms_expect(mitdvp_engine* h, int op_id, double out[2]);     /* sum over state pairs, _mps_cls.py:540-612 */
int mitdvp_ms_autocorr(mitdvp_engine* h, double out[2]);              /* sum over states, wavefunction.py:226-257 */
/* Simulator.operate with several states (apply_dipole_along_sweep, _mps_cls.py:718-796): every state must
 * be fed by at least one block or scalar term (the reference fails otherwise, _contraction.py:555) */
int mitdvp_ms_operate(mitdvp_engine* h, int op_id, int maxstep, double conv_tol, double* norm_out, int* iters_out);
int mitdvp_ms_pops(mitdvp_engine* h, double* out /* [nstate] */);      /* pop_states, _mps_cls.py:682-703 */

/* -- one block of a site-range sharded chain ----------------------------------
 * Real-space parallel one-site TDVP (MPSCoefParallel.propagate, _mps_parallel.py:106-268): every rank owns a
 * contiguous range of sites as an engine whose outer bonds are wider than 1.  The environment blocks at its
 * two ends come from the neighbours (send_op_block / recv_op_block, :1610-1635), and a half-sweep is driven
 * through the pieces of propagate_along_sweep (_mps_cls.py:798-1014) so that the end site can be left to the
 * joint update of two ranks (skip_end_site, :876-877; propagate_joint_two_sites, _mps_parallel.py:270-470).
 * Host side: pytdscf_amd/parallel_sites.py.
 *   side 0: block left of site 0, L[a][c][b] (d, m, d); side 1: block right of the last site, R[r][t][s]. */
int mitdvp_set_boundary_env(mitdvp_engine* h, int side, const double* reim, int d, int m);
/* new values for a site tensor of unchanged shape; the environment cache is kept (mitdvp_set_site drops it) */
int mitdvp_replace_site(mitdvp_engine* h, int isite, const double* reim, int gauge);
/* environment block left (side 0) / right (side 1) of bond `bond` (the bond left of site `bond`); reim_out may be
 * NULL to query the shape (d, m, d) */
int mitdvp_get_env(mitdvp_engine* h, int side, int bond, double* reim_out, int* d, int* m);
/* construct_op_sites (_mps_cls.py:1738-1796): all blocks on one side of the centre (0: left, 1: right) */
int mitdvp_build_envs(mitdvp_engine* h, int side);
int mitdvp_site_exp(mitdvp_engine* h, double dt_au);      /* exp_superH_propagation_direct, _mps_cls.py:1016-1100 */
/* ONE H_eff apply at the centre site, sigma = H_eff x (multiplyH_MPS_direct_MPO.dot, _contraction.py:1182-1243), issued
 * exactly as the local exponential issues its applies: identity blocks of the environments short-circuited when the
 * numerical check finds them (the reference's eye shortcut, _mps_mpo.py:510-523), zero blocks of the MPO core skipped.
 * reim_in NULL: x = the centre tensor.  *flags (may be NULL): bit 0 first stage trimmed, bit 1 third stage trimmed,
 * bit 2 block-sparse W stage, bit 3 the one-launch small-bond kernel, bit 4 the two-product form of an edge-structured
 * core (reducing epilogue, no intermediates; then bits 0-2 are clear).  For parity tests of the kernels a sweep runs. */
int mitdvp_heff_apply_center(mitdvp_engine* h, const double* reim_in, double* reim_out, int* flags);
/* trans_next_psite_AsigmaB (:1798-1850): centre -> A sigma (forward) or sigma B; the block through the site is built,
 * sigma stays in the engine as the pending bond matrix */
int mitdvp_split_center(mitdvp_engine* h, int forward);
int mitdvp_bond_exp(mitdvp_engine* h, double dt_au);      /* exp_superK_propagation_direct, :1102-1170 */
int mitdvp_absorb_bond(mitdvp_engine* h, int forward);    /* trans_next_psite_APsiB, :1172-1206 */
int mitdvp_get_bond(mitdvp_engine* h, double* reim_out, int* dim);   /* the pending bond matrix (joint_sigvec) */
int mitdvp_set_bond(mitdvp_engine* h, int bond, const double* reim, int dim);
/* Where the tensor arguments of mitdvp_set_site / replace_site / set_boundary_env / set_bond / fold_block (sources) and
 * mitdvp_get_site / get_env / get_bond / fold_block (destinations) live: MITDVP_POINTER_HOST (default) or
 * MITDVP_POINTER_DEVICE = device memory on the handle's GPU (the halo messages of the site-sharded sweep go from
 * engine to RCCL and back without touching the host: the reference pickles NumPy arrays through mpi4py,
 * _mps_parallel.py:541-807).  Device operands are copied by a kernel on the handle's stream, so they may have been
 * allocated by another HIP runtime instance of the process (torch's); every call returns after the copy completed. */
#define MITDVP_POINTER_HOST 0
#define MITDVP_POINTER_DEVICE 1
int mitdvp_set_pointer_mode(mitdvp_engine* h, int mode);
/* Observables of a site-sharded state without gathering it (MPSCoefParallel.ovlp, _mps_parallel.py:855-983;
 * expectation, :1210-1302): a boundary block is carried through ALL sites of this engine.  from_left: reim_in sits
 * left of site 0, reim_out right of the last site (shape from the last site / MPO core); else the other way round.
 * op_id >= 0: environment block (d, m, d) of that operator, bra conjugated; op_id < 0: transfer block (d, 1, d),
 * conj_bra = 0 gives <Psi*|Psi> (the autocorrelation of the t/2 trick). */
int mitdvp_fold_block(mitdvp_engine* h, int op_id, int conj_bra, int from_left, const double* reim_in, int d, int m,
                      double* reim_out);
/* the same through the sites [first, first + count) only (count = 0: the block is copied) */
int mitdvp_fold_block_range(mitdvp_engine* h, int op_id, int conj_bra, int from_left, int first, int count, const double* reim_in,
                            int d, int m, double* reim_out);
/* Reduced density of ONE site of a site-sharded state (MPSCoefParallel.get_reduced_densities, _mps_parallel.py:1035-1208):
 * rho[j][j'] from the site tensor and the transfer blocks of everything left / right of it, left[bra][ket] (D_l x D_l),
 * right[bra][ket] (D_r x D_r) as mitdvp_fold_block (op_id < 0, conj_bra = 1) produces them.  reim_out: d x d, host. */
int mitdvp_site_rdm_blocks(mitdvp_engine* h, int isite, const double* left, const double* right, double* reim_out);

/* -- one RANK of the site-range sharded sweep, driven natively -------------------------------------
 * MPSCoefParallel.propagate (_mps_parallel.py:106-268) as ONE call per time step: the block's two half-sweeps, the
 * joint update of the two sites facing each other across a rank boundary (propagate_joint_two_sites, :270-470:
 * psi_L X^+ psi_R -> A X' B through the pseudo-inverse of the joint matrix, multiply_sigvec_pinv _site_cls.py:709-754)
 * and the neighbour messages around it -- the reference's mpi4py send / recv pairs send_Psi_to_left (:698-707),
 * send_op_sys_to_left (:761-807), send_B_to_right (:728-740), send_joint_sigvec_to_right (:541-597),
 * send_op_sys_to_right (:612-628) -- as grouped ncclSend / ncclRecv of device buffers between chain neighbours on the
 * block engine's stream (RCCL over xGMI, librccl resolved with dlopen).  A shard owns two engines: its block
 * (nsite_block sites, outer bonds wider than 1) and, except on the last rank, the two-site engine of the junction to
 * its right; mitdvp_shard_engine hands them out (borrowed: never mitdvp_destroy them) so that set-up -- MPO cores,
 * site tensors, boundary blocks, mitdvp_build_envs -- uses the entry points above.  State between steps as in the
 * reference's diagram (:115-122): even ranks [Psi B .. B], odd ranks [A .. A Psi], every block's junction-side end
 * site carrying the weight of the joint matrix X (held by the LEFT rank of each junction: joint_sigvec_not_pinv). */
typedef struct mitdvp_shard mitdvp_shard;
/* dr_next: right bond dimension of the first site of rank + 1 (ignored on the last rank) */
int mitdvp_shard_create(const mitdvp_config* cfg, int rank, int world, int nsite_block, int dr_next, mitdvp_shard** out);
void mitdvp_shard_destroy(mitdvp_shard* h);
const char* mitdvp_shard_last_error(const mitdvp_shard* h); /* h may be NULL */
int mitdvp_shard_engine(mitdvp_shard* h, int which /* 0 block, 1 junction engine, 2 left junction engine (pair mode) */, mitdvp_engine** out);
/* the reference's treatment of small singular values at a junction: regularize != 0 lifts singular values below
 * SQRT_EPSRHO = 1e-4 to s + eps exp(-s / eps) -- in the (D_l D_r x d) unfolding of the left junction site before its
 * QR (SiteCoef.gauge_trf(regularize=True), _site_cls.py:207-246) and in the new joint matrix; p_svd >= 0 replaces
 * the new joint matrix by truncate_sigvec(p = p_svd, keepdim = True) (:586-690: SVD, cumulative-weight cut, zeros on
 * the cut values, A <- A U, B <- Vh B, blocks rebuilt).  The reference runs with both (regularize = 1, p_svd =
 * const.p_svd, default 1e-7; _mps_parallel.py:369, :437-444); defaults here: 0, -1 (off). */
int mitdvp_shard_set_options(mitdvp_shard* h, int regularize, double p_svd);
/* Pair mode (call on EVERY rank before the first step, after the transport is attached): both ranks of a junction run
 * its update on identical copies of the two facing sites, bond-sharded over the pair -- every apply and environment
 * update of the two-site engine contracts half of the bra-side bond index on each GPU (the engine's exact tensor
 * parallelism), combined by two-rank all-gathers / all-reduces over the same point-to-point transport -- instead of
 * the right rank waiting while the left one works (in the reference the partner of propagate_joint_two_sites sits in
 * comm.recv, _mps_parallel.py:196-203).  A rank > 0 gets a second two-site engine for the junction to its LEFT
 * (mitdvp_shard_engine(h, 2, ...): give it the MPO cores of sites (first - 1, first)); dl_prev = left bond dimension of
 * the left neighbour's last site (ignored on rank 0).  Same results as the default mode to rounding. */
int mitdvp_shard_enable_pair(mitdvp_shard* h, int dl_prev);
int mitdvp_shard_set_joint(mitdvp_shard* h, const double* reim, int dim);       /* host, (dim, dim) */
int mitdvp_shard_get_joint(mitdvp_shard* h, double* reim_out, int* dim);        /* reim_out may be NULL */
/* Transport.  Production: every rank calls mitdvp_shard_attach_rccl with the 128-byte id one rank got from
 * mitdvp_rccl_unique_id (distributed out of band), one rank per GPU.  Ranks SHARING a GPU (one-GPU test boxes; RCCL
 * refuses two ranks on one device) install a callback instead: op 0 sends / op 1 receives nbytes of host memory
 * to / from rank `peer`, blocking, 0 on success; the shard stages the device buffers through it. */
typedef int (*mitdvp_p2p_fn)(void* user, int op, int peer, void* host_buf, size_t nbytes);
int mitdvp_shard_set_transport(mitdvp_shard* h, mitdvp_p2p_fn fn, void* user);
int mitdvp_shard_attach_rccl(mitdvp_shard* h, const char id[128]);
/* neighbour ping over every junction, both directions (collective over the ranks); number of wrong values */
int mitdvp_shard_selftest(mitdvp_shard* h, int* mismatches);
/* one grouped ncclSend + ncclRecv of `elems` complex numbers from this rank to itself (needs attach_rccl) */
int mitdvp_shard_self_sendrecv(mitdvp_shard* h, size_t elems, int* mismatches);
int mitdvp_shard_step(mitdvp_shard* h, double dt_au);                            /* MPSCoefParallel.propagate */
/* the pieces of a step (tests, other schedules): propagate_along_sweep over the block with skip_end_site (:147-175),
 * and the joint updates of the junctions whose LEFT rank has parity `parity` (0: (3)->(4), 1: (2)->(1) of :115-122) */
int mitdvp_shard_sweep(mitdvp_shard* h, double dt_au, int forward, int skip_end);
int mitdvp_shard_junctions(mitdvp_shard* h, double dt_au, int parity);
int mitdvp_shard_traffic(mitdvp_shard* h, double* bytes, long* messages);        /* halo traffic sent so far */
/* host wall time this rank has spent in its block half-sweeps / in (and waiting for) the junction updates, over `steps`
 * mitdvp_shard_step calls: the load balance of the site-sharded sweep */
int mitdvp_shard_phase_times(mitdvp_shard* h, double* block_ms, double* junction_ms, long* steps);
/* Panel factorisation of the QR gauge move (SiteCoef.gauge_trf, _site_cls.py:138-292 = LAPACK zgeqrf + zungqr):
 * 1 (default) = CholeskyQR2 of each 32-column panel + reconstruction of LAPACK's Householder vectors / T / tau / signs
 * of diag(R) from the orthonormal panel, falling back to 0 = one Householder step per launch (unconditionally stable)
 * when a conditioning check on the device fails.  Process-wide, like mitdvp_set_gemm_mode. */
int mitdvp_set_qr_fast(int on);
int mitdvp_get_qr_fast(void);
/* device time (HIP events) of one m x n factorisation with Q and R formed, random full-rank input */
int mitdvp_bench_qr(int device, int m, int n, int reps, double* ms_out, long* launches);
/* The thin factorisation as the SWEEP issues it (csrc/qr.h qr_thin): gauge_free != 0 = the sign convention of LAPACK's
 * diag(R) is not asked for (SiteCoef.gauge_trf's Q R = psi and Q^H Q = 1 are all a gauge move needs, _site_cls.py:264-282;
 * SURVEY appendix B item 6) -> block Gram-Schmidt over Cholesky factors of 128-column blocks, R with a positive diagonal;
 * falls back to the Householder panels when a conditioning check on the device fails.  a: m x n row-major complex128 on
 * the host, or NULL = a random full-rank matrix generated on the device (timing); q_out (m x n), r_out (n x n) may be
 * NULL.  *ms_out = device time per factorisation over `reps`, *launches per factorisation, *path_out = 1 when the
 * gauge-free path delivered the result, 0 when the Householder panels did. */
int mitdvp_qr_thin(int device, const double* a, int m, int n, int gauge_free, double* q_out, double* r_out, int reps,
                   double* ms_out, long* launches, int* path_out);
/* warm-up memory of the local solves at a site (_Debug.niter_krylov[isite], _integrator.py:178-186) */
int mitdvp_get_krylov_memory(mitdvp_engine* h, int isite, int* k);
int mitdvp_set_krylov_memory(mitdvp_engine* h, int isite, int k);
/* the one-launch small-bond kernels need all their workgroups resident: switch them off (0) for engines that share
 * their GPU with other processes; default on (or MITDVP_SMALL_KERNELS) */
int mitdvp_set_small_kernels(mitdvp_engine* h, int on);

/* -- observables -------------------------------------------------------- */
int mitdvp_expect(mitdvp_engine* h, int op_id, double out[2]);       /* _mps_cls.py:540-612 */
int mitdvp_autocorr(mitdvp_engine* h, double out[2]);                /* wavefunction.py:226-257, conj=False */
/* Autocorrelation without the t/2 trick (Simulator(t2_trick=False): Properties._get_autocorr,
 * properties.py:222-232, wf_zero._ints_wf_ovlp_mpo): keep a device copy of the current state, later
 * return <copy|current state> (bra conjugated). */
int mitdvp_save_reference(mitdvp_engine* h);
int mitdvp_overlap_reference(mitdvp_engine* h, double out[2]);
int mitdvp_norm(mitdvp_engine* h, double* out);                      /* _mps_cls.py:706-716 */
int mitdvp_site_rdm(mitdvp_engine* h, int isite, double* reim_out);  /* _mps_cls.py:1208-1436, key (isite,isite) */
/* General reduced density, MPSCoef.get_reduced_densities / _get_pure_reduced_density
 * (_mps_cls.py:1208-1283, :1628-1678): remain_nleg[isite] in {0,1,2} legs kept per site
 * (2 = ket and bra, 1 = diagonal).  Two calls: with out == NULL the number of complex
 * elements is returned in *n_out; then with a buffer of that size.  Axes: kept sites
 * ascending, (ket, bra) per 2-leg site. */
int mitdvp_reduced_density(mitdvp_engine* h, const int* remain_nleg, int nlen, double* reim_out, size_t* n_out);
/* SVD truncation of the bond right of the centre ("Psi") site, truncate_sigvec
 * (_site_cls.py:586-690): keep the leading singular values whose cumulative weight
 * sum s_k / sum s reaches 1 - p (at most max_dim if max_dim > 0); the kept values,
 * normalised to unit 2-norm, are written to svals_out (may be NULL) and their number
 * to *new_dim.  The engine's one-sided Jacobi SVD kernel does the decomposition. */
int mitdvp_truncate_bond(mitdvp_engine* h, double p, int max_dim, int* new_dim, double* svals_out);
/* One-site gates, Model(one_gate_to_apply=TensorHamiltonian of single-site operators):
 *   mitdvp_set_gate    : register U[d_out][d_in] (d x d, row-major, interleaved re/im) for a
 *                        site; NULL removes it.  While gates are registered mitdvp_step applies
 *                        them between its two half-sweeps with the last site as
 *                        re-orthogonalisation centre (MPSCoef.propagate, _mps_cls.py:489-490).
 *   mitdvp_apply_gates : MPSCoef.apply_one_gate (_mps_cls.py:2314-2373, :2420-2451) now, with
 *                        the current centre site as reorth_center (WFunc.apply_one_gate,
 *                        wavefunction.py:588-598, uses the resting centre 0): U on the physical
 *                        leg, canonicalizeB / canonicalizeA over the touched span (:3539-3598),
 *                        environment blocks that saw a touched site dropped (op_sys_sites = None). */
int mitdvp_set_gate(mitdvp_engine* h, int isite, const double* U_reim, int d);
int mitdvp_apply_gates(mitdvp_engine* h);
/* Kraus maps on purified states, Model(kraus_op={(site,): B} or {(site, site+1): B}) with
 * B (k, d, d) = the Kraus operators B_q[x][d'] (kraus.py:17-123):
 *   mitdvp_set_kraus   : register B for `isite` (NULL removes it).  two_site = 0: the site's
 *                        physical index is (system d, ancilla K), K = dim / d
 *                        (_kraus_contract_single_site_np, kraus.py:146-228); two_site = 1: system
 *                        site `isite` (dimension d) and ancilla site `isite + 1`
 *                        (_kraus_contract_two_site_np, :281-358; the bond between them is
 *                        re-split by the engine's Jacobi SVD).  mitdvp_step applies the maps after
 *                        the gates, between its half-sweeps (MPSCoef.propagate, _mps_cls.py:491-492).
 *   mitdvp_apply_kraus : MPSCoef.apply_kraus (_mps_cls.py:2375-2418) now, re-orthogonalising
 *                        towards the current centre site. */
int mitdvp_set_kraus(mitdvp_engine* h, int isite, int two_site, const double* B_reim, int k, int d);
int mitdvp_apply_kraus(mitdvp_engine* h);
/* Simulator.operate (simulator_cls.py:286-331, :356-360): variational application of operator
 * op_id to the state, WFunc.apply_dipole (wavefunction.py:303-351) -> MPSCoef.apply_dipole
 * (_mps_cls.py:421-450, :718-796, :2733-2778): at most maxstep double sweeps in which every site
 * tensor is replaced by the mixed-environment apply (bra = new state, ket = initial state),
 * stopped when |1 - |<phi_i|phi_{i-1}>|| < conv_tol (1e-8 in the reference).  On return the
 * engine holds phi ~ O|psi> / ||O|psi>|| (site-0 centred); *norm_out is the norm of the last
 * apply (the value Simulator.operate returns), *iters_out the number of double sweeps done. */
int mitdvp_operate(mitdvp_engine* h, int op_id, int maxstep, double conv_tol, double* norm_out, int* iters_out);
/* Adaptive bond dimension (a1TDVP), Simulator.propagate(adaptive=True, adaptive_Dmax,
 * adaptive_dD, adaptive_p_proj) -> const.adaptive / Dmax / dD / p_proj
 * (_const_cls.py:120-124, :212-216).  While enabled, every half-sweep widens the
 * neighbour tensors by up to dD orthogonal-complement vectors (get_superblock_full,
 * _mps_cls.py:3699-3755), picks each bond's new rank from the projection-error
 * functional (get_rank_and_projection_error, :1985-2105; stop when the relative
 * increment falls below p_proj, never above Dmax) and propagates the zero-padded
 * centre tensor (propagate_along_sweep, :863-987).  Real-time propagation only. */
int mitdvp_set_adaptive(mitdvp_engine* h, int enable, int dmax, int dd, double p_proj);
/* SiteCoef.thin_to_full (_site_cls.py:294-405) on caller data: gauge 0 = "A": site
 * (l, c, r) isometric over (l c) x r -> out (l, c, r + extra); gauge 1 = "B": isometric
 * over l x (c r) -> out (l + extra, c, r).  The added vectors are the leading ones of
 * the orthogonal complement in LAPACK's full-mode QR of the isometry. */
int mitdvp_thin_to_full(int device, int gauge, const double* site, int l, int c, int r, int extra, double* out);
/* kernel-level hook: A (r x c) = U diag(S) Vh with the engine's Jacobi SVD */
int mitdvp_svd(int device, const double* A, int r, int c, double* U, double* S, double* Vh, int* sweeps);

/* Liouville space (Model(space="liouville")): the MPS is a vectorised density matrix with
 * site dimension n*n, physical index = row*n + col (_mps_mpo.py:135-194).
 *   mitdvp_set_trace_op_core : core O[a][out][in][f] (M_l, n, n, M_r) of a full-chain
 *                              observable acting on the n-dimensional Hilbert-space legs
 *   mitdvp_expect_trace      : Tr(O rho), _exp_liouville (_mps_cls.py:3769-3838)
 *   mitdvp_partial_trace     : get_partial_trace (_mps_cls.py:1438-1510); size query with
 *                              out == NULL like mitdvp_reduced_density; the right-most kept
 *                              site always keeps both legs */
int mitdvp_set_trace_op_core(mitdvp_engine* h, int op_id, int isite, const double* reim, int ml, int n, int mr);
int mitdvp_expect_trace(mitdvp_engine* h, int op_id, double out[2]);
int mitdvp_partial_trace(mitdvp_engine* h, const int* remain_nleg, int nlen, double* reim_out, size_t* n_out);
/* Subspace projection of a Liouville-space site (Model(subspace_inds={site: P_inds}), model_cls.py:110-118,
 * MPSCoefMPO.project_subspace / define_reshape_mat, _mps_mpo.py:135-220): the site's physical leg holds the entries
 * inds[0..ninds) of the n*n vectorised density matrix.  The caller hands over MPO cores and site tensors with the
 * shorter leg (TensorHamiltonian.project_subspace, hamiltonian_cls.py:852-880); this call tells the trace
 * observables how to embed the leg back into n x n.  Call it BEFORE mitdvp_set_trace_op_core for that site
 * (those cores are gathered on arrival); ninds = 0 removes the projection. */
int mitdvp_set_subspace(mitdvp_engine* h, int isite, int n, const int* inds, int ninds);
/* MPSCoef.hermitise (_mps_cls.py:2289-2312, svd_conj_mpdo :2516-2562): rho <- (rho + rho^dagger) / 2 of a
 * Liouville-space chain as a direct sum, re-truncated bond by bond to the old bond dimensions by two-site SVDs,
 * left in the site-0-centred canonical form with its normalisation untouched. */
int mitdvp_hermitise(mitdvp_engine* h);
int mitdvp_krylov_stats(mitdvp_engine* h, int* per_site);            /* _Debug.niter_krylov, _helper.py:29 */

/* -- counters: _ElpTime / _NFlops equivalents (_helper.py:33-101) ------- */
typedef struct {
  double heff_flops;     /* algorithmic flops in H_eff applies (SURVEY 8d F_H)  */
  double heff_ms;        /* HIP-event time of those applies (profiling on)      */
  double env_flops, env_ms;
  double keff_flops, keff_ms;
  double qr_flops, qr_ms;
  double krylov_vec_ms;  /* Krylov vector kernels                               */
  long long n_heff, n_keff, n_env, n_qr;
  long long n_exp_site, n_exp_bond;
  long long n_launch;    /* kernel launches issued                              */
  double heff_stage_ms[3]; /* L.psi, W., .R stages of the H_eff applies (profiling on) */
  double n_collectives;    /* collectives issued (bond-sharded mode)                   */
  double collective_bytes; /* sum of the collective buffer sizes                       */
  double heff_flops_skipped; /* part of heff_flops NOT executed: zero blocks of W skipped by the block-sparse W stage */
  double n_host_waits;     /* device->host records the host waited for inside local exponentials (multi-launch regime) */
  double n_heff_edge;      /* H_eff applies that took the two-product "edge" form (the others ran the three-stage chain or the one-launch kernel) */
  double heff_stage_flops[3]; /* 8-flop-per-complex-MAC count of what the three stages EXECUTE (trimmed identity blocks,
                                 skipped zero blocks and tile padding of the W stage accounted; the 3M product's 6 / 8 is
                                 applied by the reader): with heff_stage_ms the roofline of each stage's kernel */
} mitdvp_counters;
int mitdvp_counters_get(mitdvp_engine* h, mitdvp_counters* out);
int mitdvp_counters_reset(mitdvp_engine* h);
int mitdvp_set_profiling(mitdvp_engine* h, int on); /* HIP-event timing per phase */

/* -- fine seam for unit-level parity tests (SURVEY 8b "internal seam 1"):
 *    one H_eff / K_eff apply and one environment update on caller data.
 *    contract_with_site_mpo (_contraction.py:148-397),
 *    multiplyH_MPS_direct_MPO._op_lcr_dot (:1038-1173),
 *    multiplyK_MPS_direct_MPO._op_lr_dot (:1297-1352). */
int mitdvp_heff_apply(int device, const double* L, const double* W, const double* R, const double* psi,
                      int dl, int d, int dr, int ml, int mr, double* sigma_out, int reps, double* ms_out);
int mitdvp_keff_apply(int device, const double* L, const double* R, const double* sval,
                      int dl, int dr, int m, double* out);
int mitdvp_env_update(int device, int left, const double* env, const double* site, const double* W,
                      int dl, int d, int dr, int ml, int mr, double* out);
/* SiteCoef.gauge_trf (_site_cls.py:138-292): key 0 "Psi2Asigma", 1 "Psi2sigmaB". */
int mitdvp_gauge_trf(int device, int key, const double* psi, int dl, int d, int dr,
                     double* site_out, double* sigma_out);
/* short_iterative_lanczos / _arnoldi on a dense operator (_integrator.py:287-655). */
int mitdvp_expm_dense(int device, int integrator, int conserve_norm, int lanczos_variant,
                      const double* mat, int n, const double* x, double scale_re, double scale_im,
                      double thresh, int k_prev, double* y_out, int* k_out);

/* -- kernel-level test / bench hooks (no reference counterpart) --------- */
/* C = alpha*op(A)*op(B) + beta*C on the MFMA zgemm kernel, row-major. */
int mitdvp_zgemm(int device, int transA, int conjA, int transB, int conjB, int m, int n, int k,
                 const double* A, const double* B, double* C, const double alpha[2], const double beta[2],
                 int tile_cfg, int reps, double* ms_out);
/* Complex-product form of the MFMA zgemm kernel for everything that follows:
 * 0 = "4M" (textbook, 4 real MFMA products), 1 = "3M" (Karatsuba, 3 products,
 * normwise stable); the library default is 1 unless MITDVP_ZGEMM=4m is set. */
int mitdvp_set_gemm_mode(int mode3m);
int mitdvp_get_gemm_mode(void);
/* Device-resident timing of the three-stage H_eff apply at a given shape
 * with random operands (bench roofline leg).  Returns avg ms per apply. */
int mitdvp_bench_heff(int device, int dl, int d, int dr, int ml, int mr, int reps, int warmup, double* ms_out);
/* Size-independent self checks of the H_eff apply on device-resident random
 * operands at a given (possibly full BASELINE) shape, without host copies:
 *   out[0] = ||H_3m(x) - H_4m(x)|| / ||H_4m(x)||        (two complex-product forms agree)
 *   out[1] = ||H(x + 2i y) - H(x) - 2i H(y)|| / ||H(x + 2i y)||   (linearity)
 *   out[2] = ||H(x)||, out[3] = ms of the last apply */
int mitdvp_heff_selfcheck(int device, int dl, int d, int dr, int ml, int mr, double out[4]);
/* raw v_mfma_f64_16x16x4_f64 issue-rate probe: returns TFLOP/s */
int mitdvp_mfma_peak_probe(int device, double* tflops_out);
/* dumps the C/D lane map of v_mfma_f64_16x16x4_f64: out[64*4*2] = (row, col) */
/* shader-clock probe: out[0] = shader cycles, out[1] = 100 MHz reference ticks spent by a
 * single-wavefront dependent-FMA loop of `iters` iterations (clock in MHz = 100 * out[0] / out[1]). */
int mitdvp_clock_probe(int device, long iters, double* cycles_ticks_out);
/* Where the workgroups of a launch on a stream created with a CU mask run (hipExtStreamCreateWithCUMask; mask NULL: an
 * ordinary stream): out[b] = XCC id (bits 3:0) | CU / shader-array / shader-engine id (bits 15:8) of workgroup b.  Every
 * workgroup holds lds_bytes of LDS for spin_us microseconds.  Used to find the mask bits of one XCD (ensemble mode). */
int mitdvp_cu_mask_probe(int device, const unsigned* mask, int nwords, int nblocks, size_t lds_bytes, int spin_us, int* out);
int mitdvp_mfma_layout_probe(int device, int* out);

#ifdef __cplusplus
}
#endif
#endif /* MITDVP_H */
