"""CPU oracle of the SITE-RANGE SHARDED (real-space parallel) one-site TDVP -- test infrastructure.

PARITY: PINNED to the reference (round 3).  The reference's implementation of this scheme
(``/root/reference/pytdscf/_mps_parallel.py``: ``MPSCoefParallel.propagate`` :106-268,
``propagate_joint_two_sites`` :270-470, split rule ``_const_cls.py:236-250``) was run in the development container
on 2 and 3 ranks -- forked processes under a process-backed stand-in for the third-party ``mpi4py``
(``tests/golden/make_golden_parallel.py``) -- and its per-step states, norms and <Psi*|Psi> are committed as
``tests/golden/parallel_chain_r2.npz`` / ``parallel_chain_r3.npz`` (well-conditioned entangled start, where the
regularisations below are inactive: this file equals them to 1e-14, ``tests/test_oracle_parallel.py``) and
``parallel_exciton.npz`` (the model of the reference's own ``tests/test_mpi_exiciton_propagate.py`` from its
zero-padded product start, where they are what shapes the result; reference-held pin :220).

What the fixtures taught about the reference (all reproduced or stated here):
  * ``distribute_superblock_states`` (:1520-1607) is consistent for PRODUCT starts only: its A world goes through
    ``CC2ALambdaB`` (``_mps_cls.py:3601-3630``), whose right factor is the row space of the two-site SVD, i.e. rotated
    against the B world the even ranks keep, and the even ranks start with ``joint_sigvec_not_pinv = pinv(Lambda)``.
    For Lambda = (1, 0, ..) both are harmless.  ``ParallelOracle`` distributes any canonical chain consistently (QR
    bond matrices); the chain fixtures were produced by filling the reference's state object with the equally
    consistent Gamma-Lambda form of the same chain.
  * the four local solves of a junction update share ONE warm-up memory, that of the last site the left rank's sweep
    propagated (``_Debug.site_now`` is set by the sweep only, ``_mps_cls.py:880``; the junction update never sets it).
  * with ``regularize=True`` the left junction site is rebuilt from an SVD of its (D_l D_r x d) unfolding with small
    singular values lifted (``SiteCoef.gauge_trf``, ``_site_cls.py:207-246``), and the new joint matrix is
    ``truncate_sigvec(p=const.p_svd, regularize=True, keepdim=True)`` (``_site_cls.py:586-690``): SVD, cumulative-weight
    cut, kept values below SQRT_EPSRHO = 1e-4 lifted to s + eps exp(-s/eps), zeros kept on the cut ones, A <- A U,
    B <- Vh B, boundary blocks rebuilt.  Both are options here (``regularize``, ``p_svd``; the reference runs with both
    on, ``_mps_parallel.py:369, :437-444``); off, the scheme is Secular et al., PRB 101, 235123 as it stands.
  * ``to_MPSCoefMPO`` (the checkpoint) concatenates per-rank B worlds saved at different half steps; the fixtures
    therefore hold the state as ``MPSCoefParallel.ovlp`` reads it (:872-897), not that chain.

Adaptive bond dimensions (``adaptive={"Dmax", "dD", "p_proj"}``, round 5): the block half-sweeps run the serial
adaptive step with the frozen boundary blocks (``propagate_along_sweep`` with ``const.adaptive``, _mps_cls.py:863-987;
the end site of a block is an orthonormal tensor at the start of the sweep and is widened like any other, the bond
beyond it is not touched), and the junction update widens the right site's B tensor and chooses the junction's new
rank with the same functional (``get_superblock_full`` / ``get_adaptive_rank_and_block`` on the two-site superblock,
_mps_parallel.py:319-345, :371-374); the joint matrix, both facing tensors and both boundary blocks come out at the
new rank.  Pinned against ``parallel_adaptive_r2.npz`` at the looseness the reference itself accepts (from a product
start the lifted null directions are arbitrary; from a full-rank start the reference's own junction update fails, see
``make_golden_parallel.py``) and, where nothing can grow, exactly against the non-adaptive scheme.

All ranks are simulated in one process, in the order the real ranks would act; what one rank reads
from another is exactly what ``pytdscf_amd/parallel_sites.py`` sends over RCCL / gloo.

Per time step (reference diagram, ``_mps_parallel.py:115-122``):
  S0   even ranks [psi B .. B], odd ranks [A .. A psi]; junctions hold X_j
  (a)  every rank sweeps its block (even ->, odd <-) with FROZEN boundary environments and leaves its
       end site unpropagated (``skip_end_site``), except at the ends of the chain
  (b)  joint update of every even|odd junction: psi_L X^+ psi_R -> A X' B (site L +dt/2, bond -dt/2,
       site R +dt/2, bond -dt/2), new boundary environments for both neighbours
  (c)  psi_L = A X', psi_R = X' B; sweeps in the opposite directions; joint update of the odd|even
       junctions; absorb again.  Every site has had two half steps, every bond two backward half steps.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.linalg

from . import tdvp_oracle as orc

RCOND = 1e-13  # _site_cls.py:24
SQRT_EPSRHO = 1e-4  # _site_cls.py:22


def _lift(sig: np.ndarray) -> np.ndarray:
    """Small singular values lifted: s -> s + eps exp(-s / eps) below eps (_site_cls.py:232-236, :657-664)."""
    return np.where(sig > SQRT_EPSRHO, sig, sig + SQRT_EPSRHO * np.exp(-sig / SQRT_EPSRHO))


def regularize_site(psi: np.ndarray) -> np.ndarray:
    """SiteCoef.gauge_trf(regularize=True), NumPy branch (_site_cls.py:207-246): SVD of the (D_l D_r x d)
    unfolding, singular values lifted, tensor rebuilt."""
    ldim, ndim, rdim = psi.shape
    U, sig, Vh = scipy.linalg.svd(np.ascontiguousarray(psi.transpose(0, 2, 1).reshape(-1, ndim)), full_matrices=False)
    return np.dot(U, np.dot(np.diag(_lift(sig)), Vh)).reshape(ldim, rdim, ndim).transpose(0, 2, 1)


def truncate_joint(A: np.ndarray, sigma: np.ndarray, B: np.ndarray, p: float, regularize: bool):
    """truncate_sigvec(Asite, sigvec, Bsite, p, regularize, keepdim=True) (_site_cls.py:586-690): returns
    (A U, diag(s' / |s'|) padded with zeros to the old size, Vh B)."""
    U, s, Vh = scipy.linalg.svd(sigma, full_matrices=False)
    cumsum = np.cumsum(s.real)
    idx = int(np.argmax(cumsum / cumsum[-1] >= (1 - p)) + 1)
    thin = s[:idx]
    A2 = np.tensordot(A, U, axes=(2, 0))
    B2 = np.tensordot(Vh, B, axes=(1, 0))
    if regularize and sigma.shape != (1, 1):
        thin = _lift(thin)
    full = np.zeros_like(s)
    full[:idx] = thin
    return A2, np.diag(full / np.linalg.norm(thin)).astype(np.complex128), B2


def split_sites(nsite: int, nrank: int) -> list[tuple[int, int]]:
    """Contiguous, near-equal site ranges [lo, hi) per rank (parallel_split_indices, _const_cls.py:236-250)."""
    base, rem = divmod(nsite, nrank)
    out, lo = [], 0
    for r in range(nrank):
        n = base + (1 if r < rem else 0)
        out.append((lo, lo + n))
        lo += n
    return out


@dataclass
class Block:
    """One rank's block Phi_r: its site tensors, MPO cores, environment cache and boundary blocks."""

    cores: list
    mpo: list
    lo: int  # global index of the first site
    left_b: np.ndarray  # environment left of the block (through the orthonormal blocks of the ranks to the left)
    right_b: np.ndarray
    integrator: str = "lanczos"
    thresh: float = 1e-9
    conserve_norm: bool = True
    adaptive: dict | None = None  # {"Dmax", "dD", "p_proj"}: const.adaptive (_const_cls.py:120-124, :212-216)
    kprev: dict = field(default_factory=dict)
    left: dict = field(default_factory=dict)  # left[i]: environment left of local site i
    right: dict = field(default_factory=dict)  # right[i]: environment right of local site i

    def __post_init__(self):
        self.n = len(self.cores)
        self.left = {0: self.left_b}
        self.right = {self.n - 1: self.right_b}

    def _exp(self, scale, mv, x, site, size=None):
        fn = orc.sil_lanczos if self.integrator == "lanczos" else orc.sil_arnoldi
        out, k = fn(scale, mv, x, self.thresh, self.kprev.get(site, 0), self.conserve_norm, size)
        self.kprev[site] = k
        return out

    def build_right(self):
        for p in range(self.n - 1, 0, -1):
            self.right[p - 1] = orc.env_update_right(self.right[p], self.cores[p], self.mpo[p])

    def build_left(self):
        for p in range(0, self.n - 1):
            self.left[p + 1] = orc.env_update_left(self.left[p], self.cores[p], self.mpo[p])

    def sweep(self, dt, forward, skip_end):
        """propagate_along_sweep (_mps_cls.py:798-1014) over the block; with skip_end the end site keeps
        the centre but is not propagated (:876-877)."""
        n = self.n
        sites = range(0, n) if forward else range(n - 1, -1, -1)
        end = n - 1 if forward else 0
        view = full = None
        if self.adaptive:  # get_superblock_full over the block (_mps_cls.py:863-874): every site but the centre widened
            full = orc.superblock_full(self.cores, 0 if forward else n - 1, self.adaptive["dD"])
            view = _AdaptiveView(self, self.cores, self.mpo, self.left, self.right, lambda q: self.lo + q)
        for p in sites:
            g = self.lo + p
            if skip_end and p == end:
                return
            if view is not None and p != end and orc.OracleMPS._adaptive_site(view, p, dt, forward, full):
                continue
            L, W, R = self.left[p], self.mpo[p], self.right[p]
            self.cores[p] = self._exp(-0.5j * dt, lambda x: orc.heff_apply(L, W, R, x), self.cores[p], g)
            if p == end:
                return
            if forward:
                A, s = orc.qr_psi2Asigma(self.cores[p])
                self.cores[p] = A
                self.left[p + 1] = orc.env_update_left(self.left[p], A, W)
                Ln, Rn = self.left[p + 1], self.right[p]
                s = self._exp(+0.5j * dt, lambda x: orc.keff_apply(Ln, Rn, x), s, g)
                self.cores[p + 1] = np.tensordot(s, self.cores[p + 1], axes=(1, 0))
            else:
                s, B = orc.qr_psi2sigmaB(self.cores[p])
                self.cores[p] = np.ascontiguousarray(B)
                self.right[p - 1] = orc.env_update_right(self.right[p], self.cores[p], W)
                Ln, Rn = self.left[p], self.right[p - 1]
                s = self._exp(+0.5j * dt, lambda x: orc.keff_apply(Ln, Rn, x), s, g)
                self.cores[p - 1] = np.tensordot(self.cores[p - 1], s, axes=(2, 0))


class _AdaptiveView:
    """What ``OracleMPS._adaptive_site`` (the serial adaptive step, pinned by the a1TDVP fixtures) reads and writes of
    its object, laid over a block's -- or a junction's two -- tensors, blocks and warm-up memories."""

    relax = False
    shift = 0.0

    def __init__(self, blk: Block, cores, mpo, left, right, key, site_hook=None):
        self.blk, self.cores, self.mpo, self.left, self.right, self.key = blk, cores, mpo, left, right, key
        self.integrator = blk.integrator
        self.Dmax, self.p_proj = int(blk.adaptive["Dmax"]), float(blk.adaptive["p_proj"])
        self.site_hook = site_hook

    def _exp(self, scale, mv, x, site, size=None):
        return self.blk._exp(scale, mv, x, self.key(site), size)

    @staticmethod
    def _keff(L, R):
        return lambda x: orc.keff_apply(L, R, x)


def joint_update(bl: Block, br: Block, X: np.ndarray, dt: float, regularize: bool = False, p_svd: float | None = None):
    """propagate_joint_two_sites (_mps_parallel.py:270-470) for the junction between ``bl`` (centre on
    its last site) and ``br`` (centre on its first site).  Returns the new X; both blocks end with
    orthonormal junction sites (A | B) and refreshed boundary environments.  All four solves use the warm-up
    memory of the last site the left block's sweep propagated (see the module header)."""
    pl, pr = bl.n - 1, 0
    Wl, Wr = bl.mpo[pl], br.mpo[pr]
    mem = bl.lo + bl.n - 2
    Lenv = bl.left[pl]  # through the A sites of the left block
    Renv = br.right[pr]  # through the B sites of the right block
    # psi_L X^+ (multiply_sigvec_pinv, _site_cls.py:709-754), then centre on the left site
    theta = np.tensordot(bl.cores[pl], np.linalg.pinv(X, rcond=RCOND), axes=(2, 0))
    s, B = orc.qr_psi2sigmaB(br.cores[pr])
    B = np.ascontiguousarray(B)
    theta = np.tensordot(theta, s, axes=(2, 0))
    R1 = orc.env_update_right(Renv, B, Wr)
    grown = False
    if bl.adaptive:
        # const.adaptive, :319-345, :371-374: the two-site superblock [Psi, B] widened (get_superblock_full), the
        # junction's rank chosen by get_adaptive_rank_and_block, the left site propagated into the widened bond with the
        # truncated applies, regularised, split, the bond matrix propagated in the blocks at the new rank -- the serial
        # adaptive step on two sites whose outer blocks are the two ranks' environments
        cores2, left2, right2 = [theta, B], {0: Lenv}, {0: R1, 1: Renv}
        view = _AdaptiveView(bl, cores2, [Wl, Wr], left2, right2, lambda q: mem, regularize_site if regularize else None)
        full = orc.superblock_full(cores2, 0, bl.adaptive["dD"])
        grown = orc.OracleMPS._adaptive_site(view, 0, dt, True, full)
        if grown:
            A, L1, psi_r = cores2[0], left2[1], cores2[1]
    if not grown:
        theta = bl._exp(-0.5j * dt, lambda x: orc.heff_apply(Lenv, Wl, R1, x), theta, mem)
        if regularize:  # trans_next_psite_AsigmaB(..., regularize=True), :362-370
            theta = regularize_site(theta)
        A, s = orc.qr_psi2Asigma(theta)
        L1 = orc.env_update_left(Lenv, A, Wl)
        s = bl._exp(+0.5j * dt, lambda x: orc.keff_apply(L1, R1, x), s, mem)
        psi_r = np.tensordot(s, B, axes=(1, 0))
    psi_r = bl._exp(-0.5j * dt, lambda x: orc.heff_apply(L1, Wr, Renv, x), psi_r, mem)
    s, B = orc.qr_psi2sigmaB(psi_r)
    B = np.ascontiguousarray(B)
    R2 = orc.env_update_right(Renv, B, Wr)
    s = bl._exp(+0.5j * dt, lambda x: orc.keff_apply(L1, R2, x), s, mem)
    if p_svd is not None:  # truncate=True branch, :437-466
        A, s, B = truncate_joint(A, s, B, p_svd, regularize)
        B = np.ascontiguousarray(B)
        L1 = orc.env_update_left(Lenv, A, Wl)
        R2 = orc.env_update_right(Renv, B, Wr)
    bl.cores[pl], br.cores[pr] = A, B
    # what each side needs next: its own block's environment through the junction site, and the
    # neighbour's as its new boundary block
    bl.left[pl + 1] = L1  # (kept for completeness: left block's full left environment)
    bl.right_b = R2
    bl.right = {pl: R2}
    br.left_b = L1
    br.left = {0: L1}
    br.right_after_first = R2
    return s


class ParallelOracle:
    """N blocks in one process.  ``cores``: site-0-centred canonical MPS (Psi, B, ..., B)."""

    def __init__(self, cores, mpo, nrank, integrator="lanczos", thresh=1e-9, conserve_norm=True, ranges=None,
                 regularize=False, p_svd=None, adaptive=None):
        """``regularize`` / ``p_svd``: the reference's lifting of small singular values and the cumulative-weight
        truncation of the joint matrix (it runs with regularize=True, p_svd=const.p_svd, default 1e-7)."""
        self.regularize = regularize
        self.p_svd = p_svd
        self.adaptive = dict(adaptive) if adaptive else None  # {"Dmax", "dD", "p_proj"}
        self.nsite = len(cores)
        self.nrank = nrank
        # explicit [lo, hi) ranges = the reference's parallel_split_indices [(first, last), ...] (_const_cls.py:236-250)
        self.ranges = [tuple(r) for r in ranges] if ranges is not None else split_sites(self.nsite, nrank)
        assert len(self.ranges) == nrank and self.ranges[0][0] == 0 and self.ranges[-1][1] == self.nsite
        if any(hi - lo < 2 for lo, hi in self.ranges) and nrank > 1:
            raise ValueError("every rank needs at least two sites")
        cores = [np.array(c, dtype=np.complex128) for c in cores]
        mpo = [np.array(w, dtype=np.complex128) for w in mpo]
        one = np.ones((1, 1, 1), dtype=np.complex128)
        # B world: right environments through every site
        Rb = {self.nsite - 1: one}
        for p in range(self.nsite - 1, 0, -1):
            Rb[p - 1] = orc.env_update_right(Rb[p], cores[p], mpo[p])
        # A world built incrementally; X_j = centre matrix at junction j (distribute_superblock_states, :1520-1607)
        A = [c.copy() for c in cores]
        La = {0: one}
        self.X = []
        cuts = [hi for _, hi in self.ranges[:-1]]
        xs = {}
        for p in range(self.nsite - 1):
            Q, s = orc.qr_psi2Asigma(A[p])
            A[p] = Q
            La[p + 1] = orc.env_update_left(La[p], Q, mpo[p])
            if p + 1 in cuts:
                xs[p + 1] = s
            A[p + 1] = np.tensordot(s, A[p + 1], axes=(1, 0))
        self.blocks = []
        for r, (lo, hi) in enumerate(self.ranges):
            if r % 2 == 0:  # [psi B .. B]: B-world slice, first site carries X of the junction to its left
                cs = [cores[p].copy() for p in range(lo, hi)]
                if r > 0:
                    cs[0] = np.tensordot(xs[lo], cores[lo], axes=(1, 0))
            else:  # [A .. A psi]: A-world slice, last site carries X of the junction to its right
                cs = [A[p].copy() for p in range(lo, hi)]
                if r < nrank - 1:
                    cs[-1] = np.tensordot(A[hi - 1], xs[hi], axes=(2, 0))
                # (the last rank's last site already is the A-world centre)
            blk = Block(cs, mpo[lo:hi], lo, La[lo], Rb[hi - 1], integrator, thresh, conserve_norm, self.adaptive)
            self.blocks.append(blk)
        self.X = [xs[c] for c in cuts]
        for r, blk in enumerate(self.blocks):
            if r % 2 == 0:
                blk.build_right()
            else:
                blk.build_left()

    def step(self, dt):
        nr = self.nrank
        if nr == 1:
            b = self.blocks[0]
            b.sweep(dt, True, False)
            b.sweep(dt, False, False)
            return
        # (a) even ->, odd <-
        for r, b in enumerate(self.blocks):
            fwd = r % 2 == 0
            at_chain_end = (fwd and r == nr - 1) or (not fwd and r == 0)
            b.sweep(dt, fwd, skip_end=not at_chain_end)
        # (b) even|odd junctions
        self._junctions(dt, 0)
        # (c) even <-, odd ->
        for r, b in enumerate(self.blocks):
            fwd = r % 2 == 1
            at_chain_end = (fwd and r == nr - 1) or (not fwd and r == 0)
            b.sweep(dt, fwd, skip_end=not at_chain_end)
        self._junctions(dt, 1)

    def _junctions(self, dt, parity):
        for j in range(parity, self.nrank - 1, 2):
            bl, br = self.blocks[j], self.blocks[j + 1]
            Xn = joint_update(bl, br, self.X[j], dt, self.regularize, self.p_svd)
            self.X[j] = Xn
            # A X' B -> psi x^+ psi: both neighbours take the weight (send_joint_sigvec_to_right, :541-597)
            R2 = br.right_after_first
            bl.cores[-1] = np.tensordot(bl.cores[-1], Xn, axes=(2, 0))
            br.cores[0] = np.tensordot(Xn, br.cores[0], axes=(1, 0))
            # the blocks' own caches: left block sweeps <- next (needs its left environments, which its
            # -> sweep left behind), right block sweeps -> next
            bl.right = {bl.n - 1: bl.right_b}
            br.left = {0: br.left_b}
            # the right block's remaining right environments are still valid (its <- sweep built them)
            _ = R2

    # ---- the whole state as one MPS (tests, observables) ---------------------------------------
    def gather(self):
        """Site tensors of Psi = Phi_0 X_0^+ Phi_1 ... as one chain (not canonical)."""
        out = []
        for r, b in enumerate(self.blocks):
            cs = [c.copy() for c in b.cores]
            if r < self.nrank - 1:
                cs[-1] = np.tensordot(cs[-1], np.linalg.pinv(self.X[r], rcond=RCOND), axes=(2, 0))
            out.extend(cs)
        return out

    def norm(self):
        g = self.gather()
        return float(np.sqrt(abs(orc.overlap(g, g))))

    def bond_dims(self):
        """right bond of every site but the last, over the whole chain"""
        return [c.shape[2] for b in self.blocks for c in b.cores][:-1]
