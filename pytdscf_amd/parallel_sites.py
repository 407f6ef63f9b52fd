"""Site-range sharding of one TDVP sweep over several GPUs (BASELINE configs 4 and 5).

The reference's multi-process path (``/root/reference/pytdscf/_mps_parallel.py``): contiguous site ranges
per rank (``parallel_split_indices``, ``_const_cls.py:236-250``), every rank sweeps its block concurrently
(even ranks ->, odd ranks <-; ``MPSCoefParallel.propagate`` :106-268), neighbours exchange environment
blocks, boundary tensors and the joint bond matrix (``send_op_block`` / ``recv_op_block`` :1610-1635,
``send_Psi_to_left`` :698-707, ``send_B_to_right`` :728-740, ``send_joint_sigvec_to_right`` :541-597), and
the two sites facing each other across a rank boundary are updated together through the pseudo-inverse
of the joint bond matrix (``propagate_joint_two_sites`` :270-470; ``multiply_sigvec_pinv``,
``_site_cls.py:709-754``, RCOND = 1e-13).  The state is Psi = Phi_0 X_0^+ Phi_1 X_1^+ ... (Secular et al.,
PRB 101, 235123); the scheme is an approximation to the serial sweep whose error vanishes as dt^2.

Here: one process per GPU and ONE library call per time step (``mitdvp_shard_step``, ``csrc/shard.hip``): the
block's half-sweeps, the joint update on the left rank of every junction (a persistent two-site engine) and the
neighbour messages around it -- per junction and half step one centre tensor (D d D), one environment block
(D M D), one B tensor, one bond matrix (D D) and one environment block back -- as grouped ``ncclSend`` / ``ncclRecv``
of device buffers issued by the library itself on the engine's stream (RCCL over xGMI; no collective on the data
path, no Python and no torch between two time steps).  ``torch.distributed`` is the CONTROL plane only: rendezvous,
the 128-byte ncclUniqueId, barriers and scalar reductions; when ranks share a GPU (tests; RCCL refuses that) it also
carries the messages, through the library's host-callback transport.

Set-up replicates a full-chain engine on every rank (same seed / same input tensors) and derives the
block from it on the device; it is not part of the timed region.  ``regularize`` / ``p_svd`` switch on the
reference's lifting of small singular values (SQRT_EPSRHO = 1e-4, ``_site_cls.py:22, :207-246, :657-664``) and the
cumulative-weight truncation of the joint matrix (``truncate_sigvec``, :586-690); with both on (the reference's
setting) the path reproduces ``MPSCoefParallel`` to 1e-8 (``tests/test_gpu_site_sharding.py`` against
``tests/golden/parallel_*.npz``); both are off by default.  Hilbert space, one electronic state.

``adaptive={"Dmax", "dD", "p_proj"}`` (round 5): adaptive bond dimensions in the blocks AND across the junctions
(``const.adaptive`` in ``propagate_along_sweep``, _mps_cls.py:863-987, and in ``propagate_joint_two_sites``,
_mps_parallel.py:319-345, :371-374): the left rank of a junction widens the right site's tensor, chooses the junction's
new rank and hands B, X' and the boundary block back at that rank; the shapes travel ahead of the tensors.  In the pair
mode both ranks of the junction run that update on identical copies and take the same decisions.
"""

from __future__ import annotations

import os

import ctypes as C

import numpy as np

from . import _lib
from .engine import TDVPEngine, svd as device_svd, zgemm as device_zgemm

RCOND = 1e-13  # _site_cls.py:24


def split_sites(nsite: int, nrank: int) -> list[tuple[int, int]]:
    """Contiguous, near-equal site ranges [lo, hi) per rank (parallel_split_indices, _const_cls.py:236-250)."""
    base, rem = divmod(nsite, nrank)
    out, lo = [], 0
    for r in range(nrank):
        n = base + (1 if r < rem else 0)
        out.append((lo, lo + n))
        lo += n
    return out


def pinv_device(x: np.ndarray, device: int = 0) -> np.ndarray:
    """Moore-Penrose inverse (np.linalg.pinv(x, rcond=RCOND)): SVD and the product V diag(1/s) U^H on the device."""
    U, s, Vh, _ = device_svd(x, device=device)
    keep = s > RCOND * s.max()
    inv = np.where(keep, 1.0 / np.where(keep, s, 1.0), 0.0)
    return device_zgemm(Vh, U * inv[None, :], transA=True, conjA=True, transB=True, conjB=True, device=device)


class _Link:
    """Neighbour send / recv of complex128 HOST arrays over torch.distributed: the control plane's messages (set-up,
    observables folded rank by rank, gathers in tests).  The halo of a time step does not go through here."""

    def __init__(self, comm):
        self.comm = comm
        self.dist = comm.dist
        self.bytes = 0
        self.messages = 0

    def send(self, arr, dst: int):
        import torch

        a = np.ascontiguousarray(arr, dtype=np.complex128)
        t = torch.from_numpy(a.view(np.float64).reshape(-1))
        if self.comm.backend == "nccl":
            t = t.to(self.comm.device)
        self.dist.send(t, dst)
        self.bytes += a.nbytes
        self.messages += 1

    def recv(self, shape, src: int):
        import torch

        n = int(np.prod(shape))
        nccl = self.comm.backend == "nccl"
        t = torch.empty(2 * n, dtype=torch.float64, device=self.comm.device if nccl else "cpu")
        self.dist.recv(t, src)
        return t.cpu().numpy().view(np.complex128).reshape(shape).copy()

    def raw_callback(self):
        """mitdvp_p2p_fn over torch.distributed (ranks sharing a GPU, or RCCL unavailable to the library)."""
        import ctypes as C

        import torch

        from . import _lib

        def cb(user, op, peer, buf, nbytes):
            try:
                t = torch.frombuffer((C.c_char * nbytes).from_address(buf), dtype=torch.uint8)
                if self.comm.backend == "nccl":
                    if op == 0:
                        self.dist.send(t.to(self.comm.device), peer)
                    else:
                        d = torch.empty(nbytes, dtype=torch.uint8, device=self.comm.device)
                        self.dist.recv(d, peer)
                        t.copy_(d.cpu())
                elif op == 0:
                    self.dist.send(t, peer)
                else:
                    self.dist.recv(t, peer)
                return 0
            except Exception:  # noqa: BLE001 -- reported through the library's error code
                import traceback

                traceback.print_exc()
                return 1

        return _lib.P2P_FN(cb)


class SiteShardedTDVP:
    """One rank of the site-sharded sweep.  All ranks construct it with the same arguments.

    ``split``: explicit ranges [(first, last), ...] as the reference's ``parallel_split_indices``; default: contiguous
    near-equal ranges.  ``regularize`` / ``p_svd``: the reference's junction regularisation (module docstring).
    ``transport``: "rccl" (library-native, default with one rank per GPU), "callback" (torch.distributed carries the
    messages; default when ranks share a GPU).  ``junction``: "pair" (default) = both ranks of a junction run its
    update, bond-sharded over the pair, so that the partner works instead of waiting; "single" = the left rank alone,
    like the reference (``MITDVP_JUNCTION`` overrides); same results to rounding."""

    def __init__(self, comm, mpo, *, cores=None, dims=None, bond_dim=None, seed=1, integrator="lanczos", thresh=1e-9,
                 conserve_norm=True, device=None, split=None, regularize=False, p_svd=None, transport=None, junction=None,
                 adaptive=None):
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.device = comm.gpu if device is None else device
        if self.device is None:
            raise RuntimeError("SiteShardedTDVP needs a GPU: the MI355X engine has no CPU fallback")
        self.mpo = [np.ascontiguousarray(w, dtype=np.complex128) for w in mpo]
        self.nsite = len(self.mpo)
        if split is not None:
            self.ranges = [(int(a), int(b) + 1) for a, b in split]
            if len(self.ranges) != self.world or self.ranges[0][0] != 0 or self.ranges[-1][1] != self.nsite or any(
                self.ranges[k][1] != self.ranges[k + 1][0] for k in range(self.world - 1)
            ):
                raise ValueError("split must be contiguous [(first, last), ...] ranges, one per rank, covering the chain")
        else:
            self.ranges = split_sites(self.nsite, self.world)
        if self.world > 1 and any(hi - lo < 2 for lo, hi in self.ranges):
            raise ValueError("site sharding needs at least two sites per rank")
        self.lo, self.hi = self.ranges[self.rank]
        self.n = self.hi - self.lo
        self.kw = dict(integrator=integrator, thresh=thresh, conserve_norm=conserve_norm)
        self.regularize, self.p_svd = bool(regularize), p_svd
        self.link = _Link(comm) if self.world > 1 else None
        shared = bool(getattr(comm, "shared_gpu", False))
        self.transport = transport or os.environ.get("MITDVP_HALO_TRANSPORT") or ("callback" if shared else "rccl")
        if self.transport not in ("rccl", "callback"):
            raise ValueError("transport must be 'rccl' or 'callback'")
        self._junction_explicit = bool(junction or os.environ.get("MITDVP_JUNCTION"))
        self.junction = junction or os.environ.get("MITDVP_JUNCTION") or "pair"
        if self.junction not in ("pair", "single"):
            raise ValueError("junction must be 'pair' or 'single'")
        self.adaptive = None
        if adaptive:
            self.adaptive = dict(Dmax=int(adaptive["Dmax"]), dD=int(adaptive["dD"]), p_proj=float(adaptive["p_proj"]))
        self._h = None
        self._cb = None
        import time as _time

        t0 = _time.perf_counter()
        self._setup(cores, dims, bond_dim, seed, shared)
        self.setup_s = _time.perf_counter() - t0

    # ------------------------------------------------------------------ set-up (not timed)
    def _ck(self, rc):
        _lib.check(rc, self._h, shard=True)

    def _replicated_state(self, cores, dims, bond_dim, seed, shared):
        L, r, N = self.nsite, self.rank, self.world
        g = TDVPEngine(L, device=self.device, **self.kw)  # replicated full chain: B world, then A world
        if shared:
            g.set_small_kernels(False)  # persistent kernels need the GPU to themselves
        g.set_mpo(self.mpo)
        if cores is not None:
            g.set_mps(cores)
        else:
            g.init_random(list(dims), bond_dim, seed=seed)
        self.shapes = [g.get_site_shape(p)[:3] for p in range(L)]
        even = r % 2 == 0
        lo, hi, n = self.lo, self.hi, self.n
        g.build_envs(1)
        right_b = g.get_env(1, hi)  # through the B world right of the block
        bcores = [g.get_site(p) for p in range(lo, hi)] if even else None
        # A world, incrementally: X at the block's junctions, the block's A tensors (odd ranks), left boundary block
        acores, X_left, X_right = [], None, None
        left_b = g.get_env(0, 0) if lo == 0 else None
        for p in range(0, hi if hi < L else L - 1):
            g.split_center(True)
            if lo <= p < hi and not even:
                acores.append(g.get_site(p))
            if p + 1 == lo:
                X_left = g.get_bond()
                left_b = g.get_env(0, lo)
            if p + 1 == hi:
                X_right = g.get_bond()
            g.absorb_bond(True)
        if not even and hi == L:
            acores.append(g.get_site(L - 1))  # the A world's centre
        g.close()
        return bcores, acores, left_b, right_b, X_left, X_right

    def _setup_pipeline(self, b, cores, dims, bond_dim, seed, shared):
        """The block engine ``b`` (MPO set) filled rank by rank; returns the junction matrix to the right (or None).

        B world, from the last rank down: a rank takes the weight matrix sigma and the right boundary block from its right
        neighbour, brings its own sites into gauge B (C2sigmaB site by site, _mps_cls.py:2684-2693; given ``cores`` are in
        that gauge already) and passes sigma of its first site and the block through its sites on.  A world, from rank 0
        up: a rank takes the junction matrix X and the left boundary block, absorbs X into its first site and walks the
        centre through its block (Psi -> A sigma, sigma B -> Psi); the matrix left after its last site is the junction
        matrix to the right.  Even ranks keep the state before that walk (it runs on a scratch copy of the block), odd
        ranks the state at its end."""
        L, r, N, link = self.nsite, self.rank, self.world, self.link
        lo, hi, n = self.lo, self.hi, self.n
        even = r % 2 == 0
        one = np.ones((1, 1, 1), dtype=np.complex128)
        Dl, Dr = self.shapes[lo][0], self.shapes[hi - 1][2]
        ml = 1 if lo == 0 else self.mpo[lo].shape[0]
        mr = 1 if hi == L else self.mpo[hi].shape[0]
        # ---- B world ----------------------------------------------------------------------------------------
        if r > 0:  # a segment wants both boundary blocks in place; the left one arrives with the A world below
            b.set_boundary_env(0, np.zeros((Dl, ml, Dl), dtype=np.complex128))
        if cores is not None:
            for i in range(n):
                b.set_site(i, cores[lo + i], "Psi" if lo + i == 0 else "B")
            right_b = one if r == N - 1 else link.recv((Dr, mr, Dr), r + 1)
            b.set_boundary_env(1, right_b)
            if r > 0:
                link.send(b.fold_block(right_b, op_id=0, conj=True, from_left=False, out_shape=(Dl, ml, Dl)), r - 1)
        else:
            b.init_random_block(list(dims), lo, bond_dim, seed=seed)
            if r == N - 1:
                right_b = one
                b.set_boundary_env(1, right_b)
                b.set_site(n - 1, b.get_site(n - 1), "Psi")  # (d, d, 1) at most: marks the chain's last site as the centre
            else:
                sig = link.recv((Dr, Dr), r + 1)
                right_b = link.recv((Dr, mr, Dr), r + 1)
                b.set_boundary_env(1, right_b)
                b.set_bond(n, sig)
                b.absorb_bond(False)
            for _ in range(n - 1):
                b.split_center(False)
                b.absorb_bond(False)
            if r > 0:
                b.split_center(False)
                sig = b.get_bond()
                link.send(sig / np.linalg.norm(sig), r - 1)
                link.send(b.get_env(1, 0), r - 1)
            else:  # the chain's first site carries the norm (alloc_superblock_random, _mps_cls.py:2695-2699)
                x = b.get_site(0)
                b.set_site(0, x / np.linalg.norm(x), "Psi")
        # ---- A world ----------------------------------------------------------------------------------------
        if r == 0:
            left_b = one
            b.set_boundary_env(0, left_b)
        else:
            X_left = link.recv((Dl, Dl), r - 1)
            left_b = link.recv((Dl, ml, Dl), r - 1)
            b.set_boundary_env(0, left_b)
            b.set_bond(0, X_left)
            b.absorb_bond(True)
        X_right = None
        if even:
            b.build_envs(1)
            if r < N - 1:  # the walk to the right junction, on a scratch copy: this rank keeps the B world
                g = TDVPEngine(n, device=self.device, **self.kw)
                if shared:
                    g.set_small_kernels(False)
                g.set_mpo(self.mpo[lo:hi])
                for i in range(n):
                    g.set_site(i, b.get_site(i), "Psi" if i == 0 else "B")
                g.set_boundary_env(0, left_b)
                g.set_boundary_env(1, right_b)
                for i in range(n):
                    g.split_center(True)
                    if i < n - 1:
                        g.absorb_bond(True)
                X_right = g.get_bond()
                link.send(X_right, r + 1)
                link.send(g.get_env(0, n), r + 1)
                g.close()
        else:
            for _ in range(n - 1):
                b.split_center(True)
                b.absorb_bond(True)
            if hi < L:
                b.split_center(True)
                X_right = b.get_bond()
                link.send(X_right, r + 1)
                link.send(b.get_env(0, n), r + 1)
                b.absorb_bond(False)  # back into the last site: this rank's centre
            b.build_envs(0)
        return X_right

    def _setup(self, cores, dims, bond_dim, seed, shared):
        """Every rank ends with its block in the mixed gauge of the parallel scheme (even ranks: B world, centre on the
        first site; odd ranks: A world, centre on the last), the two boundary blocks and the junction matrices.
        ``MITDVP_SHARD_SETUP=pipeline`` (default): the ranks canonicalise their own blocks one after the other and hand
        the weight matrix and the boundary block on (_setup_pipeline) -- a block per rank in memory; ``replicated``: every
        rank walks a copy of the whole chain (rounds 2-4; kept for comparison)."""
        L, r, N = self.nsite, self.rank, self.world
        mode = os.environ.get("MITDVP_SHARD_SETUP", "pipeline")
        if mode not in ("pipeline", "replicated"):
            raise ValueError("MITDVP_SHARD_SETUP must be 'pipeline' or 'replicated'")
        self.setup_mode = mode
        even = r % 2 == 0
        lo, hi, n = self.lo, self.hi, self.n
        if mode == "replicated":
            bcores, acores, left_b, right_b, X_left, X_right = self._replicated_state(cores, dims, bond_dim, seed, shared)
        elif cores is not None:
            self.shapes = [tuple(int(x) for x in np.shape(c)) for c in cores]
        else:
            from .mps import bond_dims

            self.shapes = [(bl, int(dd), br) for (bl, br), dd in zip(bond_dims(list(dims), bond_dim), dims)]
        # the native shard: block engine + two-site junction engine + transport
        lib = _lib.load()
        cfg = _lib.Config()
        cfg.nsite = n
        cfg.device = self.device
        cfg.integrator = {"lanczos": _lib.LANCZOS, "arnoldi": _lib.ARNOLDI}[self.kw["integrator"]]
        cfg.conserve_norm = int(bool(self.kw["conserve_norm"]))
        cfg.thresh = self.kw["thresh"]
        cfg.max_krylov = 20
        h = C.c_void_p()
        dr_next = self.shapes[hi][2] if r < N - 1 else 0
        _lib.check(lib.mitdvp_shard_create(C.byref(cfg), r, N, n, dr_next, C.byref(h)), None, shard=True)
        self._h, self._lib = h, lib
        self._ck(lib.mitdvp_shard_set_options(h, int(self.regularize), -1.0 if self.p_svd is None else float(self.p_svd)))
        if N > 1:
            self._attach_transport()
            if self.transport == "callback" and not shared and self.junction == "pair" and not self._junction_explicit:
                # host-staged messages between different GPUs: the pair mode's 6x larger halo (both ranks need both
                # halves of the junction) would cost more than the partner's idle time it saves
                self.junction = "single"
        eh = C.c_void_p()
        self._ck(lib.mitdvp_shard_engine(h, 0, C.byref(eh)))
        b = TDVPEngine.borrow(eh, n, self.device)
        self.joint = None
        if r < N - 1:
            jh = C.c_void_p()
            self._ck(lib.mitdvp_shard_engine(h, 1, C.byref(jh)))
            self.joint = TDVPEngine.borrow(jh, 2, self.device)
            self.joint.set_mpo([self.mpo[hi - 1], self.mpo[hi]])
        self.joint_left = None
        if N > 1 and self.junction == "pair":
            self._ck(lib.mitdvp_shard_enable_pair(h, self.shapes[lo - 1][0] if r > 0 else 0))
            if r > 0:
                lh = C.c_void_p()
                self._ck(lib.mitdvp_shard_engine(h, 2, C.byref(lh)))
                self.joint_left = TDVPEngine.borrow(lh, 2, self.device)
                self.joint_left.set_mpo([self.mpo[lo - 1], self.mpo[lo]])
        if shared:
            b.set_small_kernels(False)
            for j in (self.joint, self.joint_left):
                if j is not None:
                    j.set_small_kernels(False)
        b.set_mpo(self.mpo[lo:hi])
        if mode == "pipeline":
            X_right = self._setup_pipeline(b, cores, dims, bond_dim, seed, shared)
        elif even:
            for i, c in enumerate(bcores):
                b.set_site(i, c, "Psi" if (i == 0 and r == 0) else "B")
            b.set_boundary_env(0, left_b)
            b.set_boundary_env(1, right_b)
            if r > 0:  # first site takes the weight of the junction to its left
                b.set_bond(0, X_left)
                b.absorb_bond(True)
            b.build_envs(1)
        else:
            last_is_center = hi == L
            for i, c in enumerate(acores):
                b.set_site(i, c, "Psi" if (i == n - 1 and last_is_center) else "A")
            b.set_boundary_env(0, left_b)
            b.set_boundary_env(1, right_b)
            if not last_is_center:
                b.set_bond(n, X_right)
                b.absorb_bond(False)
            b.build_envs(0)
        self.block = b
        if self.adaptive:  # the junction engine takes the same settings inside the library
            b.set_adaptive(True, **self.adaptive)
        if r < N - 1:  # joint matrix of the junction to the right (held by the left rank of every junction)
            x = np.ascontiguousarray(X_right, dtype=np.complex128)
            self._ck(lib.mitdvp_shard_set_joint(h, x.ctypes.data_as(C.POINTER(C.c_double)), x.shape[0]))

    def _attach_transport(self):
        """library-native RCCL between chain neighbours, or torch.distributed through the callback transport; all
        ranks end up with the same kind (the verdict of the self-test is shared)."""
        lib, h, comm = self._lib, self._h, self.comm
        if self.transport == "rccl":
            # The communicator is created on a helper thread with a deadline: an RCCL bootstrap that cannot reach its
            # peers blocks for ever and cannot be cancelled, and a bench that hangs measures nothing.  On a time-out
            # (or any error) ALL ranks fall back to the callback transport together; the helper thread is left behind
            # (a communicator that completes later is never used: the callback takes precedence in the library).
            import threading

            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # one node: bootstrap over loopback (the host name may not resolve)
            state = {"ok": False, "err": None}
            ident = C.create_string_buffer(128)
            try:
                if comm.rank == 0:
                    _lib.check(lib.mitdvp_rccl_unique_id(ident))
                box = [ident.raw]
                comm.dist.broadcast_object_list(box, src=0)

                def attach():
                    try:
                        self._ck(lib.mitdvp_shard_attach_rccl(h, box[0]))
                        state["ok"] = True
                    except Exception as exc:  # noqa: BLE001
                        state["err"] = exc

                th = threading.Thread(target=attach, daemon=True)
                th.start()
                th.join(float(os.environ.get("MITDVP_RCCL_ATTACH_TIMEOUT", "120")))
                if th.is_alive():
                    state["err"] = TimeoutError("ncclCommInitRank did not return in time")
                    # the helper is still inside mitdvp_shard_attach_rccl(h): the handle must outlive it, so close()
                    # leaks the shard instead of destroying it under the helper's feet (as after a wedged transfer)
                    self._wedged = True
            except Exception as exc:  # noqa: BLE001 -- all ranks fall back together below
                state["err"] = exc
            if state["err"] is not None or not state["ok"]:
                print(f"[site sharding] rank {comm.rank}: RCCL attach failed ({state['err']}); falling back to the callback transport", flush=True)
            if comm.min_over_ranks(1.0 if state["ok"] else 0.0) < 1.0:
                self.transport = "callback"
        if self.transport == "callback":
            self._cb = self.link.raw_callback()
            self._ck(lib.mitdvp_shard_set_transport(h, self._cb, None))

    @property
    def X(self):
        """joint matrix of the junction to the right (joint_sigvec_not_pinv), host copy"""
        dim = C.c_int()
        self._ck(self._lib.mitdvp_shard_get_joint(self._h, None, C.byref(dim)))
        x = np.empty((dim.value, dim.value), dtype=np.complex128)
        self._ck(self._lib.mitdvp_shard_get_joint(self._h, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(dim)))
        return x

    # ------------------------------------------------------------------ the step
    def step(self, dt):
        """One time step = two half-sweeps of every block + one joint update of every junction: one library call."""
        self._ck(self._lib.mitdvp_shard_step(self._h, float(dt)))
        if self.adaptive:
            self._refresh_shapes()

    def _refresh_shapes(self):
        """adaptive ranks: every rank learns the new site shapes of the whole chain (gathers and folds post receives
        with them).  Collective."""
        mine = [tuple(int(x) for x in self.block.get_site_shape(i)[:3]) for i in range(self.n)]
        if self.world == 1:
            self.shapes = mine
            return
        box = [None] * self.world
        self.comm.dist.all_gather_object(box, mine)
        self.shapes = [s for part in box for s in part]

    def bond_dims(self):
        """right bond of every site but the last, over the whole chain (properties.py:255-262)"""
        return [int(s[2]) for s in self.shapes[:-1]]

    # ------------------------------------------------------------------ the whole state (tests, observables)
    def gather(self):
        """Psi = Phi_0 X_0^+ Phi_1 ... as one list of site tensors on rank 0 (None elsewhere)."""
        cs = [self.block.get_site(i) for i in range(self.n)]
        if self.rank < self.world - 1:
            cs[-1] = np.tensordot(cs[-1], pinv_device(self.X, self.device), axes=(2, 0))
        if self.world == 1:
            return cs
        if self.rank == 0:
            out = list(cs)
            for src in range(1, self.world):
                lo, hi = self.ranges[src]
                out.extend(self.link.recv(self.shapes[p], src) for p in range(lo, hi))
            return out
        for c in cs:
            self.link.send(c, 0)
        return None

    # ------------------------------------------------------------------ observables without gathering the state
    # MPSCoefParallel.ovlp / norm / autocorr / expectation (_mps_parallel.py:855-1033, :1210-1302): transfer blocks
    # are folded from both ends of the chain towards the middle junction, rank by rank, on the devices
    # (mitdvp_fold_block); a rank's effective tensors are its block followed by X^+ of the junction to its right,
    # which enters as one more site of physical dimension 1.
    def _fold(self, block, from_left, op_id, conj, op_cores):
        eng = self.block
        parts = ["block"]
        if self.rank < self.world - 1:
            parts = parts + ["x"] if from_left else ["x"] + parts
        for kind in parts:
            if kind == "block":
                m_out = 1 if op_id < 0 else (op_cores[self.hi - 1].shape[3] if from_left else op_cores[self.lo].shape[0])
                shp = self.shapes[self.hi - 1][2] if from_left else self.shapes[self.lo][0]
                block = eng.fold_block(block, op_id=op_id, conj=conj, from_left=from_left, out_shape=(shp, m_out, shp))
            else:
                block = self._fold_x(block, from_left, op_id, conj)
        return block

    def _fold_x(self, block, from_left, op_id, conj):
        """the block through X^+ of the junction to the right: one more site of physical dimension 1"""
        D = self.X.shape[0]
        x = TDVPEngine(1, device=self.device, **self.kw)
        try:
            x.set_site(0, pinv_device(self.X, self.device).reshape(D, 1, D), "C")
            if op_id >= 0:
                M = block.shape[1]
                x.set_mpo([np.eye(M, dtype=np.complex128).reshape(M, 1, 1, M)], op_id=op_id)
            return x.fold_block(block, op_id=op_id, conj=conj, from_left=from_left, out_shape=block.shape)
        finally:
            x.close()

    def _fold_chain(self, op_id, conj, op_cores):
        """every rank returns the scalar (it is formed at the middle junction and shared)."""
        r, N = self.rank, self.world
        one = np.ones((1, 1, 1), dtype=np.complex128)
        if N == 1:
            return complex(self._fold(one, True, op_id, conj, op_cores)[0, 0, 0])
        mid = N // 2  # ranks < mid fold from the left, the others from the right (_mps_parallel.py:858-861)
        val = 0.0 + 0.0j
        if r < mid:
            D = self.shapes[self.lo][0]
            M = 1 if op_id < 0 else op_cores[self.lo].shape[0]
            blk = one if r == 0 else self.link.recv((D, M, D), r - 1)
            blk = self._fold(blk, True, op_id, conj, op_cores)
            if r < mid - 1:
                self.link.send(blk, r + 1)
            else:
                other = self.link.recv(blk.shape, mid)
                val = complex(np.sum(blk * other))  # "ab,ab->" over bra and ket (and the MPO bond)
        else:
            D = self.shapes[self.hi - 1][2]
            M = 1 if op_id < 0 else op_cores[self.hi - 1].shape[3]
            blk = one if r == N - 1 else self.link.recv((D, M, D), r + 1)
            blk = self._fold(blk, False, op_id, conj, op_cores)
            self.link.send(blk, r - 1)
        re = self.comm.sum_over_ranks(val.real)
        im = self.comm.sum_over_ranks(val.imag)
        return complex(re, im)

    def overlap(self, conj=True):
        """<Psi|Psi> (conj) or <Psi*|Psi> of the sharded state; collective over all ranks."""
        return self._fold_chain(-1, conj, None)

    def norm(self):
        return float(np.sqrt(max(self.overlap(True).real, 0.0)))

    def autocorr(self):
        return self.overlap(False)

    def expectation(self, op_cores=None):
        """<Psi|O|Psi> for an operator given as MPO cores over the WHOLE chain (default: the Hamiltonian)."""
        if op_cores is None:
            return self._fold_chain(0, True, self.mpo)
        cores = [np.ascontiguousarray(w, dtype=np.complex128) for w in op_cores]
        if len(cores) != self.nsite:
            raise ValueError("operator needs one MPO core per site of the chain")
        self.block.set_mpo(cores[self.lo : self.hi], op_id=1)
        return self._fold_chain(1, True, cores)

    def site_rdm(self, site: int):
        """Reduced density rho[j][j'] of one site of the sharded state (MPSCoefParallel.get_reduced_densities,
        _mps_parallel.py:1035-1208, one-site keys): transfer blocks are folded towards the owner of the site from both
        ends of the chain, rank by rank; every rank returns the (d, d) matrix.  Collective."""
        r, N = self.rank, self.world
        if not 0 <= site < self.nsite:
            raise ValueError("site index out of range")
        owner = next(k for k, (lo, hi) in enumerate(self.ranges) if lo <= site < hi)
        one = np.ones((1, 1, 1), dtype=np.complex128)
        d = self.shapes[site][1]
        rho = np.zeros((d, d), dtype=np.complex128)
        if r < owner:  # pass the left part on
            D = self.shapes[self.lo][0]
            blk = one if r == 0 else self.link.recv((D, 1, D), r - 1)
            self.link.send(self._fold(blk, True, -1, True, None), r + 1)
        elif r > owner:
            D = self.shapes[self.hi - 1][2]
            blk = one if r == N - 1 else self.link.recv((D, 1, D), r + 1)
            self.link.send(self._fold(blk, False, -1, True, None), r - 1)
        else:
            Dl, Dr = self.shapes[self.lo][0], self.shapes[self.hi - 1][2]
            tl = one if r == 0 else self.link.recv((Dl, 1, Dl), r - 1)
            tr = one if r == N - 1 else self.link.recv((Dr, 1, Dr), r + 1)
            p = site - self.lo
            tl = self.block.fold_block(tl, op_id=-1, conj=True, from_left=True, first=0, count=p,
                                       out_shape=(self.shapes[site][0], 1, self.shapes[site][0]))
            if r < N - 1:  # X^+ of the junction to the right sits between this block and the next
                tr = self._fold_x(tr, False, -1, True)
            tr = self.block.fold_block(tr, op_id=-1, conj=True, from_left=False, first=p + 1, count=self.n - p - 1,
                                       out_shape=(self.shapes[site][2], 1, self.shapes[site][2]))
            rho = self.block.site_rdm_blocks(p, tl[:, 0, :], tr[:, 0, :])
        if N > 1:  # share: only the owner holds non-zero entries
            import torch

            t = torch.from_numpy(np.ascontiguousarray(rho).view(np.float64).reshape(-1).copy())
            if self.comm.backend == "nccl":
                t = t.to(self.comm.device)
            self.comm.dist.all_reduce(t)
            rho = t.cpu().numpy().view(np.complex128).reshape(d, d)
        return rho

    def reduced_density(self, key):
        """General reduced density of the sharded state for a key like the reference's ``reduced_density=([key, ...], n)``
        -- ascending site indices, an index given twice keeps ket and bra of that site, once its diagonal
        (``MPSCoefParallel.get_reduced_densities``, _mps_parallel.py:1035-1208; ``properties.py:69-82``).  One-site keys
        (i, i) are folded rank by rank without moving tensors (``site_rdm``); any other key gathers the chain on rank 0
        (Psi = Phi_0 X_0^+ Phi_1 ...), re-canonicalises it there keeping its norm and uses the single-GPU contraction --
        an analysis call for states that fit one GPU, not part of a time step.  Every rank returns the array.  Collective."""
        key = tuple(int(k) for k in key)
        if key != tuple(sorted(key)) or any(not 0 <= k < self.nsite for k in key):
            raise ValueError(f"Reduced density key {key} must be ascending site indices")
        if len(key) == 2 and key[0] == key[1]:
            return self.site_rdm(key[0])
        legs = [0] * self.nsite
        for k in key:
            legs[k] += 1
        if max(legs) > 2:
            raise ValueError("a site index may appear at most twice in a reduced-density key")
        cores = self.gather()
        rho = None
        if self.rank == 0:
            g = TDVPEngine(self.nsite, device=self.device, **self.kw)
            try:
                if getattr(self.comm, "shared_gpu", False):
                    g.set_small_kernels(False)
                g.set_mps(cores, canonicalize=True, scale=None)
                rho = g.reduced_density(legs)
            finally:
                g.close()
        if self.world > 1:
            box = [rho]
            self.comm.dist.broadcast_object_list(box, src=0)
            rho = box[0]
        return rho

    def site_rdms(self, sites=None):
        """One-site reduced densities of many sites in ONE two-way pass over the ranks (2 (N - 1) messages whatever the
        number of sites): the left transfer blocks travel rank 0 -> N - 1, the right ones back, every rank then walks
        its own sites with both.  Returns {site: rho (d, d)} on every rank.  Collective."""
        r, N = self.rank, self.world
        sites = list(range(self.nsite)) if sites is None else sorted(set(int(p) for p in sites))
        if any(not 0 <= p < self.nsite for p in sites):
            raise ValueError("site index out of range")
        one = np.ones((1, 1, 1), dtype=np.complex128)
        Dl, Dr = self.shapes[self.lo][0], self.shapes[self.hi - 1][2]
        # left block at this rank's first site; pass the block at the next rank's first site on
        tl_in = one if r == 0 else self.link.recv((Dl, 1, Dl), r - 1)
        if r < N - 1:
            self.link.send(self._fold(tl_in, True, -1, True, None), r + 1)
        tr_in = one if r == N - 1 else self.link.recv((Dr, 1, Dr), r + 1)
        if r > 0:
            self.link.send(self._fold(tr_in, False, -1, True, None), r - 1)
        mine = [p for p in sites if self.lo <= p < self.hi]
        out = {}
        if mine:
            b, n = self.block, self.n
            tr = self._fold_x(tr_in, False, -1, True) if r < N - 1 else tr_in
            rights = {n - 1: tr}  # block right of local site q, for q down to the first wanted site
            for q in range(n - 2, mine[0] - self.lo - 1, -1):
                D = self.shapes[self.lo + q][2]
                rights[q] = b.fold_block(rights[q + 1], op_id=-1, conj=True, from_left=False, first=q + 1, count=1, out_shape=(D, 1, D))
            tl, at = tl_in, 0
            for p in mine:
                q = p - self.lo
                if q > at:
                    D = self.shapes[p][0]
                    tl = b.fold_block(tl, op_id=-1, conj=True, from_left=True, first=at, count=q - at, out_shape=(D, 1, D))
                    at = q
                out[p] = b.site_rdm_blocks(q, tl[:, 0, :], rights[q][:, 0, :])
        if N == 1:
            return out
        # share: every rank contributes the entries of its own sites
        import torch

        ds = [self.shapes[p][1] for p in sites]
        flat = np.zeros(sum(d * d for d in ds), dtype=np.complex128)
        off = 0
        for p, d in zip(sites, ds):
            if p in out:
                flat[off : off + d * d] = out[p].reshape(-1)
            off += d * d
        t = torch.from_numpy(flat.view(np.float64).copy())
        if self.comm.backend == "nccl":
            t = t.to(self.comm.device)
        self.comm.dist.all_reduce(t)
        flat = t.cpu().numpy().view(np.complex128)
        res, off = {}, 0
        for p, d in zip(sites, ds):
            res[p] = flat[off : off + d * d].reshape(d, d).copy()
            off += d * d
        return res

    def selftest(self) -> bool:
        """Neighbour ping through the step's own transport (every junction, both directions) before the sweep relies
        on it; the verdict is shared, so all ranks agree.  Collective."""
        if self.world == 1:
            return True
        bad = C.c_int(-1)

        def ping():
            try:
                self._ck(self._lib.mitdvp_shard_selftest(self._h, C.byref(bad)))
            except Exception as exc:  # noqa: BLE001 -- shared below
                print(f"[site sharding] rank {self.rank}: transport self-test raised {exc}", flush=True)

        # with a deadline: a point-to-point transfer whose peer never arrives blocks for ever in the stream, and a run that
        # hangs measures nothing.  A rank that times out leaves the handle behind (its stream is stuck: close() would block)
        import threading

        th = threading.Thread(target=ping, daemon=True)
        th.start()
        th.join(float(os.environ.get("MITDVP_SELFTEST_TIMEOUT", "180")))
        if th.is_alive():
            print(f"[site sharding] rank {self.rank}: transport self-test did not return in time", flush=True)
            self._wedged = True
            return self.comm.min_over_ranks(0.0) >= 1.0
        return self.comm.min_over_ranks(1.0 if bad.value == 0 else 0.0) >= 1.0

    def traffic(self):
        """(bytes, messages) of halo traffic this rank has SENT in its time steps"""
        if self._h is None or self.world == 1:
            return (0, 0)
        b, m = C.c_double(), C.c_long()
        self._ck(self._lib.mitdvp_shard_traffic(self._h, C.byref(b), C.byref(m)))
        return (int(b.value), int(m.value))

    def phase_times(self):
        """(block_ms, junction_ms, steps): host wall time of this rank's block half-sweeps / junction updates so far"""
        if self._h is None:
            return (0.0, 0.0, 0)
        b, j, n = C.c_double(), C.c_double(), C.c_long()
        self._ck(self._lib.mitdvp_shard_phase_times(self._h, C.byref(b), C.byref(j), C.byref(n)))
        return (b.value, j.value, int(n.value))

    def close(self):
        if getattr(self, "_wedged", False):  # a transfer still blocks the shard's stream: leak it rather than hang
            self._h = None
            return
        if self._h is not None:
            self.block.close()
            for j in (self.joint, getattr(self, "joint_left", None)):
                if j is not None:
                    j.close()
            self._lib.mitdvp_shard_destroy(self._h)
            self._h = None
