"""Thin Python handle over the C ABI (``include/mitdvp.h``).

Mirrors what ``WFunc``/``MPSCoefMPO`` expose for the hot path in the reference:
``propagate`` (= ``MPSCoef.propagate``, _mps_cls.py:452-503), ``expectation``,
``autocorr``, ``norm`` (_mps_cls.py:540-716).  Arrays are complex128, C order.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _c128(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.complex128))


def _is_dev(x) -> bool:
    """a torch tensor living on a GPU (complex128): handed to the library as a device pointer"""
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


class _Operand:
    """What a setter passes down: a host array (NumPy) or a device tensor (torch, complex128, contiguous).
    ``mode`` is the pointer mode the call needs (include/mitdvp.h, mitdvp_set_pointer_mode)."""

    def __init__(self, x):
        if _is_dev(x):
            import torch

            if x.dtype != torch.complex128:
                raise TypeError("device operands must be complex128")
            self.keep = x.contiguous()
            self.shape = tuple(self.keep.shape)
            self.ptr = C.cast(C.c_void_p(self.keep.data_ptr()), C.POINTER(C.c_double))
            self.mode = 1
        else:
            self.keep = _c128(x)
            self.shape = self.keep.shape
            self.ptr = _dp(self.keep)
            self.mode = 0
        self.ndim = len(self.shape)


class TDVPEngine:
    def __init__(
        self,
        nsite: int,
        *,
        device: int = 0,
        integrator: str = "lanczos",
        conserve_norm: bool = True,
        relax: bool | str = False,
        thresh: float = 1e-9,
        max_krylov: int = 20,
        lanczos_variant: str = "reference",
        cu_range: tuple[int, int] | None = None,
    ):
        """``cu_range = (first, count)``: confine the engine to ``count`` compute units starting at ``first`` (multiples of 8:
        8 k units are k CUs on every XCD); engines with disjoint ranges never compete for a compute unit (TDVPEnsemble)."""
        lib = _lib.load()
        cfg = _lib.Config()
        cfg.nsite = nsite
        cfg.device = device
        cfg.integrator = {"lanczos": _lib.LANCZOS, "arnoldi": _lib.ARNOLDI}[integrator]
        cfg.conserve_norm = int(bool(conserve_norm))
        cfg.relax = 2 if relax == "improved" else int(bool(relax))
        cfg.thresh = thresh
        cfg.max_krylov = max_krylov
        cfg.lanczos_variant = {"reference": 0, "orthodox": 1}[lanczos_variant]
        if cu_range is not None:
            cfg.cu_first, cfg.cu_count = int(cu_range[0]), int(cu_range[1])
        self._lib = lib
        self._h = C.c_void_p()
        self.nsite = nsite
        self.device = device
        _lib.check(lib.mitdvp_create(C.byref(cfg), C.byref(self._h)))

    @classmethod
    def borrow(cls, handle, nsite: int, device: int):
        """Wrap an engine handle that something else owns (``mitdvp_shard_engine``): ``close`` only forgets it."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = handle
        self._borrowed = True
        self.nsite = nsite
        self.device = device
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if not getattr(self, "_borrowed", False):
                self._lib.mitdvp_destroy(self._h)
            self._h = None

    def heff_apply_center(self, x=None):
        """sigma = H_eff x at the centre site, through the very kernels a local exponential uses; returns
        (sigma, flags): bit 0 / 1 identity block of the left / right environment short-circuited, bit 2 block-sparse
        W stage, bit 3 one-launch small-bond kernel."""
        c = next(p for p in range(self.nsite) if self.get_site_shape(p)[3] == _lib.GAUGE_PSI)
        l, n, r, _ = self.get_site_shape(c)
        out = np.empty((l, n, r), dtype=np.complex128)
        flags = C.c_int(0)
        src = None if x is None else _c128(x)
        self._ck(self._lib.mitdvp_heff_apply_center(self._h, None if src is None else _dp(src), _dp(out), C.byref(flags)))
        return out, flags.value

    def krylov_memory(self, isite: int) -> int:
        k = C.c_int()
        self._ck(self._lib.mitdvp_get_krylov_memory(self._h, isite, C.byref(k)))
        return k.value

    def set_small_kernels(self, on: bool):
        self._ck(self._lib.mitdvp_set_small_kernels(self._h, int(bool(on))))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(rc, self._h)

    # ---- host or device operands ------------------------------------------
    def _with_mode(self, mode: int, call):
        """run one library call with the pointer mode its operands need (0 host, 1 device)"""
        if mode == 0:
            return self._ck(call())
        self._ck(self._lib.mitdvp_set_pointer_mode(self._h, 1))
        try:
            return self._ck(call())
        finally:
            self._ck(self._lib.mitdvp_set_pointer_mode(self._h, 0))

    def _out(self, shape, device: bool):
        """destination of a getter: NumPy array, or a torch complex128 tensor on this engine's GPU"""
        if not device:
            a = np.empty(shape, dtype=np.complex128)
            return a, _dp(a), 0
        import torch

        t = torch.empty(tuple(shape), dtype=torch.complex128, device=torch.device("cuda", self.device))
        torch.cuda.current_stream(t.device).synchronize()  # the allocator may hand back memory with work pending
        return t, C.cast(C.c_void_p(t.data_ptr()), C.POINTER(C.c_double)), 1

    # ---- state ---------------------------------------------------------
    def set_site(self, isite: int, data, gauge: str = "C"):
        a = _Operand(data)
        if a.ndim != 3:
            raise ValueError("site tensor must be (D_l, d, D_r)")
        g = {"Psi": _lib.GAUGE_PSI, "A": _lib.GAUGE_A, "B": _lib.GAUGE_B, "C": _lib.GAUGE_C}[gauge]
        self._with_mode(a.mode, lambda: self._lib.mitdvp_set_site(self._h, isite, a.ptr, a.shape[0], a.shape[1], a.shape[2], g))

    def set_mps(self, cores, canonicalize: bool = False, scale: float = 1.0):
        """cores: site-0-centred canonical MPS (gauges Psi,B,...,B), or arbitrary
        cores with ``canonicalize=True`` (alloc_superblock_random's QR sweep)."""
        for i, c in enumerate(cores):
            self.set_site(i, c, "C" if canonicalize else ("Psi" if i == 0 else "B"))
        if canonicalize:  # scale=None: keep the state's own normalisation (Liouville space)
            self._ck(self._lib.mitdvp_canonicalize(self._h, -1.0 if scale is None else scale))

    def get_site(self, isite: int, device: bool = False):
        l, n, r, g = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self._lib.mitdvp_get_site_shape(self._h, isite, C.byref(l), C.byref(n), C.byref(r), C.byref(g)))
        out, ptr, mode = self._out((l.value, n.value, r.value), device)
        self._with_mode(mode, lambda: self._lib.mitdvp_get_site(self._h, isite, ptr))
        return out

    def get_mps(self):
        return [self.get_site(i) for i in range(self.nsite)]

    def init_random(self, dims, bond_dim: int, seed: int = 1):
        arr = (C.c_int * len(dims))(*dims)
        self._ck(self._lib.mitdvp_init_random(self._h, arr, bond_dim, seed))

    def init_random_block(self, dims, first: int, bond_dim: int, seed: int = 1, balance: bool = True):
        """the raw tensors ``init_random`` draws for sites [first, first + nsite) of the chain with physical dimensions
        ``dims`` (global shapes and seeds), not canonicalised: one block of a site-sharded state (parallel_sites.py)"""
        arr = (C.c_int * len(dims))(*dims)
        self._ck(self._lib.mitdvp_init_random_block(self._h, arr, len(dims), int(first), bond_dim, seed, int(bool(balance))))

    def set_mpo(self, cores, op_id: int = 0, shift: complex = 0.0):
        cores = list(cores)
        if cores:
            self._mpo_ends = getattr(self, "_mpo_ends", {})
            self._mpo_ends[op_id] = (np.shape(cores[0])[0], np.shape(cores[-1])[-1])
            # MPO bond left / right of every site: the block a ranged fold hands back has the bond at the END of its range
            self._mpo_bonds = getattr(self, "_mpo_bonds", {})
            self._mpo_bonds[op_id] = [(np.shape(c)[0], np.shape(c)[-1]) for c in cores]
        for i, w in enumerate(cores):
            a = _c128(w)
            if a.ndim != 4:
                raise ValueError("MPO core must be (M_l, d, d, M_r)")
            self._ck(
                self._lib.mitdvp_set_mpo_core(self._h, op_id, i, _dp(a), a.shape[0], a.shape[1], a.shape[2], a.shape[3])
            )
        self._ck(self._lib.mitdvp_set_shift(self._h, op_id, complex(shift).real, complex(shift).imag))

    # ---- one block of a site-range sharded chain (include/mitdvp.h, "one block ...") ---------
    _GAUGE = {"Psi": _lib.GAUGE_PSI, "A": _lib.GAUGE_A, "B": _lib.GAUGE_B, "C": _lib.GAUGE_C}

    def replace_site(self, isite: int, data, gauge: str):
        a = _Operand(data)
        self._with_mode(a.mode, lambda: self._lib.mitdvp_replace_site(self._h, isite, a.ptr, self._GAUGE[gauge]))

    def set_boundary_env(self, side: int, block):
        a = _Operand(block)
        if a.ndim != 3 or a.shape[0] != a.shape[2]:
            raise ValueError("boundary block must be (D, M, D)")
        self._with_mode(a.mode, lambda: self._lib.mitdvp_set_boundary_env(self._h, side, a.ptr, a.shape[0], a.shape[1]))

    def get_env(self, side: int, bond: int, device: bool = False):
        d, m = C.c_int(), C.c_int()
        self._ck(self._lib.mitdvp_get_env(self._h, side, bond, None, C.byref(d), C.byref(m)))
        out, ptr, mode = self._out((d.value, m.value, d.value), device)
        self._with_mode(mode, lambda: self._lib.mitdvp_get_env(self._h, side, bond, ptr, C.byref(d), C.byref(m)))
        return out

    def build_envs(self, side: int):
        self._ck(self._lib.mitdvp_build_envs(self._h, side))

    def site_exp(self, dt_au: float):
        self._ck(self._lib.mitdvp_site_exp(self._h, dt_au))

    def split_center(self, forward: bool):
        self._ck(self._lib.mitdvp_split_center(self._h, int(forward)))

    def bond_exp(self, dt_au: float):
        self._ck(self._lib.mitdvp_bond_exp(self._h, dt_au))

    def absorb_bond(self, forward: bool):
        self._ck(self._lib.mitdvp_absorb_bond(self._h, int(forward)))

    def get_bond(self) -> np.ndarray:
        n = C.c_int()
        self._ck(self._lib.mitdvp_get_bond(self._h, None, C.byref(n)))
        out = np.empty((n.value, n.value), dtype=np.complex128)
        self._ck(self._lib.mitdvp_get_bond(self._h, _dp(out), C.byref(n)))
        return out

    def set_bond(self, bond: int, x):
        a = _c128(x)
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("bond matrix must be square")
        self._ck(self._lib.mitdvp_set_bond(self._h, bond, _dp(a), a.shape[0]))

    def site_rdm_blocks(self, isite: int, left, right) -> np.ndarray:
        """rho[j][j'] of one site given the transfer blocks (bra, ket) of everything left / right of it."""
        a, b = _c128(left), _c128(right)
        l, n, r = self.get_site_shape(isite)[:3]
        if a.shape != (l, l) or b.shape != (r, r):
            raise ValueError("transfer blocks must be (D_l, D_l) and (D_r, D_r)")
        out = np.empty((n, n), dtype=np.complex128)
        self._ck(self._lib.mitdvp_site_rdm_blocks(self._h, isite, _dp(a), _dp(b), _dp(out)))
        return out

    def fold_block(self, block, *, op_id: int = -1, conj: bool = True, from_left: bool = True, out_shape=None, first: int = 0,
                   count: int | None = None):
        """Carry a boundary block (D, M, D) through all sites of this engine (``mitdvp_fold_block``): the piece of
        ``MPSCoefParallel.ovlp`` / ``expectation`` one rank computes.  ``op_id < 0``: plain transfer (M = 1)."""
        a = _c128(block)
        if a.ndim != 3 or a.shape[0] != a.shape[2]:
            raise ValueError("boundary block must be (D, M, D)")
        if count is None:
            count = self.nsite - first
        if count == 0:
            return a.copy()
        if out_shape is None:
            last = self.get_site_shape(first + count - 1 if from_left else first)
            dn = last[2] if from_left else last[0]
            m_out = 1
            if op_id >= 0:
                bonds = getattr(self, "_mpo_bonds", {}).get(op_id)
                if bonds is None:
                    raise ValueError("fold_block: this operator's cores were not set through set_mpo; pass out_shape")
                m_out = bonds[first + count - 1][1] if from_left else bonds[first][0]
            out_shape = (dn, m_out, dn)
        out = np.empty(out_shape, dtype=np.complex128)
        self._ck(self._lib.mitdvp_fold_block_range(self._h, op_id, int(conj), int(from_left), first, count, _dp(a), a.shape[0],
                                                   a.shape[1], _dp(out)))
        return out

    # ---- hot path ------------------------------------------------------
    def propagate(self, dt_au: float):
        self._ck(self._lib.mitdvp_step(self._h, dt_au))

    def sweep(self, dt_au: float, forward: bool):
        self._ck(self._lib.mitdvp_sweep(self._h, dt_au, int(forward)))

    def invalidate_env(self):
        self._ck(self._lib.mitdvp_invalidate_env(self._h))

    # ---- observables ---------------------------------------------------
    def expectation(self, op_id: int = 0) -> complex:
        out = np.zeros(2)
        self._ck(self._lib.mitdvp_expect(self._h, op_id, _dp(out)))
        return complex(out[0], out[1])

    def autocorr(self) -> complex:
        out = np.zeros(2)
        self._ck(self._lib.mitdvp_autocorr(self._h, _dp(out)))
        return complex(out[0], out[1])

    def save_reference(self) -> None:
        """Keep a device copy of the current state (the t = 0 state of a run without the t/2 trick)."""
        self._ck(self._lib.mitdvp_save_reference(self._h))

    def overlap_reference(self) -> complex:
        """<saved state | current state>."""
        out = np.zeros(2)
        self._ck(self._lib.mitdvp_overlap_reference(self._h, _dp(out)))
        return complex(out[0], out[1])

    def norm(self) -> float:
        out = C.c_double()
        self._ck(self._lib.mitdvp_norm(self._h, C.byref(out)))
        return out.value

    def site_rdm(self, isite: int) -> np.ndarray:
        d = self.get_site_shape(isite)[1]
        out = np.empty((d, d), dtype=np.complex128)
        self._ck(self._lib.mitdvp_site_rdm(self._h, isite, _dp(out)))
        return out

    def reduced_density(self, remain_nleg) -> np.ndarray:
        """``get_reduced_densities`` for one key: legs kept per site (0, 1 or 2)."""
        legs = [int(x) for x in remain_nleg]
        arr = (C.c_int * len(legs))(*legs)
        n = C.c_size_t(0)
        self._ck(self._lib.mitdvp_reduced_density(self._h, arr, len(legs), None, C.byref(n)))
        out = np.empty(n.value, dtype=np.complex128)
        self._ck(self._lib.mitdvp_reduced_density(self._h, arr, len(legs), _dp(out), C.byref(n)))
        shape = []
        for p, k in enumerate(legs):
            shape += [self.get_site_shape(p)[1]] * k
        return out.reshape(shape)

    def operate(self, op_id: int = 0, maxstep: int = 10, conv_tol: float = 1.0e-8):
        """``Simulator.operate``: replace the state by O|psi> / ||O|psi>|| fitted in the current
        bond dimensions; returns (norm of the last apply, double sweeps done)."""
        nrm, it = C.c_double(), C.c_int()
        self._ck(self._lib.mitdvp_operate(self._h, op_id, maxstep, conv_tol, C.byref(nrm), C.byref(it)))
        return nrm.value, it.value

    def set_gates(self, gates: dict | None) -> None:
        """Register one-site gates ``{site: U}`` (``Model(one_gate_to_apply=...)``): U is
        d x d (U[d_out, d_in]) or a length-d diagonal; ``propagate`` applies them between its
        half-sweeps.  ``None`` / ``{}`` removes all gates."""
        for i in range(self.nsite):
            self._ck(self._lib.mitdvp_set_gate(self._h, i, None, 0))
        self._step_gates = dict(gates or {})
        for site, U in (gates or {}).items():
            U = np.asarray(U, dtype=np.complex128)
            if U.ndim == 1:
                U = np.diag(U)
            if U.ndim != 2 or U.shape[0] != U.shape[1]:
                raise ValueError("a gate must be a square matrix or a diagonal")
            U = np.ascontiguousarray(U)
            self._ck(self._lib.mitdvp_set_gate(self._h, int(site), _dp(U), U.shape[0]))

    def apply_gates(self) -> None:
        """``apply_one_gate`` now, re-orthogonalising towards the current centre site."""
        self._ck(self._lib.mitdvp_apply_gates(self._h))

    def set_kraus(self, kraus: dict | None) -> None:
        """Register Kraus maps ``{(site,): B}`` / ``{(site, site + 1): B}`` with B of shape
        (k, d, d) (``Model(kraus_op=...)``); ``propagate`` applies them after the gates,
        between its half-sweeps.  ``None`` / ``{}`` removes all maps."""
        for i in range(self.nsite):
            self._ck(self._lib.mitdvp_set_kraus(self._h, i, 0, None, 0, 0))
        for sites, B in (kraus or {}).items():
            sites = tuple(int(x) for x in (sites if isinstance(sites, (tuple, list)) else (sites,)))
            B = np.ascontiguousarray(np.asarray(B, dtype=np.complex128))
            if B.ndim != 3 or B.shape[1] != B.shape[2]:
                raise ValueError("a Kraus tensor must have shape (k, d, d)")
            if len(sites) == 2 and sites[0] + 1 != sites[1]:
                raise ValueError(f"site_inds={sites} is not nearest neighbour")
            if len(sites) not in (1, 2):
                raise ValueError(f"site_inds={sites} is not yet implemented")
            self._ck(self._lib.mitdvp_set_kraus(self._h, sites[0], int(len(sites) == 2), _dp(B), B.shape[0], B.shape[1]))

    def apply_kraus(self) -> None:
        """``apply_kraus`` now, re-orthogonalising towards the current centre site."""
        self._ck(self._lib.mitdvp_apply_kraus(self._h))

    def set_adaptive(self, enable: bool = True, Dmax: int = 20, dD: int = 5, p_proj: float = 1.0e-4) -> None:
        """Adaptive bond dimension (``Simulator.propagate(adaptive=True, adaptive_Dmax=...,
        adaptive_dD=..., adaptive_p_proj=...)``): ranks grow by up to ``dD`` per half-sweep
        where the projection error asks for it, never above ``Dmax``."""
        self._ck(self._lib.mitdvp_set_adaptive(self._h, int(bool(enable)), int(Dmax), int(dD), float(p_proj)))

    def bond_dims(self) -> list[int]:
        """Current bond dimensions (``WFunc.bonddim``)."""
        return [self.get_site_shape(i)[2] for i in range(self.nsite - 1)]

    def truncate_bond(self, p: float, max_dim: int = 0):
        """SVD-truncate the bond right of the centre site (``truncate_sigvec``).
        Returns (new bond dimension, kept singular values normalised to unit norm)."""
        l, n, r, g = self.get_site_shape(self._center())
        sv = np.zeros(r)
        nd = C.c_int()
        self._ck(self._lib.mitdvp_truncate_bond(self._h, p, max_dim, C.byref(nd), _dp(sv)))
        return nd.value, sv[: nd.value].copy()

    def _center(self) -> int:
        for i in range(self.nsite):
            if self.get_site_shape(i)[3] == _lib.GAUGE_PSI:
                return i
        raise ValueError("no centre site")

    # ---- Liouville space (vectorised density matrices) -------------------
    def set_trace_op(self, cores, op_id: int):
        """Full-chain observable with n-dimensional physical legs (site dim = n*n)."""
        for i, w in enumerate(cores):
            a = _c128(w)
            if a.ndim != 4 or a.shape[1] != a.shape[2]:
                raise ValueError("trace operator core must be (M_l, n, n, M_r)")
            self._ck(self._lib.mitdvp_set_trace_op_core(self._h, op_id, i, _dp(a), a.shape[0], a.shape[1], a.shape[3]))

    def set_subspace(self, isite: int, n: int, inds):
        """Subspace projection of a Liouville-space site (``Model(subspace_inds=...)``, _mps_mpo.py:135-220): the
        physical leg of ``isite`` holds the entries ``inds`` of the n*n vectorised density matrix.  Before
        ``set_trace_op`` for that site; the sweep itself only sees the shorter leg."""
        inds = [int(x) for x in inds]
        arr = (C.c_int * max(1, len(inds)))(*inds)
        self._ck(self._lib.mitdvp_set_subspace(self._h, int(isite), int(n), arr, len(inds)))
        self._sub_n = dict(getattr(self, "_sub_n", {}))
        if inds:
            self._sub_n[int(isite)] = int(n)
        else:
            self._sub_n.pop(int(isite), None)

    def hermitise(self):
        """``MPSCoef.hermitise`` (_mps_cls.py:2289-2312): rho <- (rho + rho^dagger) / 2, bonds re-truncated to their
        old dimensions, site-0-centred canonical form."""
        self._ck(self._lib.mitdvp_hermitise(self._h))

    def expect_trace(self, op_id: int) -> complex:
        out = np.zeros(2)
        self._ck(self._lib.mitdvp_expect_trace(self._h, op_id, _dp(out)))
        return complex(out[0], out[1])

    def partial_trace(self, remain_nleg) -> np.ndarray:
        legs = [int(x) for x in remain_nleg]
        arr = (C.c_int * len(legs))(*legs)
        n = C.c_size_t(0)
        self._ck(self._lib.mitdvp_partial_trace(self._h, arr, len(legs), None, C.byref(n)))
        out = np.empty(n.value, dtype=np.complex128)
        self._ck(self._lib.mitdvp_partial_trace(self._h, arr, len(legs), _dp(out), C.byref(n)))
        center = max(i for i, k in enumerate(legs) if k)
        shape = []
        for p in range(center + 1):
            nn = getattr(self, "_sub_n", {}).get(p) or int(round(self.get_site_shape(p)[1] ** 0.5))
            shape += [nn] * (2 if p == center else legs[p])
        return out.reshape(shape)

    def get_site_shape(self, isite: int):
        l, n, r, g = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self._lib.mitdvp_get_site_shape(self._h, isite, C.byref(l), C.byref(n), C.byref(r), C.byref(g)))
        return l.value, n.value, r.value, g.value

    def krylov_stats(self):
        arr = (C.c_int * self.nsite)()
        self._ck(self._lib.mitdvp_krylov_stats(self._h, arr))
        return list(arr)

    def counters(self) -> dict:
        c = _lib.Counters()
        self._ck(self._lib.mitdvp_counters_get(self._h, C.byref(c)))
        return c.as_dict()

    def counters_reset(self):
        self._ck(self._lib.mitdvp_counters_reset(self._h))

    def set_profiling(self, on: bool):
        self._ck(self._lib.mitdvp_set_profiling(self._h, int(on)))


class MultiStateEngine(TDVPEngine):
    """Several electronic states (MPS-SM, ``nstate > 1``): one MPS per state, one MPO block and
    one scalar ``coupleJ`` per (bra, ket) state pair; the local solves act on the states' centre
    tensors stacked into one vector (reference: ``superblock_states[istate][isite]``,
    ``TensorHamiltonian.mpo[i][j]``, ``multiplyH_MPS_direct_MPO.dot``)."""

    def __init__(self, nsite: int, nstate: int, **kw):
        super().__init__(nsite, **kw)
        self.nstate = nstate
        self._ck(self._lib.mitdvp_ms_configure(self._h, nstate))

    def set_state(self, istate: int, cores, canonicalize: bool = False, scale: float = 1.0):
        """Site-0-centred cores of one state (gauges Psi,B,...,B), or arbitrary cores with
        ``canonicalize=True`` once ALL states are set (``canonicalize_states``)."""
        for p, c in enumerate(cores):
            a = _c128(c)
            if a.ndim != 3:
                raise ValueError("site tensor must be (D_l, d, D_r)")
            g = _lib.GAUGE_C if canonicalize else (_lib.GAUGE_PSI if p == 0 else _lib.GAUGE_B)
            self._ck(self._lib.mitdvp_ms_set_site(self._h, istate, p, _dp(a), a.shape[0], a.shape[1], a.shape[2], g))

    def set_states(self, states, weights=None):
        """states[istate] = arbitrary cores; QR sweep per state and site 0 scaled to
        sqrt(weight / sum(weights)) (alloc_superblock_random with init_weight_ESTATE)."""
        if len(states) != self.nstate:
            raise ValueError("one list of cores per state")
        for s, cores in enumerate(states):
            self.set_state(s, cores, canonicalize=weights is not None)
        if weights is not None:
            w = np.asarray(weights, dtype=float)
            if len(w) != self.nstate or w.min() < 0 or w.sum() <= 0:
                raise ValueError("weights must be non-negative, one per state")
            w = w / w.sum()
            for s in range(self.nstate):
                self._ck(self._lib.mitdvp_ms_canonicalize(self._h, s, float(np.sqrt(w[s]))))

    def get_state_site(self, istate: int, isite: int) -> np.ndarray:
        l, n, r, g = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self._lib.mitdvp_ms_get_site_shape(self._h, istate, isite, C.byref(l), C.byref(n), C.byref(r), C.byref(g)))
        out = np.empty((l.value, n.value, r.value), dtype=np.complex128)
        self._ck(self._lib.mitdvp_ms_get_site(self._h, istate, isite, _dp(out)))
        return out

    def get_states(self):
        return [[self.get_state_site(s, p) for p in range(self.nsite)] for s in range(self.nstate)]

    def set_block(self, ibra: int, iket: int, cores, op_id: int = 0):
        for p, w in enumerate(cores):
            a = _c128(w)
            if a.ndim != 4:
                raise ValueError("MPO core must be (M_l, d, d, M_r)")
            self._ck(self._lib.mitdvp_ms_set_mpo_core(self._h, op_id, ibra, iket, p, _dp(a), *a.shape))

    def set_hamiltonian(self, blocks, coupleJ=None, op_id: int = 0):
        """blocks[i][j] = full-chain MPO cores or None; coupleJ[i][j] scalars."""
        for i in range(self.nstate):
            for j in range(self.nstate):
                if blocks[i][j] is not None:
                    self.set_block(i, j, blocks[i][j], op_id)
                c = 0.0 if coupleJ is None else complex(coupleJ[i][j])
                self._ck(self._lib.mitdvp_ms_set_coupleJ(self._h, op_id, i, j, complex(c).real, complex(c).imag))

    def propagate(self, dt_au: float):
        self._ck(self._lib.mitdvp_ms_step(self._h, dt_au))

    def expectation(self, op_id: int = 0) -> complex:
        out = np.zeros(2)
        self._ck(self._lib.mitdvp_ms_expect(self._h, op_id, _dp(out)))
        return complex(out[0], out[1])

    def autocorr(self) -> complex:
        out = np.zeros(2)
        self._ck(self._lib.mitdvp_ms_autocorr(self._h, _dp(out)))
        return complex(out[0], out[1])

    def operate(self, op_id: int = 0, maxstep: int = 10, conv_tol: float = 1.0e-8):
        nrm, it = C.c_double(), C.c_int()
        self._ck(self._lib.mitdvp_ms_operate(self._h, op_id, maxstep, conv_tol, C.byref(nrm), C.byref(it)))
        return nrm.value, it.value

    def pop_states(self) -> list[float]:
        out = np.zeros(self.nstate)
        self._ck(self._lib.mitdvp_ms_pops(self._h, _dp(out)))
        return [float(x) for x in out]

    def norm(self) -> float:
        return float(np.sqrt(sum(self.pop_states())))


# ---- unit-level seam (SURVEY 8b "internal seam 1") ---------------------------
def heff_apply(L, W, R, psi, device=0, reps=0):
    L, W, R, psi = map(_c128, (L, W, R, psi))
    dl, d, dr = psi.shape
    out = np.empty_like(psi)
    ms = C.c_double()
    _lib.check(
        _lib.load().mitdvp_heff_apply(
            device, _dp(L), _dp(W), _dp(R), _dp(psi), dl, d, dr, W.shape[0], W.shape[3], _dp(out), reps, C.byref(ms)
        )
    )
    return (out, ms.value) if reps else out


def keff_apply(L, R, sval, device=0):
    L, R, sval = map(_c128, (L, R, sval))
    out = np.empty_like(sval)
    _lib.check(_lib.load().mitdvp_keff_apply(device, _dp(L), _dp(R), _dp(sval), sval.shape[0], sval.shape[1], L.shape[1], _dp(out)))
    return out


def env_update(env, site, W, left: bool, device=0):
    env, site, W = map(_c128, (env, site, W))
    dl, d, dr = site.shape
    ml, mr = W.shape[0], W.shape[3]
    out = np.empty((dr, mr, dr) if left else (dl, ml, dl), dtype=np.complex128)
    _lib.check(_lib.load().mitdvp_env_update(device, int(left), _dp(env), _dp(site), _dp(W), dl, d, dr, ml, mr, _dp(out)))
    return out


def gauge_trf(psi, key: str, device=0):
    """key: "Psi2Asigma" -> (A, sigma);  "Psi2sigmaB" -> (B, sigma)."""
    psi = _c128(psi)
    dl, d, dr = psi.shape
    k = {"Psi2Asigma": 0, "Psi2sigmaB": 1}[key]
    site = np.empty_like(psi)
    sig = np.empty((dr, dr) if k == 0 else (dl, dl), dtype=np.complex128)
    _lib.check(_lib.load().mitdvp_gauge_trf(device, k, _dp(psi), dl, d, dr, _dp(site), _dp(sig)))
    return site, sig


def expm_dense(mat, x, scale, integrator="lanczos", conserve_norm=True, thresh=1e-9, k_prev=0, variant="reference", device=0):
    mat, xx = _c128(mat), _c128(x)
    y = np.empty_like(xx)
    k = C.c_int()
    s = complex(scale)
    _lib.check(
        _lib.load().mitdvp_expm_dense(
            device,
            {"lanczos": 0, "arnoldi": 1}[integrator],
            int(conserve_norm),
            {"reference": 0, "orthodox": 1}[variant],
            _dp(mat),
            mat.shape[0],
            _dp(xx),
            s.real,
            s.imag,
            thresh,
            k_prev,
            _dp(y),
            C.byref(k),
        )
    )
    return y, k.value


def zgemm(A, B, C0=None, transA=False, conjA=False, transB=False, conjB=False, alpha=1.0, beta=0.0, tile_cfg=-1, reps=0, device=0):
    """C = alpha*op(A)*op(B) + beta*C0 on the MFMA kernel (row-major)."""
    A, B = _c128(A), _c128(B)
    m, k = (A.shape[1], A.shape[0]) if transA else A.shape
    n = B.shape[0] if transB else B.shape[1]
    Cm = np.zeros((m, n), dtype=np.complex128) if C0 is None else _c128(C0).copy()
    al = np.array([complex(alpha).real, complex(alpha).imag])
    be = np.array([complex(beta).real, complex(beta).imag])
    ms = C.c_double()
    _lib.check(
        _lib.load().mitdvp_zgemm(
            device, int(transA), int(conjA), int(transB), int(conjB), m, n, k, _dp(A), _dp(B), _dp(Cm), _dp(al), _dp(be), tile_cfg, reps, C.byref(ms)
        )
    )
    return (Cm, ms.value) if reps else Cm


def thin_to_full(site, gauge: str, delta_rank: int, device=0) -> np.ndarray:
    """``SiteCoef.thin_to_full``: widen an "A" (or "B") isometry by ``delta_rank``
    orthonormal columns (rows) of its orthogonal complement (capped at its dimension)."""
    site = _c128(site)
    l, c, r = site.shape
    if gauge not in ("A", "B"):
        raise ValueError(f"Invalid gauge: {gauge}")
    extra = min(int(delta_rank), l * c - r if gauge == "A" else c * r - l)
    out = np.empty((l, c, r + extra) if gauge == "A" else (l + extra, c, r), np.complex128)
    _lib.check(_lib.load().mitdvp_thin_to_full(device, 0 if gauge == "A" else 1, _dp(site), l, c, r, extra, _dp(out)))
    return out


def clock_probe(iters: int = 200000, device=0) -> dict:
    """Shader clock seen by a short single-wavefront kernel (MHz) and its duration (us)."""
    out = np.zeros(2)
    _lib.check(_lib.load().mitdvp_clock_probe(device, int(iters), _dp(out)))
    return {"mhz": 100.0 * out[0] / max(out[1], 1.0), "us": out[1] / 100.0, "cycles_per_iter": out[0] / iters}


def svd(A, device=0):
    """A = U diag(S) Vh with the engine's one-sided Jacobi kernel; returns (U, S, Vh, sweeps)."""
    A = _c128(A)
    r, c = A.shape
    k = min(r, c)
    U, S, Vh = np.empty((r, k), np.complex128), np.empty(k), np.empty((k, c), np.complex128)
    sw = C.c_int()
    _lib.check(_lib.load().mitdvp_svd(device, _dp(A), r, c, _dp(U), _dp(S), _dp(Vh), C.byref(sw)))
    return U, S, Vh, sw.value


def heff_selfcheck(dl, d, dr, ml, mr, device=0) -> dict:
    """Size-independent checks of one H_eff apply on device-resident operands."""
    out = np.zeros(4)
    _lib.check(_lib.load().mitdvp_heff_selfcheck(device, dl, d, dr, ml, mr, _dp(out)))
    return {"rel_3m_vs_4m": out[0], "linearity_defect": out[1], "norm_Hx": out[2], "ms": out[3]}


class TDVPEnsemble:
    """B independent trajectories (replicas) of one model on ONE GPU, each on its own slice of the chip.

    The small-bond regime (C2) is latency bound: one trajectory's one-launch local exponentials keep 32-128
    of the 256 compute units busy for tens of microseconds at a time.  Here every replica is an engine whose stream is
    confined to ``n_cu / B`` compute units of its own (``mitdvp_config.cu_first / cu_count``), so the replicas' launches
    overlap without admission control, and ``propagate`` is ONE library call (``mitdvp_ensemble_step``: a host thread
    per replica inside the library).  The reference runs trajectories one after the other in a Python loop
    (tests/test_mixedstate.py:269-308).  Results are those of the same engines stepped one at a time, bit for bit.
    """

    def __init__(self, n_replicas: int, nsite: int, *, device: int = 0, n_cu: int | None = None, **engine_kw):
        if n_replicas < 1:
            raise ValueError("n_replicas must be >= 1")
        if n_cu is None:
            n_cu = device_cu_count(device)
        per = (n_cu // n_replicas) // 8 * 8
        if per < 8:
            raise ValueError(f"{n_replicas} replicas do not fit {n_cu} compute units (8 per replica at least)")
        self.cu_per_replica = per
        self.engines = []
        try:
            for r in range(n_replicas):
                self.engines.append(TDVPEngine(nsite, device=device, cu_range=(r * per, per), **engine_kw))
        except Exception:
            self.close()
            raise
        self._lib = _lib.load()

    def __len__(self):
        return len(self.engines)

    def __getitem__(self, i):
        return self.engines[i]

    def set_mpo(self, cores, op_id: int = 0):
        for e in self.engines:
            e.set_mpo(cores, op_id)

    def propagate(self, dt_au: float, nsteps: int = 1):
        """``nsteps`` time steps of every replica, all replicas at once."""
        n = len(self.engines)
        hs = (C.c_void_p * n)(*[e._h for e in self.engines])
        st = (C.c_int * n)()
        rc = self._lib.mitdvp_ensemble_step(hs, n, float(dt_au), int(nsteps), st)
        if rc != 0:
            bad = next(i for i in range(n) if st[i] != 0)
            _lib.check(st[bad], self.engines[bad]._h)

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []


def device_cu_count(device: int = 0) -> int:
    """compute units of the device (256 on MI355X)"""
    n = C.c_int()
    _lib.check(_lib.load().mitdvp_device_cu_count(device, C.byref(n)))
    return n.value


def set_gemm_mode(mode: str):
    """"4m" (textbook complex product) or "3m" (Karatsuba, library default)."""
    _lib.load().mitdvp_set_gemm_mode({"4m": 0, "3m": 1}[mode.lower()])


def set_qr_fast(on: bool):
    """QR panels by CholeskyQR2 + Householder reconstruction (True, library default) or one Householder step per
    launch only (False); process-wide."""
    _lib.load().mitdvp_set_qr_fast(int(bool(on)))


def bench_qr(m: int, n: int, reps: int = 10, device: int = 0):
    """(milliseconds, launches) per m x n QR gauge move on the device (HIP events around `reps` factorisations)."""
    ms, nl = C.c_double(), C.c_long()
    _lib.check(_lib.load().mitdvp_bench_qr(device, m, n, reps, C.byref(ms), C.byref(nl)))
    return ms.value, nl.value


def qr_thin(a=None, shape=None, gauge_free: bool = True, reps: int = 1, device: int = 0):
    """The thin QR of the sweep's gauge moves (``mitdvp_qr_thin``): ``a`` (m x n) or, with ``shape=(m, n)``, a random
    matrix generated on the device.  Returns (Q, R, info) with info = {"ms", "launches", "gauge_free_path"}; Q, R are None
    for a device-generated input."""
    if a is not None:
        a = _c128(a)
        m, n = a.shape
        q, r = np.empty((m, n), dtype=np.complex128), np.empty((n, n), dtype=np.complex128)
    else:
        m, n = shape
        q = r = None
    ms, nl, path = C.c_double(), C.c_long(), C.c_int()
    _lib.check(_lib.load().mitdvp_qr_thin(device, _dp(a) if a is not None else None, m, n, int(bool(gauge_free)),
                                          _dp(q) if q is not None else None, _dp(r) if r is not None else None, reps,
                                          C.byref(ms), C.byref(nl), C.byref(path)))
    return q, r, {"ms": ms.value, "launches": nl.value, "gauge_free_path": bool(path.value)}


def get_qr_fast() -> bool:
    return bool(_lib.load().mitdvp_get_qr_fast())


def device_count() -> int:
    """HIP devices visible to this process (0 on a CPU-only box); does not create a context."""
    n = C.c_int()
    _lib.check(_lib.load().mitdvp_device_count(C.byref(n)))
    return n.value


def device_sync(device: int = 0) -> None:
    _lib.check(_lib.load().mitdvp_device_sync(device))


def get_gemm_mode() -> str:
    return "3m" if _lib.load().mitdvp_get_gemm_mode() else "4m"


def bench_heff(dl, d, dr, ml, mr, reps=3, warmup=1, device=0) -> float:
    ms = C.c_double()
    _lib.check(_lib.load().mitdvp_bench_heff(device, dl, d, dr, ml, mr, reps, warmup, C.byref(ms)))
    return ms.value


def mfma_peak_probe(device=0) -> float:
    out = C.c_double()
    _lib.check(_lib.load().mitdvp_mfma_peak_probe(device, C.byref(out)))
    return out.value


def mfma_layout_probe(device=0) -> np.ndarray:
    out = (C.c_int * 512)()
    _lib.check(_lib.load().mitdvp_mfma_layout_probe(device, out))
    return np.array(out).reshape(64, 4, 2)
