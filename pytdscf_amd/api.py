"""Thin Python shell with PyTDSCF's user surface for the MPO / MPS-SM path
(SURVEY 8b): ``TensorOperator``, ``TensorHamiltonian``, ``BasInfo``, ``Model``,
``Simulator`` and a ``WFunc`` handle.  Same names, argument meaning and error
behaviour as the reference classes cited in each docstring; all numerics go
through ``libmitdvp.so`` (no CPU fallback).  Not supported (raise
``NotImplementedError`` like the reference does for unsupported combos):
multi-state direct-product MPS, SoP/PolynomialHamiltonian, MCTDH SPFs,
MPI site sharding, subspace projection in Liouville space.  ``Simulator.relax``, ``.operate`` and
``.propagate`` chain through ``restart=True`` like in the reference's spectrum workflow.
"""

from __future__ import annotations

import os
import warnings

import numpy as np

from . import units
from .engine import MultiStateEngine, TDVPEngine
from .mps import bond_dims, product_state_cores
from .operators import compress_mpo, merge_operator_terms


class TensorOperator:
    """``TensorOperator(mpo=[cores], legs=(...))`` or a dense grid tensor
    (``shape=`` / ``tensor=``, ``only_diag``) that is decomposed into MPO cores on
    demand -- dvr_operator_cls.py:92-180, :359-485.

    ``legs``: one entry per diagonal 3-leg core, two equal entries per 4-leg core.
    Dense tensors: ``only_diag=True`` -- a function on the DVR grid, shape (n_1, .., n_f),
    decomposed by a tensor-train SVD sweep (``decompose``, the reference's ``_SVD``:
    left-to-right SVDs, ranks cut by the contribution ``rate`` or a bond dimension cap);
    a one-site tensor (vector or matrix) becomes a single core.
    """

    def __init__(self, *, shape=None, tensor=None, only_diag=False, legs=None, name=None, mpo=None):
        self.name = name
        if mpo is not None:
            self.tensor_decomposed = [np.asarray(c) for c in mpo]
            self.only_diag = all(c.ndim == 3 for c in self.tensor_decomposed)
            if legs is None:
                legs = []
                for i, c in enumerate(self.tensor_decomposed):
                    legs += [i] if c.ndim == 3 else [i, i]
            self.legs = tuple(legs)
            self.bond_dimension = [1] + [c.shape[-1] for c in self.tensor_decomposed]
            self._set_sites()
            return
        if tensor is None:
            if shape is None:
                raise ValueError("TensorOperator needs mpo=, tensor= or shape=")
            tensor = np.zeros(tuple(shape), dtype=np.float64)
        self.tensor_orig = np.array(tensor)
        self.shape = self.tensor_orig.shape
        self.only_diag = bool(only_diag)
        if legs is None:
            legs = tuple(range(len(self.shape))) if self.only_diag else tuple(i // 2 for i in range(len(self.shape)))
        self.legs = tuple(legs)
        if not self.only_diag and len(set(self.legs)) > 1:
            raise NotImplementedError("dense non-diagonal multi-site tensors: pass finished MPO cores (mpo=...)")
        self.sites = sorted(set(int(x) for x in self.legs))

    def _set_sites(self):
        sites, it = [], iter(self.legs)
        for c in self.tensor_decomposed:
            s = next(it)
            if c.ndim == 4 and next(it) != s:
                raise ValueError("4-leg core needs two equal consecutive legs")
            sites.append(int(s))
        self.sites = sites

    def decompose(self, bond_dimension=None, decompose_type="SVD", rate=None, square_sum=True, overwrite=False):
        """MPO cores of a dense tensor (dvr_operator_cls.py:423-485).  Tensor-train SVD for
        both ``decompose_type`` values (the reference's "QRD" is the exact, untruncated
        decomposition: that is rate=None, no bond cap here); ranks are cut where the
        cumulative (squared, if ``square_sum``) singular values reach ``rate``."""
        if hasattr(self, "tensor_decomposed") and not overwrite:
            return self.tensor_decomposed
        if rate is not None and not 0.0 < rate < 1.0:
            raise ValueError(f"Contribution rate must be in (0.0, 1.0), but {rate}")
        if decompose_type.lower() not in ("svd", "sv", "qrd", "qr"):
            raise ValueError('decompose_type must be "QRD" or "SVD"')
        if decompose_type.lower() in ("qrd", "qr"):
            rate, bond_dimension = None, None
        t = self.tensor_orig
        if len(set(self.legs)) == 1:  # one site: a diagonal (vector) or a matrix
            self.tensor_decomposed = [t.reshape((1,) + t.shape + (1,))]
        else:
            nsite = t.ndim
            caps = [bond_dimension] * (nsite - 1) if isinstance(bond_dimension, int) else (
                list(bond_dimension)[1:-1] if bond_dimension is not None else [None] * (nsite - 1))
            cores, r, rank = [], t.reshape(1, -1), 1
            for i, n in enumerate(self.shape[:-1]):
                mat = r.reshape(rank * n, -1)
                U, sv, Vh = np.linalg.svd(mat, full_matrices=False)
                keep = len(sv)
                if caps[i] is not None:
                    keep = min(keep, int(caps[i]))
                if rate is not None:
                    w = sv**2 if square_sum else sv
                    tot, cum, k = w.sum(), 0.0, 0
                    while k < keep and (tot == 0.0 or cum / tot < rate):
                        cum += w[k]
                        k += 1
                    keep = max(k, 1)
                cores.append(U[:, :keep].reshape(rank, n, keep))
                r = sv[:keep, None] * Vh[:keep]
                rank = keep
            cores.append(r.reshape(rank, self.shape[-1], 1))
            self.tensor_decomposed = cores
        self.bond_dimension = [1] + [c.shape[-1] for c in self.tensor_decomposed]
        if self.only_diag or len(set(self.legs)) > 1:
            self.legs = tuple(self.legs)
        self._set_sites()
        return self.tensor_decomposed

    def get_tensor_full(self):
        """Dense tensor restored from the cores (dvr_operator_cls.py:267-277, :547-554)."""
        out = self.tensor_decomposed[0]
        for c in self.tensor_decomposed[1:]:
            out = np.tensordot(out, c, axes=(out.ndim - 1, 0))
        return out.reshape(out.shape[1:-1])


def _as_term_blocks(x, nstate=None):
    """potential / kinetic -> [istate][jstate] -> dict (hamiltonian_cls.py:663-669)."""
    if x is None:
        return None
    if isinstance(x, dict):
        return [[x]]
    if isinstance(x, list):  # [[{...}, ...], ...] : [istate][jstate]
        n = len(x)
        if any(not isinstance(row, list) or len(row) != n for row in x):
            raise ValueError("potential/kinetic must be a square [istate][jstate] list of dicts")
        return [[(b or {}) for b in row] for row in x]
    raise TypeError("potential/kinetic must be a dict or [[dict]]")


class TensorHamiltonian:
    """``TensorHamiltonian(ndof, potential, name, kinetic, ..., backend)`` --
    hamiltonian_cls.py:628-752.  Holds the operator dictionaries per (bra, ket) state pair
    (``potential=[[d00, d01], [d10, d11]]``; a plain dict is one state), the scalar terms
    ``coupleJ[i][j]`` (key ``()``, :672-678); ``as_mpo(dims)`` / ``block_mpo(i, j, dims)``
    reduce a pair's dictionary to the single direct-sum MPO the engine contracts."""

    def __init__(self, ndof, potential, name="hamiltonian", kinetic=None, decompose_type="QRD", rate=None,
                 bond_dimension=None, backend="hip"):
        self.ndof = int(ndof)
        self.name = name
        self.backend = backend
        pot = _as_term_blocks(potential)
        kin = _as_term_blocks(kinetic)
        if pot is None:  # kinetic-only operators
            n_ = len(kin) if kin is not None else 1
            pot = [[{} for _ in range(n_)] for _ in range(n_)]
        self.nstate = len(pot)
        if kin is not None and len(kin) != self.nstate:
            raise ValueError("kinetic and potential must have the same number of states")
        self.coupleJ = [[0.0] * self.nstate for _ in range(self.nstate)]  # hamiltonian_cls.py:337-358
        self.terms_ij = [[{} for _ in range(self.nstate)] for _ in range(self.nstate)]
        for i in range(self.nstate):
            for j in range(self.nstate):
                t = self.terms_ij[i][j]
                for k, v in pot[i][j].items():
                    if k == ():
                        if not isinstance(v, (float, complex, int)):
                            raise ValueError(f"scalar term must be scalar but {v} is {type(v)}")
                        self.coupleJ[i][j] = v
                        continue
                    t[k] = v
                for k, v in (kin[i][j] if kin is not None else {}).items():
                    if k in t:
                        raise ValueError(f"key {k} is already set in potential. Concatenate KEO and PEO or set KEO as SOP")
                    t[k] = v
                for op in t.values():  # dense grid tensors -> MPO cores (hamiltonian_cls.py:705-722)
                    if not hasattr(op, "tensor_decomposed"):
                        op.decompose(bond_dimension=bond_dimension, decompose_type=decompose_type, rate=rate)
        self.terms = self.terms_ij[0][0]

    def block_mpo(self, i, j, dims, compress=True):
        """Full-chain 4-leg MPO of the (bra i, ket j) block, or None when the pair has no operator."""
        t = self.terms_ij[i][j]
        if not t:
            return None
        mpo = merge_operator_terms([(op.tensor_decomposed, op.sites) for op in t.values()], dims)
        return compress_mpo(mpo) if compress and len(mpo) > 1 else mpo

    def as_mpo(self, dims, compress=True):
        """One full-chain 4-leg MPO: exact direct sum of the operator terms, then a lossless
        rounding (``compress_mpo``, singular values below 1e-13 relative dropped) that removes
        the linear dependencies the direct sum introduces."""
        if self.nstate != 1:
            raise ValueError("as_mpo is the single-state form; use block_mpo(i, j, dims)")
        mpo = merge_operator_terms([(op.tensor_decomposed, op.sites) for op in self.terms.values()], dims)
        return compress_mpo(mpo) if compress and len(mpo) > 1 else mpo

    def apply_backend(self, backend):
        self.backend = backend

    def one_site_gates(self, dims):
        """``{site: U}`` of a gate operator (every key acts on one site; diagonal 3-leg
        or full 4-leg core with unit bonds), what ``apply_one_gate`` reads from
        ``mpo.calc_point`` (_mps_cls.py:2346-2361, :2420-2451)."""
        gates = {}
        for key, op in self.terms.items():
            if len(op.tensor_decomposed) != 1:
                raise ValueError(f"a gate operator acts on one site per key, got {key}")
            core, site = op.tensor_decomposed[0], op.sites[0]
            if site in gates:
                raise ValueError("Multiple one gate on same site is not supported. Contract gates in advance!")
            if core.shape[0] != 1 or core.shape[-1] != 1:
                raise ValueError("a gate core must have unit MPO bonds")
            U = np.diag(core[0, :, 0]) if core.ndim == 3 else core[0, :, :, 0]
            if U.shape[0] != dims[site]:
                raise ValueError(f"gate on site {site} has dimension {U.shape[0]}, the basis {dims[site]}")
            gates[site] = np.asarray(U, dtype=np.complex128)
        return gates


class BasInfo:
    """``BasInfo(prim_info)`` -- model_cls.py:323-.  prim_info[istate][idof]."""

    def __init__(self, prim_info, spf_info=None, ndof_per_sites=None):
        if not isinstance(prim_info[0], (list, tuple)):
            prim_info = [prim_info]
        self.prim_info = [list(p) for p in prim_info]
        if any(len(p) != len(self.prim_info[0]) for p in self.prim_info):
            raise ValueError("every electronic state needs the same number of degrees of freedom")
        self.is_DVR = True
        self.is_standard_method = True

    def get_nstate(self):
        return len(self.prim_info)

    def get_ndof(self):
        return len(self.prim_info[0])

    def get_primbas(self, istate, idof):
        return self.prim_info[istate][idof]

    def get_nprim(self, istate, idof):
        b = self.prim_info[istate][idof]
        return b.nprim if hasattr(b, "nprim") else len(b)


class Model:
    """``Model(basinfo, operators, *, bond_dim, ...)`` -- model_cls.py:64-120.

    operators: TensorHamiltonian | list of MPO cores | dict with keys
    "hamiltonian" or "potential"(+"kinetic") and observables (model_cls.py:215-284).
    Attributes ``m_aux_max``, ``init_HartreeProduct`` (weights or explicit 3-D
    cores, _site_cls.py:446-471), ``init_weight_VIBSTATE`` (HO-eigenbasis weights
    rotated to the DVR grid, _mps_mpo.py:96-110)."""

    def __init__(self, basinfo, operators, *, bond_dim=None, build_td_hamiltonian=None, space="hilbert",
                 subspace_inds=None, one_gate_to_apply=None, kraus_op=None):
        self.basinfo = basinfo if isinstance(basinfo, BasInfo) else BasInfo(basinfo)
        if space.lower() not in ("hilbert", "liouville"):
            raise ValueError(f"space must be 'hilbert' or 'liouville' but got {space}")
        if build_td_hamiltonian is not None:
            raise NotImplementedError("time-dependent Hamiltonians (const.doTDHamil is never enabled by the reference either)")
        if kraus_op is not None and not isinstance(kraus_op, dict):
            raise TypeError("kraus_op must be a dict {(site,) | (site, site + 1): array (k, d, d)}")
        self.kraus_op = kraus_op
        if one_gate_to_apply is not None and not isinstance(one_gate_to_apply, TensorHamiltonian):
            raise TypeError("one_gate_to_apply must be a TensorHamiltonian of one-site operators")
        self.one_gate_to_apply = one_gate_to_apply
        self.space = space.lower()
        ops = {"hamiltonian": operators} if isinstance(operators, (TensorHamiltonian, list)) or hasattr(operators, "to_tensor_hamiltonian") else dict(operators)
        self.dims = [self.basinfo.get_nprim(0, i) for i in range(self.basinfo.get_ndof())]
        self.nstate = self.basinfo.get_nstate()
        for s_ in range(1, self.nstate):
            if [self.basinfo.get_nprim(s_, i) for i in range(len(self.dims))] != self.dims:
                raise NotImplementedError("electronic states with different primitive-basis sizes per site")
        if self.nstate > 1 and (space.lower() != "hilbert" or one_gate_to_apply is not None or kraus_op is not None):
            raise NotImplementedError("several electronic states: Hilbert space without gates / Kraus maps only")
        out = {}
        if "potential" in ops:
            if "hamiltonian" in ops:
                raise ValueError("Cannot specify 'hamiltonian' when 'potential' is given.")
            pot, kin = ops.pop("potential"), ops.pop("kinetic", None)
            out["hamiltonian"] = TensorHamiltonian(
                len(self.dims), {"potential": TensorOperator(mpo=pot)},
                kinetic=None if kin is None else {"kinetic": TensorOperator(mpo=kin)})
        for name, op in ops.items():
            if hasattr(op, "to_tensor_hamiltonian"):  # PolynomialHamiltonian: exact MPO of the sum of products
                op = op.to_tensor_hamiltonian(self.basinfo)
            if isinstance(op, TensorHamiltonian):
                out[name] = op
            elif isinstance(op, list):
                if len(op) != len(self.dims):
                    raise ValueError(f"Operator {name} length must be equal to ndof of basis. But, got {len(op)} and {len(self.dims)}.")
                out[name] = TensorHamiltonian(len(self.dims), {name: TensorOperator(mpo=op)})
            else:
                raise TypeError(f"Operator {name} must be HamiltonianMixin or list of arrays.")
        self.hamiltonian = out.pop("hamiltonian")
        self.observables = out
        for name, op in dict(out, hamiltonian=self.hamiltonian).items():
            if op.nstate != self.nstate:
                raise ValueError(f"operator {name} has {op.nstate} electronic state(s), the basis {self.nstate}")
        self.m_aux_max = bond_dim
        self.use_mpo = True
        # subspace projection of Liouville-space sites (model_cls.py:110-120): ignored in Hilbert space, as there
        self.subspace_inds = None
        if self.space == "liouville" and subspace_inds is not None:
            if not isinstance(subspace_inds, dict):
                raise TypeError("subspace_inds must be a dict {site: tuple of kept physical indices}")
            sub = {}
            for site, inds in subspace_inds.items():
                inds = tuple(int(x) for x in inds)
                if not 0 <= int(site) < len(self.dims) or not inds or len(set(inds)) != len(inds) or min(inds) < 0 or max(inds) >= self.dims[int(site)]:
                    raise ValueError(f"subspace_inds[{site}] = {inds} does not select distinct entries of the site's {self.dims[int(site)]} physical indices")
                sub[int(site)] = inds
            self.subspace_inds = sub
            if self.nstate > 1:
                raise NotImplementedError("Only one state is supported")  # _mps_mpo.py:139, hamiltonian_cls.py:853
        self.init_HartreeProduct = None
        self.init_weight_VIBSTATE = None
        self.init_weight_ESTATE = None  # _get_initial_condition, _mps_cls.py:150-167

    def get_nstate(self):
        return self.nstate

    def estate_weights(self):
        """Normalised weights of the electronic states (_mps_cls.py:150-167); default: all in state 0."""
        if self.init_weight_ESTATE is None:
            return [1.0] + [0.0] * (self.nstate - 1)
        w = np.asarray(self.init_weight_ESTATE, dtype=float)
        if len(w) != self.nstate:
            raise ValueError("The length of weight_estate must be equal to nstate.")
        w = w / w.sum()
        if w.min() < 0.0:
            raise ValueError("The elements of weight_estate must be positive.")
        return [float(x) for x in w]

    def get_ndof(self):
        return len(self.dims)

    def projected_dims(self):
        """Site dimensions after the subspace projection (LatticeInfo.dim_of_sites, _mps_mpo.py:212-213)."""
        sub = self.subspace_inds or {}
        return [len(sub[i]) if i in sub else d for i, d in enumerate(self.dims)]

    def project_mpo(self, cores):
        """``TensorHamiltonian.project_subspace`` (hamiltonian_cls.py:852-880) on the merged MPO: bra and ket legs of
        the named sites restricted to the kept indices (slicing commutes with the direct sum of the terms)."""
        sub = self.subspace_inds or {}
        out = []
        for i, w in enumerate(cores):
            w = np.asarray(w)
            if i in sub:
                ket, bra = np.ix_(sub[i], sub[i])
                w = w[:, ket, bra, :] if w.ndim == 4 else w[:, list(sub[i]), :]
            out.append(np.ascontiguousarray(w))
        return out

    def initial_cores(self, istate=0):
        D = self.m_aux_max if self.m_aux_max is not None else 10**9  # _get_initial_condition, _mps_cls.py:147-148
        if self.init_HartreeProduct is not None:
            if len(self.init_HartreeProduct) != self.nstate:
                raise ValueError("init_HartreeProduct needs one list of site weights / cores per electronic state")
            return product_state_cores(self.init_HartreeProduct[istate], D, space=self.space)
        if self.init_weight_VIBSTATE is not None:
            if len(self.init_weight_VIBSTATE) != self.nstate:
                raise ValueError("The length of weight_vib must be equal to nstate.")
            w = self.init_weight_VIBSTATE[istate]
        else:
            w = [[1.0] + [0.0] * (d - 1) for d in self.dims]
        cores = product_state_cores(w, D)
        rot = []
        for c, b in zip(cores, self.basinfo.prim_info[istate]):
            rot.append(np.einsum("abc,bd->adc", c, b.get_unitary()) if hasattr(b, "get_unitary") else c)
        return rot


class WFunc:
    """Handle returned by ``Simulator.propagate`` (wavefunction.py:34-): the
    state lives on the GPU inside ``self.engine``."""

    def __init__(self, engine: TDVPEngine, op_ids: dict, space: str = "hilbert"):
        self.engine = engine
        self._op_ids = op_ids
        self.space = space

    def norm(self):
        return self.engine.norm()

    def pop_states(self):
        if hasattr(self.engine, "pop_states"):  # several electronic states
            return self.engine.pop_states()
        return [self.engine.norm() ** 2]

    def autocorr(self):
        return self.engine.autocorr()

    def bonddim(self):
        """Bond dimensions of the MPS (wavefunction.py:151-167)."""
        return self.engine.bond_dims()

    def hermitise(self):
        """``wf.ci_coef.hermitise()`` (MPSCoef.hermitise, _mps_cls.py:2289-2312): rho <- (rho + rho^dagger) / 2 of
        the matrix-product density operator, on the device (two-site SVDs back to the old bond dimensions)."""
        if self.space != "liouville":
            raise ValueError("hermitise needs space='liouville'")
        self.engine.hermitise()

    def apply_one_gate(self, matOp, reorth_center: int = 0):
        """``WFunc.apply_one_gate`` (wavefunction.py:588-598): one-site operators applied
        to the resting state; ``reorth_center`` must be the current centre site."""
        eng = self.engine
        if reorth_center != eng._center():
            raise ValueError(f"reorth_center={reorth_center} is not the centre site of the MPS ({eng._center()})")
        dims = [eng.get_site_shape(i)[1] for i in range(eng.nsite)]
        keep = dict(getattr(eng, "_step_gates", {}))
        eng.set_gates(matOp.one_site_gates(dims))
        eng.apply_gates()
        eng.set_gates(keep)

    def expectation(self, matOp):
        name = matOp if isinstance(matOp, str) else self._name_of(matOp)
        if self.space == "liouville":  # Tr(O rho), _exp_liouville
            if name == "hamiltonian":
                raise NotImplementedError("Liouville space: the 'hamiltonian' is a super-operator; expectation values are for n-dimensional observables")
            v = self.engine.expect_trace(self._op_ids[name])
        else:
            v = self.engine.expectation(self._op_ids[name])
        if abs(np.angle(v)) > 1e-2 and abs(abs(np.angle(v)) - np.pi) > 1e-2:
            warnings.warn(f"Expectation value {v} is not real, probably due to non-Hermitian operator or numerical error.")
        return v.real

    def get_reduced_densities(self, remain_nleg):
        """``WFunc.get_reduced_densities`` (wavefunction.py:67-88): one tuple of kept legs
        per site, e.g. (0, 0, 0, 2) = both legs of site 3, or a list of such tuples."""
        keys = remain_nleg if isinstance(remain_nleg, list) else [remain_nleg]
        if self.space == "liouville":  # get_partial_trace, _mps_cls.py:1438-1510
            return [self.engine.partial_trace(k) for k in keys]
        return [self.engine.reduced_density(k) for k in keys]

    def _name_of(self, matOp):
        for name, op in getattr(self, "_ops_by_name", {}).items():
            if op is matOp:
                return name
        return getattr(matOp, "name", "hamiltonian")

    def get_mps(self):
        if hasattr(self.engine, "get_states"):  # [istate][isite], superblock_states
            return self.engine.get_states()
        return self.engine.get_mps()

    @property
    def ci_coef(self):
        """``wf.ci_coef.<method>`` of reference scripts: the MPS coefficients live in this handle."""
        return self

    def get_CI_coef_state(self, J=None, trans_arrays=None, istate: int = 0):
        """Coefficient <j_1 j_2 ... j_f|Psi> of one basis configuration ``J``, or the contraction of
        every physical leg with a vector (``trans_arrays``, e.g. coherent-state overlaps)
        (MPSCoef.get_CI_coef_state, _mps_cls.py:1680-1736)."""
        mps = self.get_mps()
        cores = mps[istate] if hasattr(self.engine, "get_states") else mps
        if trans_arrays is None:
            if J is None:
                raise ValueError("Either `J` or `trans_arrays` must be set.")
            trans_arrays = []
            for c, j in zip(cores, J):
                v = np.zeros(c.shape[1], dtype=complex)
                v[j] = 1.0
                trans_arrays.append(v)
        elif len(trans_arrays) != len(cores):
            raise ValueError("The length of `trans_arrays` must be equal to the number of DOFs.")
        row = np.ones((1,), dtype=complex)
        for c, t in zip(cores, trans_arrays):
            row = row @ np.tensordot(c, np.asarray(t), axes=(1, 0))
        return complex(row[0])


def _add_scalar_to_mpo(mpo, c):
    """MPO of H + c: one more bond route carrying c on the first site and identities after it."""
    out = []
    n = len(mpo)
    for i, w in enumerate(mpo):
        ml, d, _, mr = w.shape
        a, b = (ml if i == 0 else ml + 1), (mr if i == n - 1 else mr + 1)
        z = np.zeros((a, d, d, b), dtype=np.complex128)
        z[:ml, :, :, :mr] = w
        eye = np.eye(d, dtype=np.complex128) * (c if i == 0 else 1.0)
        z[a - 1, :, :, b - 1] += eye
        out.append(z)
    return out


class _ShardedEngine:
    """What the property loop of ``Simulator.propagate`` asks of an engine, answered by one rank of the site-sharded
    sweep; every call is collective over the ranks."""

    def __init__(self, sh, ops, ham):
        self.sh, self.ops, self.ham, self.rank = sh, ops, ham, sh.rank

    def autocorr(self):
        return self.sh.autocorr()

    def norm(self):
        return self.sh.norm()

    def expectation(self, k):
        return self.sh.expectation(None if k == 0 else self.ops[k])

    def reduced_density(self, legs):
        return self.sh.reduced_density([i for i, n in enumerate(legs) for _ in range(n)])

    def propagate(self, dt):
        self.sh.step(dt)

    def bond_dims(self):
        return self.sh.bond_dims()

    def close(self):
        self.sh.close()


class Simulator:
    """``Simulator(jobname, model, ci_type="MPS", backend="hip", ...)`` --
    simulator_cls.py:58-94; ``propagate`` follows :160-285 and the loop of
    ``_execute`` (:400-454): observables are evaluated BEFORE each step and the
    returned energy is the one of the last loop iteration (Appendix B.1)."""

    def __init__(self, jobname, model, ci_type="MPS", backend="hip", proj_gs=False, t2_trick=True, verbose=2):
        if ci_type.lower() not in ("mps", "mps-sm", "standard-method"):
            raise NotImplementedError("only the MPS standard method (MPO Hamiltonian) is on the accelerated path")
        if backend.lower() not in ("hip", "numpy", "jax"):
            raise ValueError(f"unknown backend {backend}")
        if backend.lower() != "hip":
            warnings.warn("pytdscf_amd always runs on the MI355X HIP engine; backend string accepted for script compatibility")
        if proj_gs:
            raise NotImplementedError("proj_gs")
        self.jobname, self.model, self.t2_trick, self.verbose = jobname, model, t2_trick, verbose

    # ---- checkpoint / restart (simulator_cls.py:413-418, :500-506, :577-589) ------------
    # dill pickles wf_{jobname}{ext}.pkl at the reference's call sites and with the attribute graph its readers use
    # (wf.ci_coef.superblock_states[istate][isite].data / .gauge / .isite): pytdscf_amd/checkpoint.py.  Restart also
    # accepts the wf_*.npz container earlier versions of the shell wrote.
    def _wf_path(self, ext):
        return f"wf_{self.jobname}{ext}.pkl"

    def save_wavefunction(self, wf, ext=""):
        from . import checkpoint

        if wf is None:  # not the writing rank of a sharded run (simulator_cls.py:584)
            return

        checkpoint.save(self._wf_path(ext), wf.engine, self.model.space, self.model.nstate)

    def _load_states(self, ext):
        """[[(tensor, gauge), ...] per state] from wf_{jobname}{ext}.pkl (or the older .npz)"""
        from . import checkpoint

        path = self._wf_path(ext)
        n = len(self.model.dims)
        if os.path.exists(path):
            ci = checkpoint.load(path).ci_coef
            states = [[(np.asarray(s_.data), s_.gauge) for s_ in st] for st in ci.superblock_states]
        elif os.path.exists(path[:-4] + ".npz"):
            z = np.load(path[:-4] + ".npz")
            names = {0: "Psi", 1: "A", 2: "B", -1: "C"}
            if "nstate" in z:
                states = [[(z[f"site{s_}_{i}"], "C") for i in range(int(z["nsite"]))] for s_ in range(int(z["nstate"]))]
            else:
                states = [[(z[f"site{i}"], names[int(g)]) for i, g in zip(range(int(z["nsite"])), z["gauges"])]]
        else:
            raise FileNotFoundError(f"restart=True but {path} does not exist")
        if len(states) != self.model.nstate or any(len(st) != n or [c.shape[1] for c, _ in st] != list(self.model.projected_dims()) for st in states):
            raise ValueError(f"{path} does not match the model's sites / states")
        return states

    def _load_cores(self, ext):
        st = self._load_states(ext)[0]
        return [c for c, _ in st], [g for _, g in st]

    def _engine_multistate(self, integrator, conserve_norm, thresh, relax=False, restart_ext=None):
        """nstate > 1: one MPS per electronic state, Hamiltonian / observable blocks per state pair."""
        m = self.model
        eng = MultiStateEngine(len(m.dims), m.nstate, integrator=integrator, conserve_norm=conserve_norm, thresh=thresh,
                               relax=relax)
        ids = {}
        for k, (name, op) in enumerate([("hamiltonian", m.hamiltonian)] + list(m.observables.items())):
            blocks = [[op.block_mpo(i, j, m.dims) for j in range(m.nstate)] for i in range(m.nstate)]
            eng.set_hamiltonian(blocks, op.coupleJ, op_id=k)
            ids[name] = k
        if restart_ext is None:
            eng.set_states([m.initial_cores(s_) for s_ in range(m.nstate)], weights=m.estate_weights())
        else:
            for s_, st in enumerate(self._load_states(restart_ext)):  # saved in the site-0-centred canonical form
                eng.set_state(s_, [c for c, _ in st])
        return eng, ids

    def _engine(self, integrator, conserve_norm, thresh, relax=False, restart_ext=None):
        m = self.model
        if m.nstate > 1:
            return self._engine_multistate(integrator, conserve_norm, thresh, relax, restart_ext)
        liou = m.space == "liouville"
        eng = TDVPEngine(len(m.dims), integrator=integrator, conserve_norm=conserve_norm, thresh=thresh, relax=relax)
        ids = {"hamiltonian": 0}
        sub = m.subspace_inds or {}
        eng.set_mpo(m.project_mpo(m.hamiltonian.as_mpo(m.dims)), 0, shift=m.hamiltonian.coupleJ[0][0])
        for site, inds in sub.items():  # before the trace observables: their cores are gathered on arrival
            eng.set_subspace(site, int(round(m.dims[site] ** 0.5)), inds)
        for k, (name, op) in enumerate(m.observables.items(), start=1):
            if liou:  # observables act on the n-dimensional Hilbert-space legs, site dim = n*n
                eng.set_trace_op(op.as_mpo([int(round(d ** 0.5)) for d in m.dims]), k)
            else:
                eng.set_mpo(op.as_mpo(m.dims), k)
            ids[name] = k
        # Liouville space keeps the (trace) normalisation of the initial state (_mps_cls.py:2695-2699)
        if restart_ext is None and sub:
            # MPSCoefMPO.project_subspace (_mps_mpo.py:196-220) slices the ALREADY canonicalised cores and trims the
            # bonds to the projected lattice's caps; nothing is re-orthogonalised afterwards, and neither is it here
            full = TDVPEngine(len(m.dims))
            full.set_mps(m.initial_cores(), canonicalize=True, scale=None)
            cores = [c[:, list(sub[i]), :] if i in sub else c for i, c in enumerate(full.get_mps())]
            full.close()
            caps = bond_dims(m.projected_dims(), m.m_aux_max if m.m_aux_max is not None else 10**9)
            for i, (c, (dl_, dr_)) in enumerate(zip(cores, caps)):
                eng.set_site(i, np.ascontiguousarray(c[:dl_, :, :dr_]), "Psi" if i == 0 else "B")
        elif restart_ext is None:
            eng.set_mps(m.initial_cores(), canonicalize=True, scale=None if liou else 1.0)
        else:  # const.doRestart: continue from the saved state, gauges as saved
            cores, gauges = self._load_cores(restart_ext)
            if gauges.count("Psi") != 1 or gauges.index("Psi") != 0 or any(g != "B" for g in gauges[1:]):
                raise ValueError("the saved wavefunction is not in the site-0-centred canonical form")
            for i, (c, g_) in enumerate(zip(cores, gauges)):
                eng.set_site(i, c, g_)
        if m.one_gate_to_apply is not None:  # applied between the half-sweeps of every step (_mps_cls.py:489-490)
            gates = m.one_gate_to_apply.one_site_gates(m.dims)
            for site, inds in sub.items():  # project_subspace of the gate operator, model_cls.py:117-118
                if site in gates:
                    U = np.asarray(gates[site])
                    gates[site] = U[list(inds)] if U.ndim == 1 else U[np.ix_(inds, inds)]
            eng.set_gates(gates)
        if m.kraus_op:  # after the gates, _mps_cls.py:491-492
            eng.set_kraus(m.kraus_op)
        return eng, ids

    def _wfunc(self, eng, ids):
        wf = WFunc(eng, ids, self.model.space)
        wf._ops_by_name = dict(self.model.observables, hamiltonian=self.model.hamiltonian)
        return wf

    def propagate(self, stepsize=0.1, maxstep=5000, restart=False, savefile_ext="", loadfile_ext="_operate",
                  backup_interval=1000, autocorr=True, energy=True, norm=True, populations=True, observables=False,
                  reduced_density=None, Δt=None, thresh_sil=1.0e-09, autocorr_per_step=1, observables_per_step=1,
                  energy_per_step=1, norm_per_step=1, populations_per_step=1, parallel_split_indices=None,
                  adaptive=False, adaptive_Dmax=20, adaptive_dD=5, adaptive_p_proj=1.0e-04, adaptive_p_svd=1.0e-07,
                  integrator="lanczos", display_time_unit="fs", conserve_norm=True):
        if integrator not in ("lanczos", "arnoldi"):
            raise ValueError(f"Invalid integrator: {integrator}")
        sharded = parallel_split_indices is not None
        dt_fs = Δt if Δt is not None else stepsize
        dt_au = dt_fs / units.au_in_fs
        liou = self.model.space == "liouville"
        if liou:
            if conserve_norm:
                conserve_norm = False  # forced in Liouville space, _const_cls.py:219-224
            if energy:
                raise NotImplementedError("Liouville space: pass energy=False (the Hamiltonian entry is the super-operator), like tests/test_mixedstate.py:434")
            if autocorr:
                autocorr = False
        multi = self.model.nstate > 1
        if multi and (adaptive or reduced_density is not None or not self.t2_trick):
            raise NotImplementedError("several electronic states: adaptive bonds, reduced densities and t2_trick=False are not implemented")
        lead = True  # the rank that writes the files (const.mpi_rank == 0 in the reference)
        if sharded:
            # const.mpi_size > 1 (simulator_cls.py:364-370, :526-531): one site range per rank, each on its own GPU
            if multi or liou or restart or not self.t2_trick or self.model.one_gate_to_apply is not None or self.model.kraus_op:
                raise NotImplementedError("parallel_split_indices: one electronic state in Hilbert space with t2_trick, "
                                          "without restart, gates or Kraus operators")
            eng, ids = self._sharded_engine(parallel_split_indices, integrator, conserve_norm, thresh_sil, adaptive_p_svd,
                                            dict(Dmax=adaptive_Dmax, dD=adaptive_dD, p_proj=adaptive_p_proj) if adaptive else None)
            lead = eng.rank == 0
        else:
            eng, ids = self._engine(integrator, conserve_norm, thresh_sil, restart_ext=loadfile_ext if restart else None)
        if adaptive and not sharded:  # const.adaptive / Dmax / dD / p_proj (_const_cls.py:212-216); p_svd is unused there too (:968-983)
            eng.set_adaptive(True, Dmax=adaptive_Dmax, dD=adaptive_dD, p_proj=adaptive_p_proj)
        wf = None if sharded else self._wfunc(eng, ids)
        outdir = f"{self.jobname}_prop"
        if lead:
            os.makedirs(outdir, exist_ok=True)
        tconv = {"fs": units.au_in_fs, "ps": units.au_in_fs * 1e-3, "au": 1.0}[display_time_unit]
        names = ("autocorr", "populations", "expectations") + (("bonddim",) if adaptive else ())
        files = {k: open(os.path.join(outdir, f"{k}.dat") if lead else os.devnull, "w") for k in names}
        self.rdm_trace = []
        ener = None
        try:
            for istep in range(maxstep):
                t = istep * dt_au * tconv
                if autocorr and istep == 0 and not self.t2_trick:
                    eng.save_reference()  # wf_init = deepcopy(wf), simulator_cls.py:386-393
                if autocorr and istep % autocorr_per_step == 0:
                    a = eng.autocorr() if self.t2_trick else eng.overlap_reference()
                    if istep == 0:
                        files["autocorr"].write(f"# time [{display_time_unit}]\t auto-correlation\n")
                    files["autocorr"].write(f"{(2 * t if self.t2_trick else t):6.9f}\t{a.real: 6.9f}{a.imag:+6.9f}j\n")
                if populations and istep % populations_per_step == 0:
                    pops = eng.pop_states() if multi else [eng.norm() ** 2]
                    if istep == 0:
                        files["populations"].write(f"# time [{display_time_unit}]\t" + "".join(f"pop_{i_:<7}" for i_ in range(len(pops))) + "\n")
                    files["populations"].write(f"{t:6.9f}\t" + "".join(f"{p_:6.9f}\t" for p_ in pops) + "\n")
                row = {}
                if energy and istep % energy_per_step == 0:
                    ener = eng.expectation(0).real
                    row["energy"] = ener
                if observables and istep % observables_per_step == 0:
                    for name, k in ids.items():
                        if k:
                            row[name] = (eng.expect_trace(k) if liou else eng.expectation(k)).real
                if row:
                    if istep == 0:
                        files["expectations"].write(f"# time [{display_time_unit}]\t" + "\t".join(f"{k:<11}" for k in row) + "\n")
                    files["expectations"].write(f"{t:6.9f}\t" + "".join(f"{v:6.9f}\t" for v in row.values()) + "\n")
                if adaptive:  # properties.py:255-262, :344-356
                    bd = eng.bond_dims()
                    if istep == 0:
                        files["bonddim"].write(f"# time [{display_time_unit}]\t" + "\t".join(f"{i}" for i in range(len(bd))) + "\n")
                    files["bonddim"].write(f"{t:6.9f}\t" + "\t".join(f"{b}" for b in bd) + "\n")
                if reduced_density is not None and istep % reduced_density[1] == 0:
                    rec = {}
                    for key in reduced_density[0]:
                        # key (3, 3) -> remain_nleg (0, 0, 0, 2), properties.py:69-82
                        legs = [0] * (max(key) + 1)
                        for site in key:
                            legs[site] += 1
                        rec[tuple(key)] = eng.partial_trace(legs) if liou else eng.reduced_density(legs)
                    self.rdm_trace.append((t, rec))
                if istep % backup_interval == backup_interval - 1:
                    self.save_wavefunction(self._gathered_wfunc(eng, integrator, conserve_norm, thresh_sil) if sharded else wf,
                                           savefile_ext)
                eng.propagate(dt_au)
            if sharded:
                wf = self._gathered_wfunc(eng, integrator, conserve_norm, thresh_sil)
            self.save_wavefunction(wf, savefile_ext)
            if lead and reduced_density is not None and self.rdm_trace:
                # reduced_density.nc in the reference's layout (properties.py:156-209): time(step) and
                # rho_{key}_{istate}(step, Q.., Q..); NETCDF4 compound type with netCDF4, NetCDF-3 + (re, im) axis without
                from .util.nc_writer import write_reduced_density_nc

                write_reduced_density_nc(os.path.join(outdir, "reduced_density.nc"), [t for t, _ in self.rdm_trace],
                                         [r for _, r in self.rdm_trace])
        finally:
            for f in files.values():
                f.close()
            if sharded:
                eng.close()
        return ener, wf

    def _sharded_engine(self, split, integrator, conserve_norm, thresh, p_svd, adaptive=None):
        """The engine of ``propagate(parallel_split_indices=...)``: this rank's site range of the real-space parallel
        sweep (MPSCoefParallel, _mps_parallel.py), with the reference's junction regularisation and ``adaptive_p_svd``
        truncation always on as there.  Ranks come from RANK / WORLD_SIZE / MASTER_* (torchrun), one GPU each."""
        from .dist import world_comm
        from .parallel_sites import SiteShardedTDVP

        m = self.model
        comm = world_comm(site_sharding=True)
        if len(split) != comm.world:  # _const_cls.py:237
            raise ValueError(f"parallel_split_indices has {len(split)} ranges but the job has {comm.world} rank(s)")
        ids = {"hamiltonian": 0}
        ids.update({name: k for k, name in enumerate(m.observables, start=1)})
        box = [None]
        if comm.rank == 0:  # the operators may live on rank 0 alone, as in the reference's MPI test
            ops = {0: m.hamiltonian.as_mpo(m.dims)}
            ops.update({ids[name]: op.as_mpo(m.dims) for name, op in m.observables.items()})
            box = [(ops, m.hamiltonian.coupleJ[0][0])]
        if comm.world > 1:  # TensorHamiltonian.distribute_mpo_cores (simulator_cls.py:364-370)
            comm.dist.broadcast_object_list(box, src=0)
        ops, shift = box[0]
        mpo = [np.array(w, dtype=np.complex128) for w in ops[0]]
        if shift:  # the scalar term of the Hamiltonian rides on the first core's identity route
            mpo = _add_scalar_to_mpo(mpo, shift)
        sh = SiteShardedTDVP(comm, mpo, cores=m.initial_cores(), integrator=integrator, thresh=thresh,
                             conserve_norm=conserve_norm, split=[tuple(r) for r in split], regularize=True, p_svd=p_svd,
                             adaptive=adaptive)
        return _ShardedEngine(sh, ops, mpo), ids

    def _gathered_wfunc(self, eng, integrator, conserve_norm, thresh):
        """MPSCoefParallel.to_MPSCoefMPO (simulator_cls.py:578-583): the whole chain on rank 0 as an ordinary wave
        function (site-0-centred canonical form, norm kept); None on the other ranks.  Collective."""
        cores = eng.sh.gather()
        if eng.rank != 0:
            return None
        full, ids = self._engine(integrator, conserve_norm, thresh)
        full.set_mps(cores, canonicalize=True, scale=None)
        return self._wfunc(full, ids)

    def relax(self, stepsize=0.1, maxstep=20, improved=True, restart=False, savefile_ext="_gs", loadfile_ext="",
              backup_interval=10, norm=True, populations=True, observables=False, integrator="lanczos",
              display_time_unit="fs", thresh_sil=1.0e-09):
        """Imaginary-time relaxation (simulator_cls.py:95-159; exp(-H dt/2) with
        renormalisation, _mps_cls.py:1086-1094) or, with ``improved=True`` (the
        reference's default), Lanczos diagonalisation of H_eff per site
        (_integrator.py:74-138, _mps_cls.py:1078-1084)."""
        eng, ids = self._engine(integrator, True, thresh_sil, relax="improved" if improved else True,
                                restart_ext=loadfile_ext if restart else None)
        wf = self._wfunc(eng, ids)
        dt_au = stepsize / units.au_in_fs
        ener = None
        for istep in range(maxstep):
            ener = eng.expectation(0).real
            if istep % backup_interval == backup_interval - 1:
                self.save_wavefunction(wf, savefile_ext)
            eng.propagate(dt_au)
        self.save_wavefunction(wf, savefile_ext)
        return ener, wf

    def operate(self, maxstep=10, restart=False, savefile_ext="_operate", loadfile_ext="_gs", verbose=2):
        """``Simulator.operate`` (simulator_cls.py:286-331, :356-360): apply the Model's
        "hamiltonian" entry (e.g. a dipole operator) to the wavefunction variationally
        (``WFunc.apply_dipole``, at most ``maxstep`` double sweeps, converged when
        |1 - |<phi_i|phi_(i-1)>|| < 1e-8); returns (norm of O|Psi>, WFunc) and saves the state."""
        eng, ids = self._engine("lanczos", True, 1.0e-9, restart_ext=loadfile_ext if restart else None)
        norm, iters = eng.operate(0, maxstep)
        if iters >= maxstep:
            warnings.warn(f"Operate O|Ψ> is not converged in {maxstep - 1} iterations")
        wf = self._wfunc(eng, ids)
        self.save_wavefunction(wf, savefile_ext)
        return norm, wf
