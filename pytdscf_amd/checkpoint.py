"""Wavefunction checkpoints ``wf_{jobname}{savefile_ext}.pkl`` written with ``dill`` at the reference's call sites
(``Simulator.save_wavefunction``, pytdscf/simulator_cls.py:577-589: start of a run, every ``backup_interval`` steps,
end of a run; restart loads ``wf_{jobname}{loadfile_ext}.pkl``, :478-507).

The reference pickles its ``WFunc`` object graph; what a restart or an analysis script reads from it is
``wf.ci_coef.superblock_states[istate][isite]`` with ``.data`` (ndarray (D_l, d, D_r)), ``.gauge``
("Psi" | "A" | "B" | "C") and ``.isite``, plus ``wf.ci_coef.nsite`` / ``nstate``.  The state here lives on the GPU, so the
pickled object is a host-side snapshot with exactly that attribute graph (classes ``SavedWFunc`` / ``SavedMPSCoef`` /
``SavedSiteCoef`` of this module; time is not stored, like in the reference)."""

from __future__ import annotations

import os

import numpy as np

_GAUGE_NAMES = {0: "Psi", 1: "A", 2: "B", -1: "C"}


class SavedSiteCoef:
    def __init__(self, data, gauge, isite):
        self.data = np.ascontiguousarray(data, dtype=np.complex128)
        self.gauge = gauge
        self.isite = isite

    @property
    def shape(self):
        return self.data.shape


class SavedMPSCoef:
    def __init__(self, superblock_states, space):
        self.superblock_states = superblock_states
        self.nstate = len(superblock_states)
        self.nsite = len(superblock_states[0])
        self.space = space


class SavedWFunc:
    def __init__(self, ci_coef):
        self.ci_coef = ci_coef
        self.spf_coef = None  # MPS standard method: no single-particle functions


def snapshot(engine, space: str, nstate: int = 1) -> SavedWFunc:
    if nstate > 1:
        states = [[SavedSiteCoef(c, "C", i) for i, c in enumerate(cs)] for cs in engine.get_states()]
    else:
        gauges = [_GAUGE_NAMES[int(engine.get_site_shape(i)[3])] for i in range(engine.nsite)]
        states = [[SavedSiteCoef(c, g, i) for i, (c, g) in enumerate(zip(engine.get_mps(), gauges))]]
    return SavedWFunc(SavedMPSCoef(states, space))


def save(path: str, engine, space: str, nstate: int = 1) -> None:
    import dill

    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        dill.dump(snapshot(engine, space, nstate), f)
    os.replace(tmp, path)  # a crash while writing never leaves a truncated checkpoint behind


def load(path: str) -> SavedWFunc:
    import dill

    with open(path, "rb") as f:
        wf = dill.load(f)
    if not hasattr(wf, "ci_coef") or not hasattr(wf.ci_coef, "superblock_states"):
        raise ValueError(f"{path} is not a wavefunction checkpoint")
    return wf


def to_reference_checkpoint(path_in: str, path_out: str, model) -> None:
    """PyTDSCF side of the wire format (needs the reference package importable; nothing here runs on the GPU box).

    The reference's restart does ``wf = dill.load(f); wf = WFunc(wf.ci_coef, wf.spf_coef, ints_prim)``
    (simulator_cls.py:501-507): ``ci_coef`` has to be a genuine ``MPSCoefMPO`` with its methods, which only the reference
    can construct.  This converter does that on the reference's side: an ``MPSCoefMPO`` allocated for ``model`` (the
    reference ``Model`` of the run; ``alloc_random``, _mps_mpo.py:56-120) takes the saved tensors and gauge tags site by
    site, is wrapped in a ``WFunc`` and written with dill where ``Simulator.propagate(restart=True, loadfile_ext=...)``
    looks for it.  tests/golden/crosscheck_reference.py restarts the reference from such a file and compares the next
    step with the oracle's (development container)."""
    import dill
    from pytdscf._mps_mpo import MPSCoefMPO  # the reference
    from pytdscf._site_cls import SiteCoef
    from pytdscf._spf_cls import SPFCoef
    from pytdscf.wavefunction import WFunc

    saved = load(path_in)
    ci = MPSCoefMPO.alloc_random(model)
    if len(ci.superblock_states) != saved.ci_coef.nstate or len(ci.superblock_states[0]) != saved.ci_coef.nsite:
        raise ValueError("the model and the checkpoint disagree on the number of states / sites")
    for istate, sites in enumerate(saved.ci_coef.superblock_states):
        for isite, sc in enumerate(sites):
            ci.superblock_states[istate][isite] = SiteCoef(np.array(sc.data, dtype=np.complex128), sc.gauge, isite)
    ci.op_sys_sites = None  # environments are rebuilt from the tensors (_mps_cls.py:835-843)
    with open(path_out, "wb") as f:
        # the standard method carries trivial single-particle functions ("uniform (all 1.0)", simulator_cls.py:523-524)
        dill.dump(WFunc(ci, SPFCoef.alloc_eye(model), None), f)
