"""GPU: site-range sharding (pytdscf_amd/parallel_sites.py over csrc/shard.hip) with 2 and 3 ranks sharing the test
GPU (messages through the library's callback transport over gloo), against (a) the REFERENCE's MPSCoefParallel --
fixtures tests/golden/parallel_*.npz, 1e-8 -- (b) the oracle of the same algorithm at 1e-8 and (c) the serial sweep at
the looser bar SURVEY 8(e) sets for this approximate scheme (the reference accepts 1e-2 on the norm and 1e-1 on
energies; here infidelity < 1e-6, norm to 1e-4 at dt = 0.2).  Plus the library's RCCL point-to-point path on a
one-rank communicator and the CPU test of the neighbour link (world size 2, gloo)."""

import json
import os
import socket
import sys
import textwrap

import numpy as np
import pytest

from helpers.ranks import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


LINK_WORKER = """
import sys
sys.path.insert(0, {root!r})
import numpy as np
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import _Link, split_sites
c = Comm(n_devices=0)
link = _Link(c)
rng = np.random.default_rng(5)
a = rng.standard_normal((3, 4, 5)) + 1j * rng.standard_normal((3, 4, 5))
if c.rank == 0:
    link.send(a, 1)
    b = link.recv((5, 2), 1)
    assert np.array_equal(b, (a.reshape(-1)[:10] * 2).reshape(5, 2))
else:
    got = link.recv((3, 4, 5), 0)
    assert np.array_equal(got, a)
    link.send((got.reshape(-1)[:10] * 2).reshape(5, 2), 0)
assert link.messages == 1 and split_sites(7, 2) == [(0, 4), (4, 7)]
c.barrier()
print("LINK OK", flush=True)
c.close()
"""


def test_neighbour_link_two_ranks_gloo(tmp_path):
    script = tmp_path / "link.py"
    script.write_text(textwrap.dedent(LINK_WORKER.format(root=ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", CUDA_VISIBLE_DEVICES="",
               MITDVP_DIST_BACKEND="gloo")
    rcs, outs = run_ranks([[sys.executable, str(script)]] * 2, [dict(env, RANK=str(r), LOCAL_RANK=str(r)) for r in range(2)],
                          timeout=120)
    assert rcs == [0, 0] and all("LINK OK" in o for o in outs), outs


WORKER = """
import os, sys, json
os.environ["MITDVP_SMALL_KERNELS"] = "0"   # several processes share one GPU here: no persistent kernels
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from oracle import tdvp_parallel_oracle as par
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import SiteShardedTDVP
comm = Comm()
L, d, M, D, dt, nstep = {L}, {d}, 4, {D}, 0.2, 2
mpo = orc.synthetic_mpo(L, d, M, seed=0)
mps = orc.synthetic_mps([d] * L, D, seed=1)
eng = SiteShardedTDVP(comm, mpo, cores=mps, integrator={integ!r}, conserve_norm={cn}, junction={junction!r}, **{opts!r})
assert eng.selftest()
g0 = eng.gather()
for _ in range(nstep):
    eng.step(dt)
obs = dict(norm=eng.norm(), auto=eng.autocorr(), energy=eng.expectation())
op2 = orc.synthetic_mpo(L, d, 3, seed=7)
obs["op2"] = eng.expectation(op2)
rdms = {{p: eng.site_rdm(p) for p in (0, L // 2 - 1, L // 2, L - 1)}}
rdms_all = eng.site_rdms()
rdms_some = eng.site_rdms([1, L - 2])
rd_multi = {{k: eng.reduced_density(k) for k in [(1, 1, L - 2, L - 2), (0, L // 2, L // 2), (L - 1, L - 1)]}}
g = eng.gather()
def sandwich(bra, ket, ops=None):
    e = np.ones((1, 1, 1), complex)
    for p, (x, y) in enumerate(zip(bra, ket)):
        w = ops[p] if ops is not None else np.eye(x.shape[1]).reshape(1, x.shape[1], x.shape[1], 1)
        e = np.einsum("acb,aix,cijt,bjy->xty", e, x, w, y, optimize=True)
    return complex(e[0, 0, 0])
if comm.rank == 0:
    gc = [c.conj() for c in g]
    def rdm(cores, p):
        l = np.ones((1, 1), complex)
        for x in cores[:p]:
            l = np.einsum("ab,aic,bid->cd", l, x.conj(), x, optimize=True)   # [bra][ket]
        r = np.ones((1, 1), complex)
        for x in reversed(cores[p + 1:]):
            r = np.einsum("cd,aic,bid->ab", r, x.conj(), x, optimize=True)
        x = cores[p]
        return np.einsum("ab,bjs,ts,akt->jk", l, x, r, x.conj(), optimize=True)  # rho[j][j'] = ket j, bra j'
    rdm_gap = max(np.abs(rdms[p] - rdm(g, p)).max() for p in rdms)
    assert sorted(rdms_all) == list(range(L)) and sorted(rdms_some) == [1, L - 2]
    rdm_gap = max([rdm_gap] + [np.abs(rdms_all[p] - rdm(g, p)).max() for p in range(L)]
                  + [np.abs(rdms_some[p] - rdm(g, p)).max() for p in rdms_some])
    # multi-site keys (pairs across ranks, a diagonal leg): against the oracle's contraction of the gathered chain
    gcan = orc.canonicalize_site0([c.copy() for c in g], scale=None)   # the oracle's contraction wants the canonical form
    rdm_gap = max([rdm_gap] + [np.abs(rd_multi[k] - orc.reduced_density(gcan, [k.count(p) for p in range(L)])).max() for k in rd_multi])
    obs_gap = max(rdm_gap, abs(obs["norm"] - np.sqrt(sandwich(gc, g).real)), abs(obs["auto"] - sandwich(g, g)),
                  abs(obs["energy"] - sandwich(gc, g, mpo)), abs(obs["op2"] - sandwich(gc, g, op2)))
    ref = par.ParallelOracle([c.copy() for c in mps], mpo, comm.world, integrator={integ!r}, conserve_norm={cn}, **{opts!r})
    ser = orc.OracleMPS([c.copy() for c in mps], mpo, integrator={integ!r}, conserve_norm={cn})
    ser.build_right_envs()
    for _ in range(nstep):
        ref.step(dt)
        ser.propagate(dt)
    go = ref.gather()
    nrm = float(np.sqrt(abs(orc.overlap(g, g))))
    out = dict(init=abs(abs(orc.overlap(g0, mps)) - 1),
               vs_oracle=abs(abs(orc.overlap(go, g)) / (nrm * ref.norm()) - 1),
               norm_gap=abs(nrm - ref.norm()),
               vs_serial=abs(abs(orc.overlap(ser.cores, g)) / nrm - 1), norm=nrm, obs_gap=obs_gap,
               energy=obs["energy"].real, transport=eng.transport, collectives=(eng.joint.counters()["n_collectives"] if eng.joint is not None else 0),
               bytes=eng.traffic()[0], messages=eng.traffic()[1])
    print("RESULT " + json.dumps(out), flush=True)
comm.barrier()
eng.close()
comm.close()
"""


def _launch(script, world, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               MITDVP_DIST_BACKEND="gloo")
    rcs, outs = run_ranks([[sys.executable, str(script)]] * world, [dict(env, RANK=str(r), LOCAL_RANK="0") for r in range(world)],
                          timeout=timeout)
    assert rcs == [0] * world, "\n".join(outs)
    return json.loads([l for l in outs[0].splitlines() if l.startswith("RESULT ")][0][7:])


def _run(world, tmp_path, L=8, integ="lanczos", cn=True, d=3, D=8, junction="single", opts=None):
    script = tmp_path / f"ss{world}.py"
    script.write_text(textwrap.dedent(WORKER.format(root=ROOT, L=L, integ=integ, cn=cn, d=d, D=D, junction=junction, opts=opts or {})))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               MITDVP_DIST_BACKEND="gloo")
    rcs, outs = run_ranks([[sys.executable, str(script)]] * world, [dict(env, RANK=str(r), LOCAL_RANK="0") for r in range(world)],
                          timeout=300)
    assert rcs == [0] * world, "\n".join(outs)
    return json.loads([l for l in outs[0].splitlines() if l.startswith("RESULT ")][0][7:])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_site_sharded_matches_its_oracle_and_the_serial_sweep(world, tmp_path):
    r = _run(world, tmp_path, L=9 if world == 3 else 8)
    assert r["init"] < 1e-12                      # Phi_0 X_0^+ Phi_1 ... is the input state
    assert r["vs_oracle"] < 1e-8 and r["norm_gap"] < 1e-8
    assert r["vs_serial"] < (1e-12 if world == 1 else 1e-6)
    assert abs(r["norm"] - 1) < (1e-12 if world == 1 else 1e-4)  # O(dt^2) per junction (oracle: 1.2e-5 at N=3, dt=0.2)
    assert r["obs_gap"] < 1e-10                   # norm, <Psi*|Psi>, energy, a second operator: folded rank by rank
    if world > 1:
        assert r["messages"] > 0                  # neighbour traffic only: 5 messages per junction and half step (+ gather)


REF_WORKER = """
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import SiteShardedTDVP
comm = Comm()
g = np.load(os.path.join({root!r}, "tests", "golden", {name!r}))
n = 8
mpo = [g[f"mpo{{i}}"] for i in range(n)]
start = [g[f"start{{i}}"] for i in range(n)]
dt = float(g["dt_au"])
eng = SiteShardedTDVP(comm, mpo, cores=start, split=[tuple(int(x) for x in r) for r in g["split"]],
                      regularize=True, p_svd=float(g["p_svd"]), junction={junction!r})
assert eng.selftest() and eng.junction == {junction!r}
worst = dict(fid=0.0, norm=0.0, auto=0.0, sv=0.0, norm_folded=0.0)
for k in range(int(g["nstep"]) + 1):
    mine = eng.gather()
    n2f, acf = eng.overlap(True), eng.overlap(False)       # folded rank by rank on the devices
    sv = np.linalg.svd(eng.X, compute_uv=False) if comm.rank < comm.world - 1 else None
    box = [None] * comm.world
    comm.dist.all_gather_object(box, sv)
    if comm.rank == 0:
        ref = [g[f"step{{k}}_site{{i}}"] for i in range(n)]
        n2r, n2 = abs(orc.overlap(ref, ref)), abs(orc.overlap(mine, mine))
        worst["fid"] = max(worst["fid"], abs(abs(orc.overlap(ref, mine)) / np.sqrt(n2 * n2r) - 1))
        worst["norm"] = max(worst["norm"], abs(n2 - float(g["norm"][k])))
        worst["norm_folded"] = max(worst["norm_folded"], abs(n2f.real - float(g["norm"][k])))
        worst["auto"] = max(worst["auto"], abs(acf - complex(g["autocorr"][k])))
        for j in range(comm.world - 1):
            worst["sv"] = max(worst["sv"], np.abs(box[j] - np.linalg.svd(g[f"step{{k}}_joint{{j}}"], compute_uv=False)).max())
    if k < int(g["nstep"]):
        eng.step(dt)
kry = [eng.block.krylov_memory(i) for i in range(eng.n)]
box = [None] * comm.world
comm.dist.all_gather_object(box, kry)
if comm.rank == 0:
    worst["krylov_equal"] = all(box[r] == [int(x) for x in g["krylov"][r][: len(box[r])]] for r in range(comm.world))
    worst["norm_after_first_step"] = float(g["norm"][1])
    worst["transport"] = eng.transport
    print("RESULT " + json.dumps(worst), flush=True)
comm.barrier()
eng.close()
comm.close()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("junction", ["pair", "single"])
@pytest.mark.parametrize(
    "name, world", [("parallel_chain_r2.npz", 2), ("parallel_chain_r3.npz", 3), ("parallel_chain_graded.npz", 2)]
)
def test_site_sharded_reproduces_the_reference_parallel_tdvp(name, world, junction, tmp_path):
    """MPSCoefParallel.propagate (_mps_parallel.py:106-470) run by the REFERENCE on 2 / 3 ranks
    (tests/golden/make_golden_parallel.py): the state after every step, <Psi|Psi>, <Psi*|Psi>, the spectrum of every
    joint matrix and every rank's Krylov counts -- with the junction update on the left rank alone ("single", the
    reference's arrangement) and bond-sharded over both ranks of the junction ("pair", the default).  ``graded``: Schmidt values down to 1e-6 at the junction, p_svd = 1e-5 --
    the lifting of small singular values and the truncation of the joint matrix act (the reference's <Psi|Psi> falls to
    0.39 after one step); entries of X^+ reach 1e6 there, hence the looser bar."""
    script = tmp_path / "ref.py"
    script.write_text(textwrap.dedent(REF_WORKER.format(root=ROOT, name=name, junction=junction)))
    r = _launch(script, world)
    tol = 1e-6 if "graded" in name else 1e-8
    assert r["transport"] == "callback"            # ranks share the GPU: gloo carries the library's messages
    assert r["fid"] < tol and r["norm"] < tol and r["norm_folded"] < tol and r["auto"] < tol and r["sv"] < tol, r
    assert r["krylov_equal"], r
    if "graded" in name:
        assert r["norm_after_first_step"] < 0.5


EXC_WORKER = """
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import mps as M, operators as O
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import SiteShardedTDVP
comm = Comm()
g = np.load(os.path.join({root!r}, "tests", "golden", "parallel_exciton.npz"))
pot = [g[f"pot{{i}}"] for i in range(4)]
kin = [g[f"kin{{i}}"] for i in range(3)]
mpo = O.merge_operator_terms([(pot, [0, 1, 2, 3]), (kin, [0, 1, 2])], dims=[8, 8, 8, 2])
start = orc.canonicalize_site0(M.product_state_cores([g[f"weight{{i}}"] for i in range(4)], bond_dim=int(g["bond_dim"])))
eng = SiteShardedTDVP(comm, mpo, cores=start, split=[(0, 1), (2, 3)], regularize=True, p_svd=float(g["p_svd"]))
dt = float(g["dt_au"])
out = dict(energy=[], norm2=[], infid=[])
for k in range(21):
    if k in (0, 1, 2, 5, 10, 19, 20):
        e, n2 = eng.expectation().real, eng.overlap(True).real
        mine = eng.gather()
        if comm.rank == 0:
            out["energy"].append(e / n2)
            out["norm2"].append(n2)
            if k != 19:
                ref = [g[f"step{{k}}_site{{i}}"] for i in range(4)]
                out["infid"].append(1 - abs(orc.overlap(ref, mine)) / np.sqrt(abs(orc.overlap(ref, ref)) * abs(orc.overlap(mine, mine))))
    eng.step(dt)
if comm.rank == 0:
    out["energy_ref"] = [float(g["energy_ref"][k].real) for k in (0, 1, 2, 5, 10, 19, 20)]
    print("RESULT " + json.dumps(out), flush=True)
comm.barrier()
eng.close()
comm.close()
"""


@pytest.mark.gpu
def test_reference_mpi_exciton_model_two_ranks(tmp_path):
    """The reference's own MPI test (tests/test_mpi_exiciton_propagate.py: 4 sites, parallel_split_indices
    [(0, 1), (2, 3)], zero-padded product start, 20 steps of 0.05 fs): its pin -- energy 0.01000 to rel 1e-1 (:220) --
    and the serial run's pin 0.010000180312707298 (tests/test_exiciton_propagate.py:178).  From a rank-1 junction the
    lifted null directions are whatever each SVD's completion is (the reference's authors: "not always reproducible";
    its own <Psi|Psi> drifts to 1.03 here), so beyond the pins: energy within 2e-2 of the reference's estimator at the
    same step (a fifth of its own bar), <Psi|Psi> within 0.1 of 1, state within 5e-3 of the reference's in fidelity."""
    script = tmp_path / "exc.py"
    script.write_text(textwrap.dedent(EXC_WORKER.format(root=ROOT)))
    r = _launch(script, 2)
    print(json.dumps(r))
    e19 = r["energy"][5]
    assert e19 == pytest.approx(0.01000, rel=1e-1)                       # the reference-held pin
    assert e19 == pytest.approx(0.010000180312707298, rel=2e-2)          # the serial pin, at the scheme's accuracy
    for e, er in zip(r["energy"], r["energy_ref"]):
        assert e == pytest.approx(er, rel=2e-2)
    assert r["infid"][0] < 1e-12 and max(r["infid"]) < 5e-3, r
    assert max(abs(x - 1) for x in r["norm2"]) < 0.1, r


SHELL_WORKER = """
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
import pytest
from pytdscf_amd import Exciton, HarmonicOscillator as HO, Model, Simulator, TensorHamiltonian, TensorOperator
rank = int(os.environ["RANK"])
os.chdir({cwd!r})
g = np.load(os.path.join({root!r}, "tests", "golden", "parallel_exciton.npz"))
prim_info = [HO(8, f, units="cm-1") for f in (1000, 2000, 3000)] + [Exciton(nstate=2, names=["S0", "S1"])]
if rank == 0:  # like the reference's test: the operators exist on rank 0 only
    pot = [g[f"pot{{i}}"] for i in range(4)]
    kin = [g[f"kin{{i}}"] for i in range(3)]
    potential = [[{{(0, 1, 2, (3, 3)): TensorOperator(mpo=pot, legs=(0, 1, 2, 3, 3))}}]]
    kinetic = [[{{((0, 0), (1, 1), (2, 2)): TensorOperator(mpo=kin, legs=(0, 0, 1, 1, 2, 2))}}]]
else:
    potential = kinetic = None
hamiltonian = TensorHamiltonian(ndof=4, potential=potential, kinetic=kinetic, backend="hip")
model = Model(prim_info, {{"hamiltonian": hamiltonian}})
model.m_aux_max = 10
model.init_HartreeProduct = [[ho.get_unitary()[0].tolist() for ho in prim_info[:3]] + [np.array([0.0, 1.0]).tolist()]]
simulator = Simulator("mpi_LVC_Exciton_test_10m", model, backend="hip")
ener_calc, wf = simulator.propagate(stepsize=0.05, maxstep=20, reduced_density=([(3, 3)], 1),
                                    parallel_split_indices=[(0, 1), (2, 3)], adaptive=False, adaptive_p_svd=1e-06)
if rank == 0:
    assert pytest.approx(ener_calc, rel=1.0e-01) == 0.01000          # the reference's assertion, verbatim
    t, rdm = simulator.rdm_trace[-1]
    print("RESULT " + json.dumps(dict(energy=ener_calc, norm=wf.norm(), tr=float(np.trace(rdm[(3, 3)]).real),
                                      herm=float(abs(rdm[(3, 3)] - rdm[(3, 3)].conj().T).max()))), flush=True)
else:
    assert wf is None
from pytdscf_amd.dist import world_comm
world_comm().close()
"""


@pytest.mark.gpu
def test_reference_mpi_test_script_through_the_shell(tmp_path):
    """tests/test_mpi_exiciton_propagate.py of the reference as a user would run it here: ``Simulator.propagate(
    parallel_split_indices=[(0, 1), (2, 3)])`` under two ranks (operators on rank 0 only, broadcast like
    ``distribute_mpo_cores``), its own assertion on the returned energy, and the files rank 0 writes."""
    from pytdscf_amd.util import read_nc

    script = tmp_path / "shell.py"
    script.write_text(textwrap.dedent(SHELL_WORKER.format(root=ROOT, cwd=str(tmp_path))))
    r = _launch(script, 2)
    g = np.load(os.path.join(ROOT, "tests", "golden", "parallel_exciton.npz"))
    # the reference's run at the same step; <H> is not divided by <Psi|Psi> (drifts by a few per cent here, see above)
    assert r["energy"] == pytest.approx(float(g["energy_ref"][19].real), rel=6e-2)
    # wf is the state after the 20th step, the last density the one before it
    assert abs(r["norm"] - 1) < 0.05 and abs(r["tr"] - 1) < 0.1 and r["herm"] < 1e-12
    out = tmp_path / "mpi_LVC_Exciton_test_10m_prop"
    lines = open(out / "expectations.dat").read().splitlines()
    assert lines[0].startswith("# time [fs]") and len(lines) == 21
    assert len(open(out / "autocorr.dat").read().splitlines()) == 21
    data = read_nc(str(out / "reduced_density.nc"), [(3, 3)])
    assert data[(3, 3)].shape == (20, 2, 2)
    assert (tmp_path / "wf_mpi_LVC_Exciton_test_10m.pkl").exists()


@pytest.mark.gpu
def test_library_rccl_point_to_point_on_a_one_rank_communicator():
    """ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd resolved from librccl by the library itself (csrc/rccl_dyn.h)
    and used on the engine's stream: a grouped send + receive from the rank to itself -- what the one-GPU box can
    exercise of the halo transport that runs between GPUs at world size > 1."""
    import ctypes as C

    from pytdscf_amd import _lib

    lib = _lib.load()
    cfg = _lib.Config()
    cfg.nsite, cfg.device, cfg.integrator, cfg.conserve_norm, cfg.thresh, cfg.max_krylov = 2, 0, 0, 1, 1e-9, 20
    h = C.c_void_p()
    _lib.check(lib.mitdvp_shard_create(C.byref(cfg), 0, 1, 2, 0, C.byref(h)), None, shard=True)
    try:
        ident = C.create_string_buffer(128)
        _lib.check(lib.mitdvp_rccl_unique_id(ident))
        _lib.check(lib.mitdvp_shard_attach_rccl(h, ident.raw), h, shard=True)
        bad = C.c_int(-1)
        for elems in (7, 1 << 16, (1 << 20) + 3):
            _lib.check(lib.mitdvp_shard_self_sendrecv(h, elems, C.byref(bad)), h, shard=True)
            assert bad.value == 0
    finally:
        lib.mitdvp_shard_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_junction_update_bond_sharded_over_the_pair(world, tmp_path):
    """Pair mode at a bond dimension the engine's tensor parallelism shards (D = 16: rows 0-7 on the left rank of a
    junction, 8-15 on the right one): all-gathers / all-reduces of the two-site engine over the point-to-point transport
    must actually run, the state must equal the oracle's and the left-rank-only mode's."""
    L = 9 if world == 3 else 8
    r = _run(world, tmp_path, L=L, d=4, D=16, junction="pair")
    s = _run(world, tmp_path, L=L, d=4, D=16, junction="single")
    assert r["collectives"] > 20 and s["collectives"] == 0   # rank 0's junction engine: sharded applies in pair mode only
    assert r["init"] < 1e-12 and r["vs_oracle"] < 1e-8 and r["norm_gap"] < 1e-8 and r["obs_gap"] < 1e-10
    assert abs(r["energy"] - s["energy"]) < 1e-12 and abs(r["norm"] - s["norm"]) < 1e-12
    if world == 2:  # the reference's regularisation + truncation of the joint matrix inside the sharded junction engine
        opts = dict(regularize=True, p_svd=1e-8)
        r = _run(world, tmp_path, L=L, d=4, D=16, junction="pair", opts=opts)
        s = _run(world, tmp_path, L=L, d=4, D=16, junction="single", opts=opts)
        assert r["vs_oracle"] < 1e-8 and r["norm_gap"] < 1e-8
        assert abs(r["energy"] - s["energy"]) < 1e-12 and abs(r["norm"] - s["norm"]) < 1e-12


SETUP_WORKER = """
import os, sys, json
os.environ["MITDVP_SMALL_KERNELS"] = "0"
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd.dist import Comm
from pytdscf_amd.parallel_sites import SiteShardedTDVP
comm = Comm()
L, d, M, D, dt = {L}, 4, 4, 16, 0.2
mpo = orc.synthetic_mpo(L, d, M, seed=0)
out = {{}}
for mode in ("replicated", "pipeline"):
    os.environ["MITDVP_SHARD_SETUP"] = mode
    eng = SiteShardedTDVP(comm, mpo, dims=[d] * L, bond_dim=D, seed=3)
    assert eng.setup_mode == mode and eng.selftest()
    g0 = eng.gather()
    n0, e0 = eng.norm(), eng.expectation()
    x0 = eng.X if comm.rank < comm.world - 1 else None
    eng.step(dt)
    out[mode] = (g0, n0, e0, x0, eng.gather(), eng.expectation())
    eng.close()
a, b = out["replicated"], out["pipeline"]
if a[3] is not None:   # the junction matrix this rank holds: the same up to the gauge of the QR factors (singular values)
    sa, sb = np.linalg.svd(a[3], compute_uv=False), np.linalg.svd(b[3], compute_uv=False)
    assert np.abs(sa - sb).max() < 1e-12, (sa, sb)
if comm.rank == 0:
    ov = lambda x, y: abs(orc.overlap(x, y)) / np.sqrt(abs(orc.overlap(x, x)) * abs(orc.overlap(y, y)))
    print("RESULT " + json.dumps(dict(start=abs(ov(a[0], b[0]) - 1), norm0=abs(a[1] - b[1]), norm=abs(b[1] - 1), e0=abs(a[2] - b[2]),
                                      after=abs(ov(a[4], b[4]) - 1), e1=abs(a[5] - b[5]))), flush=True)
comm.barrier()
comm.close()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_pipelined_setup_gives_the_state_of_the_replicated_one(world, tmp_path):
    """Random start (what bench.py's site-sharded leg uses): every rank draws and canonicalises only its own block and the
    ranks hand weight matrix and boundary blocks on (MITDVP_SHARD_SETUP=pipeline, the default), against the set-up of
    rounds 2-4 in which every rank walked a copy of the whole chain: same state, norm, energy and junction spectra at the
    start, same state after a time step."""
    script = tmp_path / f"setup{world}.py"
    script.write_text(textwrap.dedent(SETUP_WORKER.format(root=ROOT, L=9 if world == 3 else 8)))
    r = _launch(script, world)
    assert r["start"] < 1e-12 and r["norm0"] < 1e-12 and r["norm"] < 1e-12 and r["e0"] < 1e-11, r
    assert r["after"] < 1e-10 and r["e1"] < 1e-10, r


@pytest.mark.gpu
def test_site_sharded_arnoldi_without_renormalisation(tmp_path):
    r = _run(2, tmp_path, integ="arnoldi", cn=False)
    assert r["vs_oracle"] < 1e-8 and r["vs_serial"] < 1e-6
    p = _run(2, tmp_path, integ="arnoldi", cn=False, d=4, D=16, junction="pair")  # sharded Hessenberg solves in the junction engine
    assert p["collectives"] > 20 and p["vs_oracle"] < 1e-8 and p["vs_serial"] < 1e-6


@pytest.mark.gpu
def test_pseudo_inverse_on_the_device_matches_numpy():
    """multiply_sigvec_pinv's pseudo-inverse (_site_cls.py:709-754, RCOND 1e-13): device SVD + device product."""
    from pytdscf_amd.parallel_sites import RCOND, pinv_device

    rng = np.random.default_rng(2)
    x = rng.standard_normal((24, 24)) + 1j * rng.standard_normal((24, 24))
    u, s, vh = np.linalg.svd(x)
    s[-5:] = 0.0                      # rank-deficient: the cut-off branch
    s[:19] = np.logspace(0, -9, 19)   # graded
    x = (u * s) @ vh
    got = pinv_device(x)
    inv = np.where(s > RCOND * s.max(), 1.0 / np.where(s > 0, s, 1.0), 0.0)
    ref = (vh.conj().T * inv) @ u.conj().T          # the pseudo-inverse by construction
    # entries up to 1e9; a relative error eps * s_max / s_min = 1e-7 in the smallest kept singular value is what
    # double precision allows (np.linalg.pinv itself is no closer to the constructed inverse)
    assert np.abs(got - ref).max() < 1e-5 * np.abs(ref).max()
    assert np.abs(x @ got @ x - x).max() < 1e-6     # Moore-Penrose: x x^+ x = x (rounding: eps |x| |x^+| |x| ~ 1e-7)
