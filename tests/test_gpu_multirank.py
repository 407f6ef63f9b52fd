"""GPU: bond-sharded (tensor-parallel) execution, 2 and 3 ranks sharing the one
GPU of the test box over gloo (host-staged collectives), against the 1-rank
engine; plus the nccl device-memory path at world_size 1."""

import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from helpers.ranks import run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = """
import sys, json
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
from pytdscf_amd.dist import Comm, attach_parallel
comm = Comm()
L, d, M, D = 8, 4, 5, {D}           # 48 = 2 * 24 = 3 * 16: shards for 2 and 3 ranks
mpo = orc.synthetic_mpo(L, d, M, seed=3)
mps = orc.synthetic_mps([d] * L, D, seed=4)
eng = TDVPEngine(L, device=0, integrator={integ!r}, conserve_norm={cn})
eng.set_mpo(mpo)
eng.set_mps(mps)
attach_parallel(eng, comm)
if {adaptive}:
    eng.set_adaptive(True, Dmax=32, dD=8, p_proj=1e-9)
e0 = eng.expectation()
for _ in range(2):
    eng.propagate(0.4)
out = dict(rank=comm.rank, e0=[e0.real, e0.imag], e=[eng.expectation().real, eng.expectation().imag],
           norm=eng.norm(), ac=[eng.autocorr().real, eng.autocorr().imag], k=eng.krylov_stats(),
           ncoll=eng.counters()["n_collectives"], bd=eng.bond_dims())
np.save({out!r} + f".rank{{comm.rank}}.npy", np.concatenate([c.reshape(-1) for c in eng.get_mps()]))
print("RESULT " + json.dumps(out), flush=True)
comm.barrier()
comm.close()
"""


def _run(world, tmp_path, integ="lanczos", cn=True, backend_env=None, D=48, adaptive=False):
    import json

    script = tmp_path / f"w{world}.py"
    out = str(tmp_path / f"mps_w{world}")
    script.write_text(textwrap.dedent(WORKER.format(root=ROOT, integ=integ, cn=cn, out=out, D=D, adaptive=adaptive)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
               MITDVP_DIST_BACKEND="gloo")
    if backend_env:
        env.update(backend_env)
    rcs, outs = run_ranks([[sys.executable, str(script)]] * world, [dict(env, RANK=str(r), LOCAL_RANK="0") for r in range(world)],
                          timeout=300)
    assert rcs == [0] * world, "\n".join(outs)
    res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs]
    vecs = [np.load(out + f".rank{r}.npy") for r in range(world)]
    return res, vecs


@pytest.mark.parametrize("world", [2, 3])
def test_bond_sharded_matches_single_rank(world, tmp_path):
    ref, vref = _run(1, tmp_path)
    res, vecs = _run(world, tmp_path)
    assert ref[0]["ncoll"] == 0 and all(r["ncoll"] > 0 for r in res)
    for r, v in zip(res, vecs):
        assert r["k"] == ref[0]["k"]                       # identical control flow on every rank
        assert abs(r["norm"] - 1) < 1e-12
        assert abs(complex(*r["e"]) - complex(*ref[0]["e"])) < 1e-10 * abs(complex(*ref[0]["e"]))
        assert abs(complex(*r["ac"]) - complex(*ref[0]["ac"])) < 1e-10
        assert np.abs(v - vref[0]).max() < 1e-10           # same tensors up to summation order
    for v in vecs[1:]:
        assert np.array_equal(v, vecs[0])                  # replicated state stays bit-identical across ranks


def test_bond_sharded_arnoldi(tmp_path):
    ref, vref = _run(1, tmp_path, integ="arnoldi", cn=False)
    res, vecs = _run(2, tmp_path, integ="arnoldi", cn=False)
    assert res[0]["k"] == ref[0]["k"]
    assert np.abs(vecs[0] - vref[0]).max() < 1e-10 and np.array_equal(vecs[0], vecs[1])


def test_bond_sharded_adaptive(tmp_path):
    """Adaptive bond dimension under bond sharding: the rank decisions are taken from
    replicated data, so every rank grows the same bonds; blocks whose leading dimension
    is not divisible by the world size stay replicated."""
    ref, vref = _run(1, tmp_path, D=16, adaptive=True)
    res, vecs = _run(2, tmp_path, D=16, adaptive=True)
    assert max(ref[0]["bd"]) > 16 and all(r["ncoll"] > 0 for r in res)
    for r, v in zip(res, vecs):
        assert r["bd"] == ref[0]["bd"] and r["k"] == ref[0]["k"]
        assert abs(r["norm"] - 1) < 1e-12
        assert abs(complex(*r["ac"]) - complex(*ref[0]["ac"])) < 1e-9
        assert v.shape == vref[0].shape and np.abs(v - vref[0]).max() < 1e-8
    assert np.array_equal(vecs[0], vecs[1])


def test_nccl_device_collectives_world1(tmp_path):
    """The production path (RCCL on the engine's device buffers through the CUDA
    array interface) at world_size 1: all-gather / all-reduce are identities."""
    script = tmp_path / "n.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import numpy as np, torch, torch.distributed as dist
        from pytdscf_amd.dist import _DevPtr
        from pytdscf_amd import engine as E
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        # engine-owned device memory, viewed zero-copy by torch
        x = torch.arange(64, dtype=torch.float64, device="cuda")
        t = torch.as_tensor(_DevPtr(x.data_ptr(), 64), device="cuda:0")
        dist.all_reduce(t)
        dist.all_gather_into_tensor(t, t.clone())
        torch.cuda.synchronize()
        assert torch.equal(x.cpu(), torch.arange(64, dtype=torch.float64))
        t += 1.0
        assert float(x[3]) == 4.0  # zero copy
        dist.destroy_process_group()
        print("OK")
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr


def test_native_rccl_collectives_world1(tmp_path):
    """The library's own RCCL path (dlopen'ed librccl, communicator owned by the engine,
    collectives on the engine's stream) at world size 1: unique id, communicator, all-gather and
    all-reduce self-test.  More ranks need more GPUs than the test box has."""
    script = tmp_path / "r.py"
    script.write_text(textwrap.dedent(f"""
        import sys, ctypes as C
        sys.path.insert(0, {ROOT!r})
        from oracle import tdvp_oracle as orc
        from pytdscf_amd import TDVPEngine, _lib
        lib = _lib.load()
        ident = C.create_string_buffer(128)
        _lib.check(lib.mitdvp_rccl_unique_id(ident))
        assert any(ident.raw)
        eng = TDVPEngine(4)
        eng.set_mpo(orc.synthetic_mpo(4, 3, 3, seed=1))
        eng.set_mps(orc.synthetic_mps([3] * 4, 4), canonicalize=True)
        _lib.check(lib.mitdvp_set_parallel_rccl(eng._h, 1, 0, ident.raw), eng._h)
        bad = C.c_int(-1)
        _lib.check(lib.mitdvp_rccl_selftest(eng._h, C.byref(bad)), eng._h)
        assert bad.value == 0, bad.value
        eng.propagate(0.1)
        assert abs(eng.norm() - 1) < 1e-12 and eng.counters()["n_collectives"] == 2
        eng.close()
        print("OK")
    """))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr


WORKER_MS = """
import sys, json
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import MultiStateEngine
from pytdscf_amd.dist import Comm, attach_parallel
comm = Comm()
L, d, M, S = 6, 4, 4, 2
rng = np.random.default_rng(7)
crandn = lambda *s: rng.standard_normal(s) + 1j * rng.standard_normal(s)
Ds = [32, 16]                      # both shard over 2 ranks in the middle of the chain
raw = [[crandn(a, d, b) for a, b in orc.bond_dims([d] * L, D)] for D in Ds]
mpo = [[orc.synthetic_mpo(L, d, M, seed=1), None], [None, orc.synthetic_mpo(L, d, M, seed=2)]]
w = [0.1 * crandn(a, d, d, b) for a, b in zip([1] + [2] * (L - 1), [2] * (L - 1) + [1])]
mpo[0][1] = w
mpo[1][0] = [np.ascontiguousarray(np.conj(c.transpose(0, 2, 1, 3))) for c in w]
eng = MultiStateEngine(L, S, device=0)
eng.set_hamiltonian(mpo, [[0.0, 0.05j], [-0.05j, 0.1]])
eng.set_states(raw, weights=[0.7, 0.3])
attach_parallel(eng, comm)
e0 = eng.expectation()
for _ in range(2):
    eng.propagate(0.3)
out = dict(rank=comm.rank, e0=[e0.real, e0.imag], e=[eng.expectation().real, eng.expectation().imag],
           norm=eng.norm(), pops=eng.pop_states(), k=eng.krylov_stats(), ncoll=eng.counters()["n_collectives"])
np.save({out!r} + f".rank{{comm.rank}}.npy", np.concatenate([c.reshape(-1) for st in eng.get_states() for c in st]))
print("RESULT " + json.dumps(out), flush=True)
comm.barrier()
comm.close()
"""


def test_multistate_bond_sharded_matches_single_rank(tmp_path):
    """Several electronic states under bond sharding: the per-pair applies / environment updates
    shard over the bra-side bond like the single-state ones (2 ranks sharing the GPU over gloo)."""
    import json

    def run(world):
        script = tmp_path / f"ms{world}.py"
        out = str(tmp_path / f"ms_w{world}")
        script.write_text(textwrap.dedent(WORKER_MS.format(root=ROOT, out=out)))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
                   MITDVP_DIST_BACKEND="gloo")
        rcs, outs = run_ranks([[sys.executable, str(script)]] * world,
                              [dict(env, RANK=str(r), LOCAL_RANK="0") for r in range(world)], timeout=300)
        assert rcs == [0] * world, "\n".join(outs)
        res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs]
        return res, [np.load(out + f".rank{r}.npy") for r in range(world)]

    ref, vref = run(1)
    res, vecs = run(2)
    assert ref[0]["ncoll"] == 0 and all(r["ncoll"] > 0 for r in res)
    for r, v in zip(res, vecs):
        assert r["k"] == ref[0]["k"]
        np.testing.assert_allclose(r["e"], ref[0]["e"], atol=1e-10)
        np.testing.assert_allclose(r["pops"], ref[0]["pops"], atol=1e-10)
        assert abs(r["norm"] - 1) < 1e-12
        np.testing.assert_allclose(v, vref[0], atol=1e-9)
    assert np.array_equal(vecs[0], vecs[1])  # the replicated state is bit-identical across ranks


WORKER_OPS = """
import sys, json
sys.path.insert(0, {root!r})
import numpy as np
from oracle import tdvp_oracle as orc
from pytdscf_amd import TDVPEngine
from pytdscf_amd.dist import Comm, attach_parallel
comm = Comm()
L, d, M, D = 8, 4, 5, 32
mpo = orc.synthetic_mpo(L, d, M, seed=3)
dip = orc.synthetic_mpo(L, d, 3, seed=9)
mps = orc.synthetic_mps([d] * L, D, seed=4)
rng = np.random.default_rng(2)
q, _ = np.linalg.qr(rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d)))
eng = TDVPEngine(L, device=0)
eng.set_mpo(mpo, shift=0.2)
eng.set_mpo(dip, op_id=1, shift=0.1)
eng.set_mps(mps)
attach_parallel(eng, comm)
eng.set_gates({{2: q, 5: np.exp(1j * np.arange(d))}})          # applied between the half-sweeps
eng.propagate(0.3)
nrm, it = eng.operate(1, maxstep=3)                          # mixed bra / ket blocks, shift through overlap chains
eng.propagate(0.3)
out = dict(rank=comm.rank, nrm=nrm, it=it, e=[eng.expectation().real, eng.expectation().imag], norm=eng.norm(),
           k=eng.krylov_stats(), ncoll=eng.counters()["n_collectives"])
np.save({out!r} + f".rank{{comm.rank}}.npy", np.concatenate([c.reshape(-1) for c in eng.get_mps()]))
print("RESULT " + json.dumps(out), flush=True)
comm.barrier()
comm.close()
"""


def test_bond_sharded_gates_and_operate(tmp_path):
    """One-site gates between the half-sweeps and Simulator.operate (mixed bra/ket blocks and the
    scalar term's overlap chain) under bond sharding."""
    import json

    def run(world):
        script = tmp_path / f"ops{world}.py"
        out = str(tmp_path / f"ops_w{world}")
        script.write_text(textwrap.dedent(WORKER_OPS.format(root=ROOT, out=out)))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world),
                   MITDVP_DIST_BACKEND="gloo")
        rcs, outs = run_ranks([[sys.executable, str(script)]] * world,
                              [dict(env, RANK=str(r), LOCAL_RANK="0") for r in range(world)], timeout=300)
        assert rcs == [0] * world, "\n".join(outs)
        res = [json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][0][7:]) for o in outs]
        return res, [np.load(out + f".rank{r}.npy") for r in range(world)]

    ref, vref = run(1)
    res, vecs = run(2)
    assert all(r["ncoll"] > 0 for r in res)
    for r, v in zip(res, vecs):
        assert r["k"] == ref[0]["k"] and r["it"] == ref[0]["it"]
        assert abs(r["nrm"] - ref[0]["nrm"]) < 1e-10 * ref[0]["nrm"]
        np.testing.assert_allclose(r["e"], ref[0]["e"], atol=1e-10)
        assert v.shape == vref[0].shape and np.abs(v - vref[0]).max() < 1e-8
    assert np.array_equal(vecs[0], vecs[1])
