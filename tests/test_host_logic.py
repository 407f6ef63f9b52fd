"""CPU: host-side logic of the shell (operator reduction, initial states) and
the C-ABI library's load/export contract.  No compute calls (no GPU here)."""

import numpy as np
import pytest

from oracle import tdvp_oracle as orc
from pytdscf_amd import mps as M
from pytdscf_amd import operators as O


def test_library_exports_every_declared_symbol():
    from pytdscf_amd import _lib

    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mitdvp_version().startswith(b"mitdvp")


def test_engine_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pytdscf_amd import TDVPEngine
    from pytdscf_amd._lib import MitdvpError

    with pytest.raises(MitdvpError):
        TDVPEngine(4)


def test_site_shard_fails_loudly_without_gpu_and_checks_its_arguments():
    """The native site shard (mitdvp_shard_*, csrc/shard.hip) has no CPU fallback either; bad rank / block arguments
    are refused before anything touches a device."""
    import ctypes as C

    import torch

    from pytdscf_amd import _lib

    lib = _lib.load()
    cfg = _lib.Config()
    cfg.nsite, cfg.device, cfg.integrator, cfg.conserve_norm, cfg.thresh, cfg.max_krylov = 4, 0, 0, 1, 1e-9, 20
    h = C.c_void_p()
    for rank, world, n, dr in ((2, 2, 4, 8), (0, 2, 1, 8), (0, 2, 4, 0)):  # rank out of range; one-site block; dr_next missing
        assert lib.mitdvp_shard_create(C.byref(cfg), rank, world, n, dr, C.byref(h)) == _lib.EINVAL
        assert not h.value and lib.mitdvp_shard_last_error(None)
    assert lib.mitdvp_shard_step(None, 0.1) == _lib.EINVAL  # null handle
    if not torch.cuda.is_available():
        assert lib.mitdvp_shard_create(C.byref(cfg), 0, 2, 4, 8, C.byref(h)) == _lib.EHIP
        assert not h.value


def test_merge_terms_is_exact_sum():
    rng = np.random.default_rng(0)
    dims = [3, 2, 4, 2]

    def term(sites, bonds, diag=False):
        cores = []
        for k, s in enumerate(sites):
            shp = (bonds[k], dims[s], bonds[k + 1]) if diag else (bonds[k], dims[s], dims[s], bonds[k + 1])
            cores.append(rng.standard_normal(shp) + 1j * rng.standard_normal(shp))
        return cores, sites

    t1 = term([0, 1, 2, 3], [1, 3, 2, 3, 1], diag=True)
    t2 = term([0, 1, 2], [1, 2, 2, 1])
    t3 = term([1, 3], [1, 2, 1])  # gap at site 2 carries the bond
    merged = O.merge_operator_terms([t1, t2, t3], dims)
    assert merged[0].shape[0] == 1 and merged[-1].shape[3] == 1
    dense = O.mpo_to_dense(merged)
    ref = sum(O.mpo_to_dense(O.full_chain(c, s, dims)) for c, s in (t1, t2, t3))
    np.testing.assert_allclose(dense, ref, atol=1e-13)


def test_product_state_and_bond_dims():
    assert M.bond_dims([3] * 5, 6) == orc.bond_dims([3] * 5, 6)
    cores = M.product_state_cores([[1, 1, 0], [0, 2, 0], [1, 0, 0]], bond_dim=4)
    assert [c.shape for c in cores] == [(1, 3, 3), (3, 3, 3), (3, 3, 1)]
    assert abs(np.linalg.norm(cores[0]) - 1) < 1e-15
    assert np.count_nonzero(cores[1]) == 1


def test_exciton_pin_through_merged_mpo(golden):
    """Reference regression pin (tests/test_exiciton_propagate.py:174-184) reproduced
    by the oracle on the direct-sum MPO: validates the reduction of the operator
    dictionary (diag cores + partial-span kinetic term) on the CPU."""
    g = golden("exciton.npz")
    pot = [g[f"pot{i}"] for i in range(4)]
    kin = [g[f"kin{i}"] for i in range(3)]
    mpo = O.merge_operator_terms([(pot, [0, 1, 2, 3]), (kin, [0, 1, 2])], dims=[8, 8, 8, 2])
    w = [g[f"w{i}"] for i in range(3)] + [np.array([0.0, 1.0])]
    init = M.product_state_cores(w, bond_dim=2)
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    dt = float(g["dt_au"])
    e = None
    for step in range(20):
        if step == 19:
            rdm = orc.site_rdm(st.cores, 3)
        e = st.expectation()
        st.propagate(dt)
    assert e.real == pytest.approx(float(g["ref_pin_energy"]))
    np.testing.assert_allclose(rdm, g["ref_pin_rdm33"], atol=1e-9)
    np.testing.assert_allclose(rdm, g["n19_rdm33"], atol=1e-10)
    # general reduced densities (keys of the reference test + diagonal-only forms)
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    for _ in range(19):
        st.propagate(dt)
    for tag, legs in (("rdm33", (0, 0, 0, 2)), ("rdm00", (2,)), ("rdm0033", (2, 0, 0, 2)), ("rdm1", (0, 1)), ("rdm013", (1, 2, 0, 1))):
        np.testing.assert_allclose(orc.reduced_density(st.cores, legs), g[f"n19_{tag}"], atol=1e-10)


def test_henon_heiles_pin_through_merged_mpo(golden):
    """tests/test_henon_heiles.py NumPy case (energy 0.018225341011652626)."""
    g = golden("henon_heiles.npz")
    pot = [g["pot0"], g["pot1"]]
    kin = [g["kin0"], g["kin1"]]
    mpo = O.merge_operator_terms([(pot, [0, 1]), (kin, [0, 1])], dims=[5, 5])
    # init_weight_VIBSTATE: HO-eigenbasis weights rotated to the DVR grid with
    # basis.get_unitary() (_mps_mpo.py:96-110): einsum("abc,bd->adc", core, U)
    wts = [np.array([0.0, 1.0, 0, 0, 0]), np.array([1.0, 0, 0, 0, 0])]
    init = M.product_state_cores(wts, bond_dim=4)
    init = [np.einsum("abc,bd->adc", c, g[f"unitary{i}"]) for i, c in enumerate(init)]
    st = orc.OracleMPS(orc.canonicalize_site0(init), mpo)
    dt = float(g["dt_au"])
    e = None
    for _ in range(3):
        e = st.expectation()
        st.propagate(dt)
    assert e.real == pytest.approx(float(g["ref_pin_energy"]))
    ref = [g["n3_final0"], g["n3_final1"]]
    fid = abs(orc.overlap(ref, st.cores))
    assert abs(fid - 1) < 1e-9


def test_product_side_synthetic_inputs_equal_the_oracles():
    """bench.py / tools build their inputs with pytdscf_amd.synthetic (so that oracle/ is only
    imported by the CPU-baseline leg); the two sets of builders must stay identical."""
    from pytdscf_amd import synthetic as syn

    for L, d, M in ((6, 3, 4), (3, 2, 3), (1, 2, 3)):
        assert all(np.array_equal(a, b) for a, b in zip(orc.synthetic_mpo(L, d, M, seed=5), syn.synthetic_mpo(L, d, M, seed=5)))
    assert all(np.array_equal(a, b) for a, b in zip(orc.synthetic_liouvillian_mpo(5, 8, seed=2), syn.synthetic_liouvillian_mpo(5, 8, seed=2)))
    assert syn.bond_dims([3] * 5, 4) == orc.bond_dims([3] * 5, 4)
    assert [c.shape for c in syn.random_mps_cores([3] * 5, 4)] == [(a, 3, b) for a, b in orc.bond_dims([3] * 5, 4)]


def test_reduced_density_nc_layout_and_reader(tmp_path):
    """reduced_density.nc as the reference lays it out (Properties._create_nc_file / _export_reduced_densities,
    properties.py:122-209): dimension step (unlimited), state, Q{idof}; time(step); rho_{key}_{istate}(step, Q.., Q..).
    Without the netCDF4 package the compound complex type becomes a trailing (real, imag) dimension of a NetCDF-3 file;
    the reader (the reference's util/read_nc.py contract) returns {"time": ..., key: complex array} either way."""
    from scipy.io import netcdf_file

    from pytdscf_amd.util import read_nc
    from pytdscf_amd.util.nc_writer import write_reduced_density_nc

    rng = np.random.default_rng(3)
    times = [0.0, 0.1, 0.2]
    recs = [{(3, 3): rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2)),
             (0, 0, 3, 3): rng.standard_normal((4, 4, 2, 2)) + 1j * rng.standard_normal((4, 4, 2, 2)),
             (1,): rng.standard_normal(5) + 0j} for _ in times]
    path = str(tmp_path / "reduced_density.nc")
    fmt = write_reduced_density_nc(path, times, recs)
    assert fmt in ("NETCDF4", "NETCDF3")
    got = read_nc(path, [(3, 3), (0, 0, 3, 3), (1,)])
    np.testing.assert_array_equal(got["time"], times)
    for k in recs[0]:
        np.testing.assert_array_equal(got[k], np.array([r[k] for r in recs]))
    with pytest.raises(ValueError):
        read_nc(path, [(2, 2)])
    if fmt == "NETCDF3":
        with netcdf_file(path, "r", mmap=False) as f:
            assert f.dimensions["step"] is None and f.dimensions["state"] == 1  # unlimited record dimension
            assert f.dimensions["Q3"] == 2 and f.dimensions["Q0"] == 4 and f.dimensions["Q1"] == 5
            assert f.variables["rho_(3, 3)_0"].dimensions == ("step", "Q3", "Q3", "complex")
            assert f.variables["rho_(0, 0, 3, 3)_0"].dimensions == ("step", "Q0", "Q0", "Q3", "Q3", "complex")
            assert f.variables["time"].dimensions == ("step",)


def test_subspace_projection_host_side(golden):
    """Host side of ``Model(space="liouville", subspace_inds=...)`` (no GPU): the shell's operator projection equals the
    oracle's restatement of ``TensorHamiltonian.project_subspace`` (hamiltonian_cls.py:852-880), projected site dimensions
    and trimmed bond caps follow ``LatticeInfo.get_bond_dim`` on the projected lattice (_mps_mpo.py:196-220)."""
    from oracle import tdvp_oracle as orc
    from pytdscf_amd import Exciton, Model
    from pytdscf_amd.mps import bond_dims, product_state_cores

    g = golden("liouville_subspace.npz")
    n, D = int(g["nsite"]), int(g["bond_dim"])
    inds = {int(q): tuple(int(x) for x in g[f"sub{int(q)}"]) for q in g["sub_sites"]}
    mpo = [g[f"mpo{i}"] for i in range(n)]
    model = Model([Exciton(nstate=4) for _ in range(n)], operators={"hamiltonian": mpo}, bond_dim=D, space="liouville",
                  subspace_inds=inds)
    for a, b in zip(model.project_mpo(model.hamiltonian.as_mpo(model.dims)), orc.project_subspace_mpo(mpo, inds)):
        assert a.shape[1:3] == b.shape[1:3]
    assert model.projected_dims() == [4, 2, 4, 3, 4]
    cores = orc.project_subspace_cores(orc.canonicalize_site0(product_state_cores([g[f"rho{i}"] for i in range(n)], D, space="liouville"),
                                                             scale=None), inds, D)
    assert [c.shape for c in cores] == [g[f"n1_final{i}"].shape for i in range(n)]
    assert [(c.shape[0], c.shape[2]) for c in cores] == bond_dims(model.projected_dims(), D)
    # the merged, projected MPO represents the projected operator: contract both chains to dense matrices
    def dense(cs):
        t = cs[0]
        for c in cs[1:]:
            t = np.tensordot(t, c, axes=(-1, 0))
        return t
    da = dense(model.project_mpo(model.hamiltonian.as_mpo(model.dims)))  # whole chain: the outer MPO bonds are 1
    db = dense(orc.project_subspace_mpo(mpo, inds))
    assert da.shape == db.shape
    np.testing.assert_allclose(da, db, atol=1e-12 * max(1.0, np.abs(db).max()))


def _parse_cdf2(buf):
    """A NetCDF classic / 64-bit-offset header read BYTE BY BYTE from the format's definition (magic, numrecs, dimension /
    attribute / variable lists with tags 0x0A / 0x0C / 0x0B, names and values padded to 4 bytes, big endian): independent of
    scipy's reader and of pytdscf_amd.util.read_nc.  Returns (version, numrecs, dims [(name, size)], gatts {name: bytes},
    vars [(name, dim ids, nc_type, vsize, begin)], record size)."""
    import struct

    pos = [0]

    def take(n):
        out = buf[pos[0] : pos[0] + n]
        assert len(out) == n
        pos[0] += n
        return out

    def u32():
        return struct.unpack(">I", take(4))[0]

    def name():
        n = u32()
        s = take(n).decode()
        take((-n) % 4)
        return s

    def att_list():
        tag, n = u32(), u32()
        assert (tag, n) == (0, 0) or tag == 0x0C
        out = {}
        for _ in range(n):
            k = name()
            typ, cnt = u32(), u32()
            size = {1: 1, 2: 1, 3: 2, 4: 4, 5: 4, 6: 8}[typ] * cnt
            out[k] = take(size)
            take((-size) % 4)
        return out

    assert take(3) == b"CDF"
    version = take(1)[0]
    numrecs = u32()
    tag, ndim = u32(), u32()
    assert tag == 0x0A
    dims = []
    for _ in range(ndim):
        k = name()
        dims.append((k, u32()))
    gatts = att_list()
    tag, nvar = u32(), u32()
    assert tag == 0x0B
    vs = []
    for _ in range(nvar):
        k = name()
        ids = [u32() for _ in range(u32())]
        att_list()
        typ, vsize = u32(), u32()
        begin = struct.unpack(">Q", take(8))[0] if version == 2 else u32()
        vs.append((k, ids, typ, vsize, begin))
    recsize = sum(v[3] for v in vs if v[1] and dims[v[1][0]][1] == 0)
    return version, numrecs, dims, gatts, vs, recsize


def test_reduced_density_nc_bytes_against_the_layout_written_out_by_hand(tmp_path):
    """The NetCDF-3 form of reduced_density.nc, checked at the byte level against an expectation written out from the
    reference's layout (Properties._create_nc_file, properties.py:156-209: dimensions step (record), state, then Q{idof} in
    order of first appearance over the keys; variable time(step) f8; rho_{key}_{istate}(step, Q.., Q..) per key -- here
    with the trailing (real, imag) dimension that stands in for the NETCDF4 compound type): header fields, variable
    order, types, record layout, and the numbers themselves at the offsets the header promises."""
    import struct

    from pytdscf_amd.util.nc_writer import write_reduced_density_nc

    rng = np.random.default_rng(4)
    times = [0.0, 0.25]
    recs = [{(0, 0): rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3)),
             (0, 2): rng.standard_normal((3, 2)) + 1j * rng.standard_normal((3, 2))} for _ in times]
    path = str(tmp_path / "reduced_density.nc")
    assert write_reduced_density_nc(path, times, recs, fmt="NETCDF3") == "NETCDF3"
    buf = open(path, "rb").read()
    version, numrecs, dims, gatts, vs, recsize = _parse_cdf2(buf)
    assert version == 2 and numrecs == len(times)  # 64-bit offsets; one record per saved step
    assert dims == [("step", 0), ("state", 1), ("complex", 2), ("Q0", 3), ("Q2", 2)]  # size 0 = THE record dimension
    assert b"trailing dimension 'complex'" in gatts["layout"]
    NC_DOUBLE = 6
    names = [v[0] for v in vs]
    assert names == ["time", "rho_(0, 0)_0", "rho_(0, 2)_0"]  # the reference's names: f"rho_{key}_{istate}" with key a tuple
    did = {n: i for i, (n, _) in enumerate(dims)}
    want_dims = {"time": ["step"], "rho_(0, 0)_0": ["step", "Q0", "Q0", "complex"], "rho_(0, 2)_0": ["step", "Q0", "Q2", "complex"]}
    want_vsize = {"time": 8, "rho_(0, 0)_0": 3 * 3 * 2 * 8, "rho_(0, 2)_0": 3 * 2 * 2 * 8}  # bytes of ONE record
    for k, ids, typ, vsize, begin in vs:
        assert ids == [did[d] for d in want_dims[k]] and typ == NC_DOUBLE and vsize == want_vsize[k], k
    assert recsize == sum(want_vsize.values())
    # records are interleaved: record r of a variable sits at begin + r * recsize
    begins = {k: b for k, _, _, _, b in vs}
    assert begins["rho_(0, 0)_0"] == begins["time"] + 8 and begins["rho_(0, 2)_0"] == begins["rho_(0, 0)_0"] + want_vsize["rho_(0, 0)_0"]
    assert len(buf) == begins["time"] + numrecs * recsize
    for r, (t, rec) in enumerate(zip(times, recs)):
        assert struct.unpack(">d", buf[begins["time"] + r * recsize :][:8])[0] == t
        for key in rec:
            k = f"rho_{key}_0"
            n = want_vsize[k] // 8
            vals = np.frombuffer(buf, dtype=">f8", count=n, offset=begins[k] + r * recsize).reshape(rec[key].shape + (2,))
            np.testing.assert_array_equal(vals[..., 0], rec[key].real)
            np.testing.assert_array_equal(vals[..., 1], rec[key].imag)


def test_bench_line_helpers():
    """bench.py's bookkeeping: the traffic pass of the apply form that ran (newest round first), the compact `secondary`
    object and its summary inside `config`, one socket's physical cores for the CPU baseline."""
    import json
    import os

    import bench

    L, d, D, M = bench.WORKLOADS["C5"][:4]
    t, src = bench.committed_traffic("heff", L, d, D, M, "edge")
    assert src and src.endswith("_heff_traffic_C5_edge.json") and t > 0
    t2, src2 = bench.committed_traffic("heff", L, d, D, M, "chain")
    assert src2 and "edge" not in src2 and t2 > 0 and src2 != src
    assert int(os.path.basename(src)[1:3]) >= 4  # never an older round's file when a newer one of the same form exists
    t3, src3 = bench.committed_traffic("heff", 7, 3, 48, 5, "chain")
    assert t3 is None and src3 is None  # a shape nobody measured: no number
    pkg, cpus = bench.one_socket_physical_cores()
    assert pkg is not None and 1 <= len(cpus) <= len(os.sched_getaffinity(0)) and len(set(cpus)) == len(cpus)
    rec = {"workload": "C3: x", "value": 64.94690741, "unit": "sweeps/s", "steps": 20,
           "roofline": {"bound": "mfma", "frac": 0.5204772, "note": "long " * 200, "kernel": "k" * 300, "traffic_note": "t" * 200},
           "breakdown_ms": {"wall": 307.94383899, "phases_from": "p" * 100},
           "cpu_baseline": {"value": 0.107, "sample": "s" * 300, "pinned": True},
           "ensemble": {"replicas": {"2": {"value": 407.123456}, "4": {"value": 566.0}}, "note": "n" * 300}}
    c = bench.compact_record(rec)
    assert len(json.dumps(c)) < 600 and c["value"] == pytest.approx(rec["value"], rel=1e-5) and len(c["cpu_baseline"]["sample"]) == 90
    assert "note" not in c["roofline"] and c["roofline"]["frac"] == pytest.approx(0.5204772, rel=1e-5)
    s = bench.secondary_summary({"C3": rec, "C5": {"error": "boom"}})
    assert s["C3"]["value"] == pytest.approx(64.947, rel=1e-4) and s["C3"]["frac"] == pytest.approx(0.5205, rel=1e-3)
    assert s["C3_ensemble"] == {"2": pytest.approx(407.12, rel=1e-4), "4": 566.0} and "error" in s["C5"]
