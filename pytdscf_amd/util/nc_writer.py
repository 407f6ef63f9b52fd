"""``{jobname}_prop/reduced_density.nc`` in the reference's layout (``Properties._create_nc_file`` /
``_export_reduced_densities``, pytdscf/properties.py:122-209): dimensions ``step`` (unlimited), ``state``, ``Q{idof}``
per degree of freedom of the requested keys; variable ``time(step)``; one variable ``rho_{key}_{istate}(step, Q.., Q..)``
per key and state.

The reference writes NETCDF4 (HDF5) with a compound type ``complex128 = {real: f8, imag: f8}``.  With the ``netCDF4``
package importable the file is written exactly so.  This image has no HDF5 stack: the file is then NetCDF-3 64-bit
(``scipy.io.netcdf_file``), which has no compound types, and every ``rho_*`` variable carries a trailing dimension
``complex`` of length 2 (real, imag) instead; ``pytdscf_amd.util.read_nc`` reads both forms (and the reference's own
files), returning the same dictionary as the reference's ``util/read_nc.py``."""

from __future__ import annotations

import os

import numpy as np


def _have_netcdf4():
    try:
        import netCDF4  # noqa: F401

        return True
    except Exception:  # noqa: BLE001
        return False


def write_reduced_density_nc(path: str, times, records, nstate: int = 1, fmt: str | None = None) -> str:
    """``records[k]``: {key tuple: ndarray of rho at ``times[k]``} (one electronic state, index 0, like the reference's
    MPS standard-method path).  Returns the format written: "NETCDF4" or "NETCDF3"."""
    times = np.asarray(times, dtype=np.float64)
    keys = list(records[0]) if records else []
    for key in keys:
        assert tuple(key) == tuple(sorted(key)), f"Reduced density key {key} must be ascending order"  # properties.py:176
    if os.path.exists(path):
        os.remove(path)
    fmt = fmt or ("NETCDF4" if _have_netcdf4() else "NETCDF3")
    qdims = {}
    for key in keys:
        shape = np.asarray(records[0][key]).shape
        # one axis per requested leg, in key order (two equal entries = ket and bra of that site)
        for ax, idof in enumerate(key):
            qdims.setdefault(f"Q{idof}", shape[ax])
    if fmt == "NETCDF4":
        import netCDF4 as nc

        with nc.Dataset(path, "w", format="NETCDF4") as f:
            f.createDimension("step", None)
            f.createDimension("state", nstate)
            c128 = np.dtype([("real", np.float64), ("imag", np.float64)])
            c128_t = f.createCompoundType(c128, "complex128")
            for name, n in qdims.items():
                f.createDimension(name, n)
            tv = f.createVariable("time", "f8", ("step",))
            vs = {key: f.createVariable(f"rho_{tuple(key)}_0", c128_t, ("step",) + tuple(f"Q{i}" for i in key)) for key in keys}
            for row, (t, rec) in enumerate(zip(times, records)):
                tv[row] = t
                for key in keys:
                    d = np.asarray(rec[key])
                    data = np.empty(d.shape, c128)
                    data["real"], data["imag"] = d.real, d.imag
                    vs[key][row] = data
        return fmt
    from scipy.io import netcdf_file

    with netcdf_file(path, "w", version=2) as f:
        f.createDimension("step", None)
        f.createDimension("state", nstate)
        f.createDimension("complex", 2)
        for name, n in qdims.items():
            f.createDimension(name, n)
        f.layout = "rho_* variables: trailing dimension 'complex' = (real, imag); NETCDF4 compound type unavailable"
        tv = f.createVariable("time", "f8", ("step",))
        vs = {key: f.createVariable(f"rho_{tuple(key)}_0", "f8", ("step",) + tuple(f"Q{i}" for i in key) + ("complex",))
              for key in keys}
        for row, (t, rec) in enumerate(zip(times, records)):
            tv[row] = t
            for key in keys:
                d = np.asarray(rec[key], dtype=np.complex128)
                vs[key][row] = np.stack([d.real, d.imag], axis=-1)
    return fmt
