"""``read_nc(filename, sites)`` -- the reader analysis scripts use for the reduced densities a
propagation saved (pytdscf/util/read_nc.py): ``{"time": t[step], (i, j): rho[step, ...], ...}``.

The reference writes ``{jobname}_prop/reduced_density.nc`` with netCDF4 compound types (HDF5;
properties.py:156-209); this image has no HDF5 stack, so the shell writes the same content as
``reduced_density.npz`` (arrays ``time`` and ``rho_{key}``).  The reader takes either: an ``.npz``
written by the shell, or -- where the ``netCDF4`` package exists -- an ``.nc`` file written by the
reference (variables ``rho_{key}_0`` with real / imag members)."""

from __future__ import annotations

import os

import numpy as np


def read_nc(filename: str, sites) -> dict:
    if not os.path.exists(filename) and filename.endswith(".nc") and os.path.exists(filename[:-3] + ".npz"):
        filename = filename[:-3] + ".npz"  # same directory, the shell's container format
    data = {}
    if filename.endswith(".npz"):
        with np.load(filename) as z:
            data["time"] = np.array(z["time"])
            for key in sites:
                name = f"rho_{tuple(key)}"
                if name not in z.files:
                    raise ValueError(f"Density data for site {key} varname='{name}' not found in {filename}")
                data[key] = np.array(z[name])
        return data
    try:
        import netCDF4 as nc
    except ImportError as e:  # pragma: no cover - not installable in this image
        raise ImportError("reading a netCDF4 file needs the netCDF4 package; the shell writes reduced_density.npz") from e
    with nc.Dataset(filename, "r") as f:  # pragma: no cover
        data["time"] = np.array(f.variables["time"][:])
        for key in sites:
            name = f"rho_{key}_0"
            if name not in f.variables:
                raise ValueError(f"Density data for site {key} varname='{name}' not found in {filename}")
            v = f.variables[name][:]
            data[key] = np.array(v["real"]) + 1.0j * np.array(v["imag"])
    return data
