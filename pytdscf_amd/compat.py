"""Run scripts written for the reference WITHOUT editing them.

``pytdscf_amd.compat.install()`` registers this package under the reference's import names
(``pytdscf``, ``pytdscf.model_cls``, ``pytdscf.simulator_cls``, ``pytdscf.hamiltonian_cls``,
``pytdscf.dvr_operator_cls``, ``pytdscf.basis`` (+ ``pytdscf.basis._primints_cls``),
``pytdscf.units``, ``pytdscf.spectra``, ``pytdscf.kraus``, ``pytdscf.util`` (+ ``.read_nc``),
``pytdscf.wavefunction``) and, when the third-party ``discvar`` package is not installed, its
``HarmonicOscillator`` / ``Sine`` / ``Exponential`` names.  Nothing is copied; the entries are the
modules of this package.

    python -m pytdscf_amd.compat your_pytdscf_script.py [args...]      # or, in a script / notebook:
    import pytdscf_amd.compat; pytdscf_amd.compat.install(); import pytdscf

``backend="numpy"`` / ``"jax"`` in such scripts is accepted (with a warning) and runs on the HIP
engine; features outside the accelerated path raise ``NotImplementedError`` as documented."""

from __future__ import annotations

import importlib
import importlib.util
import runpy
import sys
import types


def install(name: str = "pytdscf", discvar: bool = True, force: bool = False) -> list[str]:
    """Register the aliases; returns the module names that were installed."""
    if not force and name not in sys.modules and importlib.util.find_spec(name) is not None:
        raise RuntimeError(f"a real '{name}' package is importable; pass force=True to shadow it for this process")
    import pytdscf_amd as pkg
    from pytdscf_amd import api, basis, dvr_operator_cls, hamiltonian_cls, kraus, model_cls, simulator_cls, spectra, units, util
    from pytdscf_amd.util import read_nc as read_nc_mod

    wavefunction = types.ModuleType(f"{name}.wavefunction")
    wavefunction.WFunc = api.WFunc
    primints = types.ModuleType(f"{name}.basis._primints_cls")
    primints.PrimBas_HO = basis.PrimBas_HO
    mods = {
        name: pkg,
        f"{name}.model_cls": model_cls,
        f"{name}.simulator_cls": simulator_cls,
        f"{name}.hamiltonian_cls": hamiltonian_cls,
        f"{name}.dvr_operator_cls": dvr_operator_cls,
        f"{name}.basis": basis,
        f"{name}.basis._primints_cls": primints,
        f"{name}.units": units,
        f"{name}.spectra": spectra,
        f"{name}.kraus": kraus,
        f"{name}.util": util,
        f"{name}.util.read_nc": read_nc_mod,
        f"{name}.wavefunction": wavefunction,
    }
    if discvar and "discvar" not in sys.modules and importlib.util.find_spec("discvar") is None:
        dv = types.ModuleType("discvar")
        dv.HarmonicOscillator, dv.Sine, dv.Exponential = basis.HarmonicOscillator, basis.Sine, basis.Exponential
        dv.PrimBas_HO = basis.PrimBas_HO
        mods["discvar"] = dv
    for k, v in mods.items():
        sys.modules[k] = v
    if not hasattr(pkg, "wavefunction"):
        pkg.wavefunction = wavefunction
    return sorted(mods)


def uninstall(name: str = "pytdscf") -> None:
    for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
        if getattr(sys.modules[k], "__name__", "").startswith("pytdscf_amd") or k.endswith(("wavefunction", "_primints_cls")):
            del sys.modules[k]
    dv = sys.modules.get("discvar")
    if dv is not None and getattr(dv, "__file__", None) is None:
        del sys.modules["discvar"]


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__)
        return 2
    install()
    sys.argv = argv
    runpy.run_path(argv[0], run_name="__main__")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
