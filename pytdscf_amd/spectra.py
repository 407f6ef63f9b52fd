"""Spectra from the auto-correlation file a propagation writes (``{jobname}_prop/autocorr.dat``):
the post-processing half of the reference's ``pytdscf/spectra.py`` (load -> window -> FFT ->
export), so that existing analysis scripts keep working on the engine's output files.  Pinned by
the reference's own known-answer test (tests/test_spectra.py: peak 28860.651565826236 at
2684.0796620397296 cm-1 for tests/autocorr.dat)."""

from __future__ import annotations

import numpy as np

from . import units

# the reference's rounded conversion factors (spectra.py:83, :89): kept literally so that the
# numbers of an existing analysis do not move in the 7th digit
_CM1_PER_INV_FS = 1.0e15 * 3.33564e-11


def load_autocorr(dat_file: str):
    """(time in fs, complex auto-correlation) from a two-column ``autocorr.dat`` (spectra.py:10-33)."""
    with open(dat_file) as f:
        head = f.readline()
        if "fs" not in head:
            print("WARNING: time unit is not fs")
        data = np.loadtxt(f, usecols=(0, 1), dtype=np.complex128)
    time_fs, autocorr = data[:, 0].real, data[:, 1]
    if time_fs[0] != 0.0:
        raise ValueError(f"time is not starting from 0.0 but {time_fs[0]}")
    if autocorr[0] != 1.0:
        raise ValueError(f"auto-correlation at t=0 is not 1.0 but {autocorr[0]}")
    return time_fs, autocorr


def _window(time_fs: np.ndarray, name):
    x = np.pi * time_fs / time_fs[-1] / 2
    if name == "cos2":
        return np.cos(x) ** 2
    if name == "cos":
        return np.cos(x)
    if name is None:
        return np.ones_like(time_fs)
    raise ValueError(f"window function {name} is not defined")


def ifft_autocorr(time_fs, autocorr, E_shift: float = 0.0, window="cos2", power: bool = False):
    """(wave number in cm-1, intensity): cubic resampling of the auto-correlation on a uniform
    grid of half the largest time increment, window, FFT (spectra.py:62-100).  ``power=False``
    gives the absorption line shape omega * Re FT, shifted by ``E_shift`` (eV)."""
    from scipy import interpolate

    time_fs = np.asarray(time_fs, dtype=float)
    resample = interpolate.interp1d(time_fs, autocorr, kind="cubic")
    dt = np.amax(time_fs[1:-1] - time_fs[0:-2]) / 2
    n = int((time_fs[-1] - time_fs[0]) / dt)
    t = np.arange(n) * dt
    signal = resample(t) * _window(t, window)
    omega = np.flipud(-np.fft.fftshift(np.fft.fftfreq(n, dt)) * _CM1_PER_INV_FS)
    ft = np.flipud(np.fft.fftshift(np.fft.fft(signal) * dt).real)
    if power:
        return omega, ft
    omega = omega - E_shift * units.au_in_cm1 / units.au_in_eV
    return omega, ft * omega


def export_spectrum(wave_number, intensity, filename: str = "spectrum.dat"):
    with open(filename, "w") as f:
        f.write("# wave_number[cm-1]\t intensity[arb. unit]\n")
        np.savetxt(f, np.column_stack([np.ravel(wave_number), np.ravel(intensity)]), fmt="%15.8f", delimiter="\t")


def plot_autocorr(time_fs, autocorr, gui: bool = True):
    import matplotlib.pyplot as plt

    plt.figure(figsize=(15, 6), dpi=80)
    plt.plot(time_fs, np.real(autocorr), color="blue", label="real")
    plt.plot(time_fs, np.imag(autocorr), color="red", label="imag")
    plt.xlabel("time[fs]")
    plt.title("auto-corr <Ψ(0)|Ψ(t)>")
    plt.legend()
    plt.show(block=gui)


def plot_spectrum(wave_number, intensity, lower_bound: float = 0.0, upper_bound: float = 4000.0, filename: str = "spectrum.pdf",
                  export: bool = True, show_in_eV: bool = False, show_in_nm: bool = False, gui: bool = True, normalize: bool = True,
                  figsize=(15, 6), dpi: int = 80, title: str = "IR absorption spectrum"):
    """Window [lower_bound, upper_bound] of the spectrum on a 1 cm-1 grid (cubic interpolation),
    optionally normalised to the global maximum and exported next to the figure (spectra.py:140-)."""
    from scipy import interpolate

    if show_in_eV and show_in_nm:
        raise ValueError("choose one of show_in_eV / show_in_nm")
    lo, hi = np.searchsorted(wave_number, lower_bound), np.searchsorted(wave_number, upper_bound)
    y = (intensity / max(intensity) if normalize else intensity)[lo - 1: hi + 1]
    x = wave_number[lo - 1: hi + 1]
    grid = np.arange(lower_bound, upper_bound, 1, dtype=np.float64)
    vals = interpolate.interp1d(x, y, kind="cubic")(grid)
    if export:
        stem = filename.rsplit(".", 1)[0] if "." in filename else filename
        export_spectrum(grid, vals, filename=stem + ".dat")
    import matplotlib.pyplot as plt

    plt.figure(figsize=figsize, dpi=dpi)
    plt.title(title)
    if show_in_eV:
        plt.plot(grid / units.au_in_cm1 * units.au_in_eV, vals)
        plt.xlabel("energy [eV]")
    elif show_in_nm:
        with np.errstate(divide="ignore"):
            plt.plot(1.0e7 / grid, vals)
        plt.xlabel("wavelength [nm]")
    else:
        plt.plot(grid, vals)
        plt.xlabel("wave number [cm-1]")
    plt.ylabel("intensity [arb. unit]")
    plt.savefig(filename)
    plt.show(block=gui)
    return grid, vals
