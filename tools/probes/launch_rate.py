"""Host launch rate: back-to-back launches of a tiny product on one stream (time per launch = max(host enqueue, kernel))."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytdscf_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(0)
for n in (16, 64, 256):
    a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))); b = a.copy(); c = np.zeros((n, n), complex)
    one = (C.c_double * 2)(1.0, 0.0); zero = (C.c_double * 2)(0.0, 0.0)
    ms = C.c_double()
    p = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    for reps in (200, 5000):
        t0 = time.perf_counter()
        rc = lib.mitdvp_zgemm(0, 0, 0, 0, 0, n, n, n, p(a), p(b), p(c), one, zero, -1, reps, C.byref(ms))
        assert rc == 0
        print("n=%d reps=%d: %.2f us per launch (wall of the whole call %.1f ms)" % (n, reps, ms.value * 1e3, (time.perf_counter() - t0) * 1e3))
