"""Device time of the sweep's thin QR by path (tools/r05_third.sh runs it under rocprofv3 --kernel-trace --stats)."""
import sys

sys.path.insert(0, ".")
from pytdscf_amd.engine import qr_thin

shapes = [(4096, 128), (2048, 512), (1024, 128), (4096, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for shape in shapes:
    for gf in (True, False):
        _, _, info = qr_thin(shape=shape, gauge_free=gf, reps=20)
        print(shape, "gauge-free" if gf else "householder", info, flush=True)
