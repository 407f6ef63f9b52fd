"""one shape, one panel algorithm (for rocprofv3): python tools/qr_ab_one.py m n fast reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytdscf_amd import engine as E
m, n, fast, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
E.set_qr_fast(bool(fast))
print(E.bench_qr(m, n, reps=reps))
