#!/bin/bash
set -o pipefail
O=gpurun_out/r05k
mkdir -p $O
echo "== qr tests" | tee -a $O/progress.txt
timeout -k 10 400 python -m pytest tests/test_gpu_qr_gauge_free.py tests/test_gpu_bench_contract.py::test_bench_single_gpu_contract -x -q -o faulthandler_timeout=300 > $O/tests.txt 2>&1; rc=$?; tail -6 $O/tests.txt; [ $rc -eq 0 ] || exit 1
MITDVP_QR_TRACE=1 timeout -k 10 100 python -c "
import sys; sys.path.insert(0,'.')
from pytdscf_amd.engine import qr_thin
print(qr_thin(shape=(4096,128), gauge_free=True, reps=2)[2])
" > $O/qr_trace.txt 2>&1; tail -4 $O/qr_trace.txt
timeout -k 10 200 python tools/qr_thin_probe.py 2>&1 | tee $O/qr_probe.txt || exit 1
echo "== per-site applies" | tee -a $O/progress.txt
timeout -k 10 200 python tools/heff_per_site.py C3 2>&1 | tee $O/heff_per_site_C3.txt || exit 1
echo "== profiles" | tee -a $O/progress.txt
timeout -k 20 800 bash tools/profile_r05.sh > $O/profile.log 2>&1; tail -30 $O/profile.log
