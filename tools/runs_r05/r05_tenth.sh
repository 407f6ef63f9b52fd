#!/bin/bash
set -o pipefail
O=gpurun_out/r05j
mkdir -p $O
echo "== qr tests + fixed tests" | tee -a $O/progress.txt
timeout -k 10 600 python -m pytest tests/test_gpu_qr_gauge_free.py "tests/test_gpu_kernels.py::test_block_sparse_w_stage_is_bitwise_the_dense_one" tests/test_gpu_parity.py tests/test_gpu_fullsize_oracle.py -x -q -o faulthandler_timeout=300 > $O/tests.txt 2>&1; rc=$?; tail -6 $O/tests.txt; [ $rc -eq 0 ] || exit 1
echo "== qr trace / probe" | tee -a $O/progress.txt
MITDVP_QR_TRACE=1 timeout -k 10 100 python -c "
import sys; sys.path.insert(0,'.')
from pytdscf_amd.engine import qr_thin
print(qr_thin(shape=(4096,128), gauge_free=True, reps=2)[2])
" > $O/qr_trace.txt 2>&1; tail -4 $O/qr_trace.txt
timeout -k 10 200 python tools/qr_thin_probe.py 2>&1 | tee $O/qr_probe.txt || exit 1
B="python bench.py --no-cpu-baseline --secondary none"
timeout -k 10 300 $B --workload C3 --steps 20 --warmup 2 > $O/c3.json 2> $O/c3.err || exit 1
timeout -k 10 300 $B --workload C5 --steps 4 --warmup 1 > $O/c5.json 2> $O/c5.err || exit 1
timeout -k 10 300 $B --workload C2 --steps 100 --warmup 4 > $O/c2.json 2> $O/c2.err || exit 1
python - <<'P' | tee -a gpurun_out/r05j/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05j/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.4f'%r['frac'], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
