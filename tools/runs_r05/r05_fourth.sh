#!/bin/bash
set -o pipefail
O=gpurun_out/r05d
mkdir -p $O
echo "== qr probe" | tee -a $O/progress.txt
timeout -k 10 200 python tools/qr_thin_probe.py 2>&1 | tee $O/qr_probe.txt || exit 1
echo "== qr tests" | tee -a $O/progress.txt
timeout -k 10 400 python -m pytest tests/test_gpu_qr_gauge_free.py -x -q -s -o faulthandler_timeout=120 > $O/qr_tests.txt 2>&1; rc=$?; tail -15 $O/qr_tests.txt; [ $rc -eq 0 ] || exit 1
B="python bench.py --no-cpu-baseline --secondary none"
timeout -k 10 300 $B --workload C3 --steps 20 --warmup 2 > $O/c3.json 2> $O/c3.err || exit 1
timeout -k 10 300 $B --workload C5 --steps 4 --warmup 1 > $O/c5.json 2> $O/c5.err || exit 1
python - <<'P' | tee -a gpurun_out/r05d/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05d/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.3f'%r['frac'], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
echo "== mixedstate then ensemble in one process" | tee -a $O/progress.txt
timeout -k 10 300 python -X faulthandler -m pytest tests/test_mixedstate_exact.py tests/test_gpu_ensemble.py -m gpu -x -q -s -o faulthandler_timeout=45 > $O/repro.txt 2>&1; rc=$?; tail -60 $O/repro.txt; exit $rc
