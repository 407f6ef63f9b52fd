#!/bin/bash
set -o pipefail
O=gpurun_out/r05g
mkdir -p $O
echo "== ensemble tests" | tee -a $O/progress.txt
timeout -k 10 400 python -m pytest tests/test_gpu_ensemble.py -x -q -s -o faulthandler_timeout=100 > $O/ens_tests.txt 2>&1; rc=$?; tail -30 $O/ens_tests.txt; [ $rc -eq 0 ] || exit 1
echo "== ensemble probe" | tee -a $O/progress.txt
timeout -k 10 300 python tools/ensemble_partition_probe.py 2>&1 | tee $O/ens_probe.txt || exit 1
B="python bench.py --no-cpu-baseline --secondary none"
timeout -k 10 300 $B --workload C3 --steps 20 --warmup 2 > $O/c3.json 2> $O/c3.err || exit 1
timeout -k 10 300 $B --workload C2 --steps 100 --warmup 4 > $O/c2.json 2> $O/c2.err || exit 1
python - <<'P' | tee -a gpurun_out/r05g/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05g/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.3f'%r['frac'], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
