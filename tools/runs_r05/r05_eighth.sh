#!/bin/bash
set -o pipefail
O=gpurun_out/r05h
mkdir -p $O
echo "== edge + ensemble tests" | tee -a $O/progress.txt
timeout -k 10 500 python -m pytest tests/test_gpu_edge_apply.py tests/test_gpu_ensemble.py tests/test_gpu_3m_numerics.py -x -q -o faulthandler_timeout=200 > $O/tests.txt 2>&1; rc=$?; tail -8 $O/tests.txt; [ $rc -eq 0 ] || exit 1
echo "== ensemble probe" | tee -a $O/progress.txt
timeout -k 10 300 python tools/ensemble_partition_probe.py 2>&1 | tee $O/ens_probe.txt || exit 1
B="python bench.py --no-cpu-baseline --secondary none"
timeout -k 10 300 $B --workload C3 --steps 20 --warmup 2 > $O/c3.json 2> $O/c3.err || exit 1
timeout -k 10 300 $B --workload C5 --steps 4 --warmup 1 > $O/c5.json 2> $O/c5.err || exit 1
MITDVP_EPI_B4=0 timeout -k 10 300 $B --workload C5 --steps 4 --warmup 1 > $O/c5_nob4.json 2> $O/c5_nob4.err || exit 1
python - <<'P' | tee -a gpurun_out/r05h/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.3f'%r['frac'], 'stage', [round(x,4) for x in r.get('stage_ms_per_apply',[])], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
