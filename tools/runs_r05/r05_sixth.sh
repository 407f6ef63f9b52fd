#!/bin/bash
set -o pipefail
O=gpurun_out/r05f
mkdir -p $O
echo "== edge tests" | tee -a $O/progress.txt
timeout -k 10 500 python -m pytest tests/test_gpu_edge_apply.py tests/test_gpu_kernels.py -x -q -o faulthandler_timeout=200 > $O/edge_tests.txt 2>&1; rc=$?; tail -8 $O/edge_tests.txt; [ $rc -eq 0 ] || exit 1
B="python bench.py --no-cpu-baseline --secondary none"
run() { name=$1; shift; echo "== $name" | tee -a $O/progress.txt; env "$@" timeout -k 10 300 $B --workload ${WL:-C3} --steps ${ST:-20} --warmup 2 > $O/$name.json 2> $O/$name.err || { echo "FAILED $name" | tee -a $O/progress.txt; tail -5 $O/$name.err; return 1; }; }
run c3_chain MITDVP_EDGE_APPLY=0 || exit 1
run c3_edge_b4 MITDVP_EDGE_APPLY=1 || exit 1
run c3_edge_16 MITDVP_EDGE_APPLY=1 MITDVP_EPI_B4=0 || exit 1
WL=C4 ST=1 run c4_edge_b4 MITDVP_EDGE_APPLY=1 MITDVP_BENCH_BUDGET=170 || exit 1
WL=C4 ST=1 run c4_chain MITDVP_EDGE_APPLY=0 MITDVP_BENCH_BUDGET=170 || exit 1
python - <<'P' | tee -a gpurun_out/r05f/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05f/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.3f'%r['frac'], 'stage', [round(x,4) for x in r.get('stage_ms_per_apply',[])], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
