#!/bin/bash
# round 5, second GPU call: same-box baselines with the existing switches, the gauge-free QR tests, then the ensemble tests
set -o pipefail
O=gpurun_out/r05b
mkdir -p $O
B="python bench.py --no-cpu-baseline --secondary none"
run() { name=$1; shift; echo "== $name" | tee -a $O/progress.txt; env "$@" timeout -k 10 300 $B --workload ${WL:-C3} --steps ${ST:-20} --warmup 2 > $O/$name.json 2> $O/$name.err || { echo "FAILED $name" | tee -a $O/progress.txt; tail -5 $O/$name.err; return 1; }; }
run c3_lapack MITDVP_QR_GAUGE_FREE=0 || exit 1
run c3_gram MITDVP_QR_GAUGE_FREE=1 || exit 1
run c3_bigtile MITDVP_QR_GAUGE_FREE=1 MITDVP_ZGEMM_BIGTILE=1 || exit 1
run c3_edge MITDVP_QR_GAUGE_FREE=1 MITDVP_EDGE_APPLY=1 || exit 1
WL=C5 ST=4 run c5_lapack MITDVP_QR_GAUGE_FREE=0 || exit 1
WL=C5 ST=4 run c5_gram MITDVP_QR_GAUGE_FREE=1 || exit 1
WL=C2 ST=100 run c2_base MITDVP_QR_GAUGE_FREE=1 || exit 1
python - <<'P' | tee -a gpurun_out/r05b/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05b/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.3f'%r['frac'], 'stage', [round(x,4) for x in r.get('stage_ms_per_apply',[])], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
echo "== qr tests" | tee -a $O/progress.txt
timeout -k 10 400 python -m pytest tests/test_gpu_qr_gauge_free.py -x -q -s -o faulthandler_timeout=120 > $O/qr_tests.txt 2>&1; rc=$?; tail -25 $O/qr_tests.txt; [ $rc -eq 0 ] || exit 1
echo "== ensemble tests" | tee -a $O/progress.txt
timeout -k 10 240 python -m pytest tests/test_gpu_ensemble.py -x -q -s -o faulthandler_timeout=60 > $O/ens_tests.txt 2>&1; rc=$?; tail -40 $O/ens_tests.txt; exit $rc
