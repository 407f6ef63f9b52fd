#!/bin/bash
# round 5, first GPU call: new tests + same-box baselines of the secondary workloads with the existing switches
set -o pipefail
O=gpurun_out/r05a
mkdir -p $O
python -m pytest tests/test_mixedstate_exact.py tests/test_gpu_ensemble.py -m gpu -x -q -s > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -5 $O/tests.txt
B="python bench.py --no-cpu-baseline --secondary none"
$B --workload C3 --steps 20 --warmup 2 > $O/c3_base.json 2> $O/c3_base.err
MITDVP_ZGEMM_BIGTILE=1 $B --workload C3 --steps 20 --warmup 2 > $O/c3_bigtile.json 2> $O/c3_bigtile.err
MITDVP_EDGE_APPLY=1 $B --workload C3 --steps 20 --warmup 2 > $O/c3_edge.json 2> $O/c3_edge.err
MITDVP_TRIM_IDENTITY=0 $B --workload C3 --steps 20 --warmup 2 > $O/c3_notrim.json 2> $O/c3_notrim.err
$B --workload C5 --steps 4 --warmup 1 > $O/c5_base.json 2> $O/c5_base.err
$B --workload C2 --steps 100 --warmup 4 > $O/c2_base.json 2> $O/c2_base.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05a/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.3f'%r['frac'], 'stage', [round(x,4) for x in r.get('stage_ms_per_apply',[])], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
