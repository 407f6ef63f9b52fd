#!/bin/bash
set -o pipefail
O=gpurun_out/r05i
mkdir -p $O
echo "== qr trace" | tee -a $O/progress.txt
MITDVP_QR_TRACE=1 timeout -k 10 100 python -c "
import sys; sys.path.insert(0,'.')
from pytdscf_amd.engine import qr_thin
print(qr_thin(shape=(4096,128), gauge_free=True, reps=2)[2])
" > $O/qr_trace.txt 2>&1; tail -8 $O/qr_trace.txt
B="python bench.py --no-cpu-baseline --secondary none"
echo "== ensemble probe (1024 threads)" | tee -a $O/progress.txt
timeout -k 10 300 python tools/ensemble_partition_probe.py 2>&1 | tee $O/ens_probe_1024.txt || exit 1
echo "== ensemble probe (512 threads)" | tee -a $O/progress.txt
MITDVP_LIB=$PWD/pytdscf_amd/csrc/libmitdvp_ss512.so timeout -k 10 300 python tools/ensemble_partition_probe.py 2>&1 | tee $O/ens_probe_512.txt || exit 1
timeout -k 10 300 $B --workload C2 --steps 100 --warmup 4 > $O/c2_1024.json 2> $O/c2_1024.err || exit 1
MITDVP_LIB=$PWD/pytdscf_amd/csrc/libmitdvp_ss512.so timeout -k 10 300 $B --workload C2 --steps 100 --warmup 4 > $O/c2_512.json 2> $O/c2_512.err || exit 1
python - <<'P' | tee -a gpurun_out/r05i/progress.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/r05i/*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; b=d['breakdown_ms']
        print(f.split('/')[-1], 'value %.4g'%d['value'], 'frac %.4f'%r['frac'], 'brk', {k:round(v,1) for k,v in b.items() if isinstance(v,(int,float))})
    except Exception as e: print(f, 'ERR', e)
P
