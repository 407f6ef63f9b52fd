#!/bin/bash
set -o pipefail
O=gpurun_out/r05c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== qr probe" | tee -a $O/progress.txt
timeout -k 10 200 python tools/qr_thin_probe.py 2>&1 | tee $O/qr_probe.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o qr -- python3 tools/qr_thin_probe.py 4096x128 > $O/qr_prof.txt 2>&1 || { tail -5 $O/qr_prof.txt; exit 1; }
python - <<'P' | tee -a gpurun_out/r05c/progress.txt
import csv,glob
for f in glob.glob('gpurun_out/r05c/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:25]: print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
P
echo "== qr tests" | tee -a $O/progress.txt
timeout -k 10 400 python -m pytest tests/test_gpu_qr_gauge_free.py -x -q -s -o faulthandler_timeout=120 > $O/qr_tests.txt 2>&1; rc=$?; tail -15 $O/qr_tests.txt; [ $rc -eq 0 ] || exit 1
echo "== ensemble tests" | tee -a $O/progress.txt
timeout -k 10 240 python -m pytest tests/test_gpu_ensemble.py -x -q -s -o faulthandler_timeout=60 > $O/ens_tests.txt 2>&1; rc=$?; tail -40 $O/ens_tests.txt; exit $rc
