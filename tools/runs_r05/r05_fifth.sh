#!/bin/bash
set -o pipefail
O=gpurun_out/r05e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== mfma 4x4x4 layout" | tee -a $O/progress.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_4x4x4_layout.hip -o /tmp/mfma4 && timeout -k 5 60 /tmp/mfma4 > $O/mfma4.txt 2>&1; tail -3 $O/mfma4.txt
echo "== qr probe" | tee -a $O/progress.txt
timeout -k 10 200 python tools/qr_thin_probe.py 2>&1 | tee $O/qr_probe.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o qr -- python3 tools/qr_thin_probe.py 4096x128 2048x512 > $O/qr_prof.txt 2>&1 || { tail -5 $O/qr_prof.txt; exit 1; }
python - <<'P' | tee -a gpurun_out/r05e/progress.txt
import csv,glob
for f in glob.glob('gpurun_out/r05e/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:16]: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
P
echo "== qr tests" | tee -a $O/progress.txt
timeout -k 10 400 python -m pytest tests/test_gpu_qr_gauge_free.py -x -q -s -o faulthandler_timeout=120 > $O/qr_tests.txt 2>&1; rc=$?; tail -8 $O/qr_tests.txt; exit $rc
