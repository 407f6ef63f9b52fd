#!/bin/bash
# S3 of the H_eff apply (K = 32768) with the contraction split over n launches-in-one (MITDVP_LONGK_SPLITS): kernel time and
# FETCH_SIZE of the NT kernel per setting.  Run on the GPU box:  bash tools/longk_probe.sh "1 4 8 16"
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/longk; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for n in ${1:-1 8}; do
  export MITDVP_LONGK_SPLITS=$n
  rm -rf /tmp/lk_$n
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/lk_$n -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 2 > $OUT/probe_$n.out 2>&1
  find /tmp/lk_$n -name "*counter_collection.csv" -exec cp {} $OUT/fetch_$n.csv \;
  python3 - "$OUT/fetch_$n.csv" $n <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"][:70]
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6; a[2] += float(r["Counter_Value"])
print("splits", sys.argv[2])
for k, (n, ms, f) in agg.items():
    if ms > 1.0: print(f"  {k:70s} calls {n:4d}  ms/call {ms/n:8.3f}  fetch GB/call {2*f*1024/1e9/n:8.2f}")
PY
done
