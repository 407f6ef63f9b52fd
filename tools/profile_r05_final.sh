#!/bin/bash
# end-of-round-5 evidence (GPU box, from the repo root): fabric traffic + matrix-pipe utilisation of the H_eff apply in the
# form that runs (edge) at the C3 / C5 / C4 centre shapes, kernel statistics of the C3 / C5 / C4 legs, the C3 timeline.
# Separate PMC passes, kernel trace only.  Progress in gpurun_out/r05final/progress.txt.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05final; mkdir -p $OUT
export MITDVP_ROUND=05
export TMPDIR=/tmp; cd /tmp
for spec in "C3 128 32 16 8" "C5 512 4 16 6" "C4 1024 16 32 3"; do
  set -- $spec
  tag=edge
  echo "== traffic $1 $tag $(date +%T)" | tee -a $OUT/progress.txt
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf_$1 -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/fetch_$1.out 2>&1 || { echo "fetch pass failed" | tee -a $OUT/progress.txt; exit 1; }
  find /tmp/tf_$1 -name "*counter_collection.csv" -exec cp {} $OUT/fetch_$1.csv \;
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tw_$1 -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/write_$1.out 2>&1 || { echo "write pass failed" | tee -a $OUT/progress.txt; exit 1; }
  find /tmp/tw_$1 -name "*counter_collection.csv" -exec cp {} $OUT/write_$1.csv \;
  (cd $REPO && python3 tools/heff_traffic_center.py $OUT/fetch_$1.csv $OUT/write_$1.csv $1 $2 $3 $4 $5 $tag) | cut -c1-400 | tee -a $OUT/progress.txt
  rm -f $OUT/fetch_$1.csv $OUT/write_$1.csv
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/u_$1 -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/util_$1.out 2>&1 || { echo "util pass failed" | tee -a $OUT/progress.txt; exit 1; }
  find /tmp/u_$1 -name "*counter_collection.csv" -exec cp {} $OUT/util_$1.csv \;
  (cd $REPO && python3 tools/heff_util_center.py $OUT/util_$1.csv $1 $5 $tag) | cut -c1-800 | tee -a $OUT/progress.txt
  rm -f $OUT/util_$1.csv
  rm -rf /tmp/tf_$1 /tmp/tw_$1 /tmp/u_$1
done
cp $REPO/profiles/r05_heff_traffic_*_edge.json $REPO/profiles/r05_heff_mfma_util_*_edge.json $OUT/ 2>/dev/null
for w in C3 C5 C4; do
  echo "== kernel stats $w $(date +%T)" | tee -a $OUT/progress.txt
  st=6; [ $w = C4 ] && st=2
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -- python3 $REPO/bench.py --workload $w --steps $st --warmup 1 --secondary none --no-cpu-baseline > $OUT/r05_bench_${w}_under_rocprof_final.json 2> $OUT/${w}_bench.err || { echo "stats $w failed" | tee -a $OUT/progress.txt; exit 1; }
  find /tmp/p_$w -name "*kernel_stats.csv" -exec cp {} $OUT/r05_$(echo $w | tr A-Z a-z)_kernel_stats_final.csv \;
  if [ $w = C3 ]; then
    f=$(find /tmp/p_$w -name '*kernel_trace.csv' | head -1)
    (cd $REPO && python3 tools/timeline_gaps.py $f 1 0 400 30 0.2 0.5) > $OUT/r05_c3_timeline_timed_region.txt 2>&1
  fi
  rm -rf /tmp/p_$w
done
ls $OUT | tee -a $OUT/progress.txt
