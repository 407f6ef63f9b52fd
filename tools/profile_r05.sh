#!/bin/bash
# round-5 evidence (run on the GPU box from the repo root): fabric traffic of the small-bond regime (C2) and of the H_eff
# apply at the C3 / C5 centre shapes in the form that runs now (edge) and in the three-stage chain, matrix-pipe utilisation
# of the apply's launches, kernel statistics of the C3 / C5 legs.  Separate PMC passes, kernel trace only.
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r05prof; mkdir -p $OUT
export MITDVP_ROUND=05
export TMPDIR=/tmp; cd /tmp
echo "== c2 traffic" | tee -a $OUT/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_c2f -- python3 $REPO/tools/small_trace.py > $OUT/c2_fetch.out 2>&1
find /tmp/p_c2f -name "*counter_collection.csv" -exec cp {} $OUT/c2_fetch.csv \;
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_c2w -- python3 $REPO/tools/small_trace.py > $OUT/c2_write.out 2>&1
find /tmp/p_c2w -name "*counter_collection.csv" -exec cp {} $OUT/c2_write.csv \;
(cd $REPO && python3 tools/c2_traffic.py $OUT/c2_fetch.csv $OUT/c2_write.csv 5) | cut -c1-500 | tee -a $OUT/progress.txt
rm -f $OUT/c2_fetch.csv $OUT/c2_write.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python3 $REPO/tools/small_trace.py > $OUT/c2_trace.out 2>&1
find /tmp/p_c2 -name "*kernel_stats.csv" -exec cp {} $OUT/r05_c2_kernel_stats.csv \;
for spec in "C3 128 32 16 8" "C5 512 4 16 6"; do
  set -- $spec
  for edge in 1 0; do
    tag=$([ $edge = 1 ] && echo edge || echo chain)
    echo "== traffic $1 $tag" | tee -a $OUT/progress.txt
    export MITDVP_EDGE_APPLY=$edge
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf_$1_$tag -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/fetch_$1_$tag.out 2>&1
    find /tmp/tf_$1_$tag -name "*counter_collection.csv" -exec cp {} $OUT/fetch_$1_$tag.csv \;
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tw_$1_$tag -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/write_$1_$tag.out 2>&1
    find /tmp/tw_$1_$tag -name "*counter_collection.csv" -exec cp {} $OUT/write_$1_$tag.csv \;
    (cd $REPO && python3 tools/heff_traffic_center.py $OUT/fetch_$1_$tag.csv $OUT/write_$1_$tag.csv $1 $2 $3 $4 $5 $tag) | cut -c1-400 | tee -a $OUT/progress.txt
    rm -f $OUT/fetch_$1_$tag.csv $OUT/write_$1_$tag.csv
    if [ $edge = 1 ]; then
      rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/u_$1 -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/util_$1.out 2>&1
      find /tmp/u_$1 -name "*counter_collection.csv" -exec cp {} $OUT/util_$1.csv \;
      (cd $REPO && python3 tools/heff_util_center.py $OUT/util_$1.csv $1 $5 $tag) | cut -c1-800 | tee -a $OUT/progress.txt
      rm -f $OUT/util_$1.csv
    fi
    unset MITDVP_EDGE_APPLY
  done
done
cp $REPO/profiles/r05_* $OUT/ 2>/dev/null
for w in C3 C5; do
  echo "== kernel stats $w" | tee -a $OUT/progress.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$w -- python3 $REPO/bench.py --workload $w --steps 6 --warmup 2 --secondary none --no-cpu-baseline > $OUT/r05_bench_${w}_under_rocprof.json 2> $OUT/${w}_bench.err
  find /tmp/p_$w -name "*kernel_stats.csv" -exec cp {} $OUT/r05_$(echo $w | tr A-Z a-z)_kernel_stats.csv \;
done
ls $OUT | tee -a $OUT/progress.txt
