#!/bin/bash
# round-4 evidence: bench lines of C2 / C3 / C5, fabric traffic of the apply (three-stage chain vs edge form) at the C3 / C5 / C4
# centre shapes (separate FETCH_SIZE / WRITE_SIZE passes, kernel trace only), kernel statistics of the driver's command
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r04prof; mkdir -p $OUT
for w in C2 C3 C5; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --secondary none > $OUT/r04_bench_$w.json 2> $OUT/r04_bench_$w.err
  tail -c 200 $OUT/r04_bench_$w.json; echo
done
export TMPDIR=/tmp; cd /tmp
for spec in "C3 128 32 16 8" "C5 512 4 16 6" "C4 1024 16 32 3"; do
  set -- $spec
  for edge in 0 1; do
    tag=$([ $edge = 1 ] && echo edge || echo chain)
    export MITDVP_EDGE_APPLY=$edge
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf_$1_$tag -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/fetch_$1_$tag.out 2>&1
    find /tmp/tf_$1_$tag -name "*counter_collection.csv" -exec cp {} $OUT/fetch_$1_$tag.csv \;
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tw_$1_$tag -- python3 $REPO/tools/heff_center_probe.py $1 $5 > $OUT/write_$1_$tag.out 2>&1
    find /tmp/tw_$1_$tag -name "*counter_collection.csv" -exec cp {} $OUT/write_$1_$tag.csv \;
    unset MITDVP_EDGE_APPLY
    (cd $REPO && python3 tools/heff_traffic_center.py $OUT/fetch_$1_$tag.csv $OUT/write_$1_$tag.csv $1 $2 $3 $4 $5 $tag) | cut -c1-400
    rm -f $OUT/fetch_$1_$tag.csv $OUT/write_$1_$tag.csv
  done
done
cp $REPO/profiles/r04_heff_traffic_* $OUT/ 2>/dev/null
MITDVP_BENCH_BUDGET=230 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_C4_under_rocprof.json 2> $OUT/c4_bench.err
find /tmp/p_c4 -name "*kernel_stats.csv" -exec cp {} $OUT/r04_c4_kernel_stats.csv \;
head -5 $OUT/r04_c4_kernel_stats.csv | cut -c1-200
