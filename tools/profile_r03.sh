#!/bin/bash
# round-3 evidence: C2 / C3 / C5 bench lines with the final kernels, kernel statistics of the C4 driver command under rocprofv3
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r03prof; mkdir -p $OUT
for w in C2 C3 C5; do
  python3 bench.py --workload $w --steps 20 --warmup 5 > $OUT/r03_bench_$w.json 2> $OUT/r03_bench_$w.err
  tail -c 300 $OUT/r03_bench_$w.json; echo
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_C4_under_rocprof.json 2> $OUT/c4_bench.err
find /tmp/p_c4 -name "*kernel_stats.csv" -exec cp {} $OUT/r03_c4_kernel_stats.csv \;
head -5 $OUT/r03_c4_kernel_stats.csv | cut -c1-200
