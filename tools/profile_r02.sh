#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): kernel statistics of the C2 and C4 bench commands,
# FETCH_SIZE / WRITE_SIZE passes of the small-bond regime (separate runs, counters only with --kernel-trace).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r02prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
which="${1:-all}"
if [ "$which" = all ] || [ "$which" = c2 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c2 -- python3 $REPO/tools/small_trace.py > $OUT/c2_trace.out 2>&1
  find /tmp/p_c2 -name "*kernel_stats.csv" -exec cp {} $OUT/r02_c2_kernel_stats.csv \;
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_c2f -- python3 $REPO/tools/small_trace.py > $OUT/c2_fetch.out 2>&1
  find /tmp/p_c2f -name "*counter_collection.csv" -exec cp {} $OUT/r02_c2_pmc_fetch.csv \;
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_c2w -- python3 $REPO/tools/small_trace.py > $OUT/c2_write.out 2>&1
  find /tmp/p_c2w -name "*counter_collection.csv" -exec cp {} $OUT/r02_c2_pmc_write.csv \;
fi
if [ "$which" = all ] || [ "$which" = c4 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r02_bench_C4_under_rocprof.json 2> $OUT/c4_bench.err
  find /tmp/p_c4 -name "*kernel_stats.csv" -exec cp {} $OUT/r02_c4_kernel_stats.csv \;
fi
ls -la $OUT
