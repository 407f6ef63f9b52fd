#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the H_eff apply at the C5 and C3 interior shapes (separate passes, kernel trace only)
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r03traffic; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for shape in "512 4 16 6" "128 32 16 20"; do
  set -- $shape
  tag=D$1_d$2_M$3
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tf_$tag -- python3 $REPO/tools/heff_fsm_probe.py $1 $2 $3 $4 > $OUT/fetch_$tag.out 2>&1
  find /tmp/tf_$tag -name "*counter_collection.csv" -exec cp {} $OUT/fetch_$tag.csv \;
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/tw_$tag -- python3 $REPO/tools/heff_fsm_probe.py $1 $2 $3 $4 > $OUT/write_$tag.out 2>&1
  find /tmp/tw_$tag -name "*counter_collection.csv" -exec cp {} $OUT/write_$tag.csv \;
  (cd $REPO && python3 tools/heff_traffic_generic.py $OUT/fetch_$tag.csv $OUT/write_$tag.csv $1 $2 $3 $4)
done
