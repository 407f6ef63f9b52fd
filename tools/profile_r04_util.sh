#!/bin/bash
# round 4: matrix-pipe utilisation (its own PMC pass, kernel trace only) of the apply's launches at the C4 / C5 / C3 centre shapes
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r04util; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for spec in "C4 3 chain" "C5 6 edge" "C3 8 chain"; do
  set -- $spec
  [ $3 = edge ] && export MITDVP_EDGE_APPLY=1 || export MITDVP_EDGE_APPLY=0
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/u_$1 -- python3 $REPO/tools/heff_center_probe.py $1 $2 > $OUT/util_$1.out 2>&1
  find /tmp/u_$1 -name "*counter_collection.csv" -exec cp {} $OUT/util_$1.csv \;
  unset MITDVP_EDGE_APPLY
  (cd $REPO && python3 tools/heff_util_center.py $OUT/util_$1.csv $1 $2 $3) | cut -c1-1200
  rm -f $OUT/util_$1.csv
done
cp $REPO/profiles/r04_heff_mfma_util_* $OUT/ 2>/dev/null
