#!/bin/bash
# kernel statistics, FETCH_SIZE / WRITE_SIZE and MFMA-busy passes of the H_eff apply with the block-sparse W stage
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02prof; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/h0 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_trace.out 2>&1
find /tmp/h0 -name "*kernel_trace.csv" -exec cp {} $OUT/r02_heff_kernel_trace.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/h1 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_fetch.out 2>&1
find /tmp/h1 -name "*counter_collection.csv" -exec cp {} $OUT/r02_heff_pmc_fetch.csv \;
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/h2 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_write.out 2>&1
find /tmp/h2 -name "*counter_collection.csv" -exec cp {} $OUT/r02_heff_pmc_write.csv \;
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/h3 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_util.out 2>&1
find /tmp/h3 -name "*counter_collection.csv" -exec cp {} $OUT/r02_heff_pmc_util.csv \;
tail -n 1 $OUT/heff_trace.out; ls -la $OUT | tail -6
