#!/bin/bash
# round 3, final GEMM kernel: kernel trace, FETCH_SIZE / WRITE_SIZE and MFMA-busy passes of the C4 interior H_eff apply (separate
# passes), then the C2 / C3 / C5 bench lines and the kernel statistics of the C4 driver command under rocprofv3
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r03final; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/h0 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_trace.out 2>&1
find /tmp/h0 -name "*kernel_trace.csv" -exec cp {} $OUT/r03_heff_kernel_trace.csv \;
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/h1 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_fetch.out 2>&1
find /tmp/h1 -name "*counter_collection.csv" -exec cp {} $OUT/r03_heff_pmc_fetch.csv \;
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/h2 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_write.out 2>&1
find /tmp/h2 -name "*counter_collection.csv" -exec cp {} $OUT/r03_heff_pmc_write.csv \;
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/h3 -- python3 $REPO/tools/heff_fsm_probe.py 1024 16 32 3 > $OUT/heff_util.out 2>&1
find /tmp/h3 -name "*counter_collection.csv" -exec cp {} $OUT/r03_heff_pmc_util.csv \;
tail -n 1 $OUT/heff_trace.out
cd $REPO
for w in C2 C3 C5; do
  python3 bench.py --workload $w --steps 20 --warmup 5 > $OUT/r03_bench_$w.json 2> $OUT/r03_bench_$w.err
  tail -c 300 $OUT/r03_bench_$w.json; echo
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c4 -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_C4_under_rocprof.json 2> $OUT/c4_bench.err
find /tmp/p_c4 -name "*kernel_stats.csv" -exec cp {} $OUT/r03_c4_kernel_stats.csv \;
head -5 $OUT/r03_c4_kernel_stats.csv | cut -c1-200
