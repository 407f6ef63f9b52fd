#!/bin/bash
# 2 bench ranks sharing the one GPU (gloo, host-staged collectives): functional rehearsal of the N>1 tp path
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=2 MITDVP_DIST_BACKEND=gloo
RANK=0 LOCAL_RANK=0 python bench.py --gpus 2 --workload C3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tp_r0.log 2>&1 &
P0=$!
RANK=1 LOCAL_RANK=0 python bench.py --gpus 2 --workload C3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tp_r1.log 2>&1 &
P1=$!
wait $P0; R0=$?; wait $P1; R1=$?
echo "exit $R0 $R1"; tail -1 gpurun_out/tp_r0.log | cut -c1-900; tail -2 gpurun_out/tp_r1.log | cut -c1-300
