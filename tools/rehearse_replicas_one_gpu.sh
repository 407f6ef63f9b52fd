#!/bin/bash
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29631 WORLD_SIZE=2 MITDVP_DIST_BACKEND=gloo
RANK=0 LOCAL_RANK=0 python bench.py --gpus 2 --workload C2 --steps 4 --warmup 2 --no-cpu-baseline --parallel replicas > gpurun_out/rep_r0.log 2>&1 &
P0=$!
RANK=1 LOCAL_RANK=0 python bench.py --gpus 2 --workload C2 --steps 4 --warmup 2 --no-cpu-baseline --parallel replicas > gpurun_out/rep_r1.log 2>&1 &
P1=$!
wait $P0; R0=$?; wait $P1; R1=$?
echo "exit $R0 $R1"; tail -1 gpurun_out/rep_r0.log | cut -c1-600
