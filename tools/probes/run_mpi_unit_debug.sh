mkdir -p gpurun_out/r05ad
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 WORLD_SIZE=2 MITDVP_DIST_BACKEND=gloo
RANK=1 LOCAL_RANK=0 python tools/probes/mpi_unit_debug.py > gpurun_out/r05ad/dbg2_r1.txt 2>&1 &
RANK=0 LOCAL_RANK=0 timeout 200 python tools/probes/mpi_unit_debug.py > gpurun_out/r05ad/dbg2.txt 2>&1
wait
grep -h "^bond" gpurun_out/r05ad/dbg2.txt; tail -3 gpurun_out/r05ad/dbg2_r1.txt
