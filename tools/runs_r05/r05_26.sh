#!/bin/bash
# the ensemble tests alone, up to four times; a run that sits for 150 s gets its stacks dumped (Python: SIGUSR1; native:
# rocgdb attach) and its exact PID killed; no further run after that
O=$PWD/gpurun_out/r05af
mkdir -p $O
for i in 1 2 3 4; do
  python -m pytest tests/test_gpu_ensemble.py -q -m gpu -x > $O/run_$i.txt 2>&1 &
  pid=$!
  t=0
  while kill -0 $pid 2>/dev/null && [ $t -lt 150 ]; do sleep 5; t=$((t+5)); done
  if kill -0 $pid 2>/dev/null; then
    echo "run $i: still running after ${t}s: dumping stacks" | tee -a $O/progress.txt
    kill -USR1 $pid; sleep 3
    timeout -k 5 60 rocgdb -p $pid -batch -ex "thread apply all bt 30" > $O/native_stacks_$i.txt 2>&1
    tail -5 $O/native_stacks_$i.txt
    kill $pid; sleep 2; kill -9 $pid 2>/dev/null
    tail -60 $O/run_$i.txt
    exit 1
  fi
  wait $pid; rc=$?
  echo "run $i: rc=$rc after ~${t}s: $(tail -1 $O/run_$i.txt)" | tee -a $O/progress.txt
  [ $rc = 0 ] || { tail -40 $O/run_$i.txt; exit 1; }
done
