#!/bin/bash
# the whole GPU suite; a run that writes nothing (suite.txt, the bench tests' progress file) for 330 s gets its stacks dumped
# (Python: SIGUSR1 -> real stderr; native: rocgdb attach) and its exact PID killed
O=$PWD/gpurun_out/r05suite
mkdir -p $O
rm -f $O/suite.txt $O/stderr.txt
python -m pytest tests -q -m gpu -o faulthandler_timeout=900 > $O/suite.txt 2> $O/stderr.txt &
pid=$!
quiet=0
while kill -0 $pid 2>/dev/null; do
  sleep 10
  now=$(date +%s)
  m1=$(stat -c %Y $O/suite.txt 2>/dev/null || echo 0); m2=$(stat -c %Y gpurun_out/bench_contract_progress.txt 2>/dev/null || echo 0)
  m=$(( m1 > m2 ? m1 : m2 ))
  quiet=$(( now - m ))
  if [ $quiet -gt 330 ]; then
    echo "no progress for ${quiet}s: dumping stacks of $pid" | tee -a $O/hang.txt
    kill -USR1 $pid; sleep 3
    timeout -k 5 90 rocgdb -p $pid -batch -ex "thread apply all bt 40" > $O/native_stacks.txt 2>&1
    kill $pid; sleep 3; kill -9 $pid 2>/dev/null
    tail -c 3000 $O/stderr.txt; tail -40 $O/native_stacks.txt; tail -c 600 $O/suite.txt
    exit 1
  fi
done
wait $pid; rc=$?
tail -40 $O/suite.txt
exit $rc
