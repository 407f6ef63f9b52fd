#!/bin/bash
# C5 kernel timeline of the timed region
O=$PWD/gpurun_out/r05ac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C5 --steps 3 --warmup 1 --no-cpu-baseline --secondary none > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python tools/timeline_gaps.py $f 1 0 400 30 0.2 0.5 > $O/gaps.txt
head -50 $O/gaps.txt
rm -rf $O/tr
