#!/bin/bash
# C3 kernel timeline: busy fraction, gaps, sequence
O=$PWD/gpurun_out/r05t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload C3 --steps 8 --warmup 2 --no-cpu-baseline --secondary none > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $O/tr -name '*kernel_trace.csv' | head -1)
python tools/timeline_gaps.py $f 1 6.0 400 30 0.15 0.45 > $O/gaps.txt
head -44 $O/gaps.txt
cp $f $O/c3_kernel_trace.csv; rm -rf $O/tr
