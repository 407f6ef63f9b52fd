#!/bin/bash
O=gpurun_out/r05w
mkdir -p $O
MITDVP_EDGE_TRACE=1 timeout -k 20 300 python bench.py --workload C3 --steps 4 --warmup 2 --no-cpu-baseline --secondary none > $O/trace.json 2> $O/trace.err
grep -c "kept" $O/trace.err; grep -c "rebuilt" $O/trace.err
for w in C3 C5; do
  st=30; [ $w = C5 ] && st=4
  timeout -k 20 300 python bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline --secondary none > $O/$w.json 2> $O/$w.err || { tail -20 $O/$w.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
print("$w", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
