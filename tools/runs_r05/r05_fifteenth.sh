#!/bin/bash
# identity-check records through the pinned mirror: tests, then C3 / C5 / C2
O=gpurun_out/r05v
mkdir -p $O
timeout -k 20 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_apply.py -q -m gpu -x > $O/test.txt 2>&1 || { tail -40 $O/test.txt; exit 1; }
tail -3 $O/test.txt
for w in C3 C5 C2; do
  st=30; [ $w = C5 ] && st=4; [ $w = C2 ] && st=60
  timeout -k 20 300 python bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline --secondary none > $O/$w.json 2> $O/$w.err || { tail -20 $O/$w.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
print("$w", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
