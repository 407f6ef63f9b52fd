#!/bin/bash
# the driver's command, then smoke()
O=gpurun_out/r05drv
mkdir -p $O
timeout -k 20 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_C4_driver_cmd.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<P
import json
r=json.loads(open("$O/r05_bench_C4_driver_cmd.json").read().strip().splitlines()[-1])
print(r["value"], r["ms_per_step"], r["steps"], r["roofline"]["frac"], r["roofline"]["apply_form"], r["roofline"].get("traffic"), r["cpu_baseline"])
print(json.dumps(r["config"].get("secondary_summary")))
print(len(open("$O/r05_bench_C4_driver_cmd.json").read()))
P
timeout -k 20 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
