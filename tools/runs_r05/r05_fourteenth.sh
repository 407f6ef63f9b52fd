#!/bin/bash
# unguarded 4 x 4 x 4 epilogue (MITDVP_EPI_FULL): edge-form tests, then C3 / C4-edge A/B
O=gpurun_out/r05u
mkdir -p $O
timeout -k 20 600 python -m pytest tests/test_gpu_edge_apply.py tests/test_gpu_3m_numerics.py -q -m gpu -x > $O/test.txt 2>&1 || { tail -40 $O/test.txt; exit 1; }
tail -3 $O/test.txt
for f in 1 0 1 0; do
  MITDVP_EPI_FULL=$f timeout -k 20 300 python bench.py --workload C3 --steps 30 --warmup 5 --no-cpu-baseline --secondary none > $O/c3_$f.json 2> $O/c3_$f.err || { tail -20 $O/c3_$f.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/c3_$f.json").read().strip().splitlines()[-1])
print("C3 epi_full=$f", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["config"].get("stage_ms"))
P
done
for f in 1 0; do
  MITDVP_EDGE_APPLY=1 MITDVP_EPI_FULL=$f timeout -k 20 400 python bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline --secondary none > $O/c4_$f.json 2> $O/c4_$f.err || { tail -20 $O/c4_$f.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/c4_$f.json").read().strip().splitlines()[-1])
print("C4 forced edge epi_full=$f", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
