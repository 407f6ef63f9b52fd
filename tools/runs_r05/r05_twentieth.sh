#!/bin/bash
# fragment-major cores for the unguarded epilogue: tests, per-site timing, C3 / C5 / C4 (edge forced and default)
O=gpurun_out/r05aa
mkdir -p $O
timeout -k 20 400 python -m pytest tests/test_gpu_edge_apply.py tests/test_gpu_3m_numerics.py -q -m gpu -x > $O/test.txt 2>&1 || { tail -40 $O/test.txt; exit 1; }
tail -2 $O/test.txt
for t in 0 32; do
  MITDVP_TIMING_ABLATION=1 MITDVP_ZGEMM_TUNE=$t timeout -k 10 120 python tools/heff_per_site.py C3 40 > $O/tune_$t.txt 2>&1 || { tail $O/tune_$t.txt; exit 1; }
  echo "tune $t"; sed -n 1,6p $O/tune_$t.txt | cut -c1-120
done
for w in C3 C5; do
  st=30; [ $w = C5 ] && st=4
  timeout -k 20 300 python bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline --secondary none > $O/$w.json 2> $O/$w.err || { tail -20 $O/$w.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
print("$w", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
for e in 1 0; do
  MITDVP_EDGE_APPLY=$e timeout -k 20 400 python bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline --secondary none > $O/c4_$e.json 2> $O/c4_$e.err || { tail -20 $O/c4_$e.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/c4_$e.json").read().strip().splitlines()[-1])
print("C4 MITDVP_EDGE_APPLY=$e", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
