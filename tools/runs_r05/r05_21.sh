#!/bin/bash
# static priority for the second workgroup of a CU (tune 1 / 5) on the short-K reducing products
O=gpurun_out/r05ab
mkdir -p $O
for t in 0 1 5; do
  MITDVP_ZGEMM_TUNE=$t timeout -k 10 120 python tools/heff_per_site.py C3 40 > $O/tune_$t.txt 2>&1 || { tail $O/tune_$t.txt; exit 1; }
  echo "tune $t"; sed -n 2,3p $O/tune_$t.txt | cut -c1-120
  MITDVP_ZGEMM_TUNE=$t timeout -k 20 300 python bench.py --workload C3 --steps 30 --warmup 3 --no-cpu-baseline --secondary none > $O/c3_$t.json 2> $O/c3_$t.err || { tail -20 $O/c3_$t.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/c3_$t.json").read().strip().splitlines()[-1])
print("C3 tune $t", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
