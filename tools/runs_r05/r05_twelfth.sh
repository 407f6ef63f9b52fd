#!/bin/bash
# K_eff identity-state shortcut: parity test, then C5 / C4 with and without it
O=gpurun_out/r05k
mkdir -p $O
timeout -k 20 500 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "identity" -x > $O/test.txt 2>&1 || { tail -40 $O/test.txt; exit 1; }
tail -3 $O/test.txt
for f in 1 0; do
  MITDVP_KEFF_IDENT=$f timeout -k 20 300 python bench.py --workload C5 --steps 4 --warmup 1 --no-cpu-baseline --secondary none > $O/c5_$f.json 2> $O/c5_$f.err || { tail -20 $O/c5_$f.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/c5_$f.json").read().strip().splitlines()[-1])
print("C5 keff_ident=$f", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
for f in 1 0; do
  MITDVP_KEFF_IDENT=$f timeout -k 20 400 python bench.py --workload C4 --steps 2 --warmup 1 --no-cpu-baseline --secondary none > $O/c4_$f.json 2> $O/c4_$f.err || { tail -20 $O/c4_$f.err; exit 1; }
  python - <<P
import json
r=json.loads(open("$O/c4_$f.json").read().strip().splitlines()[-1])
print("C4 keff_ident=$f", r["value"], r["ms_per_step"], r["roofline"]["frac"])
P
done
