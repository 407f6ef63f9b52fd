#!/bin/bash
# epilogue-only timing (tune 32) without the core loads (64), without the tile reads (128), without both
O=gpurun_out/r05z
mkdir -p $O
for t in 32 96 160 224; do
  MITDVP_TIMING_ABLATION=1 MITDVP_ZGEMM_TUNE=$t timeout -k 10 120 python tools/heff_per_site.py C3 40 > $O/tune_$t.txt 2>&1 || { tail $O/tune_$t.txt; exit 1; }
  echo "tune $t"; sed -n 3,3p $O/tune_$t.txt | cut -c1-120
done
