#!/bin/bash
# where an interior edge product of C3 spends its time: whole / no epilogue products / no K loop / neither
O=gpurun_out/r05x
mkdir -p $O
for t in 0 16 32 48; do
  MITDVP_TIMING_ABLATION=1 MITDVP_ZGEMM_TUNE=$t timeout -k 10 120 python tools/heff_per_site.py C3 40 > $O/tune_$t.txt 2>&1 || { tail $O/tune_$t.txt; exit 1; }
  echo "tune $t"; sed -n 3,4p $O/tune_$t.txt | cut -c1-120
done
MITDVP_VERBOSE=1 python -c "
import ctypes as C
from pytdscf_amd import _lib
lib=_lib.load(); t=C.c_double(); print(lib.mitdvp_mfma_peak_probe(0, C.byref(t)), t.value)" > $O/peak.txt 2>&1; grep -i "probe" $O/peak.txt
