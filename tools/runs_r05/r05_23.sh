#!/bin/bash
O=gpurun_out/r05ad
mkdir -p $O
timeout -k 20 900 python -m pytest tests/test_gpu_site_sharding.py tests/test_gpu_multirank.py -q -m gpu -x > $O/test.txt 2>&1 || { tail -60 $O/test.txt; exit 1; }
tail -3 $O/test.txt
