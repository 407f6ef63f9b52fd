#!/bin/bash
O=gpurun_out/r05suite
mkdir -p $O
timeout -k 20 1150 python -m pytest tests -q -m gpu -o faulthandler_timeout=600 > $O/suite.txt 2>&1
rc=$?
tail -40 $O/suite.txt
exit $rc
