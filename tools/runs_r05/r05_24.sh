#!/bin/bash
# site-sharded set-up: the L = 8 rehearsal with either set-up, then the whole C4 chain over 4 ranks sharing the GPU (set-up only)
O=gpurun_out/r05ae
mkdir -p $O
for m in replicated pipeline; do
  MITDVP_SHARD_SETUP=$m timeout -k 10 300 python tools/rehearse_sites_one_gpu.py > $O/rehearsal_$m.json 2> $O/rehearsal_$m.err || { tail $O/rehearsal_$m.err; exit 1; }
  tail -1 $O/rehearsal_$m.json
done
RS_WORLD=4 RS_L=64 RS_SETUP_ONLY=1 timeout -k 10 500 python tools/rehearse_sites_one_gpu.py > $O/c4_setup_4ranks.json 2> $O/c4_setup.err || { tail $O/c4_setup.err; exit 1; }
grep setup_s $O/c4_setup_4ranks.json
