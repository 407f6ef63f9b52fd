#!/bin/bash
# A/B of GEMM-kernel builds on the H_eff apply (C4 / C5 / C3 interior shapes): tools/zgemm_variants.sh tag1 tag2 ...
# (libraries pytdscf_amd/csrc/libmitdvp_<tag>.so from `make variant`; "main" = the shipped one)
cd "$(dirname "$0")/.."
for tag in "$@"; do
  lib=pytdscf_amd/csrc/libmitdvp_$tag.so
  [ "$tag" = main ] && lib=pytdscf_amd/csrc/libmitdvp.so
  echo "== $tag"
  MITDVP_LIB=$PWD/$lib python3 tools/heff_probe.py 1024 16 32 3 || exit 1
  MITDVP_LIB=$PWD/$lib python3 tools/heff_probe.py 512 4 16 20 || exit 1
  MITDVP_LIB=$PWD/$lib python3 tools/heff_probe.py 128 32 16 50 || exit 1
done
