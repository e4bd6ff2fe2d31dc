#!/bin/bash
# Dev: time the multi-chain streaming kernel of every library in build_variants/ named on the command line (tools/exp_mc.py, 8 chains).
#   tools/exp_variants.sh <out dir> <variant> ...
out=$1; shift
mkdir -p "$out"
for v in "$@"; do
  echo "== $v" >> "$out/variants.txt"
  MAGI_HIP_LIB=build_variants/$v.so timeout -k 10 120 python tools/exp_mc.py 1024 8 2>/dev/null | grep -v "^kernel" >> "$out/variants.txt" || exit 1
done
