#!/bin/bash
# sweep of the two-pass SpMV probe over panel/block sizes; stops at the first failure
set -e
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for cfg in "18 14" "18 13" "18 12" "14 14"; do
  set -- $cfg
  for m in "rmat 24 merge" "rand 10000000 panel"; do
    echo "== PSH=$1 BSH=$2 $m"
    PSH=$1 BSH=$2 timeout -k 10 300 python -u scripts/probe/bin_spmv_probe.py $m 2>&1 | grep -E "^(build|panel|pass|engine|gather)"
  done
done
