#!/bin/bash
# SpMV time on the random symmetric matrix for several panel widths and for the row-wise kernel
for bits in 16 17 18 19 20; do
  LANCZOS_HIP_PANEL_BITS=$bits timeout -k 10 200 python scripts/gpu_rand_spmv.py ${1:-10000000} panel || exit 1
done
timeout -k 10 200 python scripts/gpu_rand_spmv.py ${1:-10000000} pipe
