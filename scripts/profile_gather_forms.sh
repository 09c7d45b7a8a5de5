#!/bin/bash
# rocprofv3 kernel trace of the gather-bound SpMV forms: the tiled kernels on the random symmetric
# matrix (10M rows) and on the R-MAT scale-24 graph Laplacian.  Run from the repository root on the
# GPU box; the per-kernel statistics land under gpurun_out/<tag>_gather_*.
set -e
tag=${1:-r2}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_gather_rand -o kt -- python3 $root/scripts/gpu_rand_spmv.py 10000000 > $root/gpurun_out/${tag}_gather_rand.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_gather_rmat -o kt -- python3 $root/scripts/gpu_rmat_forms.py 24 auto > $root/gpurun_out/${tag}_gather_rmat.log 2>&1
cd $root
for d in rand rmat; do
  f=$(find gpurun_out/${tag}_gather_$d -name "*kernel_stats.csv" | head -1)
  echo "== $d: $f"
  head -12 "$f"
done
