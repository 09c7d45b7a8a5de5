#!/bin/bash
# Per-launch durations of the block variant's tall-skinny kernels on the 256^3 Laplacian (rocprofv3 kernel trace).
# Run on the GPU box from the repository root.
set -e
root=$(pwd)
export TMPDIR=/tmp PYTHONPATH=$root
mkdir -p $root/gpurun_out/blkprof
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/blkprof -o t -- python3 $root/scripts/gpu_block_bench.py 256 8 > $root/gpurun_out/blkprof/run.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/blkprof/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    k = r["Kernel_Name"]
    if "k_bupdate16" in k or "k_bscale16" in k or "k_gram16" in k or "k_spmm16" in k:
        print(k.split("(")[0][-14:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
