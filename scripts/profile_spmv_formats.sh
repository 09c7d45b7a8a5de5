#!/bin/bash
# rocprofv3 kernel trace of the SpMV + vector step on the three formats of the 256^3 Laplacian
# (dictionary, values from the CRS array + pattern numbers, plain CRS arrays); run on the GPU box from the repository root
set -e
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/fmt_a -o kt -- python3 $root/scripts/gpu_spmv_forms.py 256 dict rows > $root/gpurun_out/fmt_a.log 2>&1
export LANCZOS_HIP_ROWS_PLANES=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/fmt_b -o kt -- python3 $root/scripts/gpu_spmv_forms.py 256 rows > $root/gpurun_out/fmt_b.log 2>&1
cd $root
python3 - <<'PY'
import csv
print("| kernel | calls | avg us |")
print("|---|---|---|")
for d in ("fmt_a", "fmt_b"):
    for r in csv.DictReader(open(f"gpurun_out/{d}/kt_kernel_stats.csv")):
        n = r["Name"].split("(")[0].replace("void ", "")
        if "k_spmv" in n or "k_recur" in n:
            print(f"| {n} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} |")
PY
