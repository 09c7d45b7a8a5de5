#!/bin/bash
# rocprofv3 kernel statistics of the iteration without reorthogonalisation as two plane walks and as product +
# recurrence kernel, on 256^3 and 512^3 (60 steps each).  Run on the GPU box from the repository root.
set -e
root=$(pwd)
export TMPDIR=/tmp PYTHONPATH=$root
mkdir -p $root/gpurun_out/walks_prof
cd /tmp
for grid in 256 512; do
  for tp in 1 0; do
    export LANCZOS_HIP_TWO_PASS=$tp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/walks_prof/g${grid}_tp${tp} -o s -- python3 $root/scripts/gpu_none_run.py $grid 60 > $root/gpurun_out/walks_prof/g${grid}_tp${tp}.log 2>&1
  done
done
cd $root
python3 - <<'PY'
import csv, glob
print("| grid | form | kernel | calls | avg us |")
print("|---|---|---|---|---|")
for grid in (256, 512):
    for tp in (1, 0):
        f = glob.glob(f"gpurun_out/walks_prof/g{grid}_tp{tp}/**/s_kernel_stats.csv", recursive=True)[0]
        for r in csv.DictReader(open(f)):
            k = r["Name"].split("(")[0].replace("void ", "").replace("lz::", "")
            if ("k_spmv_dict_planes" in k or "k_recur_lazy" in k) and int(r["Calls"]) >= 50:
                print(f"| {grid}^3 | {'two walks' if tp else 'product + recurrence'} | `{k}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} |")
PY
