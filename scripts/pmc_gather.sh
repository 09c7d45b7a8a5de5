#!/bin/bash
# L2 request counters of the gather-bound SpMV forms (run on the GPU box from the repository root):
# the tiled form on the random symmetric matrix (10M rows) and the merge form on R-MAT scale 24.
set -e
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for pass in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_rand_$tag -o p -- python3 $root/scripts/gpu_rand_spmv.py 10000000 > $root/gpurun_out/pmc_rand_$tag.log 2>&1
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_rmat_$tag -o p -- python3 $root/scripts/gpu_rmat_forms.py 24 ${RMAT_FORM:-merge} > $root/gpurun_out/pmc_rmat_$tag.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_spmv" in k or "k_merge_fixup" in k or "k_tiles" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(f, k[:60], {c: (len(v), sum(v) / len(v)) for c, v in cs.items()})
PY
