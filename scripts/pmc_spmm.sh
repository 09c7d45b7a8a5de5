#!/bin/bash
# L2 / fabric counters of the block variant's sparse product on an R-MAT graph (run on the GPU box from the
# repository root): how many of the 128-byte gathers of X the L2s serve, and what crosses the fabric.
set -e
root=$(pwd)
scale=${1:-24}
export TMPDIR=/tmp
mkdir -p $root/gpurun_out/pmc_spmm
cd /tmp
for pass in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_spmm/$tag -o p -- python3 $root/scripts/gpu_rmat_spmm.py $scale 3 > $root/gpurun_out/pmc_spmm/$tag.log 2>&1
done
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/pmc_spmm/trace -o p -- python3 $root/scripts/gpu_rmat_spmm.py $scale 3 > $root/gpurun_out/pmc_spmm/trace.log 2>&1
cd $root
python3 - <<'PY' | tee gpurun_out/pmc_spmm/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_spmm/*/**/p_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "spmm" in k or "k_hub" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k[-40:]:40s} {c:32s} {len(v):4d} {sum(v) / len(v):.5g}")
for f in sorted(glob.glob("gpurun_out/pmc_spmm/trace/**/p_kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "spmm" in r["Name"] or "k_hub" in r["Name"]:
            print(r["Name"][:70], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
