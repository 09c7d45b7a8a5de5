#!/bin/bash
# Request and issue counters of the row-wise CRS SpMV on the 256^3 Laplacian, with and without the tile loads
# that end at the group's last entry (run on the GPU box from the repository root)
set -e
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
for mask in 1 0; do
  i=0
  for pass in "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" \
              "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    LANCZOS_HIP_ROWS_MASK=$mask timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_rows_${mask}_$i -o p -- python3 $root/scripts/gpu_lap_spmv.py 256 rows > $root/gpurun_out/pmc_rows_${mask}_$i.log 2>&1
  done
done
cd $root
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_rows_*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_spmv" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{f.split('/')[1]:16s} {k[-40:]:40s} {c:32s} {len(v):4d} {sum(v) / len(v):.4g}")
PY
