#!/bin/bash
# Request and stall counters of the row-wise CRS SpMV on the 256^3 Laplacian: tiles staged through registers
# (LANCZOS_HIP_ROWS_DMA=0) against tiles loaded straight into LDS (=1).  Run on the GPU box from the repository
# root; one rocprofv3 run per counter pass, kernel-trace in a run of its own.
set -e
root=$(pwd)
export TMPDIR=/tmp
export LANCZOS_HIP_ROWS_PLANES=0
mkdir -p $root/gpurun_out/pmc_dma
cd /tmp
for dma in 0 1; do
  i=0
  for pass in "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
              "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    i=$((i+1))
    LANCZOS_HIP_ROWS_DMA=$dma timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_dma/d${dma}_$i -o p -- python3 $root/scripts/gpu_lap_spmv.py 256 rows > $root/gpurun_out/pmc_dma/d${dma}_$i.log 2>&1
  done
  LANCZOS_HIP_ROWS_DMA=$dma timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/pmc_dma/d${dma}_trace -o p -- python3 $root/scripts/gpu_lap_spmv.py 256 rows > $root/gpurun_out/pmc_dma/d${dma}_trace.log 2>&1
done
cd $root
python3 - <<'PY' | tee gpurun_out/pmc_dma/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_dma/d*_[0-9]/**/p_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_spmv" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{f.split('/')[2]:8s} {k[-36:]:36s} {c:32s} {len(v):4d} {sum(v) / len(v):.5g}")
for f in sorted(glob.glob("gpurun_out/pmc_dma/d*_trace/**/p_kernel_stats.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_spmv" in r["Name"]:
            print(f.split('/')[2], r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"])
PY
