#!/bin/bash
# Issue and wait counters of the dictionary SpMV on the 256^3 Laplacian (run on the GPU box from the repository root)
set -e
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_dict_$i -o p -- python3 $root/scripts/gpu_lap_spmv.py 256 dict > $root/gpurun_out/pmc_dict_$i.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_dict_*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_spmv" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k[:50]:50s} {c:32s} {len(v):4d} {sum(v) / len(v):.4g}")
PY
