#!/bin/bash
# Issue and wait counters of the two plane walks of an iteration without reorthogonalisation (512^3 by default).
# Run on the GPU box from the repository root.
set -e
root=$(pwd)
grid=${1:-512}
export TMPDIR=/tmp PYTHONPATH=$root
mkdir -p $root/gpurun_out/pmc_walks
cd /tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $root/gpurun_out/pmc_walks/p$i -o p -- python3 $root/scripts/gpu_none_run.py $grid 12 > $root/gpurun_out/pmc_walks/p$i.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_walks/p*/**/p_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_spmv_dict_planes" in k:
            agg[k[-34:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k:36s} {c:30s} {len(v):4d} {sorted(v)[len(v)//2]:.4g}")
PY
