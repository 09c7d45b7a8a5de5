#!/bin/bash
# Where the row-wise kernel of the block variant's sparse product (rows of at most 16 entries on a power-law graph) spends
# its time: SQ counters in one pass, L1 / TA in another.  Run on the GPU box from the repository root.
set -e
root=$(pwd)
scale=${1:-24}
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_spmm_short
mkdir -p $out
cd /tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $out/sq -o p -- python3 $root/scripts/gpu_rmat_spmm.py $scale 3 > $out/sq.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $out/tcp -o p -- python3 $root/scripts/gpu_rmat_spmm.py $scale 3 > $out/tcp.log 2>&1
cd $root
python3 - <<'PY' | tee gpurun_out/pmc_spmm_short/summary.txt
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_spmm_short/*/**/p_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "k_spmm16<" in k or k.endswith("k_spmm16"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "k_spmm16" in k and not agg[k]:
            pass
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k[-30:]:30s} {c:32s} {len(v):3d} {sorted(v)[len(v)//2]:.5g}  (VGPRs {0})")
PY
grep -h "k_spmm16<" $out/sq/*/p_counter_collection.csv 2>/dev/null | head -1 | cut -d, -f10-15 || true
