#!/bin/bash
# Fabric requests of the block variant's tall-skinny kernels on the 256^3 Laplacian, per launch.
# Run on the GPU box from the repository root.
set -e
root=$(pwd)
export TMPDIR=/tmp PYTHONPATH=$root
out=$root/gpurun_out/pmc_block
mkdir -p $out
cd /tmp
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/ea -o p -- python3 $root/scripts/gpu_block_bench.py 256 8 > $out/ea.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum --output-format csv -d $out/tcc -o p -- python3 $root/scripts/gpu_block_bench.py 256 8 > $out/tcc.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_block/*/**/p_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if any(x in k for x in ("k_bupdate16", "k_bscale16", "k_gram16")):
            agg[k[-14:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            print(f"{k:16s} {c:26s} n={len(v):3d} first={v[0]:.4g} last={v[-1]:.4g} median={sorted(v)[len(v)//2]:.4g}")
PY
