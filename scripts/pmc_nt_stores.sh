#!/bin/bash
# Fabric requests of the SpMV and the recurrence with the recurrence's vector written cacheably or past the caches
# (LANCZOS_HIP_NT_STORES = 0 / 1), 256^3 and 512^3; run on the GPU box from the repository root.
set -e
root=$(pwd)
export TMPDIR=/tmp
out=$root/gpurun_out/pmc_nt
mkdir -p $out
cd /tmp
for N in 256 512; do
  for nt in 0 1; do
    export LANCZOS_HIP_NT_STORES=$nt
    timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/c_${N}_$nt -o p -- python3 $root/scripts/gpu_none_run.py $N 30 > $out/c_${N}_$nt.log 2>&1
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_${N}_$nt -o p -- python3 $root/scripts/gpu_none_run.py $N 30 > $out/t_${N}_$nt.log 2>&1
  done
done
cd $root
python3 - <<'PY' | tee gpurun_out/pmc_nt/summary.txt
import csv, glob, collections
for N in (256, 512):
    for nt in (0, 1):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"gpurun_out/pmc_nt/c_{N}_{nt}/**/p_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if "k_spmv" in k or "k_recur_lazy" in k:
                    agg[k.split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        times = {}
        for f in glob.glob(f"gpurun_out/pmc_nt/t_{N}_{nt}/**/p_kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                for key in ("k_spmv", "k_recur_lazy"):
                    if key in r["Name"]:
                        times[key] = float(r["AverageNs"]) / 1e3
        for k, cs in sorted(agg.items()):
            med = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
            t = next((v for key, v in times.items() if key in k), float("nan"))
            print(f"{N}^3 nt={nt} {k[-28:]:28s} {t:8.1f} us  fabric reads {med.get('TCC_EA0_RDREQ_sum', 0) * 128 / 1e6:8.1f} MB  "
                  f"writes {med.get('TCC_EA0_WRREQ_sum', 0) * 64 / 1e6:8.1f} MB  L2 hit {med.get('TCC_HIT_sum', 0):.3g} miss {med.get('TCC_MISS_sum', 0):.3g}")
PY
