#!/bin/bash
# Configuration 1 through the command-line drop-in: a 2-D Laplacian of 100 x 100 points in the reference's
# triplet format, 100 iterations, the [INIT] / [TOTAL] lines of three runs in a row (process start-up included).
set -e
mkdir -p gpurun_out/c1
python3 - <<'PY'
n = 100
with open("/tmp/lap2d_100.txt", "w") as f:
    for i in range(n):
        for j in range(n):
            r = i * n + j + 1
            f.write(f"{r} {r} 4.0\n")
            if j + 1 < n: f.write(f"{r} {r + 1} -1.0\n")
            if i + 1 < n: f.write(f"{r} {r + n} -1.0\n")
PY
for run in 1 2 3; do
  s=$(date +%s.%N)
  LANCZOS_HIP_INIT_TRACE=1 LANCZOS_HIP_JSON=1 timeout -k 10 120 lanczos_amd/bin/lanczos_main /tmp/lap2d_100.txt crs_hip 4 101 > gpurun_out/c1/out_$run.txt 2> gpurun_out/c1/err_$run.txt
  e=$(date +%s.%N)
  echo "run $run: process wall $(python3 -c "print(round($e-$s,4))") s"
  grep -E "INIT|TOTAL|TIME" gpurun_out/c1/err_$run.txt
  tail -2 gpurun_out/c1/out_$run.txt
done
