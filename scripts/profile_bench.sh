#!/bin/bash
# rocprofv3 passes over the default bench command: kernel trace, then FETCH_SIZE and WRITE_SIZE in
# runs of their own.  Run from the repository root on the GPU box; output under gpurun_out/<tag>_*.
set -e
tag=${1:-r2}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_kt -o kt -- python3 $root/bench.py --main-only > $root/gpurun_out/${tag}_kt.json
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_fetch -o f -- python3 $root/bench.py --main-only > $root/gpurun_out/${tag}_fetch.json
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_write -o w -- python3 $root/bench.py --main-only > $root/gpurun_out/${tag}_write.json
cd $root
find gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write -name "*.csv" | head -20
