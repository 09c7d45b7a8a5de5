#!/bin/bash
# The command-line driver built with the address sanitizer on the HOST code of the whole product
# (make -C lanczos_amd/csrc asan, beforehand, on the CPU) run on a GPU box over small matrices that take the
# different set-up paths: dictionary, row-wise, tiled and merge SpMV, every reorthogonalisation rule, several slabs
# (loopback transport, halo and all-gather exchange, build threads), checkpoint and resume, Matrix Market input.
# Any sanitizer report ends the run with exit code 99.  Output under gpurun_out/asan/.
set -e
root=$(pwd)
out=$root/gpurun_out/asan
mkdir -p $out
export ASAN_OPTIONS=detect_leaks=0:exitcode=99:abort_on_error=0:protect_shadow_gap=0
bin=$root/lanczos_amd/bin/lanczos_main_asan
python3 - "$out" <<'PY'
import sys
import numpy as np
out = sys.argv[1]
def write(name, rows, cols, vals):
    with open(f"{out}/{name}", "w") as f:
        for r, c, v in zip(rows, cols, vals):
            f.write(f"{r + 1} {c + 1} {float(v)!r}\n")
def lap(d, N):
    n = N ** d
    idx = np.arange(n)
    rows, cols, vals = [idx], [idx], [np.full(n, 2.0 * d)]
    for k in range(d):
        s = N ** k
        coord = (idx // s) % N
        for sel, off in ((coord > 0, -s), (coord < N - 1, s)):
            rows.append(idx[sel]); cols.append(idx[sel] + off); vals.append(np.full(int(sel.sum()), -1.0))
    return np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
write("lap2d.txt", *lap(2, 60))
write("lap3d.txt", *lap(3, 20))
write("lap3d64.txt", *lap(3, 64))   # large enough for the plane-walking dictionary form
rng = np.random.default_rng(3)
n = 6000
r = rng.integers(0, n, 8 * n); c = rng.integers(0, n, 8 * n); v = rng.standard_normal(8 * n)
keep = r != c
r, c, v = r[keep], c[keep], v[keep]
write("rand.txt", np.concatenate([r, c, np.arange(n)]), np.concatenate([c, r, np.arange(n)]), np.concatenate([v, v, np.full(n, 30.0)]))
# a few very long rows beside many short ones: the merge form
hub = rng.integers(0, n, 3000)
hr = np.concatenate([np.zeros(3000, dtype=np.int64), hub, np.arange(n)]); hc = np.concatenate([hub, np.zeros(3000, dtype=np.int64), np.arange(n)])
write("hub.txt", hr, hc, np.concatenate([np.full(6000, 0.01), np.full(n, 5.0)]))
with open(f"{out}/sym.mtx", "w") as f:
    f.write("%%MatrixMarket matrix coordinate real symmetric\n% 1-D Laplacian\n500 500 999\n")
    for i in range(500):
        f.write(f"{i + 1} {i + 1} 2.0\n")
        if i: f.write(f"{i + 1} {i} -1.0\n")
PY
run() {  # name, then env assignments and arguments
  name=$1; shift
  echo "== $name: $*" >> $out/log.txt
  timeout -k 10 120 env "$@" >> $out/log.txt 2>&1
  echo "   rc=$?" >> $out/log.txt
}
cd $out
run dict            LANCZOS_HIP_QUIET=1 $bin lap3d.txt crs_hip 4 60
run rows            LANCZOS_HIP_QUIET=1 LANCZOS_HIP_NO_DICT=1 $bin lap3d.txt crs_hip 4 60
run coo_mode        LANCZOS_HIP_QUIET=1 $bin lap2d.txt coo 3 40
run crs_mode        LANCZOS_HIP_QUIET=1 $bin lap2d.txt crs 3 40
for mode in full2 mgs cuda_compat none selective; do
  run reorth_$mode  LANCZOS_HIP_QUIET=1 LANCZOS_HIP_REORTH=$mode $bin lap2d.txt crs_hip 3 50
done
run random          LANCZOS_HIP_QUIET=1 $bin rand.txt crs_hip 4 40
run hub_rows        LANCZOS_HIP_QUIET=1 $bin hub.txt crs_hip 4 40
run matrix_market   LANCZOS_HIP_QUIET=1 $bin sym.mtx crs_hip 2 80
run slabs3_halo     LANCZOS_HIP_QUIET=1 LANCZOS_HIP_NGPU=3 LANCZOS_HIP_MULTI_LOOPBACK=1 LANCZOS_HIP_CHECK_AFFINITY=1 $bin lap3d.txt crs_hip 4 60
run slabs4_threads  LANCZOS_HIP_QUIET=1 LANCZOS_HIP_NGPU=4 LANCZOS_HIP_MULTI_LOOPBACK=1 LANCZOS_HIP_MULTI_BUILD_THREADS=1 $bin lap3d.txt crs_hip 4 60
run slabs3_gather   LANCZOS_HIP_QUIET=1 LANCZOS_HIP_NGPU=3 LANCZOS_HIP_MULTI_LOOPBACK=1 $bin rand.txt crs_hip 4 40
run slabs2_none     LANCZOS_HIP_QUIET=1 LANCZOS_HIP_NGPU=2 LANCZOS_HIP_MULTI_LOOPBACK=1 LANCZOS_HIP_REORTH=none $bin lap3d.txt crs_hip 4 60
run slabs2_select   LANCZOS_HIP_QUIET=1 LANCZOS_HIP_NGPU=2 LANCZOS_HIP_MULTI_LOOPBACK=1 LANCZOS_HIP_REORTH=selective $bin lap3d.txt crs_hip 4 60
run one_slab_rccl   LANCZOS_HIP_QUIET=1 LANCZOS_HIP_MULTI_FORCE=1 $bin lap3d.txt crs_hip 4 40
run checkpoint      LANCZOS_HIP_QUIET=1 LANCZOS_HIP_CHECKPOINT=$out/ck.bin LANCZOS_HIP_CHECKPOINT_EVERY=10 $bin lap2d.txt crs_hip 3 35
run resume          LANCZOS_HIP_QUIET=1 LANCZOS_HIP_RESUME=$out/ck.bin $bin lap2d.txt crs_hip 3 60
run json_vectors    LANCZOS_HIP_JSON=1 $bin lap2d.txt crs_hip 3 30
# forms that only large matrices take by default, forced on a 64^3 grid: the iteration as two plane walks (one GPU, two
# slabs), the reorthogonalisation kernels on groups of tiles
run walks           LANCZOS_HIP_QUIET=1 LANCZOS_HIP_REORTH=none LANCZOS_HIP_TWO_PASS=1 $bin lap3d64.txt crs_hip 4 40
run walks_slabs2    LANCZOS_HIP_QUIET=1 LANCZOS_HIP_REORTH=none LANCZOS_HIP_TWO_PASS=1 LANCZOS_HIP_NGPU=2 LANCZOS_HIP_MULTI_LOOPBACK=1 $bin lap3d64.txt crs_hip 4 40
run tile_groups     LANCZOS_HIP_QUIET=1 LANCZOS_HIP_UPDATE_BIG=1 LANCZOS_HIP_DOTS_BIG=1 $bin lap3d64.txt crs_hip 4 60
cd $root
if grep -q "AddressSanitizer" $out/log.txt; then echo "sanitizer reports:"; grep -c "ERROR: AddressSanitizer" $out/log.txt; fi
grep "rc=" $out/log.txt | sort | uniq -c
grep -B1 "rc=[1-9]" $out/log.txt | head -40 || true
# the C ABI beyond the driver (engine handle, block variant, slabs) under the same sanitizer build
echo "== api_check" >> $out/log.txt
timeout -k 10 300 $root/lanczos_amd/bin/api_check_asan $out > $out/api_check.txt 2>&1
echo "   api_check rc=$?" | tee -a $out/log.txt
tail -40 $out/api_check.txt
# once more with the block variant's regrouped stream forced on (the small power-law matrix would not ask for it)
timeout -k 10 300 env LANCZOS_HIP_SPMM_HUB=2 $root/lanczos_amd/bin/api_check_asan $out > $out/api_check_hub.txt 2>&1
echo "   api_check (hub stream forced) rc=$?" | tee -a $out/log.txt
grep "block variant\|ok$\|failed" $out/api_check_hub.txt
rm -f $out/lap3d64.txt   # 37 MB of triplets nobody needs afterwards
