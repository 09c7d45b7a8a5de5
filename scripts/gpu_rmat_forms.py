"""SpMV time of chosen forms on an R-MAT graph Laplacian generated on the GPU: gpu_rmat_forms.py SCALE form [form ...]"""
import sys
import time

import numpy as np

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lanczos_amd as la
from matrices import rmat_laplacian_gpu

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t0 = time.perf_counter()
a = la.CrsMatrix(*rmat_laplacian_gpu(scale))
print(f"scale {scale}: n={a.dimension} nnz={a.length} max row {int(np.diff(a.row_head_indexes).max())} "
      f"({time.perf_counter() - t0:.1f}s)", flush=True)
x = np.random.default_rng(0).standard_normal(a.dimension)
ref = None
for kind in sys.argv[2:] or ["auto"]:
    t0 = time.perf_counter()
    with la.Engine(a, 4, reorth="none", spmv=kind, profile=True) as e:
        up = time.perf_counter() - t0
        y = e.spmv(x)
        if ref is None:
            ref = y
        err = float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))
        e.set_start_seed(1)
        e.run(3)
        e.profile_reset()
        e.run(10)
        p = e.profile()
        ms = p["spmv"]["ms"] / 10
        print(f"{e.spmv_name:14s} {ms:8.3f} ms per SpMV  {p['spmv']['bytes'] / 10 / ms / 1e6:7.0f} GB/s  "
              f"{a.length / ms / 1e6:6.1f} G entries/s  create {up:.1f}s  max rel diff vs first form {err:.1e}", flush=True)
