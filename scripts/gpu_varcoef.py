"""SpMV time of the row-wise CRS form on grids with variable coefficients (values do not repeat, column offsets do):
as a plane walk over a dictionary of offsets, and on the CRS arrays as they are (LANCZOS_HIP_ROWS_PLANES=0).
gpu_varcoef.py d N [d N ...]   (run on the GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanczos_amd as la

args = [int(v) for v in sys.argv[1:]] or [3, 256, 2, 4096]
for d, N in zip(args[::2], args[1::2]):
    a = la.synth_laplacian(d, N)
    b = la.CrsMatrix(a.values * (1.0 + 0.5 * np.random.default_rng(1).random(a.length)), a.column_index, a.row_head_indexes)
    for plain in (False, True):
        if plain:
            os.environ["LANCZOS_HIP_ROWS_PLANES"] = "0"
        else:
            os.environ.pop("LANCZOS_HIP_ROWS_PLANES", None)
        with la.Engine(b, 4, reorth="none", profile=True) as e:
            e.set_start_seed(1)
            e.run(10)
            e.profile_reset()
            e.run(50)
            p = e.profile()
            ms = p["spmv"]["ms"] / 50
            crs = 12.0 * b.length + 4.0 * (b.dimension + 1) + 16.0 * b.dimension
            print(f"{d}-D N={N} {e.spmv_name:10s} {ms:.4f} ms per SpMV  {p['spmv']['bytes'] / 50 / ms / 1e6:.0f} GB/s moved  "
                  f"{crs / ms / 1e6:.0f} GB/s in CRS bytes", flush=True)
