"""A few Lanczos steps (no reorthogonalisation) on the random symmetric matrix: for rocprofv3 passes."""
import sys

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanczos_amd as la

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "auto"
a = la.synth_random_symmetric(rows, 10, 42)
with la.Engine(a, 4, reorth="none", spmv=kind, profile=True) as e:
    e.set_start_seed(1)
    e.run(10)
    p = e.profile()
    print(e.spmv_name, p["spmv"]["ms"] / 10, "ms per SpMV")
