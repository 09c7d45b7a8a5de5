"""SpMV time of chosen forms on the 3-D Laplacian N^3 (run on the GPU box): gpu_spmv_forms.py N form [form ...]"""
import sys

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
a = la.synth_laplacian(3, N)
for kind in sys.argv[2:] or ["rows"]:
    with la.Engine(a, 4, reorth="none", spmv=kind, profile=True) as e:
        e.set_start_seed(1)
        e.run(10)
        e.profile_reset()
        e.run(50)
        p = e.profile()
        ms = p["spmv"]["ms"] / 50
        print(f"{e.spmv_name:12s} {ms:.4f} ms per SpMV  {p['spmv']['bytes'] / 50 / ms / 1e6:.0f} GB/s  "
              f"recur {p['recur']['ms'] / 50:.4f} ms", flush=True)
