"""The block variant's 16-column sparse product on an R-MAT graph Laplacian: a few launches for profiler
passes, HIP-event time per product printed.  python scripts/gpu_rmat_spmm.py [scale] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la
from lanczos_amd.synth_gpu import rmat_laplacian_gpu

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
a = la.CrsMatrix(*rmat_laplacian_gpu(scale))
with la.BlockEngine(a, 2, profile=True) as e:
    e.set_start_seed(1)
    e.spmm(0, 1)
    e.profile_reset()
    for _ in range(reps):
        e.spmm(0, 1)
    p = e.profile()["spmm"]
    print(f"scale {scale}: n={a.dimension} nnz={a.length}: {p['ms'] / reps:.3f} ms per 16-column product, "
          f"{p['bytes'] / (p['ms'] * 1e-3) / 1e9:.0f} GB/s on its algorithmic bytes")
