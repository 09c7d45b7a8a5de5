"""The regrouped 16-column product with its stream loads / partial-row stores marked non-temporal or not
(LANCZOS_HIP_SPMM_NT, read at every product): HIP-event time per product, alternating, on one R-MAT graph.
python scripts/gpu_spmm_nt.py [scale]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la
from lanczos_amd.synth_gpu import rmat_laplacian_gpu

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
a = la.CrsMatrix(*rmat_laplacian_gpu(scale))
with la.BlockEngine(a, 2, profile=True) as e:
    e.set_start_seed(1)
    e.spmm(0, 1)
    for rnd in range(3):
        for nt in ("0", "1", "2", "3"):
            os.environ["LANCZOS_HIP_SPMM_NT"] = nt
            e.spmm(0, 1)
            e.profile_reset()
            for _ in range(4):
                e.spmm(0, 1)
            print(f"scale {scale} nt={nt}: {e.profile()['spmm']['ms'] / 4:.3f} ms per product", flush=True)
