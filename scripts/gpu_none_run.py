"""A short run without reorthogonalisation on the 3-D Laplacian (SpMV + deferred-normalisation recurrence per step), for
profiler passes.  python scripts/gpu_none_run.py [grid] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
a = la.synth_laplacian(3, N)
with la.Engine(a, 4, reorth="none") as e:
    e.set_start_seed(1)
    e.run(steps)
    print(N, e.spmv_name, "done")
