"""Full reorthogonalisation with the basis kernels on 16 KiB tiles and on groups of 128 KiB (LANCZOS_HIP_UPDATE_BIG,
LANCZOS_HIP_DOTS_BIG), alternating, on grids whose vectors are not a power of two long.
python scripts/gpu_reorth_tiles.py [grid] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 320
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
a = la.synth_laplacian(3, N)
with la.Engine(a, steps + 1, reorth="full", profile=False) as e:
    res = {}
    for rnd in range(2):
        for big in ("0", "1"):
            os.environ["LANCZOS_HIP_UPDATE_BIG"] = big
            os.environ["LANCZOS_HIP_DOTS_BIG"] = big
            e.set_start_seed(1)
            e.run(3)
            e.set_start_seed(1)
            t0 = time.perf_counter()
            al, be = e.run(steps)
            dt = time.perf_counter() - t0
            res[big] = (al, be)
            print(f"{N}^3 big={big} round {rnd}: {dt / steps * 1e3:.4f} ms per step, {steps / dt:.2f} it/s")
    print("alpha rel diff", float(np.max(np.abs(res['0'][0] - res['1'][0]) / np.abs(res['0'][0]))),
          "beta", float(np.max(np.abs(res['0'][1] - res['1'][1]) / np.abs(res['0'][1]))),
          "ritz diff", float(np.max(np.abs(la.ritz_values(*res['0'], steps, 4) - la.ritz_values(*res['1'], steps, 4)))))
