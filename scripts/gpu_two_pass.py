"""The iteration without reorthogonalisation as two plane walks (LANCZOS_HIP_TWO_PASS) against the product followed by
the recurrence kernel: alpha/beta of both against each other, time per step of both, alternating.
python scripts/gpu_two_pass.py [grid] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
a = la.synth_laplacian(3, N)
n = a.dimension
with la.Engine(a, 4, reorth="none", profile=False) as e:
    print(N, e.spmv_name)
    res = {}
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["LANCZOS_HIP_TWO_PASS"] = mode
            e.set_start_seed(1)
            e.run(10)
            e.set_start_seed(1)
            t0 = time.perf_counter()
            al, be = e.run(steps)
            dt = time.perf_counter() - t0
            res[mode] = (al, be)
            moved = (50.0 if mode == "0" else 36.0) * n
            print(f"two_pass={mode} round {rnd}: {dt / steps * 1e3:.4f} ms per step, {steps / dt:.1f} it/s, "
                  f"{moved / (dt / steps) / 1e12:.3f} TB/s on the bytes it moves")
    a0, b0 = res["0"]
    a1, b1 = res["1"]
    m = min(steps, 60)
    print("alpha rel diff (first %d)" % m, float(np.max(np.abs(a0[:m] - a1[:m]) / np.abs(a0[:m]))),
          "beta", float(np.max(np.abs(b0[:m] - b1[:m]) / np.abs(b0[:m]))))
