"""Selective reorthogonalisation on one GPU with the recurrence as product + kernel and as two plane walks
(LANCZOS_HIP_TWO_PASS), alternating: time per step, alpha/beta against each other, reorthogonalisation counts.
python scripts/gpu_selective_walks.py [grid] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
a = la.synth_laplacian(3, N)
with la.Engine(a, steps + 1, reorth="selective", profile=False) as e:
    res = {}
    for rnd in range(2):
        for mode in ("0", "1"):
            os.environ["LANCZOS_HIP_TWO_PASS"] = mode
            e.set_start_seed(1)
            e.run(3)
            e.set_start_seed(1)
            t0 = time.perf_counter()
            al, be = e.run(steps)
            dt = time.perf_counter() - t0
            res[mode] = (al, be, e.reorth_steps)
            print(f"{N}^3 selective two_pass={mode} round {rnd}: {dt / steps * 1e3:.4f} ms per step, {steps / dt:.1f} it/s, "
                  f"{e.reorth_steps} reorthogonalisations")
    m = min(steps, 60)
    print("alpha rel diff (first %d)" % m, float(np.max(np.abs(res['0'][0][:m] - res['1'][0][:m]) / np.abs(res['0'][0][:m]))),
          "ritz diff", float(np.max(np.abs(la.ritz_values(res['0'][0], res['0'][1], steps, 4) - la.ritz_values(res['1'][0], res['1'][1], steps, 4)))))
