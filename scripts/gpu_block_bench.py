"""Timing of the block variant's kernels on a 3-D Laplacian (HIP events per kernel class)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a = la.synth_laplacian(3, N)
n = a.dimension
with la.BlockEngine(a, steps, profile=True) as e:
    e.set_start_seed(1)
    e.profile_reset()
    t0 = time.perf_counter()
    e.run(steps)
    dt = time.perf_counter() - t0
    prof = e.profile()
    ritz = e.ritz(4)
out = {"n": n, "steps": steps, "ms_per_block_step": dt / steps * 1e3, "ritz": ritz.tolist(), "kernels": {}}
for k, v in prof.items():
    if v["launches"]:
        out["kernels"][k] = {"ms_total": v["ms"], "launches": v["launches"],
                             "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0}
print(json.dumps(out, indent=1))
