"""The iteration as two plane walks against product + recurrence kernel over many grid sizes (lines that end inside, at the
end of and across wavefronts and steps), both spans, without reorthogonalisation and with the selective rule; the
reorthogonalisation kernels on groups of tiles against single tiles on the same grids.  Prints the worst differences;
exit status 1 beyond 1e-11.  python scripts/gpu_walks_sweep.py [first] [last] [stride]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 112
st = int(sys.argv[3]) if len(sys.argv) > 3 else 3
worst = 0.0
walked = 0


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


for N in range(lo, hi + 1, st):
    a = la.synth_laplacian(3, N)
    v0 = la.gaussian_fill(N, 0, a.dimension)
    for span in ("1024", "2048"):
        os.environ["LANCZOS_HIP_DICT_SPAN"] = span
        for reorth in ("none", "selective"):
            out = {}
            for mode in ("0", "1"):
                os.environ["LANCZOS_HIP_TWO_PASS"] = mode
                with la.Engine(a, 41, reorth=reorth) as e:
                    name = e.spmv_name
                    e.set_start(v0)
                    out[mode] = e.run(40)
            d = max(rel(out["1"][0], out["0"][0]), rel(out["1"][1], out["0"][1]))
            worst = max(worst, d)
            walked += name.endswith("p")
            if d > 1e-11:
                print("DIFF", N, span, reorth, name, d)
    os.environ.pop("LANCZOS_HIP_TWO_PASS", None)
    out = {}
    for big in ("0", "1"):
        os.environ["LANCZOS_HIP_UPDATE_BIG"] = big
        os.environ["LANCZOS_HIP_DOTS_BIG"] = big
        with la.Engine(a, 61, reorth="full") as e:
            e.set_start(v0)
            out[big] = e.run(60)
    os.environ.pop("LANCZOS_HIP_UPDATE_BIG", None)
    os.environ.pop("LANCZOS_HIP_DOTS_BIG", None)
    d = max(rel(out["1"][0], out["0"][0]), rel(out["1"][1], out["0"][1]))
    worst = max(worst, d)
    if d > 1e-11:
        print("DIFF tiles", N, d)
    print(N, name, "ok", flush=True)
print(f"grids {lo}..{hi}: worst relative difference {worst:.2e}, {walked} runs on the plane-walking form")
sys.exit(1 if worst > 1e-11 else 0)
