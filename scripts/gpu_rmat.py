"""Power-law graph Laplacian (R-MAT): the block variant and the single-vector path side by side."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import lanczos_amd as la
from matrices import rmat_laplacian

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t0 = time.perf_counter()
lap = rmat_laplacian(scale)
a = la.CrsMatrix(lap.data.astype(np.float64), lap.indices.astype(np.int64), lap.indptr.astype(np.int64))
n, nnz = a.dimension, a.length
lens = np.diff(a.row_head_indexes)
out = {"scale": scale, "n": n, "nnz": nnz, "max_row": int(lens.max()), "rows_over_512": int((lens > 512).sum()),
       "empty_rows": int((lens <= 1).sum()), "generate_s": time.perf_counter() - t0}
with la.BlockEngine(a, steps, profile=True) as e:
    e.set_start_seed(1)
    e.profile_reset()
    t0 = time.perf_counter()
    e.run(steps)
    dt = time.perf_counter() - t0
    prof = e.profile()
    out["block"] = {"ms_per_block_step": dt / steps * 1e3, "top_ritz_of_minus_L": None,
                    "kernels": {k: {"ms": v["ms"], "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0}
                                for k, v in prof.items()}}
    out["block"]["lowest4"] = e.ritz(4).tolist()
for reorth in ("none", "full"):
    with la.Engine(a, 16 * steps + 1, reorth=reorth, profile=True) as e:
        e.set_start_seed(1)
        e.run(3)
        e.set_start_seed(1)
        e.profile_reset()
        t0 = time.perf_counter()
        alpha, beta = e.run(16 * steps)
        dt = time.perf_counter() - t0
        prof = e.profile()
        out["single_" + reorth] = {"spmv_kernel": e.spmv_name, "ms_per_step": dt / (16 * steps) * 1e3,
                                   "spmv_ms": prof["spmv"]["ms"] / (16 * steps),
                                   "spmv_GBps": prof["spmv"]["bytes"] / (prof["spmv"]["ms"] * 1e-3) / 1e9,
                                   "lowest4": la.ritz_values(alpha, beta, 16 * steps, 4).tolist()}
print(json.dumps(out, indent=1))
