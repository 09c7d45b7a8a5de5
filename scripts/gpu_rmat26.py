"""R-MAT graph Laplacian at full size (scale 26 by default: n = 67 108 864, about 2.2e9 entries).

The graph is generated on the GPU with torch (level-by-level quadrant draws, symmetrised, self-loops
dropped, duplicates merged, L = D - A), brought to the host as the int64 CRS container of the
reference (src/crs.h:4-21), and handed to the engine like any other matrix.  torch's sort takes at
most 2^31-1 keys, so the (row, column) keys are sorted and merged per row range.

Checks that need no second implementation: L * ones == 0 exactly, symmetry u.(Lv) == v.(Lu), sampled
rows against direct evaluation.  Then timings of the single-vector path (SpMV form from RMAT_SPMV, default auto) and of the
block variant (second argument: block steps, 0 skips it)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lanczos_amd as la

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
block_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
edge_factor, seed = 16, 1
pa, pb, pc = 0.57, 0.19, 0.19
dev = torch.device("cuda")


def log(msg):
    print(f"[{time.perf_counter() - T0:7.1f}s] {msg}", flush=True)


T0 = time.perf_counter()
sys.path.insert(0, "tests")
from matrices import rmat_laplacian_gpu  # noqa: E402

n = 1 << scale
h_val, h_col, h_head = rmat_laplacian_gpu(scale, edge_factor, seed, (pa, pb, pc))
nnz = h_val.shape[0]
lens_h = np.diff(h_head)
max_row, rows_over_512, isolated = int(lens_h.max()), int((lens_h > 512).sum()), int((lens_h <= 1).sum())
del lens_h
log(f"on the host: n={n} nnz={nnz}, max row {max_row}, rows over 512 entries {rows_over_512}, isolated vertices {isolated}")

a = la.CrsMatrix(h_val, h_col, h_head)
out = {"scale": scale, "n": n, "nnz": nnz, "max_row": max_row, "rows_over_512": rows_over_512,
       "isolated_vertices": isolated, "generate_s": time.perf_counter() - T0}
rng = np.random.default_rng(1)
t0 = time.perf_counter()
with la.Engine(a, 33, reorth="none", spmv=os.environ.get("RMAT_SPMV", "auto"), profile=True) as e:
    out["engine_create_s"] = time.perf_counter() - t0
    out["spmv_kernel"] = e.spmv_name
    log(f"engine up in {out['engine_create_s']:.1f}s, kernel {e.spmv_name}")
    y = e.spmv(np.ones(n))
    assert not y.any(), "L * ones must vanish exactly"
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    au, av = e.spmv(u), e.spmv(v)
    lhs, rhs = float(v @ au), float(u @ av)
    assert abs(lhs - rhs) <= 1e-11 * (np.linalg.norm(u) * np.linalg.norm(av)), (lhs, rhs)
    heavy = np.argsort(np.diff(h_head))[-5:]
    for r in np.concatenate([rng.integers(0, n, 2000), [0, 1, n - 1], heavy]):
        s = slice(h_head[r], h_head[r + 1])
        want = float(np.dot(h_val[s], u[h_col[s]]))
        scale_r = float(np.abs(h_val[s]) @ np.abs(u[h_col[s]]))
        assert abs(au[r] - want) <= 1e-13 * (1 + scale_r), (r, au[r], want)
    log("SpMV checks passed")
    e.set_start_seed(1)
    e.run(3)
    e.set_start_seed(1)
    e.profile_reset()
    t0 = time.perf_counter()
    alpha, beta = e.run(32)
    dt = time.perf_counter() - t0
    assert np.all(np.isfinite(alpha)) and np.all(beta > 0)
    p = e.profile()
    out["single_none"] = {"ms_per_step": dt / 32 * 1e3, "spmv_ms": p["spmv"]["ms"] / 32,
                          "spmv_GBps": p["spmv"]["bytes"] / (p["spmv"]["ms"] * 1e-3) / 1e9}
    log(f"single vector: {out['single_none']}")
del au, av, u, v, y

if block_steps == 0:
    print(json.dumps(out, indent=1))
    sys.exit(0)
with la.BlockEngine(a, block_steps, profile=True) as e:
    e.set_start_seed(1)
    e.run(1)
    e.set_start_seed(1)
    e.profile_reset()
    t0 = time.perf_counter()
    e.run(block_steps)
    dt = time.perf_counter() - t0
    prof = e.profile()
    out["block"] = {"steps": block_steps, "ms_per_block_step": dt / block_steps * 1e3,
                    "kernels": {k: {"ms_per_step": v["ms"] / block_steps,
                                    "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0}
                                for k, v in prof.items()},
                    "lowest4": e.ritz(4).tolist()}
    log(f"block: {out['block']['ms_per_block_step']:.1f} ms per block step")
print(json.dumps(out, indent=1))
