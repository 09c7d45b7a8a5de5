"""A matrix with more than 2^31 entries (64-bit row offsets on the device): 3-D Laplacian 680^3,
n = 314 432 000, nnz = 2 198 249 600.  Checks one SpMV kernel form (argv[1]) through properties
that need no second implementation: A * ones, symmetry, and sampled rows against direct evaluation."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import lanczos_amd as la

kind = sys.argv[1] if len(sys.argv) > 1 else "auto"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 680
t0 = time.perf_counter()
a = la.synth_laplacian(3, N)
n, nnz = a.dimension, a.length
print(f"n={n} nnz={nnz} (2^31={2**31}) generated in {time.perf_counter() - t0:.1f}s", flush=True)
assert nnz >= 2 ** 31
t0 = time.perf_counter()
with la.Engine(a, 3, reorth="none", spmv=kind, profile=True) as e:
    print(f"engine up in {time.perf_counter() - t0:.1f}s, kernel {e.spmv_name}", flush=True)
    ones = np.ones(n)
    y = e.spmv(ones)
    # row sum of the Dirichlet Laplacian: 6 minus the number of neighbours inside the cube
    idx = np.arange(n)
    x_, y_, z_ = idx % N, (idx // N) % N, idx // (N * N)
    nb = (x_ > 0).astype(np.int8) + (x_ < N - 1) + (y_ > 0) + (y_ < N - 1) + (z_ > 0) + (z_ < N - 1)
    assert np.array_equal(y, 6.0 - nb), "A * ones"
    del idx, x_, y_, z_, nb
    rng = np.random.default_rng(1)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    au, av = e.spmv(u), e.spmv(v)
    lhs, rhs = float(v @ au), float(u @ av)
    assert abs(lhs - rhs) <= 1e-12 * (np.linalg.norm(u) * np.linalg.norm(av)), (lhs, rhs)
    h = a.row_head_indexes
    for r in np.concatenate([rng.integers(0, n, 2000), [0, 1, n - 1, n - 2, n // 2]]):
        s = slice(h[r], h[r + 1])
        want = float(np.dot(a.values[s], u[a.column_index[s]]))
        assert abs(au[r] - want) <= 1e-13 * (1 + abs(want)), (r, au[r], want)
    e.set_start(u)
    t0 = time.perf_counter()
    alpha, beta = e.run(20)
    dt = time.perf_counter() - t0
    assert np.all(np.isfinite(alpha)) and np.all(beta > 0)
    p = e.profile()
    print(f"{kind}: ok; 20 steps {dt * 1e3 / 20:.2f} ms/step, spmv {p['spmv']['ms'] / max(p['spmv']['launches'], 1):.2f} ms",
          flush=True)
