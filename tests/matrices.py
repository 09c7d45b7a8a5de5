"""Test matrices shared by the CPU and GPU tests (numpy only)."""
import numpy as np


def laplacian_triplets_rowwise(n):
    """1-D Laplacian as the 1-based triplet text the reference's inputs use:
    'i i-1 -1', 'i i 2', 'i i+1 -1' row by row."""
    rows, cols, vals = [], [], []
    for i in range(n):
        if i > 0:
            rows.append(i); cols.append(i - 1); vals.append(-1.0)
        rows.append(i); cols.append(i); vals.append(2.0)
        if i < n - 1:
            rows.append(i); cols.append(i + 1); vals.append(-1.0)
    return np.array(rows, np.int32), np.array(cols, np.int32), np.array(vals)


def laplacian3d_triplets_by_column(N):
    """3-D 7-point Laplacian N^3, triplets sorted by column then row (x fastest)."""
    n = N ** 3
    idx = np.arange(n)
    x, y, z = idx % N, (idx // N) % N, idx // (N * N)
    cand = np.stack([idx - N * N, idx - N, idx - 1, idx, idx + 1, idx + N, idx + N * N], 1)
    valid = np.stack([z > 0, y > 0, x > 0, np.ones(n, bool), x < N - 1, y < N - 1, z < N - 1], 1)
    vals = np.where(np.arange(7) == 3, 6.0, -1.0)[None, :].repeat(n, 0)
    colj = idx[:, None].repeat(7, 1)
    return cand[valid].astype(np.int32), colj[valid].astype(np.int32), vals[valid]


def laplacian_eigs(d, N, count):
    """Lowest `count` eigenvalues of the d-dimensional Dirichlet Laplacian on N^d."""
    one = 2.0 - 2.0 * np.cos(np.pi * np.arange(1, N + 1) / (N + 1))
    if d == 1:
        allv = one
    elif d == 2:
        allv = (one[:, None] + one[None, :]).ravel()
    else:
        k = min(N, 12)
        allv = (one[:k, None, None] + one[None, :k, None] + one[None, None, :k]).ravel()
    return np.sort(allv)[:count]


def rel_err(a, b, floor=0.0):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor if floor > 0 else 1e-300)))


def rmat_laplacian(scale, edge_factor=16, seed=1, abcd=(0.57, 0.19, 0.19, 0.05)):
    """Graph Laplacian L = D - A of a symmetrised R-MAT graph (self-loops dropped,
    duplicate edges merged); returns scipy CSR with sorted columns."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = edge_factor * n
    a, b, c, _ = abcd
    rows = np.zeros(m, np.int64)
    cols = np.zeros(m, np.int64)
    for level in range(scale):
        r = rng.random(m)
        right = (r >= a) & (r < a + b) | (r >= a + b + c)      # quadrants b and d
        down = r >= a + b                                      # quadrants c and d
        rows |= down.astype(np.int64) << level
        cols |= right.astype(np.int64) << level
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    adj = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(n, n)).tocsr()
    adj = adj + adj.T
    adj.data[:] = 1.0
    deg = np.asarray(adj.sum(1)).ravel()
    lap = (sp.diags(deg) - adj).tocsr()
    lap.sort_indices()
    return lap


def rmat_laplacian_gpu(*args, **kwargs):
    """The R-MAT graph Laplacian drawn on the GPU (lanczos_amd/synth_gpu.py, shared with bench.py)."""
    from lanczos_amd.synth_gpu import rmat_laplacian_gpu as gen
    return gen(*args, **kwargs)


class plain_rows:
    """Environment for engines made inside the `with`: the row-wise CRS SpMV stays on the CRS arrays
    (LANCZOS_HIP_ROWS_PLANES=0, read by every engine at creation) -- the form the other SpMV forms'
    bits are compared with."""

    def __enter__(self):
        import os
        self._old = os.environ.get("LANCZOS_HIP_ROWS_PLANES")
        os.environ["LANCZOS_HIP_ROWS_PLANES"] = "0"

    def __exit__(self, *exc):
        import os
        if self._old is None:
            del os.environ["LANCZOS_HIP_ROWS_PLANES"]
        else:
            os.environ["LANCZOS_HIP_ROWS_PLANES"] = self._old
        return False
