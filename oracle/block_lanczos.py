"""CPU model of the block-Lanczos variant (block width 16): test infrastructure only.

The reference (src/lanczos.c) iterates on one vector and has no block form, so
nothing of the reference pins this file directly: PARITY UNPINNED for the
block-specific quantities (the blocks of T).  What ties it to the reference is
the spectrum: converged Ritz values of the block run must equal those of the
single-vector run that follows src/lanczos.c:104-191, and closed-form Laplacian
eigenvalues.  The arithmetic mirrors the GPU variant step by step: three-term
projection, one full projection against the basis, CholQR twice.
"""
from __future__ import annotations

import numpy as np

W = 16


def spmm(row_head, cols, values, x):
    """y = A x for a CRS matrix (src/crs.c:155-178 applied to every column)."""
    n = row_head.shape[0] - 1
    counts = np.diff(row_head)
    rows = np.repeat(np.arange(n), counts)
    y = np.zeros_like(x)
    np.add.at(y, rows, values[:, None] * x[cols])
    return y


def cholqr2(w):
    """-> (q, r) with w = q r, r upper triangular with positive diagonal."""
    r_total = np.eye(w.shape[1])
    for _ in range(2):
        g = w.T @ w
        g = 0.5 * (g + g.T)
        r = np.linalg.cholesky(g).T
        w = np.linalg.solve(r.T, w.T).T
        r_total = r @ r_total
    return w, r_total


def block_lanczos(row_head, cols, values, x0, steps):
    """-> (A[steps,16,16], B[steps,16,16], basis list) with full reorthogonalisation."""
    q, _ = cholqr2(np.array(x0, dtype=np.float64))
    basis = [q]
    a_blocks, b_blocks = [], []
    for j in range(steps):
        w = spmm(row_head, cols, values, basis[j])
        lo = max(j - 1, 0)
        h = [basis[t].T @ w for t in range(lo, j + 1)]
        a_blocks.append(0.5 * (h[-1] + h[-1].T))
        for t, ht in zip(range(lo, j + 1), h):
            w = w - basis[t] @ ht
        h = [basis[t].T @ w for t in range(j + 1)]
        for t, ht in enumerate(h):
            w = w - basis[t] @ ht
        q, r = cholqr2(w)
        basis.append(q)
        b_blocks.append(r)
    return np.array(a_blocks), np.array(b_blocks), basis


def assemble_t(a_blocks, b_blocks, steps):
    n = steps * W
    t = np.zeros((n, n))
    for s in range(steps):
        t[s * W:(s + 1) * W, s * W:(s + 1) * W] = a_blocks[s]
        if s + 1 < steps:
            t[(s + 1) * W:(s + 2) * W, s * W:(s + 1) * W] = b_blocks[s]
            t[s * W:(s + 1) * W, (s + 1) * W:(s + 2) * W] = b_blocks[s].T
    return t


def ritz(a_blocks, b_blocks, steps, nth):
    return np.linalg.eigvalsh(assemble_t(a_blocks, b_blocks, steps))[:nth]
