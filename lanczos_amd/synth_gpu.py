"""Synthetic matrices that are drawn on the GPU because numpy cannot reach their size in seconds.

Workload generators for the benchmark and the full-size tests; not part of the solver (the host
generators of the grid Laplacians and the random symmetric matrix live in host/synth.c).
"""
import numpy as np  # noqa: F401  (the arrays returned are numpy arrays)


def rmat_laplacian_gpu(scale, edge_factor=16, seed=1, abc=(0.57, 0.19, 0.19)):
    """Graph Laplacian L = D - A of a symmetrised R-MAT graph at sizes numpy cannot reach in seconds
    (scale 26: n = 67 108 864, about 2.2e9 entries): drawn, symmetrised, merged and laid out as CRS on
    the GPU with torch, then brought to the host.  torch's sort takes at most 2^31-1 keys, so the
    (row, column) keys are sorted and merged per row range.  Returns (values, column_index,
    row_head_indexes) as numpy arrays in the reference's Mat_Crs layout (src/crs.h:4-21)."""
    import torch
    dev = torch.device("cuda")
    pa, pb, pc = abc
    n = 1 << scale
    m = edge_factor * n
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rows = torch.zeros(m, dtype=torch.int64, device=dev)
    cols = torch.zeros(m, dtype=torch.int64, device=dev)
    for level in range(scale):
        r = torch.rand(m, generator=g, device=dev, dtype=torch.float32)
        down = r >= pa + pb                                        # quadrants c and d
        right = ((r >= pa) & (r < pa + pb)) | (r >= pa + pb + pc)  # quadrants b and d
        rows |= down.to(torch.int64) << level
        cols |= right.to(torch.int64) << level
        del r, down, right
    keep = rows != cols
    rows, cols = rows[keep], cols[keep]
    del keep
    fwd = (rows << 32) | cols
    rev = (cols << 32) | rows
    del rows, cols
    parts = 16
    cut = [(n * p // parts) << 32 for p in range(parts + 1)]
    pieces = []
    for p in range(parts):
        lo, hi = cut[p], cut[p + 1]
        diag = torch.arange(n * p // parts, n * (p + 1) // parts, dtype=torch.int64, device=dev)
        k = torch.cat([fwd[(fwd >= lo) & (fwd < hi)], rev[(rev >= lo) & (rev < hi)], (diag << 32) | diag])
        assert k.numel() < 2 ** 31 - 1, "row range too large for one sort"
        pieces.append(torch.unique(k))
        del k, diag
    del fwd, rev
    keys = torch.cat(pieces)
    del pieces
    bounds = torch.arange(n + 1, dtype=torch.int64, device=dev) << 32
    row_head = torch.searchsorted(keys, bounds)
    del bounds
    lens = row_head[1:] - row_head[:-1]
    col = keys & 0xFFFFFFFF
    row = keys >> 32
    del keys
    val = torch.where(col == row, (lens - 1).to(torch.float64)[row],
                      torch.tensor(-1.0, dtype=torch.float64, device=dev))
    del row, lens
    out = val.cpu().numpy(), col.cpu().numpy(), row_head.cpu().numpy()
    del col, val, row_head
    torch.cuda.empty_cache()
    return out
