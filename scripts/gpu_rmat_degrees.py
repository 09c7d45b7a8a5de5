"""Share of the entries of an R-MAT graph Laplacian held by its K most popular columns, and by rows above a length.
python scripts/gpu_rmat_degrees.py [scale]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanczos_amd.synth_gpu import rmat_laplacian_gpu

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
values, column_index, row_head = rmat_laplacian_gpu(scale)
rp = np.asarray(row_head).astype(np.int64)
deg = np.diff(rp)
nnz = int(rp[-1])
d = np.sort(deg)[::-1]
cs = np.cumsum(d)
print(f"n={len(deg)} nnz={nnz} longest {d[0]}")
for k in (1024, 8192, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304):
    print(f"top {k:8d} columns hold {cs[k - 1] / nnz:.3f} of the entries (degree >= {d[k - 1]})")
for t in (16, 64, 256, 1024, 4096, 16384):
    sel = deg > t
    print(f"rows of more than {t:6d} entries: {int(sel.sum()):9d} rows, {deg[sel].sum() / nnz:.3f} of the entries")
