"""Does renumbering the vertices of a power-law graph by falling degree pay for the SpMV?
R-MAT Laplacian generated on the GPU (as scripts/gpu_rmat26.py, nnz < 2^31 only), once in its
natural numbering and once renumbered; both through the same engine."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import lanczos_amd as la

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
edge_factor, seed = 16, 1
pa, pb, pc = 0.57, 0.19, 0.19
dev = torch.device("cuda")
n = 1 << scale
m = edge_factor * n
g = torch.Generator(device=dev)
g.manual_seed(seed)
rows = torch.zeros(m, dtype=torch.int64, device=dev)
cols = torch.zeros(m, dtype=torch.int64, device=dev)
for level in range(scale):
    r = torch.rand(m, generator=g, device=dev, dtype=torch.float32)
    down = r >= pa + pb
    right = ((r >= pa) & (r < pa + pb)) | (r >= pa + pb + pc)
    rows |= down.to(torch.int64) << level
    cols |= right.to(torch.int64) << level
    del r, down, right
keep = rows != cols
rows, cols = rows[keep], cols[keep]
diag = torch.arange(n, dtype=torch.int64, device=dev)
keys = torch.unique(torch.cat([(rows << 32) | cols, (cols << 32) | rows, (diag << 32) | diag]))
del rows, cols, keep


def to_host_crs(keys):
    bounds = torch.arange(n + 1, dtype=torch.int64, device=dev) << 32
    head = torch.searchsorted(keys, bounds)
    lens = head[1:] - head[:-1]
    col, row = keys & 0xFFFFFFFF, keys >> 32
    val = torch.where(col == row, (lens - 1).to(torch.float64)[row],
                      torch.tensor(-1.0, dtype=torch.float64, device=dev))
    a = la.CrsMatrix(val.cpu().numpy(), col.cpu().numpy(), head.cpu().numpy())
    return a, lens


def time_engine(a, tag, spmv="auto"):
    with la.Engine(a, 33, reorth="none", spmv=spmv, profile=True) as e:
        e.set_start_seed(1)
        e.run(3)
        e.set_start_seed(1)
        e.profile_reset()
        alpha, beta = e.run(32)
        p = e.profile()
        return {"kernel": e.spmv_name, "spmv_ms": p["spmv"]["ms"] / 32,
                "lowest4": la.ritz_values(alpha, beta, 32, 4).tolist()}


out = {"scale": scale, "n": n, "nnz": int(keys.numel())}
a, lens = to_host_crs(keys)
out["natural"] = time_engine(a, "natural")
del a
order = torch.argsort(lens, descending=True, stable=True)      # new -> old
label = torch.empty_like(order)
label[order] = torch.arange(n, dtype=torch.int64, device=dev)  # old -> new
keys2 = torch.sort((label[keys >> 32] << 32) | label[keys & 0xFFFFFFFF]).values
del keys
a2, _ = to_host_crs(keys2)
out["by_degree"] = time_engine(a2, "by_degree")
for kind in ("pipe", "vector"):
    try:
        out["by_degree_" + kind] = time_engine(a2, kind, spmv=kind)
    except Exception as ex:  # a kernel form may decline the matrix
        out["by_degree_" + kind] = str(ex)
print(json.dumps(out, indent=1))
