"""lanczos_hip_spmv_noalloc against lanczos_hip_spmv, through the C ABI.

The two entry points run the same kernels, so y must agree bit for bit; the dot product
x . y is summed from per-workgroup partials and is compared with the float64 host sum at
1e-12 relative to |x| . |y|.  Every case makes two calls in a row on one GPU, the second on
a smaller matrix, so that what the first call left in the fixed scratch array would show."""
import numpy as np
import pytest
import torch

import lanczos_amd as la
from matrices import rmat_laplacian

pytestmark = pytest.mark.gpu


def crs_from_scipy(m):
    m = m.tocsr()
    return la.CrsMatrix(m.data.astype(np.float64), m.indices.astype(np.int64), m.indptr.astype(np.int64))


def check(a, spmv, seed):
    x = np.random.default_rng(seed).standard_normal(a.dimension)
    with la.Engine(a, 3, spmv=spmv) as e:
        want = e.spmv(x)
        xd = torch.from_numpy(x).cuda()
        yd = torch.full((a.dimension,), np.nan, dtype=torch.float64, device="cuda")
        dd = torch.full((1,), np.nan, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        e.spmv_noalloc(xd.data_ptr(), yd.data_ptr(), dd.data_ptr())
        e.history(0)  # waits for the engine's stream
        got, dot = yd.cpu().numpy(), float(dd.cpu()[0])
        name = e.spmv_name
    assert np.array_equal(got, want), (name, int(np.sum(got != want)))
    assert abs(dot - float(x @ want)) <= 1e-12 * float(np.abs(x) @ np.abs(want)), (name, dot, float(x @ want))
    return name


def test_noalloc_two_calls_second_smaller(gpu):
    # merge form (the carry lists live in the scratch array): a larger graph, then a smaller one
    assert check(crs_from_scipy(rmat_laplacian(15)), "merge", 1).startswith("merge")
    assert check(crs_from_scipy(rmat_laplacian(12)), "merge", 2).startswith("merge")
    # tiled form on the same kind of graph (the sums of the pieces of split row blocks live there too)
    assert check(crs_from_scipy(rmat_laplacian(13)), "panel", 7).startswith("panel")
    assert check(crs_from_scipy(rmat_laplacian(12)), "panel", 8).startswith("panel")
    # automatic choice on stencils (dictionary form), again large then small
    assert check(la.synth_laplacian(3, 40), "auto", 3).startswith("dict")
    assert check(la.synth_laplacian(3, 11), "auto", 4).startswith("dict")
    # the other forms on a random matrix, large then small
    for kind in ("pipe", "vector", "stream", "panel", "rows"):
        for n, seed in ((60000, 5), (3000, 6)):
            m = la.synth_random_symmetric(n, 3, seed) if kind != "rows" else la.synth_laplacian(2, int(n ** 0.5))
            check(m, kind, seed)


def test_noalloc_rejects_what_the_scratch_cannot_hold(gpu):
    # 2^17 tiles of 512 entries is the limit of the merge form: one entry more is refused
    n = 1 << 16
    per_row = 1025
    head = np.arange(n + 1, dtype=np.int64) * per_row
    cols = (np.arange(n * per_row, dtype=np.int64) * 7919) % n
    a = la.CrsMatrix(np.ones(n * per_row), cols, head)
    assert a.length > 131072 * 512
    with la.Engine(a, 3, spmv="merge") as e:
        xd = torch.ones(n, dtype=torch.float64, device="cuda")
        yd = torch.zeros(n, dtype=torch.float64, device="cuda")
        with pytest.raises(la.LanczosHipError, match="at most 67108864"):
            e.spmv_noalloc(xd.data_ptr(), yd.data_ptr(), None)
        with pytest.raises(la.LanczosHipError, match="NULL"):
            e.spmv_noalloc(None, yd.data_ptr(), None)
    # tiled form: the sums of at most 32 pieces of split 8192-row blocks fit the scratch
    g = crs_from_scipy(rmat_laplacian(16))
    with la.Engine(g, 3, spmv="panel") as e:
        xd = torch.ones(g.dimension, dtype=torch.float64, device="cuda")
        yd = torch.zeros(g.dimension, dtype=torch.float64, device="cuda")
        with pytest.raises(la.LanczosHipError, match="takes at most 32"):
            e.spmv_noalloc(xd.data_ptr(), yd.data_ptr(), None)
