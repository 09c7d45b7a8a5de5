"""Randomised differential run (scripts/gpu_fuzz.py) at a size that takes a few seconds: every SpMV
kernel form, the 16-column SpMM and the device COO->CRS conversion on 200 matrices of varied shape."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_matrices_through_every_kernel_form(gpu, capsys):
    spec = importlib.util.spec_from_file_location("gpu_fuzz", os.path.join(ROOT, "scripts", "gpu_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.CASES, mod.SEED0 = 200, 5000
    rc = mod.main()
    out = capsys.readouterr().out
    assert rc == 0, out[-4000:]
    assert "200 cases" in out and " 0 failures" in out


def _random_symmetric(rng, n):
    """Symmetric CRS with unsorted rows: a random pattern plus its mirror plus a diagonal."""
    import numpy as np
    import scipy.sparse as sp

    per = int(rng.integers(1, 12))
    rows = np.repeat(np.arange(n), per)
    cols = rng.integers(0, n, rows.size)
    vals = rng.standard_normal(rows.size)
    m = sp.coo_matrix((vals, (rows, cols)), shape=(n, n))
    m = (m + m.T + sp.diags(rng.standard_normal(n) * 3)).tocsr()
    # shuffle the entries inside every row: the reference's container keeps file order, not column order
    head = m.indptr.astype(np.int64)
    key = np.repeat(np.arange(n), np.diff(head)) + rng.random(m.nnz)
    order = np.argsort(key)
    return m.data[order].astype(np.float64), m.indices[order].astype(np.int64), head


@pytest.mark.parametrize("reorth,oracle_mode,tol", [("full", "full", 1e-11), ("mgs", "full", 1e-11),
                                                    ("full2", "full", 1e-11), ("none", "none", 1e-9),
                                                    ("cuda_compat", "cuda_compat", 1e-9),
                                                    ("selective", "selective", 1e-9)])
def test_random_symmetric_matrices_follow_the_oracle(gpu, reorth, oracle_mode, tol):
    """alpha and beta of short runs on 25 random symmetric matrices of awkward sizes, every
    reorthogonalisation mode against the CPU restatement of src/lanczos.c:104-191."""
    import numpy as np

    import lanczos_amd as la
    import oracle
    from matrices import rel_err

    rng = np.random.default_rng(99)
    sizes = [3, 5, 17, 63, 64, 65, 127, 129, 255, 257, 1023, 1025, 4095, 4097] + list(rng.integers(20, 30000, 11))
    for n in sizes:
        n = int(n)
        vals, cols, head = _random_symmetric(rng, n)
        a = la.CrsMatrix(vals, cols, head)
        steps = min(n - 1, 12)
        v0 = la.gaussian_fill(int(rng.integers(1, 1 << 30)), 0, n)
        o = oracle.lanczos_crs(head, cols, vals, v0, min(2, steps), steps + 1, 0.0, oracle_mode)
        with la.Engine(a, steps + 1, reorth=reorth) as e:
            e.set_start(v0)
            alpha, beta = e.run(steps)
        scale = np.abs(o.alpha).max() + np.abs(o.beta).max()
        assert np.max(np.abs(alpha - o.alpha[:steps])) < tol * scale, (reorth, n)
        assert rel_err(beta, o.beta[:steps]) < tol, (reorth, n)


def test_random_stencils_through_the_dictionary_forms(gpu):
    """Banded matrices with constant diagonals (rows repeat as patterns) at random offsets: a few within the
    window's reach, a few far ones placed around one or two planes' distance, symmetric or one-sided, even or
    odd, sizes that end inside a plane.  Whatever form the dictionary SpMV picks for them (plane walk, loads
    one step ahead with 16- or 8-byte loads, loads inside the step), its product has the bits of the row-wise
    kernel where that one applies (up to seven entries per row) and stays within rounding of the CPU
    checker's otherwise; the alpha partial is checked through one Lanczos step."""
    import numpy as np
    import scipy.sparse as sp

    import lanczos_amd as la
    import oracle

    def spmv_tolerance(a, x):
        absrow = oracle.matvec_crs(a.row_head_indexes, a.column_index, np.abs(a.values), np.abs(x))
        return (int(np.diff(a.row_head_indexes).max()) + 4) * np.finfo(float).eps * absrow + 1e-300

    rng = np.random.default_rng(int(os.environ.get("LANCZOS_FUZZ_STENCIL_SEED", "2024")))
    seen = set()
    for case in range(int(os.environ.get("LANCZOS_FUZZ_STENCILS", "40"))):
        n = int(rng.integers(30_000, 260_000))
        near = set(int(o) for o in rng.choice([-3, -2, -1, 1, 2, 3, -17, 17, -130, 130, -600, 600, -2000, 2000],
                                               size=int(rng.integers(1, 4)), replace=False))
        plane = int(rng.choice([2100, 2101, 4096, 5000, 6561, 9999, 10240]))
        kind = int(rng.integers(0, 5))
        far = {0: {-plane, plane}, 1: {-plane}, 2: {plane}, 3: {-plane - 1, -plane, plane, plane + 2},
               4: {-2 * plane, -plane, plane}}[kind]
        offsets = sorted({0} | near | far)
        values = rng.standard_normal(len(offsets))
        m = sp.diags([np.full(n - abs(o), v) for o, v in zip(offsets, values)], offsets, shape=(n, n), format="csr")
        m.sort_indices()
        a = la.CrsMatrix(m.data.astype(np.float64), m.indices.astype(np.int64), m.indptr.astype(np.int64))
        x = rng.standard_normal(n)
        with la.Engine(a, 3, spmv="dict") as e:
            seen.add(e.spmv_name[-1] if e.spmv_name[-1] in "ap" else "0")
            y = e.spmv(x)
            e.set_start(x)
            alpha, _ = e.run(1)
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        assert np.all(np.abs(y - want) <= spmv_tolerance(a, x)), (case, offsets, n)
        xn = x / np.linalg.norm(x)
        assert abs(alpha[0] - xn @ oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, xn)) < 1e-12 * np.abs(values).sum()
        if len(offsets) <= 7:
            from matrices import plain_rows
            with plain_rows(), la.Engine(a, 3, spmv="rows") as e:
                assert e.spmv_name == "rows64" and e.spmv(x).tobytes() == y.tobytes(), (case, offsets, n)
            with la.Engine(a, 3, spmv="rows") as e:      # as a plane walk over the offsets' dictionary where that applies
                seen.add(e.spmv_name)
                assert e.spmv(x).tobytes() == y.tobytes(), (case, offsets, n)
    assert {"a", "p", "rows64", "rows64p"} <= seen, seen
