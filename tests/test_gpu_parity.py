"""Parity of the HIP path with the CPU restatement, through the C ABI.

Every test here runs on an MI355X (-m gpu).  Tolerances: SpMV within a few ulp of
the row's |a||x| sum; alpha/beta of the first 20 steps 1e-12 relative; Ritz values
1e-10 relative (the parity bar of the reference comparison), with an absolute
floor of 1e-10 * ||T|| where an eigenvalue may be ~0."""
import json
import os
import subprocess

import numpy as np
import pytest

import lanczos_amd as la
import oracle
from matrices import (plain_rows, laplacian3d_triplets_by_column, laplacian_eigs, laplacian_triplets_rowwise, rel_err,
                      rmat_laplacian)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPMV_KINDS = ["vector", "stream", "pipe", "pipe_nt", "merge"]


def crs_from_scipy(m):
    m = m.tocsr()
    return la.CrsMatrix(m.data.astype(np.float64), m.indices.astype(np.int64), m.indptr.astype(np.int64))


def irregular_matrix(n, seed, long_rows=()):
    """Row lengths from 0 to ~60 plus a few very long rows; unsorted columns, duplicates."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 60, n)
    lens[rng.integers(0, n, n // 10)] = 0             # empty rows
    for r, L in long_rows:
        lens[r] = L
    head = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cols = rng.integers(0, n, head[-1]).astype(np.int64)
    vals = rng.standard_normal(head[-1])
    return la.CrsMatrix(vals, cols, head)


def spmv_tolerance(a, x):
    absrow = oracle.matvec_crs(a.row_head_indexes, a.column_index, np.abs(a.values), np.abs(x))
    maxlen = int(np.diff(a.row_head_indexes).max()) if a.dimension else 1
    return (maxlen + 4) * np.finfo(float).eps * absrow + 1e-300


@pytest.mark.parametrize("kind", SPMV_KINDS + ["rows"])
def test_spmv_matches_oracle_on_laplacians(gpu, kind):
    for d, N in ((1, 1000), (1, 4097), (2, 67), (3, 20)):
        a = la.synth_laplacian(d, N)
        x = np.random.default_rng(d).standard_normal(a.dimension)
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        with la.Engine(a, 3, spmv=kind) as e:
            got = e.spmv(x)
        assert np.all(np.abs(got - want) <= spmv_tolerance(a, x)), (d, N, kind)


@pytest.mark.parametrize("kind", SPMV_KINDS)
def test_spmv_matches_oracle_on_irregular_rows(gpu, kind):
    # empty rows, rows longer than one 512-entry tile, n not a multiple of 64, trailing empty rows
    for n, seed, long_rows in ((1, 0, ()), (63, 1, ()), (65, 2, ((3, 700),)), (1000, 3, ((0, 2000), (999, 513), (500, 512))),
                               (5000, 4, ((4999, 0), (4998, 0), (77, 5000)))):
        a = irregular_matrix(n, seed, long_rows)
        x = np.random.default_rng(seed + 100).standard_normal(n)
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        with la.Engine(a, 3, spmv=kind) as e:
            got = e.spmv(x)
        assert np.all(np.abs(got - want) <= spmv_tolerance(a, x)), (n, kind)


@pytest.mark.parametrize("kind", SPMV_KINDS)
def test_power_law_graph_laplacian(gpu, kind):
    """R-MAT scale 16: hub rows of several thousand entries next to rows of one entry."""
    lap = rmat_laplacian(16)
    a = crs_from_scipy(lap)
    lens = np.diff(a.row_head_indexes)
    assert lens.max() > 2000 and lens.min() <= 2
    x = np.random.default_rng(3).standard_normal(a.dimension)
    want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
    v0 = la.gaussian_fill(4, 0, a.dimension)
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 31, 0.0)
    with la.Engine(a, 31, spmv=kind) as e:
        assert np.all(np.abs(e.spmv(x) - want) <= spmv_tolerance(a, x))
        e.set_start(v0)
        alpha, beta = e.run(30)
    assert rel_err(alpha, o.alpha) < 1e-11 and rel_err(beta, o.beta) < 1e-11
    # lambda_min of a graph Laplacian is 0: compare Ritz values against ||T||, not against themselves
    ritz = la.ritz_values(alpha, beta, 30, 4)
    tnorm = np.abs(alpha).max() + 2 * np.abs(beta).max()
    assert np.max(np.abs(ritz - o.eigenvalues)) < 1e-10 * tnorm
    if kind == "merge":        # what the engine picks by itself for such rows, and bitwise repeatable
        with la.Engine(a, 3) as e:
            assert e.spmv_name == "merge"
            y1 = e.spmv(x)
            assert e.spmv(x).tobytes() == y1.tobytes()


@pytest.mark.parametrize("kind", SPMV_KINDS + ["rows", "panel", "dict", "auto"])
def test_spmv_on_tiny_and_odd_sizes(gpu, kind):
    """n = 1, 2, 3 and sizes around the wavefront / workgroup widths, sparse and dense-ish rows."""
    rng = np.random.default_rng(17)
    for n in (1, 2, 3, 17, 63, 64, 65, 255, 256, 257, 1023, 4097):
        if kind == "dict":
            a = la.synth_laplacian(1, n)
        else:
            per = min(n, 5)
            lens = rng.integers(0, per + 1, n)
            head = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            if head[-1] == 0:
                lens[0], head = 1, np.concatenate([[0], np.cumsum(np.r_[1, lens[1:]])]).astype(np.int64)
            cols = rng.integers(0, n, head[-1]).astype(np.int64)
            a = la.CrsMatrix(rng.standard_normal(head[-1]), cols, head)
        x = rng.standard_normal(n)
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        with la.Engine(a, 3, spmv=kind) as e:
            assert np.all(np.abs(e.spmv(x) - want) <= spmv_tolerance(a, x)), (kind, n)


def test_spmv_zero_matrix_and_auto_kernel(gpu, monkeypatch):
    n = 300
    a = la.CrsMatrix(np.zeros(0), np.zeros(0, np.int64), np.zeros(n + 1, np.int64))
    with la.Engine(a, 3) as e:
        assert np.all(e.spmv(np.ones(n)) == 0.0)
    r = la.synth_random_symmetric(20000, 10, 42)       # ~21 nnz per row: 16 rows per wavefront group
    with la.Engine(r, 3) as e:
        assert e.spmv_name == "pipe16"
        x = np.random.default_rng(9).standard_normal(r.dimension)
        want = oracle.matvec_crs(r.row_head_indexes, r.column_index, r.values, x)
        assert np.all(np.abs(e.spmv(x) - want) <= spmv_tolerance(r, x))
    with la.Engine(la.synth_laplacian(3, 16), 3) as e:
        assert e.spmv_name == "dict27a"          # a cube's rows: interior, 6 faces, 12 edges, 8 corners; N^2 <= 2048: no far entries
    monkeypatch.setenv("LANCZOS_HIP_NO_DICT", "1")
    with la.Engine(la.synth_laplacian(3, 16), 3) as e:
        assert e.spmv_name == "rows64p"          # stencil: neighbouring rows hold neighbouring columns (offsets as patterns)
    monkeypatch.delenv("LANCZOS_HIP_NO_DICT")
    with la.Engine(la.synth_laplacian(3, 16), 3, spmv="pipe") as e:
        assert e.spmv_name == "pipe64"


def test_rows_spmv_on_short_irregular_rows_and_its_limit(gpu):
    # rows of 0..11 entries (more than one 8-entry chunk), unsorted columns, n not a multiple of 64
    rng = np.random.default_rng(12)
    n = 1000
    lens = rng.integers(0, 8, n)
    lens[rng.integers(0, n, 50)] = 11
    head = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    a = la.CrsMatrix(rng.standard_normal(head[-1]), rng.integers(0, n, head[-1]).astype(np.int64), head)
    assert int((head[64::64] - head[:-64:64]).max()) <= 481
    x = rng.standard_normal(n)
    want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
    with la.Engine(a, 3, spmv="rows") as e:
        assert np.all(np.abs(e.spmv(x) - want) <= spmv_tolerance(a, x))
    with la.Engine(a, 3) as e:                       # random columns: the heuristic keeps the pipe form
        assert e.spmv_name.startswith("pipe")
    big = irregular_matrix(1000, 3, ((0, 2000),))
    with pytest.raises(la.LanczosHipError, match="rows SpMV needs"):
        la.Engine(big, 3, spmv="rows")
    # groups at the limit (481 entries from any alignment of their first entry) and the array ends:
    # the kernel's tiles start on 128-byte lines and the 16-byte granules past the last entry are clamped
    for tail in (0, 1, 2, 3):
        targets = [481, 478 + tail, 481, 479, 481, 481]
        lens = []
        for t in targets:
            g = np.full(64, 7)
            extra = rng.integers(0, 64, t - 448)
            np.add.at(g, extra, 1)
            lens.append(g)
        lens = np.concatenate(lens)[:64 * len(targets) - 5 * tail]     # n not a multiple of 64
        n2 = lens.size
        head = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        b = la.CrsMatrix(rng.standard_normal(head[-1]), rng.integers(0, n2, head[-1]).astype(np.int64), head)
        x2 = rng.standard_normal(n2)
        want2 = oracle.matvec_crs(b.row_head_indexes, b.column_index, b.values, x2)
        with la.Engine(b, 3, spmv="rows") as e:
            assert np.all(np.abs(e.spmv(x2) - want2) <= spmv_tolerance(b, x2)), tail


def test_dictionary_spmv_forms_on_grids(gpu):
    """The dictionary SpMV picks its form from the grid: an even plane beyond the window's reach is walked
    plane by plane (N = 46: three columns of steps, the last of 68 rows; 64 and 96: whole steps; 100: ten
    columns, segments of planes; N = 47: an odd plane is walked in steps of the next even number of rows, the
    far entries one slot off the middle of the neighbouring windows), small grids (N^2 <= 2048: N = 12, 32, 33, 40) have no far entries at all,
    and N = 45 (a window of 1024 + 2 * 2026 doubles plus the other regions is beyond 64 KiB of LDS) keeps the
    first form.  Every form gives the bits of the row-wise kernel (entries in CRS order, src/crs.c:172-175), on
    odd vector lengths too, and the alpha partial that goes with them."""
    for N in (12, 32, 33, 40, 45, 46, 47, 64, 96, 100):
        a = la.synth_laplacian(3, N)
        rng = np.random.default_rng(N)
        x = rng.standard_normal(a.dimension)
        with la.Engine(a, 4, spmv="dict") as e:
            assert e.spmv_name == ("dict27p" if N in (46, 47, 64, 96, 100) else "dict27" if N == 45 else "dict27a"), e.spmv_name
            y = e.spmv(x)
            e.set_start(x)
            alpha, _ = e.run(2)
        with plain_rows(), la.Engine(a, 4, spmv="rows") as e:
            assert e.spmv_name == "rows64" and e.spmv(x).tobytes() == y.tobytes(), N
            e.set_start(x)
            alpha_rows, _ = e.run(2)
        # the same grid with the values out of the CRS array (the row-wise form on a dictionary of column offsets)
        with la.Engine(a, 4, spmv="rows") as e:
            # (12^3: too few rows; 32 to 45: three windows with halos of N^2 entries are beyond 64 KiB of LDS)
            assert e.spmv_name == ("rows64p" if N in (46, 47, 64, 96, 100) else "rows64"), (N, e.spmv_name)
            assert e.spmv(x).tobytes() == y.tobytes(), N
            e.set_start(x)
            alpha_p, _ = e.run(2)
        assert rel_err(alpha_p, alpha_rows) < 1e-13, N
        assert rel_err(alpha, alpha_rows) < 1e-13, N
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        assert np.all(np.abs(y - want) <= spmv_tolerance(a, x)), N
    # a 27-point stencil (nine far entries around each of -N^2 and +N^2: all inside the halos of the neighbouring
    # planes' windows), even and odd N; LANCZOS_HIP_DICT_PLANES=0 would leave it to the first form
    import scipy.sparse as sp
    for N in (48, 53):
        t = sp.diags([0.5, 2.0, -1.5], [-1, 0, 1], shape=(N, N))
        m = sp.kron(sp.kron(t, t), t).tocsr()
        m.sort_indices()
        b = la.CrsMatrix(m.data.astype(np.float64), m.indices.astype(np.int64), m.indptr.astype(np.int64))
        x = np.random.default_rng(N).standard_normal(b.dimension)
        with la.Engine(b, 3, spmv="dict") as e:
            assert e.spmv_name == "dict27p", e.spmv_name
            y = e.spmv(x)
        want = oracle.matvec_crs(b.row_head_indexes, b.column_index, b.values, x)   # (27 entries per row: no row-wise form)
        assert np.all(np.abs(y - want) <= spmv_tolerance(b, x)), N
    # a grid cut off in the middle of a plane (the last plane is partial), and one row fewer (odd length)
    a = la.synth_laplacian(3, 64)
    for n in (64 * 64 * 40 + 1500, 64 * 64 * 40 + 1501):
        h = a.row_head_indexes[:n + 1]
        keep = a.column_index[:h[-1]] < n
        head = np.concatenate(([0], np.cumsum(np.add.reduceat(keep.astype(np.int64), h[:-1]))))
        b = la.CrsMatrix(a.values[:h[-1]][keep], a.column_index[:h[-1]][keep], head)
        x = np.random.default_rng(n).standard_normal(n)
        with la.Engine(b, 3, spmv="dict") as e:
            assert e.spmv_name.endswith("p")
            y = e.spmv(x)
        with plain_rows(), la.Engine(b, 3, spmv="rows") as e:
            assert e.spmv(x).tobytes() == y.tobytes(), n
        with la.Engine(b, 3, spmv="rows") as e:
            assert e.spmv_name == "rows64p" and e.spmv(x).tobytes() == y.tobytes(), n


def test_stencil_with_variable_coefficients_walks_the_planes(gpu):
    """Rows whose column offsets repeat while their values do not (a 7-point stencil with random coefficients):
    no dictionary of (offset, value) patterns exists, the automatic choice is the row-wise form, and that one
    runs as a plane walk over a dictionary of offsets with the values read from the CRS array -- on grids in
    one and two dimensions too, whose "planes" no entry reaches across.  Bits of the plain row-wise kernel
    (src/crs.c:172-175 order), Lanczos sequences of the oracle."""
    for d, N in ((3, 64), (3, 47), (2, 300), (1, 50001)):
        a = la.synth_laplacian(d, N)
        rng = np.random.default_rng(N)
        b = la.CrsMatrix(a.values * (1.0 + 0.5 * rng.random(a.length)), a.column_index, a.row_head_indexes)
        x = rng.standard_normal(b.dimension)
        with la.Engine(b, 12) as e:
            assert e.spmv_name == "rows64p", e.spmv_name
            y = e.spmv(x)
            v0 = la.gaussian_fill(5, 0, b.dimension)
            e.set_start(v0)
            alpha, beta = e.run(10)
        with plain_rows(), la.Engine(b, 3) as e:
            assert e.spmv_name == "rows64" and e.spmv(x).tobytes() == y.tobytes(), N
        o = oracle.lanczos_crs(b.row_head_indexes, b.column_index, b.values, v0, 4, 11, 0.0)
        assert rel_err(alpha, o.alpha) < 1e-12 and rel_err(beta, o.beta) < 1e-12


def test_dictionary_spmv_on_repeating_rows(gpu):
    """One pattern number per row: the same bits as the row-wise kernel (entries in CRS order,
    src/crs.c:172-175), the Lanczos sequences of the oracle, and a clean refusal when rows do not repeat."""
    for a in (la.synth_laplacian(1, 5000), la.synth_laplacian(2, 70), la.synth_laplacian(3, 21)):
        x = np.random.default_rng(1).standard_normal(a.dimension)
        with la.Engine(a, 31, spmv="dict") as e:
            assert e.spmv_name.startswith("dict")
            y = e.spmv(x)
            v0 = la.gaussian_fill(3, 0, a.dimension)
            e.set_start(v0)
            alpha, beta = e.run(30)
        with plain_rows(), la.Engine(a, 3, spmv="rows") as e:
            assert e.spmv(x).tobytes() == y.tobytes()
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        assert np.all(np.abs(y - want) <= spmv_tolerance(a, x))
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 31, 0.0)
        assert rel_err(alpha, o.alpha) < 1e-12 and rel_err(beta, o.beta) < 1e-12
    # a stencil with a few perturbed entries and one changed row: still a dictionary, a larger one
    a = la.synth_laplacian(2, 50)
    vals = a.values.copy()
    vals[[7, 1000, 5555]] *= 1.5
    b = la.CrsMatrix(vals, a.column_index, a.row_head_indexes)
    x = np.random.default_rng(2).standard_normal(b.dimension)
    with la.Engine(b, 3, spmv="dict") as e:
        assert int(e.spmv_name[4:].rstrip("ap")) == 9 + 3
        assert np.all(np.abs(e.spmv(x) - oracle.matvec_crs(b.row_head_indexes, b.column_index, b.values, x))
                      <= spmv_tolerance(b, x))
    with pytest.raises(la.LanczosHipError, match="dictionary SpMV needs rows"):
        la.Engine(irregular_matrix(3000, 1), 3, spmv="dict")          # rows up to 60 entries
    r = la.synth_random_symmetric(100000, 3, 5)                        # short rows, 100000 distinct patterns
    with pytest.raises(la.LanczosHipError, match="dictionary SpMV needs rows"):
        la.Engine(r, 3, spmv="dict")


def test_panel_spmv_on_random_sparsity(gpu, monkeypatch):
    """Column-panel form: same products as the row-wise oracle (src/crs.c:155-178) whatever the panel
    width, the same Lanczos sequences, bitwise repeatable, dense rows included."""
    a = la.synth_random_symmetric(30000, 10, 9)
    x = np.random.default_rng(3).standard_normal(a.dimension)
    want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
    for bits in ("10", "12", "18"):
        monkeypatch.setenv("LANCZOS_HIP_PANEL_BITS", bits)
        with la.Engine(a, 21, spmv="panel") as e:
            assert e.spmv_name == "panel" + bits
            y = e.spmv(x)
            assert np.all(np.abs(y - want) <= spmv_tolerance(a, x))
            assert e.spmv(x).tobytes() == y.tobytes()
            v0 = la.gaussian_fill(5, 0, a.dimension)
            e.set_start(v0)
            alpha, beta = e.run(20)
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 21, 0.0)
        assert rel_err(alpha, o.alpha) < 1e-12 and rel_err(beta, o.beta) < 1e-12
    monkeypatch.delenv("LANCZOS_HIP_PANEL_BITS")
    # ragged rows, empty rows, duplicates
    b = irregular_matrix(20000, 8)
    xb = np.random.default_rng(4).standard_normal(b.dimension)
    monkeypatch.setenv("LANCZOS_HIP_PANEL_BITS", "11")
    with la.Engine(b, 3, spmv="panel") as e:
        yb = e.spmv(xb)
    assert np.all(np.abs(yb - oracle.matvec_crs(b.row_head_indexes, b.column_index, b.values, xb))
                  <= spmv_tolerance(b, xb))
    dense = irregular_matrix(20000, 8, long_rows=((5, 9000), (19999, 300)))
    with la.Engine(dense, 3, spmv="panel") as e:
        yd = e.spmv(xb)
    assert np.all(np.abs(yd - oracle.matvec_crs(dense.row_head_indexes, dense.column_index, dense.values, xb))
                  <= spmv_tolerance(dense, xb))
    monkeypatch.delenv("LANCZOS_HIP_PANEL_BITS")
    # chosen by itself when the rows reach across more of x than the L2s hold
    big = la.synth_random_symmetric(5_000_000, 4, 1)
    with la.Engine(big, 3, reorth="none") as e:
        assert e.spmv_name.startswith("panel")
        xg = la.gaussian_fill(7, 0, big.dimension)
        yg = e.spmv(xg)
    assert np.all(np.abs(yg - oracle.matvec_crs(big.row_head_indexes, big.column_index, big.values, xg))
                  <= spmv_tolerance(big, xg))
    with la.Engine(la.synth_laplacian(3, 64), 3) as e:
        assert e.spmv_name == "dict27p"          # planes of 4096 rows: walked plane by plane


@pytest.mark.parametrize("log2n", [24, 25])
def test_panel_spmv_far_columns(gpu, log2n):
    """Even row blocks whose chunks reach across 2^20 columns and more: a few entries per chunk keep their
    column in the side list of the packed stream (2^24 rows), or the unpacked stream is built (2^25 rows:
    more such entries than the list holds).  One random entry per row, checked against direct evaluation."""
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    cols = rng.integers(0, n, n).astype(np.int64)
    vals = rng.standard_normal(n)
    a = la.CrsMatrix(vals, cols, np.arange(n + 1, dtype=np.int64))
    x = rng.standard_normal(n)
    with la.Engine(a, 3, reorth="none", spmv="panel") as e:
        y = e.spmv(x)
        assert e.spmv(x).tobytes() == y.tobytes()
    assert np.array_equal(y, vals * x[cols])


def test_panel_spmv_on_power_law_graph(gpu, monkeypatch):
    """The tiled form on hub rows: the fullest row blocks are cut into pieces whose sums a second kernel
    adds in order; same products as the row-wise oracle (src/crs.c:155-178), same Lanczos sequences,
    bit for bit the same on every run."""
    a = crs_from_scipy(rmat_laplacian(16))
    x = np.random.default_rng(3).standard_normal(a.dimension)
    want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
    v0 = la.gaussian_fill(4, 0, a.dimension)
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 31, 0.0)
    for bits in ("10", "12", "16"):
        monkeypatch.setenv("LANCZOS_HIP_PANEL_BITS", bits)
        with la.Engine(a, 31, spmv="panel") as e:
            assert e.spmv_name == "panel" + bits
            y = e.spmv(x)
            assert np.all(np.abs(y - want) <= spmv_tolerance(a, x))
            assert e.spmv(x).tobytes() == y.tobytes()
            assert not e.spmv(np.ones(a.dimension)).any()          # L * ones vanishes exactly
            e.set_start(v0)
            alpha, beta = e.run(30)
            e.set_start(v0)
            alpha2, beta2 = e.run(30)
        assert alpha.tobytes() == alpha2.tobytes() and beta.tobytes() == beta2.tobytes()
        assert rel_err(alpha, o.alpha) < 1e-11 and rel_err(beta, o.beta) < 1e-11
    monkeypatch.delenv("LANCZOS_HIP_PANEL_BITS")
    # one row block, far more entries than one workgroup takes: cut into pieces all the same
    n = 3000
    rng = np.random.default_rng(8)
    head = np.arange(n + 1, dtype=np.int64) * 200
    dense = la.CrsMatrix(rng.standard_normal(n * 200), rng.integers(0, n, n * 200).astype(np.int64), head)
    xd = rng.standard_normal(n)
    with la.Engine(dense, 3, spmv="panel") as e:
        yd = e.spmv(xd)
    assert np.all(np.abs(yd - oracle.matvec_crs(dense.row_head_indexes, dense.column_index, dense.values, xd))
                  <= spmv_tolerance(dense, xd))


def test_device_coo_to_crs_is_bit_exact(gpu, golden_dir):
    """Device build against the reference's own crs.c output and against the host build."""
    from test_oracle_golden import CASES, read_ref_bin
    for name in CASES:
        for swap in (0, 1):
            ref = read_ref_bin(os.path.join(golden_dir, f"ref_{name}_swap{swap}.bin"))
            coo, dim = la.read_mat_coo_incremental(os.path.join(golden_dir, f"triplets_{name}.txt"))
            with la.Engine.from_coo(coo, dim, swap, 3) as e:
                got = e.download_crs(coo.shape[0])
            assert got.values.tobytes() == ref["values"].tobytes(), (name, swap)
            assert got.column_index.tobytes() == ref["cols"].tobytes()
            assert got.row_head_indexes.tobytes() == ref["head"].tobytes()
    # larger: shuffled duplicates, empty rows at both ends, 1.5M entries
    rng = np.random.default_rng(21)
    n, m = 200000, 1500000
    coo = np.zeros(m, la.COO_DTYPE)
    coo["index_row"] = rng.integers(100, n - 100, m)
    coo["index_column"] = rng.integers(0, n, m)
    coo["value"] = rng.standard_normal(m)
    for swap in (0, 1):
        want = la.convert_from_coo(coo, n, swap)
        with la.Engine.from_coo(coo, n, swap, 3) as e:
            got = e.download_crs(m)
            x = rng.standard_normal(n)
            y = e.spmv(x)
        assert got.values.tobytes() == want.values.tobytes()
        assert got.column_index.tobytes() == want.column_index.tobytes()
        assert got.row_head_indexes.tobytes() == want.row_head_indexes.tobytes()
        ref_y = oracle.matvec_crs(want.row_head_indexes, want.column_index, want.values, x)
        assert np.all(np.abs(y - ref_y) <= spmv_tolerance(want, x))
    bad = coo[:10].copy()
    bad["index_row"][7] = n
    with pytest.raises(la.LanczosHipError, match=r"index out of range at entry 7 \(row=200000"):
        la.Engine.from_coo(bad, n, 0, 3)
    empty = np.zeros(0, la.COO_DTYPE)
    with la.Engine.from_coo(empty, 5, 0, 3) as e:
        assert np.all(e.spmv(np.ones(5)) == 0)


def test_create_rejects_bad_matrices(gpu):
    a = la.synth_laplacian(1, 100)
    bad = la.CrsMatrix(a.values, a.column_index.copy(), a.row_head_indexes)
    bad.column_index[5] = 100                          # out of range: would be an out-of-bounds gather
    with pytest.raises(la.LanczosHipError, match="column index out of range"):
        la.Engine(bad, 3)
    bad = la.CrsMatrix(a.values, a.column_index, a.row_head_indexes.copy())
    bad.row_head_indexes[10] = bad.row_head_indexes[11] + 1
    with pytest.raises(la.LanczosHipError, match="not monotone"):
        la.Engine(bad, 3)
    with pytest.raises(la.LanczosHipError, match="max_iter must be >= 2"):
        la.Engine(a, 1)


REORTH_TO_ORACLE = {"full": "full", "mgs": "full", "full2": "full", "none": "none", "cuda_compat": "cuda_compat"}


@pytest.mark.parametrize("reorth", list(REORTH_TO_ORACLE))
def test_alpha_beta_sequences_match_oracle(gpu, reorth):
    for a in (la.synth_laplacian(1, 10000), la.synth_laplacian(2, 100), la.synth_laplacian(3, 24),
              la.synth_random_symmetric(3000, 10, 7)):
        v0 = la.gaussian_fill(123456789, 0, a.dimension)
        steps = 20
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, steps + 1, 0.0,
                               REORTH_TO_ORACLE[reorth])
        with la.Engine(a, steps + 1, reorth=reorth) as e:
            e.set_start(v0)
            alpha, beta = e.run(steps)
        assert rel_err(alpha, o.alpha) < 1e-12 and rel_err(beta, o.beta) < 1e-12, reorth


@pytest.mark.parametrize("reorth", ["full", "mgs", "full2"])
def test_ritz_values_at_100_steps_match_oracle(gpu, reorth):
    """The reference's CPU-runnable configuration: ~10k rows, 100 Lanczos steps, 4 values."""
    for a in (la.synth_laplacian(1, 10000), la.synth_laplacian(2, 100)):
        v0 = la.gaussian_fill(123456789, 0, a.dimension)
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 101)
        with la.Engine(a, 101, reorth=reorth) as e:
            e.set_start(v0)
            alpha, beta = e.run(100)
        ritz = la.ritz_values(alpha, beta, 100, 4)
        assert rel_err(ritz, o.eigenvalues) < 1e-10
        # every intermediate step too (what the reference prints per iteration)
        for k in (1, 2, 3, 10, 50, 99):
            assert rel_err(la.ritz_values(alpha, beta, k, 4)[:min(k, 4)], o.ritz_hist[k - 1][:min(k, 4)]) < 1e-10


def _known(golden_dir, name, threads):
    cases = json.load(open(os.path.join(golden_dir, "reference_known_answers.json")))["cases"]
    return [c for c in cases if c["name"] == name and c["threads"] == threads][0]


def test_gpu_reproduces_the_references_recorded_ritz_values(gpu, golden_dir):
    """GPU against numbers the unmodified reference printed (not against the oracle)."""
    # 1-D, n = 10000, triplets -> convert_from_coo(swap=1) through the product's host code
    n = 10000
    r, c, v = laplacian_triplets_rowwise(n)
    coo = np.zeros(r.size, la.COO_DTYPE)
    coo["index_row"], coo["index_column"], coo["value"] = r, c, v
    a = la.convert_from_coo(coo, n, 1)
    with la.Engine(a, 100) as e:
        e.set_start(oracle.xorshift_start(n))
        alpha, beta = e.run(99)
    assert rel_err(la.ritz_values(alpha, beta, 99, 4), _known(golden_dir, "lap1d_10000", 1)["eig"]) < 1e-10
    # 3-D 64^3, max_iter = 101
    N = 64
    r, c, v = laplacian3d_triplets_by_column(N)
    coo = np.zeros(r.size, la.COO_DTYPE)
    coo["index_row"], coo["index_column"], coo["value"] = r, c, v
    a = la.convert_from_coo(coo, N ** 3, 1)
    with la.Engine(a, 101) as e:
        e.set_start(oracle.xorshift_start(N ** 3))
        alpha, beta = e.run(100)
    assert rel_err(la.ritz_values(alpha, beta, 100, 4), _known(golden_dir, "lap3d_64", 1)["eig"]) < 1e-10
    # 3-D 128^3, max_iter = 51
    N = 128
    r, c, v = laplacian3d_triplets_by_column(N)
    coo = np.zeros(r.size, la.COO_DTYPE)
    coo["index_row"], coo["index_column"], coo["value"] = r, c, v
    a = la.convert_from_coo(coo, N ** 3, 1)
    with la.Engine(a, 51) as e:
        e.set_start(oracle.xorshift_start(N ** 3))
        alpha, beta = e.run(50)
    assert rel_err(la.ritz_values(alpha, beta, 50, 4), _known(golden_dir, "lap3d_128", 8)["eig"]) < 1e-10


def test_basis_is_orthonormal_and_runs_are_bitwise_repeatable(gpu):
    a = la.synth_laplacian(3, 20)
    v0 = la.gaussian_fill(5, 0, a.dimension)
    with la.Engine(a, 61) as e:
        e.set_start(v0)
        alpha, beta = e.run(60)
        V = np.stack([e.get_vector(k) for k in range(61)])
        gram = V @ V.T
        assert np.max(np.abs(gram - np.eye(61))) < 1e-12
        # T = V A V^T is tridiagonal with alpha on the diagonal and beta beside it
        AV = np.stack([oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, V[k]) for k in range(60)])
        T = V[:60] @ AV.T
        assert np.max(np.abs(np.diag(T) - alpha)) < 1e-12
        assert np.max(np.abs(np.diag(T, 1) - beta[:59])) < 1e-12
        e.set_start(v0)
        alpha2, beta2 = e.run(60)
        assert alpha.tobytes() == alpha2.tobytes() and beta.tobytes() == beta2.tobytes()


def test_converged_values_match_closed_form(gpu):
    a = la.synth_laplacian(2, 40)
    with la.Engine(a, 400) as e:
        e.set_start_seed(3)
        alpha, beta = e.run(399)
    exact = laplacian_eigs(2, 40, 2)
    got = la.ritz_values(alpha, beta, 399, 2)
    assert abs(got[0] - exact[0]) < 1e-10 * exact[0]


def test_ritz_vectors_have_small_residuals(gpu):
    a = la.synth_laplacian(2, 40)
    with la.Engine(a, 300) as e:
        e.set_start_seed(3)
        e.run(299)
        eig, vecs, est = e.ritz_vectors(299, 3)
    exact = laplacian_eigs(2, 40, 3)
    assert abs(eig[0] - exact[0]) < 1e-10 * exact[0]
    for i in range(3):
        y = vecs[i]
        assert abs(np.linalg.norm(y) - 1) < 1e-10
        r = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, y) - eig[i] * y
        # the true residual is bounded by the Lanczos estimate (plus rounding)
        assert np.linalg.norm(r) <= est[i] + 1e-10 * 8
    assert est[0] < 1e-8                                  # the lowest pair has converged
    with la.Engine(a, 5, reorth="none") as e:
        e.set_start_seed(3)
        e.run(4)
        with pytest.raises(la.LanczosHipError, match="needs the whole basis"):
            e.ritz_vectors(4, 1)


def test_selective_reorthogonalisation(gpu, tmp_path):
    """Same decisions and sequences as the CPU model of the rule; Ritz values as with full
    reorthogonalisation (src/lanczos.c:156-167) at equal step counts; basis semi-orthogonal."""
    a = la.synth_laplacian(3, 24)
    v0 = la.gaussian_fill(1, 0, a.dimension)
    steps = 300
    args = (a.row_head_indexes, a.column_index, a.values, v0, 4, steps + 1, 0.0)
    full = oracle.lanczos_crs(*args, "full")
    sel = oracle.lanczos_crs(*args, "selective")
    with la.Engine(a, steps + 1, reorth="selective") as e:
        e.set_start(v0)
        alpha, beta = e.run(steps)
        assert e.reorth_steps == sel.reorth_steps
        assert 2 <= e.reorth_steps <= steps // 10
        V = np.stack([e.get_vector(k) for k in range(steps + 1)])
    # early sequences agree closely; late ones of a semi-orthogonal run are sensitive to summation order
    assert rel_err(alpha[:100], sel.alpha[:100]) < 1e-9 and rel_err(beta[:100], sel.beta[:100]) < 1e-9
    # The cube's spectrum is degenerate: later copies of a multiple eigenvalue grow out of rounding
    # noise at a run-dependent step, so beyond step 100 only the settled values are compared.
    for k in (50, 100):
        assert rel_err(la.ritz_values(alpha, beta, k, 4), full.ritz_hist[k - 1]) < 1e-10
    assert rel_err(la.ritz_values(alpha, beta, 300, 2), full.ritz_hist[299][:2]) < 1e-10
    gram = V @ V.T - np.eye(steps + 1)
    assert np.max(np.abs(gram)) < 4 * np.sqrt(np.finfo(float).eps)
    assert np.max(np.abs(gram)) > 1e-12          # and it really did skip work
    # interrupted and resumed: bit for bit the same
    path = str(tmp_path / "sel.bin")
    with la.Engine(a, steps + 1, reorth="selective") as e:
        e.set_start(v0)
        a1, b1 = e.run(130)
        e.checkpoint_save(path)
    with la.Engine(a, steps + 1, reorth="selective") as e:
        e.checkpoint_load(path)
        a2, b2 = e.run(steps - 130)
    assert np.concatenate([a1, a2]).tobytes() == alpha.tobytes()
    assert np.concatenate([b1, b2]).tobytes() == beta.tobytes()
    # an irregular matrix with clustered converging values
    # an irregular matrix with a simple spectrum: every step comparable
    a = la.synth_random_symmetric(4000, 10, 3)
    v0 = la.gaussian_fill(2, 0, a.dimension)
    full = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 201, 0.0, "full")
    with la.Engine(a, 201, reorth="selective") as e:
        e.set_start(v0)
        alpha, beta = e.run(200)
        assert 2 <= e.reorth_steps < 100
    for k in (25, 50, 100, 150, 200):
        assert rel_err(la.ritz_values(alpha, beta, k, 4), full.ritz_hist[k - 1]) < 1e-10


def test_selective_rule_with_the_recurrence_as_two_plane_walks(gpu, monkeypatch, tmp_path):
    """The selective rule on one GPU keeps the basis unnormalised like the run without reorthogonalisation, so on a
    grid's plane-walking form its iteration is the same two walks (the product launched ahead while the host decides
    is then the next iteration's first walk).  A run long enough to reorthogonalise several times: the same decisions
    and sequences as with product + recurrence kernel, interrupted and resumed bit for bit."""
    a = la.synth_laplacian(3, 64)
    v0 = la.gaussian_fill(1, 0, a.dimension)
    steps = 600
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", mode)
        with la.Engine(a, steps + 1, reorth="selective") as e:
            assert e.spmv_name.endswith("p")
            e.set_start(v0)
            alpha, beta = e.run(steps)
            out[mode] = (alpha, beta, e.reorth_steps)
    assert out["1"][2] == out["0"][2] and out["1"][2] >= 3
    assert rel_err(out["1"][0][:200], out["0"][0][:200]) < 1e-11 and rel_err(out["1"][1][:200], out["0"][1][:200]) < 1e-11
    for k in (100, 300):
        assert rel_err(la.ritz_values(*out["1"][:2], k, 4), la.ritz_values(*out["0"][:2], k, 4)) < 1e-10
    assert rel_err(la.ritz_values(*out["1"][:2], steps, 2), la.ritz_values(*out["0"][:2], steps, 2)) < 1e-9
    path = str(tmp_path / "selw.bin")
    with la.Engine(a, steps + 1, reorth="selective") as e:
        e.set_start(v0)
        a1, b1 = e.run(430)
        e.checkpoint_save(path)
    with la.Engine(a, steps + 1, reorth="selective") as e:
        e.checkpoint_load(path)
        a2, b2 = e.run(steps - 430)
    assert np.concatenate([a1, a2]).tobytes() == out["1"][0].tobytes()
    assert np.concatenate([b1, b2]).tobytes() == out["1"][1].tobytes()


def test_thick_restart_converges_with_a_small_basis(gpu):
    """40 basis vectors, restarted: the lowest values of a 60 x 60 grid Laplacian to the closed form and
    to what an unrestarted run with full reorthogonalisation (src/lanczos.c:156-167) converges to."""
    a = la.synth_laplacian(2, 60)
    # the square's second eigenvalue is double; one start vector spans one copy of it, so a run that
    # stays orthogonal finds the distinct values
    exact = laplacian_eigs(2, 60, 6)
    exact = exact[np.concatenate([[True], np.diff(exact) > 1e-9])]
    v0 = la.gaussian_fill(17, 0, a.dimension)
    with la.Engine(a, 41) as e:
        e.set_start(v0)
        r = e.thick_restart(3, 12, 200, 1e-10, vectors=True)
    assert 2 <= r["cycles"] < 200 and r["products"] == 40 + (r["cycles"] - 1) * 28
    assert rel_err(r["eig"], exact[:3]) < 1e-10
    for i in range(3):
        y = r["vectors"][i]
        assert abs(np.linalg.norm(y) - 1) < 1e-10
        res = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, y) - r["eig"][i] * y
        assert np.linalg.norm(res) <= r["resid"][i] + 1e-10 * 8
    with la.Engine(a, 700) as e:
        e.set_start(v0)
        alpha, beta = e.run(699)
    assert rel_err(r["eig"][:2], la.ritz_values(alpha, beta, 699, 2)) < 1e-10
    # argument checks
    with la.Engine(a, 41, reorth="none") as e:
        e.set_start(v0)
        with pytest.raises(la.LanczosHipError, match="whole basis"):
            e.thick_restart(3, 12, 5, 1e-8)
    with la.Engine(a, 10) as e:
        e.set_start(v0)
        with pytest.raises(la.LanczosHipError, match="nth <= keep <= max_iter - 3"):
            e.thick_restart(3, 8, 5, 1e-8)


def test_engine_limits(gpu):
    a = la.synth_laplacian(1, 500)
    with la.Engine(a, 5) as e:
        with pytest.raises(la.LanczosHipError, match="no start vector"):
            e.run(1)
        e.set_start_seed(1)
        e.run(4)
        with pytest.raises(la.LanczosHipError, match="basis full"):
            e.run(1)
    with la.Engine(a, 5, reorth="none") as e:          # ring of 3 vectors: not bounded by max_iter
        e.set_start_seed(1)
        alpha, beta = e.run(50)
        assert np.all(np.isfinite(alpha)) and np.all(beta > 0)


@pytest.mark.parametrize("reorth", ["full", "none", "cuda_compat"])
def test_checkpoint_resume_is_bitwise(gpu, tmp_path, reorth):
    a = la.synth_laplacian(3, 20)
    v0 = la.gaussian_fill(8, 0, a.dimension)
    path = str(tmp_path / "ck.bin")
    with la.Engine(a, 41, reorth=reorth) as e:
        e.set_start(v0)
        alpha, beta = e.run(40)
    with la.Engine(a, 41, reorth=reorth) as e:
        e.set_start(v0)
        a1, b1 = e.run(17)
        e.checkpoint_save(path)
    with la.Engine(a, 41, reorth=reorth) as e:
        e.checkpoint_load(path)
        assert e.iterations == 17
        a2, b2 = e.run(23)
        ah, bh = e.history(40)
    assert np.concatenate([a1, a2]).tobytes() == alpha.tobytes()
    assert np.concatenate([b1, b2]).tobytes() == beta.tobytes()
    assert ah.tobytes() == alpha.tobytes() and bh.tobytes() == beta.tobytes()
    with la.Engine(la.synth_laplacian(3, 21), 41, reorth=reorth) as e:
        with pytest.raises(la.LanczosHipError, match="does not fit this engine"):
            e.checkpoint_load(path)


@pytest.mark.parametrize("reorth", ["full", "full2", "cuda_compat"])
def test_basis_kernels_on_groups_of_tiles_match_those_on_single_tiles(gpu, monkeypatch, reorth):
    """On long vectors k_dots and k_update take 128 KiB of every basis vector per workgroup at a time (the default from
    2^23 rows; forced here) and the tiles that do not fill a group one by one: the same iteration as on single tiles to
    rounding, and the CPU restatement's."""
    for a in (la.synth_laplacian(3, 30), la.synth_laplacian(1, 100000), la.synth_laplacian(2, 181)):
        v0 = la.gaussian_fill(17, 0, a.dimension)
        out = {}
        for big in ("0", "1"):
            monkeypatch.setenv("LANCZOS_HIP_UPDATE_BIG", big)
            monkeypatch.setenv("LANCZOS_HIP_DOTS_BIG", big)
            with la.Engine(a, 81, reorth=reorth) as e:
                e.set_start(v0)
                out[big] = e.run(80)
        assert rel_err(out["1"][0], out["0"][0]) < 1e-12 and rel_err(out["1"][1], out["0"][1]) < 1e-12
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 81, 0.0,
                               reorth="full" if reorth == "full2" else reorth)
        assert rel_err(la.ritz_values(*out["1"], 80, 4), o.ritz_hist[79]) < 1e-10


def test_iteration_as_two_plane_walks_matches_product_then_recurrence(gpu, monkeypatch):
    """Without reorthogonalisation a grid's dictionary form steps as two plane walks from 128^3 (the first leaves
    u . A u and writes nothing, the second forms A u again and applies the deferred-normalisation recurrence in its
    epilogue); LANCZOS_HIP_TWO_PASS forces either form at any size.  Same alpha/beta as the product followed by the
    recurrence kernel, and as the CPU restatement; the run continues across a switch of form."""
    for N in (64, 70, 90):
        a = la.synth_laplacian(3, N)
        v0 = la.gaussian_fill(3, 0, a.dimension)
        out = {}
        for mode in ("0", "1"):
            monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", mode)
            with la.Engine(a, 61, reorth="none") as e:
                assert e.spmv_name.endswith("p"), e.spmv_name
                e.set_start(v0)
                out[mode] = e.run(60)
        assert rel_err(out["1"][0], out["0"][0]) < 1e-13 and rel_err(out["1"][1], out["0"][1]) < 1e-13
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 61, 0.0, reorth="none")
        assert rel_err(out["1"][0][:25], o.alpha[:25]) < 1e-11 and rel_err(out["1"][1][:25], o.beta[:25]) < 1e-11
        with la.Engine(a, 61, reorth="none") as e:
            e.set_start(v0)
            monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", "1")
            a1, b1 = e.run(17)
            monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", "0")
            a2, b2 = e.run(20)
            monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", "1")
            a3, b3 = e.run(23)
        assert rel_err(np.concatenate([a1, a2, a3]), out["0"][0]) < 1e-13
        assert rel_err(np.concatenate([b1, b2, b3]), out["0"][1]) < 1e-13


def test_basis_that_does_not_fit_the_gpu_is_refused_cleanly(gpu):
    """A basis of 4 TB: creation fails with the runtime's message, and the next engine works."""
    a = la.synth_laplacian(3, 64)
    with pytest.raises(la.LanczosHipError, match="(?i)memory"):
        la.Engine(a, 2_000_000, reorth="full")
    v0 = la.gaussian_fill(5, 0, a.dimension)
    with la.Engine(a, 11, reorth="full") as e:
        e.set_start(v0)
        alpha, beta = e.run(10)
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 2, 11, 0.0)
    assert rel_err(alpha, o.alpha[:10]) < 1e-12


def run_dropin(a, nth, max_iter, threshold, env=None):
    """lanczos_hip_crs() in a child process so that its stdout/stderr can be read."""
    code = (
        "import sys, json, numpy as np, lanczos_amd as la\n"
        "d=np.load(sys.argv[1])\n"
        "a=la.CrsMatrix(d['v'],d['c'],d['h'])\n"
        f"rc,eig=la.lanczos_hip_crs(a,{nth},{max_iter},{threshold!r})\n"
        "sys.stdout.flush()\n"
        "print('RESULT '+json.dumps({'rc':rc,'eig':[float(x) for x in eig]}))\n")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.npz")
        np.savez(path, v=a.values, c=a.column_index, h=a.row_head_indexes)
        full_env = dict(os.environ)
        full_env.update(env or {})
        r = subprocess.run(["python", "-c", code, path], cwd=ROOT, capture_output=True, text=True, env=full_env,
                           timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    return res, r.stdout, r.stderr


def test_dropin_matches_oracle_and_prints_reference_lines(gpu):
    a = la.synth_laplacian(2, 100)
    v0 = la.gaussian_fill(123456789, 0, a.dimension)   # the drop-in's default seed
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 101, 0.0)
    res, out, err = run_dropin(a, 4, 101, 0.0)
    assert res["rc"] == 0
    assert rel_err(res["eig"], o.eigenvalues) < 1e-10
    lines = out.splitlines()
    assert "Reached max iterations (101)" in lines
    step_lines = [ln for ln in lines if ln[:1].isdigit() and "\t" in ln]
    assert len(step_lines) == 100 and step_lines[0].split("\t")[0] == "1"
    want = "\t".join(f"{x:.7f}" for x in o.ritz_hist[49])
    assert step_lines[49] == "50\t" + want + "\t"
    for name in ("init", "spmv", "diagonalization", "reorthogonalization"):
        assert f"[TOTAL] name={name}" in err
    assert "lanczos_hip" not in err                       # the JSON line is opt-in
    # selective mode through the environment, with the machine-readable timing line
    res, out, err = run_dropin(a, 4, 101, 0.0, env={"LANCZOS_HIP_REORTH": "selective", "LANCZOS_HIP_JSON": "1",
                                                    "LANCZOS_HIP_QUIET": "1"})
    assert res["rc"] == 0 and rel_err(res["eig"], o.eigenvalues) < 1e-10
    rec = json.loads([ln for ln in err.splitlines() if ln.startswith("{")][0])["lanczos_hip"]
    assert rec["iterations"] == 100 and rec["dimension"] == 10000 and rec["status"] == 0
    assert rec["reorth_steps"] < 50 and set(rec["sec"]) == {"init", "spmv", "diagonalization",
                                                            "reorthogonalization", "wall"}


def _exit_margin(o, nth, thr):
    """How clearly the oracle's run decides its Ritz-convergence exit (src/lanczos.c:120-126): the
    smallest factor by which any comparison of the run sits away from the threshold (> 1: no
    comparison is within that factor of flipping)."""
    margin = np.inf
    prev = np.zeros(nth)
    for k in range(o.steps):
        cnt = min(k + 1, nth)
        d = np.abs(o.ritz_hist[k][:cnt] - prev[:cnt])
        worst = d.max()                       # the step exits iff worst <= thr
        margin = min(margin, worst / thr if worst > thr else thr / max(worst, 1e-300))
        prev = o.ritz_hist[k].copy()
    return margin


def test_dropin_convergence_exits_follow_the_cpu_rule(gpu):
    """Exit kind, exit iteration and the exit line equal the CPU rule's (src/lanczos.c:120-133, 185-188)
    wherever the oracle's run decides with a clear margin; knife-edge thresholds compare values only."""
    a = la.synth_laplacian(1, 2000)
    v0 = la.gaussian_fill(123456789, 0, a.dimension)
    clear = 0
    for thr in (10e-5, 1e-2, 3e-4, 2e-3):
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 2, 400, thr)
        res, out, err = run_dropin(a, 2, 400, thr, env={"LANCZOS_HIP_JSON": "1"})
        rec = json.loads([ln for ln in err.splitlines() if ln.startswith('{"lanczos_hip"')][0])["lanczos_hip"]
        assert res["rc"] == 0
        assert o.exit_kind == 1                      # these runs end on standing Ritz values
        if _exit_margin(o, 2, thr) > 1.01:           # a 1 % margin is ~1e10 times the parity tolerance
            clear += 1
            assert rec["exit_kind"] == 1 and rec["exit_step"] == o.steps, (thr, rec, o.steps)
            step_lines = [ln for ln in out.splitlines() if ln[:1].isdigit() and "\t" in ln]
            assert len(step_lines) == o.steps - 1    # the exit step prints "converged", not its values
            diffs = o.ritz_hist[o.steps - 2] - o.ritz_hist[o.steps - 1]
            conv = [ln for ln in out.splitlines() if ln.startswith("converged ")]
            assert len(conv) == 1
            got = [float(x) for x in conv[0].split()[1:]]
            assert len(got) == 2 and np.allclose(got, diffs, rtol=0.06, atol=0)   # printed at %.1E
            assert "Reached max iterations" not in out
            assert rel_err(res["eig"], o.eigenvalues) < 1e-10
        else:
            assert rel_err(res["eig"], o.eigenvalues) < 1e-6
    assert clear >= 3
    # beta exit: a matrix with five distinct eigenvalues spans its Krylov space in five steps
    n = 1000
    d = np.repeat([1.0, 2.5, 4.0, 7.0, 11.0], n // 5)
    diag = la.CrsMatrix(d, np.arange(n, dtype=np.int64), np.arange(n + 1, dtype=np.int64))
    v0 = la.gaussian_fill(123456789, 0, n)
    o = oracle.lanczos_crs(diag.row_head_indexes, diag.column_index, diag.values, v0, 2, 50, 1e-6)
    res, out, err = run_dropin(diag, 2, 50, 1e-6, env={"LANCZOS_HIP_JSON": "1"})
    rec = json.loads([ln for ln in err.splitlines() if ln.startswith('{"lanczos_hip"')][0])["lanczos_hip"]
    assert o.exit_kind == 2 and o.steps == 5
    assert res["rc"] == 0 and rec["exit_kind"] == 2 and rec["exit_step"] == 5
    assert [ln for ln in out.splitlines() if ln.endswith("beta converged")] == ["0.0000000 beta converged"]
    assert rel_err(res["eig"], o.eigenvalues) < 1e-10
    # max-iter exit
    res, out, err = run_dropin(a, 2, 30, 0.0, env={"LANCZOS_HIP_JSON": "1"})
    rec = json.loads([ln for ln in err.splitlines() if ln.startswith('{"lanczos_hip"')][0])["lanczos_hip"]
    assert rec["exit_kind"] == 0 and rec["exit_step"] == 29 and "Reached max iterations (30)" in out
    # bad arguments: EXIT_FAILURE and the reference's message (src/lanczos_cuda.c:324-327)
    res, out, _ = run_dropin(a, 2, 1, 0.0)
    assert res["rc"] == 1 and "max_iter must be >= 2" in out


def test_lanczos_takes_a_custom_matvec_plugin(gpu):
    """lanczos() with a Matvec_General that is neither matvec_crs nor matvec_coo (src/lanczos.h:5,
    call site src/lanczos.c:110): the plugin runs on host memory once per iteration, the vectors stay
    on the GPU; Ritz values against the oracle run on the same operator, 1e-10 relative."""
    import scipy.sparse as sp
    a = la.synth_laplacian(2, 60)
    n = a.dimension
    m = sp.csr_matrix((a.values, a.column_index, a.row_head_indexes), shape=(n, n))
    shift = np.linspace(0.0, 1.0, n)
    calls = []

    def apply(x):                     # A x + diag(shift) x, never materialised as a CRS matrix
        calls.append(1)
        return m @ x + shift * x

    full = (m + sp.diags(shift)).tocsr()
    full.sort_indices()
    v0 = la.gaussian_fill(123456789, 0, n)     # the start vector lanczos() draws by default
    o = oracle.lanczos_crs(full.indptr.astype(np.int64), full.indices.astype(np.int64), full.data, v0, 4, 81, 0.0)
    eig = la.lanczos_plugin(apply, n, 4, 81, 0.0)
    assert len(calls) == 80                    # one product per iteration (src/lanczos.c:104, 110)
    assert rel_err(eig, o.eigenvalues) < 1e-10
    # the handle API: the same operator set on an engine that holds an empty matrix
    def cb(_mat, vec, dist):
        np.ctypeslib.as_array(dist, (n,))[:] = apply(np.ctypeslib.as_array(vec, (n,)))
    fn = la.MATVEC_FN(cb)
    empty = la.CrsMatrix(np.zeros(0), np.zeros(0, np.int64), np.zeros(n + 1, np.int64))
    for reorth in ("full", "selective", "none", "mgs"):
        oo = oracle.lanczos_crs(full.indptr.astype(np.int64), full.indices.astype(np.int64), full.data, v0, 4, 41,
                                0.0, reorth if reorth != "mgs" else "full")
        with la.Engine(empty, 41, reorth=reorth) as e:
            e.set_operator(fn)
            assert e.spmv_name == "host_operator"
            e.set_start(v0)
            alpha, beta = e.run(40)
        assert rel_err(la.ritz_values(alpha, beta, 40, 4), oo.eigenvalues) < 1e-10, reorth
        assert rel_err(alpha[:20], oo.alpha[:20]) < 1e-11 and rel_err(beta[:20], oo.beta[:20]) < 1e-11, reorth


def test_rows_spmv_with_tiles_loaded_straight_into_lds_gives_the_same_bits(gpu, monkeypatch):
    """LANCZOS_HIP_ROWS_DMA=1 (read at every launch): the row-wise CRS kernel whose col/val tiles go from
    memory into LDS without passing through registers; same sums in the same order as the register-staged
    form, on grids of one, two and three dimensions, awkward sizes and a band with random values."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    band = sp.diags([rng.standard_normal(5000 - abs(k)) for k in range(-3, 4)], list(range(-3, 4))).tocsr()  # 7 per row
    cases = [la.synth_laplacian(3, 33), la.synth_laplacian(2, 129), la.synth_laplacian(1, 70001),
             la.CrsMatrix(band.data, band.indices.astype(np.int64), band.indptr.astype(np.int64))]
    with plain_rows():
        for a in cases:
            x = rng.standard_normal(a.dimension)
            with la.Engine(a, 3, reorth="none", spmv="rows") as e:
                assert e.spmv_name == "rows64"
                monkeypatch.setenv("LANCZOS_HIP_ROWS_DMA", "0")
                y0 = e.spmv(x)
                monkeypatch.setenv("LANCZOS_HIP_ROWS_DMA", "1")
                y1 = e.spmv(x)
                monkeypatch.delenv("LANCZOS_HIP_ROWS_DMA")
            assert y0.tobytes() == y1.tobytes()
            assert np.max(np.abs(y0 - oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x))) < 1e-12


@pytest.mark.parametrize("reorth", ["full", "full2", "cuda_compat"])
def test_recurrence_fused_with_the_dot_products_gives_the_same_bits(gpu, monkeypatch, reorth):
    """The three-term recurrence (src/lanczos.c:150-154) and the first half of the reorthogonalisation pass
    (src/lanczos.c:156-167 as a blocked product) run as one sweep over w; LANCZOS_HIP_FUSE_RECUR=0 keeps the two
    launches.  Same tiles and the same order of every sum: alpha and beta equal bit for bit, on a size with
    several tiles per workgroup and on one smaller than a tile."""
    for a in (la.synth_laplacian(3, 40), la.synth_laplacian(1, 700), la.synth_random_symmetric(30000, 6, 3)):
        v0 = la.gaussian_fill(5, 0, a.dimension)
        runs = {}
        for fuse in ("1", "0", "short"):
            monkeypatch.setenv("LANCZOS_HIP_FUSE_RECUR", "0" if fuse == "0" else "1")
            # "1": the rule for vectors in HBM (at most 32 basis vectors ride with the recurrence); "short": the default
            # for vectors of at most 2 MiB, as these are (every pass fused: such a step is the sum of its launches)
            if fuse == "1":
                monkeypatch.setenv("LANCZOS_HIP_FUSE_MAX", "32")
            else:
                monkeypatch.delenv("LANCZOS_HIP_FUSE_MAX", raising=False)
            with la.Engine(a, 41, reorth=reorth, profile=True) as e:
                e.set_start(v0)
                runs[fuse] = e.run(40)
                launches = e.profile()["recur"]["launches"]
                # iteration 0 has no pass (cuda_compat: 0..2); with the limit of 32, iterations up to k = 32 (34) ride
                # with the dot products and the rest launch k_recur again
                first = 3 if reorth == "cuda_compat" else 1
                assert launches == {"0": 40, "1": first + (5 if reorth == "cuda_compat" else 7), "short": first}[fuse]
        assert all(runs[f][0].tobytes() == runs["0"][0].tobytes() and runs[f][1].tobytes() == runs["0"][1].tobytes()
                   for f in ("1", "short"))
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 41, 0.0,
                               "cuda_compat" if reorth == "cuda_compat" else "full")
        assert rel_err(la.ritz_values(*runs["1"], 40, 4), o.eigenvalues) < 1e-10


def test_a_host_operator_has_no_checkpoint_and_no_device_product(gpu, tmp_path):
    """An engine whose operator is a host function holds an empty matrix: nothing identifies the operator
    in a checkpoint file, and there is no device product for the entry point that allocates nothing."""
    import torch
    n = 4096
    fn = la.MATVEC_FN(lambda _m, vec, dist: None)
    empty = la.CrsMatrix(np.zeros(0), np.zeros(0, np.int64), np.zeros(n + 1, np.int64))
    with la.Engine(empty, 11) as e:
        e.set_operator(fn)
        e.set_start(np.ones(n))
        with pytest.raises(la.LanczosHipError, match="host function"):
            e.checkpoint_save(str(tmp_path / "op.ckpt"))
        x = torch.ones(n, dtype=torch.float64, device="cuda")
        y = torch.full((n,), 7.0, dtype=torch.float64, device="cuda")
        with pytest.raises(la.LanczosHipError, match="host function"):
            e.spmv_noalloc(x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        assert bool((y == 7.0).all())          # nothing was launched


def test_checkpoint_of_another_spmv_form_is_refused(gpu, tmp_path):
    """The saved partial sums belong to the SpMV form that left them: a file written under the dictionary
    form is not continued by an engine of the same matrix on the row-wise form."""
    a = la.synth_laplacian(3, 20)
    path = str(tmp_path / "form.ckpt")
    with la.Engine(a, 21, spmv="dict") as e:
        e.set_start_seed(3)
        e.run(5)
        e.checkpoint_save(path)
    with la.Engine(a, 21, spmv="stream") as e:
        with pytest.raises(la.LanczosHipError, match="another matrix or SpMV form"):
            e.checkpoint_load(path)
    with la.Engine(a, 21, spmv="dict") as e:
        e.checkpoint_load(path)
        assert e.iterations == 5


def test_c_program_with_a_matrix_free_plugin(gpu):
    """lanczos_amd/bin/plugin_demo: a C program whose only operator is a function (2-D Laplacian applied
    matrix-free) calls lanczos() of liblanczos_host; values against the oracle on the assembled matrix."""
    exe = os.path.join(ROOT, "lanczos_amd", "bin", "plugin_demo")
    a = la.synth_laplacian(2, 80)
    v0 = la.gaussian_fill(123456789, 0, a.dimension)
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 3, 61, 0.0)
    p = subprocess.run([exe, "80", "3", "61"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, LANCZOS_HIP_QUIET="1"))
    assert p.returncode == 0, p.stderr[-1000:]
    assert "operator calls\t60" in p.stdout
    eig = [float(ln.split("\t")[-1]) for ln in p.stdout.splitlines() if ln.startswith("eigenvalue")]
    assert len(eig) == 3 and np.max(np.abs(np.array(eig) - o.eigenvalues)) < 5e-12   # printed at %.12f


def test_dictionary_plane_walk_forms_give_the_same_bits(gpu, monkeypatch):
    """The plane-walking dictionary SpMV in its forms -- 1024 or 2048 rows of a plane per step (LANCZOS_HIP_DICT_SPAN; the long
    span is the default for long grid lines), a wavefront's pattern read from the LDS tables for every row or kept in registers
    (LANCZOS_HIP_DICT_CACHE 0 / 1 / 2), each a kernel of its own chosen when an engine first multiplies: the same bits in
    every form, on grids whose lines end inside, at the end of and across wavefronts and steps, and against the oracle's
    product (src/crs.c:155-178).  Also as row slabs' engines see it: through a run of a few steps."""
    rng = np.random.default_rng(4)
    for N in (64, 70, 97, 129):
        a = la.synth_laplacian(3, N)
        x = rng.standard_normal(a.dimension)
        got, runs = {}, {}
        for span in ("1024", "2048"):
            for c in ("0", "1", "2"):
                monkeypatch.setenv("LANCZOS_HIP_DICT_SPAN", span)
                monkeypatch.setenv("LANCZOS_HIP_DICT_CACHE", c)
                with la.Engine(a, 6, reorth="full", spmv="dict") as e:
                    got[span, c] = e.spmv(x)
                    e.set_start_seed(3)
                    runs[span, c] = e.run(5)
                    assert e.spmv_name == "dict27p"
        first = got["1024", "0"]
        assert all(v.tobytes() == first.tobytes() for v in got.values()), N
        a0, b0 = runs["1024", "0"]
        for (span, c), (al, be) in runs.items():      # the fused dot product sums per workgroup: another grid, another order
            assert rel_err(al, a0) < 1e-13 and rel_err(be, b0) < 1e-13, (N, span, c)
        want = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        assert np.max(np.abs(first - want)) <= 8 * np.finfo(float).eps * np.max(np.abs(want)), N


def test_c_program_walks_the_abi(gpu, tmp_path):
    """lanczos_amd/bin/api_check: a C program on include/lanczos_hip.h alone -- eight SpMV forms against each other on
    two Laplacians, five on a random and a power-law matrix (alpha/beta of 30 steps, products), Ritz vectors, a
    checkpoint resumed bit for bit, thick restart, the device COO->CRS build byte for byte the host's
    (src/crs.c:8-153), the block variant's product against the host product (src/crs.c:155-178), 2 and 5 slabs
    against one engine.  Once as it is, once with the block variant's regrouped stream forced on."""
    exe = os.path.join(ROOT, "lanczos_amd", "bin", "api_check")
    for extra in ({}, {"LANCZOS_HIP_SPMM_HUB": "2"}):
        p = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        assert "api_check: ok" in p.stdout
        assert p.stdout.count(" agrees ") == 22 and "refused" not in p.stderr


def test_driver_binary_on_triplet_file(gpu, tmp_path):
    """main -> read_mat_coo_incremental -> convert_from_coo(swap=1) -> lanczos_hip_crs."""
    n = 10000
    r, c, v = laplacian_triplets_rowwise(n)
    path = tmp_path / "lap1d.txt"
    with open(path, "w") as f:
        f.write("".join(f"{i + 1} {j + 1} {x:g}\n" for i, j, x in zip(r, c, v)))
    exe = os.path.join(ROOT, "lanczos_amd", "bin", "lanczos_main")
    coo = np.zeros(r.size, oracle.COO_DTYPE)
    coo["row"], coo["col"], coo["val"] = r, c, v
    vals, cols, head = oracle.coo_to_crs(coo, n, 1)
    v0 = la.gaussian_fill(123456789, 0, n)
    o = oracle.lanczos_crs(head, cols, vals, v0, 4, 101, 10e-5)
    for mode in ("crs_hip", "crs", "coo"):
        p = subprocess.run([exe, str(path), mode, "4", "101"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-1000:]
        assert f"mat dim: {n}" in p.stdout and f"mat num-of-non-zero: {r.size}" in p.stdout
        eig = [float(ln.split("\t")[-1]) for ln in p.stdout.splitlines() if ln.startswith("eigenvalue")]
        assert len(eig) == 4
        assert np.max(np.abs(np.array(eig) - o.eigenvalues)) < 5e-12   # printed at %.12f


def test_driver_binary_on_matrix_market_file(gpu, tmp_path):
    """The same matrix as a symmetric Matrix Market file (one triangle stored): the driver sees the
    banner, mirrors the triangle and arrives at the same eigenvalues as from the triplet file."""
    n = 10000
    r, c, v = laplacian_triplets_rowwise(n)
    keep = r >= c
    path = tmp_path / "lap1d.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n% 1-D Laplacian, lower triangle\n")
        f.write(f"{n} {n} {int(keep.sum())}\n")
        f.write("".join(f"{i + 1} {j + 1} {x:g}\n" for i, j, x in zip(r[keep], c[keep], v[keep])))
    exe = os.path.join(ROOT, "lanczos_amd", "bin", "lanczos_main")
    a = la.synth_laplacian(1, n)
    v0 = la.gaussian_fill(123456789, 0, n)
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 101, 10e-5)
    p = subprocess.run([exe, str(path), "crs_hip", "4", "101"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1000:]
    assert f"mat dim: {n}" in p.stdout and f"mat num-of-non-zero: {r.size}" in p.stdout
    eig = [float(ln.split("\t")[-1]) for ln in p.stdout.splitlines() if ln.startswith("eigenvalue")]
    assert len(eig) == 4 and np.max(np.abs(np.array(eig) - o.eigenvalues)) < 5e-12


def test_dropin_fills_eigenvectors_on_request(gpu):
    """LANCZOS_HIP_EIGENVECTORS=1: the `eigenvectors` argument both reference solvers ignore
    (src/lanczos.c:20, src/lanczos_cuda.c:284) receives the Ritz vectors; without it nothing is written."""
    a = la.synth_laplacian(1, 200)          # simple eigenvalues; 200 steps span the whole space
    n = a.dimension
    rc, eig, y = la.lanczos_hip_crs(a, 3, 201, 0.0, vectors=True)
    assert rc == 0 and y.shape == (3, n)
    want = laplacian_eigs(1, 200, 3)
    assert np.max(np.abs(eig - want) / want) < 1e-10
    for i in range(3):
        r = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, y[i]) - eig[i] * y[i]
        assert abs(np.linalg.norm(y[i]) - 1) < 1e-10 and np.linalg.norm(r) < 1e-8, (i, np.linalg.norm(r))
    assert abs(y[0] @ y[1]) < 1e-10 and abs(y[0] @ y[2]) < 1e-10 and abs(y[1] @ y[2]) < 1e-10
    # off by default: a pointer array handed in stays untouched
    import ctypes as C
    buf = np.full((3, n), 7.0)
    ptrs = (C.c_void_p * 3)(*[buf[i].ctypes.data for i in range(3)])
    e2 = np.zeros(3)
    s = a.as_struct()
    assert la._hip.lanczos_hip_crs(C.byref(s), e2.ctypes.data_as(la._dp), ptrs, 3, 101, 0.0) == 0
    assert np.all(buf == 7.0)


@pytest.mark.parametrize("reorth", ["full", "selective", "none"])
def test_dropin_checkpoints_and_resumes(gpu, tmp_path, reorth):
    """LANCZOS_HIP_CHECKPOINT / _EVERY / LANCZOS_HIP_RESUME: a run picked up from the last checkpoint ends
    with the eigenvalues of the uninterrupted run, bit for bit, and prints only the iterations it ran."""
    a = la.synth_laplacian(2, 100)
    ck = str(tmp_path / "run.ckpt")
    base = {"LANCZOS_HIP_REORTH": reorth, "LANCZOS_HIP_JSON": "1"}
    whole, out_whole, _ = run_dropin(a, 4, 101, 0.0, env=base)
    # the "interrupted" run: same matrix, stops at 70 iterations, checkpoints every 30 (the file holds >= 60)
    part, _, err_part = run_dropin(a, 4, 71, 0.0, env=dict(base, LANCZOS_HIP_CHECKPOINT=ck,
                                                            LANCZOS_HIP_CHECKPOINT_EVERY="30"))
    assert os.path.exists(ck) and not os.path.exists(ck + ".tmp")
    resumed, out_res, err_res = run_dropin(a, 4, 101, 0.0, env=dict(base, LANCZOS_HIP_RESUME=ck))
    assert resumed["rc"] == 0 and resumed["eig"] == whole["eig"]
    line = [ln for ln in err_res.splitlines() if ln.startswith("[INIT] name=resumed")][0]
    saved = int(line.split("iterations=")[1].split()[0])
    printed = int(line.split("printed=")[1].split()[0])
    assert 60 <= saved <= 70 and printed == 60       # the last checkpoint was written after line 60
    # every line the interrupted run had not printed when it wrote the file comes out now
    its = [int(ln.split("\t")[0]) for ln in out_res.splitlines() if ln[:1].isdigit() and "\t" in ln]
    assert its == list(range(printed + 1, 101))
    info = json.loads([ln for ln in err_res.splitlines() if ln.startswith('{"lanczos_hip"')][0])["lanczos_hip"]
    assert info["iterations"] == 100
    # a file that is no checkpoint is refused
    bad = str(tmp_path / "bad.ckpt")
    open(bad, "wb").write(b"not a checkpoint")
    res_bad, _, _ = run_dropin(a, 4, 101, 0.0, env=dict(base, LANCZOS_HIP_RESUME=bad))
    assert res_bad["rc"] != 0
    if reorth == "full":
        # a checkpoint of another matrix of the same size and mode is refused, not continued
        other = la.CrsMatrix(a.values * 1.5, a.column_index, a.row_head_indexes)
        res_o, out_o, _ = run_dropin(other, 4, 101, 0.0, env=dict(base, LANCZOS_HIP_RESUME=ck))
        assert res_o["rc"] != 0 and "another matrix" in out_o
        # damaged header fields (counts that would become loop bounds and buffer sizes) are refused
        raw = bytearray(open(ck, "rb").read())
        import struct
        # layout: magic[8], n, ldv, nnz (int64), fingerprint (uint64), then int32 fields:
        # reorth, max_iter, nvec, k, lazy, lazy_flushed_k, alpha_count, norm_count, ...
        base_off = 8 + 4 * 8
        for field, value in ((3, -5), (6, 1 << 20), (7, -1)):
            hurt = bytearray(raw)
            struct.pack_into("<i", hurt, base_off + 4 * field, value)
            open(bad, "wb").write(bytes(hurt))
            res_h, out_h, _ = run_dropin(a, 4, 101, 0.0, env=dict(base, LANCZOS_HIP_RESUME=bad))
            assert res_h["rc"] != 0, (field, value)
