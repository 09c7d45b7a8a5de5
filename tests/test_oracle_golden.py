"""The CPU restatement (oracle/) against everything that pins it:
the reference's own coo.c/crs.c outputs (tests/golden/ref_*.bin), the Ritz
values the unmodified reference printed for fixed start vectors
(reference_known_answers.json), MKL dsyevr outputs (mkl_dsyevr.json) and
closed-form Laplacian spectra.  No GPU needed."""
import json
import os

import numpy as np
import pytest

import oracle
from matrices import laplacian3d_triplets_by_column, laplacian_eigs, laplacian_triplets_rowwise, rel_err

CASES = ["lap1d_sorted", "lap1d_shuffled", "dups_empty_rows", "single_row", "random40"]


def read_ref_bin(path):
    """Layout documented in oracle/ref_driver.c."""
    raw = open(path, "rb").read()
    dim, length = np.frombuffer(raw, np.uint64, 2)
    dim, length = int(dim), int(length)
    off = 16
    coo = np.frombuffer(raw, oracle.COO_DTYPE, length, off); off += 16 * length
    values = np.frombuffer(raw, np.float64, length, off); off += 8 * length
    cols = np.frombuffer(raw, np.int64, length, off); off += 8 * length
    head = np.frombuffer(raw, np.int64, dim + 1, off); off += 8 * (dim + 1)
    x = np.frombuffer(raw, np.float64, dim, off); off += 8 * dim
    y_crs = np.frombuffer(raw, np.float64, dim, off); off += 8 * dim
    y_coo = np.frombuffer(raw, np.float64, dim, off); off += 8 * dim
    assert off == len(raw)
    return dict(dim=dim, length=length, coo=coo, values=values, cols=cols, head=head, x=x, y_crs=y_crs, y_coo=y_coo)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("swap", [0, 1])
def test_ingest_and_crs_bit_exact_with_reference(golden_dir, name, swap):
    ref = read_ref_bin(os.path.join(golden_dir, f"ref_{name}_swap{swap}.bin"))
    coo, dim = oracle.read_triplets(os.path.join(golden_dir, f"triplets_{name}.txt"))
    assert dim == ref["dim"] and coo.shape[0] == ref["length"]
    assert coo.tobytes() == ref["coo"].tobytes()
    values, cols, head = oracle.coo_to_crs(coo, dim, swap)
    assert values.tobytes() == ref["values"].tobytes()
    assert cols.tobytes() == ref["cols"].tobytes()
    assert head.tobytes() == ref["head"].tobytes()
    # matvecs: the reference's row sum may be reassociated by its simd pragma (src/crs.c:172)
    y = oracle.matvec_crs(head, cols, values, ref["x"])
    scale = np.abs(ref["y_crs"]).max() + 1.0
    assert np.max(np.abs(y - ref["y_crs"])) <= 4 * np.finfo(float).eps * scale
    y2 = oracle.matvec_coo(coo, dim, ref["x"])
    assert y2.tobytes() == ref["y_coo"].tobytes()  # serial scatter-add in entry order: exact


def test_parse_line_forms():
    assert oracle.parse_line("3 4 1.5") == (0, 2, 3, 1.5)
    assert oracle.parse_line("  7\t9   -2e-3") == (0, 6, 8, -2e-3)
    assert oracle.parse_line("x 1 2")[0] == 1       # the reference exits here (src/coo.c:19-22)
    assert oracle.parse_line("1 y 2")[0] == 2       # and here (src/coo.c:27-30)
    rc, r, c, v = oracle.parse_line("0 5 1")        # a zero index becomes -1 (src/coo.c:37)
    assert (rc, r, c) == (0, -1, 4)


def test_coo_to_crs_rejects_out_of_range():
    coo = np.zeros(2, oracle.COO_DTYPE)
    coo["row"] = [0, 5]
    coo["col"] = [0, 0]
    with pytest.raises(IndexError):
        oracle.coo_to_crs(coo, 3, 0)


def _known_answers(golden_dir):
    return json.load(open(os.path.join(golden_dir, "reference_known_answers.json")))["cases"]


def _crs_from_triplets(rows, cols, vals, n):
    coo = np.zeros(rows.size, oracle.COO_DTYPE)
    coo["row"], coo["col"], coo["val"] = rows, cols, vals
    return oracle.coo_to_crs(coo, n, 1)


def test_lanczos_reproduces_reference_ritz_values_1d(golden_dir):
    n = 10000
    v, c, h = _crs_from_triplets(*laplacian_triplets_rowwise(n), n)
    r = oracle.lanczos_crs(h, c, v, oracle.xorshift_start(n), 4, 100)
    assert r.steps == 99 and r.exit_kind == 0
    for case in _known_answers(golden_dir):
        if case["name"] == "lap1d_10000":
            # tolerance of the parity target: 1e-10 relative
            assert rel_err(r.eigenvalues, case["eig"]) < 1e-10


def test_lanczos_reproduces_reference_ritz_values_3d_64(golden_dir):
    N = 64
    v, c, h = _crs_from_triplets(*laplacian3d_triplets_by_column(N), N ** 3)
    r = oracle.lanczos_crs(h, c, v, oracle.xorshift_start(N ** 3), 4, 101)
    assert r.steps == 100
    for case in _known_answers(golden_dir):
        if case["name"] == "lap3d_64":
            assert rel_err(r.eigenvalues, case["eig"]) < 1e-10


def test_diag_partial_against_mkl_dsyevr(golden_dir):
    data = json.load(open(os.path.join(golden_dir, "mkl_dsyevr.json")))
    for case in data["cases"]:
        alpha, beta = np.array(case["alpha"]), np.array(case["beta"])
        n = alpha.size
        t = np.zeros((n + 3, n + 3), order="F")      # ld > n, as the loop's ld = max_iter
        t[np.arange(n), np.arange(n)] = alpha
        t[np.arange(n - 1), np.arange(1, n)] = beta
        t[np.arange(1, n), np.arange(n - 1)] = 123.0  # lower triangle must be ignored ('U')
        eig = oracle.diag_partial(t, n, case["nth"])
        tnorm = np.abs(alpha).max() + 2 * (np.abs(beta).max() if n > 1 else 0.0)
        iu = min(case["nth"], n)
        assert np.max(np.abs(eig[:iu] - np.array(case["eig"]))) <= 64 * np.finfo(float).eps * tnorm
        assert np.all(eig[iu:] == 0.0)               # zero fill, src/util.c:269-271


def test_converged_ritz_values_match_closed_form():
    # 2-D 5-point Laplacian 24 x 24: run to the full Krylov dimension budget that
    # converges the lowest eigenvalue
    import lanczos_amd as la
    a = la.synth_laplacian(2, 24)
    v0 = la.gaussian_fill(1, 0, a.dimension)
    r = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 2, 200)
    exact = laplacian_eigs(2, 24, 1)
    assert abs(r.eigenvalues[0] - exact[0]) < 1e-10 * exact[0] + 1e-13


def test_reorth_modes_and_exits():
    import lanczos_amd as la
    a = la.synth_laplacian(1, 200)
    v0 = la.gaussian_fill(2, 0, 200)
    full = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 3, 40)
    none = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 3, 40, reorth="none")
    compat = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 3, 40, reorth="cuda_compat")
    # first three steps do not reorthogonalise in any mode
    assert np.array_equal(full.alpha[:1], none.alpha[:1])
    assert rel_err(full.alpha[:10], none.alpha[:10]) < 1e-8
    assert rel_err(full.alpha[:10], compat.alpha[:10]) < 1e-8
    # a huge threshold converges at the second step (first step compares with zeros)
    conv = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 3, 40, threshold=100.0)
    assert conv.exit_kind == 1 and conv.steps == 1
    # max_iter < 2 is rejected (src/lanczos.c:32-35)
    with pytest.raises(ValueError):
        oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 3, 1)


def test_lanczos_reproduces_reference_ritz_values_3d_128(golden_dir):
    N = 128
    v, c, h = _crs_from_triplets(*laplacian3d_triplets_by_column(N), N ** 3)
    r = oracle.lanczos_crs(h, c, v, oracle.xorshift_start(N ** 3), 4, 51)
    assert r.steps == 50
    case = [k for k in _known_answers(golden_dir) if k["name"] == "lap3d_128"][0]
    assert rel_err(r.eigenvalues, case["eig"]) < 1e-10
