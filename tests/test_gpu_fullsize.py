"""Full-size checks on the benchmark configuration (3-D 7-point Laplacian 256^3,
n = 16 777 216, nnz = 117 047 296) through size-independent properties, plus a
short direct comparison with the CPU restatement."""
import numpy as np
import pytest

import lanczos_amd as la
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lap256():
    return la.synth_laplacian(3, 256)


def test_spmv_symmetry_linearity_and_constant_vector(gpu, lap256):
    a = lap256
    n = a.dimension
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    with la.Engine(a, 3, reorth="none") as e:
        ax, ay = e.spmv(x), e.spmv(y)
        # symmetric operator: x.Ay == y.Ax
        assert abs(x @ ay - y @ ax) <= 1e-12 * (np.linalg.norm(x) * np.linalg.norm(ay))
        # linearity
        axy = e.spmv(2.0 * x - 3.0 * y)
        assert np.max(np.abs(axy - (2.0 * ax - 3.0 * ay))) < 1e-12 * 12 * 8
        # A * ones = number of missing neighbours (Dirichlet boundary), exactly
        ones = e.spmv(np.ones(n))
        idx = np.arange(n)
        cx, cy, cz = idx % 256, (idx // 256) % 256, idx // 65536
        want = ((cx == 0).astype(float) + (cx == 255) + (cy == 0) + (cy == 255) + (cz == 0) + (cz == 255))
        assert np.array_equal(ones, want)
        # against the CPU restatement on the full matrix
        ref = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, x)
        assert np.max(np.abs(ax - ref)) <= 16 * np.finfo(float).eps * 12 * np.abs(x).max()


def test_first_steps_match_oracle_and_basis_stays_orthonormal(gpu, lap256):
    a = lap256
    n = a.dimension
    v0 = la.gaussian_fill(123456789, 0, n)
    steps = 8
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, steps + 1, 0.0)
    with la.Engine(a, 41) as e:
        e.set_start(v0)
        alpha, beta = e.run(steps)
        assert np.max(np.abs(alpha - o.alpha) / np.abs(o.alpha)) < 1e-12
        assert np.max(np.abs(beta - o.beta) / np.abs(o.beta)) < 1e-12
        alpha2, beta2 = e.run(32)
        alpha, beta = np.concatenate([alpha, alpha2]), np.concatenate([beta, beta2])
        # spectrum of the 256^3 Dirichlet Laplacian lies in (0, 12)
        ritz = la.tridiag_eigs(alpha, beta, 40, 40)
        assert ritz[0] > 0 and ritz[-1] < 12
        vecs = [e.get_vector(k) for k in (0, 1, 17, 39, 40)]
        for i, u in enumerate(vecs):
            for j, w in enumerate(vecs):
                assert abs(u @ w - (1.0 if i == j else 0.0)) < 1e-12
