"""GPU tests of the block variant (16 vectors per step, fp64 MFMA contractions).

The reference has no block form (src/lanczos.c iterates on one vector): the block-specific
quantities are checked against the numpy model in oracle/block_lanczos.py (parity unpinned by the
reference), the spectrum against the single-vector path that follows src/lanczos.c:104-191 and
against closed forms.
"""
import numpy as np
import pytest

import lanczos_amd as la
import oracle
from oracle import block_lanczos as bl
from matrices import laplacian_eigs, rel_err

pytestmark = pytest.mark.gpu
W = la.BLOCK_WIDTH


def start_block(n, seed):
    return la.gaussian_fill(seed, 0, n * W).reshape(n, W)


@pytest.mark.parametrize("make", [lambda: la.synth_laplacian(3, 20), lambda: la.synth_laplacian(1, 777),
                                  lambda: la.synth_random_symmetric(5000, 10, 3)])
def test_spmm_matches_oracle(gpu, make):
    a = make()
    x = start_block(a.dimension, 11)
    with la.BlockEngine(a, 2) as e:
        e.set_block(0, x)
        e.spmm(0, 1)
        y = e.get_block(1)
    want = bl.spmm(a.row_head_indexes, a.column_index, a.values, x)
    bound = bl.spmm(a.row_head_indexes, a.column_index, np.abs(a.values), np.abs(x))
    assert np.all(np.abs(y - want) <= 1e-14 * bound + 1e-300)
    # column c of the block product is the single-vector product of column c (src/crs.c:155-178)
    col = oracle.matvec_crs(a.row_head_indexes, a.column_index, a.values, np.ascontiguousarray(x[:, 5]))
    assert np.all(np.abs(y[:, 5] - col) <= 1e-14 * bound[:, 5] + 1e-300)


def test_spmm_with_empty_and_very_long_rows(gpu):
    """Ragged rows, empty rows, unsorted duplicate columns, hubs of 600 / 5000 / 40000 entries."""
    from test_gpu_parity import irregular_matrix
    a = irregular_matrix(50000, 4, long_rows=((7, 600), (31000, 5000), (49999, 40000), (0, 513)))
    x = start_block(a.dimension, 12)
    with la.BlockEngine(a, 2) as e:
        e.set_block(0, x)
        e.spmm(0, 1)
        y = e.get_block(1)
    want = bl.spmm(a.row_head_indexes, a.column_index, a.values, x)
    bound = bl.spmm(a.row_head_indexes, a.column_index, np.abs(a.values), np.abs(x))
    maxlen = int(np.diff(a.row_head_indexes).max())
    assert np.all(np.abs(y - want) <= (maxlen + 4) * np.finfo(float).eps * bound + 1e-300)
    assert np.all(y[np.diff(a.row_head_indexes) == 0] == 0)


@pytest.mark.parametrize("hot_columns", ["196608", "0", "300"])
def test_spmm_of_hub_rows_regrouped_by_column_class(gpu, monkeypatch, hot_columns):
    """Power-law graphs: the rows of more than 16 entries hold nearly all of the matrix; their entries are regrouped
    by a hash of the column (so that each XCD's L2 caches its own eighth of X) and by column popularity.
    Against the numpy product, against the row-wise kernels (LANCZOS_HIP_SPMM_HUB=0), twice for the same bits;
    with every column in the popular set (the default at this size), with no such split, and with 300 popular
    columns (both kinds of classes in use).  The ragged matrix adds unsorted duplicate columns, empty rows and
    rows of 513 to 40000 entries, some of which cross dozens of 64-entry slices."""
    from matrices import rmat_laplacian
    from test_gpu_parity import crs_from_scipy, irregular_matrix
    monkeypatch.setenv("LANCZOS_HIP_SPMM_HOT_COLUMNS", hot_columns)
    cases = [crs_from_scipy(rmat_laplacian(15)), crs_from_scipy(rmat_laplacian(12, edge_factor=40)),
             irregular_matrix(50000, 4, long_rows=((7, 600), (31000, 5000), (49999, 40000), (0, 513), (123, 17),
                                                   (124, 16), (125, 257)))]
    for a in cases:
        x = start_block(a.dimension, 12)
        want = bl.spmm(a.row_head_indexes, a.column_index, a.values, x)
        bound = bl.spmm(a.row_head_indexes, a.column_index, np.abs(a.values), np.abs(x))
        maxlen = int(np.diff(a.row_head_indexes).max())
        got = {}
        for mode in ("2", "0"):                      # 2: regrouped form whatever the share of such rows; 0: row-wise
            monkeypatch.setenv("LANCZOS_HIP_SPMM_HUB", mode)
            with la.BlockEngine(a, 2) as e:
                e.set_block(0, x)
                e.spmm(0, 1)
                got[mode] = e.get_block(1)
                e.set_block(1, np.full_like(x, 3.0))   # a second product must not depend on what the first left
                e.spmm(0, 1)
                assert e.get_block(1).tobytes() == got[mode].tobytes()
            assert np.all(np.abs(got[mode] - want) <= (maxlen + 4) * np.finfo(float).eps * bound + 1e-300), mode
        assert np.all(got["2"][np.diff(a.row_head_indexes) == 0] == 0)
        # rows the regrouped form does not touch (up to 16 entries) come from the same kernel as before
        short = np.diff(a.row_head_indexes) <= 16
        assert got["2"][short].tobytes() == got["0"][short].tobytes()


def test_block_run_on_a_power_law_graph_laplacian(gpu):
    """R-MAT graph Laplacian (hubs, isolated vertices): the block run against the numpy model, and the
    single-vector path with full reorthogonalisation (src/lanczos.c:156-167) on the top of the spectrum."""
    from matrices import rmat_laplacian
    from test_gpu_parity import crs_from_scipy
    lap = rmat_laplacian(13)
    a = crs_from_scipy(-lap)                    # lowest of -L = largest of L: well separated hubs
    assert int(np.diff(a.row_head_indexes).max()) > 512
    x0 = start_block(a.dimension, 6)
    steps = 10
    ma, mb, _ = bl.block_lanczos(a.row_head_indexes, a.column_index, a.values, x0, steps)
    with la.BlockEngine(a, steps) as e:
        e.set_start(x0)
        e.run(steps)
        for k in (2, 5, 10):
            assert rel_err(e.ritz(4, k), bl.ritz(ma, mb, k, 4)) < 1e-10
        got = e.ritz(3)
    with la.Engine(a, 121) as e1:
        e1.set_start_seed(2)
        alpha, beta = e1.run(120)
    assert rel_err(got[:3], la.ritz_values(alpha, beta, 120, 3)) < 1e-10


def test_three_term_part_folded_into_the_full_pass_is_the_same_iteration(gpu, monkeypatch):
    """By default the three-term part of a block step is not written out (the full pass's products are taken of W
    less that part, formed in registers; one update applies both sets of coefficients); LANCZOS_HIP_BLOCK_FUSE=0
    runs it as a pass of its own.  Twelve steps (two segments of basis blocks in the Gram kernel): the same T,
    the same basis, to rounding."""
    a = la.synth_laplacian(3, 24)
    x0 = start_block(a.dimension, 4)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("LANCZOS_HIP_BLOCK_FUSE", fuse)
        with la.BlockEngine(a, 12) as e:
            e.set_start(x0)
            e.run(12)
            out[fuse] = (e.tmatrix(), e.ritz(6), e.get_block(12))
    (ta0, tb0), r0, q0 = out["0"]
    (ta1, tb1), r1, q1 = out["1"]
    assert np.max(np.abs(ta0 - ta1)) < 1e-12 * np.max(np.abs(ta0)) and np.max(np.abs(tb0 - tb1)) < 1e-12 * np.max(np.abs(ta0))
    assert rel_err(r1, r0) < 1e-12
    assert np.max(np.abs(q0 - q1)) < 1e-9
    assert np.max(np.abs(q1.T @ q1 - np.eye(W))) < 1e-12


@pytest.mark.parametrize("n,nblocks", [(16, 1), (4000, 3), (33333, 11), (70001, 20)])
def test_projection_kernels_match_numpy(gpu, n, nblocks):
    """H = Q^T W and W -= Q H on arbitrary (non-orthogonal) data: checks the MFMA operand maps."""
    a = la.synth_laplacian(1, n)
    rng = np.random.default_rng(n)
    q = [rng.standard_normal((n, W)) for _ in range(nblocks)]
    w = rng.standard_normal((n, W))
    with la.BlockEngine(a, nblocks + 1) as e:
        for t, qt in enumerate(q):
            e.set_block(t, qt)
        e.set_block(nblocks, w)
        h = e.project(0, nblocks, nblocks)
        w_new = e.get_block(nblocks)
        # a sub-range that does not start at block 0
        if nblocks >= 3:
            e.set_block(nblocks, w)
            h_sub = e.project(1, nblocks - 1, nblocks)
            w_sub = e.get_block(nblocks)
    scale = np.sqrt(n) * 4
    for t in range(nblocks):
        assert np.max(np.abs(h[t] - q[t].T @ w)) < 1e-13 * scale * np.sqrt(n)
    want = w - sum(q[t] @ (q[t].T @ w) for t in range(nblocks))
    assert np.max(np.abs(w_new - want)) < 1e-12 * np.max(np.abs(want))
    if nblocks >= 3:
        for t in range(1, nblocks - 1):
            assert np.max(np.abs(h_sub[t - 1] - q[t].T @ w)) < 1e-13 * scale * np.sqrt(n)
        want = w - sum(q[t] @ (q[t].T @ w) for t in range(1, nblocks - 1))
        assert np.max(np.abs(w_sub - want)) < 1e-12 * np.max(np.abs(want))


def test_orthonormalise_is_a_qr_factorisation(gpu):
    n = 12345
    a = la.synth_laplacian(1, n)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, W)) @ rng.standard_normal((W, W))     # mildly ill conditioned
    with la.BlockEngine(a, 1) as e:
        e.set_block(0, x)
        r = e.orthonormalise(0)
        q = e.get_block(0)
        assert np.max(np.abs(q.T @ q - np.eye(W))) < 1e-13
        assert np.allclose(np.tril(r, -1), 0) and np.all(np.diag(r) > 0)
        assert np.max(np.abs(q @ r - x)) < 1e-12 * np.max(np.abs(x))
        dep = x.copy()
        dep[:, 7] = dep[:, 3]
        with pytest.raises(la.LanczosHipError, match="rank deficient"):
            e.set_start(dep)


def test_block_run_matches_the_numpy_model(gpu):
    a = la.synth_random_symmetric(6000, 8, 5)
    x0 = start_block(a.dimension, 21)
    steps = 12
    ma, mb, basis = bl.block_lanczos(a.row_head_indexes, a.column_index, a.values, x0, steps)
    with la.BlockEngine(a, steps) as e:
        e.set_start(x0)
        e.run(steps)
        assert e.steps == steps
        ga, gb = e.tmatrix()
        v = np.hstack([e.get_block(t) for t in range(steps + 1)])
        for k in (1, 2, 6, 12):
            assert rel_err(e.ritz(4, k), bl.ritz(ma, mb, k, 4)) < 1e-10
        with pytest.raises(la.LanczosHipError, match="basis full"):
            e.run(1)
    assert np.max(np.abs(v.T @ v - np.eye(v.shape[1]))) < 1e-12
    scale = np.max(np.abs(ma))
    assert np.max(np.abs(ga[:4] - ma[:4])) < 1e-10 * scale and np.max(np.abs(gb[:4] - mb[:4])) < 1e-10 * scale
    # T is the projection of A on the basis: V^T A V is block tridiagonal with these blocks
    av = bl.spmm(a.row_head_indexes, a.column_index, a.values, v[:, :steps * W])
    proj = v[:, :steps * W].T @ av
    assert np.max(np.abs(proj - bl.assemble_t(ga, gb, steps))) < 1e-11 * scale


def test_block_run_finds_the_spectrum_with_multiplicities(gpu):
    """Converged values: closed form (the cube's second eigenvalue is triple: a single-vector run
    sees one copy until rounding breeds the others, a block sees all at once) and the
    single-vector path that follows src/lanczos.c."""
    a = la.synth_laplacian(3, 12)
    exact = laplacian_eigs(3, 12, 7)
    with la.BlockEngine(a, 45) as e:
        e.set_start_seed(9)
        e.run(45)
        got = e.ritz(7)
    assert rel_err(got, exact) < 1e-10
    assert abs(exact[1] - exact[3]) < 1e-14                          # the triple one
    with la.Engine(a, 400) as e1:
        e1.set_start_seed(9)
        alpha, beta = e1.run(399)
    single = la.ritz_values(alpha, beta, 399, 1)
    assert rel_err(got[:1], single) < 1e-10


def test_block_limits(gpu):
    small = la.synth_laplacian(1, 10)
    with pytest.raises(la.LanczosHipError, match="below the block width"):
        la.BlockEngine(small, 2)
    a = la.synth_laplacian(1, 100)
    with pytest.raises(la.LanczosHipError, match="max_steps must be >= 1"):
        la.BlockEngine(a, 0)
    with la.BlockEngine(a, 2) as e:
        with pytest.raises(la.LanczosHipError, match="no start block"):
            e.run(1)


@pytest.mark.parametrize("slabs", [2, 3, 5])
def test_block_variant_on_row_slabs_matches_one_engine(gpu, slabs, monkeypatch):
    """The block variant with its rows cut into slabs (one host process, every slab on this GPU through the loopback
    transport): per block step an all-gather of Q_j into every slab's table and four reductions of 16 x 16 matrices
    over the slabs.  Against the one-GPU block engine on the same start block: the blocks of T, the Ritz values and
    the basis blocks agree to rounding (the sums over the rows are taken slab by slab); against the numpy model as the
    one-GPU engine is.  A 3-D Laplacian, a random symmetric matrix, and a power-law graph with the regrouped sparse
    product forced on in every slab."""
    from matrices import rmat_laplacian
    from test_gpu_parity import crs_from_scipy
    cases = [("lap3d", la.synth_laplacian(3, 22), 7, {}), ("random", la.synth_random_symmetric(30011, 6, 5), 5, {}),
             ("rmat", crs_from_scipy(-rmat_laplacian(13)), 6, {"LANCZOS_HIP_SPMM_HUB": "2"})]
    for name, a, steps, env in cases:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x0 = start_block(a.dimension, 3)
        with la.BlockEngine(a, steps) as e:
            e.set_start(x0)
            e.run(steps)
            ta, tb = e.tmatrix()
            ritz = e.ritz(4)
            q = [e.get_block(t) for t in (0, 1, steps)]
        with la.BlockMultiEngine(a, steps, slabs, loopback=True) as m:
            assert m.slabs == slabs
            m.set_start(x0)
            m.run(steps - 2)
            m.run(2)                                  # a run goes on where the last one stopped
            assert m.steps == steps
            ma, mb = m.tmatrix()
            mritz = m.ritz(4)
            mq = [m.get_block(t) for t in (0, 1, steps)]
        scale = np.abs(ta).max()
        assert np.abs(ma - ta).max() < 1e-10 * scale and np.abs(mb - tb).max() < 1e-10 * scale, name
        assert rel_err(mritz, ritz) < 1e-10, name
        for got, want in zip(mq, q):
            assert np.abs(got - want).max() < 1e-8, name
        # the basis the slabs hold is orthonormal
        assert np.abs(mq[2].T @ mq[2] - np.eye(16)).max() < 1e-12 and np.abs(mq[0].T @ mq[2]).max() < 1e-12, name
        for k in env:
            monkeypatch.delenv(k)


def test_block_variant_on_row_slabs_by_seed_and_limits(gpu):
    """The start block drawn by seed is the one-GPU engine's (a function of seed and global row); slabs of fewer rows
    than the block width, a full basis and a missing start block are refused."""
    a = la.synth_laplacian(2, 70)
    with la.BlockEngine(a, 4) as e:
        e.set_start_seed(9)
        e.run(4)
        want = e.ritz(3)
    with la.BlockMultiEngine(a, 4, 4, loopback=True) as m:
        with pytest.raises(la.LanczosHipError, match="no start block"):
            m.run(1)
        m.set_start_seed(9)
        m.run(4)
        assert rel_err(m.ritz(3), want) < 1e-10
        with pytest.raises(la.LanczosHipError, match="basis full"):
            m.run(1)
    with pytest.raises(la.LanczosHipError, match="fewer rows than the block width"):
        la.BlockMultiEngine(la.synth_laplacian(1, 40), 2, 3, loopback=True)
    # one slab through librccl (ncclCommInitAll over one device, in-place all-gather, all-reduce): the calls a node makes
    with la.BlockMultiEngine(a, 4, 1) as m:
        m.set_start_seed(9)
        m.run(4)
        assert rel_err(m.ritz(3), want) < 1e-12
