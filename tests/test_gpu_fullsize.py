"""Full-size checks on the benchmark configuration (3-D 7-point Laplacian 256^3,
n = 16 777 216, nnz = 117 047 296) through size-independent properties, plus a
short direct comparison with the CPU restatement."""
import os

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


def test_bench_line_keeps_its_contract(gpu):
    """bench.py at N = 1 (shortened): exactly one JSON line on stdout with the agreed keys."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "2",
                        "--cpu-steps", "6", "--rand-rows", "300000", "--big-grid", "128", "--converge-steps", "40"], capture_output=True, text=True, timeout=600,
                       cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2 and d["dtype"] == "f64"
    assert d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["gpu_same_sample_rel_err"]["alpha"] < 1e-10 and cb["gpu_same_sample_rel_err"]["beta"] < 1e-10
    assert d["spmv_vector_step"]["iters_per_s"] > 0 and d["random_symmetric"]["iters_per_s"] > 0
    assert d["selective_reorth"]["ritz_rel_diff_vs_full"] < 1e-10
    assert d["strong_scaling_laplacian"]["iters_per_s"] > 0
    assert d["converged_run"]["steps"] == 40 and d["converged_run"]["seconds"] > 0
