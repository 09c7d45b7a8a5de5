"""Host-side C API of the product (lanczos_amd/host, include/lanczos_host.h and
the host-only parts of include/lanczos_hip.h) against the reference's outputs
and the oracle.  No GPU needed: nothing here launches a kernel."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import lanczos_amd as la
import oracle
from test_oracle_golden import CASES, read_ref_bin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("swap", [0, 1])
def test_ingest_and_convert_from_coo_bit_exact(golden_dir, name, swap):
    ref = read_ref_bin(os.path.join(golden_dir, f"ref_{name}_swap{swap}.bin"))
    coo, dim = la.read_mat_coo_incremental(os.path.join(golden_dir, f"triplets_{name}.txt"))
    assert dim == ref["dim"]
    assert coo.tobytes() == ref["coo"].tobytes()
    a = la.convert_from_coo(coo, dim, swap)
    assert a.values.tobytes() == ref["values"].tobytes()
    assert a.column_index.tobytes() == ref["cols"].tobytes()
    assert a.row_head_indexes.tobytes() == ref["head"].tobytes()
    y = la.matvec_crs(a, ref["x"])
    scale = np.abs(ref["y_crs"]).max() + 1.0
    assert np.max(np.abs(y - ref["y_crs"])) <= 4 * np.finfo(float).eps * scale


def test_convert_from_coo_large_random_matches_oracle():
    rng = np.random.default_rng(5)
    n, m = 5000, 200000
    coo = np.zeros(m, la.COO_DTYPE)
    coo["index_row"] = rng.integers(0, n, m)
    coo["index_column"] = rng.integers(0, n, m)
    coo["value"] = rng.standard_normal(m)
    for swap in (0, 1):
        a = la.convert_from_coo(coo, n, swap)
        v, c, h = oracle.coo_to_crs(coo.view(oracle.COO_DTYPE), n, swap)
        assert a.values.tobytes() == v.tobytes() and a.column_index.tobytes() == c.tobytes()
        assert a.row_head_indexes.tobytes() == h.tobytes()


def _run_py(code):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)


def test_error_behaviour_is_exit_with_message():
    # out-of-range index: stderr message + exit(EXIT_FAILURE), src/crs.c:68-77
    r = _run_py("import numpy as np, lanczos_amd as la\n"
                "c=np.zeros(1,la.COO_DTYPE); c['index_row']=7\n"
                "la.convert_from_coo(c,3,0)\n")
    assert r.returncode == 1 and "convert_from_coo: index out of range at entry 0 (row=7, col=0, dim=3)" in r.stderr
    # malformed line: src/coo.c:19-22
    r = _run_py("import lanczos_amd as la\nla.parse_to_coo('abc 1 2')\n")
    assert r.returncode == 1 and "Error: Invalid integer abc 1 2" in r.stderr
    # missing file: perror + exit, src/coo.c:113-117
    r = _run_py("import lanczos_amd as la\nla.read_mat_coo_incremental('/nonexistent/file.txt')\n")
    assert r.returncode == 1 and "fopen" in r.stderr


def test_parse_to_coo_forms():
    assert la.parse_to_coo("3 4 1.5") == (2, 3, 1.5)
    assert la.parse_to_coo("  7\t9   -2e-3") == (6, 8, -2e-3)
    for line in ("3 4 1.5", " 12   1 1E+2", "5 5 -0.0"):
        rc, r, c, v = oracle.parse_line(line)
        assert (r, c, v) == la.parse_to_coo(line)


def test_tridiag_eigs_against_mkl_and_oracle(golden_dir):
    data = json.load(open(os.path.join(golden_dir, "mkl_dsyevr.json")))
    for case in data["cases"]:
        alpha, beta = np.array(case["alpha"]), np.array(case["beta"])
        n = alpha.size
        eig = la.tridiag_eigs(alpha, beta, n, case["nth"])
        tnorm = np.abs(alpha).max() + 2 * (np.abs(beta).max() if n > 1 else 0.0)
        iu = min(case["nth"], n)
        assert np.max(np.abs(eig[:iu] - np.array(case["eig"]))) <= 64 * np.finfo(float).eps * tnorm
        assert np.all(eig[iu:] == 0.0)
        if case["kind"] == "toeplitz":   # closed form, and relative accuracy of the tiny ones
            exact = 2 - 2 * np.cos(np.pi * np.arange(1, iu + 1) / (n + 1))
            assert np.max(np.abs(eig[:iu] - exact) / exact) < 1e-11


def test_tridiag_eigvecs():
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 60, 250):
        alpha = rng.standard_normal(n) * 2 + 4
        beta = np.abs(rng.standard_normal(max(n - 1, 0))) + 0.05
        m = min(5, n)
        eig = la.tridiag_eigs(alpha, beta, n, m)
        S = la.tridiag_eigvecs(alpha, beta, n, eig)
        T = np.diag(alpha) + np.diag(beta, 1) + np.diag(beta, -1) if n > 1 else np.array([[alpha[0]]])
        tnorm = np.abs(T).sum(1).max()
        for i in range(m):
            assert abs(np.linalg.norm(S[i]) - 1) < 1e-12 and S[i][0] >= 0
            assert np.linalg.norm(T @ S[i] - eig[i] * S[i]) < 1e-11 * tnorm
        assert np.max(np.abs(S @ S.T - np.eye(m))) < 1e-8
    # a cluster: two almost equal eigenvalues still give two independent vectors
    alpha = np.array([1.0, 1.0 + 1e-13, 5.0])
    beta = np.array([1e-14, 1e-14])
    eig = la.tridiag_eigs(alpha, beta, 3, 2)
    S = la.tridiag_eigvecs(alpha, beta, 3, eig)
    assert abs(S[0] @ S[1]) < 1e-6


def test_gaussian_fill_is_partition_invariant_and_normal():
    g = la.gaussian_fill(123456789, 0, 200000)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1) < 0.01
    assert np.array_equal(la.gaussian_fill(123456789, 777, 1000), g[777:1777])
    assert not np.array_equal(la.gaussian_fill(1, 0, 100), g[:100])


def test_synthetic_matrices():
    import scipy.sparse as sp
    for d, N in ((1, 50), (2, 12), (3, 7)):
        a = la.synth_laplacian(d, N)
        n = N ** d
        m = sp.csr_matrix((a.values, a.column_index, a.row_head_indexes), shape=(n, n))
        assert abs(m - m.T).max() == 0 and np.all(m.diagonal() == 2 * d)
        assert np.all(np.diff(a.row_head_indexes) == np.asarray((m != 0).sum(1)).ravel())
        for i in range(n):   # ascending columns inside a row
            seg = a.column_index[a.row_head_indexes[i]:a.row_head_indexes[i + 1]]
            assert np.all(np.diff(seg) > 0)
    r = la.synth_random_symmetric(2000, 10, 42)
    m = sp.csr_matrix((r.values, r.column_index, r.row_head_indexes), shape=(2000, 2000))
    assert abs(m - m.T).max() == 0
    off = abs(m).sum(1).A.ravel() - m.diagonal()
    assert np.allclose(m.diagonal(), 1 + off)
    assert 18 < r.length / 2000 < 22


def _declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#define\b.*?(?<!\\)\n", "\n", text.replace("\\\n", " "))
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)) - {"Matvec_General", "lanczos_hip_matvec_fn"})   # function typedefs


def test_every_declared_symbol_is_exported():
    hip = ctypes.CDLL(la.LIB_HIP)
    host = ctypes.CDLL(la.LIB_HOST)
    names = _declared_functions(os.path.join(ROOT, "include", "lanczos_hip.h"))
    assert "lanczos_hip_crs" in names and "lanczos_hip_create" in names and len(names) >= 20
    for name in names:
        assert hasattr(hip, name), f"{name} declared in lanczos_hip.h but not exported"
    names = _declared_functions(os.path.join(ROOT, "include", "lanczos_host.h"))
    assert "convert_from_coo" in names and "lanczos" in names
    for name in names:
        assert hasattr(host, name), f"{name} declared in lanczos_host.h but not exported"


def test_driver_usage_line():
    exe = os.path.join(ROOT, "lanczos_amd", "bin", "lanczos_main")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1
    assert "<input_file> <coo|crs|crs_hip> <number_of_eigenvalues> <max_iter>" in r.stderr


def test_product_does_not_reach_into_the_oracle():
    """The product tree must not import, link or dlopen anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "lanczos_amd")):
        if os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                if re.search(r"\boracle\b", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_orthogonality_tracker_follows_the_cpu_model():
    """The product's decision rule against the oracle's selective run: fed the oracle's own
    alpha/beta it asks for reorthogonalisation in the same number of steps, in pairs."""
    import oracle
    a = la.synth_laplacian(3, 16)
    v0 = la.gaussian_fill(4, 0, a.dimension)
    steps = 200
    sel = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, steps + 1, 0.0, "selective")
    none = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, steps + 1, 0.0, "none")
    t = la.OmegaTracker(steps)
    fired = []
    for j in range(steps):
        if j > 0:
            t.set_beta(j - 1, sel.beta[j - 1])
        # sel.beta[j] is the norm after a possible reorthogonalisation, which differs from the
        # norm before it by O(eps): the decisions are the same
        need, est = t.step(j, sel.alpha[j], sel.beta[j])
        if need:
            fired.append(j)
    assert t.reorths == sel.reorth_steps == len(fired) and len(fired) >= 2
    assert all(fired[i + 1] == fired[i] + 1 for i in range(0, len(fired) - 1, 2))   # pairs
    with pytest.raises(la.LanczosHipError, match="out of order"):
        t.step(5, 1.0, 1.0)
    # without reorthogonalisation the estimate bounds the real loss: first firing precedes it
    assert none.steps == steps


def test_small_dense_symmetric_eigen_solvers_match_numpy():
    """Host solvers behind the block variant and the thick restart (the reference calls MKL's
    LAPACKE_dsyevr on its T, src/util.c:118-272): against numpy's LAPACK."""
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 40, 150):
        a = rng.standard_normal((n, n))
        a = a + a.T
        want = np.linalg.eigvalsh(a)
        nth = min(n, 6)
        assert np.max(np.abs(la.sym_lowest_eigs(a, nth) - want[:nth])) < 1e-13 * max(1.0, np.abs(want).max()) * n
        w, v = la.sym_eig(a)
        assert np.max(np.abs(w - want)) < 1e-12 * max(1.0, np.abs(want).max())
        assert np.max(np.abs(v @ v.T - np.eye(n))) < 1e-12
        assert np.max(np.abs(v @ a @ v.T - np.diag(w))) < 1e-11 * max(1.0, np.abs(want).max())
    # arrowhead joined to a tridiagonal: the projected matrix after a thick restart
    n, p = 50, 12
    t = np.diag(rng.standard_normal(n))
    t[p, :p] = t[:p, p] = rng.standard_normal(p)
    for i in range(p, n - 1):
        t[i, i + 1] = t[i + 1, i] = rng.standard_normal()
    w, v = la.sym_eig(t)
    assert np.max(np.abs(w - np.linalg.eigvalsh(t))) < 1e-12


def test_block_model_reaches_the_closed_form_spectrum():
    """The numpy model of the block variant on a 2-D grid: orthonormal basis, T = V^T A V, and the
    lowest values (the double one twice) to the closed form."""
    from oracle import block_lanczos as bl
    from matrices import laplacian_eigs
    a = la.synth_laplacian(2, 20)
    x0 = la.gaussian_fill(3, 0, a.dimension * 16).reshape(-1, 16)
    ab, bb, basis = bl.block_lanczos(a.row_head_indexes, a.column_index, a.values, x0, 22)
    v = np.hstack(basis[:22])
    assert np.max(np.abs(v.T @ v - np.eye(v.shape[1]))) < 1e-12
    t = bl.assemble_t(ab, bb, 22)
    av = bl.spmm(a.row_head_indexes, a.column_index, a.values, v)
    assert np.max(np.abs(v.T @ av - t)) < 1e-11
    assert np.max(np.abs(bl.ritz(ab, bb, 22, 3) - laplacian_eigs(2, 20, 3))) < 1e-9
    assert np.max(np.abs(la.sym_lowest_eigs(t, 5) - np.linalg.eigvalsh(t)[:5])) < 1e-13


def test_matrix_market_reader_against_scipy(tmp_path):
    """The Matrix Market extension of the host library: general / symmetric / skew-symmetric / pattern
    files with comments and blank lines, against scipy.io.mmread; malformed files are refused."""
    import scipy.io
    import scipy.sparse as sp

    rng = np.random.default_rng(5)
    n = 37
    dense = sp.random(n, n, density=0.15, random_state=7, format="coo")
    sym = (dense + dense.T).tocoo()
    skew = (sp.triu(dense, 1) - sp.triu(dense, 1).T).tocoo()
    cases = {"general": (dense, {}), "symmetric": (sym, {"symmetry": "symmetric"}),
             "skew": (skew, {"symmetry": "skew-symmetric"}),
             "pattern": (sym, {"field": "pattern", "symmetry": "symmetric"}),
             "integer": (sp.coo_matrix(np.triu(rng.integers(-3, 4, (n, n)))), {"field": "integer"})}
    for name, (m, kw) in cases.items():
        path = str(tmp_path / f"{name}.mtx")
        scipy.io.mmwrite(path, m, comment="written by the test\nsecond comment line", **kw)
        text = open(path).read().replace("\n", "\n\n", 3)       # blank lines must not matter
        open(path, "w").write(text)
        assert la.is_matrix_market(path)
        coo, dim = la.read_matrix_market(path)
        want = scipy.io.mmread(path).toarray()
        got = np.zeros((dim, dim))
        np.add.at(got, (coo["index_row"], coo["index_column"]), coo["value"])
        assert dim == n and np.array_equal(got, want), name
        # and on through the bit-exact builder to the product
        crs = la.convert_from_coo(coo, dim, 0)
        x = rng.standard_normal(n)
        assert np.allclose(la.matvec_crs(crs, x), want @ x, rtol=0, atol=1e-13)
    plain = str(tmp_path / "plain.txt")
    open(plain, "w").write("1 1 2.0\n1 2 -1.0\n2 1 -1.0\n2 2 2.0\n")
    assert not la.is_matrix_market(plain)
    for bad in ("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n",
                "%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1 0\n",
                "%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n",
                "%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n",
                "%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1.0\n2 2 1.0\n",
                "%%MatrixMarket matrix coordinate real general\n"):
        path = str(tmp_path / "bad.mtx")
        open(path, "w").write(bad)
        with pytest.raises(ValueError):
            la.read_matrix_market(path)


def test_bench_builds_the_launch_line_for_several_gpus():
    """bench.py --gpus N without WORLD_SIZE starts `python -m torch.distributed.run` itself."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launch_command(["--gpus", "8", "--steps", "20", "--warmup", "3"], 8, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]


def test_bench_result_line_outlives_a_crash_of_the_process():
    """bench.py hands its line to a process of its own: the last complete line published is printed,
    exactly once, whether the run returns or aborts inside a library."""
    import subprocess
    import sys
    prog = (
        "import os, sys, importlib.util\n"
        f"spec = importlib.util.spec_from_file_location('bench_mod', {os.path.join(ROOT, 'bench.py')!r})\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "fd = os.dup(1); os.dup2(2, 1)\n"
        "k = b.LineKeeper(fd); os.close(fd)\n"
        "k.publish({'metric': 'm', 'value': 1})\n"
        "k.publish({'metric': 'm', 'value': 2, 'pad': 'x' * 200000})\n"
        "print('noise on stdout')\n"
        "if sys.argv[1] == 'abort': os.abort()\n"
        "k.publish({'metric': 'm', 'value': 3}); k.close()\n")
    for how, want, rc_ok in (("abort", 2, False), ("return", 3, True)):
        p = subprocess.run([sys.executable, "-c", prog, how], capture_output=True, text=True, timeout=60)
        assert (p.returncode == 0) == rc_ok
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, p.stdout[:300]
        import json
        assert json.loads(lines[0])["value"] == want


def test_random_symmetric_rows_are_the_rows_of_the_whole_matrix():
    a = la.synth_random_symmetric(5000, 10, 42)
    h = a.row_head_indexes
    for r0, r1 in ((0, 5000), (0, 1234), (1234, 4000), (4000, 5000)):
        s = la.synth_random_symmetric_rows(5000, 10, 42, r0, r1)
        assert np.array_equal(s.row_head_indexes, h[r0:r1 + 1] - h[r0])
        assert np.array_equal(s.column_index, a.column_index[h[r0]:h[r1]])
        assert np.array_equal(s.values, a.values[h[r0]:h[r1]])


def test_host_c_code_under_the_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The host-side C code (triplet and Matrix Market readers, COO->CRS build, host matvecs, generators,
    eigen-solvers, orthogonality tracker, start vector) built with -fsanitize=address,undefined and walked
    by lanczos_amd/host/host_check.c over small awkward inputs: no report from either sanitizer, and the
    closed-form checks of the driver hold.  (The reference's build has no sanitizer target, Makefile:3.)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mk = subprocess.run(["make", "-C", os.path.join(root, "lanczos_amd", "csrc"), "host_check"], capture_output=True, text=True)
    assert mk.returncode == 0, mk.stdout[-2000:] + mk.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:exitcode=99")
    r = subprocess.run([os.path.join(root, "lanczos_amd", "bin", "host_check"), str(tmp_path)], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host_check: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
