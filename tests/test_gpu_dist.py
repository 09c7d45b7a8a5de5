"""Row-partitioned engines on real kernels.  RCCL refuses two ranks on one GPU,
so two ranks share cuda:0 here and move their payloads through host memory with
gloo (lanczos_amd.dist.StagedComm); everything else -- slab upload, remapped
columns, x_ext gather buffer, phase-level C ABI, iteration order -- is the code
bench.py runs on several GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, kind, mode, reorth, steps, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lanczos_amd as la
    from lanczos_amd import dist as ld
    torch.cuda.set_device(0)
    if kind.startswith("lap3d"):
        N = 48 if kind == "lap3d48" else 40   # 48: planes of 2304 rows lie beyond the window: the slabs walk them
        n = N ** 3
        bounds = ld.partition_rows(n, world)
        a_local = la.synth_laplacian_rows(3, N, bounds[rank], bounds[rank + 1])
        full = la.synth_laplacian(3, N) if rank == 0 else None
    else:
        # "random_big": rows reach across more of x than the caches hold, so the slabs take the tiled SpMV
        full = la.synth_random_symmetric(2_600_000 if kind == "random_big" else 30000, 8, 3)
        n = full.dimension
        bounds = ld.partition_rows_by_nnz(full.row_head_indexes, world)
        h = full.row_head_indexes
        r0, r1 = bounds[rank], bounds[rank + 1]
        a_local = la.CrsMatrix(full.values[h[r0]:h[r1]], full.column_index[h[r0]:h[r1]], h[r0:r1 + 1] - h[r0])
    comm = ld.StagedComm(dist, torch, torch.device("cuda", 0))
    runner = ld.build_gpu_rank(la, torch, comm, a_local, bounds, steps + 1, reorth=reorth, mode=mode)
    v0 = la.gaussian_fill(99, 0, n)
    runner.set_start(v0[bounds[rank]:bounds[rank + 1]])
    runner.run(steps)
    alpha, beta = runner.history()
    if rank == 0:
        with la.Engine(full, steps + 1, reorth=reorth) as e:
            e.set_start(v0)
            a1, b1 = e.run(steps)
        q.put((runner.plan.mode, float(np.max(np.abs(alpha - a1) / np.abs(a1))),
               float(np.max(np.abs(beta - b1) / np.abs(b1))), runner.engine.e.spmv_name,
               runner.interior is not None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,mode,reorth,expect", [
    ("lap3d", "auto", "full", "halo"),
    ("lap3d", "allgather", "none", "allgather"),
    ("lap3d", "auto", "full2", "halo"),
    ("random", "auto", "full", "allgather"),
    ("lap3d", "auto", "none", "halo"),
    ("lap3d", "auto", "selective", "halo"),
    ("lap3d", "auto", "cuda_compat", "halo"),
    ("random_big", "auto", "none", "allgather"),
    ("lap3d48", "auto", "full", "halo"),
    ("lap3d48", "auto", "none", "halo"),
])
def test_two_ranks_on_one_gpu_match_one_engine(gpu, kind, mode, reorth, expect):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, mode, reorth, 30, q)) for r in range(2)]
    for p in procs:
        p.start()
    chosen, ea, eb, kernel, overlapped = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert chosen == expect
    assert ea < 1e-11 and eb < 1e-11, (ea, eb)
    if kind.startswith("lap3d") and expect == "halo":
        assert kernel.startswith("dict") and overlapped      # split product, remote entries in the vectors' tails
    if kind == "lap3d48":
        assert kernel.endswith("p")    # plane walk, the neighbour's plane (in the vector's tail) before the first / after the last
    if kind == "random_big":
        assert kernel.startswith("panel")


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_step_as_two_plane_walks(gpu, monkeypatch, overlap):
    """Without reorthogonalisation a slab of whole planes steps as two plane walks like one GPU does (from 2^21 rows by
    default; forced here): the first walk beside the exchange, its edge rows' share added when the planes have arrived,
    the second walk in place of the recurrence kernel.  Against one engine running product + recurrence kernel."""
    monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", "1")      # inherited by the spawned ranks; the one engine they compare
    if not overlap:                                       # with runs the two walks too (they match the kernels to 1e-13:
        monkeypatch.setenv("LANCZOS_DIST_NO_OVERLAP", "1")  # tests/test_gpu_parity.py)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "lap3d48", "auto", "none", 30, q)) for r in range(2)]
    for p in procs:
        p.start()
    chosen, ea, eb, kernel, overlapped = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert chosen == "halo" and overlapped == overlap and kernel.endswith("p")
    assert ea < 1e-11 and eb < 1e-11, (ea, eb)


@pytest.mark.parametrize("switch,reorth", [("LANCZOS_DIST_NO_INPLACE", "none"), ("LANCZOS_DIST_NO_INPLACE", "full"),
                                           ("LANCZOS_DIST_NO_OVERLAP", "full")])
def test_two_ranks_with_the_other_exchange_forms(gpu, monkeypatch, switch, reorth):
    """The gather-buffer form of the halo exchange, and exchange-then-product, stay selectable for
    comparisons on a real node: same results."""
    monkeypatch.setenv(switch, "1")      # inherited by the spawned ranks
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, "lap3d", "auto", reorth, 30, q)) for r in range(2)]
    for p in procs:
        p.start()
    chosen, ea, eb, kernel, overlapped = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert chosen == "halo" and not overlapped
    assert ea < 1e-11 and eb < 1e-11, (ea, eb)


def test_bench_distributed_code_path_with_one_rank(gpu):
    """bench.py's N > 1 path (nccl process group, halo/allgather plumbing) with world size 1."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--workload", "lap64", "--steps", "40",
                        "--warmup", "2", "--big-grid", "96", "--rand-rows", "200000", "--converge-steps", "60"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["steps"] == 40 and out["value"] > 0
    assert 0 < out["spmv_hbm_GBps"] <= 8000.0 and out["spmv_hbm_kernel"]   # the metric's second half: achieved SpMV GB/s
    assert out["strong_scaling_laplacian"]["iters_per_s"] > 0 and out["random_symmetric"]["iters_per_s"] > 0
    assert out["converged_run"]["steps"] == 60 and out["converged_run"]["seconds"] > 0
    inproc = out["in_process_rccl"]                    # rank 0 driving the slabs through the library's own RCCL calls
    assert inproc["gpus"] == 1 and inproc["full"]["iters_per_s"] > 0 and inproc["none"]["iters_per_s"] > 0
    assert inproc["full"]["ritz_rel_diff_vs_primary"] < 1e-10
    import lanczos_amd as la
    a = la.synth_laplacian(3, 64)
    with la.Engine(a, 41) as e:
        e.set_start(la.gaussian_fill(123456789, 0, a.dimension))
        alpha, beta = e.run(40)
    want = la.ritz_values(alpha, beta, 40, 4)
    assert np.max(np.abs(np.array(out["ritz_lowest4"]) - want) / want) < 1e-10


def test_bench_gpus_flag_launches_its_own_ranks(gpu):
    """`python bench.py --gpus N` with no launcher around it starts the ranks itself (a child
    torch.distributed.run, before this process touches the GPU) and relays rank 0's line; rehearsed
    with one rank, the most a one-GPU box can hold."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--workload", "lap64", "--steps", "20",
                        "--warmup", "2", "--big-grid", "0", "--rand-rows", "100000", "--converge-steps", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, lines                      # the one JSON line and nothing else
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["value"] > 0
    assert out["config"]["parallelism"] == "row-partition x1"
    assert out["random_symmetric"]["iters_per_s"] > 0


def test_caller_owned_basis_is_validated(gpu):
    """options.basis: leading dimension and column numbering are checked before anything is launched."""
    import torch
    import lanczos_amd as la
    a = la.synth_laplacian(1, 3000)
    base = la.gather_base(3000)
    assert base == 4096
    basis = torch.zeros(5, base + 2048, dtype=torch.float64, device="cuda")
    with pytest.raises(la.LanczosHipError, match="basis_ld"):
        la.Engine(a, 5, n_global=3000, n_ext=base, basis=basis.data_ptr(), basis_ld=base - 2)
    with pytest.raises(la.LanczosHipError, match="basis_ld"):
        la.Engine(a, 5, n_global=3000, n_ext=base + 10, basis=basis.data_ptr(), basis_ld=base + 2047)
    cols = a.column_index.copy()
    cols[5] = 3500                                   # inside the zero padding [n, gather base)
    bad = la.CrsMatrix(a.values, cols, a.row_head_indexes)
    with pytest.raises(la.LanczosHipError, match="padding"):
        la.Engine(bad, 5, n_global=3000, n_ext=base + 10, basis=basis.data_ptr(), basis_ld=base + 2048)
    with pytest.raises(la.LanczosHipError, match="exclude"):
        la.Engine(a, 5, n_global=3000, n_ext=base, x_ext=basis.data_ptr(), basis=basis.data_ptr(), basis_ld=base)
    # a well-formed one: no remote entries, the engine works in the caller's memory
    with la.Engine(a, 5, reorth="full", n_global=3000, n_ext=base, basis=basis.data_ptr(), basis_ld=base + 2048) as e:
        v0 = la.gaussian_fill(3, 0, 3000)
        e.set_start_scaled(v0, 1.0 / float(np.linalg.norm(v0)))
        red = torch.zeros(16, dtype=torch.float64, device="cuda")
        e.phase_spmv(red.data_ptr())
        torch.cuda.synchronize()
        w = basis[1, :3000].cpu().numpy()
        want = la.matvec_crs(a, v0 / np.linalg.norm(v0))
        assert np.max(np.abs(w - want)) < 1e-14
        assert abs(float(red[0]) - float(want @ (v0 / np.linalg.norm(v0)))) < 1e-12
        assert not basis[:, 3000:].any()


@pytest.mark.parametrize("kind", ["auto", "rows", "stream", "vector", "pipe"])
def test_split_product_matches_the_whole_one(gpu, kind):
    """phase_spmv_begin / phase_spmv_end (interior rows first, the rest after the exchange) against
    phase_spmv on the same vector: same y up to the order of a row's sum, same u.Au."""
    import torch
    import lanczos_amd as la
    a = la.synth_laplacian(3, 24)
    n = a.dimension
    base = la.gather_base(n)
    basis = torch.zeros(3, base + 2048, dtype=torch.float64, device="cuda")
    v0 = la.gaussian_fill(11, 0, n)
    v0 /= np.linalg.norm(v0)
    red = torch.zeros(16, dtype=torch.float64, device="cuda")
    with la.Engine(a, 5, reorth="none", spmv=kind, n_global=n, n_ext=base, basis=basis.data_ptr(),
                   basis_ld=base + 2048) as e:
        assert e.spmv_name.startswith("dict" if kind == "auto" else kind) and e.can_overlap
        e.set_start_scaled(v0, 1.0)
        e.phase_spmv(red.data_ptr())
        torch.cuda.synchronize()
        whole, dot = basis[1, :n].cpu().numpy().copy(), float(red[0])
        basis[1].zero_()
        e.phase_spmv_begin(600, n - 1000)
        e.phase_spmv_end(red.data_ptr())
        torch.cuda.synchronize()
        split = basis[1, :n].cpu().numpy()
        assert np.array_equal(split[640:n - 1000], whole[640:n - 1000])
        assert np.max(np.abs(split - whole)) < 1e-14
        assert abs(float(red[0]) - dot) < 1e-13 * abs(dot)
        # an empty edge on either side
        e.phase_spmv_begin(0, n)
        e.phase_spmv_end(red.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(basis[1, :n].cpu().numpy(), whole) and abs(float(red[0]) - dot) < 1e-13 * abs(dot)
        # no interior at all: every row through the edge kernel
        e.phase_spmv_begin(700, 700)
        e.phase_spmv_end(red.data_ptr())
        torch.cuda.synchronize()
        assert np.max(np.abs(basis[1, :n].cpu().numpy() - whole)) < 1e-14 and abs(float(red[0]) - dot) < 1e-13 * abs(dot)
        with pytest.raises(la.LanczosHipError, match="row range"):
            e.phase_spmv_begin(10, 5)


def test_split_product_on_random_banded_matrices(gpu):
    """Random sizes, row lengths, band widths and interior ranges through every SpMV form that can be
    split: the two-part product equals the one-part product row by row."""
    import torch
    import lanczos_amd as la
    rng = np.random.default_rng(314)
    red = torch.zeros(16, dtype=torch.float64, device="cuda")
    for case in range(40):
        n = int(rng.choice([65, 129, 1000, 4097, int(rng.integers(200, 30000))]))
        if case % 4 == 0:                       # a stencil, so that the dictionary form takes it
            a = la.synth_laplacian(1, n) if case % 8 else la.synth_laplacian(2, int(np.sqrt(n)) + 2)
            n = a.dimension
            kinds = ["auto"]
        else:
            lens = rng.integers(0, 9, n)
            lens[rng.integers(0, n)] = int(rng.integers(1, 700))          # one long row
            head = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            w = int(rng.integers(1, 300))
            rows = np.repeat(np.arange(n), lens)
            cols = np.clip(rows + rng.integers(-w, w + 1, head[-1]), 0, n - 1).astype(np.int64)
            a = la.CrsMatrix(rng.standard_normal(head[-1]), cols, head)
            kinds = ["stream", "vector", "pipe"] + (["rows"] if lens.max() <= 8 else [])
        base = la.gather_base(n)
        basis = torch.zeros(3, base + 2048, dtype=torch.float64, device="cuda")
        v0 = rng.standard_normal(n)
        for kind in kinds:
            try:
                e = la.Engine(a, 5, reorth="none", spmv=kind, n_global=n, n_ext=base, basis=basis.data_ptr(),
                              basis_ld=base + 2048)
            except la.LanczosHipError as ex:
                assert "needs" in str(ex)
                continue
            with e:
                if not e.can_overlap:
                    continue
                e.set_start_scaled(v0, 1.0)
                e.phase_spmv(red.data_ptr())
                torch.cuda.synchronize()
                whole, dot = basis[1, :n].cpu().numpy().copy(), float(red[0])
                lo = int(rng.integers(0, n + 1))
                hi = int(rng.integers(lo, n + 1))
                basis[1].zero_()
                e.phase_spmv_begin(lo, hi)
                e.phase_spmv_end(red.data_ptr())
                torch.cuda.synchronize()
                split = basis[1, :n].cpu().numpy()
                scale = np.abs(whole).max() + 1e-300
                assert np.max(np.abs(split - whole)) <= 1e-13 * scale, (case, kind, n, lo, hi)
                assert abs(float(red[0]) - dot) <= 1e-12 * (np.abs(whole) @ np.abs(v0) + 1e-300), (case, kind)
                assert not basis[1, n:].any()
