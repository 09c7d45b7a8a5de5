#!/usr/bin/env python3
"""Benchmark of the Lanczos hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload lap256] [--reorth full]

One "step" is one Lanczos iteration (SpMV, alpha, three-term recurrence,
reorthogonalisation against the resident basis, beta, normalisation) of the
workload; the timed region is a fresh run of exactly K iterations from the
start vector with the matrix, the basis and every scalar resident in HBM.
Rank 0 prints one JSON line.  For N > 1 the matrix is row-partitioned over the
ranks (strong scaling, one process per GPU, collectives through
torch.distributed / RCCL).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak, GB/s

WORKLOADS = {
    # name: (dimension d, grid N)
    "lap256": (3, 256),   # 3-D 7-point Laplacian 256^3: n = 16 777 216, nnz = 117 047 296
    "lap512": (3, 512),   # 512^3: n = 134 217 728, nnz = 937 951 232
    "lap128": (3, 128),
    "lap64": (3, 64),
}


# Rate at which an XCD's L2 was measured to serve whole 128-byte lines to gathers, chip-wide
# (hardware guide, "Indexed rows: gather into LDS": 16.8-18.8 TB/s of 128-byte lines)
L2_GATHER_LINES_PER_S = (16.8e12 / 128, 18.8e12 / 128)


def algorithmic_bytes(n: int, nnz: int) -> dict:
    """The shared per-iteration figures on the CRS format (4-byte columns): each operand read once, each
    result written once.  `vec_charged` = alpha dot 16 n + recurrence 32 n + scaling 16 n, what a step
    with separate kernels would move; `vec_moved` = the 32 n the fused deferred-normalisation
    recurrence of the SpMV + vector step really moves (alpha comes out of the SpMV, no scaling pass)."""
    wp = 4 if nnz < 2**31 else 8
    b_spmv = 12 * nnz + wp * (n + 1) + 16 * n
    return {"spmv": b_spmv, "vec_charged": 64 * n, "vec_moved": 32 * n, "iter_charged": b_spmv + 64 * n,
            "iter_moved": b_spmv + 32 * n}


def gather_ceiling(kind: str, kernel: str, n: int, nnz: int, spmv_seconds: float):
    """The gather-bound SpMV forms against the L2s' measured line rate instead of the HBM roofline:
    L2 requests per SpMV = nnz x the requests-per-entry ratio that the PMC passes counted for this
    form (TCC_REQ_sum, profiles/gather_requests.json), over the SpMV's time.  None without a record,
    for any other kernel than the tiled form the record was counted on (a matrix whose x fits the
    caches runs on the row-wise forms, whose gathers mostly hit in L1), and for a vector of less than
    32 MiB (the L2s' total: below it the counted ratio of requests per entry does not hold)."""
    if not kernel.startswith("panel") or 8 * n < 32 * 2**20:
        return None
    try:
        rec = json.load(open(os.path.join(HERE, "profiles", "gather_requests.json")))[kind]
    except (OSError, ValueError, KeyError):
        return None
    req_s = rec["l2_requests_per_entry"] * nnz / spmv_seconds
    lo, hi = L2_GATHER_LINES_PER_S
    return {"l2_requests_per_entry": rec["l2_requests_per_entry"], "counted_on": rec["counted_on"],
            "l2_requests_per_s": req_s, "l2_line_rate_measured_per_s": [lo, hi],
            "frac_of_l2_line_rate": req_s / hi, "frac_of_l2_line_rate_low_end": min(req_s / lo, 1.0)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="lap256", choices=sorted(WORKLOADS))
    ap.add_argument("--reorth", default="full", choices=["full", "full2", "mgs", "cuda_compat", "none", "selective"])
    ap.add_argument("--spmv", default="auto", choices=["auto", "vector", "stream", "pipe", "pipe_nt", "rows", "panel", "dict", "merge"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-block", action="store_true", help="skip the block-variant section")
    ap.add_argument("--rand-rows", type=int, default=10_000_000,
                    help="rows of the random symmetric matrix (~21 entries per row) of the secondary section; "
                         "50000000 is the full-size configuration, 0 skips the section")
    ap.add_argument("--rmat-scale", type=int, default=24,
                    help="scale of the R-MAT graph Laplacian section (n = 2^scale; 26 is the full configuration; 0 skips it)")
    ap.add_argument("--big-grid", type=int, default=512,
                    help="grid size N of the N^3 Laplacian of the strong-scaling section (SpMV + vector step, "
                         "no reorthogonalisation: the basis of 300 vectors would not fit one GPU); 0 skips it")
    ap.add_argument("--converge-steps", type=int, default=1200,
                    help="steps of the time-to-solution section (selective reorthogonalisation, whole basis "
                         "resident: 1200 vectors of the 256^3 problem are 161 GB); 0 skips it")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the row-partitioned code path even with one rank (rehearsal)")
    ap.add_argument("--no-in-process", action="store_true",
                    help="multi-rank runs: skip the section in which rank 0 alone drives all GPUs through the "
                         "library's own RCCL calls")
    ap.add_argument("--main-only", action="store_true",
                    help="only the timed region and its warm-up (for profiler passes): no secondary sections")
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU sample length; 0 = size it for ~15 s")
    return ap.parse_args()


def cpu_baseline(a, v0, steps: int) -> dict:
    """Time the CPU restatement of the reference loop (oracle/) on a bounded
    sample: the first `steps` iterations of the same matrix and start vector,
    full reorthogonalisation, all host threads OpenMP gives it.  With steps <= 0
    a 16-step probe sizes the sample for 10-30 s of CPU work (the reference's
    cost per step grows linearly with the step index)."""
    import oracle

    def run(s):
        t0 = time.perf_counter()
        r = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, s + 1, 0.0, "full")
        return r, time.perf_counter() - t0

    if steps <= 0:
        probe, _ = run(16)
        per_spmv = probe.timing["spmv"] / 16
        per_pair = probe.timing["reorthogonalization"] / (16 * 17 / 2)   # ~ sum_k (k + 1) vector passes
        target, steps = 30.0, 16
        while steps < 300 and per_spmv * (steps + 8) + per_pair * (steps + 8) * (steps + 9) / 2 < target:
            steps += 8
    r, wall = run(steps)
    loop = r.timing["spmv"] + r.timing["diagonalization"] + r.timing["reorthogonalization"]
    return {
        "value": steps / loop,
        "unit": "iters/s",
        "cores": oracle.num_threads(),
        "kind": "port",
        "sample": f"first {steps} iterations (k < {steps}) of the same matrix and start vector, "
                  f"full reorthogonalisation, threshold 0; loop {loop:.2f} s of {wall:.2f} s wall "
                  f"(spmv {r.timing['spmv']:.2f} s, reorth {r.timing['reorthogonalization']:.2f} s)",
        "steps": steps,
        "ritz": [float(x) for x in r.eigenvalues],
        "alpha": [float(x) for x in r.alpha[:steps]],
        "beta": [float(x) for x in r.beta[:steps]],
    }


def measured_traffic(kernel_class: str, workload: str, reorth: str, algorithmic_bytes_per_launch: float):
    """HBM bytes per launch of one kernel class: the ratio (measured bytes / algorithmic bytes) that the
    PMC passes gave for that class (FETCH_SIZE and WRITE_SIZE collected by rocprofv3 in runs of their
    own over this command, condensed into profiles/traffic.json by scripts/summarize_profile.py) times
    the algorithmic bytes of this run's launches.  The ratio does not depend on the number of steps
    (the reorthogonalisation kernels read the k basis vectors once whatever k is), so the record holds
    for any --steps.  None when no measurement exists for this workload and reorthogonalisation mode."""
    try:
        rec = json.load(open(os.path.join(HERE, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    if (rec.get("workload"), rec.get("reorth")) != (workload, reorth):
        return None
    ratio = rec.get("measured_over_algorithmic", {}).get(kernel_class)
    return None if ratio is None else ratio * algorithmic_bytes_per_launch


def block_variant(la, a, steps: int = 8) -> dict:
    """The block variant (16 vectors per step, fp64 MFMA contractions) on the same matrix: a few block
    steps, per-kernel achieved GB/s from HIP events, and the matrix-core rate of the Gram kernel."""
    n = a.dimension
    with la.BlockEngine(a, steps, profile=True) as e:
        e.set_start_seed(123456789)
        e.profile_reset()
        t0 = time.perf_counter()
        e.run(steps)
        dt = time.perf_counter() - t0
        prof = e.profile()
    kernels = {}
    for k, v in prof.items():
        if v["launches"] and v["ms"] > 0:
            kernels[k] = {"ms_total": v["ms"], "launches": v["launches"],
                          "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9}
    # Gram: 2 * n * 16 * 16 flops per basis block streamed; bytes counted are 128 n per block
    gram_blocks = prof["gram"]["bytes"] / (8.0 * 16 * n)
    return {
        "block_width": 16,
        "block_steps": steps,
        "ms_per_block_step": dt / steps * 1e3,
        "vector_steps_per_s": 16 * steps / dt,
        "kernels": kernels,
        "gram_mfma_TFLOPs": 2.0 * n * 256 * gram_blocks / (prof["gram"]["ms"] * 1e-3) / 1e12 if prof["gram"]["ms"] else 0,
    }


def random_symmetric_section(la, torch, rows: int, steps: int = 100) -> dict:
    """Irregular rows: random symmetric matrix, 10 draws per row mirrored (about 21 entries per row),
    diagonally dominant.  The SpMV + vector step (no reorthogonalisation) and the SpMV alone, against
    the same algorithmic bytes as the Laplacian figures."""
    t0 = time.perf_counter()
    a = la.synth_random_symmetric(rows, 10, 42)
    t_gen = time.perf_counter() - t0
    n, nnz = a.dimension, a.length
    b = algorithmic_bytes(n, nnz)
    with la.Engine(a, 4, reorth="none", profile=True) as e:
        e.set_start_seed(123456789)
        e.run(5)
        e.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = e.profile()
        name = e.spmv_name
    spmv_gbps = prof["spmv"]["bytes"] / (prof["spmv"]["ms"] * 1e-3) / 1e9
    return {
        "workload": f"random symmetric n={n}, nnz={nnz} ({nnz / n:.1f} per row), seed 42",
        "steps": steps,
        "iters_per_s": steps / dt,
        "ms_per_iter": dt / steps * 1e3,
        # CRS algorithmic bytes + the 32 n the recurrence moves: what the step must at least move
        "algorithmic_bytes_per_iter": b["iter_moved"],
        "frac_of_hbm_peak": b["iter_moved"] * steps / dt / 1e9 / HBM_PEAK_GBPS,
        "spmv_kernel": name,
        "spmv_avg_ms": prof["spmv"]["ms"] / steps,
        "spmv_GBps": spmv_gbps,
        "spmv_frac_of_hbm_peak": spmv_gbps / HBM_PEAK_GBPS,
        # one L2 request per gather: the ceiling this kernel runs against
        "gather_ceiling": gather_ceiling("random_symmetric", name, n, nnz, prof["spmv"]["ms"] * 1e-3 / steps),
        "generate_s": t_gen,
    }


def power_law_section(la, torch, scale: int, steps: int = 50, block_steps: int = 3) -> dict:
    """Power-law rows: the graph Laplacian of an R-MAT graph (edge factor 16, a/b/c = 0.57/0.19/0.19),
    drawn on the GPU.  The SpMV + vector step of the single-vector path, and block steps of the 16-column
    variant (fp64 matrix-core reorthogonalisation) on the same matrix."""
    from lanczos_amd.synth_gpu import rmat_laplacian_gpu
    t0 = time.perf_counter()
    a = la.CrsMatrix(*rmat_laplacian_gpu(scale))
    t_gen = time.perf_counter() - t0
    n, nnz = a.dimension, a.length
    b = algorithmic_bytes(n, nnz)
    with la.Engine(a, 4, reorth="none", profile=True) as e:
        e.set_start_seed(123456789)
        e.run(3)
        e.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = e.profile()
        name = e.spmv_name
    out = {
        "workload": f"R-MAT scale {scale} graph Laplacian, n={n}, nnz={nnz}, longest row "
                    f"{int(np.diff(a.row_head_indexes).max())}",
        "steps": steps,
        "iters_per_s": steps / dt,
        "ms_per_iter": dt / steps * 1e3,
        "algorithmic_bytes_per_iter": b["iter_moved"],
        "frac_of_hbm_peak": b["iter_moved"] * steps / dt / 1e9 / HBM_PEAK_GBPS,
        "spmv_kernel": name,
        "spmv_avg_ms": prof["spmv"]["ms"] / steps,
        "spmv_G_entries_per_s": nnz * steps / (prof["spmv"]["ms"] * 1e-3) / 1e9,
        "gather_ceiling": gather_ceiling("power_law_graph", name, n, nnz, prof["spmv"]["ms"] * 1e-3 / steps),
        "generate_s": t_gen,
    }
    if block_steps > 0:
        with la.BlockEngine(a, block_steps, profile=True) as be:
            be.set_start_seed(1)
            be.run(1)
            be.set_start_seed(1)
            be.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            be.run(block_steps)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            pb = be.profile()
        out["block_variant"] = {"block_steps": block_steps, "ms_per_block_step": dtb / block_steps * 1e3,
                                "kernel_ms_per_step": {k: v["ms"] / block_steps for k, v in pb.items()}}
    return out


def big_laplacian_section(la, torch, N: int, steps: int = 100) -> dict:
    """SpMV + vector step on the N^3 7-point Laplacian (N = 512: n = 134 217 728, nnz = 937 951 232)
    on one GPU: the single-GPU leg of the strong-scaling figure that `--gpus 2/4/8` report on the same
    matrix, row-partitioned."""
    t0 = time.perf_counter()
    a = la.synth_laplacian(3, N)
    t_gen = time.perf_counter() - t0
    n, nnz = a.dimension, a.length
    b = algorithmic_bytes(n, nnz)
    t0 = time.perf_counter()
    with la.Engine(a, 4, reorth="none", profile=True) as e:
        t_up = time.perf_counter() - t0
        e.set_start_seed(123456789)
        e.run(5)
        # timed with no kernel bracketed by events, then the same steps again with all of them for the kernel figures
        e.profile_select(())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        e.profile_select(None)
        e.set_start_seed(123456789)
        e.run(5)
        e.profile_reset()
        e.run(steps)
        prof = e.profile()
        name = e.spmv_name
        # the same steps as the product followed by the recurrence kernel (50 bytes a row moved where the two plane
        # walks of the default move 36)
        os.environ["LANCZOS_HIP_TWO_PASS"] = "0"
        try:
            e.profile_select(())
            e.set_start_seed(123456789)
            e.run(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.run(steps)
            torch.cuda.synchronize()
            dt_unfused = time.perf_counter() - t0
            e.profile_select(None)
            e.profile_reset()
            e.run(steps)
            prof_unfused = e.profile()
        finally:
            os.environ.pop("LANCZOS_HIP_TWO_PASS", None)
    # bytes moved per iteration: what the product's and the recurrence's launches were charged (the format's SpMV bytes
    # + the recurrence's 32 n; with the recurrence in the second plane walk's epilogue 10 n + 26 n)
    fmt_iter = (prof["spmv"]["bytes"] + prof["recur"]["bytes"]) / steps
    unfused_iter = (prof_unfused["spmv"]["bytes"] + prof_unfused["recur"]["bytes"]) / steps
    return {
        "product_then_recurrence": {
            "iters_per_s": steps / dt_unfused,
            "bytes_moved_per_iter": unfused_iter,
            "frac_of_hbm_peak": unfused_iter * steps / dt_unfused / 1e9 / HBM_PEAK_GBPS,
            "kernel_ms_per_iter": {k: v["ms"] / steps for k, v in prof_unfused.items()},
        },
        "workload": f"3-D 7-point Laplacian {N}^3 (n={n}, nnz={nnz}), SpMV + vector step, reorth=none",
        "steps": steps,
        "iters_per_s": steps / dt,
        "ms_per_iter": dt / steps * 1e3,
        "spmv_kernel": name,
        "bytes_moved_per_iter": fmt_iter,
        "frac_of_hbm_peak": fmt_iter * steps / dt / 1e9 / HBM_PEAK_GBPS,
        "kernel_ms_per_iter": {k: v["ms"] / steps for k, v in prof.items()},
        "kernel_GBps": {k: v["bytes"] / (v["ms"] * 1e-3) / 1e9 for k, v in prof.items() if v["ms"] > 0},
        # for comparison with a solver on the CRS arrays: its charged bytes over this run's time (a rate in
        # another format's bytes, not a fraction of anything this run moved)
        "crs_charged_bytes_per_iter": b["iter_charged"],
        "crs_charged_bytes_over_time_GBps": b["iter_charged"] * steps / dt / 1e9,
        "spmv_avg_ms": prof["spmv"]["ms"] / steps,
        "setup_s": {"generate": t_gen, "upload": t_up},
    }


def roofline_from_profile(prof: dict) -> dict:
    """Dominant kernel class by accumulated HIP-event time."""
    name = max(prof, key=lambda k: prof[k]["ms"])
    p = prof[name]
    achieved = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
    return {
        "bound": "hbm",
        "kernel": name,
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": None,
        "launches": p["launches"],
        "avg_launch_ms": p["ms"] / max(p["launches"], 1),
        "algorithmic_bytes_per_launch": p["bytes"] / max(p["launches"], 1),
    }


def single_gpu(args, keeper=None) -> dict:
    import torch
    import lanczos_amd as la

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    d, N = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    a = la.synth_laplacian(d, N)
    n, nnz = a.dimension, a.length
    v0 = la.gaussian_fill(123456789, 0, n)
    t_gen = time.perf_counter() - t0
    bytes_ = algorithmic_bytes(n, nnz)
    K, W = args.steps, args.warmup

    t0 = time.perf_counter()
    eng = la.Engine(a, max(K, W) + 1, reorth=args.reorth, spmv=args.spmv, profile=True)
    t_upload = time.perf_counter() - t0

    # warm-up: W iterations of a throw-away run
    eng.set_start(v0)
    if W > 0:
        eng.run(W)
    # The timed run brackets with HIP events only the kernel classes one of which is the dominant one -- the two
    # reorthogonalisation kernels, or the product and the recurrence when there is no reorthogonalisation: two events
    # around every launch of every class cost the run 6 % at 20 steps (1460 against 1550 it/s).  The roofline record
    # comes from this run; the table of all classes from a second pass over the same K steps, every class timed.
    timed_classes = ("spmv", "recur") if args.reorth in ("none", "selective") else ("dots", "update")
    eng.profile_select(timed_classes)
    eng.set_start(v0)
    eng.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    alpha, beta = eng.run(K)          # enqueues K iterations, then drains the stream
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof_timed = {k: v for k, v in eng.profile().items() if k in timed_classes}
    spmv_name_main = eng.spmv_name
    ritz = la.ritz_values(alpha, beta, K, 4)
    eng.profile_select(None)
    eng.set_start(v0)
    eng.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(K)
    torch.cuda.synchronize()
    dt_all_timed = time.perf_counter() - t0
    prof = eng.profile()

    # the SpMV + vector step on its own (no reorthogonalisation): the part of the
    # iteration whose bytes do not grow with k
    eng.close()

    def vector_step(spmv_choice, plain_rows=False, two_pass=True):
        # plain_rows: the row-wise form on the CRS arrays as they are (by default it walks a grid's planes over a
        # dictionary of column offsets, with the values read from the CRS array); read by the engine at creation
        # two_pass=False: the dictionary form's step as the product followed by the recurrence kernel (by default,
        # from 128^3, it is two plane walks that move 36 bytes a row in place of 50); read by the engine per iteration
        if not two_pass:
            os.environ["LANCZOS_HIP_TWO_PASS"] = "0"
        try:
            return vector_step_run(spmv_choice, plain_rows)
        finally:
            os.environ.pop("LANCZOS_HIP_TWO_PASS", None)

    def vector_step_run(spmv_choice, plain_rows):
        if plain_rows:
            os.environ["LANCZOS_HIP_ROWS_PLANES"] = "0"
        try:
            e2 = la.Engine(a, 4, reorth="none", spmv=spmv_choice, profile=True)
        finally:
            os.environ.pop("LANCZOS_HIP_ROWS_PLANES", None)
        e2.set_start(v0)
        e2.run(5)
        S = 400   # a run() call costs about 0.6 ms besides its steps (drain, history): 4 % of 100 steps of 0.14 ms, 1 % of 400
        # timed with no class bracketed by events (they cost this run of short kernels 6-9 %), then the same S steps
        # again with all of them for the per-kernel figures
        e2.profile_select(())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e2.run(S)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        e2.profile_select(None)
        e2.set_start(v0)
        e2.run(5)
        e2.profile_reset()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e2.run(S)
        torch.cuda.synchronize()
        dt2_events = time.perf_counter() - t1
        p2 = e2.profile()
        name = e2.spmv_name
        e2.close()
        # bytes this run moves per iteration: the SpMV's bytes in the matrix format in use (CRS = 12 nnz +
        # 4 (n + 1) + 16 n; CRS values + pattern numbers = 8 nnz + 4 (n + 1) + 2 n + 16 n; dictionary = 2 n + 16 n)
        # plus the 32 n of the deferred-normalisation recurrence (alpha's partials come out of the SpMV,
        # there is no scaling pass)
        # (with the recurrence in the second plane walk's epilogue: 10 n for the first walk, 26 n for the second)
        moved = (p2["spmv"]["bytes"] + p2["recur"]["bytes"]) / S
        spmv_gb = p2["spmv"]["bytes"] / (p2["spmv"]["ms"] * 1e-3) / 1e9
        plain_crs = not (name.startswith("dict") or (name.startswith("rows") and name.endswith("p")))
        two_walks = abs(p2["recur"]["bytes"] / S - 26.0 * n) < 1.0
        rec = {
            "iters_per_s": S / dt2,
            "ms_per_iter": dt2 / S * 1e3,
            "iters_per_s_with_every_kernel_timed": S / dt2_events,
            "spmv_kernel": name,
            # two walks: class "spmv" below is the first walk (u . A u, reads 10 n), class "recur" the second
            # (A u again with the recurrence in its epilogue, 26 n)
            "step_form": "two plane walks: u.Au, then Au with the recurrence in its epilogue" if two_walks
                         else "product, then the recurrence kernel",
            "matrix_format": "row-pattern dictionary" if name.startswith("dict") else
                             "CRS values and row offsets + column-offset patterns" if name.startswith("rows") and name.endswith("p")
                             else "CRS",
            "bytes_moved_per_iter": moved,
            "achieved_GBps": moved * S / dt2 / 1e9,
            "frac_of_hbm_peak": moved * S / dt2 / 1e9 / HBM_PEAK_GBPS,
            "spmv_avg_ms": p2["spmv"]["ms"] / S,
            "spmv_GBps": spmv_gb,
            "spmv_frac_of_hbm_peak": spmv_gb / HBM_PEAK_GBPS,
            "kernel_ms_per_iter": {k: v["ms"] / S for k, v in p2.items()},
            "kernel_GBps": {k: v["bytes"] / (v["ms"] * 1e-3) / 1e9 for k, v in p2.items() if v["ms"] > 0},
            # a rate in another format's bytes (what a solver on the CRS arrays with separate vector kernels would
            # be charged, B_spmv + 64 n, over this run's time): for comparisons across formats, not a fraction
            "crs_charged_bytes_per_iter": bytes_["iter_charged"],
            "crs_charged_bytes_over_time_GBps": bytes_["iter_charged"] * S / dt2 / 1e9,
        }
        if plain_crs:
            # on the CRS arrays themselves the charged figure is this kernel's own format: B_spmv + 64 n
            # (alpha dot, recurrence, scaling as separate passes) against the B_spmv + 32 n it moves
            rec["crs_charged_frac_of_hbm_peak"] = bytes_["iter_charged"] * S / dt2 / 1e9 / HBM_PEAK_GBPS
        return rec

    sv_crs_planes = None
    if args.main_only:
        sv = sv_crs = {"spmv_kernel": spmv_name_main, "spmv_GBps": None}
    else:
        sv = vector_step(args.spmv)
        if sv["spmv_kernel"].startswith("dict"):
            unfused = vector_step(args.spmv, two_pass=False)
            sv["product_then_recurrence"] = {k: unfused[k] for k in (
                "iters_per_s", "ms_per_iter", "bytes_moved_per_iter", "achieved_GBps", "frac_of_hbm_peak", "spmv_avg_ms",
                "spmv_GBps", "spmv_frac_of_hbm_peak", "kernel_ms_per_iter", "kernel_GBps")}
        # the same step on the plain CRS arrays (what the shared algorithmic-byte figure describes), and with the
        # column array replaced by pattern numbers (what a grid with variable coefficients runs on)
        if sv["spmv_kernel"].startswith("dict"):
            sv_crs = vector_step("rows", plain_rows=True)
            sv_crs_planes = vector_step("rows")
        else:
            sv_crs = sv
    spmv_name = sv["spmv_kernel"]

    out = {
        "metric": "Lanczos iters/sec",
        "value": K / dt,
        "unit": "iters/s",
        "n_gpus": 1,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"3-D 7-point Laplacian {N}^3 (n={n}, nnz={nnz}), {K} Lanczos steps from a fixed "
                        f"N(0,1) start vector, reorth={args.reorth}, basis resident in HBM",
            "reorth": args.reorth,
            "spmv_kernel": spmv_name,
            "parallelism": "1 GPU",
        },
        "roofline": roofline_from_profile(prof_timed),
        # the SpMV kernel of the timed run, in the bytes of the format it reads
        "spmv_hbm_GBps": (prof["spmv"]["bytes"] / (prof["spmv"]["ms"] * 1e-3) / 1e9) if prof["spmv"]["ms"] > 0 else None,
        "spmv_hbm_kernel": spmv_name_main,
        "spmv_hbm_GBps_plain_crs_kernel": sv_crs["spmv_GBps"],
        "spmv_vector_step": sv,
        # the two forms of that step side by side: the one that reaches the 70 % of the HBM peak on the bytes it moves,
        # and the faster one, which moves 28 % fewer bytes at a lower rate
        "spmv_vector_step_forms": None if "product_then_recurrence" not in sv else {
            "product_then_recurrence": {"iters_per_s": sv["product_then_recurrence"]["iters_per_s"],
                                        "bytes_moved_per_iter": sv["product_then_recurrence"]["bytes_moved_per_iter"],
                                        "frac_of_hbm_peak": sv["product_then_recurrence"]["frac_of_hbm_peak"]},
            "two_plane_walks": {"iters_per_s": sv["iters_per_s"], "bytes_moved_per_iter": sv["bytes_moved_per_iter"],
                                "frac_of_hbm_peak": sv["frac_of_hbm_peak"]},
            "default": sv["step_form"],
        },
        "spmv_vector_step_crs": sv_crs,
        "spmv_vector_step_crs_values": sv_crs_planes,
        # second pass over the same K steps with every kernel class bracketed by events (the timed run brackets only
        # the classes named in kernel_classes_timed_in_the_timed_run)
        "kernel_classes_timed_in_the_timed_run": list(timed_classes),
        "value_with_every_class_timed": K / dt_all_timed,
        "kernel_ms_total": {k: v["ms"] for k, v in prof.items()},
        "kernel_launches": {k: v["launches"] for k, v in prof.items()},
        "kernel_algorithmic_bytes_total": {k: v["bytes"] for k, v in prof.items()},
        "ritz_lowest4": [float(x) for x in ritz],
        "setup_s": {"generate": t_gen, "upload": t_upload},
    }
    out["roofline"]["traffic"] = measured_traffic(out["roofline"]["kernel"], args.workload, args.reorth,
                                                  out["roofline"]["algorithmic_bytes_per_launch"])
    if keeper is not None:
        keeper.publish(out)           # the primary figure stands from here on

    def section(name, fn):
        """A secondary section: an error is recorded in the line instead of ending the run."""
        try:
            fn()
        except Exception as exc:  # noqa: BLE001
            out.setdefault("section_errors", {})[name] = f"{type(exc).__name__}: {exc}"[:300]
            import gc
            gc.collect()
            torch.cuda.empty_cache()
        if keeper is not None:
            keeper.publish(out)

    # the same K steps with the orthogonality tracker deciding when to reorthogonalise
    def sec_selective():
        with la.Engine(a, K + 1, reorth="selective", spmv=args.spmv) as es:
            es.set_start(v0)
            es.run(min(W, K))
            es.set_start(v0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a_s, b_s = es.run(K)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
            ritz_s = la.ritz_values(a_s, b_s, K, 4)
            out["selective_reorth"] = {
                "iters_per_s": K / dts,
                "ms_per_step": dts / K * 1e3,
                "steps_that_reorthogonalised": es.reorth_steps,
                "ritz_rel_diff_vs_full": float(np.max(np.abs(ritz_s - ritz) / np.abs(ritz))),
            }
    # the configuration the metric is quoted on is 300 steps; when the caller asked for another count the
    # full-length run is reported beside it (the cost of a step grows with its index)
    def sec_300():
        with la.Engine(a, 301, reorth=args.reorth, spmv=args.spmv, profile=True) as ef:
            ef.set_start(v0)
            ef.run(min(W, 5))
            ef.profile_select(timed_classes)
            ef.set_start(v0)
            ef.profile_reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            af, bf = ef.run(300)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t0
            rf = roofline_from_profile({k: v for k, v in ef.profile().items() if k in timed_classes})
            rf["traffic"] = measured_traffic(rf["kernel"], args.workload, args.reorth, rf["algorithmic_bytes_per_launch"])
            out["config_300_steps"] = {
                "steps": 300, "value": 300 / dtf, "unit": "iters/s", "ms_per_step": dtf / 300 * 1e3, "roofline": rf,
                "ritz_lowest4": [float(x) for x in la.ritz_values(af, bf, 300, 4)],
            }

    # time to solution: selective reorthogonalisation until the lowest eigenvalue stands (closed form known)
    def sec_converge():
        S = args.converge_steps
        one = 2.0 - 2.0 * np.cos(np.pi / (N + 1))
        exact = 3.0 * one
        with la.Engine(a, S + 1, reorth="selective", spmv=args.spmv) as ec:
            ec.set_start(v0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ec.run(S)
            torch.cuda.synchronize()
            dtc = time.perf_counter() - t0
            a_c, b_c = ec.history(S)
            lowest = float(la.ritz_values(a_c, b_c, S, 1)[0])
            out["converged_run"] = {
                "reorth": "selective", "steps": S, "seconds": dtc, "iters_per_s": S / dtc,
                "steps_that_reorthogonalised": ec.reorth_steps,
                "lowest_ritz": lowest, "lowest_exact": exact, "rel_err": abs(lowest - exact) / exact,
                "basis_GB": (S + 1) * 8.0 * n / 1e9,
            }

    def sec_cpu():
        cb = cpu_baseline(a, v0, args.cpu_steps)
        s = cb["steps"]
        # the SAME sample on the GPU: a run of its own over the first s iterations (the cost of a step grows
        # with k on both sides), its rate, and the agreement of the two runs
        eng3 = la.Engine(a, s + 1, reorth=args.reorth, spmv=args.spmv)
        eng3.set_start(v0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        a3, b3 = eng3.run(s)
        torch.cuda.synchronize()
        cb["gpu_same_sample_value"] = s / (time.perf_counter() - t3)
        eng3.close()
        ca, cbeta, cr = np.array(cb["alpha"][:s]), np.array(cb["beta"][:s]), np.array(cb["ritz"])
        cb["gpu_same_sample_rel_err"] = {
            "steps": s,
            "alpha": float(np.max(np.abs(a3 - ca) / np.abs(ca))),
            "beta": float(np.max(np.abs(b3 - cbeta) / np.abs(cbeta))),
        }
        m = min(4, s)
        cb["ritz_rel_err"] = float(np.max(np.abs(la.ritz_values(a3, b3, s, m) - cr[:m]) / np.abs(cr[:m])))
        del cb["alpha"], cb["beta"], cb["ritz"]
        out["cpu_baseline"] = cb

    # the CPU leg first: the line the round driver reads must carry it whatever the later sections do
    if not args.no_cpu_baseline:
        section("cpu_baseline", sec_cpu)
    if args.reorth == "full" and not args.main_only:
        section("selective_reorth", sec_selective)
    if K != 300 and args.workload == "lap256" and not args.main_only:
        section("config_300_steps", sec_300)
    if args.converge_steps > 0 and d == 3:
        section("converged_run", sec_converge)
    if not args.no_block:
        section("block_variant", lambda: out.__setitem__("block_variant", block_variant(la, a)))
    if args.rand_rows > 0:
        section("random_symmetric",
                lambda: out.__setitem__("random_symmetric", random_symmetric_section(la, torch, args.rand_rows)))
    if args.rmat_scale > 0:
        section("power_law_graph",
                lambda: out.__setitem__("power_law_graph", power_law_section(la, torch, args.rmat_scale)))
    if args.big_grid > 0:
        def sec_big():
            nonlocal a
            a = None                  # the 256^3 host arrays go before the 512^3 ones are generated
            out["strong_scaling_laplacian"] = big_laplacian_section(la, torch, args.big_grid)
        section("strong_scaling_laplacian", sec_big)
    return out


class LineKeeper:
    """Holds the result line outside this process.  A child forked before anything touches the GPU
    reads what `publish` sends and, when this process ends -- by returning, by an exception or by an
    abort inside a library -- writes the last complete line to the real stdout.  The primary figure is
    published as soon as it is measured, so a failure in a later section cannot take it away."""

    def __init__(self, out_fd: int):
        r, w = os.pipe()
        self.pid = os.fork()
        if self.pid == 0:
            os.close(w)
            buf = b""
            while True:
                chunk = os.read(r, 1 << 16)
                if not chunk:
                    break
                buf += chunk
                cut = buf.rfind(b"\n", 0, len(buf) - 1)      # keep the last complete line and what follows it
                if cut >= 0 and buf.endswith(b"\n"):
                    buf = buf[cut + 1:]
            lines = [ln for ln in buf.split(b"\n") if ln.strip()]
            if buf.endswith(b"\n") and lines:
                os.write(out_fd, lines[-1] + b"\n")
            elif len(lines) > 1:
                os.write(out_fd, lines[-2] + b"\n")
            os._exit(0)
        os.close(r)
        self.w = w

    def publish(self, obj: dict):
        os.write(self.w, (json.dumps(obj) + "\n").encode())

    def close(self):
        os.close(self.w)
        os.waitpid(self.pid, 0)


def launch_command(argv, gpus: int, port: int) -> list:
    """The command that starts one rank per GPU on this node for `bench.py <argv>`."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(args, argv) -> int:
    """`bench.py --gpus N` called without a launcher around it: the ranks are started as a child
    process (this process has not touched the GPU and never will), rank 0's JSON line is relayed,
    the child's exit code is returned."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(launch_command(argv, args.gpus, port), stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for cand in p.stdout.splitlines():
        cand = cand.strip()
        if cand.startswith("{") and '"metric"' in cand:
            line = cand
    if line is not None:
        print(line, flush=True)
    elif p.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return p.returncode


def main():
    args = parse_args()
    if args.main_only:
        args.no_block = args.no_cpu_baseline = True
        args.rand_rows = args.big_grid = args.converge_steps = args.rmat_scale = 0
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    # stdout carries the one JSON line and nothing else: whatever libraries print while the run
    # lasts (RCCL's version banner goes to stdout) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    keeper = LineKeeper(json_fd)          # forks: before torch is imported or the GPU touched
    os.close(json_fd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    try:
        if args.gpus > 1 or world > 1 or args.force_dist:
            from lanczos_amd.dist import bench_distributed
            out = bench_distributed(args, WORKLOADS, algorithmic_bytes, HBM_PEAK_GBPS, keeper)
        else:
            out = single_gpu(args, keeper)
        sys.stdout.flush()
        if out is not None:
            keeper.publish(out)
    finally:
        keeper.close()


if __name__ == "__main__":
    main()
