#!/usr/bin/env python3
"""Regenerates the fixtures in this directory.  Run in the build container only
(it needs /root/reference and, for the MKL vectors, /opt/conda/lib/libmkl_rt.so.1):

    make -C oracle ref && python tests/golden/make_golden.py

What it writes
  triplets_*.txt        small headerless triplet files (inputs, written by this script)
  ref_*_swap{0,1}.bin   what the reference's own read_mat_coo_incremental,
                        convert_from_coo, matvec_crs and matvec_coo produce for them
                        (oracle/_ref/ref_driver, built from the reference's coo.c and
                        crs.c as they lie; layout in oracle/ref_driver.c)
  mkl_dsyevr.json       lowest eigenvalues that Intel MKL's LAPACKE_dsyevr_work
                        (the routine behind diagonalize_double_partial,
                        src/util.c:186-259, called here through ctypes with the same
                        arguments: 'N','I','U', il=1, iu, abstol 0) returns for
                        dense tridiagonal matrices built from recorded alpha/beta
  reference_known_answers.json is NOT generated here: it holds the Ritz values the
                        unmodified reference printed for fixed start vectors
                        when it was compiled and run against MKL in the build container
                        before this repository had code; they are copied from that record.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def write(path, text):
    with open(os.path.join(HERE, path), "w") as f:
        f.write(text)


def triplet_cases():
    rng = np.random.default_rng(20260128)
    cases = {}
    # 1-D Laplacian n = 12, the row-by-row order the reference's inputs use
    n = 12
    lines = []
    for i in range(1, n + 1):
        if i > 1:
            lines.append(f"{i} {i-1} -1")
        lines.append(f"{i} {i} 2")
        if i < n:
            lines.append(f"{i} {i+1} -1")
    cases["lap1d_sorted"] = "\n".join(lines) + "\n"
    # same entries shuffled: within-row order of the CRS must follow the file
    perm = rng.permutation(len(lines))
    cases["lap1d_shuffled"] = "\n".join(lines[p] for p in perm) + "\n"
    # duplicates kept separate, empty rows (rows 3 and 6 of 8), values in several formats,
    # blank lines, tabs, leading blanks, no trailing newline
    cases["dups_empty_rows"] = (
        "1 1 1.5\n\n2 2 -2.25e0\n  4\t4   3\n1 1 0.5\n8 1 7\n1 8 7.0\n5 7 1e-3\n7 5 1E-3\n"
        "2 2 4\n\n\n4 2 -1\n2 4 -1\n8 8 9")
    # a single row
    cases["single_row"] = "1 1 3\n1 1 4\n"
    # unsymmetric, random, with repeated coordinates
    m = 300
    r = rng.integers(1, 41, m)
    c = rng.integers(1, 41, m)
    v = rng.standard_normal(m)
    cases["random40"] = "".join(f"{a} {b} {x:.17g}\n" for a, b, x in zip(r, c, v))
    return cases


def make_ref_bins():
    if not os.path.exists(DRIVER):
        sys.exit("build oracle/_ref first: make -C oracle ref")
    for name, text in triplet_cases().items():
        write(f"triplets_{name}.txt", text)
        for swap in (0, 1):
            out = os.path.join(HERE, f"ref_{name}_swap{swap}.bin")
            subprocess.run([DRIVER, os.path.join(HERE, f"triplets_{name}.txt"), str(swap), out], check=True)


def mkl_dsyevr_vectors():
    try:
        mkl = C.CDLL("/opt/conda/lib/libmkl_rt.so.1")
    except OSError as e:
        print("MKL runtime not loadable, mkl_dsyevr.json left as it is:", e)
        return
    f = mkl.LAPACKE_dsyevr_work
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_char, C.c_char, C.c_char, C.c_int, dp, C.c_int, C.c_double, C.c_double,
                  C.c_int, C.c_int, C.c_double, ip, dp, dp, C.c_int, ip, dp, C.c_int, ip, C.c_int]

    def lowest(alpha, beta, nth):
        n = len(alpha)
        a = np.zeros((n, n), order="F")
        a[np.arange(n), np.arange(n)] = alpha
        a[np.arange(n - 1), np.arange(1, n)] = beta      # upper triangle, as 'U' reads it
        a[np.arange(1, n), np.arange(n - 1)] = beta
        iu = min(nth, n)
        w = np.zeros(n)
        m = C.c_int(0)
        z = C.c_double(0)
        isuppz = np.zeros(2 * n, np.int32)
        wq, iwq = C.c_double(0), C.c_int(0)
        args = lambda work, lwork, iwork, liwork: (102, b"N", b"I", b"U", n, a.ctypes.data_as(dp), n, 0.0, 0.0, 1, iu,
                                                  0.0, C.byref(m), w.ctypes.data_as(dp), C.byref(z), 1,
                                                  isuppz.ctypes.data_as(ip), work, lwork, iwork, liwork)
        info = f(*args(C.byref(wq), -1, C.byref(iwq), -1))
        assert info == 0
        work = np.zeros(int(wq.value))
        iwork = np.zeros(max(iwq.value, 1), np.int32)
        info = f(*args(work.ctypes.data_as(dp), work.size, iwork.ctypes.data_as(ip), iwork.size))
        assert info == 0 and m.value == iu
        return [float(x) for x in w[:iu]]

    rng = np.random.default_rng(7)
    cases = []
    # (a) random tridiagonals of the sizes the loop sees
    for n in (1, 2, 3, 5, 17, 100, 300):
        alpha = rng.standard_normal(n) * 3 + 6
        beta = np.abs(rng.standard_normal(max(n - 1, 0))) + 0.1
        cases.append({"kind": "random", "alpha": alpha.tolist(), "beta": beta.tolist(), "nth": 4,
                      "eig": lowest(alpha, beta, 4)})
    # (b) the Toeplitz (2,-1) matrix: eigenvalues 2 - 2 cos(pi j / (n+1)), tiny lowest ones
    for n in (50, 300):
        alpha = np.full(n, 2.0)
        beta = np.full(n - 1, -1.0)
        cases.append({"kind": "toeplitz", "alpha": alpha.tolist(), "beta": beta.tolist(), "nth": 6,
                      "eig": lowest(alpha, beta, 6)})
    # (c) graded: entries spanning 12 orders of magnitude
    n = 60
    alpha = 10.0 ** np.linspace(-6, 6, n)
    beta = 10.0 ** np.linspace(-7, 5, n - 1)
    cases.append({"kind": "graded", "alpha": alpha.tolist(), "beta": beta.tolist(), "nth": 4,
                  "eig": lowest(alpha, beta, 4)})
    with open(os.path.join(HERE, "mkl_dsyevr.json"), "w") as fjs:
        json.dump({"source": "Intel MKL 2021.4 LAPACKE_dsyevr_work via ctypes, jobz N, range I, uplo U, il 1, abstol 0",
                   "cases": cases}, fjs)
    print("mkl_dsyevr.json:", len(cases), "cases")


if __name__ == "__main__":
    make_ref_bins()
    mkl_dsyevr_vectors()
