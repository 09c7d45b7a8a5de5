"""TEST INFRASTRUCTURE ONLY: ctypes loader for the CPU restatement (liboracle.so).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package, and only as the checker.  The product
package never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

COO_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("val", np.float64)], align=True)
assert COO_DTYPE.itemsize == 16


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds) if it is missing or stale."""
    src = os.path.join(_HERE, "lanczos_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int64)
        L.orc_num_threads.restype = C.c_int
        L.orc_parse_line.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), dp]
        L.orc_parse_line.restype = C.c_int
        L.orc_read_triplets.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.POINTER(C.c_uint64)]
        L.orc_read_triplets.restype = C.c_long
        L.orc_coo_to_crs.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, dp, ip, ip]
        L.orc_coo_to_crs.restype = C.c_long
        L.orc_matvec_crs.argtypes = [C.c_uint64, ip, ip, dp, dp, dp]
        L.orc_matvec_crs.restype = None
        L.orc_matvec_coo.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, dp, dp]
        L.orc_matvec_coo.restype = None
        L.orc_dot.argtypes = [dp, dp, C.c_int]
        L.orc_dot.restype = C.c_double
        L.orc_diag_partial.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int]
        L.orc_diag_partial.restype = C.c_int
        L.orc_lanczos_crs.argtypes = [C.c_uint64, ip, ip, dp, dp, dp, C.c_int, C.c_int, C.c_double,
                                      C.c_int, dp, dp, dp, dp, C.c_int, C.POINTER(C.c_int)]
        L.orc_lanczos_crs.restype = C.c_int
        L.orc_lanczos_coo.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, dp, dp, C.c_int, C.c_int,
                                      C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.POINTER(C.c_int)]
        L.orc_lanczos_coo.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def num_threads() -> int:
    return lib().orc_num_threads()


def parse_line(line: str):
    r, c, v = C.c_int(0), C.c_int(0), C.c_double(float("nan"))
    rc = lib().orc_parse_line(line.encode(), C.byref(r), C.byref(c), C.byref(v))
    return rc, r.value, c.value, v.value


def read_triplets(path: str):
    """-> (coo structured array, dimension) as read_mat_coo_incremental would."""
    dim = C.c_uint64(0)
    n = lib().orc_read_triplets(path.encode(), None, 0, C.byref(dim))
    if n < 0:
        raise OSError(f"orc_read_triplets({path}) -> {n}")
    coo = np.zeros(n, dtype=COO_DTYPE)
    m = lib().orc_read_triplets(path.encode(), coo.ctypes.data, n, C.byref(dim))
    if m != n:
        raise ValueError(f"orc_read_triplets second pass -> {m}")
    return coo, int(dim.value)


def coo_to_crs(coo: np.ndarray, dim: int, swap: int):
    """-> (values f64[len], column_index i64[len], row_head i64[dim+1])."""
    coo = np.ascontiguousarray(coo, dtype=COO_DTYPE)
    n = coo.shape[0]
    values = np.empty(n, np.float64)
    cols = np.empty(n, np.int64)
    head = np.empty(dim + 1, np.int64)
    rc = lib().orc_coo_to_crs(coo.ctypes.data, n, dim, int(swap), _dp(values), _ip(cols), _ip(head))
    if rc != 0:
        raise IndexError(f"entry {rc - 1} out of range")
    return values, cols, head


def matvec_crs(row_head, cols, values, x):
    dim = row_head.shape[0] - 1
    y = np.empty(dim, np.float64)
    lib().orc_matvec_crs(dim, _ip(row_head), _ip(cols), _dp(values), _dp(np.ascontiguousarray(x)), _dp(y))
    return y


def matvec_coo(coo, dim, x):
    y = np.empty(dim, np.float64)
    lib().orc_matvec_coo(coo.ctypes.data, coo.shape[0], dim, _dp(np.ascontiguousarray(x)), _dp(y))
    return y


def diag_partial(tmat: np.ndarray, n: int, nth: int):
    """tmat: column-major ld x ld (pass a Fortran-ordered array)."""
    assert tmat.flags.f_contiguous
    ld = tmat.shape[0]
    eig = np.empty(nth, np.float64)
    rc = lib().orc_diag_partial(_dp(tmat), ld, _dp(eig), n, nth)
    if rc != 0:
        raise RuntimeError(f"orc_diag_partial -> {rc}")
    return eig


class LanczosResult:
    def __init__(self, steps, exit_kind, eigenvalues, alpha, beta, ritz_hist, timing):
        self.steps = steps
        self.exit_kind = exit_kind
        self.eigenvalues = eigenvalues
        self.alpha = alpha
        self.beta = beta
        self.ritz_hist = ritz_hist
        self.timing = dict(zip(("init", "spmv", "diagonalization", "reorthogonalization"), timing))
        self.reorth_steps = int(timing[4]) if len(timing) > 4 else None  # selective mode only


REORTH = {"full": 0, "cuda_compat": 1, "none": 2, "selective": 3}


def lanczos_crs(row_head, cols, values, v0, nth, max_iter, threshold=0.0, reorth="full", verbose=False):
    dim = row_head.shape[0] - 1
    eig = np.zeros(nth, np.float64)
    alpha = np.zeros(max_iter - 1, np.float64)
    beta = np.zeros(max_iter - 1, np.float64)
    hist = np.zeros((max_iter - 1, nth), np.float64)
    timing = np.zeros(5, np.float64)
    kind = C.c_int(0)
    steps = lib().orc_lanczos_crs(dim, _ip(row_head), _ip(cols), _dp(values),
                                  _dp(np.ascontiguousarray(v0, dtype=np.float64)), _dp(eig), nth, max_iter,
                                  float(threshold), REORTH[reorth], _dp(alpha), _dp(beta), _dp(hist),
                                  _dp(timing), int(verbose), C.byref(kind))
    if kind.value < 0:
        raise ValueError(f"orc_lanczos_crs rejected its arguments (exit_kind={kind.value})")
    return LanczosResult(steps, kind.value, eig, alpha, beta, hist, timing)


def lanczos_coo(coo, dim, v0, nth, max_iter, threshold=0.0, reorth="full", verbose=False):
    eig = np.zeros(nth, np.float64)
    alpha = np.zeros(max_iter - 1, np.float64)
    beta = np.zeros(max_iter - 1, np.float64)
    hist = np.zeros((max_iter - 1, nth), np.float64)
    timing = np.zeros(5, np.float64)
    kind = C.c_int(0)
    steps = lib().orc_lanczos_coo(coo.ctypes.data, coo.shape[0], dim,
                                  _dp(np.ascontiguousarray(v0, dtype=np.float64)), _dp(eig), nth, max_iter,
                                  float(threshold), REORTH[reorth], _dp(alpha), _dp(beta), _dp(hist),
                                  _dp(timing), int(verbose), C.byref(kind))
    if kind.value < 0:
        raise ValueError(f"orc_lanczos_coo rejected its arguments (exit_kind={kind.value})")
    return LanczosResult(steps, kind.value, eig, alpha, beta, hist, timing)


def xorshift_start(n: int) -> np.ndarray:
    """Start vector of the recorded known answers: xorshift64 from
    0x9E3779B97F4A7C15, r[i] = (s >> 11) / 2**53 - 0.5."""
    out = np.empty(n, np.float64)
    f = lib().orc_xorshift_start
    f.argtypes = [C.c_uint64, C.POINTER(C.c_double)]
    f.restype = None
    f(n, _dp(out))
    return out
