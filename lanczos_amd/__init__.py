"""Python host-side mirror of the reference's C interface, over the C ABI.

Every function here is a thin ctypes call into ``lib/liblanczos_hip.so`` (the
HIP engine, ``include/lanczos_hip.h``) or ``lib/liblanczos_host.so`` (the host
C API kept from the reference, ``include/lanczos_host.h``).  There is no
Python or CPU implementation behind them: if the libraries are missing the
import fails.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_HIP = os.path.join(_HERE, "lib", "liblanczos_hip.so")
LIB_HOST = os.path.join(_HERE, "lib", "liblanczos_host.so")

if not (os.path.exists(LIB_HIP) and os.path.exists(LIB_HOST)):
    raise ImportError(
        f"{LIB_HIP} / {LIB_HOST} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C lanczos_amd/csrc`); this package has no fallback path")

_hip = C.CDLL(LIB_HIP, mode=C.RTLD_GLOBAL)
_host = C.CDLL(LIB_HOST)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)

COO_DTYPE = np.dtype([("index_row", np.int32), ("index_column", np.int32), ("value", np.float64)], align=True)


class Coo(C.Structure):
    _fields_ = [("index_row", C.c_int), ("index_column", C.c_int), ("value", C.c_double)]


class Mat_Coo(C.Structure):
    _fields_ = [("data", C.POINTER(Coo)), ("length", C.c_size_t), ("dimension", C.c_size_t)]


class Mat_Crs(C.Structure):
    _fields_ = [("values", _dp), ("column_index", _ip), ("row_head_indexes", _ip),
                ("length", C.c_size_t), ("dimension", C.c_size_t)]


class Options(C.Structure):
    _fields_ = [("struct_size", C.c_size_t), ("device", C.c_int), ("stream", C.c_void_p),
                ("reorth", C.c_int), ("spmv_kernel", C.c_int), ("profile", C.c_int),
                ("n_global", C.c_int64), ("row_begin", C.c_int64), ("n_ext", C.c_int64),
                ("x_ext", C.c_void_p), ("basis", C.c_void_p), ("basis_ld", C.c_int64)]


REORTH = {"full": 0, "cuda_compat": 1, "none": 2, "mgs": 3, "full2": 4, "selective": 5}
SPMV = {"auto": 0, "vector": 1, "stream": 2, "pipe": 3, "pipe_nt": 4, "rows": 5, "panel": 6, "dict": 7, "merge": 8}
KERNEL_CLASSES = ("spmv", "recur", "scale", "dots", "update", "small")

# ---- prototypes -----------------------------------------------------------
_hip.lanczos_hip_crs.argtypes = [C.POINTER(Mat_Crs), _dp, C.c_void_p, C.c_int, C.c_int, C.c_double]
_hip.lanczos_hip_crs.restype = C.c_int
_hip.lanczos_hip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, _ip, _ip, _dp, C.c_int,
                                    C.POINTER(Options)]
_hip.lanczos_hip_create.restype = C.c_int
_hip.lanczos_hip_create_from_coo.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Mat_Coo), C.c_int, C.c_int,
                                             C.POINTER(Options)]
_hip.lanczos_hip_create_from_coo.restype = C.c_int
_hip.lanczos_hip_download_crs.argtypes = [C.c_void_p, _ip, _ip, _dp]
_hip.lanczos_hip_download_crs.restype = C.c_int
_hip.lanczos_hip_destroy.argtypes = [C.c_void_p]
_hip.lanczos_hip_destroy.restype = None
_hip.lanczos_hip_set_start.argtypes = [C.c_void_p, _dp]
_hip.lanczos_hip_set_start.restype = C.c_int
_hip.lanczos_hip_set_start_scaled.argtypes = [C.c_void_p, _dp, C.c_double]
_hip.lanczos_hip_set_start_scaled.restype = C.c_int
_hip.lanczos_hip_set_start_seed.argtypes = [C.c_void_p, C.c_uint64]
_hip.lanczos_hip_set_start_seed.restype = C.c_int
_hip.lanczos_hip_gaussian_fill.argtypes = [C.c_uint64, C.c_int64, C.c_int64, _dp]
_hip.lanczos_hip_gaussian_fill.restype = None
_hip.lanczos_hip_run.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
_hip.lanczos_hip_run.restype = C.c_int
_hip.lanczos_hip_iterations.argtypes = [C.c_void_p]
_hip.lanczos_hip_iterations.restype = C.c_int
_hip.lanczos_hip_thick_restart.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
_hip.lanczos_hip_reorth_steps.argtypes = [C.c_void_p]
_hip.lanczos_hip_reorth_steps.restype = C.c_int
_hip.lanczos_hip_omega_create.argtypes = [C.c_int]
_hip.lanczos_hip_omega_create.restype = C.c_void_p
_hip.lanczos_hip_omega_destroy.argtypes = [C.c_void_p]
_hip.lanczos_hip_omega_destroy.restype = None
_hip.lanczos_hip_omega_reset.argtypes = [C.c_void_p]
_hip.lanczos_hip_omega_reset.restype = None
_hip.lanczos_hip_omega_step.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
_hip.lanczos_hip_omega_step.restype = C.c_int
_hip.lanczos_hip_omega_set_beta.argtypes = [C.c_void_p, C.c_int, C.c_double]
_hip.lanczos_hip_omega_set_beta.restype = None
_hip.lanczos_hip_omega_reorths.argtypes = [C.c_void_p]
_hip.lanczos_hip_omega_reorths.restype = C.c_int
_hip.lanczos_hip_spmv.argtypes = [C.c_void_p, _dp, _dp]
_hip.lanczos_hip_spmv.restype = C.c_int
_hip.lanczos_hip_spmv_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
_hip.lanczos_hip_spmv_noalloc.restype = C.c_int
_hip.lanczos_hip_get_vector.argtypes = [C.c_void_p, C.c_int, _dp]
_hip.lanczos_hip_get_vector.restype = C.c_int
_hip.lanczos_hip_profile.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64), _dp]
_hip.lanczos_hip_profile.restype = C.c_int
_hip.lanczos_hip_profile_select.argtypes = [C.c_void_p, C.c_uint]
_hip.lanczos_hip_profile_reset.argtypes = [C.c_void_p]
_hip.lanczos_hip_profile_reset.restype = None
_hip.lanczos_hip_spmv_name.argtypes = [C.c_void_p]
_hip.lanczos_hip_spmv_name.restype = C.c_char_p
_hip.lanczos_hip_tridiag_eigs.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp]
_hip.lanczos_hip_tridiag_eigs.restype = C.c_int
_hip.lanczos_hip_gather_base.argtypes = [C.c_int64]
_hip.lanczos_hip_gather_base.restype = C.c_int64
_hip.lanczos_hip_last_error.argtypes = []
_hip.lanczos_hip_last_error.restype = C.c_char_p
for _name in ("lanczos_hip_phase_spmv", "lanczos_hip_phase_update", "lanczos_hip_phase_scale",
              "lanczos_hip_phase_recur_lazy", "lanczos_hip_phase_lazy_norm", "lanczos_hip_phase_lazy_close"):
    getattr(_hip, _name).argtypes = [C.c_void_p, C.c_void_p]
    getattr(_hip, _name).restype = C.c_int
_hip.lanczos_hip_phase_spmv_begin.argtypes = [C.c_void_p, C.c_int64, C.c_int64]
_hip.lanczos_hip_phase_spmv_begin.restype = C.c_int
_hip.lanczos_hip_phase_spmv_end.argtypes = [C.c_void_p, C.c_void_p]
_hip.lanczos_hip_phase_spmv_end.restype = C.c_int
_hip.lanczos_hip_phase_spmv_can_overlap.argtypes = [C.c_void_p]
_hip.lanczos_hip_phase_spmv_can_overlap.restype = C.c_int
_hip.lanczos_hip_phase_recur.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
_hip.lanczos_hip_phase_recur.restype = C.c_int
_hip.lanczos_hip_phase_dots.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
_hip.lanczos_hip_phase_dots.restype = C.c_int
_hip.lanczos_hip_export_vector.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
_hip.lanczos_hip_export_vector.restype = C.c_int
_hip.lanczos_hip_tridiag_eigvecs.argtypes = [_dp, _dp, C.c_int, C.c_int, _dp, _dp]
_hip.lanczos_hip_tridiag_eigvecs.restype = C.c_int
_hip.lanczos_hip_ritz_vectors.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp]
_hip.lanczos_hip_ritz_vectors.restype = C.c_int
_hip.lanczos_hip_checkpoint_save.argtypes = [C.c_void_p, C.c_char_p]
_hip.lanczos_hip_checkpoint_save.restype = C.c_int
_hip.lanczos_hip_checkpoint_load.argtypes = [C.c_void_p, C.c_char_p]
_hip.lanczos_hip_checkpoint_load.restype = C.c_int
_hip.lanczos_hip_local_vector.argtypes = [C.c_void_p, C.c_int]
_hip.lanczos_hip_local_vector.restype = C.c_void_p
_hip.lanczos_hip_history.argtypes = [C.c_void_p, _dp, _dp, C.c_int]
_hip.lanczos_hip_history.restype = C.c_int

_host.read_matrix_market.argtypes = [C.c_char_p, C.POINTER(Mat_Coo)]
_host.read_matrix_market.restype = C.c_int
_host.is_matrix_market.argtypes = [C.c_char_p]
_host.is_matrix_market.restype = C.c_int
_host.read_mat_coo_incremental.argtypes = [C.c_char_p]
_host.read_mat_coo_incremental.restype = Mat_Coo
_host.parse_to_coo.argtypes = [C.c_char_p, C.POINTER(Coo)]
_host.parse_to_coo.restype = None
_host.convert_from_coo.argtypes = [C.POINTER(Mat_Coo), C.c_int]
_host.convert_from_coo.restype = Mat_Crs
_host.matvec_crs.argtypes = [C.POINTER(Mat_Crs), _dp, _dp]
_host.matvec_crs.restype = None
_host.matvec_coo.argtypes = [C.POINTER(Mat_Coo), _dp, _dp]
_host.matvec_coo.restype = None
_host.lanczos_synth_laplacian.argtypes = [C.c_int, C.c_int64, C.POINTER(Mat_Crs)]
_host.lanczos_synth_laplacian.restype = C.c_int
_host.lanczos_synth_laplacian_rows.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.POINTER(Mat_Crs)]
_host.lanczos_synth_laplacian_rows.restype = C.c_int
_host.lanczos_synth_random_symmetric.argtypes = [C.c_int64, C.c_int, C.c_uint64, C.POINTER(Mat_Crs)]
_host.lanczos_synth_random_symmetric.restype = C.c_int
_host.lanczos_synth_random_symmetric_rows.argtypes = [C.c_int64, C.c_int, C.c_uint64, C.c_int64, C.c_int64,
                                                       C.POINTER(Mat_Crs)]
_host.lanczos_synth_random_symmetric_rows.restype = C.c_int
_host.lanczos_free_crs.argtypes = [C.POINTER(Mat_Crs)]
_host.lanczos_free_crs.restype = None
MATVEC_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp)   # Matvec_General (reference src/lanczos.h:5)


class Mat_Matvec(C.Structure):
    _fields_ = [("mat", C.c_void_p), ("matvec", MATVEC_FN), ("dimension", C.c_size_t)]


_host.lanczos.argtypes = [Mat_Matvec, _dp, C.c_void_p, C.c_int, C.c_int, C.c_double]
_host.lanczos.restype = None
_hip.lanczos_hip_operator.argtypes = [MATVEC_FN, C.c_void_p, C.c_size_t, _dp, C.c_void_p, C.c_int, C.c_int, C.c_double]
_hip.lanczos_hip_operator.restype = C.c_int
_hip.lanczos_hip_set_operator.argtypes = [C.c_void_p, MATVEC_FN, C.c_void_p]
_hip.lanczos_hip_set_operator.restype = C.c_int
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class LanczosHipError(RuntimeError):
    pass


def _check(rc: int, what: str):
    if rc != 0:
        raise LanczosHipError(f"{what}: {_hip.lanczos_hip_last_error().decode(errors='replace')}")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ---- host API kept from the reference ---------------------------------------
class CrsMatrix:
    """numpy view of a Mat_Crs (values, column_index, row_head_indexes)."""

    def __init__(self, values, column_index, row_head_indexes):
        self.values = _f64(values)
        self.column_index = _i64(column_index)
        self.row_head_indexes = _i64(row_head_indexes)
        self.dimension = self.row_head_indexes.shape[0] - 1
        self.length = self.values.shape[0]

    def as_struct(self) -> Mat_Crs:
        return Mat_Crs(self.values.ctypes.data_as(_dp), self.column_index.ctypes.data_as(_ip),
                       self.row_head_indexes.ctypes.data_as(_ip), self.length, self.dimension)

    @staticmethod
    def _take(m: Mat_Crs) -> "CrsMatrix":
        n, nnz = int(m.dimension), int(m.length)
        out = CrsMatrix(np.ctypeslib.as_array(m.values, (nnz,)).copy() if nnz else np.zeros(0),
                        np.ctypeslib.as_array(m.column_index, (nnz,)).copy() if nnz else np.zeros(0, np.int64),
                        np.ctypeslib.as_array(m.row_head_indexes, (n + 1,)).copy())
        _host.lanczos_free_crs(C.byref(m))
        return out


def read_mat_coo_incremental(path: str):
    """-> (structured array of Coo, dimension); exits the process on malformed input, as the reference."""
    m = _host.read_mat_coo_incremental(os.fsencode(path))
    n = int(m.length)
    arr = np.zeros(n, COO_DTYPE)
    if n:
        C.memmove(arr.ctypes.data, m.data, n * COO_DTYPE.itemsize)
        _libc.free(C.cast(m.data, C.c_void_p))
    return arr, int(m.dimension)


def read_matrix_market(path: str):
    """Matrix Market coordinate file -> (structured array of Coo, dimension); symmetric and
    skew-symmetric files come back with both triangles.  Raises on unsupported or malformed files."""
    m = Mat_Coo(None, 0, 0)
    if _host.read_matrix_market(os.fsencode(path), C.byref(m)) != 0:
        raise ValueError(f"read_matrix_market: cannot read {path} (see stderr)")
    n = int(m.length)
    arr = np.zeros(n, COO_DTYPE)
    if n:
        C.memmove(arr.ctypes.data, m.data, n * COO_DTYPE.itemsize)
    _libc.free(C.cast(m.data, C.c_void_p))
    return arr, int(m.dimension)


def is_matrix_market(path: str) -> bool:
    return bool(_host.is_matrix_market(os.fsencode(path)))


def parse_to_coo(line: str):
    c = Coo(0, 0, 0.0)
    _host.parse_to_coo(C.create_string_buffer(line.encode()), C.byref(c))
    return c.index_row, c.index_column, c.value


def convert_from_coo(coo: np.ndarray, dimension: int, swap_row_column: int) -> CrsMatrix:
    coo = np.ascontiguousarray(coo, dtype=COO_DTYPE)
    m = Mat_Coo(C.cast(coo.ctypes.data, C.POINTER(Coo)), coo.shape[0], dimension)
    return CrsMatrix._take(_host.convert_from_coo(C.byref(m), int(swap_row_column)))


def matvec_crs(a: CrsMatrix, x) -> np.ndarray:
    y = np.empty(a.dimension)
    s = a.as_struct()
    _host.matvec_crs(C.byref(s), _f64(x).ctypes.data_as(_dp), y.ctypes.data_as(_dp))
    return y


def synth_laplacian(d: int, N: int) -> CrsMatrix:
    m = Mat_Crs()
    if _host.lanczos_synth_laplacian(d, N, C.byref(m)) != 0:
        raise ValueError(f"lanczos_synth_laplacian({d}, {N}) failed")
    return CrsMatrix._take(m)


def synth_laplacian_rows(d: int, N: int, row_begin: int, row_end: int) -> CrsMatrix:
    """Rows [row_begin, row_end) of synth_laplacian(d, N); column indices stay global."""
    m = Mat_Crs()
    if _host.lanczos_synth_laplacian_rows(d, N, row_begin, row_end, C.byref(m)) != 0:
        raise ValueError(f"lanczos_synth_laplacian_rows({d}, {N}, {row_begin}, {row_end}) failed")
    return CrsMatrix._take(m)


def synth_random_symmetric(n: int, per_row: int, seed: int) -> CrsMatrix:
    m = Mat_Crs()
    if _host.lanczos_synth_random_symmetric(n, per_row, seed, C.byref(m)) != 0:
        raise ValueError("lanczos_synth_random_symmetric failed")
    return CrsMatrix._take(m)


def synth_random_symmetric_rows(n: int, per_row: int, seed: int, row_begin: int, row_end: int) -> CrsMatrix:
    """Rows [row_begin, row_end) of synth_random_symmetric(n, per_row, seed); column indices stay global."""
    m = Mat_Crs()
    if _host.lanczos_synth_random_symmetric_rows(n, per_row, seed, row_begin, row_end, C.byref(m)) != 0:
        raise ValueError("lanczos_synth_random_symmetric_rows failed")
    return CrsMatrix._take(m)


def gaussian_fill(seed: int, row_begin: int, n: int) -> np.ndarray:
    out = np.empty(n)
    _hip.lanczos_hip_gaussian_fill(seed, row_begin, n, out.ctypes.data_as(_dp))
    return out


def tridiag_eigs(alpha, beta, n: int, m: int) -> np.ndarray:
    alpha, beta = _f64(alpha), _f64(beta if len(beta) else np.zeros(1))
    eig = np.empty(m)
    rc = _hip.lanczos_hip_tridiag_eigs(alpha.ctypes.data_as(_dp), beta.ctypes.data_as(_dp), n, m,
                                       eig.ctypes.data_as(_dp))
    if rc != 0:
        raise LanczosHipError("lanczos_hip_tridiag_eigs failed")
    return eig


def tridiag_eigvecs(alpha, beta, n: int, eig) -> np.ndarray:
    alpha, beta, eig = _f64(alpha), _f64(beta if len(beta) else np.zeros(1)), _f64(eig)
    m = eig.shape[0]
    S = np.zeros((m, n))
    rc = _hip.lanczos_hip_tridiag_eigvecs(alpha.ctypes.data_as(_dp), beta.ctypes.data_as(_dp), n, m,
                                          eig.ctypes.data_as(_dp), S.ctypes.data_as(_dp))
    if rc != 0:
        raise LanczosHipError("lanczos_hip_tridiag_eigvecs failed")
    return S


# ---- solver entry points ----------------------------------------------------
def lanczos_hip_crs(a: CrsMatrix, nth_eig: int, max_iter: int, threshold: float, vectors: bool = False):
    """The drop-in entry; returns (status, eigenvalues[nth_eig]), and with vectors=True also the
    (nth_eig, n) array the `eigenvectors` argument was filled into (LANCZOS_HIP_EIGENVECTORS=1 for
    the duration of the call)."""
    eig = np.zeros(max(nth_eig, 1))
    s = a.as_struct()
    if not vectors:
        rc = _hip.lanczos_hip_crs(C.byref(s), eig.ctypes.data_as(_dp), None, nth_eig, max_iter, threshold)
        return rc, eig[:max(nth_eig, 0)]
    y = np.zeros((max(nth_eig, 1), a.dimension))
    ptrs = (C.c_void_p * y.shape[0])(*[y[i].ctypes.data for i in range(y.shape[0])])
    old = os.environ.get("LANCZOS_HIP_EIGENVECTORS")
    os.environ["LANCZOS_HIP_EIGENVECTORS"] = "1"
    try:
        rc = _hip.lanczos_hip_crs(C.byref(s), eig.ctypes.data_as(_dp), ptrs, nth_eig, max_iter, threshold)
    finally:
        if old is None:
            del os.environ["LANCZOS_HIP_EIGENVECTORS"]
        else:
            os.environ["LANCZOS_HIP_EIGENVECTORS"] = old
    return rc, eig[:max(nth_eig, 0)], y[:max(nth_eig, 0)]


def lanczos_plugin(apply, dimension: int, nth_eig: int, max_iter: int, threshold: float) -> np.ndarray:
    """lanczos() of the host library (reference src/lanczos.h:34-36) with a custom Matvec_General:
    `apply(x) -> y` is wrapped into the plugin's C signature and called on host memory once per
    iteration.  Exits the process on failure, as the reference's lanczos() does."""
    def cb(_mat, vec, dist):
        x = np.ctypeslib.as_array(vec, (dimension,))
        np.ctypeslib.as_array(dist, (dimension,))[:] = apply(x)

    fn = MATVEC_FN(cb)
    token = C.c_int(0)   # the reference's lanczos() refuses a NULL mat pointer
    eig = np.zeros(nth_eig)
    _host.lanczos(Mat_Matvec(C.cast(C.pointer(token), C.c_void_p), fn, dimension), eig.ctypes.data_as(_dp), None,
                  nth_eig, max_iter, threshold)
    return eig


BLOCK_WIDTH = 16
BLOCK_KERNEL_CLASSES = ("spmm", "gram", "update", "scale", "small")
_hip.lanczos_hip_block_create.argtypes = [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, _ip, _ip, _dp, C.c_int,
                                          C.POINTER(Options)]
_hip.lanczos_hip_block_create.restype = C.c_int
_hip.lanczos_hip_block_destroy.argtypes = [C.c_void_p]
_hip.lanczos_hip_block_destroy.restype = None
_hip.lanczos_hip_block_set_start.argtypes = [C.c_void_p, _dp]
_hip.lanczos_hip_block_set_start_seed.argtypes = [C.c_void_p, C.c_uint64]
_hip.lanczos_hip_block_run.argtypes = [C.c_void_p, C.c_int]
_hip.lanczos_hip_block_steps.argtypes = [C.c_void_p]
_hip.lanczos_hip_block_tmatrix.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
_hip.lanczos_hip_block_ritz.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
_hip.lanczos_hip_block_get_block.argtypes = [C.c_void_p, C.c_int, _dp]
_hip.lanczos_hip_block_set_block.argtypes = [C.c_void_p, C.c_int, _dp]
_hip.lanczos_hip_block_spmm.argtypes = [C.c_void_p, C.c_int, C.c_int]
_hip.lanczos_hip_block_project.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, _dp]
_hip.lanczos_hip_block_orthonormalise.argtypes = [C.c_void_p, C.c_int, _dp]
_hip.lanczos_hip_block_profile.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64), _dp]
_hip.lanczos_hip_block_profile_reset.argtypes = [C.c_void_p]
_hip.lanczos_hip_block_profile_reset.restype = None
_hip.lanczos_hip_block_multi_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int64, C.c_int64, _ip, _ip,
                                                _dp, C.c_int, C.POINTER(Options), C.c_int]
_hip.lanczos_hip_block_multi_create.restype = C.c_int
_hip.lanczos_hip_block_multi_destroy.argtypes = [C.c_void_p]
_hip.lanczos_hip_block_multi_destroy.restype = None
_hip.lanczos_hip_block_multi_set_start.argtypes = [C.c_void_p, _dp]
_hip.lanczos_hip_block_multi_set_start_seed.argtypes = [C.c_void_p, C.c_uint64]
_hip.lanczos_hip_block_multi_run.argtypes = [C.c_void_p, C.c_int]
_hip.lanczos_hip_block_multi_steps.argtypes = [C.c_void_p]
_hip.lanczos_hip_block_multi_slabs.argtypes = [C.c_void_p]
_hip.lanczos_hip_block_multi_tmatrix.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
_hip.lanczos_hip_block_multi_ritz.argtypes = [C.c_void_p, C.c_int, C.c_int, _dp]
_hip.lanczos_hip_block_multi_get_block.argtypes = [C.c_void_p, C.c_int, _dp]
_hip.lanczos_hip_sym_lowest_eigs.argtypes = [_dp, C.c_int, C.c_int, _dp]


_hip.lanczos_hip_sym_eig_jacobi.argtypes = [_dp, C.c_int, _dp, _dp]


def sym_eig(a: np.ndarray):
    """All eigenpairs of a small dense symmetric matrix: (values ascending, vectors as rows)."""
    a = np.array(a, dtype=np.float64, order="C")
    n = a.shape[0]
    w, v = np.zeros(n), np.zeros((n, n))
    if _hip.lanczos_hip_sym_eig_jacobi(a.ctypes.data_as(_dp), n, w.ctypes.data_as(_dp), v.ctypes.data_as(_dp)) != 0:
        raise LanczosHipError("lanczos_hip_sym_eig_jacobi failed")
    return w, v


def sym_lowest_eigs(a: np.ndarray, nth: int) -> np.ndarray:
    """Lowest nth eigenvalues of a dense symmetric matrix (Householder + bisection, host)."""
    a = np.array(a, dtype=np.float64, order="C")
    out = np.zeros(nth)
    if _hip.lanczos_hip_sym_lowest_eigs(a.ctypes.data_as(_dp), a.shape[0], nth, out.ctypes.data_as(_dp)) != 0:
        raise LanczosHipError("lanczos_hip_sym_lowest_eigs failed")
    return out


class BlockEngine:
    """Block-Lanczos variant: 16 vectors per step, blocks are (n, 16) row-major arrays."""

    def __init__(self, a: "CrsMatrix", max_steps: int, profile: bool = False, device: int = -1):
        self._h = C.c_void_p(None)
        self.n, self.max_steps = a.dimension, max_steps
        opt = Options(C.sizeof(Options), device, None, 0, 0, int(profile), 0, 0, 0, None)
        _check(_hip.lanczos_hip_block_create(C.byref(self._h), a.dimension, a.length,
                                             a.row_head_indexes.ctypes.data_as(_ip),
                                             a.column_index.ctypes.data_as(_ip), a.values.ctypes.data_as(_dp),
                                             max_steps, C.byref(opt)), "lanczos_hip_block_create")

    def close(self):
        if self._h:
            _hip.lanczos_hip_block_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_start(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert x0.shape == (self.n, BLOCK_WIDTH)
        _check(_hip.lanczos_hip_block_set_start(self._h, x0.ctypes.data_as(_dp)), "lanczos_hip_block_set_start")

    def set_start_seed(self, seed: int):
        _check(_hip.lanczos_hip_block_set_start_seed(self._h, seed), "lanczos_hip_block_set_start_seed")

    def run(self, steps: int):
        _check(_hip.lanczos_hip_block_run(self._h, steps), "lanczos_hip_block_run")

    @property
    def steps(self) -> int:
        return _hip.lanczos_hip_block_steps(self._h)

    def tmatrix(self, steps: int = None):
        steps = self.steps if steps is None else steps
        a = np.zeros((steps, BLOCK_WIDTH, BLOCK_WIDTH))
        b = np.zeros((steps, BLOCK_WIDTH, BLOCK_WIDTH))
        _check(_hip.lanczos_hip_block_tmatrix(self._h, steps, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp)),
               "lanczos_hip_block_tmatrix")
        return a, b

    def ritz(self, nth: int, steps: int = None) -> np.ndarray:
        steps = self.steps if steps is None else steps
        out = np.zeros(nth)
        _check(_hip.lanczos_hip_block_ritz(self._h, steps, nth, out.ctypes.data_as(_dp)), "lanczos_hip_block_ritz")
        return out

    def get_block(self, t: int) -> np.ndarray:
        out = np.zeros((self.n, BLOCK_WIDTH))
        _check(_hip.lanczos_hip_block_get_block(self._h, t, out.ctypes.data_as(_dp)), "lanczos_hip_block_get_block")
        return out

    def set_block(self, t: int, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.n, BLOCK_WIDTH)
        _check(_hip.lanczos_hip_block_set_block(self._h, t, x.ctypes.data_as(_dp)), "lanczos_hip_block_set_block")

    def spmm(self, t_in: int, t_out: int):
        _check(_hip.lanczos_hip_block_spmm(self._h, t_in, t_out), "lanczos_hip_block_spmm")

    def project(self, t0: int, t1: int, t_w: int) -> np.ndarray:
        h = np.zeros((t1 - t0, BLOCK_WIDTH, BLOCK_WIDTH))
        _check(_hip.lanczos_hip_block_project(self._h, t0, t1, t_w, h.ctypes.data_as(_dp)),
               "lanczos_hip_block_project")
        return h

    def orthonormalise(self, t: int) -> np.ndarray:
        r = np.zeros((BLOCK_WIDTH, BLOCK_WIDTH))
        _check(_hip.lanczos_hip_block_orthonormalise(self._h, t, r.ctypes.data_as(_dp)),
               "lanczos_hip_block_orthonormalise")
        return r

    def profile(self) -> dict:
        out = {}
        for i, name in enumerate(BLOCK_KERNEL_CLASSES):
            ms, ln, by = C.c_double(), C.c_int64(), C.c_double()
            _check(_hip.lanczos_hip_block_profile(self._h, i, C.byref(ms), C.byref(ln), C.byref(by)),
                   "lanczos_hip_block_profile")
            out[name] = {"ms": ms.value, "launches": ln.value, "bytes": by.value}
        return out

    def profile_reset(self):
        _hip.lanczos_hip_block_profile_reset(self._h)


class BlockMultiEngine:
    """The block variant on row slabs, one host process over `ngpu` GPUs (lanczos_hip_block_multi_*); several slabs on ONE
    GPU with loopback=True (copies and a summing kernel in place of the collectives)."""

    def __init__(self, a: "CrsMatrix", max_steps: int, ngpu: int, loopback: bool = False, devices=None):
        self._h = C.c_void_p(None)
        self.n, self.max_steps = a.dimension, max_steps
        opt = Options(C.sizeof(Options), -1, None, 0, 0, 0, 0, 0, 0, None)
        devs = (C.c_int * len(devices))(*devices) if devices else None
        _check(_hip.lanczos_hip_block_multi_create(C.byref(self._h), ngpu, devs, a.dimension, a.length,
                                                   a.row_head_indexes.ctypes.data_as(_ip), a.column_index.ctypes.data_as(_ip),
                                                   a.values.ctypes.data_as(_dp), max_steps, C.byref(opt),
                                                   1 if loopback else 0), "lanczos_hip_block_multi_create")

    def close(self):
        if self._h:
            _hip.lanczos_hip_block_multi_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_start(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        assert x0.shape == (self.n, BLOCK_WIDTH)
        _check(_hip.lanczos_hip_block_multi_set_start(self._h, x0.ctypes.data_as(_dp)), "lanczos_hip_block_multi_set_start")

    def set_start_seed(self, seed: int):
        _check(_hip.lanczos_hip_block_multi_set_start_seed(self._h, seed), "lanczos_hip_block_multi_set_start_seed")

    def run(self, steps: int):
        _check(_hip.lanczos_hip_block_multi_run(self._h, steps), "lanczos_hip_block_multi_run")

    @property
    def steps(self) -> int:
        return _hip.lanczos_hip_block_multi_steps(self._h)

    @property
    def slabs(self) -> int:
        return _hip.lanczos_hip_block_multi_slabs(self._h)

    def tmatrix(self, steps: int = None):
        steps = self.steps if steps is None else steps
        a = np.zeros((steps, BLOCK_WIDTH, BLOCK_WIDTH))
        b = np.zeros((steps, BLOCK_WIDTH, BLOCK_WIDTH))
        _check(_hip.lanczos_hip_block_multi_tmatrix(self._h, steps, a.ctypes.data_as(_dp), b.ctypes.data_as(_dp)),
               "lanczos_hip_block_multi_tmatrix")
        return a, b

    def ritz(self, nth: int, steps: int = None) -> np.ndarray:
        steps = self.steps if steps is None else steps
        out = np.zeros(nth)
        _check(_hip.lanczos_hip_block_multi_ritz(self._h, steps, nth, out.ctypes.data_as(_dp)), "lanczos_hip_block_multi_ritz")
        return out

    def get_block(self, t: int) -> np.ndarray:
        out = np.zeros((self.n, BLOCK_WIDTH))
        _check(_hip.lanczos_hip_block_multi_get_block(self._h, t, out.ctypes.data_as(_dp)), "lanczos_hip_block_multi_get_block")
        return out


class OmegaTracker:
    """Decision rule of selective reorthogonalisation (host arithmetic on alpha/beta only)."""

    def __init__(self, max_steps: int):
        self._h = _hip.lanczos_hip_omega_create(max_steps)
        if not self._h:
            raise LanczosHipError("lanczos_hip_omega_create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            _hip.lanczos_hip_omega_destroy(self._h)
            self._h = None

    def reset(self):
        _hip.lanczos_hip_omega_reset(self._h)

    def step(self, j: int, alpha: float, beta: float):
        """-> (reorthogonalise now?, largest estimate of |v_{j+1} . v_i|)"""
        est = C.c_double(0.0)
        rc = _hip.lanczos_hip_omega_step(self._h, j, alpha, beta, C.byref(est))
        if rc < 0:
            raise LanczosHipError(f"orthogonality tracker called out of order at step {j}")
        return bool(rc), est.value

    def set_beta(self, j: int, beta: float):
        _hip.lanczos_hip_omega_set_beta(self._h, j, beta)

    @property
    def reorths(self) -> int:
        return _hip.lanczos_hip_omega_reorths(self._h)


class Engine:
    """Handle API: one matrix resident on one GPU."""

    def __init__(self, a: CrsMatrix, max_iter: int, reorth: str = "full", spmv: str = "auto",
                 profile: bool = False, device: int = -1, stream: int | None = None,
                 n_global: int = 0, row_begin: int = 0, n_ext: int = 0, x_ext: int | None = None,
                 basis: int | None = None, basis_ld: int = 0):
        self._h = C.c_void_p(None)
        self.n = a.dimension
        self.max_iter = max_iter
        opt = Options(C.sizeof(Options), device, stream, REORTH[reorth], SPMV[spmv], int(profile),
                      n_global, row_begin, n_ext, x_ext, basis, basis_ld)
        _check(_hip.lanczos_hip_create(C.byref(self._h), a.dimension, a.length,
                                       a.row_head_indexes.ctypes.data_as(_ip),
                                       a.column_index.ctypes.data_as(_ip),
                                       a.values.ctypes.data_as(_dp), max_iter, C.byref(opt)),
               "lanczos_hip_create")

    @classmethod
    def from_coo(cls, coo: np.ndarray, dimension: int, swap_row_column: int, max_iter: int, reorth: str = "full",
                 spmv: str = "auto", profile: bool = False, device: int = -1):
        """Engine whose CRS arrays are built from the triplets on the device."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(None)
        self.n = dimension
        self.max_iter = max_iter
        coo = np.ascontiguousarray(coo, dtype=COO_DTYPE)
        m = Mat_Coo(C.cast(coo.ctypes.data, C.POINTER(Coo)), coo.shape[0], dimension)
        opt = Options(C.sizeof(Options), device, None, REORTH[reorth], SPMV[spmv], int(profile), 0, 0, 0, None)
        _check(_hip.lanczos_hip_create_from_coo(C.byref(self._h), C.byref(m), int(swap_row_column), max_iter,
                                                C.byref(opt)), "lanczos_hip_create_from_coo")
        self.nnz = coo.shape[0]
        return self

    def download_crs(self, nnz: int) -> "CrsMatrix":
        head = np.zeros(self.n + 1, np.int64)
        cols = np.zeros(nnz, np.int64)
        vals = np.zeros(nnz)
        _check(_hip.lanczos_hip_download_crs(self._h, head.ctypes.data_as(_ip), cols.ctypes.data_as(_ip),
                                             vals.ctypes.data_as(_dp)), "lanczos_hip_download_crs")
        return CrsMatrix(vals, cols, head)

    def close(self):
        if self._h:
            _hip.lanczos_hip_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def spmv_name(self) -> str:
        return _hip.lanczos_hip_spmv_name(self._h).decode()

    @property
    def iterations(self) -> int:
        return _hip.lanczos_hip_iterations(self._h)

    def thick_restart(self, nth: int, keep: int, max_cycles: int, tol: float, vectors: bool = False):
        """-> dict(eig, resid, cycles, products[, vectors]) from a freshly set start vector."""
        eig, resid = np.zeros(nth), np.zeros(nth)
        vec = np.zeros((nth, self.n)) if vectors else None
        cycles, products = C.c_int(0), C.c_int(0)
        _check(_hip.lanczos_hip_thick_restart(self._h, nth, keep, max_cycles, tol, eig.ctypes.data_as(_dp),
                                              resid.ctypes.data_as(_dp),
                                              vec.ctypes.data_as(_dp) if vectors else None,
                                              C.byref(cycles), C.byref(products)), "lanczos_hip_thick_restart")
        out = {"eig": eig, "resid": resid, "cycles": cycles.value, "products": products.value}
        if vectors:
            out["vectors"] = vec
        return out

    @property
    def reorth_steps(self) -> int:
        return _hip.lanczos_hip_reorth_steps(self._h)

    def set_start(self, v0):
        v0 = _f64(v0)
        assert v0.shape[0] == self.n
        _check(_hip.lanczos_hip_set_start(self._h, v0.ctypes.data_as(_dp)), "lanczos_hip_set_start")

    def set_start_scaled(self, v0, scale: float):
        v0 = _f64(v0)
        assert v0.shape[0] == self.n
        _check(_hip.lanczos_hip_set_start_scaled(self._h, v0.ctypes.data_as(_dp), scale),
               "lanczos_hip_set_start_scaled")

    def set_start_seed(self, seed: int):
        _check(_hip.lanczos_hip_set_start_seed(self._h, seed), "lanczos_hip_set_start_seed")

    def run(self, steps: int):
        """`steps` more iterations; returns alpha and beta of THESE iterations (history(count) gives
        the whole sequence of a run continued over several calls)."""
        alpha, beta = np.zeros(steps), np.zeros(steps)
        _check(_hip.lanczos_hip_run(self._h, steps, alpha.ctypes.data_as(_dp), beta.ctypes.data_as(_dp)),
               "lanczos_hip_run")
        return alpha, beta

    def spmv(self, x, n_ext: int | None = None):
        x = _f64(x)
        y = np.empty(self.n)
        _check(_hip.lanczos_hip_spmv(self._h, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp)), "lanczos_hip_spmv")
        return y

    def spmv_noalloc(self, x_dev: int, y_dev: int, dot_dev: int | None = None):
        """y = A x on device addresses (enqueued on the engine's stream), no allocation."""
        _check(_hip.lanczos_hip_spmv_noalloc(self._h, x_dev, y_dev, dot_dev), "lanczos_hip_spmv_noalloc")

    def set_operator(self, fn, mat_ptr=None):
        """fn: a MATVEC_FN instance (keep it alive as long as the engine runs)."""
        _check(_hip.lanczos_hip_set_operator(self._h, fn, mat_ptr), "lanczos_hip_set_operator")

    def get_vector(self, k: int):
        out = np.empty(self.n)
        _check(_hip.lanczos_hip_get_vector(self._h, k, out.ctypes.data_as(_dp)), "lanczos_hip_get_vector")
        return out

    def ritz_vectors(self, steps: int, m: int):
        """-> (eig[m], vectors[m, n], residual_est[m]) of the first `steps` iterations."""
        m = min(m, steps)
        eig, vecs, res = np.zeros(m), np.zeros((m, self.n)), np.zeros(m)
        _check(_hip.lanczos_hip_ritz_vectors(self._h, steps, m, eig.ctypes.data_as(_dp), vecs.ctypes.data_as(_dp),
                                             res.ctypes.data_as(_dp)), "lanczos_hip_ritz_vectors")
        return eig, vecs, res

    def checkpoint_save(self, path: str):
        _check(_hip.lanczos_hip_checkpoint_save(self._h, os.fsencode(path)), "lanczos_hip_checkpoint_save")

    def checkpoint_load(self, path: str):
        _check(_hip.lanczos_hip_checkpoint_load(self._h, os.fsencode(path)), "lanczos_hip_checkpoint_load")

    def profile(self) -> dict:
        out = {}
        for i, name in enumerate(KERNEL_CLASSES):
            ms, cnt, by = C.c_double(0), C.c_int64(0), C.c_double(0)
            _check(_hip.lanczos_hip_profile(self._h, i, C.byref(ms), C.byref(cnt), C.byref(by)),
                   "lanczos_hip_profile")
            out[name] = {"ms": ms.value, "launches": cnt.value, "bytes": by.value}
        return out

    def profile_reset(self):
        _hip.lanczos_hip_profile_reset(self._h)

    def profile_select(self, classes=None):
        """The kernel classes (names of KERNEL_CLASSES) whose launches are timed from now on; None: all of them.
        Launches and bytes are counted for every class."""
        mask = 0xFFFFFFFF if classes is None else sum(1 << KERNEL_CLASSES.index(c) for c in classes)
        _check(_hip.lanczos_hip_profile_select(self._h, mask), "lanczos_hip_profile_select")

    # phase-level calls for row-partitioned runs; `red` is a device address
    def phase_spmv(self, red: int):
        _check(_hip.lanczos_hip_phase_spmv(self._h, red), "lanczos_hip_phase_spmv")

    def phase_recur(self, red: int):
        n_red, is_norm = C.c_int(0), C.c_int(0)
        _check(_hip.lanczos_hip_phase_recur(self._h, red, C.byref(n_red), C.byref(is_norm)),
               "lanczos_hip_phase_recur")
        return n_red.value, bool(is_norm.value)

    def phase_dots(self, red: int) -> int:
        n_red = C.c_int(0)
        _check(_hip.lanczos_hip_phase_dots(self._h, red, C.byref(n_red)), "lanczos_hip_phase_dots")
        return n_red.value

    def export_vector(self, k: int, dst: int):
        _check(_hip.lanczos_hip_export_vector(self._h, k, dst), "lanczos_hip_export_vector")

    def phase_update(self, red: int):
        _check(_hip.lanczos_hip_phase_update(self._h, red), "lanczos_hip_phase_update")

    def phase_scale(self, red: int):
        _check(_hip.lanczos_hip_phase_scale(self._h, red), "lanczos_hip_phase_scale")

    def phase_spmv_begin(self, first_interior_row: int, end_interior_row: int):
        _check(_hip.lanczos_hip_phase_spmv_begin(self._h, first_interior_row, end_interior_row),
               "lanczos_hip_phase_spmv_begin")

    def phase_spmv_end(self, red: int):
        _check(_hip.lanczos_hip_phase_spmv_end(self._h, red), "lanczos_hip_phase_spmv_end")

    @property
    def can_overlap(self) -> bool:
        return bool(_hip.lanczos_hip_phase_spmv_can_overlap(self._h))

    def phase_recur_lazy(self, red: int):
        _check(_hip.lanczos_hip_phase_recur_lazy(self._h, red), "lanczos_hip_phase_recur_lazy")

    def phase_lazy_norm(self, red: int):
        _check(_hip.lanczos_hip_phase_lazy_norm(self._h, red), "lanczos_hip_phase_lazy_norm")

    def phase_lazy_close(self, red: int):
        _check(_hip.lanczos_hip_phase_lazy_close(self._h, red), "lanczos_hip_phase_lazy_close")

    def local_vector_ptr(self, k: int) -> int:
        return _hip.lanczos_hip_local_vector(self._h, k)

    def history(self, count: int):
        alpha, beta = np.zeros(count), np.zeros(count)
        _check(_hip.lanczos_hip_history(self._h, alpha.ctypes.data_as(_dp), beta.ctypes.data_as(_dp), count),
               "lanczos_hip_history")
        return alpha, beta


_hip.lanczos_hip_multi_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int), C.c_int64, C.c_int64,
                                          _ip, _ip, _dp, C.c_int, C.POINTER(Options), C.c_int]
_hip.lanczos_hip_multi_create.restype = C.c_int
_hip.lanczos_hip_multi_destroy.argtypes = [C.c_void_p]
_hip.lanczos_hip_multi_destroy.restype = None
_hip.lanczos_hip_multi_set_start.argtypes = [C.c_void_p, _dp]
_hip.lanczos_hip_multi_set_start_seed.argtypes = [C.c_void_p, C.c_uint64]
_hip.lanczos_hip_multi_run.argtypes = [C.c_void_p, C.c_int, _dp, _dp]
_hip.lanczos_hip_multi_step.argtypes = [C.c_void_p]
_hip.lanczos_hip_multi_engine.argtypes = [C.c_void_p, C.c_int]
_hip.lanczos_hip_multi_engine.restype = C.c_void_p
_hip.lanczos_hip_multi_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
_hip.lanczos_hip_multi_build_stats.argtypes = [C.c_void_p, _dp, C.POINTER(C.c_int)]
MULTI_LOOPBACK = 1


class MultiEngine:
    """Row slabs of one matrix on several GPUs, driven from this one process (RCCL), or all on one
    GPU with loopback=True (copies and a summing kernel in place of the collectives)."""

    def __init__(self, a: CrsMatrix, max_iter: int, ngpu: int, reorth: str = "full", spmv: str = "auto",
                 loopback: bool = False, devices=None, profile: bool = False):
        self._h = C.c_void_p(None)
        self.n, self.max_iter, self.ngpu = a.dimension, max_iter, ngpu
        opt = Options(C.sizeof(Options), -1, None, REORTH[reorth], SPMV[spmv], int(profile), 0, 0, 0, None, None, 0)
        devs = (C.c_int * ngpu)(*devices) if devices is not None else None
        _check(_hip.lanczos_hip_multi_create(C.byref(self._h), ngpu, devs, a.dimension, a.length,
                                             a.row_head_indexes.ctypes.data_as(_ip),
                                             a.column_index.ctypes.data_as(_ip), a.values.ctypes.data_as(_dp),
                                             max_iter, C.byref(opt), MULTI_LOOPBACK if loopback else 0),
               "lanczos_hip_multi_create")

    def close(self):
        if self._h:
            _hip.lanczos_hip_multi_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_start(self, v0):
        v0 = _f64(v0)
        assert v0.shape[0] == self.n
        _check(_hip.lanczos_hip_multi_set_start(self._h, v0.ctypes.data_as(_dp)), "lanczos_hip_multi_set_start")

    def set_start_seed(self, seed: int):
        _check(_hip.lanczos_hip_multi_set_start_seed(self._h, seed), "lanczos_hip_multi_set_start_seed")

    def run(self, steps: int):
        alpha, beta = np.zeros(steps), np.zeros(steps)
        _check(_hip.lanczos_hip_multi_run(self._h, steps, alpha.ctypes.data_as(_dp), beta.ctypes.data_as(_dp)),
               "lanczos_hip_multi_run")
        return alpha, beta

    def info(self) -> dict:
        v = [C.c_int(0) for _ in range(5)]
        _check(_hip.lanczos_hip_multi_info(self._h, *[C.byref(x) for x in v]), "lanczos_hip_multi_info")
        return dict(zip(("slabs", "halo", "overlapped", "iterations", "reorth_steps"), (x.value for x in v)))

    def build_stats(self) -> dict:
        sec, thr = C.c_double(0), C.c_int(0)
        _check(_hip.lanczos_hip_multi_build_stats(self._h, C.byref(sec), C.byref(thr)), "lanczos_hip_multi_build_stats")
        return {"seconds": sec.value, "host_threads": thr.value}

    def slab_spmv_name(self, slab: int) -> str:
        return _hip.lanczos_hip_spmv_name(_hip.lanczos_hip_multi_engine(self._h, slab)).decode()

    def slab_profile(self, slab: int) -> dict:
        h = _hip.lanczos_hip_multi_engine(self._h, slab)
        out = {}
        for i, name in enumerate(KERNEL_CLASSES):
            ms, cnt, by = C.c_double(0), C.c_int64(0), C.c_double(0)
            _check(_hip.lanczos_hip_profile(h, i, C.byref(ms), C.byref(cnt), C.byref(by)), "lanczos_hip_profile")
            out[name] = {"ms": ms.value, "launches": cnt.value, "bytes": by.value}
        return out


def gather_base(n: int) -> int:
    """First x index of the remote entries of an engine that owns no basis (options.basis)."""
    return int(_hip.lanczos_hip_gather_base(n))


def ritz_values(alpha, beta, steps: int, nth: int) -> np.ndarray:
    """Lowest nth Ritz values of the steps x steps tridiagonal matrix (host solve)."""
    return tridiag_eigs(alpha[:steps], beta[:max(steps - 1, 0)], steps, nth)
