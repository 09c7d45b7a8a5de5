"""Which host ranges hipHostRegister accepts on this machine (flags, alignment, size)."""
import ctypes as C
import mmap

import numpy as np

hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipGetErrorName.restype = C.c_char_p
hip.hipGetErrorName.argtypes = [C.c_int]
hip.hipInit(0)
hip.hipSetDevice(0)


def attempt(name, ptr, size, flags):
    rc = hip.hipHostRegister(ptr, size, flags)
    print(f"{name:38s} ptr%4096={ptr % 4096:5d} size={size:>12d} flags={flags}: {hip.hipGetErrorName(rc).decode()}", flush=True)
    if rc == 0:
        hip.hipHostUnregister(ptr)


a = np.ones(1 << 27)  # 1 GiB
p = a.ctypes.data
for flags in (0, 1, 2, 3):
    attempt("numpy 1 GiB as is", p, a.nbytes, flags)
lo = (p + 4095) // 4096 * 4096
attempt("numpy, page-aligned inner range", lo, (a.nbytes - 8192) // 4096 * 4096, 0)
b = np.ones(1 << 16)
attempt("numpy 512 KiB", b.ctypes.data, b.nbytes, 0)
m = mmap.mmap(-1, 1 << 26)
buf = (C.c_char * (1 << 26)).from_buffer(m)
attempt("anonymous mmap 64 MiB", C.addressof(buf), 1 << 26, 0)
attempt("anonymous mmap 64 MiB, offset 8", C.addressof(buf) + 8, (1 << 26) - 16, 0)
ro = np.ones(1 << 20)
ro.setflags(write=False)
attempt("numpy 8 MiB (numpy read-only flag)", ro.ctypes.data, ro.nbytes, 0)
