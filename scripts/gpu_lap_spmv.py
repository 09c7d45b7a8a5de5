"""A few Lanczos steps (no reorthogonalisation) on the 3-D Laplacian with a chosen SpMV kernel: for rocprofv3 passes."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "/root/repo")
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "auto"
a = la.synth_laplacian(3, N)
with la.Engine(a, 4, reorth="none", spmv=kind, profile=True) as e:
    e.set_start_seed(1)
    e.run(10)
    p = e.profile()
    print(e.spmv_name, p["spmv"]["ms"] / 10, "ms per SpMV")
