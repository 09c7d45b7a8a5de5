"""Stores past the caches in the vector kernels and the row-wise SpMV (LANCZOS_HIP_NT_STORES = 0 / 1, read at every launch):
the SpMV + vector step on the dictionary and on the CRS arrays, and runs with full reorthogonalisation; an engine per
setting, two rounds.  python scripts/gpu_recur_nt.py [grid sizes ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

for N in [int(s) for s in sys.argv[1:]] or [256, 512]:
    a = la.synth_laplacian(3, N)
    for rnd in range(2):
        for spmv in ("auto", "rows"):
            for nt in ("0", "1"):
                os.environ["LANCZOS_HIP_NT_STORES"] = nt
                if spmv == "rows":
                    os.environ["LANCZOS_HIP_ROWS_PLANES"] = "0"
                with la.Engine(a, 4, reorth="none", spmv=spmv, profile=True) as e:
                    e.set_start_seed(1)
                    e.run(5)
                    e.profile_reset()
                    t0 = time.perf_counter()
                    e.run(100)
                    dt = time.perf_counter() - t0
                    p = e.profile()
                    print(f"{N}^3 {e.spmv_name} none nt={nt}: {100 / dt:.1f} it/s, spmv {p['spmv']['ms'] / 100:.4f} ms, "
                          f"recur {p['recur']['ms'] / 100:.4f} ms", flush=True)
                os.environ.pop("LANCZOS_HIP_ROWS_PLANES", None)
        if 8.0 * a.dimension * 302 < 200e9:
            for steps in (20, 300):
                for nt in ("0", "1"):
                    os.environ["LANCZOS_HIP_NT_STORES"] = nt
                    with la.Engine(a, steps + 1, reorth="full", profile=True) as e:
                        e.set_start_seed(1)
                        e.run(5)
                        e.set_start_seed(1)
                        e.profile_reset()
                        t0 = time.perf_counter()
                        e.run(steps)
                        dt = time.perf_counter() - t0
                        p = e.profile()
                        print(f"{N}^3 full {steps} steps nt={nt}: {steps / dt:.1f} it/s; per step ms: "
                              + ", ".join(f"{k} {v['ms'] / steps:.4f}" for k, v in p.items()), flush=True)
os.environ.pop("LANCZOS_HIP_NT_STORES", None)
