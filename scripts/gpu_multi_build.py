"""Set-up time of the row slabs of one process (column remap on the host, upload, engine per slab) with one host
thread per slab against one slab after the other: the 512^3 Laplacian (16 GB of CRS arrays on the host) cut into
4 and 8 slabs on one GPU (loopback transport), then a few iterations against the single engine.
python scripts/gpu_multi_build.py [grid] [slab counts ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
counts = [int(s) for s in sys.argv[2:]] or [4, 8]
t0 = time.perf_counter()
a = la.synth_laplacian(3, N)
print(f"{N}^3 generated in {time.perf_counter() - t0:.1f} s: n={a.dimension}, nnz={a.length}", flush=True)
for P in counts:
    for threads in ("0", "1"):
        os.environ["LANCZOS_HIP_MULTI_BUILD_THREADS"] = threads
        t0 = time.perf_counter()
        with la.MultiEngine(a, 4, P, reorth="none", loopback=True) as m:
            wall = time.perf_counter() - t0
            st = m.build_stats()
            m.set_start_seed(1)
            t1 = time.perf_counter()
            al, be = m.run(20)
            dt = time.perf_counter() - t1
            print(f"{P} slabs, build threads {'per slab' if threads == '1' else 'none'}: set-up {st['seconds']:.2f} s on {st['host_threads']} "
                  f"host thread(s) (create call {wall:.2f} s); 20 steps {dt * 1e3 / 20:.3f} ms each, alpha_19 = {al[19]:.15g}", flush=True)
