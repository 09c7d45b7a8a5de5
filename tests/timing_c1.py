"""Configuration 1 (about 10k rows, 100 steps): the drop-in against the CPU port, wall time."""
import sys
import time

sys.path.insert(0, ".")  # run from the repository root: python tests/timing_c1.py
import lanczos_amd as la
import oracle

for a, name in ((la.synth_laplacian(1, 10000), "lap1d_10000"), (la.synth_laplacian(2, 100), "lap2d_100")):
    v0 = la.gaussian_fill(123456789, 0, a.dimension)
    t0 = time.perf_counter()
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 101, 0.0)
    t_cpu = time.perf_counter() - t0
    for reorth in ("full", "selective"):
        with la.Engine(a, 101, reorth=reorth) as e:
            e.set_start(v0)
            e.run(100)                       # warm
            e.set_start(v0)
            t0 = time.perf_counter()
            alpha, beta = e.run(100)
            t_gpu = time.perf_counter() - t0
        print(name, reorth, "cpu port %.2f ms (loop %.2f ms)" % (t_cpu * 1e3, sum(o.timing.values()) * 1e3),
              "gpu engine %.2f ms" % (t_gpu * 1e3), e.spmv_name)
