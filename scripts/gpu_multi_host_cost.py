"""Host time per iteration of the one-process multi-GPU path: P loopback slabs on one GPU over a matrix small
enough that the kernels take next to nothing, so that what is measured is what the host thread spends issuing
one iteration for P slabs (on a real node the GPUs work in parallel and this is the floor of a step).
python scripts/gpu_multi_host_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la

a = la.synth_laplacian(3, 48)
v0 = la.gaussian_fill(1, 0, a.dimension)
steps = 2000
for reorth in ("none", "full"):
    for P in (1, 2, 4, 8):
        st = steps if reorth == "none" else 200
        with la.MultiEngine(a, st + 1, P, reorth=reorth, loopback=True) as m:
            m.set_start(v0)
            m.run(50)
            m.set_start(v0)
            t0 = time.perf_counter()
            m.run(st)
            dt = time.perf_counter() - t0
        print(f"reorth={reorth} slabs={P}: {dt / st * 1e6:.1f} us per iteration", flush=True)
    with la.Engine(a, (steps if reorth == "none" else 200) + 1, reorth=reorth) as e:
        st = steps if reorth == "none" else 200
        e.set_start(v0)
        e.run(50)
        e.set_start(v0)
        t0 = time.perf_counter()
        e.run(st)
        dt = time.perf_counter() - t0
    print(f"reorth={reorth} single engine: {dt / st * 1e6:.1f} us per iteration", flush=True)
