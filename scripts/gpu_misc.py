import sys, time, json, os, subprocess, numpy as np
sys.path.insert(0, '.')
import lanczos_amd as la
# (1) random symmetric, ~21 nnz/row, 10M rows: SpMV rate of the pipe form
t=time.time(); a=la.synth_random_symmetric(10_000_000, 10, 42); print("gen random", time.time()-t, a.length/a.dimension, flush=True)
for kind in ("pipe","vector","stream"):
    with la.Engine(a, 4, reorth="none", spmv=kind, profile=True) as e:
        e.set_start_seed(1); e.run(3); e.profile_reset(); e.run(20); p=e.profile()
        print("random10M", e.spmv_name, "spmv ms", p["spmv"]["ms"]/20, "GB/s", p["spmv"]["bytes"]/p["spmv"]["ms"]/1e6, "recur ms", p["recur"]["ms"]/20, flush=True)
del a
# (2) drop-in wall time on lap256, 100 steps full reorth, vs engine.run
a=la.synth_laplacian(3,256)
with la.Engine(a, 101) as e:
    e.set_start_seed(123456789); t=time.time(); al,be=e.run(100); t_run=time.time()-t
os.environ["LANCZOS_HIP_QUIET"]="1"
t=time.time(); rc,eig=la.lanczos_hip_crs(a,4,101,0.0); t_drop=time.time()-t
print("lap256 100 steps: engine.run", t_run, "drop-in total (incl upload+init)", t_drop, "ritz", eig, la.ritz_values(al,be,100,4), flush=True)
