"""Wall time of the drop-in entry point on configuration 2 (256^3, 300 steps, full reorthogonalisation):
upload, dictionary build, 300 pipelined iterations with the host-side Ritz solve per step."""
import os
import sys
import time

sys.path.insert(0, ".")
os.environ.setdefault("LANCZOS_HIP_QUIET", "1")
os.environ.setdefault("LANCZOS_HIP_JSON", "1")
import lanczos_amd as la

a = la.synth_laplacian(3, 256)
for profile in ("1", "0"):
    os.environ["LANCZOS_HIP_PROFILE"] = profile
    t0 = time.perf_counter()
    rc, eig = la.lanczos_hip_crs(a, 4, 301, 0.0)
    print(f"profile={profile} rc={rc} wall={time.perf_counter() - t0:.3f}s eig={eig.tolist()}", flush=True)
