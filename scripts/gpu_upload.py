"""Matrix upload of the drop-in with and without page-locking the caller's arrays (run on the GPU box)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); import lanczos_amd as la\n"
        "a = la.synth_laplacian(3, int(sys.argv[1]))\n"
        "for i in range(2):\n"
        "    la.lanczos_hip_crs(a, 2, 6, 0.0)\n" % ROOT)
for N in sys.argv[1:] or ["256"]:
    for pin in ("1", "0"):
        env = dict(os.environ, LANCZOS_HIP_PIN_UPLOAD=pin, LANCZOS_HIP_QUIET="1")
        p = subprocess.run([sys.executable, "-c", code, N], capture_output=True, text=True, env=env)
        lines = [ln for ln in p.stderr.splitlines() if "name=upload" in ln or "name=total_wall" in ln]
        print(f"N={N} pin={pin}:", " | ".join(lines), flush=True)
