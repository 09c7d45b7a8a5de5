"""Time to solution on configuration 2's matrix (256^3 Laplacian): how many Lanczos steps the lowest
eigenvalues need, and what they cost under selective against full reorthogonalisation."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import lanczos_amd as la

N = 256
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
a = la.synth_laplacian(3, N)
one = 2.0 - 2.0 * np.cos(np.pi * np.arange(1, 4) / (N + 1))
exact = np.sort((one[:, None, None] + one[None, :, None] + one[None, None, :]).ravel())[:4]
print("exact lowest (with multiplicity):", exact, flush=True)
with la.Engine(a, steps + 1, reorth="selective") as e:
    e.set_start_seed(123456789)
    t0 = time.perf_counter()
    done = 0
    for upto in range(300, steps + 1, 300):
        e.run(upto - done)
        done = upto
        alpha, beta = e.history(done)
        ritz = la.ritz_values(alpha, beta, done, 4)
        print(f"steps {done}: {time.perf_counter() - t0:.2f}s, reorth steps {e.reorth_steps}, ritz {ritz}, "
              f"rel err of the lowest {abs(ritz[0] - exact[0]) / exact[0]:.2e}", flush=True)
