"""Per-kernel-class times of one slab of an N-way partition of the 512^3 grid (the middle slab, collectives that do
nothing: scripts/gpu_slab_cost.py's set-up with the engine's event timing on).  python scripts/gpu_slab_kernels.py [world]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import lanczos_amd as la
from lanczos_amd.dist import Comm, build_gpu_rank, partition_rows


class SilentComm(Comm):
    def __init__(self, torch_, world, rank):
        self.dist, self.torch, self.device = None, torch_, torch_.device("cuda", 0)
        self.rank, self.world = rank, world

    def all_gather_ints(self, values):
        return [list(values) for _ in range(self.world)]

    def all_reduce_sum(self, t):
        pass

    def all_gather_into(self, out, inp):
        pass

    def neighbour_exchange_start(self, send_lo, recv_lo, send_hi, recv_hi):
        return []


world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, steps = 512, 100
n = N ** 3
rank = world // 2
bounds = partition_rows(n, world)
slab = la.synth_laplacian_rows(3, N, bounds[rank], bounds[rank + 1])
r = build_gpu_rank(la, torch, SilentComm(torch, world, rank), slab, bounds, steps + 1, reorth="none", profile=True)
v0 = la.gaussian_fill(123456789, bounds[rank], bounds[rank + 1] - bounds[rank])
r.set_start(v0)
r.run(5)
r.set_start(v0)
r.engine.e.profile_reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
r.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
r.history()
p = r.engine.e.profile()
print(f"world {world}: {dt / steps * 1e3:.4f} ms per step (every class timed)",
      {k: (round(v["ms"] / steps, 4), v["launches"] // steps) for k, v in p.items() if v["launches"]})
