"""What one rank of an N-GPU row-partitioned run computes per step, measured on ONE GPU: the middle
slab of the N-way partition goes through the same driver loop and phase calls as in a real run, with
a stand-in communicator whose collectives do nothing (the remote entries stay zero; the kernels do
the same work).  step time x N against the one-GPU time is the compute-side scaling efficiency; the
collectives' latency comes on top in a real run."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import lanczos_amd as la
from lanczos_amd.dist import Comm, build_gpu_rank, partition_rows


class SilentComm(Comm):
    """rank `rank` of `world` with no peers behind it"""

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


def run(N, world, reorth, steps):
    n = N ** 3
    rank = world // 2
    bounds = partition_rows(n, world)
    slab = la.synth_laplacian_rows(3, N, bounds[rank], bounds[rank + 1])
    comm = SilentComm(torch, world, rank)
    r = build_gpu_rank(la, torch, comm, slab, bounds, steps + 1, reorth=reorth, profile=False)
    v0 = la.gaussian_fill(123456789, bounds[rank], bounds[rank + 1] - bounds[rank])
    r.set_start(v0)
    r.run(5)
    r.set_start(v0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"grid": N, "world": world, "reorth": reorth, "steps": steps, "rows_per_rank": bounds[rank + 1] - bounds[rank],
            "exchange": r.plan.mode, "interior_rows": r.interior, "ms_per_step": dt / steps * 1e3}


out = []
for N, reorth, steps in ((256, "full", 300), (256, "none", 200), (512, "none", 100)):
    base = None
    for world in (1, 2, 4, 8):
        rec = run(N, world, reorth, steps)
        base = base or rec["ms_per_step"]
        rec["compute_side_speedup"] = base / rec["ms_per_step"]
        out.append(rec)
        print(rec, flush=True)
        torch.cuda.empty_cache()
print(json.dumps(out))
