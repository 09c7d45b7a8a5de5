"""Host time per step of the row-partitioned driver loop (python + ctypes + torch.distributed calls),
measured with one RCCL rank on a matrix so small that the GPU work is negligible: what the host must
stay under for the multi-GPU step to remain GPU-bound.  Prints per-step wall times with the
collectives skipped (world size 1 skips them) and with every all-reduce actually issued."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import lanczos_amd as la
from lanczos_amd.dist import Comm, build_gpu_rank, partition_rows

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))


class EagerComm(Comm):
    """issues the all-reduces even with one rank"""

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)


N = 32
a = la.synth_laplacian(3, N)
bounds = partition_rows(a.dimension, 1)
v0 = la.gaussian_fill(1, 0, a.dimension)
for reorth, steps in (("none", 2000), ("full", 300)):
    for cls in (Comm, EagerComm):
        comm = cls(dist, torch, torch.device("cuda", 0))
        r = build_gpu_rank(la, torch, comm, a, bounds, steps + 1, reorth=reorth)
        r.set_start(v0)
        r.run(20)
        r.set_start(v0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.run(steps)
        t_host = time.perf_counter() - t0           # enqueue only: nothing in the loop waits for the GPU
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"reorth={reorth:5s} {cls.__name__:9s}: host {t_host / steps * 1e6:7.1f} us/step, "
              f"with the GPU drained {t_all / steps * 1e6:7.1f} us/step", flush=True)
        del r
dist.destroy_process_group()
