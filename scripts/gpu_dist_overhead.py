import os, sys, time, numpy as np
sys.path.insert(0, '.')
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import lanczos_amd as la
from lanczos_amd import dist as ld
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
comm = ld.Comm(dist, torch, torch.device("cuda", 0))
N = 128; n = N**3
bounds = ld.partition_rows(n, 1)
a = la.synth_laplacian_rows(3, N, 0, n)
for reorth in ("none", "full"):
    K = 200
    r = ld.build_gpu_rank(la, torch, comm, a, bounds, K + 1, reorth=reorth, mode="halo")
    v0 = la.gaussian_fill(1, 0, n)
    r.set_start(v0); r.run(5); r.set_start(v0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.run(K)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"{reorth}: host enqueue {t_host/K*1e6:.1f} us/iter, total {t_all/K*1e6:.1f} us/iter", flush=True)
    # force collectives as if world>1: time an all_reduce of 1 double and of 200 doubles
x = torch.zeros(200, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): dist.all_reduce(x[:1])
t1 = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"host cost of one all_reduce call (world 1): {t1/1000*1e6:.1f} us", flush=True)
dist.destroy_process_group()
