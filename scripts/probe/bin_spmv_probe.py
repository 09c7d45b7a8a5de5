"""Feasibility probe: SpMV as two streaming passes (products binned by row block) against the engine's forms.
bin_spmv_probe.py rmat SCALE | rand ROWS"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lanczos_amd as la  # noqa: E402
from matrices import rmat_laplacian_gpu  # noqa: E402

PSH, BSH = int(os.environ.get("PSH", "14")), int(os.environ.get("BSH", "14"))
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libbin_probe_{PSH}_{BSH}.so"))
PANEL, BLOCK, CH = 1 << PSH, 1 << BSH, 1024
E1 = int(os.environ.get("E1", "65536"))
E2 = int(os.environ.get("E2", "262144"))

kind, arg = sys.argv[1], int(sys.argv[2])
t0 = time.perf_counter()
a = la.CrsMatrix(*rmat_laplacian_gpu(arg)) if kind == "rmat" else la.synth_random_symmetric(arg, 10, 7)
n, nnz = a.dimension, a.length
print(f"{kind} {arg}: n={n} nnz={nnz} ({time.perf_counter() - t0:.1f}s)", flush=True)
dev = torch.device("cuda")
rp = torch.from_numpy(a.row_head_indexes.astype(np.int64)).to(dev)
col = torch.from_numpy(a.column_index.astype(np.int64)).to(dev)
val = torch.from_numpy(a.values).to(dev)
row = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
nb, npan = -(-n // BLOCK), -(-n // PANEL)
q, b = col >> PSH, row >> BSH
t0 = time.perf_counter()
order_src = torch.sort(q * nb + b, stable=True)[1]
order_bin = torch.sort(b * npan + q, stable=True)[1]
cb = torch.bincount(b, minlength=nb)
start = torch.cumsum(cb, 0) - cb
padded = (cb + CH - 1) // CH * CH
pstart = torch.cumsum(padded, 0) - padded
total = int(padded.sum().item())
b_bin = b[order_bin]
pos = pstart[b_bin] + (torch.arange(nnz, device=dev) - start[b_bin])
pos_of_crs = torch.empty(nnz, dtype=torch.int64, device=dev)
pos_of_crs[order_bin] = pos
dst = pos_of_crs[order_src].to(torch.int32)
s_val = val[order_src].contiguous()
s_col = (col[order_src] & (PANEL - 1)).to(torch.int16)
b_val = torch.zeros(total, dtype=torch.float64, device=dev)
b_val[pos] = val[order_bin]
b_col = torch.zeros(total, dtype=torch.int32, device=dev)
b_col[pos] = col[order_bin].to(torch.int32)
row16 = torch.zeros(total, dtype=torch.int64, device=dev)
row16[pos] = row[order_bin] & (BLOCK - 1)
del pos_of_crs, order_bin, b_bin
idx = torch.arange(total, device=dev)
head = ((idx & 63) == 0) | (row16 != torch.roll(row16, 1))
h = torch.nonzero(head).squeeze(1)
key = (h >> 10) * BLOCK + row16[h]
ks, perm = torch.sort(key, stable=True)
first = torch.ones_like(ks, dtype=torch.bool)
first[1:] = ks[1:] != ks[:-1]
gstart = torch.cummax(torch.where(first, torch.arange(ks.numel(), device=dev), torch.zeros_like(ks)), 0)[0]
rank = torch.arange(ks.numel(), device=dev) - gstart
assert int(rank.max().item()) < 256
rnd = torch.zeros(total, dtype=torch.uint8, device=dev)
rnd[h[perm]] = rank.to(torch.uint8)
chunk_rounds = torch.zeros(total // CH, dtype=torch.int64, device=dev)
chunk_rounds.scatter_reduce_(0, h[perm] >> 10, rank, "amax")
print(f"build {time.perf_counter() - t0:.1f}s: padded stream {total} ({total / nnz:.3f}x), heads {h.numel() / nnz:.3f} of entries, "
      f"rounds mean {float(chunk_rounds.float().mean()) + 1:.2f} max {int(chunk_rounds.max()) + 1}", flush=True)
chunk_rounds = chunk_rounds.to(torch.uint8)
row16 = row16.to(torch.int16)
del idx, head, h, key, ks, perm, first, gstart, rank

# pass-1 items: each panel's entries in pieces of E1
cq = torch.bincount(q, minlength=npan).cpu().numpy()
sq = np.concatenate([[0], np.cumsum(cq)])
ilo, ihi, ipan = [], [], []
for p in range(npan):
    for lo in range(int(sq[p]), int(sq[p + 1]), E1):
        ilo.append(lo)
        ihi.append(min(lo + E1, int(sq[p + 1])))
        ipan.append(p)
# pass-2 pieces: each block's padded stream in pieces of E2
pst, pad = pstart.cpu().numpy(), padded.cpu().numpy()
plo, phi, bp0 = [], [], [0]
for blk in range(nb):
    for lo in range(int(pst[blk]), int(pst[blk] + pad[blk]), E2):
        plo.append(lo)
        phi.append(min(lo + E2, int(pst[blk] + pad[blk])))
    bp0.append(len(plo))
t = lambda x_, dt: torch.tensor(np.asarray(x_), dtype=dt, device=dev)  # noqa: E731
ilo, ihi, ipan = t(ilo, torch.int64), t(ihi, torch.int64), t(ipan, torch.int32)
plo, phi, bp0 = t(plo, torch.int64), t(phi, torch.int64), t(bp0, torch.int32)
items, pieces = ilo.numel(), plo.numel()
print(f"panel {PANEL} block {BLOCK}: {items} pass-1 items, {pieces} pass-2 pieces over {nb} blocks", flush=True)
prod = torch.zeros(total, dtype=torch.float64, device=dev)
part = torch.zeros(pieces * BLOCK, dtype=torch.float64, device=dev)
y = torch.zeros(n, dtype=torch.float64, device=dev)
xh = np.random.default_rng(0).standard_normal(n)
x = torch.from_numpy(xh).to(dev)
P = lambda z: C.c_void_p(z.data_ptr())  # noqa: E731


def run(which):
    rc = lib.bin_spmv(items, P(ilo), P(ihi), P(ipan), C.c_int64(n), P(s_val), P(s_col), P(dst), P(x), P(prod),
                      pieces, P(plo), P(phi), P(row16), P(rnd), P(chunk_rounds), P(part), P(bp0), P(y), which,
                      P(b_val), P(b_col))
    assert rc == 0, rc


def timed(which, reps=10):
    run(which)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(which)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


TWO = PSH <= 14
run(7 if TWO else 12)
torch.cuda.synchronize()
y1 = y.cpu().numpy().copy()
run(12)
torch.cuda.synchronize()
print("gather form equals the two-pass form bit for bit:", bool(np.array_equal(y1, y.cpu().numpy())), flush=True)
tg = timed(8)
print(f"gather-and-sum in one pass {tg:.3f} ms = {nnz / tg / 1e6:.1f} G entries/s ({total * 15 / tg / 1e6:.0f} GB/s on 15 B/entry)", flush=True)
t1, t2, t3, tall = (timed(1), timed(2), timed(4), timed(7)) if TWO else (1, 1, 1, 1)
print(f"pass 1 {t1:.3f} ms ({nnz * 22 / t1 / 1e6:.0f} GB/s on 22 B/entry)  pass 2 {t2:.3f} ms ({total * 11 / t2 / 1e6:.0f} GB/s on 11 B/entry)  "
      f"combine {t3:.3f} ms  all {tall:.3f} ms = {nnz / tall / 1e6:.1f} G entries/s", flush=True)
del prod, part, s_val, s_col, dst, row16, rnd, row, col, val, q, b, b_val, b_col
torch.cuda.empty_cache()
for form in sys.argv[3:] or ["auto"]:
    with la.Engine(a, 4, reorth="none", spmv=form, profile=True) as e:
        ye = e.spmv(xh)
        err = float(np.max(np.abs(ye - y1)) / np.max(np.abs(ye)))
        e.set_start_seed(1)
        e.run(3)
        e.profile_reset()
        e.run(10)
        ms = e.profile()["spmv"]["ms"] / 10
        print(f"engine {e.spmv_name}: {ms:.3f} ms per SpMV = {nnz / ms / 1e6:.1f} G entries/s; max rel diff of the probe's y {err:.1e}", flush=True)
