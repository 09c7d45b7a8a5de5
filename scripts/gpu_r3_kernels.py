"""Measurements behind two kernel changes (run on the GPU box from the repository root):

  rows   the row-wise CRS SpMV with its tiles loaded straight into LDS (LANCZOS_HIP_ROWS_DMA=1) against the
         register-staged form: results compared bit for bit on several matrices, then timed on the 256^3
         Laplacian (HIP events per launch, 50 launches each, alternating so that both see the same clocks)
  recur  the recurrence / scaling kernels on whole tiles: SpMV + vector step (reorth=none) and a few full
         steps on 256^3 and 512^3, per-kernel HIP-event time and achieved GB/s on the bytes each kernel moves

    python scripts/gpu_r3_kernels.py rows|recur [grid sizes ...]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lanczos_amd as la  # noqa: E402


def rows_form(sizes):
    os.environ["LANCZOS_HIP_ROWS_PLANES"] = "0"      # the plain row-wise kernel, not the plane walk
    rng = np.random.default_rng(0)
    cases = [("lap3d-40", la.synth_laplacian(3, 40)), ("lap2d-300", la.synth_laplacian(2, 300)),
             ("lap1d-100003", la.synth_laplacian(1, 100003)), ("lap3d-97", la.synth_laplacian(3, 97))]
    out = {"bitwise": {}, "timing": {}}
    for name, a in cases:
        x = rng.standard_normal(a.dimension)
        with la.Engine(a, 3, reorth="none", spmv="rows") as e:
            os.environ["LANCZOS_HIP_ROWS_DMA"] = "0"
            y0 = e.spmv(x)
            os.environ["LANCZOS_HIP_ROWS_DMA"] = "1"
            y1 = e.spmv(x)
        out["bitwise"][name] = bool(y0.tobytes() == y1.tobytes())
        print(name, "bitwise equal:", out["bitwise"][name], flush=True)
    for N in sizes:
        a = la.synth_laplacian(3, N)
        n, nnz = a.dimension, a.length
        b_spmv = 12 * nnz + 4 * (n + 1) + 16 * n
        res = {}
        with la.Engine(a, 4, reorth="none", spmv="rows", profile=True) as e:
            e.set_start_seed(1)
            for rnd in range(3):
                for dma in ("0", "1"):
                    os.environ["LANCZOS_HIP_ROWS_DMA"] = dma
                    e.set_start_seed(1)
                    e.run(3)
                    e.profile_reset()
                    e.run(50)
                    ms = e.profile()["spmv"]["ms"] / 50
                    res.setdefault(dma, []).append(ms)
        for dma, v in res.items():
            best = min(v)
            out["timing"][f"{N}^3 dma={dma}"] = {"ms_per_spmv": v, "best_TBps": b_spmv / best / 1e9}
            print(f"{N}^3 dma={dma}: {v} ms per SpMV, best {b_spmv / best / 1e9:.3f} TB/s on {b_spmv / 1e9:.3f} GB", flush=True)
    return out


def recur_form(sizes):
    out = {}
    for N in sizes:
        a = la.synth_laplacian(3, N)
        n = a.dimension
        with la.Engine(a, 4, reorth="none", profile=True) as e:
            e.set_start_seed(1)
            e.run(5)
            e.profile_reset()
            e.run(100)
            p = e.profile()
            out[f"{N}^3 none"] = {k: {"ms": v["ms"] / 100, "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9}
                                  for k, v in p.items() if v["ms"] > 0}
            out[f"{N}^3 none"]["spmv_kernel"] = e.spmv_name
        if 8.0 * n * 22 < 200e9:
            with la.Engine(a, 21, reorth="full", profile=True) as e:
                e.set_start_seed(1)
                e.run(20)
                e.set_start_seed(1)
                e.profile_reset()
                e.run(20)
                p = e.profile()
                out[f"{N}^3 full20"] = {k: {"ms_per_launch": v["ms"] / max(v["launches"], 1),
                                            "GBps": v["bytes"] / (v["ms"] * 1e-3) / 1e9}
                                        for k, v in p.items() if v["ms"] > 0}
        print(N, json.dumps({k: v for k, v in out.items() if k.startswith(f"{N}^3")}), flush=True)
    return out


def fuse_form(sizes):
    """300 (and 20) steps with full reorthogonalisation: recurrence + dots in one sweep (while the pass is short) against
    two launches; two rounds each, alternating."""
    out = {}
    for N in sizes:
        a = la.synth_laplacian(3, N)
        v0 = la.gaussian_fill(123456789, 0, a.dimension)
        for steps in (20, 300):
            with la.Engine(a, steps + 1, reorth="full", profile=True) as e:
                for rnd in range(2):
                    for tag, env in (("two", {"LANCZOS_HIP_FUSE_RECUR": "0"}), ("fused", {"LANCZOS_HIP_FUSE_RECUR": "1"})):
                        os.environ.update(env)
                        e.set_start(v0)
                        e.run(5)
                        e.set_start(v0)
                        e.profile_reset()
                        import time
                        t0 = time.perf_counter()
                        e.run(steps)
                        dt = time.perf_counter() - t0
                        p = e.profile()
                        rec = {"its_per_s": steps / dt, **{k: round(v["ms"] / steps, 4) for k, v in p.items()}}
                        out.setdefault(f"{N}^3 {steps} {tag}", []).append(rec)
                        print(N, steps, tag, rec, flush=True)
    return out


def dict_cache_form(sizes):
    """The plane-walking dictionary SpMV: 1024 or 2048 rows of a plane per step (LANCZOS_HIP_DICT_SPAN), a wavefront's
    pattern kept in registers or not (LANCZOS_HIP_DICT_CACHE); each combination a kernel of its own, chosen when an engine
    first multiplies.  Same bits on several grids, then HIP-event time per launch on the given grids, an engine per form,
    three rounds."""
    rng = np.random.default_rng(0)
    forms = [("1024", "0"), ("1024", "2"), ("2048", "0"), ("2048", "2")]
    out = {"bitwise": {}, "timing": {}}

    def setenv(span, c):
        os.environ["LANCZOS_HIP_DICT_SPAN"] = span
        os.environ["LANCZOS_HIP_DICT_CACHE"] = c

    for N in (64, 67, 97, 130):
        a = la.synth_laplacian(3, N)
        x = rng.standard_normal(a.dimension)
        ys = []
        for span, c in forms:
            setenv(span, c)
            with la.Engine(a, 3, reorth="none", spmv="dict") as e:
                ys.append(e.spmv(x).tobytes())
                name = e.spmv_name
        same = all(y == ys[0] for y in ys)
        out["bitwise"][f"lap3d-{N} {name}"] = bool(same)
        print(N, name, "bitwise equal:", same, flush=True)
    for N in sizes:
        a = la.synth_laplacian(3, N)
        res = {}
        for rnd in range(3):
            for span, c in forms:
                setenv(span, c)
                with la.Engine(a, 4, reorth="none", spmv="dict", profile=True) as e:
                    e.set_start_seed(1)
                    e.run(5)
                    e.profile_reset()
                    e.run(100)
                    p = e.profile()
                    res.setdefault((span, c), []).append(p["spmv"]["ms"] / 100)
                    bytes_ = p["spmv"]["bytes"] / 100
                    name = e.spmv_name
        for (span, c), v in res.items():
            out["timing"][f"{N}^3 span={span} cache={c}"] = {"ms_per_spmv": v, "best_TBps": bytes_ / min(v) / 1e9}
            print(f"{N}^3 {name} span={span} cache={c}: {[round(t, 4) for t in v]} ms per SpMV, best {bytes_ / min(v) / 1e9:.3f} TB/s", flush=True)
    os.environ.pop("LANCZOS_HIP_DICT_CACHE", None)
    os.environ.pop("LANCZOS_HIP_DICT_SPAN", None)
    return out


if __name__ == "__main__":
    what = sys.argv[1]
    sizes = [int(s) for s in sys.argv[2:]] or ([256] if what == "rows" else [256, 512])
    res = (rows_form(sizes) if what == "rows" else fuse_form(sizes) if what == "fuse" else dict_cache_form(sizes) if what == "dictcache"
           else recur_form(sizes))
    tag = os.environ.get("R3_TAG", what)
    os.makedirs("gpurun_out/r3k", exist_ok=True)
    json.dump(res, open(f"gpurun_out/r3k/{tag}.json", "w"), indent=1)
