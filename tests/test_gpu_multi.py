"""Several row slabs driven from one host process (lanczos_hip_multi_*, LANCZOS_HIP_NGPU).

The reference has no multi-GPU code; the slabs must reproduce the single-GPU run, which the other
tests tie to the oracle.  A one-GPU box holds every slab on cuda:0 (loopback transport: copies and
a summing kernel where a node runs RCCL), which exercises the partition, the column remap, both
exchange forms, the split product and the iteration order; RCCL itself runs with one slab.
Tolerances: alpha/beta 1e-11 relative to the single engine (sums are taken in another order),
Ritz values 1e-10 relative to the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import lanczos_amd as la
import oracle
from matrices import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _slab_affinity_checked(monkeypatch):
    """Every call into a slab's engine is checked against the slab the driver made current: on this
    one-GPU box all slabs share a device, so HIP itself would accept a call made under the wrong one."""
    monkeypatch.setenv("LANCZOS_HIP_CHECK_AFFINITY", "1")


def single(a, v0, steps, reorth):
    with la.Engine(a, steps + 1, reorth=reorth) as e:
        e.set_start(v0)
        return e.run(steps)


@pytest.mark.parametrize("reorth", ["full", "full2", "cuda_compat", "none", "selective"])
def test_loopback_slabs_match_the_single_engine_and_the_oracle(gpu, reorth):
    steps = 60
    for a, slabs, want_halo in ((la.synth_laplacian(3, 24), 3, True), (la.synth_laplacian(2, 90), 2, True),
                                (la.synth_random_symmetric(20000, 6, 5), 3, False)):
        v0 = la.gaussian_fill(7, 0, a.dimension)
        a1, b1 = single(a, v0, steps, reorth)
        with la.MultiEngine(a, steps + 1, slabs, reorth=reorth, loopback=True) as m:
            info = m.info()
            assert info["slabs"] == slabs and bool(info["halo"]) == want_halo
            if want_halo:
                assert info["overlapped"] == 1       # stencil slabs: the split product is exercised
            m.set_start(v0)
            a2, b2 = m.run(steps)
            assert m.info()["iterations"] == steps
        # without reorthogonalisation rounding differences grow with the loss of orthogonality: the
        # sequences are compared while they are still determined (25 steps), the others throughout
        cmp = 25 if reorth == "none" else steps
        assert rel_err(a2[:cmp], a1[:cmp]) < 1e-11 and rel_err(b2[:cmp], b1[:cmp]) < 1e-11, (reorth, slabs)
        o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, cmp + 1, 0.0,
                               reorth if reorth in ("full", "none", "selective", "cuda_compat") else "full")
        assert rel_err(la.ritz_values(a2, b2, cmp, 4), o.eigenvalues) < 1e-10, (reorth, slabs)


@pytest.mark.parametrize("walks", ["0", "1"])
@pytest.mark.parametrize("span", ["1024", "2048"])
@pytest.mark.parametrize("reorth", ["full", "none"])
def test_loopback_slabs_of_whole_planes_walk_them(gpu, reorth, span, walks, monkeypatch):
    """64^3 cut into 2 and 4 slabs of whole planes: every slab's dictionary SpMV walks its planes, with the
    neighbours' planes (received into the vectors' tails) as the plane before its first and after its last;
    with 1024 and with 2048 rows of a plane per step (the form long grid lines take by default).  walks = 1: without
    reorthogonalisation the slabs' iteration is two plane walks (large slabs' default), against product + recurrence
    kernel on one engine."""
    if walks == "1" and reorth != "none":
        pytest.skip("the two walks are the iteration without reorthogonalisation")
    monkeypatch.setenv("LANCZOS_HIP_DICT_SPAN", span)
    a = la.synth_laplacian(3, 64)
    v0 = la.gaussian_fill(5, 0, a.dimension)
    steps = 30
    monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", "0")
    a1, b1 = single(a, v0, steps, reorth)
    monkeypatch.setenv("LANCZOS_HIP_TWO_PASS", walks)
    for slabs in (2, 4):
        with la.MultiEngine(a, steps + 1, slabs, reorth=reorth, loopback=True) as m:
            assert m.info()["halo"] == 1
            assert all(m.slab_spmv_name(s).startswith("dict") and m.slab_spmv_name(s).endswith("p") for s in range(slabs))
            m.set_start(v0)
            a2, b2 = m.run(steps)
        cmp = 25 if reorth == "none" else steps
        assert rel_err(a2[:cmp], a1[:cmp]) < 1e-11 and rel_err(b2[:cmp], b1[:cmp]) < 1e-11, (reorth, slabs)


@pytest.mark.parametrize("spmv", ["panel", "merge"])
def test_loopback_slabs_on_the_gather_bound_forms(gpu, spmv):
    """Slabs whose columns reach all over the vector (all-gather layout), on the tiled and the
    share-of-entries SpMV: a random matrix and a power-law graph against the single engine."""
    from matrices import rmat_laplacian
    m16 = rmat_laplacian(14).tocsr()
    graph = la.CrsMatrix(m16.data.astype(np.float64), m16.indices.astype(np.int64), m16.indptr.astype(np.int64))
    for a in (la.synth_random_symmetric(30000, 8, 2), graph):
        v0 = la.gaussian_fill(11, 0, a.dimension)
        a1, b1 = single(a, v0, 30, "full")
        with la.MultiEngine(a, 31, 3, reorth="full", spmv=spmv, loopback=True) as m:
            assert m.info()["halo"] == 0
            assert all(m.slab_spmv_name(s).startswith(spmv) for s in range(3))
            m.set_start(v0)
            a2, b2 = m.run(30)
        assert rel_err(a2, a1) < 1e-11 and rel_err(b2, b1) < 1e-11, spmv


def test_loopback_forced_allgather_and_unsplit_product(gpu, monkeypatch):
    a = la.synth_laplacian(3, 20)
    v0 = la.gaussian_fill(3, 0, a.dimension)
    a1, b1 = single(a, v0, 40, "full")
    monkeypatch.setenv("LANCZOS_HIP_MULTI_ALLGATHER", "1")
    with la.MultiEngine(a, 41, 4, loopback=True) as m:
        assert m.info()["halo"] == 0
        m.set_start(v0)
        a2, b2 = m.run(40)
    assert rel_err(a2, a1) < 1e-11 and rel_err(b2, b1) < 1e-11
    monkeypatch.delenv("LANCZOS_HIP_MULTI_ALLGATHER")
    monkeypatch.setenv("LANCZOS_HIP_MULTI_NO_OVERLAP", "1")
    with la.MultiEngine(a, 41, 4, loopback=True) as m:
        assert m.info()["halo"] == 1 and m.info()["overlapped"] == 0
        m.set_start(v0)
        a3, b3 = m.run(40)
    assert rel_err(a3, a1) < 1e-11 and rel_err(b3, b1) < 1e-11
    # runs continue over several calls; a second start vector resets the run
    monkeypatch.delenv("LANCZOS_HIP_MULTI_NO_OVERLAP")
    with la.MultiEngine(a, 41, 2, loopback=True, reorth="none") as m:
        m.set_start(v0)
        pa, pb = m.run(15)
        qa, qb = m.run(25)
        an, bn = single(a, v0, 40, "none")
        assert rel_err(np.concatenate([pa, qa])[:25], an[:25]) < 1e-11
        assert rel_err(np.concatenate([pb, qb])[:25], bn[:25]) < 1e-11
        m.set_start(v0)
        ra, rb = m.run(40)
        assert np.array_equal(ra, np.concatenate([pa, qa]))


def test_slabs_built_by_one_host_thread_each(gpu, monkeypatch):
    """LANCZOS_HIP_MULTI_BUILD_THREADS=1: every slab's column remap, upload and engine in a host thread of its own
    (the default on several devices; forced here on the loopback transport): same run as the slabs built in turn."""
    a = la.synth_laplacian(3, 32)
    v0 = la.gaussian_fill(4, 0, a.dimension)
    runs = {}
    for threads in ("0", "1"):
        monkeypatch.setenv("LANCZOS_HIP_MULTI_BUILD_THREADS", threads)
        with la.MultiEngine(a, 31, 4, reorth="full", loopback=True) as m:
            st = m.build_stats()
            assert st["host_threads"] == (4 if threads == "1" else 1) and st["seconds"] > 0
            m.set_start(v0)
            runs[threads] = m.run(30)
    assert np.array_equal(runs["0"][0], runs["1"][0]) and np.array_equal(runs["0"][1], runs["1"][1])
    # an error inside a slab's thread comes back as the call's own message
    monkeypatch.setenv("LANCZOS_HIP_MULTI_BUILD_THREADS", "1")
    with pytest.raises(la.LanczosHipError, match="rows SpMV needs every 64-row group"):
        la.MultiEngine(la.synth_random_symmetric(40000, 40, 1), 5, 2, spmv="rows", loopback=True)


def test_loopback_eight_slabs(gpu):
    """The slab count of a whole node: 4096-row slabs of a 32^3 Laplacian whose 1024-entry halos are just
    within half a slab (halo form), and the 32-row slabs of a 16 x 16 grid, too thin for it (all-gather form)."""
    for a, want_halo in ((la.synth_laplacian(3, 32), 1), (la.synth_laplacian(2, 16), 0)):
        v0 = la.gaussian_fill(9, 0, a.dimension)
        for reorth in ("full", "none"):
            a1, b1 = single(a, v0, 30, reorth)
            with la.MultiEngine(a, 31, 8, reorth=reorth, loopback=True) as m:
                assert m.info()["halo"] == want_halo
                m.set_start(v0)
                a2, b2 = m.run(30)
            cmp = 20 if reorth == "none" else 30
            assert rel_err(a2[:cmp], a1[:cmp]) < 1e-11 and rel_err(b2[:cmp], b1[:cmp]) < 1e-11, (reorth, want_halo)


def test_rccl_transport_with_one_slab(gpu):
    """ncclCommInitAll over the one device of this box: every collective of an iteration goes through
    RCCL with world size 1; the sequences are the single engine's to rounding (1e-12 relative)."""
    a = la.synth_laplacian(3, 32)
    v0 = la.gaussian_fill(11, 0, a.dimension)
    for reorth in ("none", "full", "selective"):
        a1, b1 = single(a, v0, 50, reorth)
        with la.MultiEngine(a, 51, 1, reorth=reorth, loopback=False) as m:
            m.set_start(v0)
            a2, b2 = m.run(50)
        cmp = 25 if reorth == "none" else 50
        assert rel_err(a2[:cmp], a1[:cmp]) < 1e-12 and rel_err(b2[:cmp], b1[:cmp]) < 1e-12, reorth


def test_more_gpus_than_visible_and_mgs_are_refused(gpu):
    a = la.synth_laplacian(1, 5000)
    import torch
    with pytest.raises(la.LanczosHipError, match="more GPUs than this process sees"):
        la.MultiEngine(a, 11, torch.cuda.device_count() + 1)
    with pytest.raises(la.LanczosHipError, match="mgs"):
        la.MultiEngine(a, 11, 2, reorth="mgs", loopback=True)
    with pytest.raises(la.LanczosHipError, match="fewer rows"):
        la.MultiEngine(la.synth_laplacian(1, 3), 3, 4, loopback=True)


def _dropin(a, nth, max_iter, thr, env):
    code = ("import sys, json, numpy as np, lanczos_amd as la\n"
            "d=np.load(sys.argv[1])\n"
            "a=la.CrsMatrix(d['v'],d['c'],d['h'])\n"
            f"rc,eig=la.lanczos_hip_crs(a,{nth},{max_iter},{thr!r})\n"
            "sys.stdout.flush()\n"
            "print('RESULT '+json.dumps({'rc':rc,'eig':[float(x) for x in eig]}))\n")
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.npz")
        np.savez(path, v=a.values, c=a.column_index, h=a.row_head_indexes)
        r = subprocess.run([sys.executable, "-c", code, path], cwd=ROOT, capture_output=True, text=True,
                           env=dict(os.environ, **env), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][0][7:])
    return res, r.stdout, r.stderr


@pytest.mark.parametrize("reorth", ["full", "none", "selective"])
def test_dropin_touches_slab_zero_only_with_its_device_current(gpu, reorth):
    """The drop-in's own host loop records events on slab 0's stream between calls that walk over all
    slabs' devices.  With three slabs and the affinity check on, an event created or recorded while
    another slab is current (what HIP refuses on a node with one GPU per slab) ends the run with a message."""
    a = la.synth_laplacian(3, 24)
    v0 = la.gaussian_fill(123456789, 0, a.dimension)
    max_iter = 26 if reorth == "none" else 41   # without reorthogonalisation only the first steps are determined
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, max_iter, 0.0, reorth)
    res, out, err = _dropin(a, 4, max_iter, 0.0, {"LANCZOS_HIP_NGPU": "3", "LANCZOS_HIP_MULTI_LOOPBACK": "1",
                                                  "LANCZOS_HIP_REORTH": reorth, "LANCZOS_HIP_CHECK_AFFINITY": "1"})
    assert "is current" not in out, out[-500:]
    assert res["rc"] == 0 and f"Reached max iterations ({max_iter})" in out
    assert rel_err(res["eig"], o.eigenvalues) < 1e-10, reorth


def test_dropin_honours_ngpu(gpu):
    """lanczos_hip_crs with LANCZOS_HIP_NGPU: same lines, same exits, same values as on one GPU."""
    a = la.synth_laplacian(2, 100)
    v0 = la.gaussian_fill(123456789, 0, a.dimension)
    o = oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 101, 0.0)
    for reorth in ("full", "selective"):
        oo = o if reorth == "full" else oracle.lanczos_crs(a.row_head_indexes, a.column_index, a.values, v0, 4, 101,
                                                           0.0, reorth)
        res, out, err = _dropin(a, 4, 101, 0.0, {"LANCZOS_HIP_NGPU": "3", "LANCZOS_HIP_MULTI_LOOPBACK": "1",
                                                 "LANCZOS_HIP_REORTH": reorth, "LANCZOS_HIP_JSON": "1"})
        assert res["rc"] == 0 and rel_err(res["eig"], oo.eigenvalues) < 1e-10, reorth
        assert "[INIT] name=row_slabs count=3 transport=loopback exchange=halo" in err and " sec=" in err
        step_lines = [ln for ln in out.splitlines() if ln[:1].isdigit() and "\t" in ln]
        assert len(step_lines) == 100 and "Reached max iterations (101)" in out
        assert step_lines[49] == "50\t" + "\t".join(f"{x:.7f}" for x in oo.ritz_hist[49]) + "\t"
        rec = json.loads([ln for ln in err.splitlines() if ln.startswith('{"lanczos_hip"')][0])["lanczos_hip"]
        assert rec["iterations"] == 100 and rec["exit_kind"] == 0
    # the convergence exit through the slabs, and RCCL with one slab behind the same entry
    thr = 1e-2
    b = la.synth_laplacian(1, 2000)
    ob = oracle.lanczos_crs(b.row_head_indexes, b.column_index, b.values, la.gaussian_fill(123456789, 0, 2000), 2, 400, thr)
    res, out, err = _dropin(b, 2, 400, thr, {"LANCZOS_HIP_NGPU": "2", "LANCZOS_HIP_MULTI_LOOPBACK": "1",
                                             "LANCZOS_HIP_JSON": "1"})
    rec = json.loads([ln for ln in err.splitlines() if ln.startswith('{"lanczos_hip"')][0])["lanczos_hip"]
    assert res["rc"] == 0 and rec["exit_kind"] == ob.exit_kind == 1 and rec["exit_step"] == ob.steps
    # RCCL with one slab behind the drop-in: its version banner must not land among the solver's stdout lines
    res, out, err = _dropin(a, 4, 31, 0.0, {"LANCZOS_HIP_NGPU": "1", "LANCZOS_HIP_MULTI_LOOPBACK": "0",
                                            "LANCZOS_HIP_MULTI_FORCE": "1"})
    assert res["rc"] == 0 and "transport=rccl" in err and "RCCL version" not in out
    # asking for more GPUs than there are fails cleanly with EXIT_FAILURE and a message
    import torch
    res, out, _ = _dropin(a, 4, 21, 0.0, {"LANCZOS_HIP_NGPU": str(torch.cuda.device_count() + 1)})
    assert res["rc"] == 1 and "more GPUs than this process sees" in out
    # checkpoints are refused there
    res, out, _ = _dropin(a, 4, 21, 0.0, {"LANCZOS_HIP_NGPU": "2", "LANCZOS_HIP_MULTI_LOOPBACK": "1",
                                          "LANCZOS_HIP_CHECKPOINT": "/tmp/x.ckpt", "LANCZOS_HIP_CHECKPOINT_EVERY": "5"})
    assert res["rc"] == 1 and "whole basis on one GPU" in out
