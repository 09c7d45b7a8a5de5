"""Randomised differential run: many small matrices of varied shape through every SpMV kernel form,
the 16-column SpMM and the device COO->CRS conversion, each against a host evaluation (scipy for the
products, numpy's stable sort for the conversion).  Prints one line per failure and a summary; exit
status 1 when anything failed.  argv: [cases] [first seed]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import lanczos_amd as la

CASES, SEED0 = 120, 0
# "panel/even" and "panel/drawnNN": the tiled form with its schedule forced (whole row blocks launched in step on
# the packed stream / pieces of 2^NN-row blocks drawn from a counter on the unpacked one), whatever the blocks weigh
KINDS = ["auto", "vector", "stream", "pipe", "pipe_nt", "merge", "rows", "panel", "panel/even", "panel/drawn11", "panel/drawn12",
         "panel/drawn13", "panel/drawn14", "dict"]
EPS = np.finfo(float).eps


def draw_matrix(rng):
    """(CrsMatrix, description); rows unsorted, duplicates allowed, like the reference's container"""
    family = rng.choice(["uniform", "heavy", "banded", "stencil", "equal", "sparse"])
    n = int(rng.choice([1, 2, 5, 63, 64, 65, 127, 129, 255, 257, 511, 1025, 4095, 4097,
                        int(rng.integers(1, 20000)), int(rng.integers(1, 3000))]))
    if family == "uniform":
        lens = rng.integers(0, int(rng.integers(1, 70)) + 1, n)
    elif family == "heavy":
        lens = np.minimum((rng.pareto(1.2, n) * 3).astype(np.int64), max(1, 4 * n))
    elif family == "equal":
        lens = np.full(n, int(rng.integers(1, 40)))
    elif family == "sparse":
        lens = (rng.random(n) < 0.1).astype(np.int64) * rng.integers(1, 4, n)
    else:
        lens = rng.integers(1, 12, n)
    if family == "stencil":
        offs = np.unique(np.r_[0, rng.integers(-min(n, 300), min(n, 300) + 1, int(rng.integers(1, 9)))])
        kinds = [rng.standard_normal(offs.size) for _ in range(int(rng.integers(1, 5)))]
        which = rng.integers(0, len(kinds), n)
        r = np.repeat(np.arange(n), offs.size)
        c = r + np.tile(offs, n)
        v = np.concatenate([kinds[w] for w in which]) if n else np.zeros(0)
        ok = (c >= 0) & (c < n)
        r, c, v = r[ok], c[ok], v[ok]
        head = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=n))]).astype(np.int64)
        return la.CrsMatrix(v.astype(np.float64), c.astype(np.int64), head), f"stencil n={n} offs={offs.size}"
    lens = lens.astype(np.int64)
    if lens.sum() == 0:
        lens[rng.integers(0, n)] = 1
    head = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(head[-1])
    if family == "banded":
        w = int(rng.integers(1, 200))
        rows = np.repeat(np.arange(n), lens)
        cols = np.clip(rows + rng.integers(-w, w + 1, nnz), 0, n - 1)
    else:
        cols = rng.integers(0, n, nnz)
    return la.CrsMatrix(rng.standard_normal(nnz), cols.astype(np.int64), head), f"{family} n={n} nnz={nnz}"


def to_scipy(a, magnitude=False):
    # copies: scipy canonicalises in place, and the engine must keep seeing the container as drawn
    v = np.abs(a.values) if magnitude else a.values.copy()
    return sp.csr_matrix((v, a.column_index.copy(), a.row_head_indexes.copy()), shape=(a.dimension, a.dimension))


def main():
    fails, runs, declined = 0, 0, 0
    t0 = time.perf_counter()
    for case in range(CASES):
        rng = np.random.default_rng(SEED0 + case)
        a, what = draw_matrix(rng)
        n = a.dimension
        m = to_scipy(a)
        mabs = to_scipy(a, magnitude=True)   # |values| entry by entry: duplicates must not cancel in the bound
        x = rng.standard_normal(n)
        want = m @ x
        maxlen = int(np.diff(a.row_head_indexes).max())
        tol = (maxlen + 4) * EPS * (mabs @ np.abs(x)) + 1e-300
        for kind in KINDS:
            os.environ.pop("LANCZOS_HIP_TILE_EVEN", None)
            os.environ.pop("LANCZOS_HIP_TILE_ROWBITS", None)
            if "/" in kind:
                os.environ["LANCZOS_HIP_TILE_EVEN"] = "1" if kind.endswith("even") else "0"
                if "drawn" in kind:
                    os.environ["LANCZOS_HIP_TILE_ROWBITS"] = kind[-2:]
            try:
                e = la.Engine(a, 3, spmv=kind.split("/")[0])
            except la.LanczosHipError as ex:
                if "needs" in str(ex):
                    declined += 1
                    continue
                print(f"FAIL create case={SEED0 + case} {what} kind={kind}: {ex}", flush=True)
                fails += 1
                continue
            with e:
                got = e.spmv(x)
                again = e.spmv(x)
                name = e.spmv_name
            runs += 1
            bad = np.abs(got - want) > tol
            if bad.any() or got.tobytes() != again.tobytes():
                i = int(np.argmax(bad))
                print(f"FAIL spmv case={SEED0 + case} {what} kind={kind} ({name}): {int(bad.sum())} rows off, "
                      f"first {i}: {got[i]} vs {want[i]}; repeatable={got.tobytes() == again.tobytes()}", flush=True)
                fails += 1
        # 16-column product
        if n >= la.BLOCK_WIDTH:
            xb = rng.standard_normal((n, la.BLOCK_WIDTH))
            with la.BlockEngine(a, 2) as b:
                b.set_block(0, xb)
                b.spmm(0, 1)
                gotb = b.get_block(1)
            tolb = (maxlen + 4) * EPS * (mabs @ np.abs(xb)) + 1e-300
            runs += 1
            if (np.abs(gotb - m @ xb) > tolb).any():
                print(f"FAIL spmm case={SEED0 + case} {what}", flush=True)
                fails += 1
        # device COO -> CRS: shuffled triplets, both orientations, against a stable host sort
        if a.length:
            rows = np.repeat(np.arange(n), np.diff(a.row_head_indexes))
            perm = rng.permutation(a.length)
            coo = np.zeros(a.length, dtype=la.COO_DTYPE)
            coo["index_row"], coo["index_column"], coo["value"] = rows[perm], a.column_index[perm], a.values[perm]
            for swap in (0, 1):
                key = coo["index_column"] if swap else coo["index_row"]
                other = coo["index_row"] if swap else coo["index_column"]
                order = np.argsort(key, kind="stable")
                with la.Engine.from_coo(coo, n, swap, 3) as e:
                    got = e.download_crs(a.length)
                runs += 1
                if not (np.array_equal(got.column_index, other[order]) and
                        got.values.tobytes() == coo["value"][order].tobytes() and
                        np.array_equal(got.row_head_indexes,
                                       np.concatenate([[0], np.cumsum(np.bincount(key, minlength=n))]))):
                    print(f"FAIL coo case={SEED0 + case} {what} swap={swap}", flush=True)
                    fails += 1
        if case % 10 == 9:
            print(f"... {case + 1} cases, {runs} runs, {declined} declined, {fails} failures, "
                  f"{time.perf_counter() - t0:.0f}s", flush=True)
    print(f"done: {CASES} cases, {runs} runs, {declined} declined, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    if len(sys.argv) > 1:
        CASES = int(sys.argv[1])
    if len(sys.argv) > 2:
        SEED0 = int(sys.argv[2])
    sys.exit(main())
