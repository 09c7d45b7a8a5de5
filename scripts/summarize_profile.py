#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC counter collections) into the
small text summaries kept under profiles/.

usage: summarize_profile.py <tag> <kernel_stats.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv>
                            [<traffic.json> <workload> <steps> <reorth> <bench_line.json>]]

With the last five arguments (the command profiled is `bench.py --main-only --steps <steps>`, whose JSON
line is the last argument) it also writes, for every kernel class,
the ratio of measured HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes) to the algorithmic bytes
of the same launches: what bench.py multiplies its own launches' algorithmic bytes with to report
roofline.traffic at any --steps.  Only the launches of the timed run count (the last `steps` of a class,
steps-1 for the reorthogonalisation kernels; the warm-up launched them before).
"""
import collections
import csv
import json
import sys

CLASS_OF = (("k_spmv", "spmv"), ("k_recur_dots", "dots"), ("k_recur", "recur"), ("k_scale", "scale"), ("k_dots", "dots"),
            ("k_update", "update"))


def short(name):
    name = name.replace("HIP_vector_type<double, 2u>", "double2")
    return name.split("(")[0].replace("void ", "").strip()


def main():
    tag, stats = sys.argv[1], sys.argv[2]
    out = [f"# rocprofv3 summary: {tag}", "", "## --kernel-trace --stats (per kernel)", "",
           "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats)):
        out.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                   f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    if len(sys.argv) >= 5:
        out += ["", "## HBM traffic from PMC (separate passes), per launch",
                "FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced "
                "reads, so `read MB` below is 2 x FETCH_SIZE (calibrated on k_scale/k_dot, whose byte counts are known).", "",
                "| kernel | launches | FETCH_SIZE KiB (avg) | read MB (2x) | WRITE_SIZE KiB (avg) | write MB |",
                "|---|---|---|---|---|---|"]
        agg = collections.defaultdict(lambda: {"f": [], "w": []})
        by_class = collections.defaultdict(lambda: {"f": [], "w": []})   # (dispatch id, value), several kernels a class
        for path, key in ((sys.argv[3], "f"), (sys.argv[4], "w")):
            for r in csv.DictReader(open(path)):
                agg[short(r["Kernel_Name"])][key].append(float(r["Counter_Value"]))
                cls = next((c for name, c in CLASS_OF if name in r["Kernel_Name"]), None)
                if cls:
                    by_class[cls][key].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["f"])):
            if not v["f"]:
                continue
            f = sum(v["f"]) / len(v["f"])
            w = sum(v["w"]) / max(len(v["w"]), 1)
            out.append(f"| {k} | {len(v['f'])} | {f:.0f} | {2*f*1024/1e6:.1f} | {w:.0f} | {w*1024/1e6:.1f} |")
        if len(sys.argv) >= 10:
            path, workload, steps, reorth = sys.argv[5], sys.argv[6], int(sys.argv[7]), sys.argv[8]
            # the JSON line bench.py printed in the kernel-trace pass
            bench = json.loads(open(sys.argv[9]).read().strip().splitlines()[-1])
            alg, launches = bench["kernel_algorithmic_bytes_total"], bench["kernel_launches"]
            measured, ratio = {}, {}
            for cls, v in by_class.items():
                cnt = int(launches.get(cls, 0))
                if cnt <= 0 or len(v["f"]) < cnt or len(v["w"]) < cnt:
                    continue
                f = [x for _, x in sorted(v["f"])[-cnt:]]
                w = [x for _, x in sorted(v["w"])[-cnt:]]
                measured[cls] = 2 * 1024 * sum(f) + 1024 * sum(w)
            for cls, b in measured.items():
                if alg.get(cls, 0) > 0:
                    ratio[cls] = b / alg[cls]
            json.dump({"workload": workload, "steps": steps, "reorth": reorth, "source": tag,
                       "rule": "(2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes, summed over the timed run's launches of a "
                               "class) / (algorithmic bytes of the same launches); separate --pmc passes",
                       "spmv_kernel": bench["config"]["spmv_kernel"],
                       "measured_over_algorithmic": ratio}, open(path, "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main()
