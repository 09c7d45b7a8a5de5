#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC counter collections) into the
small text summaries kept under profiles/.

usage: summarize_profile.py <tag> <kernel_stats.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv>
                            [<traffic.json> <workload> <steps> <reorth>]]

With the last four arguments it also writes the per-launch HBM bytes of every kernel class
(2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes) that bench.py reports as roofline.traffic.  For the
reorthogonalisation kernels only the last steps-1 launches count (those of the timed run; the
warm-up iterations launched them before with a nearly empty basis).
"""
import collections
import csv
import json
import sys

CLASS_OF = (("k_spmv", "spmv"), ("k_recur", "recur"), ("k_scale", "scale"), ("k_dots", "dots"),
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
        for path, key in ((sys.argv[3], "f"), (sys.argv[4], "w")):
            for r in csv.DictReader(open(path)):
                agg[short(r["Kernel_Name"])][key].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]["f"])):
            if not v["f"]:
                continue
            f = sum(v["f"]) / len(v["f"])
            w = sum(v["w"]) / max(len(v["w"]), 1)
            out.append(f"| {k} | {len(v['f'])} | {f:.0f} | {2*f*1024/1e6:.1f} | {w:.0f} | {w*1024/1e6:.1f} |")
        if len(sys.argv) >= 9:
            path, workload, steps, reorth = sys.argv[5], sys.argv[6], int(sys.argv[7]), sys.argv[8]
            per = {}
            for k, v in agg.items():
                cls = next((c for key, c in CLASS_OF if key in k), None)
                if cls is None or not v["f"] or not v["w"]:
                    continue
                f, w = v["f"], v["w"]
                if cls in ("dots", "update"):
                    f, w = f[-(steps - 1):], w[-(steps - 1):]
                per[cls] = 2 * 1024 * sum(f) / len(f) + 1024 * sum(w) / len(w)
            json.dump({"workload": workload, "steps": steps, "reorth": reorth, "source": tag,
                       "rule": "2 x FETCH_SIZE + WRITE_SIZE (KiB), averaged per launch; separate --pmc passes",
                       "bytes_per_launch": per}, open(path, "w"), indent=1)
    print("\n".join(out))


if __name__ == "__main__":
    main()
