#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC counter collections) into the
small text summaries kept under profiles/.

usage: summarize_profile.py <tag> <kernel_stats.csv> [<fetch_counter_collection.csv> <write_counter_collection.csv>]
"""
import collections
import csv
import sys


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
    print("\n".join(out))


if __name__ == "__main__":
    main()
