#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): the same kernel template serves
several layer shapes, so the per-shape average is what bench.py's `roofline.avg_launch_ms` must
agree with.  usage: summarize_trace.py <kernel_trace.csv> [out.md]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if len(name) > 70:
                name = name[:67] + "..."
            grid = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?")
            wg = (r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "?")
            rows[(name, grid, wg, r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in rows.values())
    lines = ["| kernel | grid(threads) | wg | LDS B | VGPR | AGPR | calls | avg ms | min ms | max ms | % time |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| `%s` | %s | %s | %s | %s | %s | %d | %.4f | %.4f | %.4f | %.2f |" % (
            k[0], k[1], k[2], k[3], k[4], k[5], len(v), sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6,
            100.0 * sum(v) / total))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out)
    print(out)


if __name__ == "__main__":
    main()
