#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection CSV per (kernel, grid): mean counter values, mean
duration, derived effective clock / MFMA utilisation.  usage: pmc_summary.py <counter_collection.csv> [filter]"""
import csv
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    disp = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        key = r["Dispatch_Id"]
        d = disp.setdefault(key, {"name": name[:60], "grid": r["Grid_Size"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    groups = defaultdict(list)
    for d in disp.values():
        groups[(d["name"], d["grid"])].append(d)
    for (name, grid), ds in sorted(groups.items(), key=lambda kv: -sum(x["dur"] for x in kv[1])):
        n = len(ds)
        dur = sum(x["dur"] for x in ds) / n
        print("%s grid=%s calls=%d avg=%.4f ms" % (name, grid, n, dur / 1e6))
        names = sorted({k for x in ds for k in x["c"]})
        c = {k: sum(x["c"].get(k, 0.0) for x in ds) / n for k in names}
        for k in names:
            print("    %-28s %.4g" % (k, c[k]))
        if "GRBM_GUI_ACTIVE" in c and dur > 0:
            # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs of the MI355X
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            clk = cyc / (dur * 1e-9) / 1e9
            print("    -> effective clock %.3f GHz (GRBM_GUI_ACTIVE/8 / duration)" % clk)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                # busy cycles are summed over the 1024 SIMDs (256 CUs x 4); one 32x32x2 f32 MFMA = 64 cycles
                util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
                print("    -> MFMA utilisation = MFMA_BUSY / (cycles * 1024 SIMDs) = %.3f  (= %.1f TFLOP/s fp32 at this clock)"
                      % (util, util * 1024 * 64 * clk * 1e9 / 1e12))


if __name__ == "__main__":
    main()
