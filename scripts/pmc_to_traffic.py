#!/usr/bin/env python
"""Turn the rocprofv3 --pmc passes of scripts/collect_pmc.sh into profiles/traffic.json: per roofline kernel the HBM-side
bytes per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes -- FETCH_SIZE / WRITE_SIZE are in KiB;
FETCH_SIZE tallies the 128-B fabric read requests of wide coalesced streams at 64 B on gfx950 and is doubled.  These are
L2<->fabric bytes: Infinity-Cache hits are included, so they bound the HBM bytes from above.
usage: pmc_to_traffic.py <dir with NAME.TAG/ counter CSVs> <out.json>"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csv.field_size_limit(1 << 30)

KERNELS = {   # key in traffic.json -> (pass-name prefix, kernel-name filter(s), algorithmic bytes per launch at B=24)
    # the GEMM stage runs as two launches: full rounds with 256-row blocks (<4, .>) and the last partial round as 128-row blocks (<2, .>)
    # F(6x6,3x3): T = 24 * 11 * 11 = 2904 tiles, 64 planes; 11 whole rounds of 256-row blocks (<4, 3>) + the ragged last block as 128-row items (<2, 3>)
    # (round 5: from the `bf3` pass -- scripts/bf3_check.py times the exact-fp32 stages first, then the split ones, on the same shape)
    "wino63_gemm_res2": ("bf3", ["wino43_gemm_kernel<4, 3>", "wino43_gemm_kernel<2, 3>"], 64 * 2904 * (1024 + 1024) * 4 + 64 * 1024 * 1024 * 4),
    "wino63_input_res2": ("wino63", ["wino_input_kernel"], 24 * 64 * 64 * 1024 * 4 + 64 * 2904 * 1024 * 4),
    "wino63_output_res2": ("wino63", ["wino_output_kernel"], 64 * 2904 * 1024 * 4 + 24 * 64 * 64 * 1024 * 4),
    # split (bf16x3) route, same shape: V and U are 6 bytes per element (three bf16 pieces), M stays fp32
    "wino63_gemm_bf3_res2": ("bf3", ["rnf::FmtB3, 4, 2,", "rnf::FmtB3, 2, 2,"], 64 * 2904 * (1024 * 6 + 1024 * 4) + 64 * 1024 * 1024 * 6),
    "wino63_input_bf3_res2": ("bf3", ["wino_input_bf3_kernel"], 24 * 64 * 64 * 1024 * 4 + 64 * 2904 * 1024 * 6),
    # fp16x2 route: V and U are 4 bytes per element (two fp16 pieces of the scaled value)
    "wino63_gemm_h2_res2": ("bf3", ["rnf::FmtH2, 4, 2,", "rnf::FmtH2, 2, 2,"], 64 * 2904 * (1024 * 4 + 1024 * 4) + 64 * 1024 * 1024 * 4),
    "wino63_input_h2_res2": ("bf3", ["wino_input_h2_kernel", "absmax_kernel"], 24 * 64 * 64 * 1024 * 4 * 2 + 64 * 2904 * 1024 * 4),
    "wino43_gemm_res2": ("wino43", ["wino43_gemm_kernel<4, 0>", "wino43_gemm_kernel<2, 0>"], 36 * 6144 * (1024 + 1024) * 4 + 36 * 1024 * 1024 * 4),
    "wino43_input_res2": ("wino43", ["wino_input_kernel"], 24 * 64 * 64 * 1024 * 4 + 36 * 6144 * 1024 * 4),
    "wino43_output_res2": ("wino43", ["wino_output_kernel"], 36 * 6144 * 1024 * 4 + 2 * 24 * 64 * 64 * 1024 * 4),
    "conv3d_drun_res1": ("res1", ["conv3d_k3_drun"], 24 * 64 * 64 * 32 * 32 * 4 * 2 + 27 * 32 * 32 * 4),
    "conv_wino_res1": ("res1w", ["conv_wino_kernel"], 24 * 64 * 64 * 32 * 32 * 4 * 2 + 16 * 3 * 32 * 32 * 4),
    "conv3d_wino_bf3_res1": ("res1s", ["conv3d_wino_bf3_kernel"], 24 * 64 * 64 * 32 * 32 * 4 * 2 + 16 * 3 * 32 * 32 * 6),
    "resampler": ("resample", ["resample_prepare", "resample_classify", "resample_main"], 24 * 9437184),
}


def per_launch(path, filters):
    """{kernel filter: {counter: mean per dispatch}} over the dispatches of the LARGEST grid matching each filter."""
    disp = {}
    for r in csv.DictReader(open(path)):
        for f in filters:
            if f in r["Kernel_Name"]:
                d = disp.setdefault((f, r["Dispatch_Id"]), {"grid": int(r["Grid_Size"]), "c": {},
                                                              "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
                d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = {}
    for f in filters:
        ds = [d for (ff, _), d in disp.items() if ff == f]
        if not ds:
            continue
        g = max(d["grid"] for d in ds)
        ds = [d for d in ds if d["grid"] == g]
        names = {k for d in ds for k in d["c"]}
        out[f] = {k: sum(d["c"].get(k, 0.0) for d in ds) / len(ds) for k in names}
        out[f]["_avg_ms"] = sum(d["dur"] for d in ds) / len(ds) / 1e6
        out[f]["_calls"] = len(ds)
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    from bench import kernel_sources_digest
    res = {"_comment": __doc__.split("usage")[0].strip(), "kernels": {}}
    for key, (name, filters, alg) in KERNELS.items():
        c = {}
        for tag in ("sq", "fetch", "write", "tcc"):
            fs = glob.glob(os.path.join(src, "%s.%s" % (name, tag), "**", "*counter_collection.csv"), recursive=True)
            if fs:
                for f, vals in per_launch(fs[0], filters).items():
                    c.setdefault(f, {}).update({k if not k.startswith("_") else "%s%s" % (tag, k): v for k, v in vals.items()})
        if not c:
            continue
        rd = sum(2.0 * 1024.0 * v.get("FETCH_SIZE", 0.0) for v in c.values())
        wr = sum(1024.0 * v.get("WRITE_SIZE", 0.0) for v in c.values())
        ent = {"kernels": filters, "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
               "algorithmic_bytes_per_launch": alg, "csrc_digest": kernel_sources_digest(), "git": os.environ.get("GIT_REV", "unknown")}
        hit = sum(v.get("TCC_HIT_sum", 0.0) for v in c.values())
        miss = sum(v.get("TCC_MISS_sum", 0.0) for v in c.values())
        if hit + miss > 0:
            ent["l2_hit_rate"] = hit / (hit + miss)
        for f, v in c.items():
            if "GRBM_GUI_ACTIVE" in v and v.get("sq_avg_ms"):
                cyc = v["GRBM_GUI_ACTIVE"] / 8.0
                ent.setdefault("per_kernel", {})[f] = {
                    "avg_launch_ms_profiled": v["sq_avg_ms"], "effective_clock_ghz": cyc / (v["sq_avg_ms"] * 1e-3) / 1e9,
                    "mfma_utilisation": v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024.0),
                    "lds_bank_conflict_cycles": v.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": v.get("SQ_LDS_IDX_ACTIVE")}
        res["kernels"][key] = ent
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
