#!/usr/bin/env python3
"""Condenses what profiles/run_profile.sh left under gpurun_out/prof_<tag>/ into the files kept in
profiles/<tag>/: the kernel-trace statistics as they are, per-launch PMC averages per kernel, and the
HBM traffic of the headline kernel (read by bench.py).

    python profiles/summarize.py <tag> [bench-json-to-copy]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADLINE = "render_tiles_resident<1>"


def short(name):
    n = name.replace("void mrt::", "").replace("mrt::", "")
    return n.split("(")[0]


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "render_tiles" not in k and "raster" not in k:
                continue
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    summary = {k: {c: {"launches": v[1], "per_launch": v[0] / v[1]} for c, v in sorted(cs.items())} for k, cs in sorted(agg.items())}
    h = summary.get(HEADLINE, {})
    if "SQ_INSTS_VALU" in h and "GRBM_GUI_ACTIVE" in h:
        # 157.3 Tflop/s of fp32 FMAs = 256 CUs x 128 lanes per clock, so a SIMD retires one wave64 VALU instruction every
        # 2 clocks; GRBM_GUI_ACTIVE sums the 8 XCDs
        cycles = h["GRBM_GUI_ACTIVE"]["per_launch"] / 8
        summary["_derived"] = {"headline_kernel": HEADLINE, "gpu_cycles_per_launch": cycles,
                               "valu_issue_utilisation": h["SQ_INSTS_VALU"]["per_launch"] * 2 / (1024 * cycles)}
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    if "FETCH_SIZE" in h and "WRITE_SIZE" in h:
        fk, wk = h["FETCH_SIZE"]["per_launch"], h["WRITE_SIZE"]["per_launch"]
        json.dump({"render_tiles_recursive_1080p_bytes_per_launch": int((fk + wk) * 1024),
                   "fetch_size_kb_per_launch": fk, "write_size_kb_per_launch": wk, "profile": tag,
                   "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over `bench.py --no-cpu --no-extra` "
                           "(profiles/run_profile.sh); raw counter values in KB.  No x2 correction is applied to FETCH_SIZE: the "
                           "reads of this kernel are dword private-memory and shading-attribute accesses, not the 16-byte-per-lane "
                           "streaming reads the correction was calibrated on.  Nearly all of it is the recursive shader's "
                           "private-memory frames and register spills (about 25 dword stores per wave and shading round), "
                           "not scene data: nodes and triangles are read from LDS."},
                  open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    if len(sys.argv) > 2:
        shutil.copy(sys.argv[2], os.path.join(dst, "bench_n1.json"))
    print(json.dumps(summary.get("_derived", {})))


if __name__ == "__main__":
    main()
