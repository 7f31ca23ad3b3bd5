#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (profiles/run_profile_r02.sh) into profiles/<tag>/: the kernel-trace statistics as
they are, per-launch PMC averages per configuration and kernel, and profiles/hbm_traffic.json (memory-side bytes per
launch of every configuration's dominant kernel, read by bench.py).

    python profiles/summarize_r02.py <tag> [bench-json-to-copy]
"""
import collections
import csv
import glob
import json
import re
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = {"c2": "render_tiles_resident<1, true>", "c2r": "render_tiles_resident<1, false>", "c2p": "render_tiles_two_trees_packet<true>", "c4": "render_tiles_two_trees_packet<true>",
            "c3": "render_tiles_global<0, 32>", "c5": "raster_tile<4>"}


def short(name):
    name = name.replace("void mrt::", "").replace("mrt::", "").split("(")[0]
    # render_tiles_global<mode, stack entries, analytic objects, group boxes>: the last two do not name a configuration
    return re.sub(r"render_tiles_global<(\d+), (\d+)[^>]*>", r"render_tiles_global<\1, \2>", name)


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    headline = os.path.join(src, "trace_headline", "trace_kernel_stats.csv")
    if os.path.exists(headline):
        shutil.copy(headline, os.path.join(dst, "kernel_stats_headline.csv"))
    summary = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*_*", "pmc_counter_collection.csv"))):
        cfg = os.path.basename(os.path.dirname(d)).split("_")[1]
        agg = summary.setdefault(cfg, collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0])))
        for r in csv.DictReader(open(d)):
            k = short(r["Kernel_Name"])
            if "render_tiles" not in k and "raster" not in k:
                continue
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    out = {cfg: {k: {c: {"launches": v[1], "per_launch": v[0] / v[1]} for c, v in sorted(cs.items())} for k, cs in sorted(ks.items())}
           for cfg, ks in sorted(summary.items())}
    derived, traffic = {}, {"profile": tag, "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE (separate passes over scripts/config_probe.py, "
                            "profiles/run_profile_r02.sh), raw counter values in KB converted to bytes per launch.  These are the L2's memory-side "
                            "requests, Infinity-Cache hits included.  FETCH_SIZE is NOT doubled: the x2 correction holds for 16-byte-per-lane streaming "
                            "reads, and the reads here are 16- and 32-byte node and triangle records fetched one per wave (scalar loads) or one per lane, "
                            "shading attributes and texels.  One kernel is such a stream: raster_setup reads its gathered triangle arrays 16 bytes per lane, so its "
                            "share of c5.frame_bytes_all_kernels (54 MB of fetches in r03a) stands for about twice that."}
    for cfg, kern in DOMINANT.items():
        h = out.get(cfg, {}).get(kern)
        if not h:
            continue
        e = {}
        if "SQ_INSTS_VALU" in h and "GRBM_GUI_ACTIVE" in h:
            cycles = h["GRBM_GUI_ACTIVE"]["per_launch"] / 8          # the counter sums the 8 XCDs
            e["gpu_cycles_per_launch"] = cycles
            e["valu_issue_utilisation"] = h["SQ_INSTS_VALU"]["per_launch"] * 2 / (1024 * cycles)   # a SIMD retires one wave64 VALU instruction per 2 clocks
        if "SQ_THREAD_CYCLES_VALU" in h and "SQ_ACTIVE_INST_VALU" in h:
            e["active_lanes_per_valu_instruction"] = h["SQ_THREAD_CYCLES_VALU"]["per_launch"] / h["SQ_ACTIVE_INST_VALU"]["per_launch"]
        if "SQ_WAIT_ANY" in h and "SQ_WAVE_CYCLES" in h:
            e["wave_time_waiting_on_counters"] = h["SQ_WAIT_ANY"]["per_launch"] / h["SQ_WAVE_CYCLES"]["per_launch"]
        if "FETCH_SIZE" in h and "WRITE_SIZE" in h:
            fk, wk = h["FETCH_SIZE"]["per_launch"], h["WRITE_SIZE"]["per_launch"]
            traffic[cfg] = {"kernel": kern, "bytes_per_launch": int((fk + wk) * 1024), "fetch_bytes": int(fk * 1024), "write_bytes": int(wk * 1024)}
            if cfg == "c5":      # the rasterizer frame is several kernels: all of them
                # (raster_gather runs once per scene upload, not per frame)
                tot_f = sum(v["FETCH_SIZE"]["per_launch"] for k, v in out[cfg].items() if "FETCH_SIZE" in v and "gather" not in k)
                tot_w = sum(v["WRITE_SIZE"]["per_launch"] for k, v in out[cfg].items() if "WRITE_SIZE" in v and "gather" not in k)
                traffic[cfg].update({"frame_bytes_all_kernels": int((tot_f + tot_w) * 1024)})
        derived[cfg] = dict(kernel=kern, **e)
    out["_derived"] = derived
    json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    if "c2" in traffic:
        traffic["render_tiles_recursive_1080p_bytes_per_launch"] = traffic["c2"]["bytes_per_launch"]
    json.dump(traffic, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
    if len(sys.argv) > 2:
        shutil.copy(sys.argv[2], os.path.join(dst, "bench_n1.json"))
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
