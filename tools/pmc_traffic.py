#!/usr/bin/env python3
"""Summarise a tools/profile.sh run into profiles/: per-kernel average duration (kernel-trace stats) and HBM traffic
per launch from the FETCH_SIZE / WRITE_SIZE passes.

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced (16 B/lane) streaming read, so the read side is doubled for our streaming kernels (their
loads are 16 B per lane); WRITE_SIZE is exact for 16-B streaming stores.  Both raw and corrected values are kept.

    python tools/pmc_traffic.py gpurun_out/prof_r01 r01 [out_dir [suffix]]      suffix: "_n25_k6_o1" for a non-headline configuration
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0].split("<")[0]


def main(src, tag, out_dir=None, suffix=""):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = out_dir or os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    summary = {"source": "rocprofv3 on MI355X, tools/profile.sh %s" % tag, "kernels": {}}
    # kernel-trace stats
    stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime, reverse=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(out_dir, "%s_kernel_stats%s.csv" % (tag, suffix)), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        for r in rows:
            k = short(r["Name"])
            if k.startswith("k_"):
                summary["kernels"].setdefault(k, {})["avg_us"] = float(r["AverageNs"]) / 1e3
                summary["kernels"][k]["calls"] = int(r["Calls"])
    # PMC passes
    for counter, key in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        files = sorted(glob.glob(os.path.join(src, key, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime, reverse=True)
        if not files:
            continue
        acc = defaultdict(list)
        for r in csv.DictReader(open(files[0])):
            if r.get("Counter_Name") == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, vals in acc.items():
            if not k.startswith("k_"):
                continue
            vals = vals[len(vals) // 10:]  # drop warm-up launches
            summary["kernels"].setdefault(k, {})[counter + "_KiB_raw_avg"] = sum(vals) / len(vals)
    for k, d in summary["kernels"].items():
        if "FETCH_SIZE_KiB_raw_avg" in d and "WRITE_SIZE_KiB_raw_avg" in d:
            d["hbm_bytes_per_launch_raw"] = (d["FETCH_SIZE_KiB_raw_avg"] + d["WRITE_SIZE_KiB_raw_avg"]) * 1024
            d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE_KiB_raw_avg"] + d["WRITE_SIZE_KiB_raw_avg"]) * 1024
    for name in ("bench.json", "bench_under_trace.json"):
        pth = os.path.join(src, name)
        if os.path.exists(pth) and os.path.getsize(pth):
            try:
                summary[name] = json.loads(open(pth).read().strip().splitlines()[-1])
            except Exception:
                pass
    with open(os.path.join(out_dir, "%s_pmc_traffic%s.json" % (tag, suffix)), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary["kernels"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "r01", sys.argv[3] if len(sys.argv) > 3 else None,
         sys.argv[4] if len(sys.argv) > 4 else "")
