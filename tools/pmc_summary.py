#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (tools/pmc_passes.sh) per kernel and grid size.

    python tools/pmc_summary.py gpurun_out/<tag> [name-filter]
Prints, per (kernel, grid) group, the mean of every counter per launch and the launch
duration taken from the same pass's kernel trace.
"""
import collections
import csv
import glob
import sys


def load(tag_dir, name):
    cc = glob.glob(f"{tag_dir}/{name}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{tag_dir}/{name}/**/*kernel_trace.csv", recursive=True)
    if not cc:
        return {}, {}
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(cc[0])):
        kname = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = (kname, r["Grid_Size"] if "Grid_Size" in r else "")
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            durs[key].append(dur.get(r["Dispatch_Id"], 0.0))
    return vals, durs


def main():
    tag = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    merged = collections.defaultdict(dict)
    dd = {}
    for name in ("fetch", "write", "sq", "sq2"):
        vals, durs = load(tag, name)
        for key, cs in vals.items():
            for c, v in cs.items():
                merged[key][c] = sum(v) / len(v)
            dd.setdefault(key, sum(durs[key]) / max(len(durs[key]), 1))
    for key in sorted(merged, key=lambda k: -dd.get(k, 0)):
        if flt and flt not in key[0]:
            continue
        print(f"{key[0]} grid={key[1]} avg_us={dd[key]:.1f}")
        for c, v in sorted(merged[key].items()):
            print(f"    {c:28s} {v:16.1f}")


if __name__ == "__main__":
    main()
