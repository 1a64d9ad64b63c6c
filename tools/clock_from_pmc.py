#!/usr/bin/env python3
"""Shader clock during each blur launch, from a rocprofv3 PMC pass with GRBM_GUI_ACTIVE:
busy cycles (summed over the 8 XCDs by the tool) / 8 / launch duration.
    python3 tools/clock_from_pmc.py <dir with *_counter_collection.csv> [min grid size]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    min_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or int(r["Grid_Size"]) < min_grid:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[name].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    print("| kernel | launches | duration us | busy cycles per XCD | clock GHz |")
    print("|---|---:|---:|---:|---:|")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(b for _, b in kv[1])):
        c = sum(a for a, _ in v) / len(v) / 8
        ns = sum(b for _, b in v) / len(v)
        print(f"| {k} | {len(v)} | {ns / 1e3:.0f} | {c / 1e6:.2f} M | {c / ns:.2f} |")


if __name__ == "__main__":
    main()
