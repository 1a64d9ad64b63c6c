#!/usr/bin/env python3
"""Turns a rocprofv3 --kernel-trace --stats output directory into a short markdown table.

    python tools/prof_summary.py gpurun_out/prof_x [steps] > profiles/rNN_x.md
"""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    path = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"rocprofv3 --kernel-trace --stats, {path.split('/')[-1]}; total kernel time {tot / 1e6:.3f} ms"
          + (f" over {steps:g} steps = {tot / 1e6 / steps:.3f} ms/step" if steps else ""))
    print()
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")
        print(f"| {name} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")


if __name__ == "__main__":
    main()
