#!/usr/bin/env python3
"""Average launch duration per (kernel, grid) from a rocprofv3 kernel trace directory."""
import collections
import csv
import glob
import sys

rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if flt in n:
        g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["Grid_Size_Z"])
        agg[(n, g)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    v = sorted(v)
    print(f"{k[0]:24s} blocks={k[1]} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
