#!/usr/bin/env python3
"""Average duration of the dominant kernel's launches in the roofline leg of bench.py, taken from
a rocprofv3 kernel trace of that command.

bench.py times the octave-0 blur launches with hipEvents during extra steps that run alone after
the timed region (the timed region keeps several batches in flight, which stretches individual
kernel durations).  This picks the same launches from the trace: the last `n` launches of
k_blur_fused<R, false, ...> with the octave-0 grid.

    python tools/dominant_from_trace.py <rocprof dir> <blocks of the octave-0 launch> [n=50] [blocks of the two-blur launch]
"""
import csv
import glob
import sys

d, blocks = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rows = list(csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])))
blocks2 = int(sys.argv[4]) if len(sys.argv) > 4 else -1     # blocks of the two-blur launch (k_blur_pair), if used


def wanted(r):
    nb = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    name = r["Kernel_Name"]
    return ("k_blur_fused<" in name and ", false," in name and nb == blocks) or ("k_blur_pair<" in name and nb == blocks2)


sel = [r for r in rows if wanted(r)]
sel.sort(key=lambda r: int(r["Start_Timestamp"]))
last = sel[-n:]
per_r = {}
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per_r.setdefault(name, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = [v for vs in per_r.values() for v in vs]
print(f"last {len(last)} octave-0 launches of k_blur_fused<R,false,...> ({blocks} blocks) / k_blur_pair ({blocks2} blocks) in the trace:")
for k in sorted(per_r, key=lambda k: (int(k.split("<")[1].split(",")[0]), k)):
    v = per_r[k]
    print(f"  {k:34s} n={len(v):3d}  mean {sum(v) / len(v):8.1f} us")
print(f"  all                                n={len(tot):3d}  mean {sum(tot) / len(tot):8.1f} us")
