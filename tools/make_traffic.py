#!/usr/bin/env python3
"""Builds profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_passes.sh.

HBM bytes per launch of the dominant kernel (the octave-0 blur launches), corrected as the
MI355X counter notes prescribe: FETCH_SIZE (KiB) counts half of the bytes of coalesced streaming
reads on gfx950 (checked on this machine with tools/membench: dword and float4 reads of a known
size both report exactly 1/2), WRITE_SIZE (KiB) is exact.

    python tools/make_traffic.py gpurun_out/<tag> <threads of the octave-0 launch>[,<threads of the two-blur launch>] <images per launch> [out.json]
"""
import collections
import csv
import glob
import json
import sys


def per_launch(tag, name, counter, grids):
    """Mean counter value per launch over the octave-0 launches of the dominant group: k_blur_fused
    <R, false, ...> with grids[0] threads and, when the radius-5/7 pair runs as one launch,
    k_blur_pair with grids[1] threads (a launch is a launch: the mean is over both kinds)."""
    path = glob.glob(f"{tag}/{name}/**/*counter_collection.csv", recursive=True)[0]
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        fused = "k_blur_fused" in name and ", false," in name and int(r["Grid_Size"]) == grids[0]
        pair = "k_blur_pair" in name and len(grids) > 1 and int(r["Grid_Size"]) == grids[1]
        if (fused or pair) and r["Counter_Name"] == counter:
            per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = list(per_dispatch.values())
    return sum(vals) / len(vals), len(vals)


def main():
    tag, grids, images = sys.argv[1], [int(v) for v in sys.argv[2].split(",")], int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else "profiles/traffic.json"
    fetch_kib, n1 = per_launch(tag, "fetch", "FETCH_SIZE", grids)
    write_kib, n2 = per_launch(tag, "write", "WRITE_SIZE", grids)
    rd, wr = 2.0 * fetch_kib * 1024, write_kib * 1024
    rec = {"kernel": "k_blur_fused<R,false,...> and k_blur_pair, octave 0 launches (DoG-producing blurs)",
           "launches_sampled": [n1, n2],
           "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib,
           "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
           "bytes_per_launch": rd + wr, "images_per_launch": images,
           "correction": "read = 2 x FETCH_SIZE KiB (gfx950 counts 64 B per 128 B request), write = WRITE_SIZE KiB",
           "source": tag}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
