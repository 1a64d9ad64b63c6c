#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the reference itself.

Runs the unmodified reference sources, compiled in place by oracle/Makefile into
oracle/_ref/libsift_ref.so, on deterministic synthetic inputs (sift_amd/synth.py) and
stores only data: digests of the inputs and of every scale-space octave, stage sizes per
octave, scalar tables, and -- for the small cases -- the full final keypoint table.
Only runs where /root/reference is present; the outputs are committed.

    python tests/golden/make_golden.py                 # everything
    python tests/golden/make_golden.py NAME [NAME ...]   # only these cases; the index keeps the others
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refrun  # noqa: E402
from sift_amd import abi, synth  # noqa: E402

# name, kind, w, h, seed, parameter overrides, keep the keypoint table?
CASES = [
    ("white_64x48", "white", 64, 48, 1, {}, True),
    ("white_135x101", "white", 135, 101, 2, {}, True),
    ("white_200x150", "white", 200, 150, 3, {}, True),
    ("smooth_200x150", "smooth", 200, 150, 4, {}, True),
    ("white_333x257_spo4", "white", 333, 257, 5, {"n_spo": 4}, True),
    ("white_160x120_delta1", "white", 160, 120, 6, {"delta_min": 1.0}, True),
    ("smooth_800x640", "smooth", 800, 640, 12345, {}, True),
    # stand-ins for photographs (the Oxford images are not available offline): edges, corners and
    # texture at many scales, so that the edge filter and all octaves are exercised
    ("natural_800x640_a", "natural", 800, 640, 1, {}, True),
    ("natural_800x640_b", "natural", 800, 640, 2, {}, True),
    ("smooth_1920x1080", "smooth", 1920, 1080, 12345, {}, False),
    ("white_1920x1080", "white", 1920, 1080, 12345, {}, False),
    ("smooth_3840x2160", "smooth", 3840, 2160, 12345, {}, False),
    # octaves 960 and 480 wide: the widths at which single-blur launches pick 480-column tiles
    ("white_480x270", "white", 480, 270, 7, {}, True),
    ("natural_480x270", "natural", 480, 270, 3, {}, True),
    # full-size stand-in for a photograph, and the keypoint-dense case at 4K
    ("natural_1920x1080", "natural", 1920, 1080, 12345, {}, False),
    ("white_3840x2160", "white", 3840, 2160, 12345, {}, False),
]


def digest(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    lib = refrun.ref_lib()
    index = {}
    only = set(sys.argv[1:])
    if only:
        unknown = only - {c[0] for c in CASES}
        if unknown:
            sys.exit(f"unknown case(s): {sorted(unknown)}")
        with open(os.path.join(HERE, "index.json")) as f:
            index = json.load(f)
    for name, kind, w, h, seed, kw, table in CASES:
        if only and name not in only:
            continue
        img = synth.make(kind, w, h, seed)
        p = abi.SiftParameters.default().copy(**kw)
        rec = refrun.run_sift_anatomy(lib, img, p, keep_ss=True)
        n_oct = len(rec["octaves"])
        entry = {
            "kind": kind, "w": w, "h": h, "seed": seed, "params": kw,
            "input_sha1": digest(img),
            "n_oct": n_oct,
            "dims": [[o["w"], o["h"]] for o in rec["octaves"]],
            "sigmas": [[float(s).hex() for s in o["sigmas"]] for o in rec["octaves"]],
            "gss_sha1": [digest(g) for g in rec["gss"]],
            # per image of the stack: the batched pipeline may leave the last image of an octave,
            # which only the refinement reads near a candidate, incomplete
            "gss_image_sha1": [[digest(im) for im in g] for g in rec["gss"]],
            "dog_sha1": [digest(d) for d in rec["dog"]],
            "stage_counts_per_octave": {
                s: refrun.counts_per_octave(rec["stages"][s]["ints"], n_oct).tolist() for s in "ABCDEF"},
            "n_keys": int(len(rec["keys"]["ints"])),
            "keys_per_octave": refrun.counts_per_octave(rec["keys"]["ints"], n_oct).tolist(),
            "keys_sha1": {"ints": digest(rec["keys"]["ints"]),
                          "xyst": digest(rec["keys"]["flts"][:, :4]),
                          "descr_u8": digest(rec["keys"]["descr"].astype(np.uint8))},
        }
        if table:
            np.savez_compressed(os.path.join(HERE, name + ".npz"),
                                ints=rec["keys"]["ints"], flts=rec["keys"]["flts"],
                                descr=rec["keys"]["descr"].astype(np.uint8),
                                orihist=rec["keys"]["orihist"],
                                refined_ints=rec["stages"]["C"]["ints"],
                                refined_flts=rec["stages"]["C"]["flts"])
        index[name] = entry
        print(name, entry["stage_counts_per_octave"]["A"], entry["n_keys"], flush=True)
        del rec
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
