#!/usr/bin/env python3
"""Wall time of the matching entry points (host arrays in, per-query results out, so H2D/D2H
copies and the scratch allocations of the call are included).

    python tools/match_bench.py [--n1 4000 --n2 4500 --dim 128 --iters 20]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sift_amd import api  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=4000)
    ap.add_argument("--n2", type=int, default=4500)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    d1 = rng.integers(0, 256, (a.n1, a.dim), dtype=np.uint8)
    d2 = rng.integers(0, 256, (a.n2, a.dim), dtype=np.uint8)
    for name, x, y in (("u8 (int8 MFMA)", d1, d2), ("f32 (ordered sum)", d1.astype(np.float32), d2.astype(np.float32))):
        api.match(x, y)
        t = time.perf_counter()
        for _ in range(a.iters):
            api.match(x, y)
        dt = (time.perf_counter() - t) / a.iters
        print(f"{name:20s} {a.n1} x {a.n2} x {a.dim}: {dt * 1e3:.3f} ms per call, "
              f"{2.0 * a.n1 * a.n2 * a.dim / dt / 1e12:.2f} Top/s of distance arithmetic")


if __name__ == "__main__":
    main()
