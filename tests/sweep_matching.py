"""Randomised sweep of the descriptor matcher (GPU) against the CPU oracle: random set sizes,
descriptor lengths, thresholds, both decision rules, 8-bit and float descriptors, few distinct
values (many exact distance ties) as well as full-range ones.

    python tests/sweep_matching.py [n_cases] [seed]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from sift_amd import api                 # noqa: E402
from oracle import pyoracle              # noqa: E402


FIELDS = ("iA", "iB", "dA", "dB", "pass")


def same(got, want):
    """First field that differs (floats by bit pattern), or None."""
    for f in FIELDS:
        g, w = got[f], want[f]
        if g.dtype.kind == "f":
            g, w = g.view(np.uint32), w.view(np.uint32)
        if not np.array_equal(g, w):
            return f
    return None


def random_case(rng):
    n1 = int(rng.integers(1, 400))
    n2 = int(rng.integers(2, 1500))
    dim = int(rng.choice([128, 128, 128, 72, 32, 5, 200, 256, 17]))
    levels = int(rng.choice([2, 3, 8, 256]))
    flag = int(rng.integers(2))
    thresh = float(rng.uniform(0.3, 1.0)) if flag else float(rng.uniform(10, 2500))
    kind = "u8" if rng.random() < 0.7 else "f32"
    return n1, n2, dim, levels, flag, thresh, kind


def run_case(rng, n1, n2, dim, levels, flag, thresh, kind):
    scale = 255 // (levels - 1)
    a = (rng.integers(0, levels, (n1, dim)) * scale).astype(np.uint8)
    b = (rng.integers(0, levels, (n2, dim)) * scale).astype(np.uint8)
    k = min(n1, n2) // 2                       # near and exact duplicates across the two sets
    if k:
        b[:k] = a[:k]
        b[: k // 2, 0] ^= 1
    if kind == "f32":
        a = a.astype(np.float32) + rng.random((n1, dim), np.float32) * (levels == 256)
        b = b.astype(np.float32) + rng.random((n2, dim), np.float32) * (levels == 256)
        want = pyoracle.matching(a, b, thresh, flag)
    else:
        want = pyoracle.matching(a.astype(np.float32), b.astype(np.float32), thresh, flag)
    got = api.match(a, b, thresh, flag)
    bad = same(got, want)
    assert bad is None, (bad, n1, n2, dim, levels, flag, thresh, kind)
    return int(want["pass"].sum())


def sweep(n_cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    total = 0
    for c in range(n_cases):
        case = random_case(rng)
        total += run_case(rng, *case)
        if verbose and c % 50 == 0:
            print(c, case, "matches so far", total, flush=True)
    return total


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    print("matches compared:", sweep(n, s))
