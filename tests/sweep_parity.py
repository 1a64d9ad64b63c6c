"""Randomised parity sweep (GPU): random image sizes (odd widths, widths not divisible by 4,
single-octave sizes), image kinds and parameter sets; the HIP library, through the
reference-named sift_anatomy() and through the batched throughput context, must equal the CPU
oracle bit for bit at every stage.

    python tests/sweep_parity.py [n_cases] [seed] [max_w max_h]

Not collected by pytest (tests/test_gpu_sweep.py runs a short fixed-seed sweep)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import sift_amd                                    # noqa: E402
from sift_amd import abi, api, synth               # noqa: E402
from oracle import pyoracle, refrun                # noqa: E402
from common import compare_records, same_bits      # noqa: E402

PARAM_SETS = [
    {}, {}, {},
    {"n_spo": 2}, {"n_spo": 4}, {"n_spo": 5, "C_DoG": 0.01},
    {"sigma_min": 1.0}, {"sigma_min": 1.6, "n_oct": 3},
    {"delta_min": 1.0, "sigma_min": 0.8, "sigma_in": 0.5},
    {"C_edge": 5.0}, {"lambda_ori": 1.0, "t": 0.9}, {"lambda_descr": 4.0},
    {"n_hist": 3, "n_ori": 6}, {"n_bins": 24},
]


MAX_W, MAX_H = 420, 320


def random_case(rng):
    kind = ["white", "smooth", "white", "mix"][rng.integers(4)]
    w = int(rng.integers(17, MAX_W))
    h = int(rng.integers(13, MAX_H))
    if rng.random() < 0.25:
        w = int(rng.integers(MAX_W - 20, MAX_W + 280))
    kw = PARAM_SETS[rng.integers(len(PARAM_SETS))]
    return kind, w, h, kw, int(rng.integers(1 << 30))


def make_image(kind, w, h, seed):
    if kind == "mix":      # smooth background with a noisy patch: marked and unmarked regions side by side
        img = synth.make("smooth", w, h, seed)
        noise = synth.make("white", w, h, seed + 1)
        y0, x0 = h // 3, w // 4
        img[y0:y0 + h // 3, x0:x0 + w // 2] = noise[y0:y0 + h // 3, x0:x0 + w // 2]
        return np.ascontiguousarray(img)
    return synth.make(kind, w, h, seed)


def reference_reads_out_of_bounds(octaves):
    """lib_discrete.c:243-249 mirrors an index with (i + 2l) % 2l, which is negative (an
    out-of-bounds read) once a blur radius exceeds twice the image dimension; such cases have no
    defined reference result (the HIP library extends periodically there)."""
    for o in octaves:
        sig = np.asarray(o["sigmas"], np.float64)
        for s in range(1, len(sig)):
            radius = int(np.ceil(4 * np.sqrt(max(sig[s] ** 2 - sig[s - 1] ** 2, 0.0)) / o["delta"]))
            if radius > 2 * min(o["w"], o["h"]):
                return True
    return False


class Undefined(Exception):
    pass


def run_case(lib, kind, w, h, kw, seed):
    img = make_image(kind, w, h, seed)
    p = abi.SiftParameters.default().copy(**kw)
    want = pyoracle.run(img, p)
    n_oct = len(want["octaves"])
    if reference_reads_out_of_bounds(want["octaves"]):
        raise Undefined("a blur radius exceeds twice an octave's size")
    got = refrun.run_sift_anatomy(lib, img, p)
    stats = compare_records(got, want, n_oct)
    assert stats.get("descr_exact", 1.0) == 1.0 and stats.get("theta_exact", 1.0) == 1.0, stats
    # throughput context (marks, no stored DoG), image repeated in a batch of 3
    with sift_amd.Context(w, h, max_batch=3, params=p) as ctx:
        ctx.detect_host(np.stack([img, img[::-1].copy(), img]))
        for b in (0, 2):
            std, keys = ctx.keypoints(b)
            orih, descr, _ = ctx.extras(b)
            assert len(std) == len(want["keys"]["ints"]), (len(std), len(want["keys"]["ints"]))
            if len(std):
                xyst = np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1)
                assert same_bits(xyst, want["keys"]["flts"][:, :4])
                assert np.array_equal(descr, want["keys"]["descr"])
                assert same_bits(orih, want["keys"]["orihist"])
    lib.sift_amd_release_thread_cache()
    return stats["n_keys"]


def sweep(n_cases, seed, verbose=True):
    lib = api.lib()
    rng = np.random.default_rng(seed)
    total = 0
    t0 = time.time()
    for c in range(n_cases):
        case = random_case(rng)
        try:
            nk = run_case(lib, *case)
        except (sift_amd.api.SiftAmdError, Undefined) as e:      # e.g. image too small for the parameter set
            if verbose:
                print(c, case, "refused:", str(e)[:60], flush=True)
            continue
        total += nk
        if verbose:
            print(c, case, "keys", nk, f"{time.time() - t0:.0f}s", flush=True)
    return total


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 4:
        MAX_W, MAX_H = int(sys.argv[3]), int(sys.argv[4])
    print("keypoints compared:", sweep(n, s))
