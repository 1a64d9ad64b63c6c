"""GPU parity: the HIP library, called through its C ABI, against the CPU oracle."""
import numpy as np
import pytest

import sift_amd
from sift_amd import abi, synth
from oracle import pyoracle, refrun

from common import bits, compare_records, same_bits

pytestmark = pytest.mark.gpu

CASES = [
    ("white", 64, 48, {}),
    ("white", 135, 101, {}),
    ("smooth", 200, 150, {}),
    ("white", 200, 150, {}),
    ("smooth", 800, 640, {}),
    ("white", 333, 257, {}),
    ("white", 160, 120, {"n_spo": 4}),
    ("white", 160, 120, {"n_spo": 2, "C_DoG": 0.02}),
    ("white", 128, 96, {"delta_min": 1.0, "sigma_min": 0.8, "sigma_in": 0.5}),
    ("smooth", 256, 192, {"sigma_min": 1.6, "n_oct": 3}),
]


def _params(kw):
    return abi.SiftParameters.default().copy(**kw)


@pytest.mark.parametrize("kind,w,h,kw", CASES)
def test_sift_anatomy_matches_oracle(hip, kind, w, h, kw):
    """Every stage of the reference-named sift_anatomy() against the oracle."""
    img = synth.make(kind, w, h, seed=4242 + w)
    p = _params(kw)
    got = refrun.run_sift_anatomy(hip, img, p)
    want = pyoracle.run(img, p)
    n_oct = len(want["octaves"])
    assert len(got["octaves"]) == n_oct
    for o in range(n_oct):
        assert got["octaves"][o]["w"] == want["octaves"][o]["w"]
        assert same_bits(got["octaves"][o]["sigmas"], want["octaves"][o]["sigmas"])
    stats = compare_records(got, want, n_oct)
    # gradients of the compatibility output
    for o in range(n_oct):
        for s in range(want["gss"][o].shape[0]):
            gi, gj = pyoracle.gradient(want["gss"][o][s])
            assert same_bits(got["gx"][o][s], gi) and same_bits(got["gy"][o][s], gj)
    print(kind, w, h, kw, stats)
    hip.sift_amd_release_thread_cache()


def test_batched_context_matches_single_images(hip):
    """A batch of different images gives, per image, what one-image runs give."""
    w, h, n = 320, 240, 5
    imgs = np.stack([synth.make("white" if b % 2 else "smooth", w, h, seed=100 + b) for b in range(n)])
    with sift_amd.Context(w, h, max_batch=n) as ctx:
        ctx.detect_host(imgs)
        counts = ctx.counts()
        batch = [ctx.keypoints(b) for b in range(n)]
        all_std, cnt2 = ctx.gather()
        assert np.array_equal(counts, cnt2) and len(all_std) == counts.sum()
        stage = [ctx.stage_counts(b) for b in range(n)]
    with sift_amd.Context(w, h, max_batch=n, keep_all_extrema=True) as ctx:
        ctx.detect_host(imgs)
        stage_all = [ctx.stage_counts(b) for b in range(n)]
        batch_all = [ctx.keypoints(b) for b in range(n)]
    off = 0
    for b in range(n):
        want = pyoracle.run(imgs[b], keep_ss=False)
        std, keys = batch[b]
        assert len(std) == len(want["keys"]["ints"])
        assert np.array_equal(np.stack([keys["o"], keys["s"], keys["i"], keys["j"]], 1), want["keys"]["ints"])
        assert np.abs(std["x"] - want["keys"]["flts"][:, 0]).max(initial=0) <= 1e-3
        assert np.abs(std["descriptor"].astype(np.int32) - want["keys"]["descr"].astype(np.int32)).max(initial=0) <= 1
        assert np.array_equal(all_std[off:off + counts[b]], std)
        off += counts[b]
        n_oct = len(want["octaves"])
        sizes, ext = stage_all[b]          # every discrete extremum listed: exact counts for A..F
        assert np.array_equal(ext, refrun.counts_per_octave(want["stages"]["A"]["ints"], n_oct))
        assert list(sizes) == [len(want["stages"][s]["ints"]) for s in "ABCDEF"]
        sizes, ext = stage[b]              # throughput configuration: A is not counted, B..F are
        assert sizes[0] == -1 and (ext == -1).all()
        assert list(sizes[1:]) == [len(want["stages"][s]["ints"]) for s in "BCDEF"]
        assert batch_all[b][0].tobytes() == std.tobytes() and batch_all[b][1].tobytes() == keys.tobytes()


def test_run_to_run_determinism(hip):
    img = synth.make("white", 400, 300, seed=7)
    with sift_amd.Context(400, 300) as ctx:
        ctx.detect_host(img)
        a = ctx.keypoints(0)[0].copy()
        ctx.detect_host(img)
        b = ctx.keypoints(0)[0].copy()
    assert a.tobytes() == b.tobytes()
