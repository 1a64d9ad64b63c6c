"""GPU: the launches the benchmark's number is measured on, against the golden fixtures.

The blur launches depend on the batch: large batches of wide octaves take the two-blur launches
with 1080-row strips and the single-blur launches that fetch rows two iterations ahead in strips
of ~544 rows, which a batch of one never reaches.  These tests run the full-size golden images
inside batches of the benchmark's size (320 images of 1920x1080, 64 of 3840x2160) and compare the
records and the Gaussian stacks of several slots with the fixtures the compiled reference produced;
the launch-route counters of the library show that the routes meant were the ones taken."""
import numpy as np
import pytest

import sift_amd
from sift_amd import api, synth

from test_oracle_golden import INDEX, digest

pytestmark = pytest.mark.gpu


def golden_batch(names, batch):
    imgs = []
    for n in names:
        e = INDEX[n]
        img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
        assert digest(img) == e["input_sha1"]
        imgs.append(img)
    order = [names[b % len(names)] for b in range(batch)]
    return np.stack([imgs[b % len(names)] for b in range(batch)]), order


def check_records(recs, counts, order, slots):
    assert counts.tolist() == [INDEX[n]["n_keys"] for n in order], "keypoints per image"
    starts = np.concatenate([[0], np.cumsum(counts)])
    for b in slots:
        e = INDEX[order[b]]
        std = recs[starts[b]:starts[b + 1]]
        xyst = np.ascontiguousarray(np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1))
        assert digest(xyst) == e["keys_sha1"]["xyst"], f"slot {b} ({order[b]}): x, y, sigma, theta bits"
        assert digest(np.ascontiguousarray(std["descriptor"])) == e["keys_sha1"]["descr_u8"], \
            f"slot {b} ({order[b]}): descriptors"


def check_stacks(ctx, order, slots, sparse_last):
    """Every image of the Gaussian stack of every octave, bit for bit.  The last image of an
    octave is read only by the refinement next to a candidate; a context that computes it only
    there (see sift_amd_launch_stats, SIFT_AMD_LS_SPARSE) is not compared on it."""
    for b in slots:
        e = INDEX[order[b]]
        for o in range(ctx.n_oct):
            stack = ctx.scalespace(b, 0, o)
            n_img = len(stack) - (1 if sparse_last else 0)
            got = [digest(stack[s]) for s in range(n_img)]
            assert got == e["gss_image_sha1"][o][:n_img], f"slot {b} ({order[b]}), octave {o}"


def test_bench_batch_1080p(hip):
    """320 images of 1920x1080: the benchmark's batch and launch geometry (the launch shapes are the
    same from 128 images on: every launch has more blocks than the thresholds that shorten strips)."""
    names = ["smooth_1920x1080", "white_1920x1080", "natural_1920x1080"]
    B = 320
    batch, order = golden_batch(names, B)
    api.launch_stats(reset=True)
    with sift_amd.Context(1920, 1080, B) as ctx:
        ctx.detect_host(batch)
        recs, counts = ctx.gather()
        st = api.launch_stats(reset=True)
        # octave 0: seed + s1 and s2 + s3 as two-blur launches, octaves 1-2 one pair each, 1080-row strips;
        # the radius-10/13 launches of the wide octaves fetch rows two iterations ahead
        assert st["pair_seed"] == 1 and st["pair"] >= 2 and st["pair_strip_max"] == 1080, st
        assert st["wide_ahead2"] >= 4 and st["wide_ahead2_strip_max"] >= 544, st
        check_records(recs, counts, order, [0, 1, 2, 127, 128, 129, 160, 317, 318, 319])
        check_stacks(ctx, order, [0, 1, 2, 319], sparse_last=st["sparse"] > 0)
        for _ in range(2):                         # hipGraph build, replay
            ctx.detect_host(batch)
            recs, counts = ctx.gather()
            check_records(recs, counts, order, [0, 1, 2, 319])
        # 128 images (the batch of earlier measurements) through the same context
        ctx.detect_host(batch[:128])
        recs, counts = ctx.gather()
        check_records(recs, counts, order[:128], [0, 1, 2, 63, 64, 65, 125, 126, 127])
        # a partial batch through the same context
        ctx.detect_host(batch[:37])
        recs, counts = ctx.gather()
        check_records(recs, counts, order[:37], [0, 1, 2, 36])


def test_bench_batch_4k(hip):
    """64 images of 3840x2160, the benchmark's 4K leg."""
    names = ["smooth_3840x2160", "white_3840x2160"]
    B = 64
    batch, order = golden_batch(names, B)
    api.launch_stats(reset=True)
    with sift_amd.Context(3840, 2160, B) as ctx:
        ctx.detect_host(batch)
        recs, counts = ctx.gather()
        st = api.launch_stats(reset=True)
        assert st["pair_seed"] == 1 and st["pair"] >= 2 and st["wide_ahead2"] >= 4, st
        check_records(recs, counts, order, [0, 1, 30, 31, 62, 63])
        check_stacks(ctx, order, [0, 63], sparse_last=st["sparse"] > 0)
        ctx.detect_host(batch)
        ctx.detect_host(batch)
        recs, counts = ctx.gather()
        check_records(recs, counts, order, [0, 1, 63])


@pytest.mark.parametrize("fork", ["0", "1"])
def test_bench_batch_both_schedules(hip, monkeypatch, fork):
    """The benchmark's batch as one chain of launches and as two interleaved chains (the default for
    small batches only): same records."""
    monkeypatch.setenv("SIFT_AMD_FORK", fork)
    names = ["natural_1920x1080", "smooth_1920x1080", "white_1920x1080"]
    batch, order = golden_batch(names, 48)
    with sift_amd.Context(1920, 1080, 48) as ctx:
        for _ in range(3):
            ctx.detect_host(batch)
            recs, counts = ctx.gather()
            check_records(recs, counts, order, [0, 1, 2, 45, 46, 47])


@pytest.mark.parametrize("name", ["white_480x270", "natural_480x270"])
def test_wide_tiles_default_strips(hip, name):
    """Octaves 960 and 480 wide pick the 480-column tiles; a batch of 256 gives them their default
    strips (two iterations ahead).  Full record tables against the fixture."""
    import os
    from test_oracle_golden import GOLD
    e = INDEX[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    batch, order = golden_batch([name], 256)
    api.launch_stats(reset=True)
    with sift_amd.Context(e["w"], e["h"], 256) as ctx:
        ctx.detect_host(batch)
        st = api.launch_stats(reset=True)
        assert st["wide_ahead2"] >= 2, st
        for b in (0, 100, 255):
            std, keys = ctx.keypoints(b)
            orih, _, _ = ctx.extras(b)
            xyst = np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1)
            assert np.array_equal(xyst.view(np.uint32), g["flts"][:, :4].view(np.uint32))
            assert np.array_equal(std["descriptor"], g["descr"])
            assert np.array_equal(orih.view(np.uint32), g["orihist"].view(np.uint32))
        check_stacks(ctx, order, [0, 255], sparse_last=st["sparse"] > 0)
