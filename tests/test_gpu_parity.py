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
        n_a = len(want["stages"]["A"]["ints"])  # -1 where the scan skipped unmarked regions, else exact
        assert sizes[0] in (-1, n_a) and all(e in (-1, x) for e, x in zip(ext, stage_all[b][1]))
        assert list(sizes[1:]) == [len(want["stages"][s]["ints"]) for s in "BCDEF"]
        assert batch_all[b][0].tobytes() == std.tobytes() and batch_all[b][1].tobytes() == keys.tobytes()


def test_run_to_run_determinism(hip):
    """Four runs of one context: plain launches, the run that builds the hipGraph of the launch
    sequence, and two replays of it give the same bytes."""
    img = synth.make("white", 400, 300, seed=7)
    with sift_amd.Context(400, 300) as ctx:
        runs = []
        for _ in range(4):
            ctx.detect_host(img)
            runs.append(ctx.keypoints(0)[0].tobytes())
    assert len(runs[0]) > 144 * 100 and all(r == runs[0] for r in runs)


def test_graph_replay_between_foreign_copies(hip):
    """A context's cached hipGraphs replayed after other users of the process have issued small
    fills and host<->device copies on the null stream (what a torch tensor created on the GPU or an
    .item() does), two input buffers in turn: same bytes every time.  (A memset node inside a
    captured graph replayed with the operands of such a later copy; the pipeline clears its buffers
    with a kernel of its own since.)"""
    from test_zz_no_alloc import Hip
    w, h, B = 400, 300, 4
    imgs = np.stack([synth.make("white", w, h, seed=40 + b) for b in range(B)])
    hipm = Hip()
    d_a, d_b = hipm.upload(imgs), hipm.upload(imgs)
    small = hipm.alloc(4096)
    host = np.arange(64, dtype=np.float64)
    with sift_amd.Context(w, h, max_batch=B) as ctx:
        want = None
        for k in range(14):
            ctx.detect_device(d_a if k % 2 == 0 else d_b, B, None)
            got = ctx.gather()[0].tobytes()
            want = want or got
            assert got == want, f"run {k}"
            for r in range(k % 5 + 1):                       # foreign null-stream traffic between the runs
                assert hipm.rt.hipMemcpy(small, host.ctypes.data, 8 * (r + 1), 1) == 0
                assert hipm.rt.hipMemset(small, r, 64) == 0
                assert hipm.rt.hipMemcpy(host.ctypes.data, small, 8 * (r + 1), 2) == 0
    assert len(want) > 144 * 400
    hipm.free_all()


EDGE_IMAGES = {
    "constant": lambda w, h: np.full((h, w), 0.37, np.float32),
    "zeros": lambda w, h: np.zeros((h, w), np.float32),
    "step": lambda w, h: np.tile((np.arange(w) >= w // 2).astype(np.float32), (h, 1)),
    "impulse": lambda w, h: np.pad(np.ones((1, 1), np.float32), ((h // 2, h - h // 2 - 1), (w // 2, w - w // 2 - 1))),
    "checker": lambda w, h: ((np.add.outer(np.arange(h) // 5, np.arange(w) // 7) & 1) * 0.8).astype(np.float32),
}


@pytest.mark.parametrize("name,w,h", [("constant", 96, 64), ("zeros", 64, 64), ("step", 100, 60),
                                      ("impulse", 65, 49), ("checker", 150, 90), ("checker", 31, 203),
                                      ("checker", 24, 24), ("impulse", 17, 13)])
def test_degenerate_and_tiny_images(hip, name, w, h):
    """Flat, binary and very small inputs (down to a single octave): stage sizes and final
    records equal the oracle's, including the empty results."""
    img = EDGE_IMAGES[name](w, h)
    want = pyoracle.run(img, keep_ss=False)
    with sift_amd.Context(w, h, keep_all_extrema=True) as ctx:
        ctx.detect_host(img)
        sizes, ext = ctx.stage_counts(0)
        std, keys = ctx.keypoints(0)
        _, descr, _ = ctx.extras(0)
    n_oct = len(want["octaves"])
    assert list(sizes) == [len(want["stages"][s]["ints"]) for s in "ABCDEF"]
    assert np.array_equal(ext, refrun.counts_per_octave(want["stages"]["A"]["ints"], n_oct))
    assert len(std) == len(want["keys"]["ints"])
    if len(std):
        got = np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1)
        assert same_bits(got, want["keys"]["flts"][:, :4])
        assert np.array_equal(descr, want["keys"]["descr"])
    with sift_amd.Context(w, h) as ctx:           # throughput configuration: same records
        ctx.detect_host(img)
        std2, keys2 = ctx.keypoints(0)
    assert std2.tobytes() == std.tobytes() and keys2.tobytes() == keys.tobytes()


def test_partial_batches_and_context_reuse(hip):
    """A context sized for 6 images run with 1, 6, then 3 images gives per-image results that
    do not depend on the batch they were part of."""
    w, h = 160, 120
    imgs = np.stack([synth.make("white", w, h, seed=500 + b) for b in range(6)])
    with sift_amd.Context(w, h, max_batch=6) as ctx:
        ref = []
        for b in range(6):
            ctx.detect_host(imgs[b])
            ref.append(ctx.keypoints(0)[0].copy())
        ctx.detect_host(imgs)
        for b in range(6):
            assert ctx.keypoints(b)[0].tobytes() == ref[b].tobytes()
        ctx.detect_host(imgs[[4, 1, 5]])
        assert [ctx.keypoints(k)[0].tobytes() for k in range(3)] == [ref[b].tobytes() for b in (4, 1, 5)]
        assert ctx.L.sift_amd_detect_host(ctx.ptr, imgs.ctypes.data_as(abi.c_float_p), 7) != 0   # over capacity


def test_non_default_descriptor_shapes_and_matching_between_images(hip):
    """n_hist x n_hist x n_ori other than 4x4x8 (bins of a lane straddling cells, float
    descriptor path of the matcher)."""
    w, h = 200, 150
    a = synth.make("white", w, h, seed=71)
    b = np.roll(a, (2, 3), axis=(0, 1))
    for kw in ({"n_hist": 3, "n_ori": 8}, {"n_hist": 4, "n_ori": 5}, {"n_hist": 2, "n_ori": 12},
               {"n_hist": 1, "n_ori": 72}, {"n_hist": 2, "n_ori": 64}):
        p = _params(kw)
        nd = p.n_hist * p.n_hist * p.n_ori
        with sift_amd.Context(w, h, max_batch=2, params=p) as ctx:
            ctx.detect_host(np.stack([a, b]))
            d = [ctx.extras(k)[1] for k in range(2)]
            m = ctx.match(0, 1, 0.7, 1)
        for img, got in zip((a, b), d):
            want = pyoracle.run(img, p, keep_ss=False)["keys"]["descr"]
            assert got.shape == want.shape == (len(want), nd) and np.array_equal(got, want), kw
        wm = pyoracle.matching(d[0], d[1], 0.7, 1)
        for f in ("iA", "iB", "pass"):
            assert np.array_equal(m[f], wm[f]), (kw, f)
        assert np.array_equal(m["dA"].view(np.uint32), wm["dA"].view(np.uint32))
        assert wm["pass"].sum() > 5


@pytest.mark.parametrize("kw,needle", [
    ({"sigma_min": 0.3}, "sigma_min"), ({"delta_min": 1.5}, "delta_min"), ({"n_spo": 10}, "n_spo"),
    ({"n_bins": 70}, "n_bins"), ({"n_hist": 9}, "n_hist"), ({"n_ori": 1}, "n_ori"),
    ({"n_hist": 8, "n_ori": 8}, "n_hist")])
def test_unsupported_parameters_are_refused_with_a_message(hip, kw, needle):
    """The batched interface reports bad parameters through its error channel (the reference
    asserts); nothing is launched."""
    with pytest.raises(sift_amd.api.SiftAmdError) as e:
        sift_amd.Context(128, 96, params=_params(kw))
    assert needle in str(e.value)


def test_image_too_small_for_one_octave_is_refused(hip):
    with pytest.raises(sift_amd.api.SiftAmdError):
        sift_amd.Context(4, 4)


def test_image_beyond_32_bit_offsets_is_refused(hip):
    """One scale-space image must stay below 2 GiB (the blur addresses it with 32-bit byte
    offsets); the refusal comes before any allocation."""
    with pytest.raises(sift_amd.api.SiftAmdError) as e:
        sift_amd.Context(20000, 14000)
    assert "too large" in str(e.value)


def test_list_overflow_is_reported_not_silent(hip, monkeypatch):
    """With the candidate/keypoint lists forced far too small, every kernel stays inside its
    buffers and the run reports the overflow instead of returning a truncated result."""
    w, h = 200, 150
    img = synth.make("white", w, h, seed=31)
    with sift_amd.Context(w, h) as ctx:
        ctx.detect_host(img)
        sizes, _ = ctx.stage_counts(0)
    n_cand, n_final = int(sizes[1]), int(sizes[5])
    assert n_cand > 64 and n_final > (n_cand + 8) // 4
    # candidate list too small; then candidates fit but the survivor/key lists (a quarter) do not
    for cap in (64, n_cand + 8):
        monkeypatch.setenv("SIFT_AMD_CAP", str(cap))
        with sift_amd.Context(w, h, max_batch=2) as ctx:
            ctx.detect_host(np.stack([img, img]))
            with pytest.raises(sift_amd.api.SiftAmdError) as e:
                ctx.counts()
            assert "overflow" in str(e.value)
            # every later reader of this run's results gets the same error, not a truncated list
            for read in (ctx.counts, ctx.keypoints, ctx.gather, ctx.candidates, ctx.extras, ctx.stage_counts):
                with pytest.raises(sift_amd.api.SiftAmdError) as e:
                    read()
                assert "overflow" in str(e.value), read
            buf = np.zeros(16, abi.KEYPOINT_STD_DTYPE)
            assert ctx.L.sift_amd_fetch_keypoints(ctx.ptr, 0, buf.ctypes.data, None, 16) < 0
            # the next run of the same context starts clean (it overflows again, for the same reason)
            ctx.detect_host(img)
            with pytest.raises(sift_amd.api.SiftAmdError):
                ctx.counts()
    monkeypatch.delenv("SIFT_AMD_CAP")
    with sift_amd.Context(w, h) as ctx:      # and the library is fine afterwards
        ctx.detect_host(img)
        assert ctx.counts()[0] == len(pyoracle.run(img, keep_ss=False)["keys"]["ints"])


@pytest.mark.parametrize("name,value", [("nan", np.nan), ("inf", np.inf), ("huge", 1e30), ("negative", -5.0),
                                        ("denormal-range", None)])
def test_unusual_sample_values(hip, name, value):
    """Out-of-range, non-finite and denormal-range samples: the lists from the first contrast
    test on and the final records equal the oracle's.  (The list of all discrete extrema is not
    compared for NaN/Inf: the reference lists every NaN sample of the DoG stack there.)"""
    w, h = 96, 80
    img = synth.make("white", w, h, seed=5)
    if value is None:
        img = (img * np.float32(1e-38)).astype(np.float32)
    else:
        img[40, 48] = value
    want = pyoracle.run(img, keep_ss=False)
    with sift_amd.Context(w, h, keep_all_extrema=True) as ctx:
        ctx.detect_host(img)
        sizes, _ = ctx.stage_counts(0)
        std, _ = ctx.keypoints(0)
        _, descr, _ = ctx.extras(0)
    first = 1 if name in ("nan", "inf") else 0
    assert list(sizes[first:]) == [len(want["stages"][s]["ints"]) for s in "ABCDEF"[first:]]
    assert len(std) == len(want["keys"]["ints"])
    if len(std):
        got = np.stack([std["x"], std["y"], std["scale"], std["orientation"]], 1)
        assert same_bits(got, want["keys"]["flts"][:, :4])
        assert np.array_equal(descr, want["keys"]["descr"])


def test_two_blur_launches_in_a_batch_equal_single_image_runs(hip):
    """A batch large enough for the two-blur launch (images s and s+1 of an octave in one kernel,
    chosen by the library from the number of blocks) gives the records of single-image runs, which
    take one launch per blur."""
    w, h, nb = 480, 270, 64
    imgs = np.stack([synth.make(["white", "natural", "smooth"][b % 3], w, h, 4000 + b) for b in range(nb)])
    with sift_amd.Context(w, h, max_batch=1) as one:
        single = []
        for b in (0, 1, 2, 3, 37, 63):
            one.detect_host(imgs[b])
            std, keys = one.keypoints(0)
            single.append((b, std.tobytes(), keys.tobytes(),
                           # (the last image of a stack is computed only where it is read: not compared)
                           [one.scalespace(0, 0, o)[:-1].tobytes() for o in range(one.n_oct)]))
    with sift_amd.Context(w, h, max_batch=nb) as ctx:
        ctx.detect_host(imgs)
        for b, std_b, keys_b, gss in single:
            std, keys = ctx.keypoints(b)
            assert std.tobytes() == std_b and keys.tobytes() == keys_b, f"image {b}"
            for o in range(ctx.n_oct):
                assert ctx.scalespace(b, 0, o)[:-1].tobytes() == gss[o], f"image {b} octave {o}"
    assert len(single[0][1]) > 144 * 50


def test_scale_space_only_and_whole_pipeline_runs_alternate_on_one_context(hip):
    """A throughput context computes the last image of every stack only where it is read when it runs
    the whole pipeline, and everywhere when only the scale-space is asked for: alternating the two
    kinds of run (each has its own cached hipGraph) gives the complete stack after one and the
    reference's records after the other, every time."""
    from test_zz_no_alloc import Hip
    from test_oracle_golden import INDEX, digest
    name = "natural_480x270"
    e = INDEX[name]
    img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
    want = pyoracle.run(img, keep_ss=False)
    hipm = Hip()
    d_img = hipm.upload(np.stack([img, img, img]))
    with sift_amd.Context(e["w"], e["h"], max_batch=3) as ctx:
        for rep in range(4):
            ctx.scalespace_device(d_img, 3, None)
            for o in (0, 1, ctx.n_oct - 1):
                assert [digest(im) for im in ctx.scalespace(2, 0, o)] == e["gss_image_sha1"][o], f"run {rep} octave {o}"
            ctx.detect_device(d_img, 3, None)
            std, _ = ctx.keypoints(1)
            assert len(std) == e["n_keys"]
            assert np.array_equal(std["descriptor"], want["keys"]["descr"].astype(np.uint8))
    hipm.free_all()


def test_stored_dog_stacks_of_a_batch(hip):
    """Contexts that keep the DoG stacks (and list every extremum) compute every image everywhere, also in a
    batch small enough for the two-chain schedule: all images of the batch carry the fixture's stacks."""
    from test_oracle_golden import INDEX, digest
    name = "white_480x270"
    e = INDEX[name]
    img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
    with sift_amd.Context(e["w"], e["h"], 4, keep_all_extrema=True, keep_scalespace=True) as ctx:
        for _ in range(3):
            ctx.detect_host(np.stack([img] * 4))
            for b in (0, 3):
                for o in range(ctx.n_oct):
                    assert digest(ctx.scalespace(b, 0, o)) == e["gss_sha1"][o]
                    assert digest(ctx.scalespace(b, 1, o)) == e["dog_sha1"][o]
                sizes, ext = ctx.stage_counts(b)
                assert ext.tolist() == e["stage_counts_per_octave"]["A"]
                assert len(ctx.keypoints(b)[0]) == e["n_keys"]


def test_eight_bit_input_through_both_schedules(hip, monkeypatch):
    """8-bit images converted on the GPU, then the pipeline as one chain and as two chains: the records of
    the float path on the same values."""
    rng = np.random.default_rng(5)
    w, h = 320, 200
    u8 = rng.integers(0, 256, (2, h, w), dtype=np.uint8)
    f32 = (u8.astype(np.float32).astype(np.float64) / 256.0).astype(np.float32)
    with sift_amd.Context(w, h, max_batch=2) as ref:
        ref.detect_host(f32)
        want = [ref.keypoints(b)[0].tobytes() for b in range(2)]
    assert len(want[0]) > 144 * 100
    for fork in ("0", "1"):
        monkeypatch.setenv("SIFT_AMD_FORK", fork)
        with sift_amd.Context(w, h, max_batch=2) as ctx:
            for _ in range(3):
                ctx.detect_host_u8(u8)
                assert [ctx.keypoints(b)[0].tobytes() for b in range(2)] == want
