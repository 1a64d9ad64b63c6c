"""Descriptor matching (lib_matching.c:92-135) and 8-bit ingest (io_png.c:480-520 +
sift_cli.c:280-281): oracle vs compiled reference on the CPU, HIP vs oracle on the GPU.
Everything is compared bit-exactly (indices, float distances, pass flags)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle, refrun
from sift_amd import abi, synth

FIELDS = ("iA", "iB", "dA", "dB", "pass")


def descriptor_sets(n1, n2, dim, seed, kind="u8"):
    """Two sets where most rows of the first have a noisy twin in the second."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (max(n1, n2), dim))
    a = base[:n1].copy()
    b = base[rng.permutation(max(n1, n2))[:n2]].copy()
    b = np.clip(b + rng.integers(-6, 7, b.shape), 0, 255)
    if kind == "u8":
        return a.astype(np.uint8), b.astype(np.uint8)
    a = a + rng.uniform(-0.5, 0.5, a.shape)
    b = b + rng.uniform(-0.5, 0.5, b.shape)
    return a.astype(np.float32), b.astype(np.float32)


def with_ties(a, b):
    """Plant the tie cases of lib_util.c:108-134: targets 0 and 1 identical, exact copies
    of queries, duplicated targets further down."""
    b = b.copy()
    a = a.copy()
    b[1] = b[0]
    if len(a) > 3:
        a[2] = b[0]              # nearest two are targets 0 and 1 at distance 0
    if len(b) > 9:
        b[9] = b[5]
        if len(a) > 4:
            a[4] = b[5]
        far = 255 - b[0]         # targets 0/1 tie but are not the nearest
        if len(a) > 5:
            a[5] = far
    return a, b


def reference_matching(lib, d1, d2, thresh, flag, n_hist=4, n_ori=8):
    """Run `matching` of a reference-API library; returns (query idx, iA, iB) of the pairs."""
    d1 = np.asarray(d1, np.float32)
    d2 = np.asarray(d2, np.float32)
    k1 = abi.keypoints_from_numpy(lib, d1, n_hist, n_ori)
    k2 = abi.keypoints_from_numpy(lib, d2, n_hist, n_ori)
    outs = [lib.sift_malloc_keypoints() for _ in range(3)]
    lib.matching(k1, k2, outs[0], outs[1], outs[2], thresh, flag)
    r = [abi.keypoints_to_numpy(o) for o in outs]
    res = tuple(x["flts"][:, 0].astype(np.int64) for x in r)
    descr_ok = all(np.array_equal(x["descr"], src[i]) for x, src, i in
                   zip(r, (d1, d2, d2), res))
    for lst in [k1, k2] + outs:
        lib.sift_free_keypoints(lst)
    return res, descr_ok


def assert_same(got, want, what=""):
    for f in FIELDS:
        g, w = got[f], want[f]
        if g.dtype.kind == "f":      # compare bit patterns (NaN-safe, -0 vs +0 strict)
            g, w = g.view(np.uint32), w.view(np.uint32)
        bad = np.nonzero(g != w)[0]
        assert bad.size == 0, f"{what} {f}: {bad.size} differ, first {bad[:5]} got {got[f][bad[:5]]} want {want[f][bad[:5]]}"


# ------------------------------------------------------------------ CPU: oracle pinned
@pytest.mark.skipif(not refrun.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("n1,n2,dim,flag,thresh,kind", [
    (60, 80, 128, 1, 0.6, "u8"), (60, 80, 128, 0, 250.0, "u8"), (33, 2, 128, 1, 1.1, "u8"),
    (40, 50, 72, 1, 0.7, "f32"), (1, 17, 128, 0, 400.0, "f32")])
def test_oracle_matching_equals_reference(n1, n2, dim, flag, thresh, kind):
    a, b = with_ties(*descriptor_sets(n1, n2, dim, 7 + n1, kind))
    o = pyoracle.matching(a, b, thresh, flag)
    nh, no = (4, 8) if dim == 128 else (3, 8)
    (q, ia, ib), descr_ok = reference_matching(refrun.ref_lib(), a, b, thresh, flag, nh, no)
    idx = np.nonzero(o["pass"])[0]
    assert len(idx) > 0 or n1 == 1
    assert np.array_equal(q, idx) and np.array_equal(ia, o["iA"][idx]) and np.array_equal(ib, o["iB"][idx])
    assert descr_ok


def test_oracle_matching_tie_rules():
    # lib_util.c:110-121: on d[0] == d[1] the search starts with A = 1, B = 0
    b = np.zeros((4, 8), np.float32)
    b[2] = 3
    b[3] = 1
    a = np.zeros((3, 8), np.float32)
    a[1] = 3           # nearest is target 2, then 3; 0/1 tie irrelevant
    a[2] = 0.9         # nearest target 3 (0.1 away), second: 0 and 1 tie -> 1
    o = pyoracle.matching(a, b, 10.0, 0)
    assert (o["iA"][0], o["iB"][0]) == (1, 0)
    assert (o["iA"][1], o["iB"][1]) == (2, 3)
    assert (o["iA"][2], o["iB"][2]) == (3, 1)


def test_oracle_ingest_formula():
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    got = pyoracle.ingest_u8(rgb)
    r, g, b = (rgb[..., c].astype(np.float32).astype(np.float64) for c in range(3))
    want = (0.212639005871510 * r + 0.715168678767756 * g + 0.072192315360734 * b).astype(np.float32)
    want = (want.astype(np.float64) / 256.0).astype(np.float32)
    assert np.array_equal(got, want)
    gray = rgb[..., 0].copy()
    assert np.array_equal(pyoracle.ingest_u8(gray), gray.astype(np.float32) / np.float32(256))
    rgba = np.concatenate([rgb, rng.integers(0, 256, (13, 17, 1), dtype=np.uint8)], axis=2)
    assert np.array_equal(pyoracle.ingest_u8(rgba), got)


# ------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2,dim,flag,thresh", [
    (60, 80, 128, 1, 0.6), (60, 80, 128, 0, 250.0), (33, 2, 128, 1, 0.9), (1, 3, 128, 1, 0.6),
    (257, 1031, 128, 1, 0.8), (100, 90, 72, 1, 0.7), (64, 64, 32, 0, 100.0), (50, 70, 5, 1, 0.9),
    (45, 37, 256, 1, 0.8), (70, 65, 108, 0, 300.0)])
def test_match_u8_equals_oracle(hip, n1, n2, dim, flag, thresh):
    from sift_amd import api
    a, b = with_ties(*descriptor_sets(n1, n2, dim, 3 * n1 + dim, "u8"))
    got = api.match(a, b, thresh, flag)
    want = pyoracle.matching(a.astype(np.float32), b.astype(np.float32), thresh, flag)
    assert_same(got, want, f"u8 {n1}x{n2}x{dim}")
    # same values through the float kernel
    gotf = api.match(a.astype(np.float32), b.astype(np.float32), thresh, flag)
    assert_same(gotf, want, f"f32-of-u8 {n1}x{n2}x{dim}")


@pytest.mark.gpu
def test_match_u8_extreme_values(hip):
    """Distances up to sqrt(128)*255 (squared 8.3e6 < 2^24) and float-sqrt collisions of
    distinct squared distances above 2^22."""
    from sift_amd import api
    rng = np.random.default_rng(11)
    a = (rng.integers(0, 2, (40, 128)) * 255).astype(np.uint8)
    b = (rng.integers(0, 2, (300, 128)) * 255).astype(np.uint8)
    b[:150] = np.where(rng.random((150, 128)) < 0.5, b[:150], 255 - a[rng.integers(0, 40, 150)])
    a[0] = 0
    b[7] = 255
    b[8] = 255
    b[8, 0] = 254
    for flag, th in ((1, 0.99), (0, 2000.0)):
        assert_same(api.match(a, b, th, flag), pyoracle.matching(a.astype(np.float32), b.astype(np.float32), th, flag))


@pytest.mark.gpu
@pytest.mark.parametrize("n1,n2,dim,flag,thresh", [
    (60, 80, 128, 1, 0.6), (40, 50, 72, 0, 200.0), (130, 67, 128, 1, 0.9), (3, 2, 12, 1, 0.5)])
def test_match_f32_equals_oracle(hip, n1, n2, dim, flag, thresh):
    from sift_amd import api
    a, b = with_ties(*descriptor_sets(n1, n2, dim, 5 * n1 + dim, "f32"))
    assert_same(api.match(a, b, thresh, flag), pyoracle.matching(a, b, thresh, flag), f"f32 {n1}x{n2}x{dim}")


@pytest.mark.gpu
def test_match_large(hip):
    from sift_amd import api
    a, b = descriptor_sets(4000, 4500, 128, 99, "u8")
    got = api.match(a, b, 0.6, 1)
    want = pyoracle.matching(a.astype(np.float32), b.astype(np.float32), 0.6, 1)
    assert_same(got, want, "large")
    assert want["pass"].sum() > 1000


@pytest.mark.gpu
def test_match_errors(hip):
    from sift_amd import api
    a = np.zeros((4, 128), np.uint8)
    with pytest.raises(api.SiftAmdError):
        api.match(a, a[:1])                       # fewer than two targets
    with pytest.raises(api.SiftAmdError):
        api.match(np.zeros((4, 300), np.uint8), np.zeros((4, 300), np.uint8))
    out = api.match(a[:0], a)
    assert len(out["iA"]) == 0


@pytest.mark.gpu
@pytest.mark.skipif(not refrun.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("kind", ["u8", "f32"])
def test_reference_named_matching(hip, kind):
    """`matching` of the drop-in library appends the same three lists as the reference's."""
    a, b = with_ties(*descriptor_sets(90, 120, 128, 42, kind))
    got, ok1 = reference_matching(hip, a, b, 0.6, 1)
    want, ok2 = reference_matching(refrun.ref_lib(), a, b, 0.6, 1)
    assert ok1 and ok2 and len(want[0]) > 10
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.gpu
def test_match_between_images_of_a_batch(hip):
    """Keypoints of two images of one run matched without leaving the device."""
    from sift_amd import api
    img = synth.make("white", 200, 160, 5)
    shifted = np.roll(img, (3, 5), axis=(0, 1))
    with api.Context(200, 160, max_batch=2) as ctx:
        ctx.detect_host(np.stack([img, shifted]))
        sa, _ = ctx.keypoints(0)
        sb, _ = ctx.keypoints(1)
        got = ctx.match(0, 1, 0.6, 1)
    want = pyoracle.matching(sa["descriptor"].astype(np.float32), sb["descriptor"].astype(np.float32), 0.6, 1)
    assert_same(got, want, "batch images")
    assert len(sa) > 20 and want["pass"].sum() > 10


# ------------------------------------------------------------------ GPU: 8-bit ingest
@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ch", [(96, 80, 1), (97, 53, 3), (130, 67, 4), (64, 64, 3)])
def test_ingest_u8(hip, w, h, ch):
    from sift_amd import api
    rng = np.random.default_rng(w + ch)
    base = (synth.make("smooth", w, h, 9) * 255).astype(np.uint8)
    if ch == 1:
        imgs = np.stack([base, rng.integers(0, 256, (h, w), dtype=np.uint8)])
    else:
        imgs = np.stack([np.stack([np.roll(base, c, axis=1) for c in range(ch)], axis=2),
                         rng.integers(0, 256, (h, w, ch), dtype=np.uint8)])
    want = np.stack([pyoracle.ingest_u8(im) for im in imgs])
    with api.Context(w, h, max_batch=2) as ctx:
        ctx.detect_host_u8(imgs)
        assert np.array_equal(ctx.ingested(), want)
        k_u8 = [ctx.keypoints(i) for i in range(2)]
        ctx.detect_host(want)
        k_f = [ctx.keypoints(i) for i in range(2)]
    for (s1, r1), (s2, r2) in zip(k_u8, k_f):
        assert np.array_equal(s1, s2) and np.array_equal(r1, r2)
    assert len(k_f[1][0]) > 0
    with api.Context(w, h) as ctx:
        assert ctx.L.sift_amd_detect_host_u8(ctx.ptr, imgs.ctypes.data, 1, 2) != 0   # 2 channels: refused
        assert b"channels" in ctx.L.sift_amd_last_error()
