"""The `--record` directory written by sift_amd.record: layout and, on the GPU, byte
equality with the same directory written from the compiled reference's results."""
import filecmp
import json
import os

import numpy as np
import pytest

from oracle import refrun
from sift_amd import abi, record, synth

STAGES = ["extrema", "contrast_pre", "contrast_post", "edge", "border"]


def _tree(root):
    out = []
    for d, _, files in os.walk(root):
        for f in files:
            out.append(os.path.relpath(os.path.join(d, f), root))
    return sorted(out)


def _check_layout(root, n_oct, n_spo):
    """Reads the directory back the way tests/test_sift_unittest.py:105-130,179-216,383-516 does."""
    for stem, nimg in (("gss", n_spo + 3), ("dog", n_spo + 2), ("grad_x", n_spo + 3),
                       ("grad_y", n_spo + 3)):
        meta = json.load(open(os.path.join(root, stem, stem + "_meta.json")))
        assert meta["n_octaves"] == n_oct and meta["stem"] == stem
        for o, oc in enumerate(meta["octaves"]):
            assert oc["o"] == o and len(oc["files"]) == nimg == len(oc["sigmas"])
            for s, f in enumerate(oc["files"]):
                assert f == "%s_o%03d_s%03d.f32" % (stem, o, s)
                m = np.fromfile(os.path.join(root, stem, f), np.float32)
                assert m.size == oc["w"] * oc["h"]
    counts = {}
    for st in STAGES:
        meta = json.load(open(os.path.join(root, st, "extrema_meta.json")))
        assert meta["int_order"] == ["o", "s", "i", "j"]
        assert meta["float_order"] == ["y", "x", "sigma", "val"]
        ints = np.fromfile(os.path.join(root, st, meta["int_file"]), np.int32).reshape(-1, 4)
        flts = np.fromfile(os.path.join(root, st, meta["float_file"]), np.float32).reshape(-1, 4)
        assert len(ints) == len(flts) == meta["count"]
        counts[st] = meta["count"]
    meta = json.load(open(os.path.join(root, "refined", "extrema_refined_meta.json")))
    assert np.fromfile(os.path.join(root, "refined", meta["int_file"]), np.int32).size == 4 * meta["count"]
    counts["refined"] = meta["count"]
    assert counts["extrema"] >= counts["contrast_pre"] >= counts["refined"] >= \
        counts["contrast_post"] >= counts["edge"] >= counts["border"]
    meta = json.load(open(os.path.join(root, "keys", "keys_meta.json")))
    assert meta["float_order"] == ["y", "x", "sigma", "theta"]
    n = meta["count"]
    d = np.fromfile(os.path.join(root, "keys", meta["desc_file"]), np.uint8)
    assert d.size == n * meta["desc_len"]
    assert np.fromfile(os.path.join(root, "keys", meta["float_file"]), np.float32).size == 4 * n
    return counts, n


def test_descriptor_serialisation():
    # sift_cli.c:535-545: dst[(u*NH+v)*NO+o] = src[(v*NH+u)*NO+o], clamp, +0.5, truncate
    rng = np.random.default_rng(3)
    nh, no = 4, 8
    src = rng.uniform(-5, 270, (7, nh * nh * no)).astype(np.float32)
    src[0, :4] = [254.5, 254.49, 0.5, 0.49]
    got = record.serialise_descriptors(src, nh, no)
    want = np.zeros_like(got)
    for k in range(len(src)):
        for u in range(nh):
            for v in range(nh):
                for o in range(no):
                    vv = min(max(src[k, (v * nh + u) * no + o], np.float32(0)), np.float32(255))
                    want[k, (u * nh + v) * no + o] = int(np.float32(vv + np.float32(0.5)))
    assert np.array_equal(got, want)


def test_gray_conversion():
    # io_png.c:499-516 then sift_cli.c:280-281
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    g = record.gray_from_u8(rgb)
    r, gg, b = [float(v) for v in rgb[4, 7]]
    want = np.float32(np.float32(0.212639005871510 * r + 0.715168678767756 * gg
                                 + 0.072192315360734 * b) / 256.0)
    assert g.dtype == np.float32 and g[4, 7] == want
    assert np.array_equal(record.gray_from_u8(rgb[:, :, 0]), rgb[:, :, 0].astype(np.float32) / 256)
    rgba = np.concatenate([rgb, np.full((9, 11, 1), 77, np.uint8)], axis=2)
    assert np.array_equal(record.gray_from_u8(rgba), g)


def test_cli_grammar():
    path, p, out, size = record.parse_cli(["-ss_noct", "5", "img.png", "--record", "d",
                                           "-thresh_dog", "0.02", "-size", "64x48"])
    assert (path, out, size) == ("img.png", "d", (64, 48))
    assert p.n_oct == 5 and abs(p.C_DoG - 0.02) < 1e-9 and p.n_spo == 3
    with pytest.raises(SystemExit):
        record.parse_cli(["img.png", "-bogus", "1"])
    with pytest.raises(SystemExit):
        record.parse_cli(["img.png", "-ss_noct"])
    with pytest.raises(SystemExit):
        record.parse_cli(["--record", "d"])


@pytest.mark.skipif(not refrun.have_ref(), reason="oracle/_ref not built")
def test_record_layout_from_reference_results(tmp_path):
    img = synth.make("white", 96, 80, 11)
    p = abi.SiftParameters.default()
    info = record.write_record(refrun.ref_lib(), img, str(tmp_path / "rec"), p)
    counts, n = _check_layout(str(tmp_path / "rec"), info["n_octaves"], p.n_spo)
    assert n == info["n_keys"] > 0
    # the lists agree with what the common unpacker of the parity tests sees
    rec = refrun.run_sift_anatomy(refrun.ref_lib(), img, p, keep_ss=True)
    ints = np.fromfile(tmp_path / "rec" / "extrema" / "extrema_int.i32", np.int32).reshape(-1, 4)
    assert np.array_equal(ints, rec["stages"]["A"]["ints"])
    # grad_x holds the 4th scale-space, grad_y the 3rd (sift_cli.c:315,320)
    gx = np.fromfile(tmp_path / "rec" / "grad_x" / "grad_x_o001_s002.f32", np.float32)
    assert np.array_equal(gx, rec["gy"][1][2].ravel())
    gy = np.fromfile(tmp_path / "rec" / "grad_y" / "grad_y_o001_s002.f32", np.float32)
    assert np.array_equal(gy, rec["gx"][1][2].ravel())


@pytest.mark.gpu
@pytest.mark.skipif(not refrun.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("kind,w,h,kw", [("white", 160, 120, {}),
                                         ("smooth", 131, 97, {"n_spo": 4, "n_hist": 3}),
                                         ("white", 40, 36, {"C_DoG": 0.5})])
def test_record_identical_to_reference(hip, tmp_path, kind, w, h, kw):
    """Same files, same bytes (the JSON "dir" field aside, which holds the output path)."""
    img = synth.make(kind, w, h, 21)
    p = abi.SiftParameters.default().copy(**kw)
    a, b = str(tmp_path / "hip"), str(tmp_path / "ref")
    record.write_record(hip, img, a, p)
    record.write_record(refrun.ref_lib(), img, b, p)
    files = _tree(a)
    assert files == _tree(b) and len(files) > 20
    for f in files:
        if f.endswith("_meta.json") and f.split(os.sep)[0] in ("gss", "dog", "grad_x", "grad_y"):
            ja, jb = json.load(open(os.path.join(a, f))), json.load(open(os.path.join(b, f)))
            ja.pop("dir"), jb.pop("dir")
            assert ja == jb, f
        else:
            assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f
