"""Pins the oracle: bit-for-bit comparison with the reference sources compiled in place.
Only runs where the reference is present (this container); elsewhere test_oracle_golden.py
checks the same oracle against fixtures the compiled reference produced."""
import numpy as np
import pytest

from sift_amd import abi, synth
from oracle import pyoracle, refrun

from common import same_bits

pytestmark = pytest.mark.skipif(not refrun.have_ref(), reason="compiled reference not available")

CASES = [
    ("white", 64, 48, {}),
    ("white", 135, 101, {}),
    ("smooth", 200, 150, {}),
    ("white", 257, 131, {}),
    ("smooth", 640, 480, {}),
    ("white", 160, 120, {"n_spo": 4}),
    ("white", 160, 120, {"n_spo": 2, "C_DoG": 0.02, "C_edge": 6}),
    ("white", 128, 96, {"delta_min": 1.0}),
    ("white", 150, 100, {"delta_min": 0.75}),
    ("smooth", 256, 192, {"sigma_min": 1.6, "n_oct": 3, "lambda_descr": 5, "lambda_ori": 1.2}),
    ("white", 96, 96, {"n_bins": 24, "t": 0.6, "itermax": 3}),
]


@pytest.mark.parametrize("kind,w,h,kw", CASES)
def test_every_stage_bit_identical(kind, w, h, kw):
    img = synth.make(kind, w, h, seed=99 + w + h)
    p = abi.SiftParameters.default().copy(**kw)
    ref = refrun.run_sift_anatomy(refrun.ref_lib(), img, p)
    ora = pyoracle.run(img, p)
    assert len(ref["octaves"]) == len(ora["octaves"])
    for o, (a, b) in enumerate(zip(ref["octaves"], ora["octaves"])):
        assert (a["w"], a["h"], a["delta"]) == (b["w"], b["h"], b["delta"])
        assert same_bits(a["sigmas"], b["sigmas"])
        assert same_bits(ref["gss"][o], ora["gss"][o]), f"gss {o}"
        assert same_bits(ref["dog"][o], ora["dog"][o]), f"dog {o}"
        for s in range(ref["gss"][o].shape[0]):
            gi, gj = pyoracle.gradient(ora["gss"][o][s])
            assert same_bits(ref["gx"][o][s], gi) and same_bits(ref["gy"][o][s], gj)
    for name in "ABCDEF":
        assert np.array_equal(ref["stages"][name]["ints"], ora["stages"][name]["ints"]), name
        cols = [0, 1, 2, 4]
        assert same_bits(np.ascontiguousarray(ref["stages"][name]["flts"][:, cols]),
                         np.ascontiguousarray(ora["stages"][name]["flts"][:, cols])), name
    # edge response is defined from stage D on
    for name in "DEF":
        assert same_bits(np.ascontiguousarray(ref["stages"][name]["flts"][:, 5]),
                         np.ascontiguousarray(ora["stages"][name]["flts"][:, 5])), name
    for f in ("ints", "flts", "descr", "orihist"):
        assert same_bits(ref["keys"][f], ora["keys"][f]), f


def test_scalars_match_reference_defaults():
    lib = refrun.ref_lib()
    p = lib.sift_assign_default_parameters().contents
    assert bytes(p) == bytes(abi.SiftParameters.default())
