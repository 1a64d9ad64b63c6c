"""The oracle against the committed golden fixtures (produced by the compiled reference,
see tests/golden/make_golden.py).  Runs anywhere, no reference and no GPU needed."""
import hashlib
import json
import os

import numpy as np
import pytest

from sift_amd import abi, synth
from oracle import pyoracle, refrun

from common import same_bits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))
CPU_CASES = [n for n, e in INDEX.items() if e["w"] * e["h"] <= 1920 * 1080 and n != "white_1920x1080"]


def digest(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_against_golden(rec, entry, name):
    n_oct = entry["n_oct"]
    assert len(rec["octaves"]) == n_oct
    assert [[o["w"], o["h"]] for o in rec["octaves"]] == entry["dims"]
    assert [[float(s).hex() for s in o["sigmas"]] for o in rec["octaves"]] == entry["sigmas"]
    if "gss" in rec:
        assert [digest(g) for g in rec["gss"]] == entry["gss_sha1"], "Gaussian scale-space bits"
        assert [digest(d) for d in rec["dog"]] == entry["dog_sha1"], "DoG bits"
    for s in "ABCDEF":
        got = refrun.counts_per_octave(rec["stages"][s]["ints"], n_oct).tolist()
        assert got == entry["stage_counts_per_octave"][s], f"stage {s} per-octave counts"
    keys = rec["keys"]
    assert len(keys["ints"]) == entry["n_keys"]
    path = os.path.join(GOLD, name + ".npz")
    if os.path.exists(path):
        g = np.load(path)
        assert np.array_equal(keys["ints"], g["ints"])
        return g
    return None


@pytest.mark.parametrize("name", CPU_CASES)
def test_oracle_reproduces_golden(name):
    e = INDEX[name]
    img = synth.make(e["kind"], e["w"], e["h"], e["seed"])
    assert digest(img) == e["input_sha1"], "synthetic input generator changed"
    p = abi.SiftParameters.default().copy(**e["params"])
    rec = pyoracle.run(img, p)
    g = check_against_golden(rec, e, name)
    keys = rec["keys"]
    assert digest(keys["ints"]) == e["keys_sha1"]["ints"]
    assert digest(keys["flts"][:, :4]) == e["keys_sha1"]["xyst"]
    assert digest(keys["descr"].astype(np.uint8)) == e["keys_sha1"]["descr_u8"]
    if g is not None:
        assert same_bits(keys["flts"], g["flts"])
        assert np.array_equal(keys["descr"].astype(np.uint8), g["descr"])
        assert same_bits(keys["orihist"], g["orihist"])
        assert np.array_equal(rec["stages"]["C"]["ints"], g["refined_ints"])


def test_tap_tables():
    """Blur radii of the default schedule are 5 (seed) then 5,7,8,10,13 in every octave."""
    p = abi.SiftParameters.default()
    rec = pyoracle.run(synth.white(64, 48, 1), p, keep_ss=False)
    sig = rec["octaves"][0]["sigmas"]
    radii = []
    for s in range(1, len(sig)):
        se = np.float32(np.sqrt(np.float64(np.float32(sig[s] * sig[s]) - np.float32(sig[s - 1] * sig[s - 1])))
                        / np.float64(rec["octaves"][0]["delta"]))
        r, taps = pyoracle.taps(float(se))
        radii.append(r)
        assert abs(float(taps[0] + 2 * taps[1:].sum()) - 1.0) < 1e-6
    assert radii == [5, 7, 8, 10, 13]
