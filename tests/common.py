"""Shared helpers for the parity tests."""
import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(bits(a), bits(b))


def circ_diff(a, b):
    d = np.abs(a - b) % (2 * np.pi)
    return np.minimum(d, 2 * np.pi - d)


# Tolerances stated by the target: refined (x, y, sigma, theta) within 1e-3 px/rad,
# quantised descriptors within +-1 per bin, identical discrete-extrema counts per octave.
TOL_XYS = 1e-3
TOL_THETA = 1e-3
TOL_DESCR = 1


def compare_records(got, want, n_oct, check_ss=True):
    """Compare two full-stage records (layout of oracle/refrun.py).  Returns a dict of
    statistics; raises AssertionError on anything outside the stated tolerances."""
    stats = {}
    if check_ss and "gss" in want and "gss" in got:
        for o in range(n_oct):
            assert same_bits(got["gss"][o], want["gss"][o]), f"gss octave {o} differs"
            assert same_bits(got["dog"][o], want["dog"][o]), f"dog octave {o} differs"
    for name in "ABCDEF":
        g, w = got["stages"][name], want["stages"][name]
        gc = np.bincount(g["ints"][:, 0], minlength=n_oct)
        wc = np.bincount(w["ints"][:, 0], minlength=n_oct)
        assert np.array_equal(gc, wc), f"stage {name}: per-octave counts {gc} != {wc}"
        assert np.array_equal(g["ints"], w["ints"]), f"stage {name}: (o,s,i,j) differ"
        cols = [0, 1, 2, 4]          # x, y, sigma, val
        if len(w["ints"]):
            d = np.abs(g["flts"][:, cols] - w["flts"][:, cols])
            assert d[:, :3].max() <= TOL_XYS, f"stage {name}: x/y/sigma off by {d[:, :3].max()}"
            assert d[:, 3].max() <= 1e-6, f"stage {name}: val off by {d[:, 3].max()}"
            stats[f"{name}_exact"] = float(np.mean(bits(np.ascontiguousarray(g["flts"][:, cols]))
                                                   == bits(np.ascontiguousarray(w["flts"][:, cols]))))
    gk, wk = got["keys"], want["keys"]
    assert len(gk["ints"]) == len(wk["ints"]), f"final keypoints {len(gk['ints'])} != {len(wk['ints'])}"
    assert np.array_equal(gk["ints"], wk["ints"]), "final keypoints: (o,s,i,j) differ"
    if len(wk["ints"]):
        d = np.abs(gk["flts"][:, :3] - wk["flts"][:, :3])
        assert d.max() <= TOL_XYS
        dth = circ_diff(gk["flts"][:, 3].astype(np.float64), wk["flts"][:, 3].astype(np.float64))
        assert dth.max() <= TOL_THETA, f"theta off by {dth.max()}"
        dd = np.abs(gk["descr"] - wk["descr"])
        assert dd.max() <= TOL_DESCR, f"descriptor bins off by {dd.max()}"
        stats["theta_exact"] = float(np.mean(bits(np.ascontiguousarray(gk["flts"][:, 3]))
                                             == bits(np.ascontiguousarray(wk["flts"][:, 3]))))
        stats["descr_exact"] = float(np.mean(gk["descr"] == wk["descr"]))
        stats["orihist_exact"] = float(np.mean(bits(gk["orihist"]) == bits(wk["orihist"])))
    stats["n_keys"] = len(wk["ints"])
    return stats
