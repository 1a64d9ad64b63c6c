"""Deterministic synthetic inputs (float32, row-major, values in [0,1)).

S1 "smooth": six octaves of value noise (lattice cells 64,32,...,2 px, bilinear
interpolation, amplitudes 0.5,0.25,...) -- a few hundred keypoints at 1080p.
S2 "white": i.i.d. uniform noise -- thousands of keypoints; stresses the descriptor stage.
Both are driven by the LCG  st = st*1664525 + 1013904223 (mod 2^32), u = (st>>8)/2^24,
so the same seed gives the same bits on every machine.
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def lcg_uniform(seed, n):
    """First n outputs u_k = (st_k >> 8) / 2^24 of the LCG started from `seed` (vectorised)."""
    if n == 0:
        return np.zeros(0, np.float32)
    with np.errstate(over="ignore"):
        a_pow = np.cumprod(np.full(n, _A, np.uint32), dtype=np.uint32)       # a^(k+1)
        geo = np.empty(n, np.uint32)                                          # 1+a+..+a^k
        geo[0] = 1
        if n > 1:
            geo[1:] = (np.cumsum(a_pow[:-1], dtype=np.uint32) + np.uint32(1))
        st = a_pow * np.uint32(seed & 0xFFFFFFFF) + _C * geo
    return ((st >> np.uint32(8)).astype(np.float64) / 16777216.0).astype(np.float32)


def white(w, h, seed=12345):
    return lcg_uniform(seed, w * h).reshape(h, w)


def smooth(w, h, seed=12345):
    img = np.zeros((h, w), np.float64)
    amp = 0.5
    cell = 64
    s = seed
    ii = np.arange(h, dtype=np.float64)
    jj = np.arange(w, dtype=np.float64)
    while cell >= 2:
        lh = h // cell + 2
        lw = w // cell + 2
        lat = lcg_uniform(s, lh * lw).astype(np.float64).reshape(lh, lw)
        s = (s * 7919 + 104729) & 0xFFFFFFFF
        fi = ii / cell
        fj = jj / cell
        i0 = np.floor(fi).astype(np.int64)
        j0 = np.floor(fj).astype(np.int64)
        ti = (fi - i0)[:, None]
        tj = (fj - j0)[None, :]
        a = lat[i0][:, j0]
        b = lat[i0][:, j0 + 1]
        c = lat[i0 + 1][:, j0]
        d = lat[i0 + 1][:, j0 + 1]
        img += amp * ((1 - ti) * ((1 - tj) * a + tj * b) + ti * ((1 - tj) * c + tj * d))
        amp *= 0.5
        cell //= 2
    return img.astype(np.float32)


def make(kind, w, h, seed=12345):
    if kind == "smooth":
        return smooth(w, h, seed)
    if kind == "white":
        return white(w, h, seed)
    raise ValueError(kind)
