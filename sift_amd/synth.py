"""Deterministic synthetic inputs (float32, row-major, values in [0,1)).

S1 "smooth": six octaves of value noise (lattice cells 64,32,...,2 px, bilinear
interpolation, amplitudes 0.5,0.25,...) -- a few hundred keypoints at 1080p.
S2 "white": i.i.d. uniform noise -- thousands of keypoints; stresses the descriptor stage.
S3 "natural": a stand-in for photographs - sharp-edged rotated rectangles and ellipses of
random size and brightness, partly transparent, over textured ground (value noise of cells
128..1 px with amplitude proportional to the square root of the cell size); edges are one-pixel linear ramps.
Gives keypoints at edges, corners and blobs of many scales, as natural images do.
All are driven by the LCG  st = st*1664525 + 1013904223 (mod 2^32), u = (st>>8)/2^24,
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


def natural(w, h, seed=12345, n_shapes=120):
    """Only +, -, *, /, sqrt, abs, floor, minimum/maximum on float64 (all correctly rounded), so
    the same seed gives the same bits on every machine."""
    ii = np.arange(h, dtype=np.float64)
    jj = np.arange(w, dtype=np.float64)
    img = np.zeros((h, w), np.float64)
    cell, s, norm = 128, seed & 0xFFFFFFFF, 0.0
    while cell >= 1:                                   # texture with energy at every scale
        lh, lw = h // cell + 2, w // cell + 2
        lat = lcg_uniform(s, lh * lw).astype(np.float64).reshape(lh, lw)
        s = (s * 7919 + 104729) & 0xFFFFFFFF
        fi, fj = ii / cell, jj / cell
        i0, j0 = np.floor(fi).astype(np.int64), np.floor(fj).astype(np.int64)
        ti, tj = (fi - i0)[:, None], (fj - j0)[None, :]
        a, b = lat[i0][:, j0], lat[i0][:, j0 + 1]
        c, d = lat[i0 + 1][:, j0], lat[i0 + 1][:, j0 + 1]
        amp = np.sqrt(float(cell))
        img += amp * ((1 - ti) * ((1 - tj) * a + tj * b) + ti * ((1 - tj) * c + tj * d))
        norm += amp
        cell //= 2
    img = 0.25 + 0.5 * img / norm
    u = lcg_uniform((seed * 2654435761 + 97) & 0xFFFFFFFF, 8 * n_shapes).astype(np.float64).reshape(n_shapes, 8)
    X0, Y0 = ii[:, None], jj[None, :]
    for k in range(n_shapes):
        ci, cj = u[k, 0] * h, u[k, 1] * w
        size = 6.0 + 0.22 * min(w, h) * u[k, 2] * u[k, 2]          # many small shapes, a few large ones
        ha, hb = size, size * (0.3 + 0.7 * u[k, 3])
        da, db = 2 * u[k, 4] - 1, 2 * u[k, 5] - 1                  # direction, normalised below
        nrm = np.sqrt(da * da + db * db)
        if nrm < 1e-3:
            da, db, nrm = 1.0, 0.0, 1.0
        co, si = da / nrm, db / nrm
        X = co * (X0 - ci) + si * (Y0 - cj)
        Y = co * (Y0 - cj) - si * (X0 - ci)
        if k % 2 == 0:                                              # rectangle
            dist = np.maximum(np.abs(X) - ha, np.abs(Y) - hb)
        else:                                                       # ellipse
            dist = (np.sqrt((X / ha) * (X / ha) + (Y / hb) * (Y / hb)) - 1.0) * min(ha, hb)
        cover = np.minimum(np.maximum(0.5 - dist, 0.0), 1.0)       # one-pixel linear ramp at the edge
        level = 0.08 + 0.84 * u[k, 6]
        alpha = 0.55 + 0.45 * u[k, 7]                               # the texture shows through
        img = img * (1 - alpha * cover) + level * (alpha * cover)
    return np.minimum(np.maximum(img, 0.0), 0.999).astype(np.float32)


def make(kind, w, h, seed=12345):
    if kind == "smooth":
        return smooth(w, h, seed)
    if kind == "white":
        return white(w, h, seed)
    if kind == "natural":
        return natural(w, h, seed)
    raise ValueError(kind)
