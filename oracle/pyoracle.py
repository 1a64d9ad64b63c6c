"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/liboracle.so (sift_oracle.c).

Returns the same "record" dict layout as oracle/refrun.py so the two can be diffed.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from sift_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")

OS_MAX_OCT, OS_MAX_SCA = 32, 16


class OsPoint(C.Structure):
    _fields_ = [("o", C.c_int), ("s", C.c_int), ("i", C.c_int), ("j", C.c_int),
                ("x", C.c_float), ("y", C.c_float), ("sigma", C.c_float),
                ("theta", C.c_float), ("val", C.c_float), ("edge", C.c_float)]


OS_POINT_DTYPE = np.dtype([("o", "<i4"), ("s", "<i4"), ("i", "<i4"), ("j", "<i4"),
                           ("x", "<f4"), ("y", "<f4"), ("sigma", "<f4"), ("theta", "<f4"),
                           ("val", "<f4"), ("edge", "<f4")])


class OsResult(C.Structure):
    _fields_ = [("n_oct", C.c_int), ("n_sca", C.c_int),
                ("w", C.c_int * OS_MAX_OCT), ("h", C.c_int * OS_MAX_OCT),
                ("delta", C.c_float * OS_MAX_OCT),
                ("sigmas", (C.c_float * OS_MAX_SCA) * OS_MAX_OCT),
                ("thresh", C.c_float), ("thresh_pre", C.c_float), ("thresh_edge", C.c_float),
                ("gss", abi.c_float_p * OS_MAX_OCT), ("dog", abi.c_float_p * OS_MAX_OCT),
                ("n_stage", C.c_int * 6), ("stage", C.POINTER(OsPoint) * 6),
                ("n_keys", C.c_int), ("n_descr", C.c_int), ("n_bins", C.c_int),
                ("keys", C.POINTER(OsPoint)), ("descr", abi.c_float_p),
                ("orihist", abi.c_float_p)]


def build(force=False):
    if force or not os.path.exists(SO) or \
            os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "sift_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(SO)
        L.os_sift.restype = C.POINTER(OsResult)
        L.os_sift.argtypes = [abi.c_float_p, C.c_int, C.c_int, C.POINTER(abi.SiftParameters), C.c_int]
        L.os_free.argtypes = [C.POINTER(OsResult)]
        L.os_free.restype = None
        L.os_hash_floats.restype = C.c_uint64
        L.os_hash_floats.argtypes = [abi.c_float_p, C.c_size_t]
        L.os_gaussian_taps.restype = C.c_int
        L.os_gaussian_taps.argtypes = [C.c_float, abi.c_float_p, C.c_int]
        L.os_blur.restype = None
        L.os_blur.argtypes = [abi.c_float_p, abi.c_float_p, C.c_int, C.c_int, C.c_float]
        L.os_gradient.restype = None
        L.os_gradient.argtypes = [abi.c_float_p, abi.c_float_p, abi.c_float_p, C.c_int, C.c_int]
        L.os_number_of_octaves.restype = C.c_int
        L.os_number_of_octaves.argtypes = [C.c_int, C.c_int, C.POINTER(abi.SiftParameters)]
        L.os_convert_threshold.restype = C.c_float
        L.os_convert_threshold.argtypes = [C.POINTER(abi.SiftParameters)]
        L.os_matching.restype = C.c_int
        L.os_matching.argtypes = [abi.c_float_p, C.c_int, abi.c_float_p, C.c_int, C.c_int, C.c_float,
                                  C.c_int, abi.c_int_p, abi.c_int_p, abi.c_float_p, abi.c_float_p,
                                  C.POINTER(C.c_ubyte)]
        L.os_ingest_u8.restype = None
        L.os_ingest_u8.argtypes = [C.POINTER(C.c_ubyte), C.c_int, abi.c_float_p, C.c_size_t]
        _lib = L
    return _lib


def _points(ptr, n):
    if n == 0:
        return np.zeros(0, OS_POINT_DTYPE)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_char)),
                                 shape=(n * OS_POINT_DTYPE.itemsize,)).view(OS_POINT_DTYPE).copy()


def _split(pts):
    ints = np.stack([pts["o"], pts["s"], pts["i"], pts["j"]], 1).astype(np.int32) \
        if len(pts) else np.zeros((0, 4), np.int32)
    flts = np.stack([pts["x"], pts["y"], pts["sigma"], pts["theta"], pts["val"], pts["edge"]], 1) \
        if len(pts) else np.zeros((0, 6), np.float32)
    return {"ints": ints, "flts": flts.astype(np.float32)}


def taps(sigma):
    buf = (C.c_float * 256)()
    r = lib().os_gaussian_taps(C.c_float(sigma), buf, 256)
    return r, np.array(buf[:r + 1], np.float32)


def hash_floats(a):
    a = np.ascontiguousarray(a, np.float32)
    return int(lib().os_hash_floats(a.ctypes.data_as(abi.c_float_p), a.size))


def run(img, params=None, keep_ss=True, detect_only=False):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    p = params if params is not None else abi.SiftParameters.default()
    L = lib()
    rp = L.os_sift(img.ctypes.data_as(abi.c_float_p), w, h, C.byref(p), 1 if detect_only else 0)
    r = rp.contents
    rec = {"octaves": [], "stages": {}, "thresh": (r.thresh, r.thresh_pre, r.thresh_edge)}
    for o in range(r.n_oct):
        rec["octaves"].append({"w": r.w[o], "h": r.h[o], "delta": float(r.delta[o]),
                               "sigmas": np.array(r.sigmas[o][:r.n_sca], np.float32)})
    if keep_ss:
        rec["gss"], rec["dog"] = [], []
        for o in range(r.n_oct):
            n = r.w[o] * r.h[o]
            rec["gss"].append(np.ctypeslib.as_array(r.gss[o], shape=(r.n_sca * n,)).copy()
                              .reshape(r.n_sca, r.h[o], r.w[o]))
            rec["dog"].append(np.ctypeslib.as_array(r.dog[o], shape=((r.n_sca - 1) * n,)).copy()
                              .reshape(r.n_sca - 1, r.h[o], r.w[o]))
    for k, name in enumerate("ABCDEF"):
        rec["stages"][name] = _split(_points(r.stage[k], r.n_stage[k]))
    if not detect_only:
        keys = _split(_points(r.keys, r.n_keys))
        n = r.n_keys
        keys["descr"] = (np.ctypeslib.as_array(r.descr, shape=(n * r.n_descr,)).copy()
                         .reshape(n, r.n_descr) if n else np.zeros((0, r.n_descr), np.float32))
        keys["orihist"] = (np.ctypeslib.as_array(r.orihist, shape=(n * r.n_bins,)).copy()
                           .reshape(n, r.n_bins) if n else np.zeros((0, r.n_bins), np.float32))
        rec["keys"] = keys
    L.os_free(rp)
    return rec


def gradient(im):
    im = np.ascontiguousarray(im, np.float32)
    h, w = im.shape
    gi = np.empty_like(im)
    gj = np.empty_like(im)
    lib().os_gradient(im.ctypes.data_as(abi.c_float_p), gi.ctypes.data_as(abi.c_float_p),
                      gj.ctypes.data_as(abi.c_float_p), w, h)
    return gi, gj


def matching(d1, d2, thresh=0.6, flag=1):
    """-> dict iA, iB, dA, dB, pass for every row of d1 (float descriptors n x dim)."""
    d1 = np.ascontiguousarray(d1, np.float32)
    d2 = np.ascontiguousarray(d2, np.float32)
    n1, dim = d1.shape
    n2 = d2.shape[0]
    assert n2 >= 2 and d2.shape[1] == dim
    iA, iB = np.zeros(n1, np.int32), np.zeros(n1, np.int32)
    dA, dB = np.zeros(n1, np.float32), np.zeros(n1, np.float32)
    ok = np.zeros(n1, np.uint8)
    lib().os_matching(d1.ctypes.data_as(abi.c_float_p), n1, d2.ctypes.data_as(abi.c_float_p), n2, dim,
                      C.c_float(thresh), flag, iA.ctypes.data_as(abi.c_int_p),
                      iB.ctypes.data_as(abi.c_int_p), dA.ctypes.data_as(abi.c_float_p),
                      dB.ctypes.data_as(abi.c_float_p), ok.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return {"iA": iA, "iB": iB, "dA": dA, "dB": dB, "pass": ok}


def ingest_u8(a):
    """u8 (h,w) or (h,w,3|4) -> float32 (h,w) as the reference's loader + sift_cli produce it."""
    a = np.ascontiguousarray(a, np.uint8)
    ch = 1 if a.ndim == 2 else a.shape[2]
    out = np.empty(a.shape[:2], np.float32)
    lib().os_ingest_u8(a.ctypes.data_as(C.POINTER(C.c_ubyte)), ch, out.ctypes.data_as(abi.c_float_p),
                       out.size)
    return out
