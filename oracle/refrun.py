"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Runs a library that exports the reference-named C API (the compiled reference in
oracle/_ref/, or the HIP drop-in) through ctypes and returns every stage as numpy
arrays in one common layout, so parity tests can diff any two implementations.

Layout of the returned dict ("record"):
  octaves : list of {w,h,delta,sigmas}                     geometry per octave
  gss/dog/gx/gy : list (per octave) of (nSca,h,w) float32  (present if keep_ss)
  stages  : {"A".."F"} -> {"ints": (n,4) o,s,i,j ; "flts": (n,6) x,y,sigma,theta,val,edgeResp}
            A discrete extrema, B |val|>0.8thr, C interpolated, D |val|>thr, E edge, F border
            (lib_sift_anatomy.c:843-848)
  keys    : same + "descr" (n,128) float32 holding 0..255, "orihist" (n,36)
"""
import ctypes as C
import os

import numpy as np

from sift_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libsift_ref.so")


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        _ref = abi.bind_reference_api(C.CDLL(REF_SO))
    return _ref


def run_sift_anatomy(lib, img, params=None, keep_ss=True):
    """Call lib.sift_anatomy exactly as sift_cli.c:285-292 does and unpack everything."""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    p = params if params is not None else abi.SiftParameters.default()
    P = C.POINTER
    kk = (P(abi.SiftKeypoints) * 6)(*[lib.sift_malloc_keypoints() for _ in range(6)])
    ss = (P(abi.SiftScalespace) * 4)()
    k = lib.sift_anatomy(img.ctypes.data_as(abi.c_float_p), w, h, C.byref(p), ss, kk)
    rec = {"stages": {}}
    for name, lst in zip("ABCDEF", kk):
        rec["stages"][name] = abi.keypoints_to_numpy(lst, with_descr=False)
    rec["keys"] = abi.keypoints_to_numpy(k, with_descr=True)
    g = abi.scalespace_to_numpy(ss[0])
    rec["octaves"] = [{"w": o["w"], "h": o["h"], "delta": o["delta"], "sigmas": o["sigmas"]}
                      for o in g]
    if keep_ss:
        rec["gss"] = [o["stack"] for o in g]
        rec["dog"] = [o["stack"] for o in abi.scalespace_to_numpy(ss[1])]
        rec["gx"] = [o["stack"] for o in abi.scalespace_to_numpy(ss[2])]
        rec["gy"] = [o["stack"] for o in abi.scalespace_to_numpy(ss[3])]
    lib.sift_free_keypoints(k)
    for lst in kk:
        lib.sift_free_keypoints(lst)
    for s in ss:
        lib.sift_free_scalespace(s)
    return rec


def counts_per_octave(ints, n_oct):
    return np.bincount(ints[:, 0], minlength=n_oct)[:n_oct].astype(np.int64)
