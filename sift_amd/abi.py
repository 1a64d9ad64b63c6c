"""ctypes mirrors of the C structs that cross the drop-in boundary.

Layouts follow the reference headers so that pointers can be exchanged with
unmodified host code:
  struct sift_parameters   sift_anatomy/src/lib_sift_anatomy.h:63-67
  struct keypoint          sift_anatomy/src/lib_keypoint.h:76-96
  struct sift_keypoints    sift_anatomy/src/lib_keypoint.h:106-110
  struct octa              sift_anatomy/src/lib_scalespace.h:62-69
  struct sift_scalespace   sift_anatomy/src/lib_scalespace.h:71-74
  struct sift_keypoint_std sift_anatomy/src/lib_sift.h:73-80
"""
import ctypes as C

import numpy as np

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class SiftParameters(C.Structure):
    _fields_ = [
        ("n_oct", C.c_int), ("n_spo", C.c_int), ("n_hist", C.c_int),
        ("n_bins", C.c_int), ("n_ori", C.c_int), ("itermax", C.c_int),
        ("sigma_min", C.c_float), ("delta_min", C.c_float), ("sigma_in", C.c_float),
        ("C_DoG", C.c_float), ("C_edge", C.c_float), ("lambda_ori", C.c_float),
        ("t", C.c_float), ("lambda_descr", C.c_float),
    ]

    @classmethod
    def default(cls):
        """Values of sift_assign_default_parameters (lib_sift_anatomy.c:771-789)."""
        return cls(n_oct=8, n_spo=3, n_hist=4, n_bins=36, n_ori=8, itermax=5,
                   sigma_min=0.8, delta_min=0.5, sigma_in=0.5, C_DoG=0.013333333,
                   C_edge=10, lambda_ori=1.5, t=0.80, lambda_descr=6)

    def copy(self, **kw):
        q = SiftParameters.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            setattr(q, k, v)
        return q


class Keypoint(C.Structure):
    _fields_ = [
        ("x", C.c_float), ("y", C.c_float), ("sigma", C.c_float), ("theta", C.c_float),
        ("o", C.c_int), ("s", C.c_int), ("i", C.c_int), ("j", C.c_int),
        ("val", C.c_float), ("edgeResp", C.c_float),
        ("n_hist", C.c_int), ("n_ori", C.c_int),
        ("descr", c_float_p),
        ("n_bins", C.c_int),
        ("orihist", c_float_p),
    ]


class SiftKeypoints(C.Structure):
    _fields_ = [("size", C.c_int), ("capacity", C.c_int),
                ("list", C.POINTER(C.POINTER(Keypoint)))]


class Octa(C.Structure):
    _fields_ = [("delta", C.c_float), ("w", C.c_int), ("h", C.c_int), ("nSca", C.c_int),
                ("sigmas", c_float_p), ("imStack", c_float_p)]


class SiftScalespace(C.Structure):
    _fields_ = [("nOct", C.c_int), ("octaves", C.POINTER(Octa) * 100)]


class SiftKeypointStd(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("scale", C.c_float),
                ("orientation", C.c_float), ("descriptor", C.c_ubyte * 128)]


assert C.sizeof(SiftParameters) == 56
assert C.sizeof(Keypoint) == 72 and Keypoint.descr.offset == 48 and Keypoint.orihist.offset == 64
assert C.sizeof(SiftScalespace) == 808 and C.sizeof(Octa) == 32
assert C.sizeof(SiftKeypointStd) == 144

KEYPOINT_STD_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("scale", "<f4"),
                               ("orientation", "<f4"), ("descriptor", "u1", (128,))])
assert KEYPOINT_STD_DTYPE.itemsize == 144


def bind_reference_api(lib):
    """Declare argtypes/restypes of the reference-named entry points on a loaded library.

    Used for any library exporting the lib_sift_anatomy.h / lib_sift.h / lib_keypoint.h /
    lib_scalespace.h symbols (the HIP drop-in and, in the tests, the compiled reference).
    """
    P = C.POINTER
    lib.sift_assign_default_parameters.restype = P(SiftParameters)
    lib.sift_assign_default_parameters.argtypes = []
    lib.sift_malloc_keypoints.restype = P(SiftKeypoints)
    lib.sift_malloc_keypoints.argtypes = []
    lib.sift_free_keypoints.restype = None
    lib.sift_free_keypoints.argtypes = [P(SiftKeypoints)]
    lib.sift_free_scalespace.restype = None
    lib.sift_free_scalespace.argtypes = [P(SiftScalespace)]
    lib.sift_malloc_keypoint.restype = P(Keypoint)
    lib.sift_malloc_keypoint.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.sift_add_keypoint_to_list.restype = None
    lib.sift_add_keypoint_to_list.argtypes = [P(Keypoint), P(SiftKeypoints)]
    lib.sift_anatomy.restype = P(SiftKeypoints)
    lib.sift_anatomy.argtypes = [c_float_p, C.c_int, C.c_int, P(SiftParameters),
                                 P(P(SiftScalespace)), P(P(SiftKeypoints))]
    lib.sift_anatomy_without_description.restype = P(SiftKeypoints)
    lib.sift_anatomy_without_description.argtypes = lib.sift_anatomy.argtypes
    lib.sift_anatomy_only_description.restype = None
    lib.sift_anatomy_only_description.argtypes = [c_float_p, C.c_int, C.c_int,
                                                  P(SiftParameters), P(SiftKeypoints)]
    lib.sift_anatomy_orientation_and_description.restype = None
    lib.sift_anatomy_orientation_and_description.argtypes = \
        lib.sift_anatomy_only_description.argtypes
    lib.sift_malloc_scalespace_lowe.restype = P(SiftScalespace)
    lib.sift_malloc_scalespace_lowe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_float, C.c_float]
    lib.sift_malloc_scalespace_from_model.restype = P(SiftScalespace)
    lib.sift_malloc_scalespace_from_model.argtypes = [P(SiftScalespace)]
    lib.sift_malloc_scalespace_dog_from_scalespace.restype = P(SiftScalespace)
    lib.sift_malloc_scalespace_dog_from_scalespace.argtypes = [P(SiftScalespace)]
    lib.scalespace_compute.restype = None
    lib.scalespace_compute.argtypes = [P(SiftScalespace), c_float_p, C.c_int, C.c_int, C.c_float]
    lib.scalespace_compute_gradient.restype = None
    lib.scalespace_compute_gradient.argtypes = [P(SiftScalespace), P(SiftScalespace),
                                                P(SiftScalespace)]
    lib.sift_compute_features.restype = P(SiftKeypointStd)
    lib.sift_compute_features.argtypes = [c_float_p, C.c_int, C.c_int, c_int_p]
    lib.sift_compute_points.restype = P(SiftKeypointStd)
    lib.sift_compute_points.argtypes = [c_float_p, C.c_int, C.c_int, c_int_p]
    lib.sift_fill_descriptors.restype = None
    lib.sift_fill_descriptors.argtypes = [c_float_p, C.c_int, C.c_int, P(SiftKeypointStd), C.c_int]
    lib.sift_find_ori_and_fill_descriptors.restype = None
    lib.sift_find_ori_and_fill_descriptors.argtypes = lib.sift_fill_descriptors.argtypes
    if hasattr(lib, "matching"):
        lib.matching.restype = None
        lib.matching.argtypes = [P(SiftKeypoints)] * 5 + [C.c_float, C.c_int]
    return lib


def keypoints_from_numpy(lib, descr, n_hist=4, n_ori=8, n_bins=36, xy=None):
    """Build a malloc'd struct sift_keypoints list holding the given float descriptors
    (n x n_hist*n_hist*n_ori); x,y default to (row index, 0).  Free with sift_free_keypoints."""
    descr = np.ascontiguousarray(descr, np.float32)
    lst = lib.sift_malloc_keypoints()
    for i in range(len(descr)):
        k = lib.sift_malloc_keypoint(n_ori, n_hist, n_bins)
        C.memmove(k.contents.descr, descr[i].ctypes.data, descr.shape[1] * 4)
        k.contents.x, k.contents.y = (float(i), 0.0) if xy is None else (float(xy[i][0]), float(xy[i][1]))
        lib.sift_add_keypoint_to_list(k, lst)
    return lst


def scalespace_to_numpy(ss_ptr):
    """Copy a struct sift_scalespace into a list of (nSca,h,w) float32 arrays + metadata."""
    ss = ss_ptr.contents
    out = []
    for o in range(ss.nOct):
        oc = ss.octaves[o].contents
        n = oc.nSca * oc.w * oc.h
        stack = np.ctypeslib.as_array(oc.imStack, shape=(n,)).copy().reshape(oc.nSca, oc.h, oc.w)
        sig = np.ctypeslib.as_array(oc.sigmas, shape=(oc.nSca,)).copy()
        out.append({"delta": float(oc.delta), "w": oc.w, "h": oc.h, "nSca": oc.nSca,
                    "sigmas": sig, "stack": stack})
    return out


KEY_INT_FIELDS = ("o", "s", "i", "j")
KEY_FLT_FIELDS = ("x", "y", "sigma", "theta", "val", "edgeResp")


def keypoints_to_numpy(kp_ptr, with_descr=True):
    """Copy a struct sift_keypoints list into flat arrays (ints (n,4), floats (n,6), ...)."""
    kp = kp_ptr.contents
    n = kp.size
    ints = np.zeros((n, 4), np.int32)
    flts = np.zeros((n, 6), np.float32)
    descr = orih = None
    nd = nb = 0
    if n:
        k0 = kp.list[0].contents
        nd = k0.n_hist * k0.n_hist * k0.n_ori
        nb = k0.n_bins
    if with_descr:
        descr = np.zeros((n, nd), np.float32)
        orih = np.zeros((n, nb), np.float32)
    for k in range(n):
        key = kp.list[k].contents
        ints[k] = (key.o, key.s, key.i, key.j)
        flts[k] = (key.x, key.y, key.sigma, key.theta, key.val, key.edgeResp)
        if with_descr:
            descr[k] = np.ctypeslib.as_array(key.descr, shape=(nd,))
            orih[k] = np.ctypeslib.as_array(key.orihist, shape=(nb,))
    return {"ints": ints, "flts": flts, "descr": descr, "orihist": orih}
