"""ctypes front-end of libsift_amd.so.

Two layers, both thin:
  * `Context` wraps the batched interface (Part 2 of include/sift_amd.h);
  * `lib()` returns the raw library with the reference-named entry points bound, so code
    written against lib_sift_anatomy.h / lib_sift.h can call them unchanged.
There is no CPU fallback: if the HIP library is missing or no GPU is visible, calls fail.
"""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
# SIFT_AMD_LIB: another build of the same library (A/B runs of the tools); never a fallback
LIB_PATH = os.environ.get("SIFT_AMD_LIB") or os.path.join(HERE, "libsift_amd.so")

MAX_OCT, MAX_SCA = 16, 12
K_OVERSAMPLE, K_SUBSAMPLE, K_BLUR_V, K_BLUR_H, K_EXTREMA, K_REFINE, K_RANK, K_ORIENTATION, \
    K_EXPAND, K_DESCRIPTOR, K_BLUR_FUSED, K_COUNT = range(12)
KERNEL_CLASSES = ["oversample", "subsample", "blur_v", "blur_h", "extrema", "refine", "rank",
                  "orientation", "expand", "descriptor", "blur_fused"]
KEEP_ALL_EXTREMA = 1
KEEP_SCALESPACE = 2


class Geometry(C.Structure):
    _fields_ = [("n_oct", C.c_int), ("n_sca", C.c_int),
                ("w", C.c_int * MAX_OCT), ("h", C.c_int * MAX_OCT),
                ("delta", C.c_float * MAX_OCT),
                ("sigmas", (C.c_float * MAX_SCA) * MAX_OCT),
                ("blur_radius", (C.c_int * MAX_SCA) * MAX_OCT),
                ("thresh_pre", C.c_float), ("thresh", C.c_float), ("thresh_edge", C.c_float)]


KEY_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("sigma", "<f4"), ("theta", "<f4"),
                      ("o", "<i4"), ("s", "<i4"), ("i", "<i4"), ("j", "<i4"),
                      ("val", "<f4"), ("edge", "<f4")])
assert KEY_DTYPE.itemsize == 40


class SiftAmdError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libsift_amd.so and declare every entry point.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SiftAmdError(f"{LIB_PATH} not found: build it with `python -m sift_amd.build` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    abi.bind_reference_api(L)
    P = C.POINTER
    vp = C.c_void_p
    L.sift_amd_last_error.restype = C.c_char_p
    L.sift_amd_device_count.restype = C.c_int
    L.sift_amd_create.restype = vp
    L.sift_amd_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P(abi.SiftParameters), C.c_int]
    L.sift_amd_destroy.restype = None
    L.sift_amd_destroy.argtypes = [vp]
    L.sift_amd_get_geometry.argtypes = [vp, P(Geometry)]
    L.sift_amd_detect_device.argtypes = [vp, vp, C.c_int, vp]
    L.sift_amd_detect_host.argtypes = [vp, abi.c_float_p, C.c_int]
    L.sift_amd_scalespace_device.argtypes = [vp, vp, C.c_int, vp]
    L.sift_amd_keypoint_counts.argtypes = [vp, abi.c_int_p]
    L.sift_amd_fetch_keypoints.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.sift_amd_gather_keypoints.argtypes = [vp, P(vp), abi.c_int_p]
    L.sift_amd_gather_keypoints_into.argtypes = [vp, vp, C.c_int, abi.c_int_p]
    L.sift_amd_copy_ceiling.argtypes = [C.c_int, C.c_size_t, C.c_int, P(C.c_double)]
    L.sift_amd_valu_ceiling.argtypes = [C.c_int, C.c_int, P(C.c_double)]
    L.sift_amd_profile_read_blur.argtypes = [vp, C.c_int, C.c_int, P(C.c_double), P(C.c_longlong), C.c_int]
    L.sift_amd_stage_counts.argtypes = [vp, C.c_int, abi.c_int_p, abi.c_int_p]
    L.sift_amd_fetch_candidates.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.sift_amd_copy_scalespace.argtypes = [vp, C.c_int, C.c_int, C.c_int, abi.c_float_p]
    L.sift_amd_fetch_extras.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.sift_amd_fetch_stage_f_histograms.argtypes = [vp, C.c_int, vp, C.c_int]
    L.sift_amd_profile_enable.argtypes = [vp, C.c_uint]
    L.sift_amd_profile_read.argtypes = [vp, C.c_int, C.c_int, P(C.c_double), P(C.c_longlong), C.c_int]
    L.sift_amd_detect_device_u8.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.sift_amd_detect_host_u8.argtypes = [vp, vp, C.c_int, C.c_int]
    L.sift_amd_copy_ingested.argtypes = [vp, C.c_int, abi.c_float_p]
    match_out = [abi.c_int_p, abi.c_int_p, abi.c_float_p, abi.c_float_p, P(C.c_ubyte)]
    L.sift_amd_match_u8.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int] + match_out
    L.sift_amd_match_f32.argtypes = [C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int] + match_out
    L.sift_amd_match_images.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int] + match_out + [C.c_int]
    L.sift_amd_match_u8_noalloc.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int,
                                            vp, vp, vp, vp, vp, vp]
    L.sift_amd_release_thread_cache.restype = None
    L.sift_amd_launch_stats.argtypes = [P(C.c_longlong), C.c_int]
    _lib = L
    return L


def _err(what):
    return SiftAmdError(f"{what}: {lib().sift_amd_last_error().decode()}")


def device_count():
    return lib().sift_amd_device_count()


LAUNCH_STATS = ["pair_seed", "pair", "pair_strip_max", "wide_ahead2", "wide_ahead2_strip_max",
                "wide_other", "narrow", "sparse"]


def launch_stats(reset=False):
    """Routes the blur launches of this process took since the last reset (dict name -> count)."""
    out = (C.c_longlong * len(LAUNCH_STATS))()
    lib().sift_amd_launch_stats(out, int(reset))
    return dict(zip(LAUNCH_STATS, out[:]))


def valu_ceiling(device=0, reps=3):
    """1e9 lane-operations/s of non-fused float32 multiplies and adds on `device`."""
    out = C.c_double()
    if lib().sift_amd_valu_ceiling(device, reps, C.byref(out)):
        raise _err("sift_amd_valu_ceiling")
    return out.value


def copy_ceiling(device=0, nbytes=2 << 30, reps=5):
    """GB/s (read + written) of a plain copy kernel on `device`: a practical HBM ceiling."""
    out = C.c_double()
    if lib().sift_amd_copy_ceiling(device, nbytes, reps, C.byref(out)):
        raise _err("sift_amd_copy_ceiling")
    return out.value


class Context:
    """Pyramid pool + pipeline for up to `max_batch` images of w x h on one GPU."""

    def __init__(self, w, h, max_batch=1, params=None, device=0, keep_all_extrema=False,
                 keep_scalespace=False):
        self.L = lib()
        self.w, self.h, self.max_batch = w, h, max_batch
        self.params = params if params is not None else abi.SiftParameters.default()
        flags = (KEEP_ALL_EXTREMA if keep_all_extrema else 0) | (KEEP_SCALESPACE if keep_scalespace else 0)
        self.ptr = self.L.sift_amd_create(device, w, h, max_batch, C.byref(self.params), flags)
        if not self.ptr:
            raise _err("sift_amd_create")
        g = Geometry()
        self.L.sift_amd_get_geometry(self.ptr, C.byref(g))
        self.geometry = g
        self.n_oct, self.n_sca = g.n_oct, g.n_sca
        self.dims = [(g.w[o], g.h[o]) for o in range(g.n_oct)]
        self.last_n = 0

    def close(self):
        if self.ptr:
            self.L.sift_amd_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- running
    def detect_device(self, d_ptr, n, stream=None):
        if self.L.sift_amd_detect_device(self.ptr, d_ptr, n, stream):
            raise _err("sift_amd_detect_device")
        self.last_n = n

    def scalespace_device(self, d_ptr, n, stream=None):
        if self.L.sift_amd_scalespace_device(self.ptr, d_ptr, n, stream):
            raise _err("sift_amd_scalespace_device")
        self.last_n = n

    def detect_host(self, images):
        images = np.ascontiguousarray(images, np.float32)
        if images.ndim == 2:
            images = images[None]
        n, h, w = images.shape
        if (w, h) != (self.w, self.h):
            raise ValueError(f"image size {w}x{h} does not match the context ({self.w}x{self.h})")
        if self.L.sift_amd_detect_host(self.ptr, images.ctypes.data_as(abi.c_float_p), n):
            raise _err("sift_amd_detect_host")
        self.last_n = n

    def detect_device_u8(self, d_ptr, n, channels, stream=None):
        if self.L.sift_amd_detect_device_u8(self.ptr, d_ptr, n, channels, stream):
            raise _err("sift_amd_detect_device_u8")
        self.last_n = n

    def detect_host_u8(self, images):
        """images: uint8 (n,h,w) gray or (n,h,w,3|4) interleaved colour."""
        images = np.ascontiguousarray(images, np.uint8)
        ch = 1 if images.ndim == 3 else images.shape[3]
        n, h, w = images.shape[:3]
        if (w, h) != (self.w, self.h):
            raise ValueError(f"image size {w}x{h} does not match the context ({self.w}x{self.h})")
        if self.L.sift_amd_detect_host_u8(self.ptr, images.ctypes.data, n, ch):
            raise _err("sift_amd_detect_host_u8")
        self.last_n = n

    def ingested(self, n=None):
        n = self.last_n if n is None else n
        out = np.empty((n, self.h, self.w), np.float32)
        if self.L.sift_amd_copy_ingested(self.ptr, n, out.ctypes.data_as(abi.c_float_p)):
            raise _err("sift_amd_copy_ingested")
        return out

    def match(self, img_a, img_b, thresh=0.6, flag=1):
        """Two nearest neighbours in image b of every keypoint of image a (last run)."""
        n = int(self.counts()[img_a])
        out = _match_buffers(n)
        m = self.L.sift_amd_match_images(self.ptr, img_a, img_b, thresh, flag, *_match_args(out), max(n, 1))
        if m < 0:
            raise _err("sift_amd_match_images")
        return out

    # ---- results
    def counts(self):
        out = (C.c_int * max(self.last_n, 1))()
        if self.L.sift_amd_keypoint_counts(self.ptr, out):
            raise _err("sift_amd_keypoint_counts")
        return np.array(out[:self.last_n], np.int64)

    def keypoints(self, img=0):
        """(std records (n,) KEYPOINT_STD_DTYPE, key records (n,) KEY_DTYPE) of one image."""
        n = int(self.counts()[img])
        std = np.zeros(n, abi.KEYPOINT_STD_DTYPE)
        keys = np.zeros(n, KEY_DTYPE)
        if n:
            got = self.L.sift_amd_fetch_keypoints(self.ptr, img, std.ctypes.data, keys.ctypes.data, n)
            if got != n:
                raise _err("sift_amd_fetch_keypoints")
        return std, keys

    def gather(self):
        """Host gather of the whole batch: (concatenated std records, per-image counts)."""
        out = C.c_void_p()
        cnt = (C.c_int * max(self.last_n, 1))()
        if self.L.sift_amd_gather_keypoints(self.ptr, C.byref(out), cnt):
            raise _err("sift_amd_gather_keypoints")
        counts = np.array(cnt[:self.last_n], np.int64)
        total = int(counts.sum())
        arr = np.zeros(total, abi.KEYPOINT_STD_DTYPE)
        if total:
            C.memmove(arr.ctypes.data, out.value, total * 144)
        C.CDLL(None).free(out)
        return arr, counts

    def gather_into(self, buf):
        """Host gather into a caller-owned array of 144-byte records (e.g. pinned memory).
        Returns (view of the filled part, per-image counts)."""
        cnt = (C.c_int * max(self.last_n, 1))()
        total = self.L.sift_amd_gather_keypoints_into(self.ptr, buf.ctypes.data, len(buf), cnt)
        if total < 0:
            raise _err("sift_amd_gather_keypoints_into")
        return buf[:total], np.array(cnt[:self.last_n], np.int64)

    def extras(self, img=0):
        n = int(self.counts()[img])
        nb = self.params.n_bins
        nd = self.params.n_hist ** 2 * self.params.n_ori
        orih = np.zeros((n, nb), np.float32)
        descr = np.zeros((n, nd), np.float32)
        src = np.zeros(n, np.int32)
        if n:
            got = self.L.sift_amd_fetch_extras(self.ptr, img, orih.ctypes.data, descr.ctypes.data,
                                               src.ctypes.data, n)
            if got != n:
                raise _err("sift_amd_fetch_extras")
        return orih, descr, src

    def stage_counts(self, img=0):
        sizes = (C.c_int * 6)()
        ext = (C.c_int * MAX_OCT)()
        if self.L.sift_amd_stage_counts(self.ptr, img, sizes, ext):
            raise _err("sift_amd_stage_counts")
        return np.array(sizes[:], np.int64), np.array(ext[:self.n_oct], np.int64)

    def candidates(self, img=0):
        n = self.L.sift_amd_fetch_candidates(self.ptr, img, None, None, None, 0)
        if n < 0:
            raise _err("sift_amd_fetch_candidates")
        seed = np.zeros(n, KEY_DTYPE)
        ref = np.zeros(n, KEY_DTYPE)
        stage = np.zeros(n, np.int32)
        if n:
            got = self.L.sift_amd_fetch_candidates(self.ptr, img, seed.ctypes.data, ref.ctypes.data,
                                                   stage.ctypes.data, n)
            if got != n:
                raise _err("sift_amd_fetch_candidates")
        return seed, ref, stage

    def scalespace(self, img, which, octave):
        """(n_images, h, w) float32 copy of the Gaussian (which=0) or DoG (which=1) stack."""
        w, h = self.dims[octave]
        nimg = self.n_sca if which == 0 else self.n_sca - 1
        out = np.empty((nimg, h, w), np.float32)
        if self.L.sift_amd_copy_scalespace(self.ptr, img, which, octave,
                                           out.ctypes.data_as(abi.c_float_p)):
            raise _err("sift_amd_copy_scalespace")
        return out

    # ---- event timers
    def profile_enable(self, classes):
        mask = 0
        for c in classes:
            mask |= 1 << c
        self.L.sift_amd_profile_enable(self.ptr, mask)

    def profile_read(self, cls, octave=-1, reset=False):
        ms = C.c_double()
        n = C.c_longlong()
        if self.L.sift_amd_profile_read(self.ptr, cls, octave, C.byref(ms), C.byref(n), int(reset)):
            raise _err("sift_amd_profile_read")
        return ms.value, n.value

    def profile_read_blur(self, octave, scale, reset=False):
        ms = C.c_double()
        n = C.c_longlong()
        if self.L.sift_amd_profile_read_blur(self.ptr, octave, scale, C.byref(ms), C.byref(n), int(reset)):
            raise _err("sift_amd_profile_read_blur")
        return ms.value, n.value

    # ---- sizes of the work (for roofline accounting)
    def pixels_per_octave(self):
        return [w * h for (w, h) in self.dims]


def _match_buffers(n):
    return {"iA": np.zeros(n, np.int32), "iB": np.zeros(n, np.int32), "dA": np.zeros(n, np.float32),
            "dB": np.zeros(n, np.float32), "pass": np.zeros(n, np.uint8)}


def _match_args(o):
    return (o["iA"].ctypes.data_as(abi.c_int_p), o["iB"].ctypes.data_as(abi.c_int_p),
            o["dA"].ctypes.data_as(abi.c_float_p), o["dB"].ctypes.data_as(abi.c_float_p),
            o["pass"].ctypes.data_as(C.POINTER(C.c_ubyte)))


def match(desc1, desc2, thresh=0.6, flag=1, device=0):
    """lib_matching.c:92-135 on the GPU.  uint8 descriptors use the integer (matrix-core) path,
    anything else the float path.  Returns per-query iA, iB, dA, dB, pass."""
    a, b = np.asarray(desc1), np.asarray(desc2)
    u8 = a.dtype == np.uint8 and b.dtype == np.uint8
    a = np.ascontiguousarray(a, np.uint8 if u8 else np.float32)
    b = np.ascontiguousarray(b, np.uint8 if u8 else np.float32)
    if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("descriptors must be (n1, dim) and (n2, dim)")
    out = _match_buffers(len(a))
    fn = lib().sift_amd_match_u8 if u8 else lib().sift_amd_match_f32
    m = fn(device, a.ctypes.data, len(a), b.ctypes.data, len(b), a.shape[1], thresh, flag, *_match_args(out))
    if m < 0:
        raise _err("sift_amd_match")
    return out


def sift_compute_features(img):
    """lib_sift.h sift_compute_features on one image -> structured array of 144-byte records."""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    n = C.c_int()
    L = lib()
    p = L.sift_compute_features(img.ctypes.data_as(abi.c_float_p), w, h, C.byref(n))
    out = np.zeros(n.value, abi.KEYPOINT_STD_DTYPE)
    if n.value:
        C.memmove(out.ctypes.data, p, n.value * 144)
    C.CDLL(None).free(C.cast(p, C.c_void_p))
    return out
