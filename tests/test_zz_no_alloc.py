"""sift_amd_match_u8_noalloc (device-resident inputs/outputs, scratch in a fixed array of the
code object) against sift_amd_match_u8 (which allocates its scratch per call): bit-identical
indices, distances and pass flags, on two calls in a row, the second with a smaller input that
leaves stale bytes of the first call in the shared scratch."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle

from test_matching import FIELDS, assert_same, descriptor_sets, with_ties


class Hip:
    """Device memory through the HIP runtime the library itself is linked against (loading a
    second HIP runtime into the process, as importing torch after the library would, leaves the
    second one without GPUs)."""

    def __init__(self, own_stream=False):
        """own_stream: copies and fills go through a non-blocking stream of this object instead of
        the legacy default stream (which fails while another host thread captures a hipGraph)."""
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.rt.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        self.rt.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.rt.hipStreamDestroy.argtypes = [C.c_void_p]
        self.rt.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        self.rt.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        assert self.rt.hipSetDevice(0) == 0
        self.bufs = []
        self.own = self.stream(non_blocking=True) if own_stream else None

    def alloc(self, nbytes, fill=0xA5):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        if self.own:
            assert self.rt.hipMemsetAsync(p, fill, max(nbytes, 16), self.own) == 0
            assert self.rt.hipStreamSynchronize(self.own) == 0
        else:
            assert self.rt.hipMemset(p, fill, max(nbytes, 16)) == 0
        self.bufs.append(p)
        return p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        self._copy(p, arr.ctypes.data, arr.nbytes, 1)
        return p

    def _copy(self, dst, src, nbytes, kind):
        if self.own:
            assert self.rt.hipMemcpyAsync(dst, src, nbytes, kind, self.own) == 0
            assert self.rt.hipStreamSynchronize(self.own) == 0
        else:
            assert self.rt.hipMemcpy(dst, src, nbytes, kind) == 0

    def download(self, p, n, dtype):
        out = np.empty(n, dtype)
        if n:
            self._copy(out.ctypes.data, p, out.nbytes, 2)
        return out

    def stream(self, non_blocking=False):
        s = C.c_void_p()
        if non_blocking:
            assert self.rt.hipStreamCreateWithFlags(C.byref(s), 1) == 0      # hipStreamNonBlocking
        else:
            assert self.rt.hipStreamCreate(C.byref(s)) == 0
        return s

    def sync(self):
        assert self.rt.hipDeviceSynchronize() == 0

    def free_all(self):
        for p in self.bufs:
            self.rt.hipFree(p)
        self.bufs = []


DTYPES = {"iA": np.int32, "iB": np.int32, "dA": np.float32, "dB": np.float32, "pass": np.uint8}


def _noalloc(L, hipm, a, b, thresh, flag, stream=None):
    n1 = len(a)
    da, db = hipm.upload(a), hipm.upload(b)
    out = {f: hipm.alloc(n1 * np.dtype(DTYPES[f]).itemsize) for f in FIELDS}    # pre-filled with 0xA5
    hipm.sync()
    rc = L.sift_amd_match_u8_noalloc(da, n1, db, len(b), a.shape[1], thresh, flag, *(out[f] for f in FIELDS), stream)
    assert rc == 0, L.sift_amd_last_error().decode()
    hipm.sync()
    res = {f: hipm.download(out[f], n1, DTYPES[f]) for f in FIELDS}
    hipm.free_all()
    return res


@pytest.mark.gpu
def test_noalloc_matching_equals_allocating_entry_point(hip):
    from sift_amd import api
    hipm = Hip()
    s = hipm.stream()
    cases = [(3000, 3500, 128, 0.6, 1, None),      # first call: large
             (130, 67, 128, 0.9, 1, None),          # second call: smaller, stale scratch around it
             (60, 80, 72, 200.0, 0, s),             # other length / flag, caller's stream
             (8192, 8192, 128, 0.6, 1, None),       # the stated maximum
             (33, 2, 256, 1.1, 1, None)]
    for n1, n2, dim, thresh, flag, stream in cases:
        a, b = with_ties(*descriptor_sets(n1, n2, dim, 11 * n1 + dim, "u8"))
        want = api.match(a, b, thresh, flag)
        got = _noalloc(hip, hipm, a, b, thresh, flag, stream)
        assert_same(got, want, f"noalloc {n1}x{n2}x{dim}")
        if n1 <= 3000:         # and directly against the CPU oracle (the 8192 x 8192 case takes it minutes)
            assert_same(got, pyoracle.matching(a.astype(np.float32), b.astype(np.float32), thresh, flag),
                        f"noalloc vs oracle {n1}x{n2}x{dim}")
        assert n1 < 1000 or want["pass"].sum() > 100
    hipm.rt.hipStreamDestroy(s)


@pytest.mark.gpu
def test_noalloc_matching_refuses_what_does_not_fit(hip):
    hipm = Hip()
    d = hipm.alloc(64)
    o = hipm.alloc(64)
    args = (o,) * 5 + (None,)
    assert hip.sift_amd_match_u8_noalloc(d, 8193, d, 100, 128, 0.6, 1, *args) < 0
    assert b"8192" in hip.sift_amd_last_error()
    assert hip.sift_amd_match_u8_noalloc(d, 10, d, 8193, 128, 0.6, 1, *args) < 0
    assert hip.sift_amd_match_u8_noalloc(d, 10, d, 100, 257, 0.6, 1, *args) < 0
    assert hip.sift_amd_match_u8_noalloc(d, 10, d, 1, 128, 0.6, 1, *args) < 0
    assert hip.sift_amd_match_u8_noalloc(d, 0, d, 100, 128, 0.6, 1, *args) == 0
    hipm.sync()
    hipm.free_all()
