"""GPU: the remaining reference-named entry points (flat lib_sift.h API, description-only
variants, stand-alone scale-space), called through ctypes exactly as C host code would."""
import ctypes as C
import os

import numpy as np
import pytest

import sift_amd
from sift_amd import abi, api, synth
from oracle import pyoracle, refrun

from common import same_bits

pytestmark = pytest.mark.gpu
P = C.POINTER


def _std_array(ptr, n):
    out = np.zeros(n, abi.KEYPOINT_STD_DTYPE)
    if n:
        C.memmove(out.ctypes.data, ptr, n * 144)
    return out


def test_sift_compute_features_flat_api(hip):
    img = synth.white(240, 180, 11)
    want = pyoracle.run(img, keep_ss=False)["keys"]
    got = api.sift_compute_features(img)
    assert len(got) == len(want["ints"])
    assert same_bits(np.stack([got["x"], got["y"], got["scale"], got["orientation"]], 1), want["flts"][:, :4])
    assert np.array_equal(got["descriptor"], want["descr"].astype(np.uint8))


def test_sift_compute_points_returns_unoriented_border_filtered_list(hip):
    img = synth.white(240, 180, 12)
    want = pyoracle.run(img, keep_ss=False)["stages"]["F"]
    n = C.c_int()
    p = hip.sift_compute_points(img.ctypes.data_as(abi.c_float_p), 240, 180, C.byref(n))
    got = _std_array(p, n.value)
    C.CDLL(None).free(C.cast(p, C.c_void_p))
    assert n.value == len(want["ints"])
    assert same_bits(np.stack([got["x"], got["y"], got["scale"]], 1), np.ascontiguousarray(want["flts"][:, :3]))
    assert not got["orientation"].any() and not got["descriptor"].any()


def test_without_description_variant(hip):
    img = synth.white(200, 150, 13)
    want = pyoracle.run(img)
    kk = (P(abi.SiftKeypoints) * 5)(*[hip.sift_malloc_keypoints() for _ in range(5)])
    ss = (P(abi.SiftScalespace) * 2)()
    k = hip.sift_anatomy_without_description(img.ctypes.data_as(abi.c_float_p), 200, 150,
                                             C.byref(abi.SiftParameters.default()), ss, kk)
    res = abi.keypoints_to_numpy(k, with_descr=False)
    assert np.array_equal(res["ints"], want["stages"]["F"]["ints"])
    assert same_bits(np.ascontiguousarray(res["flts"][:, [0, 1, 2, 4, 5]]),
                     np.ascontiguousarray(want["stages"]["F"]["flts"][:, [0, 1, 2, 4, 5]]))
    for name, lst in zip("ABCDE", kk):
        assert np.array_equal(abi.keypoints_to_numpy(lst, False)["ints"], want["stages"][name]["ints"])
    dog = abi.scalespace_to_numpy(ss[1])
    for o in range(len(dog)):
        assert same_bits(dog[o]["stack"], want["dog"][o])
    hip.sift_free_keypoints(k)
    for lst in kk:
        hip.sift_free_keypoints(lst)
    for s in ss:
        hip.sift_free_scalespace(s)


def _list_from(hip, rows, p):
    lst = hip.sift_malloc_keypoints()
    for (x, y, sigma, theta, o, s) in rows:
        key = hip.sift_malloc_keypoint(p.n_ori, p.n_hist, p.n_bins)
        kc = key.contents
        kc.x, kc.y, kc.sigma, kc.theta, kc.o, kc.s = x, y, sigma, theta, o, s
        hip.sift_add_keypoint_to_list(key, lst)
    return lst


def test_only_description_and_orientation_variants(hip):
    """Descriptors (and single orientations) for caller-supplied keypoints: feed the oracle's
    own keypoints back in and expect its descriptors."""
    img = synth.white(220, 160, 14)
    p = abi.SiftParameters.default()
    want = pyoracle.run(img, p, keep_ss=False)["keys"]
    rows = [(f[0], f[1], f[2], f[3], i[0], i[1]) for f, i in zip(want["flts"], want["ints"])]
    lst = _list_from(hip, rows, p)
    hip.sift_anatomy_only_description(img.ctypes.data_as(abi.c_float_p), 220, 160, C.byref(p), lst)
    got = abi.keypoints_to_numpy(lst)
    assert np.array_equal(got["descr"], want["descr"])
    hip.sift_free_keypoints(lst)

    # one orientation per keypoint: compare with the reference library when it is available
    if refrun.have_ref():
        ref = refrun.ref_lib()
        l_ref = _list_from(ref, rows, p)
        l_hip = _list_from(hip, rows, p)
        ref.sift_anatomy_orientation_and_description(img.ctypes.data_as(abi.c_float_p), 220, 160, C.byref(p), l_ref)
        hip.sift_anatomy_orientation_and_description(img.ctypes.data_as(abi.c_float_p), 220, 160, C.byref(p), l_hip)
        a, b = abi.keypoints_to_numpy(l_ref), abi.keypoints_to_numpy(l_hip)
        assert same_bits(np.ascontiguousarray(a["flts"][:, 3]), np.ascontiguousarray(b["flts"][:, 3]))
        assert np.array_equal(a["descr"], b["descr"]) and same_bits(a["orihist"], b["orihist"])
        ref.sift_free_keypoints(l_ref)
        hip.sift_free_keypoints(l_hip)


def test_standalone_scalespace_and_gradient(hip):
    img = synth.smooth(160, 120, 15)
    want = pyoracle.run(img)
    ss = hip.sift_malloc_scalespace_lowe(len(want["octaves"]), 3, 160, 120, C.c_float(0.5), C.c_float(0.8))
    hip.scalespace_compute(ss, img.ctypes.data_as(abi.c_float_p), 160, 120, C.c_float(0.5))
    g = abi.scalespace_to_numpy(ss)
    for o in range(len(g)):
        assert same_bits(g[o]["stack"], want["gss"][o])
    sx = hip.sift_malloc_scalespace_from_model(ss)
    sy = hip.sift_malloc_scalespace_from_model(ss)
    hip.scalespace_compute_gradient(ss, sx, sy)
    gx, gy = abi.scalespace_to_numpy(sx), abi.scalespace_to_numpy(sy)
    for o in range(len(g)):
        for s in range(g[o]["nSca"]):
            gi, gj = pyoracle.gradient(want["gss"][o][s])
            assert same_bits(gx[o]["stack"][s], gi) and same_bits(gy[o]["stack"][s], gj)
    for s in (ss, sx, sy):
        hip.sift_free_scalespace(s)


def test_keypoint_text_format_round_trip(hip, tmp_path):
    img = synth.white(200, 150, 16)
    got = api.sift_compute_features(img)
    path = str(tmp_path / "keys.txt").encode()
    hip.sift_write_to_file.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    hip.sift_write_to_file(path, got.ctypes.data, len(got))
    first = open(path).readline().split()
    assert len(first) == 4 + 128 and first[0] == "%f" % got["x"][0]
    hip.sift_read_from_file.restype = C.c_void_p
    hip.sift_read_from_file.argtypes = [C.c_char_p, P(C.c_int)]
    n = C.c_int()
    back = _std_array(hip.sift_read_from_file(path, C.byref(n)), n.value)
    assert n.value == len(got) and np.array_equal(back["descriptor"], got["descriptor"])
    assert np.allclose(back["x"], got["x"], atol=1e-6, rtol=1e-6)


def test_c_host_program_links_and_runs(hip, tmp_path):
    """Compiles a C host (tests/c_host/host_example.c) against include/sift_amd.h and the shared
    library with gcc and runs it: the drop-in boundary exercised from C, not from ctypes."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    img = synth.white(256, 192, 17)
    raw = str(tmp_path / "img.f32")
    img.tofile(raw)
    exe = str(tmp_path / "host_example")
    libdir = os.path.join(root, "sift_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", os.path.join(root, "tests", "c_host", "host_example.c"),
                           "-I", os.path.join(root, "include"), "-L", libdir, "-lsift_amd",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    out = str(tmp_path / "keys.txt")
    res = subprocess.run([exe, raw, "256", "192", out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    want = pyoracle.run(img)
    nums = [int(v) for v in res.stdout.split()]
    assert nums[0] == nums[1] == len(want["keys"]["ints"])
    assert nums[2:8] == [len(want["stages"][s]["ints"]) for s in "ABCDEF"]
    assert nums[8] == len(want["octaves"]) and nums[9] == 5
    lines = open(out).read().splitlines()
    assert len(lines) == len(want["keys"]["ints"])
    first = lines[0].split()
    assert first[:4] == ["%f" % v for v in want["keys"]["flts"][0, :4]]
    assert [int(v) for v in first[4:]] == want["keys"]["descr"][0].astype(int).tolist()


def test_c_matching_host_program(hip, tmp_path):
    """tests/c_host/match_example.c (the call sequence of match_cli.c) on two keypoint files
    written in the reference's text format: the printed pairs are those of the oracle."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    a = synth.white(200, 150, 31)
    b = np.roll(a, (3, 2), axis=(0, 1))
    files, descr, xy = [], [], []
    for k, img in enumerate((a, b)):
        keys = api.sift_compute_features(img)
        path = str(tmp_path / f"keys{k}.txt")
        n = len(keys)
        buf = keys.tobytes()
        hip.sift_write_to_file(path.encode(), C.cast(C.c_char_p(buf), P(abi.SiftKeypointStd)), n)
        files.append(path)
        descr.append(keys["descriptor"].astype(np.float32))
        xy.append(np.stack([keys["x"], keys["y"]], 1))
    exe = str(tmp_path / "match_example")
    libdir = os.path.join(root, "sift_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", os.path.join(root, "tests", "c_host", "match_example.c"),
                           "-I", os.path.join(root, "include"), "-L", libdir, "-lsift_amd",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    res = subprocess.run([exe, files[0], files[1]], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    want = pyoracle.matching(descr[0], descr[1], 0.6, 1)
    idx = np.nonzero(want["pass"])[0]
    n1, n2, nm = [int(v) for v in res.stderr.split()[-3:]]
    assert (n1, n2, nm) == (len(descr[0]), len(descr[1]), len(idx)) and nm > 10
    rows = [l.split() for l in res.stdout.splitlines() if l.strip()]
    assert len(rows) == nm
    for row, i in zip(rows, idx):
        # "x y sigma theta" of the query, then of its nearest neighbour (lib_keypoint.c:184-205)
        vals = [float(v) for v in row]
        assert vals[0] == float("%f" % xy[0][i, 0]) and vals[1] == float("%f" % xy[0][i, 1])
        j = want["iA"][i]
        assert vals[4] == float("%f" % xy[1][j, 0]) and vals[5] == float("%f" % xy[1][j, 1])


def test_concurrent_host_threads(hip):
    """The reference is re-entrant (no globals); the replacement must be too: four host threads
    call the reference-named sift_compute_features and a batched context of their own at the
    same time, on different images, and get what a single thread gets.  Every thread also calls
    the allocation-free matcher on a non-blocking stream of its own while the others capture
    their hipGraphs (its calls share one scratch array per GPU, so they take turns under a lock)."""
    import threading
    from test_matching import FIELDS, assert_same, descriptor_sets
    from test_zz_no_alloc import DTYPES, Hip
    w, h, n_threads, rounds = 240, 180, 4, 3
    imgs = [synth.make("white", w, h, seed=900 + t) for t in range(n_threads)]
    sets = [descriptor_sets(200 + 10 * t, 300, 128, 70 + t, "u8") for t in range(n_threads)]
    want_match = [api.match(a, b, 0.6, 1) for a, b in sets]
    scratch_lock = threading.Lock()

    def flat(img):
        return api.sift_compute_features(img)

    def batched(img):
        with sift_amd.Context(w, h, max_batch=2) as ctx:
            ctx.detect_host(np.stack([img, img]))
            return ctx.keypoints(1)[0].tobytes()

    def noalloc(t, hipm, stream):
        a, b = sets[t]
        da, db = hipm.upload(a), hipm.upload(b)
        out = {f: hipm.alloc(len(a) * np.dtype(DTYPES[f]).itemsize) for f in FIELDS}
        with scratch_lock:
            rc = hip.sift_amd_match_u8_noalloc(da, len(a), db, len(b), 128, 0.6, 1, *(out[f] for f in FIELDS), stream)
            assert rc == 0, hip.sift_amd_last_error().decode()
            assert hipm.rt.hipStreamSynchronize(stream) == 0
        res = {f: hipm.download(out[f], len(a), DTYPES[f]) for f in FIELDS}
        hipm.free_all()
        return res

    want = [(flat(im).tobytes(), batched(im)) for im in imgs]
    errors = []

    def worker(t):
        try:
            hipm = Hip(own_stream=True)
            stream = hipm.own
            for _ in range(rounds):
                assert flat(imgs[t]).tobytes() == want[t][0]
                assert batched(imgs[t]) == want[t][1]
                assert_same(noalloc(t, hipm, stream), want_match[t], f"thread {t}")
            hip.sift_amd_release_thread_cache()
        except BaseException as e:       # noqa: BLE001 - reported to the main thread
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert len(want[0][0]) > 144 * 20


def test_runs_on_streams_the_caller_destroys(hip):
    """A context keeps no handle of a caller's stream: run on stream A, destroy A, run on stream B
    (per-request streams of a serving loop), read the results, destroy the context."""
    from test_zz_no_alloc import Hip
    w, h = 200, 150
    img = synth.white(w, h, 3)
    hipm = Hip()
    d_img = hipm.upload(np.stack([img, img]))
    with sift_amd.Context(w, h, max_batch=2) as ref:
        ref.detect_host(np.stack([img, img]))
        want = ref.keypoints(1)[0].tobytes()
    ctx = sift_amd.Context(w, h, max_batch=2)
    for k in range(4):
        s = hipm.stream(non_blocking=(k % 2 == 0))
        ctx.detect_device(d_img, 2, s)
        if k % 2 == 0:
            assert hipm.rt.hipStreamSynchronize(s) == 0
        hipm.rt.hipStreamDestroy(s)       # with work of the run possibly still queued on it (odd k)
        if k >= 2:
            assert ctx.keypoints(1)[0].tobytes() == want
    s = hipm.stream(non_blocking=True)
    ctx.detect_device(d_img, 2, s)
    hipm.rt.hipStreamDestroy(s)
    ctx.close()                            # waits for the last run without the stream
    hipm.free_all()


def test_reference_named_calls_grow_their_lists_instead_of_aborting(hip, monkeypatch):
    """The reference's lists grow as needed.  With the default list capacities forced far too
    small, sift_anatomy(), sift_compute_features() and sift_compute_points() repeat the run with
    lists sized from the first run's counters and return the reference's result."""
    w, h = 200, 150
    img = synth.white(w, h, 31)
    want = pyoracle.run(img, keep_ss=False)
    monkeypatch.setenv("SIFT_AMD_CAP", "64")
    hip.sift_amd_release_thread_cache()
    with sift_amd.Context(w, h) as ctx:                     # the forced capacity does overflow
        ctx.detect_host(img)
        with pytest.raises(api.SiftAmdError):
            ctx.counts()
    got = refrun.run_sift_anatomy(hip, img, keep_ss=False)
    for name in "ABCDEF":
        assert np.array_equal(got["stages"][name]["ints"], want["stages"][name]["ints"]), name
    assert np.array_equal(got["keys"]["descr"], want["keys"]["descr"])
    assert same_bits(np.ascontiguousarray(got["keys"]["flts"][:, :4]), np.ascontiguousarray(want["keys"]["flts"][:, :4]))
    flat = api.sift_compute_features(img)
    assert np.array_equal(flat["descriptor"], want["keys"]["descr"].astype(np.uint8))
    n = C.c_int()
    p = hip.sift_compute_points(img.ctypes.data_as(abi.c_float_p), w, h, C.byref(n))
    C.CDLL(None).free(C.cast(p, C.c_void_p))
    assert n.value == len(want["stages"]["F"]["ints"])
    hip.sift_amd_release_thread_cache()


def test_alternating_reference_named_calls_keep_their_contexts(hip):
    """sift_anatomy, the flat API and scalespace_compute at one image size, in turn: each keeps
    its cached context (results stay right, and the second round is much faster than the first
    because no pool is rebuilt)."""
    import time
    w, h = 320, 240
    img = synth.white(w, h, 77)
    want = pyoracle.run(img, keep_ss=True)
    hip.sift_amd_release_thread_cache()
    times = []
    for rnd in range(3):
        t0 = time.perf_counter()
        got = refrun.run_sift_anatomy(hip, img, keep_ss=False)
        flat = api.sift_compute_features(img)
        ss = hip.sift_malloc_scalespace_lowe(len(want["octaves"]), 3, w, h, C.c_float(0.5), C.c_float(0.8))
        hip.scalespace_compute(ss, img.ctypes.data_as(abi.c_float_p), w, h, C.c_float(0.5))
        gss = abi.scalespace_to_numpy(ss)
        hip.sift_free_scalespace(ss)
        times.append(time.perf_counter() - t0)
        assert np.array_equal(got["keys"]["descr"], want["keys"]["descr"])
        assert np.array_equal(flat["descriptor"], want["keys"]["descr"].astype(np.uint8))
        for o in range(len(gss)):
            assert same_bits(gss[o]["stack"], want["gss"][o])
    print("rounds:", times)
    hip.sift_amd_release_thread_cache()


def test_scale_spaces_of_sift_anatomy_with_other_copy_thread_counts(hip):
    """The four scale-spaces sift_anatomy() hands back travel through pinned staging in 32 MB pieces
    that several host threads copy into place; one thread, and a count that does not divide the
    pieces evenly, must give the same bytes as the oracle (stacks large enough for two pieces)."""
    img = synth.make("white", 700, 500, 31)
    want = pyoracle.run(img, keep_ss=True)
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import sift_amd\n"
        "from sift_amd import synth\n"
        "from oracle import refrun\n"
        "got = refrun.run_sift_anatomy(sift_amd.lib(), synth.make('white', 700, 500, 31))\n"
        "for k in ('gss', 'dog', 'gx', 'gy'):\n"
        "    print(k, ' '.join(hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in got[k]))\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import hashlib
    import subprocess
    import sys
    expected = {"gss": want["gss"], "dog": want["dog"]}
    grads = []
    for n in ("1", "3"):
        env = dict(os.environ, SIFT_AMD_COPY_THREADS=n)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=200)
        assert out.returncode == 0, out.stderr[-1500:]
        lines = dict(l.split(" ", 1) for l in out.stdout.strip().splitlines() if l[:3] in ("gss", "dog", "gx ", "gy "))
        for k, stacks in expected.items():
            digests = [hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in stacks]
            assert lines[k].split() == digests, f"{k} differs with {n} copy thread(s)"
        grads.append((lines["gx"].split(), lines["gy"].split()))
    here = refrun.run_sift_anatomy(hip, img)          # this process: default thread count
    mine = tuple([hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in here[k]] for k in ("gx", "gy"))
    assert grads[0] == grads[1] == mine and len(mine[0]) == len(want["gss"])
