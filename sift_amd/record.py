"""Writer of the reference's `--record` directory, driven by the HIP library.

`sift_cli image --record dir` (sift_cli.c:300-349) leaves one directory holding every
intermediate of the pipeline; the reference's Python test-suite reads it back
(tests/test_sift_unittest.py:105-130, 179-216).  This module produces the same
directory from `libsift_amd.so`, so that those consumers can be pointed at the GPU
backend:

    <dir>/gss|dog|grad_x|grad_y/<stem>_oNNN_sNNN.f32 + <stem>_meta.json
                                                  (lib_io_scalespace.c:138-175)
    <dir>/extrema|contrast_pre|contrast_post|edge|border/extrema_{int.i32,float.f32,meta.json}
                                                  (sift_cli.c:382-440)
    <dir>/refined/extrema_refined_{int.i32,float.f32,meta.json}   (sift_cli.c:442-492)
    <dir>/keys/keys_{int.i32,float.f32,desc.u8,meta.json}         (sift_cli.c:494-582)

Two quirks of the reference are kept on purpose: `grad_x` holds the fourth scale-space
returned by sift_anatomy and `grad_y` the third (sift_cli.c:315,320), and descriptors
are serialised with their two spatial indices swapped (sift_cli.c:535-545).

Command line (options as sift_cli.c:98-119):

    python -m sift_amd.record image.png --record out_dir [-ss_noct 8] [-thresh_dog 0.0133] ...

`image` may be a PNG (decoded with PIL, gray conversion of io_png.c:499-516), a `.npy`
float array in [0,1), or a raw float32 file given with `-size WxH`.
"""
import ctypes as C
import json
import os
import sys

import numpy as np

from . import abi

SCALESPACE_DIRS = (("gss", 0), ("dog", 1), ("grad_x", 3), ("grad_y", 2))
STAGE_DIRS = (("extrema", 0, "extrema"), ("contrast_pre", 1, "extrema"),
              ("refined", 2, "extrema_refined"), ("contrast_post", 3, "extrema"),
              ("edge", 4, "extrema"), ("border", 5, "extrema"))

# sift_cli.c:177-227: option name -> (struct field, converter)
CLI_OPTIONS = {
    "ss_noct": ("n_oct", int), "ss_nspo": ("n_spo", int), "ss_dmin": ("delta_min", float),
    "ss_smin": ("sigma_min", float), "ss_sin": ("sigma_in", float),
    "thresh_dog": ("C_DoG", float), "thresh_edge": ("C_edge", float),
    "ori_nbins": ("n_bins", int), "ori_thresh": ("t", float), "ori_lambda": ("lambda_ori", float),
    "descr_nhist": ("n_hist", int), "descr_nori": ("n_ori", int),
    "descr_lambda": ("lambda_descr", float),
}


def _g9(v):
    return "%.9g" % float(v)


def write_scalespace(octaves, out_dir, stem):
    """octaves: list of {"w","h","delta","sigmas","stack"(nSca,h,w) float32}."""
    os.makedirs(out_dir, exist_ok=True)
    lines = []
    for o, oc in enumerate(octaves):
        names = []
        for s in range(len(oc["sigmas"])):
            name = "%s_o%03d_s%03d.f32" % (stem, o, s)
            np.ascontiguousarray(oc["stack"][s], "<f4").tofile(os.path.join(out_dir, name))
            names.append(name)
        lines.append('    {"o": %d, "w": %d, "h": %d, "delta": %s, "sigmas": [%s], "files": [%s] }'
                     % (o, oc["w"], oc["h"], _g9(oc["delta"]),
                        ", ".join(_g9(v) for v in oc["sigmas"]),
                        ", ".join('"%s"' % n for n in names)))
    with open(os.path.join(out_dir, stem + "_meta.json"), "w") as f:
        f.write('{\n  "n_octaves": %d,\n  "stem": "%s",\n  "dir": %s,\n  "octaves": [\n'
                % (len(octaves), stem, json.dumps(out_dir)))
        f.write(",\n".join(lines) + ("\n" if lines else ""))
        f.write("  ]\n}\n")


def _write_list(out_dir, prefix, ints, floats, float_order, descr=None, desc_len=0):
    os.makedirs(out_dir, exist_ok=True)
    np.ascontiguousarray(ints, "<i4").tofile(os.path.join(out_dir, prefix + "_int.i32"))
    np.ascontiguousarray(floats, "<f4").tofile(os.path.join(out_dir, prefix + "_float.f32"))
    meta = ['  "count": %d' % len(ints),
            '  "int_file": "%s_int.i32"' % prefix,
            '  "float_file": "%s_float.f32"' % prefix]
    if descr is not None:
        np.ascontiguousarray(descr, np.uint8).tofile(os.path.join(out_dir, prefix + "_desc.u8"))
        meta += ['  "desc_file": "%s_desc.u8"' % prefix, '  "desc_len": %d' % desc_len]
    meta += ['  "int_order": ["o", "s", "i", "j"]',
             '  "float_order": [%s]' % ", ".join('"%s"' % n for n in float_order)]
    with open(os.path.join(out_dir, prefix + "_meta.json"), "w") as f:
        f.write("{\n" + ",\n".join(meta) + "\n}\n")


def write_stage(lst, out_dir, prefix):
    """lst: {"ints": (n,4) o,s,i,j ; "flts": (n,6) x,y,sigma,theta,val,edgeResp}."""
    f = lst["flts"]
    _write_list(out_dir, prefix, lst["ints"], f[:, [0, 1, 2, 4]], ("y", "x", "sigma", "val"))


def serialise_descriptors(descr, n_hist, n_ori):
    """float descriptors (n, n_hist*n_hist*n_ori) -> u8 with the spatial indices swapped,
    clamped to 0..255 and rounded by +0.5 then truncation (sift_cli.c:535-545)."""
    d = np.asarray(descr, np.float32).reshape(-1, n_hist, n_hist, n_ori).transpose(0, 2, 1, 3)
    d = np.clip(d, np.float32(0), np.float32(255)) + np.float32(0.5)
    return d.astype(np.uint8).reshape(len(d), -1)


def write_keys(lst, out_dir, n_hist, n_ori):
    f = lst["flts"]
    n = len(lst["ints"])
    nd = n_hist * n_hist * n_ori if n else 0
    d = serialise_descriptors(lst["descr"], n_hist, n_ori) if n else np.zeros((0, 0), np.uint8)
    _write_list(out_dir, "keys", lst["ints"], f[:, [0, 1, 2, 3]], ("y", "x", "sigma", "theta"),
                descr=d, desc_len=nd)


def compute(lib, img, params):
    """Run `sift_anatomy` of `lib` as sift_cli.c:285-292 does; returns (scalespaces[4],
    stage lists[6], keys) as numpy structures."""
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape
    P = C.POINTER
    kk = (P(abi.SiftKeypoints) * 6)(*[lib.sift_malloc_keypoints() for _ in range(6)])
    ss = (P(abi.SiftScalespace) * 4)()
    k = lib.sift_anatomy(img.ctypes.data_as(abi.c_float_p), w, h, C.byref(params), ss, kk)
    try:
        spaces = [abi.scalespace_to_numpy(s) for s in ss]
        stages = [abi.keypoints_to_numpy(l, with_descr=False) for l in kk]
        keys = abi.keypoints_to_numpy(k, with_descr=True)
    finally:
        lib.sift_free_keypoints(k)
        for l in kk:
            lib.sift_free_keypoints(l)
        for s in ss:
            lib.sift_free_scalespace(s)
    return spaces, stages, keys


def write_record(lib, img, out_dir, params=None):
    """Everything `sift_cli img --record out_dir` writes, computed by `lib`."""
    p = params if params is not None else abi.SiftParameters.default()
    spaces, stages, keys = compute(lib, img, p)
    os.makedirs(out_dir, exist_ok=True)
    for name, idx in SCALESPACE_DIRS:
        write_scalespace(spaces[idx], os.path.join(out_dir, name), name)
    for name, idx, prefix in STAGE_DIRS:
        write_stage(stages[idx], os.path.join(out_dir, name), prefix)
    write_keys(keys, os.path.join(out_dir, "keys"), p.n_hist, p.n_ori)
    return {"n_keys": len(keys["ints"]), "n_octaves": len(spaces[0])}


def gray_from_u8(a):
    """u8 image (h,w) or (h,w,3|4) -> float32 gray in [0,1): the reference's loader
    (io_png.c:499-516, alpha stripped) followed by the division of sift_cli.c:280-281."""
    a = np.asarray(a)
    if a.ndim == 3 and a.shape[2] in (2, 4):
        a = a[:, :, :-1]
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[:, :, 0]
    if a.ndim == 2:
        g = a.astype(np.float32)
    else:
        r, g_, b = (a[:, :, c].astype(np.float32).astype(np.float64) for c in range(3))
        g = (0.212639005871510 * r + 0.715168678767756 * g_ + 0.072192315360734 * b
             ).astype(np.float32)
    return (g.astype(np.float64) / 256.0).astype(np.float32)


def load_image(path, size=None):
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    if size is not None:
        w, h = size
        return np.fromfile(path, "<f4").reshape(h, w)
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("L", "RGB", "RGBA", "LA"):
        im = im.convert("RGB")
    return gray_from_u8(np.asarray(im))


def parse_cli(argv):
    """Same grammar as sift_cli.c:130-247: `-name value` pairs anywhere, one positional image."""
    p = abi.SiftParameters.default()
    out_dir, size, rest = "", None, []
    i = 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("-"):
            name = a[1:]
            if i + 1 >= len(argv) or argv[i + 1].startswith("-"):
                raise SystemExit("Fatal error: option %s requires an argument." % name)
            v = argv[i + 1]
            if name == "-record":
                out_dir = v
            elif name == "size":
                size = tuple(int(t) for t in v.lower().split("x"))
            elif name in CLI_OPTIONS:
                field, conv = CLI_OPTIONS[name]
                setattr(p, field, conv(v))
            else:
                raise SystemExit('Fatal error: option "-%s" is unknown.' % name)
            i += 2
        else:
            rest.append(a)
            i += 1
    if len(rest) != 1:
        raise SystemExit("Usage:  python -m sift_amd.record image [options...] --record dir")
    return rest[0], p, out_dir, size


def main(argv=None):
    from .api import lib
    path, p, out_dir, size = parse_cli(list(sys.argv[1:] if argv is None else argv))
    img = load_image(path, size)
    L = lib()
    if out_dir:
        info = write_record(L, img, out_dir, p)
        print("%d keypoints, %d octaves -> %s" % (info["n_keys"], info["n_octaves"], out_dir))
    else:
        _, _, keys = compute(L, img, p)
        for f, d in zip(keys["flts"], keys["descr"]):
            print("%.3f %.3f %.3f %.3f " % tuple(f[:4]) + " ".join("%d" % int(v) for v in d))
    return 0


if __name__ == "__main__":
    sys.exit(main())
