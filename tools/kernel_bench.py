#!/usr/bin/env python3
"""Per-kernel-class event timings of the pipeline (library timers) for quick A/B runs.

    python tools/kernel_bench.py [--w 1920 --h 1080 --batch 16 --iters 10 --kind smooth]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--w", type=int, default=1920)
    ap.add_argument("--h", type=int, default=1080)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--kind", default="smooth")
    ap.add_argument("--sigma-min", type=float, default=0.8)
    ap.add_argument("--sigma-in", type=float, default=0.5)
    ap.add_argument("--roofline", action="store_true", help="print the per-kernel HBM roofline table")
    ap.add_argument("--scalespace-only", action="store_true", help="run only the scale-space launches")
    a = ap.parse_args()
    imgs = np.stack([synth.make(a.kind, a.w, a.h, 12345 + b) for b in range(min(a.batch, 4))])
    imgs = np.concatenate([imgs] * ((a.batch + len(imgs) - 1) // len(imgs)))[:a.batch]
    d = torch.from_numpy(imgs).cuda()
    from sift_amd import abi
    params = abi.SiftParameters.default().copy(sigma_min=a.sigma_min, sigma_in=a.sigma_in)
    ctx = sift_amd.Context(a.w, a.h, a.batch, params)
    g = ctx.geometry
    print("octave-0 blur radii:", [g.blur_radius[0][s] for s in range(ctx.n_sca)])
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        ctx.detect_device(d.data_ptr(), a.batch, st)
        ctx.counts()
    ctx.profile_enable(list(range(api.K_COUNT)))
    for c in range(api.K_COUNT):
        ctx.profile_read(c, -1, reset=True)
    for _ in range(a.iters):
        if a.scalespace_only:
            ctx.scalespace_device(d.data_ptr(), a.batch, st)
            torch.cuda.synchronize()
        else:
            ctx.detect_device(d.data_ptr(), a.batch, st)
            ctx.counts()
    tot = 0.0
    for c in range(api.K_COUNT):
        ms, n = ctx.profile_read(c, -1)
        if not n:
            continue
        per_oct = []
        for o in range(ctx.n_oct):
            m, k = ctx.profile_read(c, o)
            if k:
                per_oct.append(f"o{o}:{m / a.iters:.3f}")
        tot += ms / a.iters
        print(f"{api.KERNEL_CLASSES[c]:12s} {ms / a.iters:8.3f} ms/iter  ({n // a.iters} launches)  " + " ".join(per_oct))
    print(f"sum {tot:.3f} ms/iter for {a.batch} images -> {tot / a.batch * 1e3:.1f} us/image")
    g = ctx.geometry
    for o in range(min(ctx.n_oct, 3)):
        row = []
        for s in range(ctx.n_sca):
            ms, n = ctx.profile_read_blur(o, s)
            if n:
                row.append(f"s{s}(R{g.blur_radius[o][s]}):{ms / n:.3f}")
        print(f"blur launches octave {o}: " + " ".join(row))
    if a.roofline:
        roofline_table(ctx, a)


def roofline_table(ctx, a):
    """Achieved algorithmic HBM GB/s per kernel class and octave against the 8 TB/s peak.

    Algorithmic bytes per octave pixel (N_o pixels, B images), priced as one launch per blur: seed
    launch 5 (reads a quarter of the pixels, writes one image); the blur launches 8 each (read source,
    write Gaussian image; DoG images are not stored), +1 for the one that also writes the next octave's
    first image; extrema scan 4 per Gaussian image (it forms the DoG values itself).  The launch of
    the last image of a stack, which a whole-pipeline run evaluates only where it is read, is left
    out of both the bytes and the time of its octave.
    """
    B, ns = a.batch, ctx.n_sca
    sparse = api.launch_stats()["sparse"] > 0 and not a.scalespace_only
    print()
    print("| kernel class | octave | pixels/image | launches | ms/iter | algorithmic MB | GB/s | of 8 TB/s |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for cls, name in ((api.K_OVERSAMPLE, "seed blur (oversampling fused)"), (api.K_BLUR_FUSED, "blur (+ marks)"),
                      (api.K_EXTREMA, "extrema scan")):
        for o in range(ctx.n_oct):
            ms, n = ctx.profile_read(cls, o)
            if not n:
                continue
            npx = ctx.dims[o][0] * ctx.dims[o][1]
            if cls == api.K_OVERSAMPLE:
                per_px = 5
            elif cls == api.K_BLUR_FUSED:
                per_px = 8 * (ns - 1) + (1 if o + 1 < ctx.n_oct else 0)
                if o == 0 and ctx.profile_read(api.K_OVERSAMPLE, 0)[1] == 0:
                    per_px += 5          # the seed blur runs inside the first two-blur launch
                if sparse and ctx.dims[o][0] % 4 == 0:
                    ms_last, n_last = ctx.profile_read_blur(o, ns - 1)
                    ms -= ms_last
                    n -= n_last
                    per_px -= 8
            else:
                per_px = 4 * ns
            mb = B * npx * per_px / 1e6
            t = ms / a.iters
            gbs = mb / 1e3 / (t * 1e-3)
            print(f"| {name} | {o} | {npx} | {n // a.iters} | {t:.3f} | {mb:.1f} | {gbs:.0f} | {gbs / 8000:.2f} |")


if __name__ == "__main__":
    main()
