"""Where the descriptor kernel spends its time, phase by phase (cycle counters read inside the
kernel).  Needs a library built with the probes compiled in:
    SIFT_AMD_EXTRA_FLAGS=-DSA_DESCR_PROF python -m sift_amd.build --force
    python tools/descr_prof.py [smooth|white]
(rebuild without the flag afterwards: the probes cost time)."""
import ctypes, sys, numpy as np
sys.path.insert(0, '.')
import sift_amd
from sift_amd import synth
kind = sys.argv[1] if len(sys.argv) > 1 else "smooth"
B = 8
imgs = np.stack([synth.make(kind, 1920, 1080, seed=12345 + b) for b in range(B)])
names = ["setup", "scan", "geometry", "math", "Kmasks", "cellmasks", "ranks", "counts+scan", "write", "walk", "pop", "epilogue"]
with sift_amd.Context(1920, 1080, max_batch=B) as ctx:
    ctx.detect_host(imgs)
    L = ctx.L
    out = (ctypes.c_ulonglong * 16)()
    L.sift_amd_debug_prof(out, 1)
    ctx.detect_host(imgs)
    L.sift_amd_debug_prof(out, 1)
    tot = sum(out[:12])
    print(kind, "keypoints", int(ctx.counts().sum()))
    for n, v in zip(names, out):
        print(f"{n:12s} {v:14d} {100.0 * v / tot:5.1f}%")
    nb = max(out[12], 1)
    print(f"batches {out[12]}, samples per batch {out[15] / nb:.1f}, longest lane walk per batch {out[13] / nb:.2f} pairs, "
          f"mean lane walk {out[14] / nb / 64:.2f} pairs")
