"""Shader clock held while the blur launches run (library built with -DSA_BLUR_CLOCK):
    SIFT_AMD_EXTRA_FLAGS=-DSA_BLUR_CLOCK python -m sift_amd.build --force
    python tools/blur_clock.py
Per radius: s_memtime ticks / s_memrealtime ticks (100 MHz) summed over the blocks of all octaves."""
import ctypes, sys, numpy as np
sys.path.insert(0, '.')
import torch
import sift_amd
from sift_amd import synth
B = 128
imgs = np.stack([synth.make("smooth", 1920, 1080, seed=12345 + b) for b in range(4)] * (B // 4))
d = torch.from_numpy(imgs).cuda()
ctx = sift_amd.Context(1920, 1080, max_batch=B)
L = ctx.L
out = (ctypes.c_ulonglong * 34)()
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    ctx.scalespace_device(d.data_ptr(), B, st)
torch.cuda.synchronize()
L.sift_amd_debug_blur_clock(out, 1)
for _ in range(100):
    ctx.scalespace_device(d.data_ptr(), B, st)
torch.cuda.synchronize()
L.sift_amd_debug_blur_clock(out, 1)
for R in range(17):
    if out[2 * R + 1]:
        print(f"R={R:2d}  shader ticks {out[2 * R]:16d}  realtime ticks {out[2 * R + 1]:14d}  clock {out[2 * R] / out[2 * R + 1] * 0.1:.3f} GHz")
