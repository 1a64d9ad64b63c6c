#!/usr/bin/env python3
"""Wall time per batch (graph replay, inputs resident, keypoints gathered) for several batch sizes,
to place the batch size up to which the two-chain schedule (SIFT_AMD_FORK) pays."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sift_amd
from sift_amd import synth
w, h = int(sys.argv[1]), int(sys.argv[2])
imgs = np.stack([synth.make("smooth", w, h, 12345 + b) for b in range(4)])
for B in (1, 2, 4, 8, 16, 32, 64):
    if B * w * h > 128 * 1920 * 1080:
        break
    d = torch.from_numpy(np.concatenate([imgs] * ((B + 3) // 4))[:B].copy()).cuda()
    torch.cuda.synchronize()
    ctx = sift_amd.Context(w, h, max_batch=B)
    for _ in range(4):
        ctx.detect_device(d.data_ptr(), B, None); ctx.gather()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); ctx.detect_device(d.data_ptr(), B, None); ctx.gather(); ts.append(time.perf_counter() - t0)
    print(f"{w}x{h} batch {B:3d}: {1e3 * statistics.median(ts):8.3f} ms", flush=True)
    ctx.close()
