"""Where the waves of the two-blur launches spend their time (library built with -DSA_PAIR_CLOCK):
    tools/build_variant.sh pairclock -DSA_PAIR_CLOCK
    SIFT_AMD_LIB=sift_amd/build/variants/pairclock.so python tools/pair_clock.py
Shader-clock ticks per phase of an iteration, per role, summed over all waves of all two-blur launches
of the scale-space of 128 1080p images (octave 0: seed + s1 with the oversampling in the producer's
loads; s2 + s3; octaves 1-2: s1 + s2)."""
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
out = (ctypes.c_ulonglong * 20)()
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    ctx.scalespace_device(d.data_ptr(), B, st)
torch.cuda.synchronize()
L.sift_amd_debug_pair_clock(out, 1)
N = 20
for _ in range(N):
    ctx.scalespace_device(d.data_ptr(), B, st)
torch.cuda.synchronize()
L.sift_amd_debug_pair_clock(out, 1)
a = np.array(out[:], dtype=np.float64).reshape(2, 2, 5)
names = ["rows into the window (+ wait for their loads)", "column pass", "barrier", "row pass + stores + window shift"]
for ovs, kind in ((1, "seed + s1 (oversampling in the producer's loads)"), (0, "other two-blur launches (5+7, 7+8)")):
    print(kind)
    for role, rn in ((0, "producer"), (1, "consumer")):
        it = a[ovs, role, 4]
        tot = a[ovs, role, :4].sum()
        if it == 0:
            continue
        print(f"  {rn}: {tot / it:8.0f} ticks per wave and iteration")
        for k in range(4):
            print(f"      {names[k]:48s} {a[ovs, role, k] / it:8.0f}  {100 * a[ovs, role, k] / tot:5.1f} %")
