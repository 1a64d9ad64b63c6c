import sys, numpy as np
sys.path.insert(0, '.')
import sift_amd
from sift_amd import synth
imgs = np.stack([synth.make("natural", 1920, 1080, seed=12345 + b) for b in range(2)])
with sift_amd.Context(1920, 1080, max_batch=2) as ctx:
    ctx.detect_host(imgs)
    std, keys = ctx.keypoints(0)
    print("n", len(keys))
    print("octaves", np.bincount(keys["o"]))
    so = keys["sigma"] / (0.5 * 2.0 ** keys["o"])
    print("sigma_oct pct", np.percentile(so, [0, 25, 50, 75, 100]))
    print("scale idx", np.bincount(keys["s"]))
