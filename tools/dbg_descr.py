import numpy as np, sys
sys.path.insert(0, '.')
from sift_amd import api, synth
from oracle import pyoracle
img = synth.white(240, 180, 11)
want = pyoracle.run(img, keep_ss=False)["keys"]
with api.Context(240, 180) as ctx:
    ctx.detect_host(img)
    std, keys = ctx.keypoints(0)
    orih, descr, src = ctx.extras(0)
w = want["descr"]
print("n", len(std), len(w))
bad = np.nonzero((descr != w).any(axis=1))[0]
print("bad keypoints", len(bad), "of", len(w))
for k in bad[:5]:
    d = np.nonzero(descr[k] != w[k])[0]
    print("key", k, "o,s", keys["o"][k], keys["s"][k], "sigma", keys["sigma"][k], "nbad bins", len(d), d[:20])
    print(" got ", descr[k][d[:10]], " want", w[k][d[:10]])
