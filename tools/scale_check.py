import sys, numpy as np
sys.path.insert(0,'.')
import sift_amd
from sift_amd import synth
from oracle import pyoracle
w,h=int(sys.argv[1]),int(sys.argv[2])
img=synth.white(w,h,seed=5)
want=pyoracle.run(img,keep_ss=True)
for flags in (dict(keep_all_extrema=True,keep_scalespace=True), dict()):
    with sift_amd.Context(w,h,1,device=0,**flags) as ctx:
        ctx.detect_host(img)
        for o in range(ctx.n_oct):
            g=ctx.scalespace(0,0,o); wg=want["gss"][o] if "gss" in want else None
            if wg is None: print(list(want.keys())); break
            eq=[bool(np.array_equal(g[s].view(np.uint32),wg[s].view(np.uint32))) for s in range(g.shape[0])]
            bad=[(s,int((g[s]!=wg[s]).sum())) for s in range(g.shape[0]) if not eq[s]]
            print(flags.keys(),'oct',o,g.shape,eq,bad[:3])
            if o==0 and not eq[0]:
                ii,jj=np.nonzero(g[0]!=wg[0]); print('rows',np.unique(ii)[:40],'cols',np.unique(jj)[:60], 'maxdiff', np.abs(g[0]-wg[0]).max())
