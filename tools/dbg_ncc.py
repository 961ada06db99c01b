import os, sys
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R+'/image-processing_amd', R+'/oracle', R+'/tests']
import numpy as np, mistereo, oracle, synth
ctx = mistereo.Context(0)
for (win, dmin, dmax, w) in [(3,-5,5,70), (3,-5,58,70), (3,-5,5,140), (5,-5,5,70), (7,-5,5,70), (3,0,5,70), (3,-20,5,70), (15,-5,5,100)]:
    left = synth.texture(40, w, 1234 + 3)
    right = synth.shift_right_image(left, 2, fill_seed=99)
    ctx.upload_pair(left, right)
    g = ctx.ncc_dense(win, dmin, dmax)
    o = oracle.ncc_dense(left, right, win, dmin, dmax)
    gi, oi = g['idisp'], o['idisp']
    bad = (gi == -32768) != (oi == -32768)
    ys, xs = np.nonzero(bad)
    print(win, dmin, dmax, w, "bad", bad.sum(), "cols", (xs.min(), xs.max()) if bad.any() else None, "mismatch", ((gi != oi) & ~bad).sum())
