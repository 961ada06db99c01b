"""Refinement-kernel time on a coherent (constant shift) and on the benchmark's piecewise pair."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/image-processing_amd']
import numpy as np, torch, mistereo, synth
size = 8192
ctx = mistereo.Context(0)
left = synth.texture(size, size, 5)
for name, dmap in (("constant", 17), ("piecewise64", synth.planar_disparity(size, size, -200, 200, 64, 7)),
                   ("piecewise512", synth.planar_disparity(size, size, -200, 200, 512, 7))):
    right = synth.shift_right_image(left, dmap, fill_seed=9)
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    disp = torch.empty((size, size), dtype=torch.float32, device="cuda"); peak = torch.empty_like(disp)
    ctx.profile(True); ctx.profile_read()
    for it in range(3):
        ctx.bind_pair_torch(tl, tr)
        ctx.ncc_dense_torch(21, -256, 256, disp, peak)
    ctx.synchronize()
    t, m, r, n = ctx.profile_read()
    print("%-13s tables %.2f  argmax %.2f  refine %.2f ms   matched %.3f" % (name, t / n, m / n, r / n, float((peak >= 0.75).float().mean())), flush=True)
