"""Timing of the 16-bit dense NCC (31x31 window, +-512 search) followed by triangulation of every
pixel's disparity to XYZ, device resident.  Usage: python tools/bench_ncc16.py [size] [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/image-processing_amd']
import numpy as np, torch, mistereo, synth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
window, dmin, dmax = 31, -512, 512
ctx = mistereo.Context(0)
left = synth.texture(size, size, 5, np.uint16)
right = synth.shift_right_image(left, synth.planar_disparity(size, size, -400, 400, 256, 7), fill_seed=9)
tl, tr = torch.from_numpy(left.view(np.int16)).cuda().view(torch.uint16), torch.from_numpy(right.view(np.int16)).cuda().view(torch.uint16)
disp = torch.empty((size, size), dtype=torch.float32, device="cuda")
peak = torch.empty_like(disp)
ctx.profile(True)
for it in range(steps + 1):
    ctx.bind_pair_torch(tl, tr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.ncc_dense_torch(window, dmin, dmax, disp, peak)
    ctx.synchronize(); t1 = time.perf_counter()
    if it:
        print("step %d: %.2f ms  %.1f Gpx*disp/s" % (it, (t1 - t0) * 1e3, size * size * (dmax - dmin + 1) / (t1 - t0) / 1e9), flush=True)
print("profile (tables, argmax, refine ms, calls):", ctx.profile_read())
