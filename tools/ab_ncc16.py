"""A/B timing of the 16-bit dense-NCC kernels on BASELINE config[4]'s band (16384 x 1024, 31x31, +-512):
   MISTEREO_LIB=abtest/x.so python tools/ab_ncc16.py [window] [disp] [rows]   -> kernel times from the library's HIP events"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/image-processing_amd']
import numpy as np, torch, mistereo, synth
win = int(sys.argv[1]) if len(sys.argv) > 1 else 31
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
W = 16384
left = synth.texture(H, W, 0xC0FFEE + 5, np.uint16)
right = synth.shift_right_image_subpixel(left, 37.25)
tl = torch.from_numpy(left.view(np.int16)).cuda().view(torch.uint16)
tr = torch.from_numpy(right.view(np.int16)).cuda().view(torch.uint16)
disp = torch.empty((H, W), dtype=torch.float32, device="cuda"); peak = torch.empty_like(disp)
ctx = mistereo.Context(0)
ctx.profile(True)
for it in range(4):
    ctx.bind_pair_torch(tl, tr)
    if it == 1: ctx.profile_read()
    ctx.ncc_dense_torch(win, -D, D, disp, peak)
t, a, r, n = ctx.profile_read()
print("%-28s W%d +-%d %dx%d: tables %.3f  argmax %.3f  refine %.3f ms   (%.1f Gpx*disp/s kernel)" %
      (os.path.basename(mistereo.LIB_PATH), win, D, W, H, t / n, a / n, r / n, W * H * (2 * D + 1) / (a / n) / 1e6), flush=True)
