"""A/B timing of the dense-NCC kernels from the library's own HIP events:
   [MISTEREO_LIB=abtest/x.so] python tools/ab_ncc.py u8|u16 window disp width height [peak2]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/image-processing_amd']
import numpy as np, torch, mistereo, synth
dt = sys.argv[1] if len(sys.argv) > 1 else "u8"
win = int(sys.argv[2]) if len(sys.argv) > 2 else 21
D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
W = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
H = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
p2 = len(sys.argv) > 6 and sys.argv[6] == "peak2"
left = synth.texture(H, W, 0xC0FFEE + 5, np.uint16 if dt == "u16" else np.uint8)
right = synth.shift_right_image(left, 37, fill_seed=3)
if dt == "u16":
    tl = torch.from_numpy(left.view(np.int16)).cuda().view(torch.uint16); tr = torch.from_numpy(right.view(np.int16)).cuda().view(torch.uint16)
else:
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
disp = torch.empty((H, W), dtype=torch.float32, device="cuda"); peak = torch.empty_like(disp)
peak2 = torch.empty_like(disp) if p2 else None
ctx = mistereo.Context(0)
ctx.profile(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st = torch.cuda.Stream(); ctx.set_stream(st.cuda_stream)
for it in range(5):
    if it == 1:
        ctx.profile_read(); e0.record(st)
    ctx.bind_pair_torch(tl, tr)
    ctx.ncc_dense_torch(win, -D, D, disp, peak, None, peak2)
e1.record(st); st.synchronize()
t, a, r, n = ctx.profile_read()
print("%-22s %s W%d +-%d %dx%d%s: step %.3f ms | tables %.3f argmax %.3f refine %.3f  (%.1f Gpx*disp/s)" %
      (os.path.basename(mistereo.LIB_PATH), dt, win, D, W, H, " peak2" if p2 else "", e0.elapsed_time(e1) / 4, t / n, a / n, r / n,
       W * H * (2 * D + 1) / (e0.elapsed_time(e1) / 4) / 1e6), flush=True)
