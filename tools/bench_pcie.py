"""Host-buffer (PCIe-inclusive) rate of the dense NCC at BASELINE config[1]: mis_upload_pair from
host rasters, mis_ncc_dense with host outputs.  A side measurement: bench.py reports the device-resident rate."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/image-processing_amd']
import numpy as np, mistereo, synth
size, D, win = 8192, 256, 21
ctx = mistereo.Context(0)
left = synth.texture(size, size, 5)
right = synth.shift_right_image(left, synth.planar_disparity(size, size, -200, 200, 64, 7), fill_seed=9)
for it in range(3):
    t0 = time.perf_counter()
    ctx.upload_pair(left, right)
    t1 = time.perf_counter()
    out = ctx.ncc_dense(win, -D, D, want_idisp=False)
    t2 = time.perf_counter()
    print("upload %.1f ms, ncc_dense with host outputs %.1f ms, total %.1f ms -> %.1f Gpx*disp/s" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, size * size * (2 * D + 1) / (t2 - t0) / 1e9), flush=True)
# the same with page-locked rasters (mis_alloc_host): inputs by DMA, result strips by DMA straight into the caller's rasters
pl, pr = ctx.alloc_host((size, size), np.uint8), ctx.alloc_host((size, size), np.uint8)
disp, peak = ctx.alloc_host((size, size), np.float32), ctx.alloc_host((size, size), np.float32)
pl[:] = left; pr[:] = right
for it in range(4):
    t0 = time.perf_counter()
    ctx.upload_pair(pl, pr)
    t1 = time.perf_counter()
    ctx.ncc_dense_rows(win, -D, D, 0, size, disp, peak)
    t2 = time.perf_counter()
    print("page-locked: upload %.1f ms, ncc_dense with host outputs %.1f ms, total %.1f ms -> %.1f Gpx*disp/s" % (
        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, size * size * (2 * D + 1) / (t2 - t0) / 1e9), flush=True)
assert np.array_equal(disp.view(np.uint32), out["disp"].view(np.uint32)) and np.array_equal(peak.view(np.uint32), out["peak"].view(np.uint32))
# ... and through a group on page-locked rasters (every slot is GPU 0 here: its host threads share one device and one link)
for slots in (1, 2):
    g = mistereo.Group([0] * slots)
    for it in range(3):
        t0 = time.perf_counter()
        g.ncc_dense(pl, pr, win, -D, D, want_idisp=False, out=(disp, peak))
        t1 = time.perf_counter()
    print("page-locked, group of %d slot(s) on one GPU: %.1f ms -> %.1f Gpx*disp/s" % (slots, (t1 - t0) * 1e3, size * size * (2 * D + 1) / (t1 - t0) / 1e9), flush=True)
    assert np.array_equal(disp.view(np.uint32), out["disp"].view(np.uint32))
    g.close()
for a in (pl, pr, disp, peak):
    ctx.free_host(a)
# the same through a group (one process over several GPUs; here every slot is GPU 0): strips in parallel host threads
for slots in (1, 2, 4):
    g = mistereo.Group([0] * slots)
    for it in range(3):
        t0 = time.perf_counter()
        out = g.ncc_dense(left, right, win, -D, D, want_idisp=False)
        t1 = time.perf_counter()
    print("group of %d slot(s) on one GPU: %.1f ms -> %.1f Gpx*disp/s" % (slots, (t1 - t0) * 1e3, size * size * (2 * D + 1) / (t1 - t0) / 1e9), flush=True)
    g.close()
