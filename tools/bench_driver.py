"""Times the level-synchronous recursion (calcDisparityMap, image3.cpp:676-865) on the GPU for a
power-of-two frame and for one whose tiles take the general-size transforms at every level.
Usage: python tools/bench_driver.py [HxW ...]      (default: 4096x4096 2000x3000 6000x8000)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "image-processing_amd")]
import ctypes as C
import numpy as np
import mistereo, synth

ctx = mistereo.Context(0)
out = {}
sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(4096, 4096), (2000, 3000), (6000, 8000)]
for (H, W) in sizes:
    left = synth.texture(H, W, 7)
    dmap = synth.planar_disparity(H, W, -20, 20, 16, 8)
    right = synth.shift_right_image(left, dmap, fill_seed=9)
    ctx.upload_pair(left, right)
    ctx.disparity_map(min_size=32)                 # warm-up: plans, twiddles, workspaces
    t0 = time.perf_counter()
    leaves, nt = ctx.disparity_map(min_size=32)
    dt = time.perf_counter() - t0
    # the library call alone (the figure above includes the Python records of every leaf)
    buf = (mistereo.Leaf * (len(leaves) + 1))()
    n_, nt_ = C.c_int(0), C.c_int(0)
    t0 = time.perf_counter()
    ctx._lib.mis_disparity_map(ctx._h, W, H, 32, buf, len(buf), C.byref(n_), C.byref(nt_))
    dt_lib = time.perf_counter() - t0
    assert n_.value == len(leaves)
    out["%dx%d" % (W, H)] = {"ms": round(dt * 1e3, 2), "library_ms": round(dt_lib * 1e3, 2), "find_offset_calls": nt, "leaves": len(leaves)}
print(json.dumps(out))
