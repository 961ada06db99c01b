"""Randomised parity stress of the FFT tile correlation (findOffset, image3.cpp:346-517) against the CPU oracle, beyond the
cases the test-suite pins:
   python tools/stress_xcorr.py [cases] [seed]
Every case: a random transform size (powers of two, multiples of 32 -- the sizes whose Gaussian runs on the matrix cores --,
arbitrary and odd sizes, rectangles), 8/16-bit/float rasters, a batch of 1..12 tile pairs with random boxes (smaller than
the transform, displaced between the rasters); offsets equal to the oracle's (or a tie of the DoG surface within float
noise), peak within 1e-4; on the first tile the normalised surface within 3e-5 of the float64 oracle's -- or, where float32
itself is noisier than that (weak peaks: the normalisation divides by max - min), within twice the distance of the oracle's
own float32 arithmetic from its float64 one -- and the DoG within 1e-4.  Prints a line per failure and the largest surface
error relative to that float32 yardstick."""
import os, sys
R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R0 + '/image-processing_amd', R0 + '/oracle', R0 + '/tests']
import numpy as np
import mistereo, oracle, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = mistereo.Context(0)
bad = 0
ties = 0
worst = 0.0
for k in range(cases):
    rng = np.random.default_rng(seed0 * 100003 + k)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        w, h = int(2 ** rng.integers(4, 10)), int(2 ** rng.integers(4, 10))
    elif kind == 1:
        w, h = 32 * int(rng.integers(1, 13)), 32 * int(rng.integers(1, 13))
    elif kind == 2:
        w, h = int(rng.integers(20, 400)), int(rng.integers(20, 400))
    else:
        w = h = int(rng.choice([64, 96, 128, 160, 256, 320, 512]))
    dtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
    ntiles = int(rng.integers(1, 13))
    # a frame that holds a few tiles side by side; image 1's content is image 2's moved by (dx, dy)
    fw, fh = w * 2 + 40, h * 2 + 40
    dx, dy = int(rng.integers(-w // 5, w // 5 + 1)), int(rng.integers(-h // 5, h // 5 + 1))
    big = synth.texture(fh + 2 * abs(dy) + 8, fw + 2 * abs(dx) + 8, 40000 + k, np.uint16 if dtype != np.uint8 else np.uint8)
    oy, ox = abs(dy) + 4, abs(dx) + 4
    img2 = np.ascontiguousarray(big[oy:oy + fh, ox:ox + fw]).astype(dtype)
    img1 = np.ascontiguousarray(big[oy - dy:oy - dy + fh, ox - dx:ox - dx + fw]).astype(dtype)
    r1, r2 = [], []
    for t in range(ntiles):
        bw, bh = int(rng.integers(max(w // 2, 1), w + 1)), int(rng.integers(max(h // 2, 1), h + 1))
        if rng.random() < 0.5:
            bw, bh = w, h
        x1, y1 = int(rng.integers(0, fw - bw + 1)), int(rng.integers(0, fh - bh + 1))
        x2 = min(max(x1 + int(rng.integers(-6, 7)), 0), fw - bw)
        y2 = min(max(y1 + int(rng.integers(-6, 7)), 0), fh - bh)
        r1.append((x1, y1, bw, bh)); r2.append((x2, y2, bw, bh))
    try:
        ctx.upload_pair(img1, img2)
        got = ctx.find_offsets(r1, r2, w, h)
        for t, (a, b) in enumerate(zip(r1, r2)):
            ref = oracle.find_offset(img1, img2, a, b, w, h, precision=1, want_dog=True, want_surface=(t == 0))
            gx, gy = int(got["dx"][t]), int(got["dy"][t])
            if (gx, gy) != (ref["dx"], ref["dy"]):
                # a different peak position is acceptable only where the two DoG values tie within float noise
                v = ref["dog"][gy % h, gx % w]
                assert abs(v - ref["dog_peak"]) <= 2e-5, "tile %d: offset (%d, %d), oracle (%d, %d), DoG gap %g" % (
                    t, gx, gy, ref["dx"], ref["dy"], abs(v - ref["dog_peak"]))
                ties += 1
            else:
                assert abs(got["peak"][t] - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5, \
                    "tile %d: peak %g, oracle %g" % (t, got["peak"][t], ref["peak"])
            if t == 0:
                surf, dog = ctx.tile_surfaces(a, b, w, h)
                err = np.abs(surf - ref["surface"]).max()
                if err > 1e-5:
                    ref32 = oracle.find_offset(img1, img2, a, b, w, h, precision=0, want_surface=True)
                    e32 = np.abs(ref32["surface"] - ref["surface"]).max()
                    worst = max(worst, err / max(e32, 1e-12))
                    assert err <= max(3e-5, 2 * e32), "surface off by %g (float32 oracle: %g)" % (err, e32)
                assert np.abs(dog - ref["dog"]).max() <= 1e-4, "DoG off by %g" % np.abs(dog - ref["dog"]).max()
    except AssertionError as e:
        bad += 1
        print("FAIL case", k, dict(w=w, h=h, dtype=np.dtype(dtype).name, ntiles=ntiles, dx=dx, dy=dy), str(e)[:200], flush=True)
    if k % 20 == 19:
        print("...", k + 1, "cases,", bad, "failures,", ties, "DoG ties", flush=True)
print("done:", cases, "cases,", bad, "failures,", ties, "DoG ties; largest surface error / float32 oracle's: %.2f" % worst)
sys.exit(1 if bad else 0)
