"""One-off stress of the triangulation against the CPU oracle: millions of random points over random
camera geometries, bit for bit.  Usage (GPU box): python tools/stress_tri.py [n_geometries] [points]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import mistereo, oracle

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
ctx = mistereo.Context(0)
bad = 0
for g in range(ng):
    rng = np.random.default_rng(1000 + g)
    def rot(a):
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    f = float(rng.uniform(2000, 20000))
    A = [[f, 0, float(rng.uniform(1000, 5000))], [0, f * float(rng.uniform(0.97, 1.03)), float(rng.uniform(1000, 5000))], [0, 0, 1]]
    amp = float(rng.choice([0.0, 0.02, 0.3]))
    R1, R2 = rot(rng.uniform(-amp, amp, 3)), rot(rng.uniform(-amp, amp, 3))
    hgt = float(rng.uniform(200, 6000))
    base = float(rng.uniform(20, 2000))
    off = rng.choice([0.0, 1.0e5, 4.0e6]) * np.array([1.0, 1.0, 0.0])
    C1 = off + np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), hgt])
    C2 = off + np.array([base, rng.uniform(-50, 50), hgt + rng.uniform(-100, 100)])
    c1, c2 = (A, R1, -R1 @ C1), (A, R2, -R2 @ C2)
    dground = f * base / hgt
    u1 = rng.integers(0, 8192, n); v1 = rng.integers(0, 8192, n)
    d = rng.integers(int(-0.2 * dground) - 5, int(2.5 * dground) + 6, n)
    pts = np.stack([u1, v1, u1 - d, v1 + rng.integers(-3, 4, n)], axis=1).astype(np.int32)
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    nan = np.isnan(want)
    ok = np.array_equal(np.isnan(got), nan) and np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    nbad = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=1).sum()) if not ok else 0
    bad += nbad
    print("geometry %d: f %.0f height %.0f base %.0f rot %.2f offset %.0e: %s (%d differing points)" %
          (g, f, hgt, base, amp, off[0], "equal" if ok else "DIFFERENT", nbad), flush=True)
sys.exit(1 if bad else 0)
