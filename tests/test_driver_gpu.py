"""Coarse-to-fine driver (calcDisparityMap, image3.cpp:676-865): the level-synchronous product
driver against the oracle's depth-first restatement, plus the XYZ / disparity-raster writers."""
import os

import numpy as np
import pytest

import mistereo
import oracle
import synth

pytestmark = pytest.mark.gpu


def _shifted_pair(size, dx, dy, seed):
    big = synth.texture(size + 2 * abs(dy) + 8, size + 2 * abs(dx) + 8, seed)
    oy, ox = abs(dy) + 4, abs(dx) + 4
    img2 = np.ascontiguousarray(big[oy:oy + size, ox:ox + size])
    img1 = np.ascontiguousarray(big[oy - dy:oy - dy + size, ox - dx:ox - dx + size])
    return img1, img2


@pytest.mark.parametrize("size,dx,dy,min_size", [(256, 0, 0, 32), (512, 13, -6, 32), (512, -40, 25, 64), (256, 7, 3, 16)])
def test_leaves_match_the_depth_first_reference_order(ctx, size, dx, dy, min_size):
    img1, img2 = _shifted_pair(size, dx, dy, 300 + size + dx)
    ctx.upload_pair(img1, img2)
    got, ntiles = ctx.disparity_map(min_size=min_size)
    want, ncalls = oracle.disparity_map(img1, img2, min_size=min_size, precision=1)
    assert ntiles == ncalls
    assert len(got) == len(want) > 0
    for g, w in zip(got, want):
        assert g[:10] == w[:10]                      # boxes, offsets and matched pixels, in order
        assert abs(g[10] - w[10]) <= 1e-4 * max(abs(w[10]), 1e-3) + 1e-5
    if dx or dy:
        inner = [g for g in got if g[10] > 0.75]
        assert inner and all((g[4], g[5]) == (dx, dy) for g in inner)


def test_zero_shift_counts(ctx):
    """2048-px pair, zero shift, min_size 32: 1 + 4 + 16 + 64 + 256 + 1024 = 1365 evaluations and
    1024 leaves, the count the reference makes for BASELINE config[0]."""
    img = synth.texture(2048, 2048, 11)
    ctx.upload_pair(img, img)
    got, ntiles = ctx.disparity_map(min_size=32, max_leaves=4096)
    assert ntiles == 1365 and len(got) == 1024
    assert all(g[2] == 64 and g[4] == 0 and g[5] == 0 for g in got)


def test_xyz_file_and_disparity_bands(ctx, tmp_path):
    img1, img2 = _shifted_pair(256, 9, -4, 77)
    ctx.upload_pair(img1, img2)
    leaves, _ = ctx.disparity_map(min_size=32)
    A = [[10000, 0, 128], [0, 10000, 128], [0, 0, 1]]
    c1 = mistereo.make_camera(A, np.eye(3), [0, 0, -3000])
    c2 = mistereo.make_camera(A, np.eye(3), [-900, 0, -3000])
    pts = np.array([l[6:10] for l in leaves], np.int32)
    xyz = ctx.calculate_xyz(c1, c2, pts)
    bands = np.array([[float(img1[l[7], l[6]])] for l in leaves], np.float32)
    path = os.path.join(tmp_path, "out.xyz")
    mistereo.write_xyz(path, xyz, bands)
    lines = open(path).read().splitlines()
    assert len(lines) == len(leaves)
    for ln, p, b in zip(lines, xyz, bands):
        assert ln == "%f %f %f %f " % (p[0], p[1], p[2], b[0])     # image3.cpp:842-849
        assert len(ln.split()) == 4                                  # what xyz2pcd's sscanf reads
    ras = mistereo.rasterize_leaves(leaves, 256, 256, fill=-1.0)
    # later leaves overwrite earlier ones where clamped tiles overlap: check the last writer
    last = leaves[-1]
    cx, cy = last[0] + last[2] // 2, last[1] + last[3] // 2
    assert ras[1, cy, cx] == last[4] and ras[2, cy, cx] == last[5]
    assert ras[0, cy, cx] == np.float32(np.sqrt(np.float32(last[4] * last[4] + last[5] * last[5])))


def test_frame_that_is_not_a_power_of_two(ctx):
    """360 x 250 pair: the first level correlates the whole frames (any size), the levels below are
    power-of-two tiles (image3.cpp:754-761)."""
    big = synth.texture(250 + 40, 360 + 40, 91)
    img2 = np.ascontiguousarray(big[20:270, 20:380])
    img1 = np.ascontiguousarray(big[20 + 5:270 + 5, 20 - 9:380 - 9])     # content moved by (+9, -5)
    ctx.upload_pair(img1, img2)
    got, ntiles = ctx.disparity_map(min_size=16)
    want, ncalls = oracle.disparity_map(img1, img2, min_size=16, precision=1)
    assert ntiles == ncalls and len(got) == len(want) > 1
    for g, w in zip(got, want):
        assert g[:10] == w[:10]
        assert abs(g[10] - w[10]) <= 1e-4 * max(abs(w[10]), 1e-3) + 1e-5
    assert (got[0][4], got[0][5]) == (9, -5)


def _write_pgm(path, img):
    h, w = img.shape
    with open(path, "wb") as f:
        if img.dtype == np.uint8:
            f.write(b"P5\n# test raster\n%d %d\n255\n" % (w, h)); f.write(img.tobytes())
        else:
            f.write(b"P5\n%d %d\n65535\n" % (w, h)); f.write(img.astype(">u2").tobytes())


def test_cpp_host_program_on_the_c_abi(ctx, tmp_path):
    """examples/image3_mi.cpp (plain C++ against include/mi_stereo.h, the reference's main() on the
    library): its XYZ file, dispersion raster and dense raster equal what the Python binding produces."""
    import subprocess
    exe = os.path.join(os.path.dirname(mistereo.LIB_PATH), "image3_mi")
    assert os.path.exists(exe), "build it with make -C image-processing_amd all"
    img1, img2 = _shifted_pair(256, 9, -4, 78)
    left, right = os.path.join(tmp_path, "l.pgm"), os.path.join(tmp_path, "r.pgm")
    _write_pgm(left, img1); _write_pgm(right, img2)
    yaml = "%%YAML:1.0\nA: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ 10000., 0., 128., 0., 10000., 128., 0., 0., 1. ]\n" \
           "R: !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ 1., 0., 0., 0., 1., 0., 0., 0., 1. ]\n" \
           "T: !!opencv-matrix\n   rows: 3\n   cols: 1\n   dt: d\n   data: [ %s, 0., -3000. ]\n"
    open(left + ".yaml", "w").write(yaml % "0.")
    open(right + ".yaml", "w").write(yaml % "-900.")
    xyz_path, tif_path, dense_path = (os.path.join(tmp_path, n) for n in ("o.xyz", "d.tif", "n.tif"))
    r = subprocess.run([exe, left, right, "--xyz", xyz_path, "--dispersion", tif_path, "--min-size", "32",
                        "--dense", "11", "-20", "20", dense_path], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    # the same through the Python binding
    ctx.upload_pair(img1, img2)
    leaves, ntiles = ctx.disparity_map(min_size=32)
    assert "%d tiles correlated, %d leaves" % (ntiles, len(leaves)) in r.stdout
    A = [[10000, 0, 128], [0, 10000, 128], [0, 0, 1]]
    c1 = mistereo.make_camera(A, np.eye(3), [0, 0, -3000])
    c2 = mistereo.make_camera(A, np.eye(3), [-900, 0, -3000])
    xyz = ctx.calculate_xyz(c1, c2, np.array([l[6:10] for l in leaves], np.int32))
    want = ["%f %f %f %f " % (p[0], p[1], p[2], float(img1[l[7], l[6]])) for p, l in zip(xyz, leaves)]
    assert open(xyz_path).read().splitlines() == want
    planes = mistereo.rasterize_leaves(leaves, 256, 256, fill=0.0)
    raw = open(tif_path, "rb").read()
    assert np.array_equal(np.frombuffer(raw, np.float32, 3 * 256 * 256, 8).reshape(3, 256, 256), planes)
    dense = ctx.ncc_dense(11, -20, 20, want_idisp=False)
    raw = open(dense_path, "rb").read()
    got = np.frombuffer(raw, np.float32, 2 * 256 * 256, 8).reshape(2, 256, 256)
    assert np.array_equal(got[0].view(np.uint32), dense["disp"].view(np.uint32))
    assert np.array_equal(got[1].view(np.uint32), dense["peak"].view(np.uint32))
    # --devices: the same files through a group of GPUs (every slot is GPU 0 here)
    paths2 = [os.path.join(tmp_path, n) for n in ("o2.xyz", "d2.tif", "n2.tif")]
    r2 = subprocess.run([exe, left, right, "--xyz", paths2[0], "--dispersion", paths2[1], "--min-size", "32",
                         "--dense", "11", "-20", "20", paths2[2], "--devices", "0,0,0"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stderr
    assert r2.stdout == r.stdout
    for p1, p2 in zip((xyz_path, tif_path, dense_path), paths2):
        assert open(p1, "rb").read() == open(p2, "rb").read()


def test_recursion_equal_on_the_matrix_core_and_the_reference_ordered_gaussian(tmp_path):
    """The default tile-correlation path sums the 151 Gaussian taps on the matrix cores, NOT in the reference's filter-loop
    order (image3.cpp:481; the DoG differs by about 3e-8 of the surface's range); MISTEREO_XC_MFMA=0 keeps the
    reference-ordered vector kernels.  The first-maximum pick (image3.cpp:486-501) and the accept test `max > 0.75`
    (image3.cpp:751) could flip on a near-tie between the two; on these pairs (shifted, unshifted, not a power of two)
    every level's (dx, dy), every accept decision and so every leaf must be the same.  The variable is read once per
    process: each setting runs in a child process.  Pairs outside these fixtures: parity unpinned on near-ties."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, json; sys.path[:0] = %r\n"
        "import numpy as np, mistereo, synth\n"
        "def pair(h, w, dx, dy, seed):\n"
        "    big = synth.texture(h + 2 * abs(dy) + 8, w + 2 * abs(dx) + 8, seed)\n"
        "    oy, ox = abs(dy) + 4, abs(dx) + 4\n"
        "    return (np.ascontiguousarray(big[oy - dy:oy - dy + h, ox - dx:ox - dx + w]), np.ascontiguousarray(big[oy:oy + h, ox:ox + w]))\n"
        "ctx = mistereo.Context(0); out = []\n"
        "for h, w, dx, dy, seed, ms in ((512, 512, 13, -6, 825, 32), (1024, 1024, 0, 0, 5, 32), (600, 840, -25, 17, 9, 16), (512, 512, -40, 25, 472, 64)):\n"
        "    a, b = pair(h, w, dx, dy, seed); ctx.upload_pair(a, b)\n"
        "    leaves, n = ctx.disparity_map(min_size=ms, max_leaves=1 << 15)\n"
        "    out.append([n, [[int(v) for v in l[:10]] + [bool(l[10] > 0.75), float(l[10])] for l in leaves]])\n"
        "ctx.close(); print(json.dumps(out))\n" % ([p for p in sys.path if p],))
    res = {}
    for flag in ("0", "1"):
        env = dict(os.environ, MISTEREO_XC_MFMA=flag)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        res[flag] = json.loads(r.stdout.strip().splitlines()[-1])
    for a, b in zip(res["0"], res["1"]):
        assert a[0] == b[0] and len(a[1]) == len(b[1]) > 0
        for la, lb in zip(a[1], b[1]):
            assert la[:11] == lb[:11]                      # boxes, offsets, matched pixels, accept decision
            assert abs(la[11] - lb[11]) <= 1e-6


@pytest.mark.parametrize("shape1,shape2,dx,dy", [((300, 420), (256, 400), 12, -7), ((256, 256), (300, 330), -9, 5), ((512, 400), (512, 512), 0, 3)])
def test_rasters_of_different_size(ctx, shape1, shape2, dx, dy):
    """The reference only warns when the two rasters differ in size (image3.cpp:995-998): the first level correlates
    the whole of each, zero-padded to the larger extent (image3.cpp:687-688, 322-328), and overlaps are clipped against
    the second raster (image3.cpp:684-685).  mis_upload_pair2 / mis_disparity_map2 against the oracle's walk on the two
    rasters as they are."""
    H, W = max(shape1[0], shape2[0]) + 60, max(shape1[1], shape2[1]) + 60
    big = synth.texture(H, W, 4000 + shape1[0] + shape2[1])
    img2 = np.ascontiguousarray(big[30:30 + shape2[0], 30:30 + shape2[1]])
    img1 = np.ascontiguousarray(big[30 - dy:30 - dy + shape1[0], 30 - dx:30 - dx + shape1[1]])      # content moved by (dx, dy)
    ctx.upload_pair2(img1, img2)
    got, ntiles = ctx.disparity_map(min_size=32)
    want, ncalls = oracle.disparity_map(img1, img2, min_size=32, precision=1)
    assert ntiles == ncalls and len(got) == len(want) > 1
    for g, w in zip(got, want):
        assert g[:10] == w[:10]
        assert abs(g[10] - w[10]) <= 1e-4 * max(abs(w[10]), 1e-3) + 1e-5
    assert (got[0][4], got[0][5]) == (dx, dy)
    # every matched pixel of the second raster lies inside it
    assert all(0 <= g[8] < shape2[1] and 0 <= g[9] < shape2[0] for g in got)
    ctx.upload_pair(img2, img2)                       # an equal-size pair afterwards takes the one-size path again
    got, _ = ctx.disparity_map(min_size=64)
    assert all((g[4], g[5]) == (0, 0) for g in got)


def test_rasters_of_different_size_read_as_zero_beyond_their_extent(ctx):
    """mis_upload_pair2 binds both rasters at the common size with zeros beyond each raster's own extent: every entry
    point then answers as for the explicitly zero-padded pair (dense NCC bit for bit, tile correlation likewise)."""
    a = synth.texture(150, 260, 61)
    b = synth.texture(190, 220, 62)
    H, W = 190, 260
    pa = np.zeros((H, W), np.uint8); pa[:150, :260] = a
    pb = np.zeros((H, W), np.uint8); pb[:190, :220] = b
    ctx.upload_pair(pa, pb)
    want = ctx.ncc_dense(9, -12, 12)
    wo = ctx.find_offsets([(0, 0, 128, 128), (100, 40, 64, 64)], [(0, 0, 128, 128), (90, 50, 64, 64)], 128, 128)
    ctx.upload_pair2(a, b)
    got = ctx.ncc_dense(9, -12, 12)
    for k in ("disp", "peak"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    assert np.array_equal(got["idisp"], want["idisp"])
    go = ctx.find_offsets([(0, 0, 128, 128), (100, 40, 64, 64)], [(0, 0, 128, 128), (90, 50, 64, 64)], 128, 128)
    assert np.array_equal(np.asarray(go), np.asarray(wo))
