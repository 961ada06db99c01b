"""Dense disparity -> XYZ (mis_triangulate_disparity_device): every pixel of a dense-NCC result whose
peak reaches the reference's accept threshold (0.75, image3.cpp:751) is handed to calculateXYZ
(image3.cpp:148-264) as (u1,v1,u2,v2) = (x, y, x - round(disp), y), the leaf hand-over of
image3.cpp:812-824; all other pixels get NaN.  Checked bit for bit against the oracle's
triangulation of the same points, and the matched count against the oracle's own dense NCC."""
import numpy as np
import pytest

import mistereo
import oracle
import synth

pytestmark = pytest.mark.gpu

MIN_PEAK = 0.75


def _to_cuda(a):
    import torch
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).cuda().view(torch.uint16)
    return torch.from_numpy(a).cuda()


def _cameras(w, h, baseline):
    A = [[10000.0, 0.0, w / 2.0], [0.0, 10000.0, h / 2.0], [0.0, 0.0, 1.0]]
    R = np.eye(3)
    return (A, R, -R @ np.array([0.0, 0.0, 3000.0])), (A, R, -R @ np.array([baseline, 0.0, 3000.0]))


def _dense_xyz(ctx, left, right, window, dmin, dmax, c1, c2, min_peak=MIN_PEAK, subpixel=False):
    import torch
    h, w = left.shape
    tl, tr = _to_cuda(left), _to_cuda(right)
    ctx.bind_pair_torch(tl, tr)
    disp = torch.empty((h, w), dtype=torch.float32, device="cuda")
    peak = torch.empty_like(disp)
    xyz = torch.full((h, w, 3), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.ncc_dense_torch(window, dmin, dmax, disp, peak)
    n = ctx.triangulate_disparity_torch(mistereo.make_camera(*c1), mistereo.make_camera(*c2), disp, peak, min_peak, xyz,
                                        subpixel=subpixel)
    ctx.synchronize()
    return disp.cpu().numpy(), peak.cpu().numpy(), xyz.cpu().numpy(), n


def _check(disp, peak, xyz, n, c1, c2, min_peak=MIN_PEAK):
    h, w = disp.shape
    ok = (disp != np.float32(mistereo.INVALID_DISP)) & (peak >= np.float32(min_peak))
    assert n == int(ok.sum()), "matched count %d, expected %d" % (n, int(ok.sum()))
    assert np.all(np.isnan(xyz[~ok])), "unmatched pixels must be NaN"
    ys, xs = np.nonzero(ok)
    u2 = xs - np.rint(disp[ys, xs]).astype(np.int64)           # rintf: half to even, as numpy
    pts = np.stack([xs, ys, u2, ys], axis=1).astype(np.int32)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    got = xyz[ys, xs]
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    return int(ok.sum())


def test_u8_pair_to_xyz(ctx):
    h, w = 100, 1600
    left, right, p = synth.config_pair(2, h=h, w=w)            # piecewise disparity in [-200,200] + noise
    c1, c2 = _cameras(w, h, 60.0)
    disp, peak, xyz, n = _dense_xyz(ctx, left, right, p["window"], p["dmin"], p["dmax"], c1, c2)
    matched = _check(disp, peak, xyz, n, c1, c2)
    assert 0 < matched < h * w
    # end to end against the oracle's own dense NCC: same gate and same points, except where the peak
    # lies within float noise of the threshold and where the integer match is one of the stated
    # near-ties (this pair repeats left-image content in the right image at block boundaries: two
    # disparities score within 2e-5 of each other and either may win)
    ora = oracle.ncc_dense(left, right, p["window"], p["dmin"], p["dmax"], want_volume=True)
    ook = (ora["idisp"] != -32768) & (ora["peak"] >= np.float32(MIN_PEAK))
    gok = (disp != np.float32(-32768)) & (peak >= np.float32(MIN_PEAK))
    assert (ook != gok).sum() <= 1e-3 * ook.sum()
    gi = np.rint(disp).astype(np.int64)
    same = ook & gok & (gi == np.rint(ora["disp"]))
    ys, xs = np.nonzero(same)
    pts = np.stack([xs, ys, xs - gi[ys, xs], ys], axis=1).astype(np.int32)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    got = xyz[ys, xs]
    nan = np.isnan(want)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    assert same.sum() >= 0.99 * ook.sum()
    ys, xs = np.nonzero(ook & gok & ~same)
    vol = ora["volume"]
    kg = np.clip(gi[ys, xs] - p["dmin"], 0, vol.shape[2] - 1)
    ko = np.clip(np.rint(ora["disp"][ys, xs]).astype(np.int64) - p["dmin"], 0, vol.shape[2] - 1)
    gap = np.abs(vol[ys, xs, kg] - vol[ys, xs, ko])
    near_tie = gap <= 2e-5
    boundary = np.abs(disp[ys, xs] - ora["disp"][ys, xs]) <= 0.01     # x.5 rounded either way
    assert np.all(near_tie | boundary), "integer match differs outside a near-tie: max gap %g" % gap.max()


def test_u16_w31_pm512_pair_to_xyz(ctx):
    """BASELINE config[4]'s shape of work on a band: 16-bit, 31 x 31 window, +-512 search (1025
    disparities), subpixel disparity rounded to the integer match the reference triangulates."""
    h, w = 100, 1600
    rng = np.random.default_rng(5)
    left = synth.texture(h, w, 0xC0FFEE + 5, np.uint16).astype(np.int64) + rng.integers(-120, 121, (h, w))
    left = np.clip(left, 0, 65535).astype(np.uint16)
    right = synth.shift_right_image_subpixel(left, 37.25)
    right[:, 900:] = synth.shift_right_image(left, -301, fill_seed=9)[:, 900:]     # a second plane, negative disparity
    c1, c2 = _cameras(w, h, 900.0)
    disp, peak, xyz, n = _dense_xyz(ctx, left, right, 31, -512, 512, c1, c2)
    matched = _check(disp, peak, xyz, n, c1, c2)
    assert matched > 0.3 * h * w


def test_subpixel_columns_to_xyz(ctx):
    """MIS_TRI_SUBPIXEL: the fractional column x - disp goes into calculateXYZ's arithmetic.  The reference
    only triangulates integer pixels (image3.cpp:148, 812-824), so beyond integer-valued disparities this is
    PARITY UNPINNED by the reference: checked bit for bit against the oracle's same arithmetic on float
    coordinates, and against the flag-less form where the disparities are integers."""
    import torch
    h, w = 100, 1600
    left = synth.texture(h, w, 0xC0FFEE + 5, np.uint16)
    right = synth.shift_right_image_subpixel(left, 37.25)
    right[:, 900:] = synth.shift_right_image_subpixel(left, -300.6)[:, 900:]
    c1, c2 = _cameras(w, h, 900.0)
    disp, peak, xyz, n = _dense_xyz(ctx, left, right, 31, -512, 512, c1, c2, subpixel=True)
    ok = (disp != np.float32(mistereo.INVALID_DISP)) & (peak >= np.float32(MIN_PEAK))
    assert n == int(ok.sum()) and n > 0.3 * h * w
    assert np.all(np.isnan(xyz[~ok]))
    ys, xs = np.nonzero(ok)
    frac = disp[ys, xs] - np.rint(disp[ys, xs])
    assert (np.abs(frac) > 0.05).mean() > 0.5                    # the case is about fractional columns
    u2 = xs.astype(np.float32) - disp[ys, xs]                    # one float32 subtraction, as in the kernel
    pts = np.stack([xs.astype(np.float32), ys.astype(np.float32), u2, ys.astype(np.float32)], axis=1)
    want = oracle.triangulate_f(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    got = xyz[ys, xs]
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    # integer-valued disparities: both forms give the same bits
    rng = np.random.default_rng(11)
    di = torch.from_numpy(rng.integers(-200, 201, (h, w)).astype(np.float32)).cuda()
    pk = torch.ones_like(di)
    a = torch.empty((h, w, 3), dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    cam1, cam2 = mistereo.make_camera(*c1), mistereo.make_camera(*c2)
    torch.cuda.synchronize()
    na = ctx.triangulate_disparity_torch(cam1, cam2, di, pk, MIN_PEAK, a)
    nb = ctx.triangulate_disparity_torch(cam1, cam2, di, pk, MIN_PEAK, b, subpixel=True)
    ctx.synchronize()
    assert na == nb == h * w
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_strips_of_a_frame_triangulated_on_their_own(ctx):
    """A frame split into row strips (several GPUs): every strip held as a raster of its own and triangulated with
    its frame row offset gives the rows of the whole frame's result bit for bit, counts included; rounded and
    subpixel columns."""
    import torch
    h, w = 96, 700
    rng = np.random.default_rng(17)
    disp = (rng.integers(-150, 151, (h, w)) + rng.random((h, w)) - 0.5).astype(np.float32)
    peak = rng.random((h, w)).astype(np.float32)
    disp[rng.random((h, w)) < 0.05] = np.float32(mistereo.INVALID_DISP)
    c1, c2 = _cameras(w, h, 900.0)
    cam1, cam2 = mistereo.make_camera(*c1), mistereo.make_camera(*c2)
    ctx.upload_pair(np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8))     # the whole-frame entry takes its size from the pair
    td, tp = torch.from_numpy(disp).cuda(), torch.from_numpy(peak).cuda()
    for subpixel in (False, True):
        full = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        nfull = ctx.triangulate_disparity_torch(cam1, cam2, td, tp, MIN_PEAK, full, subpixel=subpixel)
        ctx.synchronize()
        total = 0
        for r0, n in ((0, 31), (31, 1), (32, 40), (72, 24)):
            sd, sp = td[r0:r0 + n].contiguous(), tp[r0:r0 + n].contiguous()      # the strip as a raster of its own
            sx = torch.full((n, w, 3), 5.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            total += ctx.triangulate_disparity_rows_torch(cam1, cam2, sd, sp, MIN_PEAK, sx, 0, n, r0, subpixel=subpixel)
            ctx.synchronize()
            assert torch.equal(sx.view(torch.int32), full[r0:r0 + n].view(torch.int32))
        assert total == nfull
        # a window of rows inside a larger raster: the other rows are left alone
        part = torch.full((h, w, 3), 5.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.triangulate_disparity_rows_torch(cam1, cam2, td, tp, MIN_PEAK, part, 10, 20, 0, subpixel=subpixel)
        ctx.synchronize()
        assert torch.equal(part[10:30].view(torch.int32), full[10:30].view(torch.int32))
        assert bool((part[:10] == 5).all()) and bool((part[30:] == 5).all())


def test_thresholds_and_empty_result(ctx):
    """min_peak above every peak: nothing matched, all NaN, count 0; min_peak below every peak: every
    valid pixel matched (including zero-disparity pixels, whose rays never meet: inf / NaN from the
    reference's own arithmetic, compared as such)."""
    h, w = 64, 400
    left = synth.texture(h, w, 31)
    right = left.copy()
    right[:, 200:] = synth.shift_right_image(left, 9, fill_seed=2)[:, 200:]
    c1, c2 = _cameras(w, h, 30.0)
    disp, peak, xyz, n = _dense_xyz(ctx, left, right, 9, -20, 20, c1, c2, min_peak=2.0)
    assert n == 0 and np.all(np.isnan(xyz))
    disp, peak, xyz, n = _dense_xyz(ctx, left, right, 9, -20, 20, c1, c2, min_peak=-2.0)
    assert _check(disp, peak, xyz, n, c1, c2, min_peak=-2.0) == (h - 8) * (w - 8)


def test_two_million_points_over_random_geometries(ctx):
    """Random intrinsics, rotations, heights, baselines and coordinate offsets (up to millions of
    metres), 500 k points each: the search's early-exit bound and hoisted division give the
    reference's bits in all of them."""
    def rot(a):
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    n = 500_000
    for g in range(4):
        rng = np.random.default_rng(1000 + g)
        f = float(rng.uniform(2000, 20000))
        A = [[f, 0, float(rng.uniform(1000, 5000))], [0, f * float(rng.uniform(0.97, 1.03)), float(rng.uniform(1000, 5000))], [0, 0, 1]]
        amp = [0.0, 0.02, 0.3, 0.02][g]
        R1, R2 = rot(rng.uniform(-amp, amp, 3)), rot(rng.uniform(-amp, amp, 3))
        hgt = float(rng.uniform(200, 6000))
        base = float(rng.uniform(20, 2000))
        off = [0.0, 1.0e5, 0.0, 4.0e6][g] * np.array([1.0, 1.0, 0.0])
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
        assert np.array_equal(np.isnan(got), nan), "geometry %d" % g
        assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan]), "geometry %d" % g


@pytest.mark.parametrize("d", [1.0, -1.0, 2.0, -3.0, 7.0, 20.0])
def test_nearly_parallel_rays(ctx, d):
    """Small disparities against a long baseline: the rays are all but parallel, the secant start of the search lies
    millions of units away and both of the reference's searches run their hundred steps past the minimum
    (image3.cpp:220-246).  Bit for bit against the oracle on every pixel of a constant-disparity raster."""
    import torch
    h, w = 24, 160
    c1, c2 = _cameras(w, h, 900.0)
    left = np.zeros((h, w), np.uint8)
    ctx.bind_pair_torch(_to_cuda(left), _to_cuda(left))
    disp = torch.full((h, w), d, dtype=torch.float32, device="cuda")
    peak = torch.ones((h, w), dtype=torch.float32, device="cuda")
    xyz = torch.full((h, w, 3), 7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n = ctx.triangulate_disparity_torch(mistereo.make_camera(*c1), mistereo.make_camera(*c2), disp, peak, MIN_PEAK, xyz)
    ctx.synchronize()
    assert _check(disp.cpu().numpy(), peak.cpu().numpy(), xyz.cpu().numpy(), n, c1, c2) == h * w


def test_rasters_higher_than_one_launch_covers(ctx):
    """A launch covers at most 65535 rows: a 65536-row raster (the highest a pair may be) is triangulated in slabs, and
    equals its two halves triangulated as strips of the frame; a negative frame row offset is refused."""
    import torch
    h, w = 65536, 8
    rng = np.random.default_rng(3)
    disp = torch.from_numpy(rng.integers(-30, 31, (h, w)).astype(np.float32)).cuda()
    peak = torch.from_numpy(rng.uniform(0.5, 1.0, (h, w)).astype(np.float32)).cuda()
    c1, c2 = (mistereo.make_camera(*c) for c in _cameras(w, h, 60.0))
    whole = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    halves = torch.ones((h, w, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n = ctx.triangulate_disparity_rows_torch(c1, c2, disp, peak, MIN_PEAK, whole, 0, h, 0)
    # the lower half held as a raster of its own: its row k is frame row k + h/2
    n0 = ctx.triangulate_disparity_rows_torch(c1, c2, disp, peak, MIN_PEAK, halves, 0, h // 2, 0)
    n1 = ctx.triangulate_disparity_rows_torch(c1, c2, disp[h // 2:], peak[h // 2:], MIN_PEAK, halves[h // 2:], 0, h // 2, h // 2)
    ctx.synchronize()
    assert n == n0 + n1 == int((peak >= MIN_PEAK).sum().item())
    a, b = whole.cpu().numpy().view(np.uint32), halves.cpu().numpy().view(np.uint32)
    assert np.array_equal(a, b)
    with pytest.raises(mistereo.MisError):
        ctx.triangulate_disparity_rows_torch(c1, c2, disp, peak, MIN_PEAK, whole, 0, 16, -1)
