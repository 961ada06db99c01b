"""Triangulation parity (calculateXYZ, image3.cpp:148-264): bit-exact float32 against the oracle."""
import numpy as np
import pytest

import mistereo
import oracle

pytestmark = pytest.mark.gpu


def cameras():
    A = [[10000, 0, 4096], [0, 10000, 4096], [0, 0, 1]]
    R = np.eye(3)
    # OpenCV convention camera = R world + t: a camera at world position C has t = -R C
    c1 = (A, R, -R @ np.array([0.0, 0.0, 3000.0]))
    c2 = (A, R, -R @ np.array([900.0, 0.0, 3000.0]))
    return c1, c2


def rotated_cameras():
    A = [[9000.5, 0, 4000.25], [0, 9100.75, 4100.5], [0, 0, 1]]
    def rot(ax, ay, az):
        cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    R1, R2 = rot(0.01, -0.02, 0.005), rot(-0.015, 0.01, -0.02)
    return (A, R1, -R1 @ np.array([10.0, -5.0, 2800.0])), (A, R2, -R2 @ np.array([850.0, 20.0, 2900.0]))


@pytest.mark.parametrize("cams", [cameras, rotated_cameras])
def test_points_bit_exact(ctx, cams):
    c1, c2 = cams()
    rng = np.random.default_rng(1)
    n = 4096
    u1 = rng.integers(0, 8192, n); v1 = rng.integers(0, 8192, n)
    d = rng.integers(-400, 401, n)
    pts = np.stack([u1, v1, u1 - d, v1 + rng.integers(-2, 3, n)], axis=1).astype(np.int32)
    pts[:8, 2] = pts[:8, 0]            # zero disparity: parallel rays, the search does not converge
    pts[8:16] = 0
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    # bit-exact, NaN payloads aside (zero-disparity rays never intersect: inf/NaN propagate)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


def test_known_geometry(ctx):
    """Ground point (X, Y, 0) seen by two nadir cameras 900 m apart at 3000 m: Z comes back near 0."""
    c1, c2 = cameras()
    X, Y = 150.0, -80.0
    f, cx, cy, Hc = 10000.0, 4096.0, 4096.0, 3000.0
    # ray = (u-cx)/f, -(v-cy)/f, -1 scaled by the height above ground (image3.cpp:185-194)
    u1, v1 = cx + f * (X - 0.0) / Hc, cy - f * Y / Hc
    u2, v2 = cx + f * (X - 900.0) / Hc, cy - f * Y / Hc
    pts = np.array([[round(u1), round(v1), round(u2), round(v2)]], np.int32)
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)[0]
    assert abs(got[0] - X) < 1.0 and abs(got[1] - Y) < 1.0 and abs(got[2]) < 2.0


def test_empty(ctx):
    c1, c2 = cameras()
    out = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), np.zeros((0, 4), np.int32))
    assert out.shape == (0, 3)


@pytest.mark.parametrize("seed", range(8))
def test_random_cameras_bit_exact(ctx, seed):
    """Random intrinsics, rotations up to a few degrees, positions and heights: still bit for bit."""
    rng = np.random.default_rng(50 + seed)
    def rot(a):
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    f = float(rng.uniform(3000, 15000))
    A = [[f, 0, float(rng.uniform(1000, 5000))], [0, f * float(rng.uniform(0.98, 1.02)), float(rng.uniform(1000, 5000))], [0, 0, 1]]
    R1, R2 = rot(rng.uniform(-0.05, 0.05, 3)), rot(rng.uniform(-0.05, 0.05, 3))
    hgt = float(rng.uniform(500, 4000))
    c1 = (A, R1, -R1 @ np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), hgt]))
    c2 = (A, R2, -R2 @ np.array([rng.uniform(300, 1200), rng.uniform(-50, 50), hgt + rng.uniform(-100, 100)]))
    n = 1500
    u1 = rng.integers(0, 8192, n); v1 = rng.integers(0, 8192, n)
    pts = np.stack([u1, v1, u1 - rng.integers(-600, 601, n), v1 + rng.integers(-3, 4, n)], axis=1).astype(np.int32)
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


@pytest.mark.parametrize("h1,h2", [(0.0, 0.0), (3.0e6, 3.0e6), (3000.0, 0.0), (3000.0, 2.0e-7)])
def test_cameras_outside_the_fast_range(ctx, h1, h2):
    """Camera heights of zero, tiny or beyond 2^20 make the library run the search with general
    division and square roots instead of its hoisted-reciprocal path: same bits either way."""
    A = [[10000, 0, 4096], [0, 10000, 4096], [0, 0, 1]]
    R = np.eye(3)
    c1 = (A, R, -R @ np.array([0.0, 0.0, h1]))
    c2 = (A, R, -R @ np.array([900.0, 0.0, h2]))
    rng = np.random.default_rng(7)
    n = 600
    u1 = rng.integers(0, 8192, n); v1 = rng.integers(0, 8192, n)
    pts = np.stack([u1, v1, u1 - rng.integers(-400, 401, n), v1 + rng.integers(-2, 3, n)], axis=1).astype(np.int32)
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


def test_many_points_bit_exact(ctx):
    """A larger sample (a million points, fractional camera parameters) through the two-step search."""
    c1, c2 = rotated_cameras()
    rng = np.random.default_rng(11)
    n = 1_000_000
    u1 = rng.integers(0, 8192, n); v1 = rng.integers(0, 8192, n)
    pts = np.stack([u1, v1, u1 - rng.integers(-600, 601, n), v1 + rng.integers(-3, 4, n)], axis=1).astype(np.int32)
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


@pytest.mark.parametrize("geom", ["short_baseline", "utm_coordinates", "oblique"])
def test_search_end_geometries(ctx, geom):
    """Geometries in which the library ends a search loop early (steep error curves), cannot end it
    early (coordinates of millions of metres: the rounding bound exceeds the slope), and oblique
    views: the result is the reference's bit for bit in all of them."""
    rng = np.random.default_rng(123)
    A = [[10000, 0, 4096], [0, 10000, 4096], [0, 0, 1]]
    def rot(ax, ay, az):
        cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    if geom == "short_baseline":
        R1 = R2 = np.eye(3)
        C1, C2 = np.array([0.0, 0.0, 3000.0]), np.array([60.0, 0.0, 3000.0])
        dlo, dhi = 150, 260                       # ground at 2300 .. 4000 m below the cameras
    elif geom == "utm_coordinates":
        R1 = R2 = np.eye(3)
        C1, C2 = np.array([512345.5, 4123456.25, 3000.0]), np.array([513245.5, 4123456.25, 3010.0])
        dlo, dhi = 2500, 3500
    else:
        R1, R2 = rot(0.3, -0.2, 0.1), rot(0.25, -0.15, 0.05)
        C1, C2 = np.array([0.0, 0.0, 2500.0]), np.array([400.0, 30.0, 2550.0])
        dlo, dhi = 800, 2400
    c1, c2 = (A, R1, -R1 @ C1), (A, R2, -R2 @ C2)
    n = 60_000
    u1 = rng.integers(0, 8192, n); v1 = rng.integers(0, 8192, n)
    pts = np.stack([u1, v1, u1 - rng.integers(dlo, dhi + 1, n), v1 + rng.integers(-3, 4, n)], axis=1).astype(np.int32)
    got = ctx.calculate_xyz(mistereo.make_camera(*c1), mistereo.make_camera(*c2), pts)
    want = oracle.triangulate(oracle.make_camera(*c1), oracle.make_camera(*c2), pts)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
