"""The C oracle against the committed golden vectors (tests/golden/, made by make_golden.py from an
independent numpy restatement).  Runs without a GPU.  Parity with the reference itself is unpinned:
the reference ships no vectors and cannot be built here."""
import os

import numpy as np
import pytest

import oracle
from golden_util import GOLDEN, load_cases
from ncc_util import compare_ncc, flat_mask


@pytest.mark.parametrize("case", ["ncc_u8_w7", "ncc_u8_w15", "ncc_u8_w21", "ncc_u16_w15"])
def test_ncc_oracle_matches_numpy_definition(case):
    c = load_cases("ncc_cases.npz")[case]
    win, dmin, dmax = int(c["window"]), int(c["dmin"]), int(c["dmax"])
    o = oracle.ncc_dense(c["left"], c["right"], win, dmin, dmax, want_peak2=True, want_volume=True)
    both_nan = np.isnan(o["volume"]) & np.isnan(c["volume"])
    assert np.array_equal(np.isnan(o["volume"]), np.isnan(c["volume"]))
    assert np.abs(np.where(both_nan, 0, o["volume"] - c["volume"])).max() <= 1e-12
    assert np.array_equal(o["idisp"], c["idisp"])
    assert np.abs(o["peak"] - c["peak"]).max() <= 1e-6
    assert np.abs(o["peak2"] - c["peak2"]).max() <= 1e-6
    flat = flat_mask(c["volume"], c["idisp"], dmin, eps=1e-9)
    assert np.abs(o["disp"] - c["disp"])[~flat].max() <= 1e-5


def test_ncc_score_pointwise():
    c = load_cases("ncc_cases.npz")["ncc_u8_w15"]
    win, dmin = int(c["window"]), int(c["dmin"])
    rng = np.random.default_rng(0)
    h, w = c["left"].shape
    for _ in range(200):
        x, y, k = int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(c["volume"].shape[2]))
        s = oracle.ncc_score(c["left"], c["right"], win, x, y, dmin + k)
        g = c["volume"][y, x, k]
        assert (np.isnan(s) and np.isnan(g)) or abs(s - g) <= 1e-12


@pytest.mark.parametrize("case", ["xc32", "xc64", "xc128", "impulse", "padded"])
@pytest.mark.parametrize("precision", [0, 1])
def test_find_offset_oracle(case, precision):
    c = load_cases("xcorr_cases.npz")[case]
    size = int(c["size"])
    b1 = tuple(int(v) for v in c["b1"]) if "b1" in c else (0, 0, size, size)
    b2 = tuple(int(v) for v in c["b2"]) if "b2" in c else (0, 0, size, size)
    o = oracle.find_offset(c["img1"], c["img2"], b1, b2, size, size, precision=precision,
                           want_surface=True, want_dog=True)
    assert (o["dx"], o["dy"]) == (int(c["dx"]), int(c["dy"]))
    tol = 2e-5 if precision == 0 else 2e-6
    assert abs(o["peak"] - float(c["peak"])) <= tol
    assert np.abs(o["surface"] - c["surface"]).max() <= tol
    assert np.abs(o["dog"] - c["dog"]).max() <= 5 * tol


def test_fft_oracle():
    z = np.load(os.path.join(GOLDEN, "fft_case.npz"))
    got = oracle.fft2d_forward(z["img"])
    assert np.linalg.norm(got - z["spectrum"]) / np.linalg.norm(z["spectrum"]) <= 2e-6
    rng = np.random.default_rng(1)
    for shape in [(8, 16), (12, 20), (7, 9), (64, 64)]:       # powers of two and Bluestein sizes
        a = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
        assert np.abs(oracle.fft2d(a, -1) - np.fft.fft2(a)).max() <= 1e-10
        assert np.abs(oracle.fft2d(a, +1) - np.fft.ifft2(a) * a.size).max() <= 1e-10
        f = oracle.fft2d(a, -1, double=False)
        assert np.linalg.norm(f - np.fft.fft2(a)) / np.linalg.norm(np.fft.fft2(a)) <= 1e-5


def test_gaussian_blur_oracle_multi_reflection():
    z = np.load(os.path.join(GOLDEN, "blur_case.npz"))
    got = oracle.gaussian_blur(z["src"])
    assert np.abs(got - z["dst"]).max() <= 1e-5
    one = oracle.gaussian_blur(np.ones((5, 1), np.float32))    # a 1-px dimension collapses the kernel
    assert np.allclose(one, 1.0, atol=1e-6)


def test_fft2shift_conventions():
    a = np.arange(6 * 8, dtype=np.float32).reshape(6, 8)
    assert np.array_equal(oracle.fft2shift(a), np.fft.fftshift(a))      # even sizes: plain fftshift
    b = (np.arange(4 * 6) % 251).astype(np.uint8).reshape(4, 6)
    assert np.array_equal(oracle.fft2shift(b), np.fft.fftshift(b))


def test_xcorr_surface_u8_oracle():
    c = load_cases("xcorr_cases.npz")["xc64"]
    s = oracle.xcorr_surface_u8(c["img1"], c["img2"], precision=1).astype(np.int32)
    want = np.fft.fftshift(np.floor(c["surface"].astype(np.float64) * 255.0)).astype(np.int32)
    d = np.abs(s - want)
    assert d.max() <= 1 and (d > 0).mean() < 0.02


def test_triangulation_oracle_bit_exact():
    z = np.load(os.path.join(GOLDEN, "tri_cases.npz"))
    for i in range(2):
        c1 = oracle.make_camera(z["A1_%d" % i], z["R1_%d" % i], z["T1_%d" % i])
        c2 = oracle.make_camera(z["A2_%d" % i], z["R2_%d" % i], z["T2_%d" % i])
        got = oracle.triangulate(c1, c2, z["pts"])
        assert np.array_equal(got.view(np.uint32), z["xyz_%d" % i].view(np.uint32))


def test_triangulation_oracle_float_coordinates_equal_the_integer_form():
    """orc_triangulate_f is the same arithmetic on coordinates that arrive as floats: on integer values it
    reproduces the golden vectors of the integer form bit for bit."""
    z = np.load(os.path.join(GOLDEN, "tri_cases.npz"))
    for i in range(2):
        c1 = oracle.make_camera(z["A1_%d" % i], z["R1_%d" % i], z["T1_%d" % i])
        c2 = oracle.make_camera(z["A2_%d" % i], z["R2_%d" % i], z["T2_%d" % i])
        got = oracle.triangulate_f(c1, c2, z["pts"].astype(np.float32))
        assert np.array_equal(got.view(np.uint32), z["xyz_%d" % i].view(np.uint32))


def test_disparity_map_recursion_counts():
    """Zero-shift 256-px pair, min_size 32: levels 256, 128, 64 -> 1 + 4 + 16 findOffset calls,
    16 leaves (image3.cpp:751-774)."""
    import synth
    img = synth.texture(256, 256, 3)
    leaves, calls = oracle.disparity_map(img, img, min_size=32, precision=1)
    assert calls == 21 and len(leaves) == 16
    assert all(l[4] == 0 and l[5] == 0 for l in leaves)
    assert sorted((l[0], l[1]) for l in leaves) == sorted((i, j) for j in range(0, 256, 64) for i in range(0, 256, 64))


@pytest.mark.parametrize("case", ["g48x40", "g45x31", "g64x20"])
def test_find_offset_oracle_general_sizes(case):
    c = load_cases("xcorr_general_cases.npz")[case]
    w, h = int(c["w"]), int(c["h"])
    o = oracle.find_offset(c["img1"], c["img2"], (0, 0, w, h), (0, 0, w, h), w, h, precision=1,
                           want_surface=True, want_dog=True)
    assert (o["dx"], o["dy"]) == (int(c["dx"]), int(c["dy"]))
    assert abs(o["peak"] - float(c["peak"])) <= 2e-6
    assert np.abs(o["surface"] - c["surface"]).max() <= 2e-6
    assert np.abs(o["dog"] - c["dog"]).max() <= 1e-5


def test_band_product_and_masked_peak_oracle():
    z = np.load(os.path.join(GOLDEN, "bands_cases.npz"))
    b1, b2 = list(z["bands1"]), list(z["bands2"])
    o = oracle.find_offset_bands(b1, b2, (0, 0, 32, 32), (0, 0, 32, 32), 32, 32, precision=1,
                                 want_surface=True, want_dog=True)
    assert (o["dx"], o["dy"]) == (int(z["dx"]), int(z["dy"]))
    assert np.abs(o["surface"] - z["surface"]).max() <= 5e-6
    assert np.abs(o["dog"] - z["dog"]).max() <= 2e-5
    for i in range(4):
        pk, ox, oy = oracle.masked_peak(z["mp%d__bands" % i], float(z["mp%d__pct" % i]))
        want = z["mp%d__result" % i]
        assert (np.float32(pk), ox, oy) == (np.float32(want[0]), int(want[1]), int(want[2]))
