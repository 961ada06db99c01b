"""libmistereo against the committed golden vectors (independent numpy restatement)."""
import os

import numpy as np
import pytest

import mistereo
from golden_util import GOLDEN, load_cases
from ncc_util import compare_ncc, flat_mask

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["ncc_u8_w7", "ncc_u8_w15", "ncc_u8_w21"])
def test_ncc_golden(ctx, case):
    c = load_cases("ncc_cases.npz")[case]
    win, dmin, dmax = int(c["window"]), int(c["dmin"]), int(c["dmax"])
    ctx.upload_pair(c["left"], c["right"])
    gpu = ctx.ncc_dense(win, dmin, dmax, want_peak2=True)
    ora = {k: c[k] for k in ("idisp", "disp", "peak", "peak2")}
    ora["flat"] = flat_mask(c["volume"], c["idisp"], dmin)
    compare_ncc(gpu, ora, volume=c["volume"], dmin=dmin, check_peak2=True)


def test_ncc_golden_u16(ctx):
    c = load_cases("ncc_cases.npz")["ncc_u16_w15"]
    win, dmin, dmax = int(c["window"]), int(c["dmin"]), int(c["dmax"])
    ctx.upload_pair(c["left"], c["right"])
    gpu = ctx.ncc_dense(win, dmin, dmax)
    ora = {k: c[k] for k in ("idisp", "disp", "peak")}
    ora["flat"] = flat_mask(c["volume"], c["idisp"], dmin)
    compare_ncc(gpu, ora, volume=c["volume"], dmin=dmin)


@pytest.mark.parametrize("case", ["xc32", "xc64", "xc128", "impulse", "padded"])
def test_find_offset_golden(ctx, case):
    c = load_cases("xcorr_cases.npz")[case]
    size = int(c["size"])
    b1 = tuple(int(v) for v in c["b1"]) if "b1" in c else (0, 0, size, size)
    b2 = tuple(int(v) for v in c["b2"]) if "b2" in c else (0, 0, size, size)
    ctx.upload_pair(c["img1"], c["img2"])
    peak, dx, dy = ctx.find_offset(b1, b2, size, size)
    assert (dx, dy) == (int(c["dx"]), int(c["dy"]))
    assert abs(peak - float(c["peak"])) <= 1e-4 * max(abs(float(c["peak"])), 1e-3) + 1e-5
    surf, dog = ctx.tile_surfaces(b1, b2, size, size)
    assert np.abs(surf - c["surface"]).max() <= 3e-5
    assert np.abs(dog - c["dog"]).max() <= 1e-4


def test_fft_golden(ctx):
    z = np.load(os.path.join(GOLDEN, "fft_case.npz"))
    got = ctx.fft2d_forward(z["img"])
    assert np.linalg.norm(got - z["spectrum"]) / np.linalg.norm(z["spectrum"]) <= 1e-5


def test_triangulation_golden_bit_exact(ctx):
    z = np.load(os.path.join(GOLDEN, "tri_cases.npz"))
    for i in range(2):
        c1 = mistereo.make_camera(z["A1_%d" % i], z["R1_%d" % i], z["T1_%d" % i])
        c2 = mistereo.make_camera(z["A2_%d" % i], z["R2_%d" % i], z["T2_%d" % i])
        got = ctx.calculate_xyz(c1, c2, z["pts"])
        assert np.array_equal(got.view(np.uint32), z["xyz_%d" % i].view(np.uint32))


@pytest.mark.parametrize("case", ["g48x40", "g45x31", "g64x20"])
def test_find_offset_golden_general_sizes(ctx, case):
    c = load_cases("xcorr_general_cases.npz")[case]
    w, h = int(c["w"]), int(c["h"])
    ctx.upload_pair(c["img1"], c["img2"])
    peak, dx, dy = ctx.find_offset((0, 0, w, h), (0, 0, w, h), w, h)
    assert (dx, dy) == (int(c["dx"]), int(c["dy"]))
    assert abs(peak - float(c["peak"])) <= 1e-4 * max(abs(float(c["peak"])), 1e-3) + 1e-5
    surf, dog = ctx.tile_surfaces((0, 0, w, h), (0, 0, w, h), w, h)
    assert np.abs(surf - c["surface"]).max() <= 3e-5
    assert np.abs(dog - c["dog"]).max() <= 1e-4


def test_band_product_and_masked_peak_golden(ctx):
    z = np.load(os.path.join(GOLDEN, "bands_cases.npz"))
    b1, b2 = list(z["bands1"]), list(z["bands2"])
    ctx.upload_pair(b1[0], b2[0])
    ctx.upload_extra_bands(b1[1:], b2[1:])
    peak, dx, dy = ctx.find_offset((0, 0, 32, 32), (0, 0, 32, 32), 32, 32)
    surf, dog = ctx.tile_surfaces((0, 0, 32, 32), (0, 0, 32, 32), 32, 32)
    assert (dx, dy) == (int(z["dx"]), int(z["dy"]))
    assert abs(peak - float(z["peak"])) <= 1e-4 * max(abs(float(z["peak"])), 1e-3) + 1e-5
    assert np.abs(surf - z["surface"]).max() <= 5e-5
    assert np.abs(dog - z["dog"]).max() <= 1e-4
    for i in range(4):
        pk, ox, oy = ctx.masked_peak(z["mp%d__bands" % i], float(z["mp%d__pct" % i]))
        want = z["mp%d__result" % i]
        assert (np.float32(pk), ox, oy) == (np.float32(want[0]), int(want[1]), int(want[2]))
