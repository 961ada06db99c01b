"""The CPU oracle against implementations neither this repository nor the reference wrote: scipy.ndimage
(border handling and window sums), scipy.signal (direct circular correlation) and numpy.fft.  The reference
holds no vectors for any of this (its parity stays unpinned); these checks only take the author of the
oracle out of the loop for the published semantics the oracle restates:
  * cv::GaussianBlur with BORDER_REFLECT_101 (image3.cpp:481) == correlate1d(mode='mirror') per axis,
  * the correlation of image3.cpp:383-405 == a direct (FFT-free) circular cross-correlation,
  * zero-mean normalised correlation of windows == box-filter moments."""
import numpy as np
import pytest

import oracle
import synth

ndimage = pytest.importorskip("scipy.ndimage")
signal = pytest.importorskip("scipy.signal")


def _gauss151():
    i = np.arange(151, dtype=np.float64)
    k = np.exp(-(i - 75.0) ** 2 / (2.0 * 99.0 ** 2))
    return k / k.sum()


@pytest.mark.parametrize("shape", [(200, 180), (97, 310), (40, 30), (8, 12), (300, 1)])
def test_gaussian_blur_against_scipy_mirror(shape):
    """Tiles narrower than 76 px reflect several times; a 1-pixel dimension collapses that kernel to one tap
    (OpenCV's createSeparableLinearFilter shrinks ksize to 1 there)."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    src = rng.random(shape).astype(np.float32)
    k = _gauss151()
    want = src.astype(np.float64)
    if shape[1] > 1:
        want = ndimage.correlate1d(want, k, axis=1, mode="mirror")
    if shape[0] > 1:
        want = ndimage.correlate1d(want, k, axis=0, mode="mirror")
    got = oracle.gaussian_blur(src)
    assert np.abs(got - want).max() <= 2e-6


@pytest.mark.parametrize("size,shift", [(16, (3, -2)), (32, (-7, 5)), (24, (0, 0))])
def test_correlation_surface_against_direct_circular_correlation(size, shift):
    """c[k] = (1/sqrt N) sum_n a[n+k] b[n] over circular shifts (image3.cpp:383-405), |c| min-max normalised
    (image3.cpp:402-416): scipy.signal.correlate2d with wrapped boundaries computes the same sums directly."""
    rng = np.random.default_rng(size)
    b = (rng.random((size, size)) * 255).astype(np.uint8)
    a = np.roll(b, shift, axis=(0, 1))                      # content of b at p appears in a at p + shift
    res = oracle.find_offset(a, b, (0, 0, size, size), (0, 0, size, size), size, size, precision=1, want_surface=True)
    full = signal.correlate2d(np.tile(a.astype(np.float64), (2, 2)), b.astype(np.float64), mode="valid")
    # both forward transforms and the backward one are scaled by 1/sqrt(N) (image3.cpp:333, 403): c = sum / sqrt(N);
    # the min-max normalisation removes any common factor anyway
    c = np.abs(full[:size, :size]) / size
    want = (c - c.min()) / (c.max() - c.min())
    assert np.abs(res["surface"] - want).max() <= 1e-6
    if size >= 32:      # the DoG peak of a clean shift sits on the correlation peak (tiny tiles are all border)
        assert (res["dy"], res["dx"]) == shift


@pytest.mark.parametrize("dtype,window", [(np.uint8, 5), (np.uint8, 21), (np.uint16, 31)])
def test_dense_ncc_volume_against_box_filter_moments(dtype, window):
    h, w, dmin, dmax = 50, 90, -6, 7
    left = synth.texture(h, w, 9, dtype)
    right = synth.shift_right_image(left, 3, fill_seed=10)
    ora = oracle.ncc_dense(left, right, window, dmin, dmax, want_volume=True)
    L, R = left.astype(np.float64), right.astype(np.float64)
    n = window * window
    box = lambda x: ndimage.uniform_filter(x, size=window, mode="constant") * n
    r = window // 2
    sl, sll = box(L), box(L * L)
    for d in range(dmin, dmax + 1):
        Rd = np.zeros_like(R)                                # Rd[y, x] = R[y, x - d]
        if d >= 0:
            Rd[:, d:] = R[:, :w - d]
        else:
            Rd[:, :w + d] = R[:, -d:]
        sr, srr, slr = box(Rd), box(Rd * Rd), box(L * Rd)
        num = slr - sl * sr / n
        den = np.sqrt(np.maximum(sll - sl * sl / n, 0) * np.maximum(srr - sr * sr / n, 0))
        want = np.where(den > 0, num / np.where(den > 0, den, 1), 0.0)
        xs = np.arange(w)
        ok = np.zeros((h, w), bool)
        ok[r:h - r, :] = True
        ok &= ((xs >= r) & (xs < w - r) & (xs - d >= r) & (xs - d < w - r))[None, :]
        got = ora["volume"][:, :, d - dmin]
        assert np.abs(got[ok] - want[ok]).max() <= 1e-7
