"""Shared checks for dense-NCC parity tests (GPU library vs CPU oracle)."""
import numpy as np

# scores closer than this (in NCC units) are a declared near-tie: float32 scoring on the GPU and
# float64 scoring in the oracle may order them differently
NEAR_TIE = 2e-5
PEAK_RTOL = 1e-4      # correlation peaks within 1e-4 relative
SUBPX_ATOL = 0.01     # subpixel disparity within 0.01 px
ARGMAX_MIN_AGREE = 0.999


def compare_ncc(gpu, ora, volume=None, dmin=None, check_peak2=False):
    """gpu / ora: dicts with disp, peak, idisp (and peak2).  volume: oracle score volume
    [H][W][nd] (optional) used to prove that argmax differences are near-ties."""
    gi, oi = gpu["idisp"].astype(np.int32), ora["idisp"].astype(np.int32)
    valid = oi != -32768
    assert np.array_equal(gi == -32768, ~valid), "validity masks differ"
    nvalid = int(valid.sum())
    if nvalid == 0:
        return {"valid": 0, "agree": 1.0}
    same = (gi == oi) & valid
    agree = same.sum() / nvalid
    assert agree >= ARGMAX_MIN_AGREE, "integer argmax agrees on only %.5f of pixels" % agree
    diff = valid & ~same
    if diff.any():
        assert volume is not None or diff.sum() <= (1 - ARGMAX_MIN_AGREE) * nvalid
        if volume is not None:
            ys, xs = np.nonzero(diff)
            sg = volume[ys, xs, gi[ys, xs] - dmin]
            so = volume[ys, xs, oi[ys, xs] - dmin]
            assert np.all(np.abs(sg - so) <= NEAR_TIE), "argmax differs outside a near-tie: max gap %g" % np.abs(sg - so).max()
    # peaks and subpixel where the integer argmax agrees
    gp, op = gpu["peak"][same], ora["peak"][same]
    assert np.all(np.abs(gp - op) <= PEAK_RTOL * np.maximum(np.abs(op), 1e-3) + 2e-6), \
        "peak mismatch: max abs %g" % np.abs(gp - op).max()
    # subpixel: NO pixel is excluded from the comparison.  Every outlier is then classified, as oracle.parity_report
    # does for the benchmark line: it must be an ill-conditioned parabola (denominator |s- - 2 s0 + s+| below 1e-4 in
    # the oracle's float64 volume, where float32 noise of the three scores moves the vertex); the count is returned.
    gd, od = gpu["disp"][same], ora["disp"][same]
    bad = np.abs(gd - od) > SUBPX_ATOL
    if bad.any():
        assert "flat" in ora, "subpixel mismatch on %d pixels: max %g (no volume to classify them)" % (bad.sum(), np.abs(gd - od).max())
        well = bad & ~ora["flat"][same]
        assert not well.any(), "subpixel mismatch on %d well-conditioned pixels: max %g" % (well.sum(), np.abs(gd - od)[well].max())
    assert np.array_equal(gpu["disp"][~valid], ora["disp"][~valid])
    assert np.array_equal(gpu["peak"][~valid], ora["peak"][~valid])
    if check_peak2:
        g2, o2 = gpu["peak2"][same], ora["peak2"][same]
        assert np.all(np.abs(g2 - o2) <= PEAK_RTOL * np.maximum(np.abs(o2), 1e-3) + 2e-6), \
            "peak2 mismatch: max abs %g" % np.abs(g2 - o2).max()
    return {"valid": nvalid, "agree": float(agree), "subpx_outliers_ill_conditioned": int(bad.sum())}


def flat_mask(volume, idisp, dmin, eps=1e-4):
    """Conditioning class of every pixel's parabola (the one oracle.parity_report uses): True where the denominator is
    below eps, so the subpixel vertex is ill-conditioned.  compare_ncc uses it to classify outliers, not to skip pixels."""
    h, w, nd = volume.shape
    k = np.clip(idisp.astype(np.int64) - dmin, 0, nd - 1)
    yy, xx = np.mgrid[0:h, 0:w]
    s0 = volume[yy, xx, k]
    sm = volume[yy, xx, np.clip(k - 1, 0, nd - 1)]
    sp = volume[yy, xx, np.clip(k + 1, 0, nd - 1)]
    den = np.abs(sm - 2 * s0 + sp)
    return ~(den > eps)   # NaN neighbours count as flat
