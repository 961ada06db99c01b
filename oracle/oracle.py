"""ctypes binding of the CPU oracle (oracle/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product path (image-processing_amd/) never does.  PARITY UNPINNED: see oracle/oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

U8, U16, F32 = 0, 1, 2
INVALID_DISP = -32768


class Rect(C.Structure):
    _fields_ = [("minx", C.c_int), ("miny", C.c_int), ("width", C.c_int), ("height", C.c_int)]


class Offset(C.Structure):
    _fields_ = [("dx", C.c_int), ("dy", C.c_int), ("peak", C.c_float), ("dog_peak", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("A", C.c_float * 9), ("R", C.c_float * 9), ("T", C.c_float * 3)]


class Leaf(C.Structure):
    _fields_ = [("bbox1", Rect), ("total_dx", C.c_int), ("total_dy", C.c_int),
                ("u1", C.c_int), ("v1", C.c_int), ("u2", C.c_int), ("v2", C.c_int),
                ("peak", C.c_float)]


def build(force=False):
    """Compile liboracle.so from the C sources next to this file."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_ncc_score.restype = C.c_double
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _dtype_code(a):
    if a.dtype == np.uint8:
        return U8
    if a.dtype == np.uint16:
        return U16
    if a.dtype == np.float32:
        return F32
    raise TypeError("unsupported pixel type %s" % a.dtype)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_camera(A, R, T):
    cam = Camera()
    cam.A[:] = [float(v) for v in np.asarray(A, np.float32).reshape(9)]
    cam.R[:] = [float(v) for v in np.asarray(R, np.float32).reshape(9)]
    cam.T[:] = [float(v) for v in np.asarray(T, np.float32).reshape(3)]
    return cam


def ncc_dense(left, right, window, dmin, dmax, want_peak2=False, want_volume=False):
    left = np.ascontiguousarray(left)
    right = np.ascontiguousarray(right)
    assert left.shape == right.shape and left.dtype == right.dtype
    h, w = left.shape
    disp = np.empty((h, w), np.float32)
    peak = np.empty((h, w), np.float32)
    idisp = np.empty((h, w), np.int16)
    peak2 = np.empty((h, w), np.float32) if want_peak2 else None
    vol = np.empty((h, w, dmax - dmin + 1), np.float64) if want_volume else None
    rc = lib().orc_ncc_dense(_ptr(left), _ptr(right), _dtype_code(left), w, h,
                             C.c_size_t(left.strides[0]), window, dmin, dmax,
                             _ptr(disp), _ptr(peak), _ptr(idisp), _ptr(peak2), _ptr(vol))
    if rc != 0:
        raise ValueError("orc_ncc_dense rejected its arguments")
    out = {"disp": disp, "peak": peak, "idisp": idisp}
    if want_peak2:
        out["peak2"] = peak2
    if want_volume:
        out["volume"] = vol
    return out


def ncc_score(left, right, window, x, y, d):
    left = np.ascontiguousarray(left)
    right = np.ascontiguousarray(right)
    h, w = left.shape
    return lib().orc_ncc_score(_ptr(left), _ptr(right), _dtype_code(left), w, h,
                               C.c_size_t(left.strides[0]), window, int(x), int(y), int(d))


def ncc_scores(left, right, window, xs, ys, ds):
    """float64 NCC of the window at (x, y) against the window at (x - d, y), for arrays of triples."""
    left = np.ascontiguousarray(left)
    right = np.ascontiguousarray(right)
    h, w = left.shape
    xs = np.ascontiguousarray(xs, np.int32); ys = np.ascontiguousarray(ys, np.int32); ds = np.ascontiguousarray(ds, np.int32)
    out = np.empty(xs.shape[0], np.float64)
    lib().orc_ncc_scores(_ptr(left), _ptr(right), _dtype_code(left), w, h, C.c_size_t(left.strides[0]), window,
                         xs.shape[0], _ptr(xs), _ptr(ys), _ptr(ds), _ptr(out))
    return out


def parity_report(gpu_disp, gpu_peak, ora, left, right, window, dmin, dmax, gpu_idisp=None, near_tie=2e-5,
                  peak_rtol=1e-4, subpx_atol=0.01, flat_eps=1e-4):
    """The three parity bars of the dense NCC, each on its own, for a GPU result (disp, peak rasters)
    against this oracle's result `ora` on the same rasters:
      argmax_agree                  fraction of valid pixels with the same integer disparity
      argmax_diffs_all_near_tie     every differing pixel's two disparities score within `near_tie` of
                                    each other (float64 scores of both, recomputed for those pixels only)
      peak_within_1e-4              fraction of agreeing pixels whose peaks agree to 1e-4 relative
      subpx_within_0.01_on_agreeing fraction of agreeing pixels whose subpixel disparities agree to 0.01 px,
                                    with NO pixel excluded; the outliers are then classified: a parabola
                                    whose denominator |s- - 2 s0 + s+| is below `flat_eps` is ill-conditioned
                                    (float32 noise of the three scores moves its vertex)."""
    oi = ora["idisp"].astype(np.int32)
    if gpu_idisp is not None:
        gi = gpu_idisp.astype(np.int32)
    else:
        # the integer disparity is the subpixel value rounded (|delta| <= 0.5); a value at .5 is resolved
        # with the oracle's integer when that is one of the two candidates
        gi = np.where(gpu_disp == np.float32(INVALID_DISP), INVALID_DISP, np.rint(gpu_disp)).astype(np.int32)
        lo, hi = np.floor(gpu_disp), np.ceil(gpu_disp)
        half = (oi != INVALID_DISP) & (np.abs(gpu_disp - lo - 0.5) < 1e-6)
        gi = np.where(half & ((lo == oi) | (hi == oi)), oi, gi)
    valid = oi != INVALID_DISP
    nvalid = int(valid.sum())
    rep = {"valid_pixels": nvalid, "validity_masks_equal": bool(np.array_equal(gi == INVALID_DISP, ~valid))}
    if nvalid == 0:
        return rep
    same = valid & (gi == oi)
    diff = valid & ~same
    rep["argmax_agree"] = float(same.sum() / nvalid)
    rep["argmax_diffs"] = int(diff.sum())
    if diff.any():
        ys, xs = np.nonzero(diff)
        sg = ncc_scores(left, right, window, xs, ys, gi[ys, xs])
        so = ncc_scores(left, right, window, xs, ys, oi[ys, xs])
        gap = np.abs(sg - so)
        rep["argmax_diffs_all_near_tie"] = bool(np.all(gap <= near_tie))
        rep["argmax_diffs_max_score_gap"] = float(np.nanmax(gap))
    else:
        rep["argmax_diffs_all_near_tie"] = True
        rep["argmax_diffs_max_score_gap"] = 0.0
    gp, op = gpu_peak[same], ora["peak"][same]
    rep["peak_within_1e-4"] = float((np.abs(gp - op) <= peak_rtol * np.maximum(np.abs(op), 1e-3) + 2e-6).mean())
    rep["peak_max_abs_diff"] = float(np.abs(gp - op).max())
    bad = same & (np.abs(gpu_disp - ora["disp"]) > subpx_atol)
    rep["subpx_within_0.01_on_agreeing"] = float(1.0 - bad.sum() / max(int(same.sum()), 1))
    rep["subpx_outliers"] = int(bad.sum())
    if bad.any():
        ys, xs = np.nonzero(bad)
        d0 = oi[ys, xs]
        s0 = ncc_scores(left, right, window, xs, ys, d0)
        sm = ncc_scores(left, right, window, xs, ys, d0 - 1)
        sp = ncc_scores(left, right, window, xs, ys, d0 + 1)
        den = np.abs(sm - 2 * s0 + sp)
        rep["subpx_outliers_ill_conditioned"] = int((~(den > flat_eps)).sum())
        rep["subpx_outliers_max_diff_well_conditioned"] = float(np.abs(gpu_disp - ora["disp"])[ys, xs][den > flat_eps].max()) \
            if (den > flat_eps).any() else 0.0
    else:
        rep["subpx_outliers_ill_conditioned"] = 0
        rep["subpx_outliers_max_diff_well_conditioned"] = 0.0
    return rep


def fft2d(data, sign, double=True):
    """In-place-style 2-D DFT of a complex array (unnormalised); returns a new array."""
    if double:
        a = np.ascontiguousarray(data, np.complex128).copy()
        lib().orc_fft2d_f64(_ptr(a), a.shape[1], a.shape[0], int(sign))
    else:
        a = np.ascontiguousarray(data, np.complex64).copy()
        lib().orc_fft2d_f32(_ptr(a), a.shape[1], a.shape[0], int(sign))
    return a


def fft2d_forward(img):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    out = np.empty((h, w), np.complex64)
    lib().orc_fft2d_forward(_ptr(img), _dtype_code(img), w, h, C.c_size_t(img.strides[0]), _ptr(out))
    return out


def find_offset(img1, img2, b1, b2, width, height, precision=0, want_surface=False, want_dog=False):
    img1 = np.ascontiguousarray(img1)
    img2 = np.ascontiguousarray(img2)
    assert img1.dtype == img2.dtype
    r1, r2 = Rect(*b1), Rect(*b2)
    off = Offset()
    surf = np.empty((height, width), np.float32) if want_surface else None
    dog = np.empty((height, width), np.float32) if want_dog else None
    lib().orc_find_offset(_ptr(img1), img1.shape[1], img1.shape[0], C.c_size_t(img1.strides[0]),
                          _ptr(img2), img2.shape[1], img2.shape[0], C.c_size_t(img2.strides[0]),
                          _dtype_code(img1), r1, r2, width, height, precision, C.byref(off),
                          _ptr(surf), _ptr(dog))
    res = {"dx": off.dx, "dy": off.dy, "peak": off.peak, "dog_peak": off.dog_peak}
    if want_surface:
        res["surface"] = surf
    if want_dog:
        res["dog"] = dog
    return res


def find_offset_bands(bands1, bands2, b1, b2, width, height, precision=0, want_surface=False, want_dog=False):
    """bands1 / bands2: lists of equally shaped rasters (one per band)."""
    bands1 = [np.ascontiguousarray(b) for b in bands1]
    bands2 = [np.ascontiguousarray(b) for b in bands2]
    nb = len(bands1)
    assert nb == len(bands2) and nb >= 1
    p1 = (C.c_void_p * nb)(*[b.ctypes.data for b in bands1])
    p2 = (C.c_void_p * nb)(*[b.ctypes.data for b in bands2])
    i1, i2 = bands1[0], bands2[0]
    r1, r2 = Rect(*b1), Rect(*b2)
    off = Offset()
    surf = np.empty((height, width), np.float32) if want_surface else None
    dog = np.empty((height, width), np.float32) if want_dog else None
    lib().orc_find_offset_bands(p1, i1.shape[1], i1.shape[0], C.c_size_t(i1.strides[0]),
                                p2, i2.shape[1], i2.shape[0], C.c_size_t(i2.strides[0]),
                                _dtype_code(i1), nb, r1, r2, width, height, precision, C.byref(off),
                                _ptr(surf), _ptr(dog))
    res = {"dx": off.dx, "dy": off.dy, "peak": off.peak, "dog_peak": off.dog_peak}
    if want_surface:
        res["surface"] = surf
    if want_dog:
        res["dog"] = dog
    return res


def masked_peak(bands, mask_pct):
    """findCPPMax: bands [n, rows, cols] float32 -> (peak, offset_x, offset_y)."""
    bands = np.ascontiguousarray(bands, np.float32)
    if bands.ndim == 2:
        bands = bands[None]
    nb, rows, cols = bands.shape
    ox, oy = C.c_int(0), C.c_int(0)
    f = lib().orc_masked_peak
    f.restype = C.c_float
    peak = f(_ptr(bands), cols, rows, nb, C.c_float(mask_pct), C.byref(ox), C.byref(oy))
    return float(peak), ox.value, oy.value


def gaussian_blur(src, ksize=151, sigma=99.0):
    src = np.ascontiguousarray(src, np.float32)
    dst = np.empty_like(src)
    lib().orc_gaussian_blur(_ptr(src), _ptr(dst), src.shape[1], src.shape[0], ksize, C.c_double(sigma))
    return dst


def xcorr_surface_u8(img1, img2, precision=0):
    img1 = np.ascontiguousarray(img1)
    img2 = np.ascontiguousarray(img2)
    h, w = img1.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_xcorr_surface_u8(_ptr(img1), _ptr(img2), _dtype_code(img1), w, h,
                               C.c_size_t(img1.strides[0]), precision, _ptr(out))
    return out


def fft2shift(a):
    a = np.ascontiguousarray(a).copy()
    if a.dtype == np.float32:
        lib().orc_fft2shift_f32(_ptr(a), a.shape[1], a.shape[0])
    elif a.dtype == np.uint8:
        lib().orc_fft2shift_u8(_ptr(a), a.shape[0], a.shape[1])
    else:
        raise TypeError(a.dtype)
    return a


def triangulate(cam1, cam2, pts):
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 4)
    out = np.empty((pts.shape[0], 3), np.float32)
    lib().orc_triangulate(C.byref(cam1), C.byref(cam2), _ptr(pts), pts.shape[0], _ptr(out))
    return out


def triangulate_f(cam1, cam2, pts):
    """float pixel coordinates (u1, v1, u2, v2); equals triangulate() on integer values"""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 4)
    out = np.empty((pts.shape[0], 3), np.float32)
    lib().orc_triangulate_f(C.byref(cam1), C.byref(cam2), _ptr(pts), pts.shape[0], _ptr(out))
    return out


def disparity_map(img1, img2, min_size=32, precision=0, max_leaves=1 << 16):
    img1 = np.ascontiguousarray(img1)
    img2 = np.ascontiguousarray(img2)
    leaves = (Leaf * max_leaves)()
    ncalls = C.c_int(0)
    # rasters of different size are allowed (image3.cpp:995-998 only warns)
    n = lib().orc_disparity_map2(_ptr(img1), img1.shape[1], img1.shape[0], C.c_size_t(img1.strides[0]),
                                 _ptr(img2), img2.shape[1], img2.shape[0], C.c_size_t(img2.strides[0]), _dtype_code(img1),
                                 min_size, precision, leaves, max_leaves, C.byref(ncalls))
    if n < 0:
        raise RuntimeError("no overlap")
    recs = [(l.bbox1.minx, l.bbox1.miny, l.bbox1.width, l.bbox1.height, l.total_dx, l.total_dy,
             l.u1, l.v1, l.u2, l.v2, l.peak) for l in leaves[:min(n, max_leaves)]]
    return recs, ncalls.value


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    """OpenMP threads for oracle calls made from the calling thread (1: serial)."""
    lib().orc_set_num_threads(int(n))
