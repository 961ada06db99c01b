#!/usr/bin/env python3
"""Generates the golden vectors in this directory with an independent numpy restatement.

The reference (klassenjs/image-processing) ships no tests, fixtures or sample imagery, and it
cannot be built or imported in the build environment (C++ against GDAL / FFTW3 / OpenCV 2.4; its
only Python file is Python 2 and runs against absent TIFFs).  PARITY IS THEREFORE UNPINNED by the
reference itself: these vectors pin the C oracle (oracle/) against a second, independently
written statement of the same formulas -- numpy.fft in complex128, numpy sliding windows with
exact integer sums, numpy float32 scalar arithmetic -- each citing the reference lines it follows.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "image-processing_amd"))
import synth  # noqa: E402  (synthetic textures only; no product code is involved)

INVALID = -32768


# ------------------------------------------------------------------------------------------
# dense NCC, straight from the definition (dense limit of image3.cpp:775-811, 866-870)
# ------------------------------------------------------------------------------------------
def ncc_dense_numpy(L, R, window, dmin, dmax):
    h, w = L.shape
    r = window // 2
    N = window * window
    nd = dmax - dmin + 1
    Li, Ri = L.astype(np.int64), R.astype(np.int64)
    vol = np.full((h, w, nd), np.nan)
    if h >= window and w >= window:
        wl = sliding_window_view(Li, (window, window))          # [h-2r, w-2r, W, W], centre (y+r, x+r)
        wr = sliding_window_view(Ri, (window, window))
        sl, sll = wl.sum((2, 3)), (wl * wl).sum((2, 3))
        sr, srr = wr.sum((2, 3)), (wr * wr).sum((2, 3))
        nx = w - 2 * r
        for k, d in enumerate(range(dmin, dmax + 1)):
            # left centre column xc pairs with right centre column xc - d; both in [r, w-r)
            lo, hi = max(0, d), min(nx, nx + d)                 # window-array indices of the left centre
            if hi <= lo:
                continue
            a = wl[:, lo:hi]
            b = wr[:, lo - d:hi - d]
            slr = (a * b).sum((2, 3))
            num = N * slr - sl[:, lo:hi] * sr[:, lo - d:hi - d]
            vl = N * sll[:, lo:hi] - sl[:, lo:hi] ** 2
            vr = N * srr[:, lo - d:hi - d] - sr[:, lo - d:hi - d] ** 2
            den = np.sqrt(vl.astype(np.float64) * vr.astype(np.float64))
            sc = np.where((vl > 0) & (vr > 0), num / np.where(den > 0, den, 1), 0.0)
            vol[r:h - r, r + lo:r + hi, k] = sc
    idisp = np.full((h, w), INVALID, np.int16)
    disp = np.full((h, w), float(INVALID), np.float32)
    peak = np.zeros((h, w), np.float32)
    peak2 = np.full((h, w), -1.0, np.float32)
    for y in range(h):
        for x in range(w):
            s = vol[y, x]
            ok = ~np.isnan(s)
            if not ok.any():
                continue
            k = int(np.nanargmax(s))                            # first maximum: lowest d wins ties
            delta = 0.0
            if 0 < k < nd - 1 and ok[k - 1] and ok[k + 1]:
                den = 2.0 * (s[k - 1] - 2.0 * s[k] + s[k + 1])
                if den != 0:
                    delta = min(0.5, max(-0.5, (s[k - 1] - s[k + 1]) / den))
            idisp[y, x] = dmin + k
            disp[y, x] = np.float32(dmin + k + delta)
            peak[y, x] = np.float32(s[k])
            far = ok & (np.abs(np.arange(nd) - k) > 1)
            if far.any():
                peak2[y, x] = np.float32(s[far].max())
    return {"idisp": idisp, "disp": disp, "peak": peak, "peak2": peak2, "volume": vol}


# ------------------------------------------------------------------------------------------
# findOffset (image3.cpp:346-517) with numpy.fft in complex128
# ------------------------------------------------------------------------------------------
def reflect101(p, n):
    if n == 1:
        return 0
    while p < 0 or p >= n:
        p = -p if p < 0 else 2 * n - 2 - p
    return p


def gaussian_kernel(n=151, sigma=99.0):
    x = np.arange(n) - (n - 1) * 0.5
    k = np.exp(-0.5 / (sigma * sigma) * x * x).astype(np.float32)      # stored as float
    return (k * (1.0 / k.astype(np.float64).sum())).astype(np.float32)


def gaussian_blur_numpy(img, ksize=151, sigma=99.0):
    """cv::GaussianBlur(img, Size(151,151), 99) with BORDER_REFLECT_101, in float64 (the taps
    are the float32 taps OpenCV would use; the accumulation order does not matter at this
    precision)."""
    h, w = img.shape
    k = gaussian_kernel(ksize, sigma).astype(np.float64)
    r = ksize // 2
    src = img.astype(np.float64)
    if w > 1:
        idx = np.array([reflect101(i - r, w) for i in range(w + 2 * r)])
        pad = src[:, idx]
        tmp = sum(k[i] * pad[:, i:i + w] for i in range(ksize))
    else:
        tmp = src
    if h > 1:
        idy = np.array([reflect101(i - r, h) for i in range(h + 2 * r)])
        pad = tmp[idy, :]
        out = sum(k[i] * pad[i:i + h, :] for i in range(ksize))
    else:
        out = tmp
    return out


def load_tile(img, bbox, width, height):
    minx, miny, bw, bh = bbox
    t = np.zeros((height, width), np.float64)                          # image3.cpp:322-323
    for y in range(min(bh, height)):
        sy = miny + y
        if 0 <= sy < img.shape[0]:
            x0, x1 = max(minx, 0), min(minx + min(bw, width), img.shape[1])
            if x1 > x0:
                t[y, x0 - minx:x1 - minx] = img[sy, x0:x1]
    return t


def find_offset_numpy(img1, img2, b1, b2, width, height):
    n = width * height
    a = np.fft.fft2(load_tile(img1, b1, width, height)) / np.sqrt(n)   # image3.cpp:329-334
    b = np.fft.fft2(load_tile(img2, b2, width, height)) / np.sqrt(n)
    c = np.fft.ifft2(a * np.conj(b)) * n / np.sqrt(n)                  # unnormalised backward, / sqrt(N)
    m = np.abs(c)
    out = (m - m.min()) / (m.max() - m.min())                          # image3.cpp:402-416
    h2, w2 = (height // 2) * 2, (width // 2) * 2
    imout = np.zeros_like(out)
    imout[:h2, :w2] = out[:h2, :w2]                                    # image3.cpp:471-478
    dog = imout - gaussian_blur_numpy(imout)                           # image3.cpp:481-483
    sub = dog[:h2, :w2]
    k = int(np.argmax(sub))                                            # first maximum, row-major
    row, col = divmod(k, w2)
    peak = out[row, col]
    if col > width // 2:
        col -= width
    if row > height // 2:
        row -= height
    return {"dx": col, "dy": row, "peak": np.float32(peak), "dog_peak": np.float32(sub.max()),
            "surface": out.astype(np.float32), "dog": dog.astype(np.float32)}


# ------------------------------------------------------------------------------------------
# calculateXYZ (image3.cpp:102-132, 148-264) in numpy float32 scalars
# ------------------------------------------------------------------------------------------
F = np.float32


def inv3_cv(R):
    """cv::Mat::inv() on 3x3 CV_32F: cofactors in double, narrowed to float."""
    S = R.astype(np.float64)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
         + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    d = 1.0 / d
    D = np.array([[(S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d, (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d,
                   (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d],
                  [(S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d, (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d,
                   (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d],
                  [(S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d, (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d,
                   (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d]])
    return D.astype(np.float32)


def gemv_cv(A, b, alpha=1.0):
    out = np.zeros(3, np.float32)
    for i in range(3):
        s = 0.0
        for k in range(3):
            s += float(A[i, k]) * float(b[k])
        out[i] = F(s * alpha)
    return out


def xyz2(ang1, t1p, ang2, t2p, z1):
    w1 = (F(z1) * ang1).astype(np.float32) + t1p                       # scaleAdd: float mul, float add
    z2 = F(F(w1[2] - t2p[2]) / ang2[2])
    w2 = (z2 * ang2).astype(np.float32) + t2p
    dx, dy = F(w1[0] - w2[0]), F(w1[1] - w2[1])
    err = F(np.sqrt(F(F(dx * dx) + F(dy * dy))))
    return err, w1


def calculate_xyz_numpy(cam1, cam2, u1, v1, u2, v2):
    (A1, R1, T1), (A2, R2, T2) = cam1, cam2
    R1p, R2p = inv3_cv(R1), inv3_cv(R2)
    t1p, t2p = gemv_cv(R1p, T1, -1.0), gemv_cv(R2p, T2, -1.0)
    a1 = np.array([F(F(u1) - A1[0, 2]) / A1[0, 0], -F(F(v1) - A1[1, 2]) / A1[1, 1], F(-1)], np.float32)
    a2 = np.array([F(F(u2) - A2[0, 2]) / A2[0, 0], -F(F(v2) - A2[1, 2]) / A2[1, 1], F(-1)], np.float32)
    ang1, ang2 = gemv_cv(R1p, a1), gemv_cv(R2p, a2)
    with np.errstate(all="ignore"):
        e0, out = xyz2(ang1, t1p, ang2, t2p, F(0))
        e1, out = xyz2(ang1, t1p, ang2, t2p, F(1000))
        z0 = F(F(0) + F(F(F(1000) - F(0)) / F(e1 - e0)) * F(F(0) - e0))
        min_e = e0
        for step in (F(1), F(-1)):
            counter = 0
            z1 = z0
            while (z1 < t1p[2]) if step > 0 else (z1 > 0):
                e, w = xyz2(ang1, t1p, ang2, t2p, z1)
                if e < min_e:
                    counter, min_e, out = 0, e, w
                else:
                    counter += 1
                    if counter - 1 > 100:
                        break
                z1 = F(z1 + step)
    return out


def main():
    rng = np.random.default_rng(20240611)
    # (i)/(iv) dense NCC crops: u8 and u16, windows 7, 15, 21
    ncc_cases = {}
    for name, (hh, ww, win, dmin, dmax, shift, dt) in {
        "ncc_u8_w7": (28, 60, 7, -9, 9, 4, np.uint8),
        "ncc_u8_w15": (36, 72, 15, -12, 12, -7, np.uint8),
        "ncc_u8_w21": (40, 80, 21, -6, 14, 9, np.uint8),
        "ncc_u16_w15": (34, 70, 15, -10, 10, 5, np.uint16),
    }.items():
        L = synth.texture(hh, ww, int(rng.integers(1 << 30)), dt)
        Rr = synth.shift_right_image(L, shift, fill_seed=int(rng.integers(1 << 30)))
        L[5:12, 20:40] = L[5, 20]                                       # a flat patch: zero-variance windows
        g = ncc_dense_numpy(L, Rr, win, dmin, dmax)
        ncc_cases[name] = dict(left=L, right=Rr, window=win, dmin=dmin, dmax=dmax, **g)
    np.savez_compressed(os.path.join(HERE, "ncc_cases.npz"),
                        **{"%s__%s" % (n, k): np.asarray(v) for n, c in ncc_cases.items() for k, v in c.items()})

    # (i)-(iii) tile correlation: 32, 64, 128 tiles with known shifts (negative / wrapped), impulse, constant
    xc = {}
    for name, (size, dx, dy) in {"xc32": (32, 3, -2), "xc64": (64, -9, 5), "xc128": (128, 20, -31)}.items():
        big = synth.texture(size + 80, size + 80, int(rng.integers(1 << 30)))
        img2 = np.ascontiguousarray(big[40:40 + size, 40:40 + size])
        img1 = np.ascontiguousarray(big[40 - dy:40 - dy + size, 40 - dx:40 - dx + size])
        r = find_offset_numpy(img1, img2, (0, 0, size, size), (0, 0, size, size), size, size)
        xc[name] = dict(img1=img1, img2=img2, size=size, **r)
    imp1 = np.zeros((32, 32), np.uint8); imp1[10, 12] = 200
    imp2 = np.zeros((32, 32), np.uint8); imp2[7, 20] = 100
    r = find_offset_numpy(imp1, imp2, (0, 0, 32, 32), (0, 0, 32, 32), 32, 32)
    xc["impulse"] = dict(img1=imp1, img2=imp2, size=32, **r)
    # padded tile: bbox smaller than the transform (image3.cpp:322-328)
    big = synth.texture(100, 100, 77)
    r = find_offset_numpy(big, big, (5, 6, 50, 40), (9, 3, 50, 40), 64, 64)
    xc["padded"] = dict(img1=big, img2=big, size=64, b1=np.array([5, 6, 50, 40]), b2=np.array([9, 3, 50, 40]), **r)
    np.savez_compressed(os.path.join(HERE, "xcorr_cases.npz"),
                        **{"%s__%s" % (n, k): np.asarray(v) for n, c in xc.items() for k, v in c.items()})

    # forward FFT (image-fft.cpp:65-87)
    img = synth.texture(32, 64, 5)
    np.savez_compressed(os.path.join(HERE, "fft_case.npz"), img=img,
                        spectrum=(np.fft.fft2(img.astype(np.float64)) / np.sqrt(img.size)).astype(np.complex64))

    # Gaussian blur on a narrow image (kernel wider than the image: repeated reflection)
    g = rng.random((20, 33)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "blur_case.npz"), src=g, dst=gaussian_blur_numpy(g).astype(np.float32))

    # (v) triangulation: 32 points, two synthetic cameras (nadir pair, then a slightly rotated pair)
    def cam(A, R, C):
        A, R = np.array(A, np.float32), np.array(R, np.float32)
        return A, R, (-(R.astype(np.float64) @ np.array(C, np.float64))).astype(np.float32)
    A = [[10000, 0, 4096], [0, 10000, 4096], [0, 0, 1]]
    ca, sa = np.cos(0.01), np.sin(0.01)
    Rz = [[ca, -sa, 0], [sa, ca, 0], [0, 0, 1]]
    pairs = [(cam(A, np.eye(3), [0, 0, 3000]), cam(A, np.eye(3), [900, 0, 3000])),
             (cam(A, Rz, [5, -3, 2950]), cam(A, np.eye(3), [880, 12, 3010]))]
    pts = np.stack([rng.integers(500, 7500, 32), rng.integers(500, 7500, 32)], 1)
    d = rng.integers(2400, 3300, 32)
    pts = np.concatenate([pts, np.stack([pts[:, 0] - d, pts[:, 1] + rng.integers(-1, 2, 32)], 1)], 1).astype(np.int32)
    tri = {"pts": pts}
    for i, (c1, c2) in enumerate(pairs):
        tri["A1_%d" % i], tri["R1_%d" % i], tri["T1_%d" % i] = c1
        tri["A2_%d" % i], tri["R2_%d" % i], tri["T2_%d" % i] = c2
        tri["xyz_%d" % i] = np.array([calculate_xyz_numpy(c1, c2, *p) for p in pts], np.float32)
    np.savez_compressed(os.path.join(HERE, "tri_cases.npz"), **tri)
    main_general()
    print("golden vectors written to", HERE)


def masked_peak_numpy(bands, mask_pct):
    """findCPPMax (image2.cpp:216-308) with numpy float32 scalars."""
    img = bands[0].astype(np.float32).copy()
    for b in bands[1:]:
        img = img * b.astype(np.float32)
    rows, cols = img.shape
    pp_row, pp_col = rows // 2, cols // 2
    pct = F(mask_pct)
    if pct > 99:
        pct = F(99)
    if pct <= 1:
        pct = F(0)
    r0 = int(F(pp_row) - F(pp_row) * pct / F(100)); r1 = int(F(pp_row) + F(pp_row) * pct / F(100))
    c0 = int(F(pp_col) - F(pp_col) * pct / F(100)); c1 = int(F(pp_col) + F(pp_col) * pct / F(100))
    sub = img[:pp_row * 2, :pp_col * 2].copy()
    sub[max(r0, 0):max(r1, 0), max(c0, 0):max(c1, 0)] = -np.inf
    k = int(np.argmax(sub))
    row, col = divmod(k, pp_col * 2)
    if not sub[row, col] > -1:
        return np.float32(-1), -pp_col, -pp_row
    return np.float32(sub[row, col]), col - pp_col, row - pp_row


def main_general():
    """Sizes that are not powers of two, the band product and the masked peak search."""
    rng = np.random.default_rng(20240612)
    xc = {}
    for name, (w, h, dx, dy) in {"g48x40": (48, 40, 5, -3), "g45x31": (45, 31, -4, 2), "g64x20": (64, 20, 7, 1)}.items():
        big = synth.texture(h + 40, w + 40, int(rng.integers(1 << 30)))
        img2 = np.ascontiguousarray(big[20:20 + h, 20:20 + w])
        img1 = np.ascontiguousarray(big[20 - dy:20 - dy + h, 20 - dx:20 - dx + w])
        r = find_offset_numpy(img1, img2, (0, 0, w, h), (0, 0, w, h), w, h)
        xc[name] = dict(img1=img1, img2=img2, w=w, h=h, **r)
    np.savez_compressed(os.path.join(HERE, "xcorr_general_cases.npz"),
                        **{"%s__%s" % (n, k): np.asarray(v) for n, c in xc.items() for k, v in c.items()})

    # band product (image3.cpp:382-418 with the band loop open): product of the normalised surfaces
    w = h = 32
    b1, b2, prod = [], [], np.ones((h, w), np.float32)
    for b in range(3):
        big = synth.texture(h + 20, w + 20, int(rng.integers(1 << 30)))
        i2 = np.ascontiguousarray(big[10:10 + h, 10:10 + w]); i1 = np.ascontiguousarray(big[10 + 2:10 + 2 + h, 10 - 3:10 - 3 + w])
        b1.append(i1); b2.append(i2)
        r = find_offset_numpy(i1, i2, (0, 0, w, h), (0, 0, w, h), w, h)
        prod = (prod.astype(np.float64) * r["surface"].astype(np.float64)).astype(np.float32)
    imout = prod.copy()
    dog = imout - gaussian_blur_numpy(imout)
    k = int(np.argmax(dog)); row, col = divmod(k, w)
    peak = prod[row, col]
    col = col - w if col > w // 2 else col
    row = row - h if row > h // 2 else row
    mp = {}
    for i, (rows, cols, nb, pct) in enumerate([(20, 30, 1, 0.0), (20, 30, 2, 40.0), (21, 33, 3, 60.0), (16, 16, 1, 120.0)]):
        bands = rng.random((nb, rows, cols)).astype(np.float32)
        bands[:, rows // 2, cols // 2] = 3.0
        pk, ox, oy = masked_peak_numpy(bands, pct)
        mp["mp%d__bands" % i] = bands; mp["mp%d__pct" % i] = np.float32(pct)
        mp["mp%d__result" % i] = np.array([pk, ox, oy], np.float64)
    np.savez_compressed(os.path.join(HERE, "bands_cases.npz"), bands1=np.stack(b1), bands2=np.stack(b2),
                        surface=prod, dog=dog.astype(np.float32), dx=col, dy=row, peak=np.float32(peak), **mp)


if __name__ == "__main__":
    main()
