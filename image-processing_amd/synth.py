"""Deterministic synthetic stereo pairs for tests and benchmarks (no reference imagery exists:
the reference ships none).  Left = band-limited random texture: splitmix64 noise, 3x3 box blur
twice, contrast-stretched to [16, 240]; right = left resampled by a known disparity field, with
image-1 position = image-2 position + d (the sign of image3.cpp:696-699)."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n outputs of splitmix64 started at `seed` (vectorised: state_i = seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _box3(a):
    p = np.pad(a, 1, mode="edge")
    s = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            s += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return s / 9.0


def texture(h, w, seed, dtype=np.uint8):
    """Band-limited texture in [16,240] (u8) or that range x257 (u16)."""
    noise = (splitmix64(seed, h * w) >> np.uint64(56)).astype(np.float32).reshape(h, w)
    t = _box3(_box3(noise))
    lo, hi = float(t.min()), float(t.max())
    t = 16.0 + (t - lo) * (224.0 / max(hi - lo, 1e-9))
    if np.dtype(dtype) == np.uint16:
        return np.clip(np.rint(t * 257.0), 0, 65535).astype(np.uint16)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def shift_right_image(left, disp, fill_seed=None):
    """right(x - d(x,y), y) = left(x, y) for an integer disparity field (scalar or HxW array):
    built by gathering right(x', y) = left(x' + d, y) with d sampled at the right-image pixel.
    Columns that fall outside the left image are filled with unrelated texture."""
    h, w = left.shape
    d = np.broadcast_to(np.asarray(disp, np.int64), (h, w))
    xs = np.arange(w, dtype=np.int64)[None, :] + d
    ok = (xs >= 0) & (xs < w)
    src = np.clip(xs, 0, w - 1)
    right = np.take_along_axis(left, src, axis=1)
    if fill_seed is not None:
        filler = texture(h, w, fill_seed, left.dtype)
        right = np.where(ok, right, filler)
    return right.astype(left.dtype)


def shift_right_image_subpixel(left, d):
    """Constant fractional shift by linear resampling: right(x') = left(x' + d)."""
    h, w = left.shape
    xs = np.arange(w, dtype=np.float64) + float(d)
    x0 = np.floor(xs).astype(np.int64)
    f = xs - x0
    a = left[:, np.clip(x0, 0, w - 1)].astype(np.float64)
    b = left[:, np.clip(x0 + 1, 0, w - 1)].astype(np.float64)
    out = a * (1 - f)[None, :] + b * f[None, :]
    info = np.iinfo(left.dtype)
    return np.ascontiguousarray(np.clip(np.rint(out), info.min, info.max).astype(left.dtype))


def planar_disparity(h, w, dlo, dhi, block, seed):
    """Piecewise-constant integer disparity in [dlo, dhi], constant on block x block tiles."""
    by, bx = (h + block - 1) // block, (w + block - 1) // block
    r = splitmix64(seed, by * bx).reshape(by, bx)
    d = (r % np.uint64(dhi - dlo + 1)).astype(np.int64) + dlo
    return np.kron(d, np.ones((block, block), np.int64))[:h, :w]


def config_pair(index, h=None, w=None):
    """Synthetic pair for BASELINE config `index` (1-based); h, w override the frame size.
    Returns (left, right, params)."""
    seed = 0xC0FFEE + index
    if index == 1:
        h, w = h or 2048, w or 2048
        left = texture(h, w, seed)
        right = shift_right_image(left, 17, fill_seed=seed + 1000)
        return left, right, {"window": 15, "dmin": -64, "dmax": 64}
    if index in (2, 4):
        h, w = h or (8192 if index == 2 else 16384), w or (8192 if index == 2 else 16384)
        left = texture(h, w, seed)
        d = planar_disparity(h, w, -200, 200, 64, seed + 1)
        right = shift_right_image(left, d, fill_seed=seed + 1000)
        noise = (splitmix64(seed + 2, h * w) % np.uint64(5)).astype(np.int16).reshape(h, w) - 2
        right = np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)
        return left, right, {"window": 21, "dmin": -256, "dmax": 256}
    if index == 5:
        h, w = h or 16384, w or 16384
        left = texture(h, w, seed, np.uint16)
        right = shift_right_image_subpixel(left, 37.25)
        return left, right, {"window": 31, "dmin": -512, "dmax": 512}
    raise ValueError("no dense-NCC pair for config %d" % index)
