"""FFT tile correlation parity: libmistereo (HIP, through the C ABI) against the CPU oracle's
restatement of findOffset (image3.cpp:346-517), image-fft.cpp and image-cross-cor.cpp."""
import numpy as np
import pytest

import oracle
import synth

pytestmark = pytest.mark.gpu


def _pair(h, w, dx, dy, seed, dtype=np.uint8):
    """img1 content = img2 content moved by (+dx, +dy): peak expected at (dx, dy)."""
    big = synth.texture(h + 2 * abs(dy) + 8, w + 2 * abs(dx) + 8, seed, dtype)
    oy, ox = abs(dy) + 4, abs(dx) + 4
    img2 = big[oy:oy + h, ox:ox + w]
    img1 = big[oy - dy:oy - dy + h, ox - dx:ox - dx + w]
    return np.ascontiguousarray(img1), np.ascontiguousarray(img2)


@pytest.mark.parametrize("size,dx,dy", [(64, 5, -3), (128, -20, 11), (256, 40, 33), (32, 3, 2), (16, 1, -1)])
def test_find_offset_single_tile(ctx, size, dx, dy):
    img1, img2 = _pair(size, size, dx, dy, 100 + size)
    ctx.upload_pair(img1, img2)
    peak, ox, oy = ctx.find_offset((0, 0, size, size), (0, 0, size, size), size, size)
    ref = oracle.find_offset(img1, img2, (0, 0, size, size), (0, 0, size, size), size, size, precision=0)
    ref64 = oracle.find_offset(img1, img2, (0, 0, size, size), (0, 0, size, size), size, size, precision=1)
    assert (ox, oy) == (ref["dx"], ref["dy"]) == (ref64["dx"], ref64["dy"])
    if size >= 64:
        assert (ox, oy) == (dx, dy)
    assert abs(peak - ref64["peak"]) <= 1e-4 * max(abs(ref64["peak"]), 1e-3) + 1e-5


def test_surfaces_match_oracle(ctx):
    size = 128
    img1, img2 = _pair(size, size, 9, -14, 7)
    ctx.upload_pair(img1, img2)
    surf, dog = ctx.tile_surfaces((0, 0, size, size), (0, 0, size, size), size, size)
    ref = oracle.find_offset(img1, img2, (0, 0, size, size), (0, 0, size, size), size, size, precision=1,
                             want_surface=True, want_dog=True)
    assert np.abs(surf - ref["surface"]).max() <= 2e-5
    assert np.abs(dog - ref["dog"]).max() <= 5e-5


def test_batched_tiles_with_padding_and_offsets(ctx):
    """512-px frame, 128-px tiles on a clamped grid (image3.cpp:764-767), tiles of image 2 displaced,
    one bbox smaller than the tile (zero padding, image3.cpp:322-328)."""
    rng = np.random.default_rng(5)
    img1, img2 = _pair(512, 512, 12, -7, 21)
    ctx.upload_pair(img1, img2)
    r1, r2 = [], []
    for j in range(0, 512, 128):
        for i in range(0, 512, 128):
            r1.append((i, j, 128, 128))
            ox, oy = (int(v) for v in rng.integers(-6, 7, 2))
            r2.append((min(max(i + ox, 0), 384), min(max(j + oy, 0), 384), 128, 128))
    r1.append((10, 20, 100, 90)); r2.append((14, 20, 100, 90))       # padded tile
    # bbox runs off the raster: zero-filled per pixel, the library's stated choice (parity unpinned:
    # the reference's RasterIO refuses such a window and calcDisparityMap never produces one)
    r1.append((450, 470, 128, 128)); r2.append((450, 470, 128, 128))
    got = ctx.find_offsets(r1, r2, 128, 128)
    for k, (a, b) in enumerate(zip(r1, r2)):
        ref = oracle.find_offset(img1, img2, a, b, 128, 128, precision=1)
        assert (got["dx"][k], got["dy"][k]) == (ref["dx"], ref["dy"]), "tile %d" % k
        assert abs(got["peak"][k] - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5


@pytest.mark.parametrize("tw,th", [(32, 32), (64, 32), (32, 64), (64, 64)])
def test_batches_of_small_tiles(ctx, tw, th):
    """Tiles of 32 / 64 samples a side, square and not (the smallest the matrix-core Gaussian passes take, one block of
    32 lines; with MISTEREO_XC_SMALL=1 an experimental fused kernel, four wavefronts of different tiles per workgroup): a
    batch of 14, displaced second boxes, one constant tile in the middle (max == min), one zero-padded tile and the DoG
    surface of a single tile, against the oracle."""
    rng = np.random.default_rng(tw + 3 * th)
    img1, img2 = _pair(300, 400, 3, -2, 77 + tw)
    img1[100:100 + th, 200:200 + tw] = 9
    img2[100:100 + th, 200:200 + tw] = 9
    ctx.upload_pair(img1, img2)
    r1, r2 = [], []
    for k in range(13):
        i, j = int(rng.integers(0, 400 - tw)), int(rng.integers(0, 300 - th))
        ox, oy = (int(v) for v in rng.integers(-3, 4, 2))
        r1.append((i, j, tw, th))
        r2.append((min(max(i + ox, 0), 400 - tw), min(max(j + oy, 0), 300 - th), tw, th))
    r1[6] = r2[6] = (200, 100, tw, th)                       # the constant tile
    r1.append((5, 7, tw - 5, th - 3)); r2.append((6, 7, tw - 5, th - 3))       # zero-padded tile
    got = ctx.find_offsets(r1, r2, tw, th)
    for k, (a, b) in enumerate(zip(r1, r2)):
        ref = oracle.find_offset(img1, img2, a, b, tw, th, precision=1)
        assert (got["dx"][k], got["dy"][k]) == (ref["dx"], ref["dy"]), "tile %d" % k
        assert abs(got["peak"][k] - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5
    assert got["peak"][6] == -1.0
    surf, dog = ctx.tile_surfaces(r1[0], r2[0], tw, th)
    ref = oracle.find_offset(img1, img2, r1[0], r2[0], tw, th, precision=1, want_surface=True, want_dog=True)
    assert np.abs(surf - ref["surface"]).max() <= 3e-5
    assert np.abs(dog - ref["dog"]).max() <= 1e-4


@pytest.mark.parametrize("tw,th", [(32, 32), (64, 32), (64, 64), (100, 60)])
def test_repeated_batches_of_small_tiles_are_bit_equal(ctx, tw, th):
    """Batches of 64 tiles and more run as two halves on two streams, call after call through the same workspace: the same
    200 tiles twice in a row must give the same bits, whatever the tiles of the call before were (a kernel that reads
    samples another call left in its caches, or a plane rewritten in place, shows here as a peak that differs in one call
    out of ten)."""
    h = w = 1024
    left = synth.texture(h, w, 41)
    right = synth.shift_right_image(left, 7, fill_seed=42)
    ctx.upload_pair(left, right)
    rng = np.random.default_rng(tw + th)
    n = 200 if tw <= 64 else 80          # 100 x 60: mixed-radix transforms and the segmented Gaussian passes
    for it in range(40 if tw <= 64 else 12):
        xs, ys = rng.integers(0, w - tw, n), rng.integers(0, h - th, n)
        r1 = np.stack([xs, ys, np.full(n, tw), np.full(n, th)], axis=1)
        r2 = r1.copy()
        r2[:, 0] = np.clip(r2[:, 0] - 7, 0, w - tw)
        a, b = ctx.find_offsets(r1, r2, tw, th), ctx.find_offsets(r1, r2, tw, th)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "call %d" % it


def test_rectangular_tile_and_u16_f32(ctx):
    for dtype in (np.uint16, np.float32):
        base1, base2 = _pair(64, 256, -17, 4, 3, np.uint16)
        img1, img2 = base1.astype(dtype), base2.astype(dtype)
        ctx.upload_pair(img1, img2)
        peak, ox, oy = ctx.find_offset((0, 0, 256, 64), (0, 0, 256, 64), 256, 64)
        ref = oracle.find_offset(img1, img2, (0, 0, 256, 64), (0, 0, 256, 64), 256, 64, precision=1)
        assert (ox, oy) == (ref["dx"], ref["dy"]) == (-17, 4)
        assert abs(peak - ref["peak"]) <= 1e-4 * abs(ref["peak"]) + 1e-5


def test_degenerate_constant_tile(ctx):
    """max == min: the reference divides 0 by 0, nothing beats the initial -1 (image3.cpp:415, 466-501)."""
    img = np.zeros((64, 64), np.uint8)
    ctx.upload_pair(img, img)
    peak, ox, oy = ctx.find_offset((0, 0, 64, 64), (0, 0, 64, 64), 64, 64)
    ref = oracle.find_offset(img, img, (0, 0, 64, 64), (0, 0, 64, 64), 64, 64)
    assert (peak, ox, oy) == (-1.0, 0, 0) == (ref["peak"], ref["dx"], ref["dy"])


def test_fft2d_forward_matches_numpy(ctx):
    rng = np.random.default_rng(11)
    imgs = rng.integers(0, 256, (3, 64, 128), dtype=np.uint8)
    got = ctx.fft2d_forward(imgs)
    for b in range(3):
        want = np.fft.fft2(imgs[b].astype(np.float64)) / np.sqrt(64 * 128)
        err = np.linalg.norm(got[b] - want) / np.linalg.norm(want)
        assert err <= 1e-5
        o = oracle.fft2d_forward(imgs[b])
        assert np.linalg.norm(got[b] - o) / np.linalg.norm(o) <= 1e-5


def test_fft2d_forward_512(ctx):
    img = synth.texture(512, 512, 2).astype(np.float32)
    got = ctx.fft2d_forward(img)
    want = np.fft.fft2(img.astype(np.float64)) / 512.0
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-5   # accuracy contract at 512^2


def _check_surface_u8(got, img1, img2):
    """Bytes against the oracle in the reference's own float32 arithmetic (precision 0) and in
    float64 (precision 1).  floor(v * 255) turns a last-bit difference of v into a whole byte step
    wherever v * 255 lies next to an integer, and float32 transforms of different butterfly order
    differ in the last bits, so a small fraction of bytes is off by one; it is reported, and bounded."""
    got = got.astype(np.int32)
    frac = {}
    for name, prec in (("float32 oracle", 0), ("float64 oracle", 1)):
        want = oracle.xcorr_surface_u8(img1, img2, precision=prec).astype(np.int32)
        d = np.abs(got - want)
        assert d.max() <= 1, name
        frac[name] = float((d > 0).mean())
    # the two oracles differ from each other by about as much as the library differs from either
    o32 = oracle.xcorr_surface_u8(img1, img2, precision=0).astype(np.int32)
    o64 = oracle.xcorr_surface_u8(img1, img2, precision=1).astype(np.int32)
    frac["float32 vs float64 oracle"] = float((o32 != o64).mean())
    print("xcorr_surface_u8 %s: bytes off by one: %s" % (got.shape, ", ".join("%s %.4f %%" % (k, 100 * v) for k, v in frac.items())))
    assert frac["float32 oracle"] <= 0.005 and frac["float64 oracle"] <= 0.005     # measured: 0.04 .. 0.10 %, the two oracles differ by 0.04 .. 0.08 %
    return frac


def test_xcorr_surface_u8(ctx):
    img1, img2 = _pair(128, 256, 30, -9, 17)
    ctx.upload_pair(img1, img2)
    got = ctx.xcorr_surface_u8().astype(np.int32)
    _check_surface_u8(got, img1, img2)
    py, px = np.unravel_index(np.argmax(got), got.shape)
    assert (px - 128, py - 64) == (30, -9)              # fft-shifted: offset relative to the centre


def test_large_tile_2048(ctx):
    img1, img2 = _pair(2048, 2048, -100, 57, 9)
    ctx.upload_pair(img1, img2)
    peak, ox, oy = ctx.find_offset((0, 0, 2048, 2048), (0, 0, 2048, 2048), 2048, 2048)
    assert (ox, oy) == (-100, 57) and peak == pytest.approx(1.0, abs=1e-5)


@pytest.mark.parametrize("w,h,dx,dy", [
    (100, 60, 7, -4),      # neither side a power of two
    (96, 64, -11, 5),      # one side a power of two
    (127, 33, 3, 2),       # odd sides: the last row / column is excluded from the peak search
    (200, 200, -25, 31),
    (8, 12, 1, -1),        # narrower than the blur radius: multiple reflections
    (300, 1, 5, 0),        # one row: the blur kernel collapses in that dimension
    (160, 96, 9, -7),      # sides that are multiples of 32: the Gaussian passes run on the matrix cores
    (32, 32, 2, 1),        # ... the smallest such tile (three reflections of a 32-sample line)
    (1024, 64, -30, 3),    # ... and the widest line their LDS image holds
    (600, 344, 21, -13),   # even sides, not multiples of 32, longer than a segment of the segmented matrix-core passes:
                           # three segments of columns (256, 256, 88), two of rows, partial blocks at both ends
    (1100, 36, -8, 2),     # a line longer than the LDS-resident kernels hold next to one shorter than the filter
])
def test_find_offset_general_sizes(ctx, w, h, dx, dy):
    """Tile sides that are not powers of two (the first level of image3.cpp's recursion works on the
    whole rasters): mixed-radix row transforms (sides whose prime factors are all <= 13) and chirp-z ones
    (127, 101, ...) against the oracle's mixed radix-2 / Bluestein FFT."""
    img1, img2 = _pair(h, w, dx, dy, 300 + w)
    ctx.upload_pair(img1, img2)
    peak, ox, oy = ctx.find_offset((0, 0, w, h), (0, 0, w, h), w, h)
    ref = oracle.find_offset(img1, img2, (0, 0, w, h), (0, 0, w, h), w, h, precision=1, want_surface=True, want_dog=True)
    assert (ox, oy) == (ref["dx"], ref["dy"])
    if min(w, h) >= 60:
        assert (ox, oy) == (dx, dy)
    assert abs(peak - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5
    surf, dog = ctx.tile_surfaces((0, 0, w, h), (0, 0, w, h), w, h)
    assert np.abs(surf - ref["surface"]).max() <= 3e-5
    assert np.abs(dog - ref["dog"]).max() <= 1e-4


@pytest.mark.parametrize("w,h", [(3000, 2000), (2006, 1511)])
def test_general_size_large_frame(ctx, w, h):
    """A whole frame as one tile: 3000 x 2000 (2^3 3 5^3 and 2^4 5^3: mixed-radix transforms of the lengths themselves)
    and 2006 x 1511 (2 17 59 and a prime: chirp-z transforms of lengths 4096)."""
    img1, img2 = _pair(h, w, -140, 77, 12)
    ctx.upload_pair(img1, img2)
    peak, ox, oy = ctx.find_offset((0, 0, w, h), (0, 0, w, h), w, h)
    assert (ox, oy) == (-140, 77) and peak == pytest.approx(1.0, abs=1e-4)


def test_fft2d_forward_general_sizes(ctx):
    rng = np.random.default_rng(12)
    for shape in ((60, 100), (33, 127), (64, 96), (1, 50)):
        img = rng.integers(0, 65536, shape).astype(np.uint16)
        got = ctx.fft2d_forward(img)
        want = np.fft.fft2(img.astype(np.float64)) / np.sqrt(shape[0] * shape[1])
        assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 2e-5, shape


@pytest.mark.parametrize("shape", [
    (6, 8000),        # 8 8 5 5 5: five passes, 1024 threads, eight elements per thread
    (3, 6000),        # 8 2 3 5 5 5
    (1001, 210),      # 7 11 13 and 2 3 5 7: every odd radix
    (4, 8190),        # 2 3^2 5 7 13: the longest non-power-of-two sequences take 1024 threads with a ragged last round
    (13, 6561),       # 3^8: eight passes of the smallest odd radix
    (12, 4100),       # 4 5 5 41: a factor above 13 sends the whole length through the chirp
    (22, 2 * 2 * 2 * 2 * 11 * 13),
    (5, 3),
])
def test_fft2d_forward_mixed_radix_lengths(ctx, shape):
    """Lengths whose prime factors are all <= 13 are transformed directly (radices 8, 4, 2, 3, 5, 7, 11, 13 in LDS)."""
    rng = np.random.default_rng(shape[1])
    img = rng.integers(0, 256, shape).astype(np.uint8)
    got = ctx.fft2d_forward(img)
    want = np.fft.fft2(img.astype(np.float64)) / np.sqrt(shape[0] * shape[1])
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 2e-5, shape
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), shape


def test_xcorr_surface_u8_odd_size(ctx):
    """Odd sizes: fft2shift swaps halves of floor(n/2) and leaves the last row / column in place
    (image-cross-cor.cpp:52-80)."""
    img1, img2 = _pair(75, 101, 8, -6, 19)
    ctx.upload_pair(img1, img2)
    got = ctx.xcorr_surface_u8().astype(np.int32)
    _check_surface_u8(got, img1, img2)


def test_more_small_tiles_than_one_grid_holds(ctx):
    """45 000 tiles of 8 x 8 (the general-size path puts two images per tile into grid.z, whose limit
    is 65535): batches are split, every tile still gets its own answer."""
    img1, img2 = _pair(600, 2400, 1, 0, 77)
    ctx.upload_pair(img1, img2)
    xs, ys = np.meshgrid(np.arange(0, 2400, 8), np.arange(0, 600, 8))
    rects = np.stack([xs.ravel(), ys.ravel(), np.full(xs.size, 8), np.full(xs.size, 8)], axis=1)
    assert len(rects) == 22500
    rects = np.concatenate([rects, rects])
    got = ctx.find_offsets(rects, rects, 8, 8)
    assert len(got) == 45000
    assert np.array_equal(got[:22500], got[22500:])            # the repeated tiles repeat
    for k in (0, 1234, 22499, 40000, 44999):
        ref = oracle.find_offset(img1, img2, tuple(rects[k]), tuple(rects[k]), 8, 8, precision=1)
        assert (got["dx"][k], got["dy"][k]) == (ref["dx"], ref["dy"])
        assert abs(got["peak"][k] - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5


@pytest.mark.parametrize("w,h", [(64, 64), (100, 60)])
def test_multi_band_product(ctx, w, h):
    """Three bands: the normalised surfaces are multiplied (image3.cpp:382-418 with n_bands not
    forced to 1), on the power-of-two path and on the general one."""
    pairs = [_pair(h, w, 6, -3, 40 + b) for b in range(3)]
    b1, b2 = [p[0] for p in pairs], [p[1] for p in pairs]
    ctx.upload_pair(b1[0], b2[0])
    ctx.upload_extra_bands(b1[1:], b2[1:])
    peak, ox, oy = ctx.find_offset((0, 0, w, h), (0, 0, w, h), w, h)
    surf, dog = ctx.tile_surfaces((0, 0, w, h), (0, 0, w, h), w, h)
    ref = oracle.find_offset_bands(b1, b2, (0, 0, w, h), (0, 0, w, h), w, h, precision=1, want_surface=True, want_dog=True)
    assert (ox, oy) == (ref["dx"], ref["dy"]) == (6, -3)
    assert abs(peak - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5
    assert np.abs(surf - ref["surface"]).max() <= 5e-5
    assert np.abs(dog - ref["dog"]).max() <= 1e-4
    single = oracle.find_offset(b1[0], b2[0], (0, 0, w, h), (0, 0, w, h), w, h, precision=1, want_surface=True)
    assert np.abs(surf - single["surface"]).max() > 1e-3        # the product differs from band 1 alone
    # binding a new pair drops the extra bands
    ctx.upload_pair(b1[0], b2[0])
    surf1, _ = ctx.tile_surfaces((0, 0, w, h), (0, 0, w, h), w, h)
    assert np.abs(surf1 - single["surface"]).max() <= 3e-5


@pytest.mark.parametrize("rows,cols,nb,pct", [(64, 96, 1, 0), (64, 96, 2, 30), (75, 101, 3, 50), (40, 40, 1, 99.5),
                                              (40, 40, 1, 1.0), (33, 64, 2, 150)])
def test_masked_peak(ctx, rows, cols, nb, pct):
    """findCPPMax (image2.cpp:216-308): band product and first maximum outside the central mask."""
    rng = np.random.default_rng(rows + cols + nb)
    bands = rng.random((nb, rows, cols)).astype(np.float32)
    bands[:, rows // 2, cols // 2] = 5.0                          # a central peak the mask must hide
    bands[:, 3, 7] = 2.0
    bands[:, rows // 2 - 1, 5] = 2.0                              # equal maxima: the first in row-major order wins
    got = ctx.masked_peak(bands, pct)
    want = oracle.masked_peak(bands, pct)
    assert got == want
    if pct in (30, 50):
        assert (got[1], got[2]) == (7 - cols // 2, 3 - rows // 2)
    elif pct <= 1:
        assert (got[1], got[2]) == (0, 0)


def test_masked_peak_nothing_above_minus_one(ctx):
    bands = np.full((1, 16, 16), -2.0, np.float32)
    assert ctx.masked_peak(bands, 10) == oracle.masked_peak(bands, 10) == (-1.0, -8, -8)


def test_largest_power_of_two_tile_8192(ctx):
    """The largest tile of the LDS-resident path: one 8192 x 8192 tile (column transform of two
    8192-point sequences plus twiddles = the whole 160 KB of LDS)."""
    img1, img2 = _pair(8192, 8192, 321, -123, 10)
    ctx.upload_pair(img1, img2)
    peak, ox, oy = ctx.find_offset((0, 0, 8192, 8192), (0, 0, 8192, 8192), 8192, 8192)
    assert (ox, oy) == (321, -123) and peak == pytest.approx(1.0, abs=1e-4)


@pytest.mark.parametrize("seed", range(24))
def test_random_tile_sizes_boxes_and_types(ctx, seed):
    """Random transform sizes (powers of two and not, odd, rectangular), bounding boxes smaller than the
    transform or running off the raster, 8/16-bit/float rasters: offsets equal to the oracle's, peak and
    surfaces within tolerance."""
    rng = np.random.default_rng(7000 + seed)
    if seed % 2:
        w, h = (int(2 ** rng.integers(4, 9)), int(2 ** rng.integers(4, 9)))
    else:
        w, h = int(rng.integers(20, 300)), int(rng.integers(20, 300))
    dx, dy = int(rng.integers(-w // 5, w // 5 + 1)), int(rng.integers(-h // 5, h // 5 + 1))
    dtype = [np.uint8, np.uint16, np.float32][seed % 3]
    base1, base2 = _pair(h + 16, w + 16, dx, dy, 900 + seed, np.uint16 if dtype != np.uint8 else np.uint8)
    img1, img2 = base1.astype(dtype), base2.astype(dtype)
    b1 = (int(rng.integers(0, 12)), int(rng.integers(0, 12)), int(rng.integers(w // 2, w + 1)), int(rng.integers(h // 2, h + 1)))
    b2 = (int(rng.integers(0, 12)), int(rng.integers(0, 12)), b1[2], b1[3])
    if seed % 6 == 0:
        b1 = (w // 2, h // 2, w, h)                    # runs off the raster on the right / bottom (parity unpinned: see mi_stereo.h)
    ctx.upload_pair(img1, img2)
    peak, ox, oy = ctx.find_offset(b1, b2, w, h)
    ref = oracle.find_offset(img1, img2, b1, b2, w, h, precision=1, want_surface=True, want_dog=True)
    surf, dog = ctx.tile_surfaces(b1, b2, w, h)
    assert np.abs(surf - ref["surface"]).max() <= 3e-5
    assert np.abs(dog - ref["dog"]).max() <= 1e-4
    if (ox, oy) != (ref["dx"], ref["dy"]):
        # a different peak position is acceptable only where the two DoG values tie within float noise
        a = ref["dog"][oy % h, ox % w]
        assert abs(a - ref["dog_peak"]) <= 2e-5, (ox, oy, ref["dx"], ref["dy"])
    else:
        assert abs(peak - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5
