"""Dense-NCC parity: libmistereo (HIP, through the C ABI) against the CPU oracle."""
import os

import numpy as np
import pytest

import mistereo
import oracle
import synth
from ncc_util import compare_ncc, flat_mask

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(ctx, left, right, window, dmin, dmax, peak2=False):
    ctx.upload_pair(left, right)
    gpu = ctx.ncc_dense(window, dmin, dmax, want_idisp=True, want_peak2=peak2)
    ora = oracle.ncc_dense(left, right, window, dmin, dmax, want_peak2=peak2, want_volume=True)
    ora["flat"] = flat_mask(ora["volume"], ora["idisp"], dmin)
    return compare_ncc(gpu, ora, volume=ora["volume"], dmin=dmin, check_peak2=peak2)


@pytest.mark.parametrize("window,dmin,dmax,shape,shift", [
    (15, -64, 64, (96, 300), 17),
    (21, -40, 90, (80, 260), -23),
    (3, -5, 5, (40, 70), 2),
    (31, -70, 70, (70, 256), 40),
    (7, 0, 63, (50, 200), 11),       # exactly one chunk
    (9, -100, 27, (64, 333), -60),   # two chunks, ragged width
])
def test_constant_shift(ctx, window, dmin, dmax, shape, shift):
    left = synth.texture(shape[0], shape[1], 1234 + window)
    right = synth.shift_right_image(left, shift, fill_seed=99)
    st = _run(ctx, left, right, window, dmin, dmax)
    assert st["valid"] > 0


def test_peak2(ctx):
    left = synth.texture(72, 280, 77)
    right = synth.shift_right_image(left, 9, fill_seed=5)
    _run(ctx, left, right, 11, -70, 70, peak2=True)


@pytest.mark.parametrize("dtype,window,dmin,dmax,shape,shift", [
    (np.uint8, 21, -256, 256, (70, 900), 63),       # 8 chunks + the one-disparity tail; winner in the last lane of a chunk
    (np.uint8, 21, -256, 256, (70, 900), -192),     # winner in lane 0 of a chunk: the neighbour across the boundary is excluded
    (np.uint8, 9, -100, 27, (64, 333), -37),        # two chunks: winner in lane 63 of the first
    (np.uint8, 7, 0, 67, (50, 260), 62),            # one chunk and a four-disparity tail, winner next to the boundary
    (np.uint8, 7, 0, 67, (50, 260), 65),            # winner inside the tail
    (np.uint8, 31, -70, 70, (70, 256), 40),         # 31 x 31 window (fewer wavefronts per SIMD)
    (np.uint8, 5, -1, 1, (40, 90), 0),              # three disparities: nothing is further than one from the winner
    (np.uint16, 31, -512, 512, (70, 1500), 333),    # 1025 disparities, 17 chunks (config[4]'s search)
    (np.uint16, 15, -64, 64, (96, 300), -1),        # winner in lane 63 of chunk 0
    (np.uint16, 11, -64, 63, (60, 300), 0),         # winner in lane 0 of chunk 1
])
def test_peak2_chunk_boundaries(ctx, dtype, window, dmin, dmax, shape, shift):
    """Second peak (best score further than one disparity from the winner) on 8- and 16-bit pairs:
    winners at chunk boundaries, where the excluded neighbour lives in another wavefront's chunk or in
    the tail plane, and ranges of up to 1025 disparities."""
    left = _texture16(shape[0], shape[1], 55 + window) if dtype == np.uint16 else synth.texture(shape[0], shape[1], 55 + window)
    right = synth.shift_right_image(left, shift, fill_seed=56)
    st = _run(ctx, left, right, window, dmin, dmax, peak2=True)
    assert st["valid"] > 0


@pytest.mark.parametrize("window,dmin,dmax,shape,shift", [
    (21, -100, 100, (60, 400), 4),       # dmax - shift = 96: in every 64th column the winner is right column 0 of the second range
    (21, -100, 100, (60, 400), 5),       # ... and right column 95 of the first range in the next column over
    (9, -256, 256, (40, 700), 64),       # 513 disparities, six ranges: winners on the boundary between ranges 2 and 3
    (5, -40, 250, (40, 500), 59),        # boundary winners in edge ranges (masked tiles on both sides)
    (15, 0, 94, (50, 300), 0),           # a single range, winner in its last right columns
    (3, -2, 2, (30, 200), 1),            # five disparities
])
def test_peak2_range_boundaries_matrix_core(ctx, window, dmin, dmax, shape, shift):
    """Second peak on the matrix-core kernels (windows up to 21): a wavefront's range of 96 right columns starts at
    disparity dmax + ((x - r) & 63), so with a constant shift the winner walks through every position of a range as
    x grows -- including the first and last one, where the excluded neighbour belongs to the next range's plane."""
    left = synth.texture(shape[0], shape[1], 91 + window)
    right = synth.shift_right_image(left, shift, fill_seed=92)
    st = _run(ctx, left, right, window, dmin, dmax, peak2=True)
    assert st["valid"] > 0


@pytest.mark.parametrize("seed", range(12))
def test_peak2_random(ctx, seed):
    window, h, w, dmin, dmax, dtype, shift = _random_case(3000 + seed)
    left = _texture16(h, w, seed) if dtype == np.uint16 else synth.texture(h, w, seed)
    # piecewise disparities: winners land in every lane, repeated content gives real second peaks
    dmap = synth.planar_disparity(h, w, max(dmin, -60), min(dmax, 60), 16, seed + 7)
    right = synth.shift_right_image(left, dmap, fill_seed=seed + 500)
    _run(ctx, left, right, window, dmin, dmax, peak2=True)


def test_piecewise_disparity_with_noise(ctx):
    left, right, p = synth.config_pair(2, h=120, w=700)
    _run(ctx, left, right, p["window"], -130, 130)


def test_flat_and_saturated_regions(ctx):
    rng = np.random.default_rng(3)
    left = synth.texture(90, 320, 5)
    left[20:60, 100:180] = 200          # flat patch: zero variance windows
    left[:, 250:] = 255
    left[70:, :40] = 0
    right = synth.shift_right_image(left, 6, fill_seed=8)
    right[30:50, 10:90] = rng.integers(0, 2, (20, 80)) * 255   # extreme contrast
    _run(ctx, left, right, 15, -32, 32)


def test_too_small_frame_is_all_invalid(ctx):
    left = synth.texture(10, 40, 1)
    ctx.upload_pair(left, left)
    out = ctx.ncc_dense(15, -4, 4)
    assert np.all(out["idisp"] == -32768) and np.all(out["peak"] == 0)


@pytest.mark.parametrize("dmin,dmax,dtype", [(-40, 40, np.uint8), (-32, 32, np.uint8), (-70, 59, np.uint16)])
def test_row_strips_equal_the_full_frame(ctx, dmin, dmax, dtype):
    """Row strips computed separately (the multi-GPU partition) equal the full-frame result; the
    second range leaves one disparity to the tail kernel."""
    import torch
    import partition
    left = synth.texture(90, 300, 21, dtype)
    right = synth.shift_right_image(left, -12, fill_seed=22)
    if dtype == np.uint16:
        tl = torch.from_numpy(left.view(np.int16)).cuda().view(torch.uint16)
        tr = torch.from_numpy(right.view(np.int16)).cuda().view(torch.uint16)
    else:
        tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.bind_pair_torch(tl, tr)
    full_d = torch.empty((90, 300), dtype=torch.float32, device="cuda")
    full_p = torch.empty_like(full_d)
    ctx.ncc_dense_torch(11, dmin, dmax, full_d, full_p)
    part_d = torch.full_like(full_d, 12345.0)
    part_p = torch.full_like(full_d, 12345.0)
    torch.cuda.synchronize()           # the fills run on torch's stream, the library on its own: order them
    for row0, n in partition.row_strips(90, 3):
        ctx.ncc_dense_torch(11, dmin, dmax, part_d, part_p, row0=row0, nrows=n)
    ctx.synchronize()
    assert torch.equal(full_d, part_d) and torch.equal(full_p, part_p)


def _texture16(h, w, seed):
    """16-bit texture using the full range (not just multiples of 257)."""
    rng = np.random.default_rng(seed)
    t = synth.texture(h, w, seed, np.uint16).astype(np.int64)
    t = t + rng.integers(-120, 121, t.shape)
    return np.clip(t, 0, 65535).astype(np.uint16)


@pytest.mark.parametrize("window,dmin,dmax,shape,shift", [
    (15, -64, 64, (96, 300), 17),
    (21, -40, 90, (80, 260), -23),
    (3, -5, 5, (40, 70), 2),
    (31, -70, 70, (70, 256), 40),
    (7, 0, 63, (50, 200), 11),
    (9, -100, 27, (64, 333), -60),
])
def test_u16_constant_shift(ctx, window, dmin, dmax, shape, shift):
    left = _texture16(shape[0], shape[1], 4321 + window)
    right = synth.shift_right_image(left, shift, fill_seed=98)
    st = _run(ctx, left, right, window, dmin, dmax)
    assert st["valid"] > 0


def test_u16_extremes(ctx):
    """Saturated and zero 16-bit regions: the largest products and sums the integer path sees."""
    rng = np.random.default_rng(11)
    left = _texture16(80, 300, 6)
    left[10:50, 60:140] = 65535
    left[55:, 200:] = 0
    right = synth.shift_right_image(left, -7, fill_seed=3)
    right[20:60, 150:260] = (rng.integers(0, 2, (40, 110)) * 65535).astype(np.uint16)
    _run(ctx, left, right, 31, -40, 40)


def test_u16_wide_search_many_chunks(ctx):
    """17 chunks of 64 disparities (more than one workgroup of the 8-bit v1 kernel could hold)."""
    left = _texture16(70, 1500, 8)
    right = synth.shift_right_image(left, 333, fill_seed=4)
    _run(ctx, left, right, 31, -512, 512)


def _random_case(seed):
    rng = np.random.default_rng(seed)
    window = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17, 21, 25, 29, 31]))
    h = int(rng.integers(window, window + 70))
    w = int(rng.integers(window + 2, 420))
    span = int(rng.choice([1, 2, 5, 33, 64, 65, 68, 100, 129, 200, 300]))
    dmin = int(rng.integers(-span, 40))
    dmax = dmin + span - 1
    dtype = np.uint16 if rng.random() < 0.35 else np.uint8
    shift = int(rng.integers(dmin - 3, dmax + 4))
    return window, h, w, dmin, dmax, dtype, shift


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_windows_and_ranges(ctx, seed):
    """Random window sizes, frame shapes (down to a single valid row), search ranges (one disparity,
    ranges that end on / just past a multiple of 64, all-positive or all-negative ranges) and pixel
    types against the oracle."""
    window, h, w, dmin, dmax, dtype, shift = _random_case(1000 + seed)
    left = _texture16(h, w, seed) if dtype == np.uint16 else synth.texture(h, w, seed)
    right = synth.shift_right_image(left, shift, fill_seed=seed + 500)
    if seed % 5 == 0:
        left[h // 3: h // 2, w // 4: w // 2] = left[0, 0]          # a flat patch
    _run(ctx, left, right, window, dmin, dmax)


@pytest.mark.parametrize("window", [3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31])
def test_every_window_of_the_matrix_core_path(ctx, window):
    """Every 8-bit window runs the matrix-core argmax kernels: up to 21 x 21 with the integer part of the mean correction
    inside the products, from 23 x 23 on with that correction as a vector instruction per score.  Search ranges that end on, just before
    and just past the kernels' 32-disparity tile grid and 96-column ranges, from one disparity to several key
    planes, on a width that is no multiple of the 64-column blocks."""
    rng = np.random.default_rng(window)
    for k, (dmin, dmax) in enumerate([(-16, 16), (0, 0), (-3, 28), (-31, 33), (-100, 92), (5, 200), (-160, 159)]):
        h, w = 2 * window + 9 + 3 * k, 230 + 17 * k + (dmax - dmin)
        left = synth.texture(h, w, 77 * window + k)
        shift = int(rng.integers(dmin, dmax + 1))
        right = synth.shift_right_image(left, shift, fill_seed=k)
        st = _run(ctx, left, right, window, dmin, dmax)
        assert st["valid"] > 0


@pytest.mark.parametrize("window", [3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31])
def test_edge_ranges_with_compile_time_tiles(ctx, window):
    """Searches of 32 G + 1 disparities (+-64: the last range holds diagonals G and G + 1; +-128: two last ranges, the very
    last one a single masked tile) run their first and last ranges of right columns as launches whose tiles are fixed at
    compile time; every window, with and without the second peak, against the oracle."""
    for k, (dmin, dmax) in enumerate([(-64, 64), (-128, 128), (-100, 28), (3, 259)]):
        h, w = 2 * window + 7 + k, 300 + 40 * k + (dmax - dmin)
        left = synth.texture(h, w, 900 + 31 * window + k)
        right = synth.shift_right_image(left, (dmin + dmax) // 2 + 5 - 3 * k, fill_seed=k + 20)
        st = _run(ctx, left, right, window, dmin, dmax, peak2=(k % 2 == 1))
        assert st["valid"] > 0


@pytest.mark.parametrize("window", [3, 7, 9, 13, 15, 19, 21, 23, 27, 31])
def test_edge_ranges_of_general_search_sizes(ctx, window):
    """Searches whose size is not 32 G + 1 mask two diagonals in their last range; the three residues of G mod 3 (201, 233
    and 265 disparities) and a short search of one range take one compile-time tile pattern each."""
    for k, (dmin, dmax) in enumerate([(-100, 100), (-116, 116), (-132, 132), (-20, 50)]):
        h, w = 2 * window + 5 + k, 280 + 24 * k + (dmax - dmin)
        left = synth.texture(h, w, 1700 + 17 * window + k)
        right = synth.shift_right_image(left, (dmin + dmax) // 2 - 7 + 4 * k, fill_seed=k + 40)
        st = _run(ctx, left, right, window, dmin, dmax, peak2=(k % 2 == 0))
        assert st["valid"] > 0


def test_widest_search_range_on_the_matrix_core_path(ctx):
    """2048 disparities (the API's limit) are 22 key planes of 96 right columns; the frame is narrower than the range,
    so most candidates of most pixels fall off the raster."""
    left = synth.texture(44, 1500, 4242)
    right = synth.shift_right_image(left, -333, fill_seed=5)
    st = _run(ctx, left, right, 21, -1024, 1023)
    assert st["valid"] > 0
    right = synth.shift_right_image(left, 700, fill_seed=6)
    _run(ctx, left, right, 11, 0, 2047)


@pytest.mark.parametrize("window", [9, 21, 23, 31])
def test_extreme_contrast_at_the_accumulator_bound(ctx, window):
    """Black-and-white images drive the window sums and the mean corrections of the matrix-core kernels to
    their largest magnitudes (128^2 per pixel): every partial sum must still be exact in f32."""
    rng = np.random.default_rng(11 + window)
    h, w = 3 * window + 20, 420
    base = (rng.integers(0, 2, (h, w + 80)) * 255).astype(np.uint8)
    base[:, 100:180] = 255                      # saturated run: windows of maximal mean
    base[: h // 2, 200:260] = 0
    base[:, 300:330] = np.where(rng.random((h, 30)) < 0.03, 0, 255)      # nearly saturated, small variance
    left = np.ascontiguousarray(base[:, 40:40 + w])
    right = np.ascontiguousarray(base[:, 40 + 13:40 + 13 + w])
    _run(ctx, left, right, window, -64, 64)
    # all-black against all-white runs: every product is 128 * 128 or -128 * 127, the window sums at their bound
    left[:, 100:180] = 0
    _run(ctx, left, right, window, -64, 64)


def test_back_to_back_calls_on_a_caller_stream():
    """The 8-bit path forks small kernels onto an internal stream: three calls enqueued without any
    synchronisation in between (two pairs, output tensors overwritten by the caller's own stream
    work between the calls) must give what the same calls give one at a time."""
    import torch
    c = mistereo.Context(0)
    try:
        rng = np.random.default_rng(77)
        pairs = []
        for k in range(2):
            left = synth.texture(300, 520, 900 + k)
            right = synth.shift_right_image(left, synth.planar_disparity(300, 520, -12, 12, 8, 910 + k), fill_seed=920 + k)
            pairs.append((torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()))
        want = []
        for tl, tr in pairs:                                   # one at a time, synchronised
            c.bind_pair_torch(tl, tr)
            d = torch.empty((300, 520), dtype=torch.float32, device="cuda"); p = torch.empty_like(d)
            c.ncc_dense_torch(11, -70, 70, d, p)
            c.synchronize()
            want.append((d.clone(), p.clone()))
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            c.use_torch_stream()
            d = torch.empty((300, 520), dtype=torch.float32, device="cuda"); p = torch.empty_like(d)
            got = []
            for rep in range(3):
                tl, tr = pairs[rep % 2]
                c.bind_pair_torch(tl, tr)
                d.fill_(-5.0); p.fill_(-5.0)                   # caller's work on the same stream, same buffers
                c.ncc_dense_torch(11, -70, 70, d, p)
                got.append((d.clone(), p.clone()))             # ordered behind the call on the caller's stream
            st.synchronize()
        for rep in range(3):
            assert torch.equal(got[rep][0], want[rep % 2][0]) and torch.equal(got[rep][1], want[rep % 2][1])
    finally:
        c.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_pixels_rewritten_in_bound_buffers(dtype):
    """Tables derived from the pixels are cached per bound pair: a caller that writes a new frame
    into the same device buffers calls mis_invalidate_pair (or rebinds) and gets the new frame's
    disparities; without it the previous frame's tables answer (the documented contract)."""
    import torch
    c = mistereo.Context(0)
    try:
        h, w = 80, 300
        def frame(seed, shift):
            left = _texture16(h, w, seed) if dtype == np.uint16 else synth.texture(h, w, seed)
            return left, synth.shift_right_image(left, shift, fill_seed=seed + 1)
        def cuda(a):
            return torch.from_numpy(a.view(np.int16)).cuda().view(torch.uint16) if dtype == np.uint16 else torch.from_numpy(a).cuda()
        l1, r1 = frame(1, 9)
        l2, r2 = frame(2, -14)
        tl, tr = cuda(l1), cuda(r1)
        c.bind_pair_torch(tl, tr)
        d = torch.empty((h, w), dtype=torch.float32, device="cuda"); p = torch.empty_like(d)
        i16 = torch.empty((h, w), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        c.ncc_dense_torch(11, -40, 40, d, p, i16)
        c.synchronize()
        first = i16.cpu().numpy().copy()
        assert (first[20:60, 100:200] == 9).all()
        tl.copy_(cuda(l2)); tr.copy_(cuda(r2))            # new pixels, same buffers
        torch.cuda.synchronize()
        c.ncc_dense_torch(11, -40, 40, d, p, i16)
        c.synchronize()
        assert np.array_equal(i16.cpu().numpy(), first)    # not invalidated: the cached tables still answer
        c.invalidate_pair()
        c.ncc_dense_torch(11, -40, 40, d, p, i16)
        c.synchronize()
        gpu = {"disp": d.cpu().numpy(), "peak": p.cpu().numpy(), "idisp": i16.cpu().numpy()}
        ora = oracle.ncc_dense(l2, r2, 11, -40, 40, want_volume=True)
        ora["flat"] = flat_mask(ora["volume"], ora["idisp"], -40)
        compare_ncc(gpu, ora, volume=ora["volume"], dmin=-40)
        # rebinding does the same
        tl.copy_(cuda(l1)); tr.copy_(cuda(r1))
        torch.cuda.synchronize()
        c.bind_pair_torch(tl, tr)
        c.ncc_dense_torch(11, -40, 40, d, p, i16)
        c.synchronize()
        assert np.array_equal(i16.cpu().numpy(), first)
    finally:
        c.close()


def test_cached_tables_reused_from_another_stream():
    """The first call after a bind builds the tables on the stream it runs on; a later call on a
    different stream reuses them and is ordered behind that build by an event."""
    import torch
    c = mistereo.Context(0)
    try:
        h, w = 600, 1500
        left = synth.texture(h, w, 5)
        right = synth.shift_right_image(left, 21, fill_seed=6)
        tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        d1 = torch.empty((h, w), dtype=torch.float32, device="cuda"); p1 = torch.empty_like(d1)
        d2 = torch.empty_like(d1); p2 = torch.empty_like(d1)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        c.bind_pair_torch(tl, tr)
        c.set_stream(s1.cuda_stream)
        c.ncc_dense_torch(21, -64, 64, d1, p1, row0=0, nrows=300)       # builds the tables on s1
        c.set_stream(s2.cuda_stream)
        c.ncc_dense_torch(21, -64, 64, d2, p2, row0=300, nrows=300)     # reuses them on s2, no host synchronisation between
        torch.cuda.synchronize()
        c.set_stream(None)
        full_d = torch.empty_like(d1); full_p = torch.empty_like(d1)
        c.bind_pair_torch(tl, tr)
        c.ncc_dense_torch(21, -64, 64, full_d, full_p)
        c.synchronize()
        assert torch.equal(d1[:300], full_d[:300]) and torch.equal(d2[300:], full_d[300:])
        assert torch.equal(p1[:300], full_p[:300]) and torch.equal(p2[300:], full_p[300:])
    finally:
        c.close()


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_strips_in_any_order_build_only_their_rows(ctx, dtype):
    """The per-pair tables are built for the rows a call reads and kept for later calls: strips taken
    bottom-up, with a gap, overlapping and repeated give the full frame's bits."""
    import torch
    h, w = 700, 400
    left = _texture16(h, w, 31) if dtype == np.uint16 else synth.texture(h, w, 31)
    right = synth.shift_right_image(left, 7, fill_seed=32)
    if dtype == np.uint16:
        tl = torch.from_numpy(left.view(np.int16)).cuda().view(torch.uint16)
        tr = torch.from_numpy(right.view(np.int16)).cuda().view(torch.uint16)
    else:
        tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.bind_pair_torch(tl, tr)
    full_d = torch.empty((h, w), dtype=torch.float32, device="cuda"); full_p = torch.empty_like(full_d)
    ctx.ncc_dense_torch(9, -20, 45, full_d, full_p)
    ctx.synchronize()
    ctx.bind_pair_torch(tl, tr)                          # drops the tables
    d = torch.full_like(full_d, 777.0); p = torch.full_like(full_d, 777.0)
    torch.cuda.synchronize()           # the fills run on torch's stream, the library on its own: order them
    order = [(600, 100), (0, 130), (300, 150), (130, 170), (440, 170), (250, 120), (0, 700)]
    for row0, n in order[:-1]:
        ctx.ncc_dense_torch(9, -20, 45, d, p, row0=row0, nrows=n)
    ctx.synchronize()
    covered = torch.zeros(h, dtype=torch.bool)
    for row0, n in order[:-1]:
        covered[row0:row0 + n] = True
    assert covered.all()
    bad = (d != full_d) | (p != full_p)
    ys, xs = torch.nonzero(bad, as_tuple=True)
    assert not bad.any(), "%d pixels differ: rows %s, columns %d..%d" % (
        int(bad.sum()), sorted(set(ys.cpu().tolist()))[:12], int(xs.min()), int(xs.max()))


def test_vector_kernels_behind_the_switch(tmp_path):
    """MISTEREO_NCC_MFMA=0 sends windows up to 21 to the vector kernels (the A/B switch of the benchmarks): the
    variable is read once per process, so the case runs in a child process; disparity, peak and second peak
    against the oracle like every other case."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = %r\n"
        "import numpy as np, mistereo, oracle, synth\n"
        "from ncc_util import compare_ncc, flat_mask\n"
        "left = synth.texture(70, 500, 4242); right = synth.shift_right_image(left, 63, fill_seed=7)\n"
        "ctx = mistereo.Context(0); ctx.upload_pair(left, right)\n"
        "for win, dmin, dmax in ((21, -128, 128), (9, -100, 27)):\n"
        "    gpu = ctx.ncc_dense(win, dmin, dmax, want_idisp=True, want_peak2=True)\n"
        "    ora = oracle.ncc_dense(left, right, win, dmin, dmax, want_peak2=True, want_volume=True)\n"
        "    ora['flat'] = flat_mask(ora['volume'], ora['idisp'], dmin)\n"
        "    st = compare_ncc(gpu, ora, volume=ora['volume'], dmin=dmin, check_peak2=True)\n"
        "    assert st['valid'] > 0\n"
        "print('ok')\n"
    ) % ([os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")],)
    env = dict(os.environ, MISTEREO_NCC_MFMA="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_page_locked_host_rasters(ctx):
    """mis_alloc_host: inputs uploaded and result strips copied back by DMA from / into page-locked rasters; the same
    bits as through pageable memory, rows outside the requested strip untouched.  mis_pin_host page-locks memory the
    caller already owns where the system allows it (it does not on every box: then the error is reported and the
    pageable path still works)."""
    h, w = 300, 1100
    left = synth.texture(h, w, 808)
    right = synth.shift_right_image(left, -19, fill_seed=9)
    ctx.upload_pair(left, right)
    want = ctx.ncc_dense(15, -40, 24, want_idisp=True, want_peak2=True)
    pl, pr = ctx.alloc_host((h, w), np.uint8), ctx.alloc_host((h, w), np.uint8)
    disp, peak, peak2 = (ctx.alloc_host((h, w), np.float32) for _ in range(3))
    idisp = ctx.alloc_host((h, w), np.int16)
    try:
        pl[:] = left; pr[:] = right
        for rep in range(2):
            disp[:] = 777.0; peak[:] = 777.0; peak2[:] = 777.0; idisp[:] = 77
            ctx.upload_pair(pl, pr)
            ctx.ncc_dense_rows(15, -40, 24, 0, h, disp, peak, idisp, peak2)
            for k, a in (("disp", disp), ("peak", peak), ("peak2", peak2)):
                assert np.array_equal(a.view(np.uint32), want[k].view(np.uint32)), k
            assert np.array_equal(idisp, want["idisp"])
            disp[:] = 777.0
            ctx.ncc_dense_rows(15, -40, 24, 100, 57, disp, peak)         # a strip: the other rows keep their contents
            assert np.array_equal(disp[100:157].view(np.uint32), want["disp"][100:157].view(np.uint32))
            assert (disp[:100] == 777.0).all() and (disp[157:] == 777.0).all()
        # uploads in a row: a pair still on its way (the page-locked upload returns at once) is not overtaken by the next
        # one, page-locked or pageable
        other = synth.shift_right_image(left, 7, fill_seed=3)
        ctx.upload_pair(left, other)
        want7 = ctx.ncc_dense(15, -40, 24, want_idisp=False)["disp"]
        pr2 = ctx.alloc_host((h, w), np.uint8)
        pr2[:] = other
        ctx.upload_pair(pl, pr); ctx.upload_pair(pl, pr2)
        ctx.ncc_dense_rows(15, -40, 24, 0, h, disp, peak)
        assert np.array_equal(disp.view(np.uint32), want7.view(np.uint32))
        ctx.upload_pair(pl, pr2); ctx.upload_pair(left, right)
        ctx.ncc_dense_rows(15, -40, 24, 0, h, disp, peak)
        assert np.array_equal(disp.view(np.uint32), want["disp"].view(np.uint32))
        ctx.free_host(pr2)
        # a page-locked output beside a pageable one: the staged path serves both
        pageable = np.full((h, w), 5.0, np.float32)
        ctx.ncc_dense_rows(15, -40, 24, 0, h, disp, pageable)
        assert np.array_equal(pageable.view(np.uint32), want["peak"].view(np.uint32))
        assert np.array_equal(disp.view(np.uint32), want["disp"].view(np.uint32))
    finally:
        for a in (pl, pr, disp, peak, peak2, idisp):
            ctx.free_host(a)
    own = np.full((h, w), 3.0, np.float32)
    try:
        ctx.pin_host(own)
    except mistereo.MisError as e:
        assert "hipHostRegister" in str(e)                   # this system does not page-lock foreign memory
        return
    ctx.pin_host(own)                                        # pinning a pinned range is not an error
    ctx.ncc_dense_rows(15, -40, 24, 0, h, own, np.empty((h, w), np.float32))
    assert np.array_equal(own.view(np.uint32), want["disp"].view(np.uint32))
    ctx.unpin_host(own)
    with pytest.raises(mistereo.MisError):
        ctx.unpin_host(own)                                  # not pinned any more


def test_exact_ties_of_negative_scores_are_the_one_declared_deviation(ctx):
    """The reference's rule is the FIRST maximum (image3.cpp:486-501): the lowest disparity among equal scores.  The
    matrix-core kernel keeps it while the tied scores are positive; an exact tie between NEGATIVE best scores inside one
    range of right columns may go to a higher disparity (the tag below a key's score bits orders the other way for
    negative floats).  Such pixels have no correlating candidate at all and fail the 0.75 accept threshold of
    image3.cpp:751, so no XYZ point changes.  Vertical stripes of period 7 under a 21-wide window, searched over
    d = 1..6, tie all six candidates exactly at -1/6: the deviation is pinned here -- a pixel may differ from the
    oracle's pick only by an exactly equal negative score, with an equal peak below the threshold -- and positive
    ties keep the first maximum."""
    h, w = 60, 300
    col = np.where(np.arange(w + 40) % 7 == 0, 200, 40).astype(np.uint8)
    base = np.repeat(col[None, :], h, axis=0)
    left = np.ascontiguousarray(base[:, 20:20 + w])
    right = left.copy()
    ctx.upload_pair(left, right)
    gpu = ctx.ncc_dense(21, 1, 6, want_idisp=True)
    ora = oracle.ncc_dense(left, right, 21, 1, 6, want_volume=True)
    valid = ora["idisp"] != -32768
    assert np.array_equal(gpu["idisp"] == -32768, ~valid) and valid.sum() > 0
    valid &= ~np.isnan(ora["volume"]).any(axis=2)            # pixels whose six candidates all fit the raster
    vol = ora["volume"][valid]
    assert np.all(vol.max(axis=1) < 0) and np.all(vol.max(axis=1) == vol.min(axis=1))      # six exactly equal negative scores
    assert np.all(ora["idisp"][valid] == 1)                                                  # the reference's first maximum
    gi = gpu["idisp"][valid].astype(int)
    assert np.all((gi >= 1) & (gi <= 6))
    deviating = int((gi != 1).sum())
    print("exact ties of negative scores resolved to a higher disparity: %d of %d pixels (picks %s)" % (deviating, gi.size, np.unique(gi)))
    assert np.array_equal(vol[np.arange(gi.size), gi - 1], vol[:, 0])                        # an exactly equal score, nothing else
    assert np.allclose(gpu["peak"][valid], ora["peak"][valid], rtol=1e-6, atol=1e-7)
    assert np.all(gpu["peak"][valid] < 0.75)
    # the same stripes searched over d = 0..13 tie positively at d = 0, 7 (score 1): the first maximum, d = 0, is kept
    gpu = ctx.ncc_dense(21, 0, 13, want_idisp=True)
    ora = oracle.ncc_dense(left, right, 21, 0, 13)
    valid = ora["idisp"] != -32768
    assert np.all(ora["idisp"][valid] == 0) and np.all(gpu["idisp"][valid] == 0)
