"""BASELINE config[1] at full size (8192 x 8192, 8-bit, 21 x 21 window, +-256 search): properties that
do not need the oracle on the whole frame, plus the oracle on a band of rows."""
import numpy as np
import pytest

import oracle
import synth
from ncc_util import compare_ncc

pytestmark = pytest.mark.gpu

SIZE, WIN, D, SHIFT = 8192, 21, 256, 37


@pytest.fixture(scope="module")
def full(ctx):
    left = synth.texture(SIZE, SIZE, 0xC0FFEE + 2)
    right = synth.shift_right_image(left, SHIFT, fill_seed=77)
    ctx.upload_pair(left, right)
    out = ctx.ncc_dense(WIN, -D, D, want_idisp=True)
    return left, right, out


def test_known_shift_recovered_everywhere(full):
    left, right, out = full
    r = WIN // 2
    idisp, peak, disp = out["idisp"], out["peak"], out["disp"]
    # frame border invalid, everything else valid
    assert np.all(idisp[:r] == -32768) and np.all(idisp[-r:] == -32768)
    assert np.all(idisp[:, :r] == -32768) and np.all(idisp[:, -r:] == -32768)
    inner = idisp[r:-r, r:-r]
    assert np.all(inner != -32768)
    # where the right window holds copied left content (x - SHIFT - r >= 0, x + r < W - SHIFT ...)
    core = (slice(r, SIZE - r), slice(SHIFT + 2 * r, SIZE - SHIFT - 2 * r))
    assert (idisp[core] == SHIFT).mean() >= 0.9999
    assert np.abs(peak[core][idisp[core] == SHIFT] - 1.0).max() <= 1e-4
    # the parabola through asymmetric neighbours does not peak exactly on the integer: small, bounded
    dev = np.abs(disp[core][idisp[core] == SHIFT] - SHIFT)
    assert dev.max() <= 0.3 and dev.mean() <= 0.02


def test_deterministic_and_strip_invariant(ctx, full):
    """A second run gives the same bits; two half-frame strips give the same bits as the full frame."""
    import torch
    left, right, out = full
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.bind_pair_torch(tl, tr)
    d = torch.empty((SIZE, SIZE), dtype=torch.float32, device="cuda")
    p = torch.empty_like(d)
    ctx.ncc_dense_torch(WIN, -D, D, d, p)
    ctx.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint32), out["disp"].view(np.uint32))
    assert np.array_equal(p.cpu().numpy().view(np.uint32), out["peak"].view(np.uint32))
    d2 = torch.full_like(d, -1.0)
    p2 = torch.full_like(d, -1.0)
    torch.cuda.synchronize()           # the fills run on torch's stream, the library on its own: order them
    ctx.ncc_dense_torch(WIN, -D, D, d2, p2, row0=0, nrows=5000)
    ctx.ncc_dense_torch(WIN, -D, D, d2, p2, row0=5000, nrows=SIZE - 5000)
    ctx.synchronize()
    assert torch.equal(d, d2) and torch.equal(p, p2)


def test_band_of_rows_against_the_oracle(full):
    left, right, out = full
    r = WIN // 2
    y0, n = 4000, 48
    crop_l = np.ascontiguousarray(left[y0 - r:y0 + n + r])
    crop_r = np.ascontiguousarray(right[y0 - r:y0 + n + r])
    ora = oracle.ncc_dense(crop_l, crop_r, WIN, -D, D)
    gpu = {k: np.ascontiguousarray(v[y0 - r:y0 + n + r]).copy() for k, v in out.items() if v is not None}
    # rows whose window leaves the crop are invalid in the oracle's crop only: compare the inner rows
    for k in gpu:
        gpu[k] = gpu[k][r:-r]
    ora = {k: v[r:-r] for k, v in ora.items() if isinstance(v, np.ndarray)}
    compare_ncc(gpu, ora)


def test_config2_tiles_recover_their_offsets(ctx):
    """BASELINE config[2]: 256 tiles of 512 x 512 over an 8192 x 8192 pair, a different integer offset
    in [-100,100]^2 per tile: every offset recovered, the batched call equal to single-tile calls and
    to the oracle on a few tiles."""
    size, tile = 8192, 512
    seed = 0xC0FFEE + 3
    left = synth.texture(size, size, seed)
    nt = size // tile
    rnd = synth.splitmix64(seed + 1, 2 * nt * nt).reshape(nt * nt, 2)
    offs = (rnd % np.uint64(201)).astype(np.int64) - 100
    right = np.empty_like(left)
    pad = np.pad(left, 128, mode="reflect")
    rects = []
    for k in range(nt * nt):
        j, i = divmod(k, nt)
        dx, dy = int(offs[k, 0]), int(offs[k, 1])
        y0, x0 = j * tile + 128 + dy, i * tile + 128 + dx
        right[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile] = pad[y0:y0 + tile, x0:x0 + tile]
        rects.append((i * tile, j * tile, tile, tile))
    ctx.upload_pair(left, right)
    got = ctx.find_offsets(rects, rects, tile, tile)
    assert np.array_equal(got["dx"], offs[:, 0]) and np.array_equal(got["dy"], offs[:, 1])
    for k in (0, 17, 255):
        peak, dx, dy = ctx.find_offset(rects[k], rects[k], tile, tile)
        assert (dx, dy) == (got["dx"][k], got["dy"][k]) and np.float32(peak) == got["peak"][k]
        ref = oracle.find_offset(left, right, rects[k], rects[k], tile, tile, precision=1)
        assert (dx, dy) == (ref["dx"], ref["dy"])
        assert abs(peak - ref["peak"]) <= 1e-4 * max(abs(ref["peak"]), 1e-3) + 1e-5


def test_a_strip_builds_only_its_rows_of_tables(ctx, full):
    """Row strip of 1024 rows of the 8192-row frame (what one of 8 ranks computes): the table kernels
    cover the strip's rows plus the window halo, rounded to 64-row bands, so their device time is about
    an eighth of the full frame's (measured with the library's own HIP events)."""
    import torch
    left, right, out = full
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    d = torch.empty((SIZE, SIZE), dtype=torch.float32, device="cuda"); p = torch.empty_like(d)
    torch.cuda.synchronize()
    ctx.profile(True)
    times = {}
    for name, (row0, n) in (("warm", (0, SIZE)), ("full", (0, SIZE)), ("strip", (3072, 1024))):
        ctx.bind_pair_torch(tl, tr)
        ctx.profile_read()
        ctx.ncc_dense_torch(WIN, -D, D, d, p, row0=row0, nrows=n)
        tables_ms, main_ms, refine_ms, calls = ctx.profile_read()
        times[name] = (tables_ms, main_ms)
    ctx.profile(False)
    assert np.array_equal(d[3072:4096].cpu().numpy().view(np.uint32), out["disp"][3072:4096].view(np.uint32))
    assert times["strip"][0] <= 0.3 * times["full"][0], times
    assert times["strip"][1] <= 0.2 * times["full"][1], times


def test_config3_strip_of_a_16384_frame(ctx):
    """BASELINE config[3]: one 16384 x 16384 8-bit pair, 21 x 21 window, +-256 search, sharded by row
    strips: the strip one of 8 ranks computes (2048 rows from row 7168) inside the full-size frame
    (64-bit row offsets, 1.1 GB tables for the strip only), bit for bit equal to the same rows computed
    as a raster of their own (strip + halo, what a rank that holds only its strip binds), and against
    the oracle on a 160-row band of it."""
    import torch
    W = H = 16384
    row0, n, r = 7168, 2048, WIN // 2
    band_l = synth.texture(n + 2 * r, W, 0xC0FFEE + 3)
    band_r = synth.shift_right_image(band_l, synth.planar_disparity(n + 2 * r, W, -200, 200, 64, 5), fill_seed=6)
    tl = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    tr = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    tl[row0 - r:row0 + n + r] = torch.from_numpy(band_l).cuda()
    tr[row0 - r:row0 + n + r] = torch.from_numpy(band_r).cuda()
    d = torch.full((H, W), -7.0, dtype=torch.float32, device="cuda"); p = torch.full_like(d, -7.0)
    torch.cuda.synchronize()
    ctx.bind_pair_torch(tl, tr)
    ctx.ncc_dense_torch(WIN, -D, D, d, p, row0=row0, nrows=n)
    ctx.synchronize()
    assert float(d[row0 - 1].max()) == -7.0 and float(d[row0 + n].max()) == -7.0      # rows outside the strip untouched
    strip_d, strip_p = d[row0:row0 + n].cpu().numpy(), p[row0:row0 + n].cpu().numpy()
    del d, p, tl, tr
    torch.cuda.empty_cache()
    # the same rows as a raster of their own
    bl, br = torch.from_numpy(band_l).cuda(), torch.from_numpy(band_r).cuda()
    d2 = torch.empty((n + 2 * r, W), dtype=torch.float32, device="cuda"); p2 = torch.empty_like(d2)
    i2 = torch.empty((n + 2 * r, W), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.bind_pair_torch(bl, br)
    ctx.ncc_dense_torch(WIN, -D, D, d2, p2, i2)
    ctx.synchronize()
    assert np.array_equal(d2[r:r + n].cpu().numpy().view(np.uint32), strip_d.view(np.uint32))
    assert np.array_equal(p2[r:r + n].cpu().numpy().view(np.uint32), strip_p.view(np.uint32))
    # the oracle on 160 rows of the band
    y0, m = 900, 160
    cl, cr = np.ascontiguousarray(band_l[y0 - r:y0 + m + r]), np.ascontiguousarray(band_r[y0 - r:y0 + m + r])
    ora = oracle.ncc_dense(cl, cr, WIN, -D, D)
    gd = np.full(cl.shape, -32768, np.float32); gp = np.zeros(cl.shape, np.float32); gi = np.full(cl.shape, -32768, np.int16)
    gd[r:-r] = d2[y0:y0 + m].cpu().numpy(); gp[r:-r] = p2[y0:y0 + m].cpu().numpy(); gi[r:-r] = i2[y0:y0 + m].cpu().numpy()
    rep = oracle.parity_report(gd, gp, ora, cl, cr, WIN, -D, D, gpu_idisp=gi)
    assert rep["validity_masks_equal"] and rep["argmax_agree"] >= 0.999 and rep["argmax_diffs_all_near_tie"], rep
    assert rep["peak_within_1e-4"] == 1.0, rep
    assert rep["subpx_outliers"] == rep["subpx_outliers_ill_conditioned"], rep


def test_config4_band_16384_wide_u16(ctx):
    """BASELINE config[4]'s frame width and search: a 16384-wide, 126-row 16-bit band, 31 x 31 window,
    +-512 search (1025 disparities, 17 chunks), subpixel shift 37.25 px, against the oracle."""
    W, Hb, win, Dm = 16384, 126, 31, 512
    rng = np.random.default_rng(9)
    left = synth.texture(Hb, W, 0xC0FFEE + 5, np.uint16).astype(np.int64) + rng.integers(-120, 121, (Hb, W))
    left = np.clip(left, 0, 65535).astype(np.uint16)
    right = synth.shift_right_image_subpixel(left, 37.25)
    right[:, 9000:] = synth.shift_right_image(left, -480, fill_seed=3)[:, 9000:]      # a plane near the end of the range
    ctx.upload_pair(left, right)
    out = ctx.ncc_dense(win, -Dm, Dm)
    ora = oracle.ncc_dense(left, right, win, -Dm, Dm)
    rep = oracle.parity_report(out["disp"], out["peak"], ora, left, right, win, -Dm, Dm, gpu_idisp=out["idisp"])
    assert rep["validity_masks_equal"] and rep["argmax_agree"] >= 0.999 and rep["argmax_diffs_all_near_tie"], rep
    assert rep["peak_within_1e-4"] == 1.0, rep
    assert rep["subpx_outliers"] == rep["subpx_outliers_ill_conditioned"], rep
    core = out["idisp"][20:-20, 600:8000]
    assert (core == 37).mean() > 0.99
    sub = out["disp"][20:-20, 600:8000][core == 37]
    assert abs(float(sub.mean()) - 37.25) < 0.05           # linear resampling biases the parabola slightly
