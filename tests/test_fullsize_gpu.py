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
