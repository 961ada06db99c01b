"""mis_triangulate_disparity_noalloc (the entry point that allocates nothing; it has no counterpart
in the reference) against mis_triangulate_disparity_device on the same inputs: XYZ bit for bit, the
same matched count; two calls in a row, the second on a smaller raster (the statically allocated
counter slots keep the first call's contents until the second clears them)."""
import numpy as np
import pytest

import mistereo
import synth

pytestmark = pytest.mark.gpu


def _cams(w, h):
    A = [[10000.0, 0.0, w / 2.0], [0.0, 10000.0, h / 2.0], [0.0, 0.0, 1.0]]
    R = np.eye(3)
    return (mistereo.make_camera(A, R, -R @ np.array([0.0, 0.0, 3000.0])),
            mistereo.make_camera(A, R, -R @ np.array([60.0, 0.0, 3000.0])))


def _case(ctx, h, w, seed, shift):
    import torch
    left = synth.texture(h, w, seed)
    right = synth.shift_right_image(left, shift, fill_seed=seed + 1)
    right[: h // 2, : w // 3] = synth.texture(h, w, seed + 2)[: h // 2, : w // 3]      # unmatched area: low peaks
    tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
    ctx.bind_pair_torch(tl, tr)
    disp = torch.empty((h, w), dtype=torch.float32, device="cuda")
    peak = torch.empty_like(disp)
    torch.cuda.synchronize()
    ctx.ncc_dense_torch(11, -40, 40, disp, peak)
    c1, c2 = _cams(w, h)
    want_xyz = torch.full((h, w, 3), 3.0, dtype=torch.float32, device="cuda")
    got_xyz = torch.full((h, w, 3), 5.0, dtype=torch.float32, device="cuda")
    got_n = torch.full((1,), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    want_n = ctx.triangulate_disparity_torch(c1, c2, disp, peak, 0.75, want_xyz)
    ctx.triangulate_disparity_noalloc_torch(c1, c2, disp, peak, 0.75, got_xyz, got_n)
    ctx.synchronize()
    a, b = want_xyz.cpu().numpy(), got_xyz.cpu().numpy()
    nan = np.isnan(a)
    assert np.array_equal(np.isnan(b), nan)
    assert np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan])
    assert int(got_n.item()) == want_n
    return want_n


def test_noalloc_equals_the_allocating_entry_point_twice_in_a_row():
    ctx = mistereo.Context(0)
    try:
        n1 = _case(ctx, 300, 900, 11, 7)
        n2 = _case(ctx, 120, 260, 12, -5)          # smaller input right after: stale counts would show
        assert 0 < n2 < n1 < 300 * 900
    finally:
        ctx.close()


def test_noalloc_count_is_optional_and_limits_are_checked():
    import torch
    ctx = mistereo.Context(0)
    try:
        h, w = 40, 100
        disp = torch.full((h, w), 3.0, dtype=torch.float32, device="cuda")
        peak = torch.ones_like(disp)
        xyz = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        c1, c2 = _cams(w, h)
        torch.cuda.synchronize()
        ctx.triangulate_disparity_noalloc_torch(c1, c2, disp, peak, 0.75, xyz, None)
        ctx.synchronize()
        assert not torch.isnan(xyz).any()
        lib = ctx._lib
        import ctypes as C
        rc = lib.mis_triangulate_disparity_noalloc(ctx._h, C.byref(c1), C.byref(c2), C.c_void_p(disp.data_ptr()),
                                                   C.c_void_p(peak.data_ptr()), C.c_float(0.75), 70000, 4,
                                                   C.c_void_p(xyz.data_ptr()), None)
        assert rc == 1 and b"outside the supported" in lib.mis_last_error(ctx._h)
        rc = lib.mis_triangulate_disparity_noalloc(ctx._h, C.byref(c1), C.byref(c2), C.c_void_p(disp.data_ptr()),
                                                   C.c_void_p(peak.data_ptr()), C.c_float(0.75), 4, 65537,
                                                   C.c_void_p(xyz.data_ptr()), None)
        assert rc == 1
    finally:
        ctx.close()
