"""The library's two-stream paths in a process that holds more streams than the device has hardware queues.

A process gets 4 hardware queues (GPU_MAX_HW_QUEUES); streams beyond that share them.  The dense NCC forks to a side
stream (tables, edge ranges) and the tile correlation runs the two halves of a large batch on two streams; both join
through events.  With a HIGH-priority side stream a join could starve: the waiting stream kept its queue and the
stream it waited for was never scheduled.  The side streams therefore have the caller's priority, and this test
builds the condition on purpose instead of leaving it to the size of the suite: six more streams with work queued
on each, then the strip case on two caller streams and a 64-tile batch, under a 60 s limit.
"""
import numpy as np
import pytest

import mistereo
import oracle
import synth

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(60, method="thread")
def test_side_streams_join_when_streams_outnumber_hardware_queues():
    import torch
    extra = [torch.cuda.Stream() for _ in range(6)]
    busy = [torch.randn(2048, 2048, device="cuda") for _ in extra]
    keep = []
    c = mistereo.Context(0)
    try:
        h, w = 600, 1500
        left = synth.texture(h, w, 5)
        right = synth.shift_right_image(left, 21, fill_seed=6)
        tl, tr = torch.from_numpy(left).cuda(), torch.from_numpy(right).cuda()
        d1 = torch.empty((h, w), dtype=torch.float32, device="cuda"); p1 = torch.empty_like(d1)
        d2 = torch.empty_like(d1); p2 = torch.empty_like(d1)
        torch.cuda.synchronize()
        for rep in range(3):
            for s, b in zip(extra, busy):               # work queued on every extra stream while the library runs
                with torch.cuda.stream(s):
                    for _ in range(8):
                        keep.append(b @ b)
                    keep.clear()
            s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
            c.bind_pair_torch(tl, tr)
            c.set_stream(s1.cuda_stream)
            c.ncc_dense_torch(21, -64, 64, d1, p1, row0=0, nrows=300)       # builds the tables on s1 (side stream forked)
            c.set_stream(s2.cuda_stream)
            c.ncc_dense_torch(21, -64, 64, d2, p2, row0=300, nrows=300)     # reuses them on s2, ordered by an event
        torch.cuda.synchronize()
        c.set_stream(None)
        full_d = torch.empty_like(d1); full_p = torch.empty_like(d1)
        c.bind_pair_torch(tl, tr)
        c.ncc_dense_torch(21, -64, 64, full_d, full_p)
        c.synchronize()
        assert torch.equal(d1[:300], full_d[:300]) and torch.equal(d2[300:], full_d[300:])
        assert torch.equal(p1[:300], full_p[:300]) and torch.equal(p2[300:], full_p[300:])

        # tile correlation: 64 tiles of 128 x 128 take the two-stream path (two halves of the batch side by side)
        size, tile = 1024, 128
        big = synth.texture(size + 64, size + 64, 77)
        img2 = np.ascontiguousarray(big[32:32 + size, 32:32 + size])
        img1 = np.ascontiguousarray(big[32 - 7:32 - 7 + size, 32 + 11:32 + 11 + size])      # content moved by (-11, +7)
        rects = np.array([(i * tile, j * tile, tile, tile) for j in range(8) for i in range(8)], np.int32)
        t1, t2 = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
        d_rects = torch.from_numpy(rects).cuda()
        out = torch.zeros((64, 4), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        s3 = torch.cuda.Stream()
        c.set_stream(s3.cuda_stream)
        c.bind_pair_torch(t1, t2)
        for s, b in zip(extra, busy):
            with torch.cuda.stream(s):
                keep.append(b @ b)
        for _ in range(3):
            c.find_offsets_torch(d_rects, d_rects, tile, tile, out)
        torch.cuda.synchronize()
        c.set_stream(None)
        got = out.cpu().numpy()
        for k in (0, 9, 63):
            ref = oracle.find_offset(img1, img2, tuple(rects[k]), tuple(rects[k]), tile, tile, precision=0)
            assert (int(got[k, 0]), int(got[k, 1])) == (ref["dx"], ref["dy"]) == (-11, 7)
        assert (got[:, 0] == -11).all() and (got[:, 1] == 7).all()
    finally:
        c.close()
