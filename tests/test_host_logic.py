"""Host-side logic that needs no GPU: synthetic generators, output writers, work partition and the
rank-0 gather (world_size 2 over gloo)."""
import os
import sys

import numpy as np
import pytest

import mistereo
import partition
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_deterministic_and_in_range():
    a = synth.texture(64, 96, 0xC0FFEE)
    b = synth.texture(64, 96, 0xC0FFEE)
    assert np.array_equal(a, b) and a.min() >= 16 and a.max() <= 240
    assert synth.splitmix64(1, 3).tolist() == [10451216379200822465, 13757245211066428519, 17911839290282890590]
    r = synth.shift_right_image(a, 5, fill_seed=2)
    assert np.array_equal(r[:, :-5], a[:, 5:])               # right(x') = left(x' + d)
    u16 = synth.texture(32, 32, 3, np.uint16)
    assert u16.dtype == np.uint16 and u16.max() > 255


def test_write_xyz_format(tmp_path):
    xyz = np.array([[1.5, -2.25, 3.0], [1e6, 0.1, -7.125]], np.float32)
    bands = np.array([[10, 20, 30], [255, 0, 1]], np.float32)
    p = os.path.join(tmp_path, "a.xyz")
    mistereo.write_xyz(p, xyz, bands)
    mistereo.write_xyz(p, xyz[:1], bands[:1], append=True)
    lines = open(p).read().split("\n")
    assert lines[0] == "1.500000 -2.250000 3.000000 10.000000 20.000000 30.000000 "   # image3.cpp:842-849
    assert lines[1] == "%f %f %f %f %f %f " % (1e6, np.float32(0.1), -7.125, 255, 0, 1)
    assert lines[2] == lines[0] and lines[3] == ""
    # 7 floats per line is what xyzrgb2pcd.cpp:49 parses when the raster has 4 bands
    mistereo.write_xyz(p, xyz, None)
    assert open(p).read().split("\n")[0] == "1.500000 -2.250000 3.000000 "


def test_write_xyz_reference_truncation(tmp_path):
    """The reference formats the band values into 10*n_bands bytes (image3.cpp:830-846): the band text
    of a line is cut after 10*n_bands-1 characters.  ref_truncate reproduces that byte for byte; the
    default writes every band in full."""
    xyz = np.array([[1.0, 2.0, 3.0]], np.float32)
    p = os.path.join(tmp_path, "t.xyz")
    # one band >= 100: 11 characters wanted, 9 kept
    mistereo.write_xyz(p, xyz, np.array([[123.0]], np.float32), ref_truncate=True)
    assert open(p).read() == "1.000000 2.000000 3.000000 123.00000\n"
    # three 8-bit bands >= 100: 33 wanted, 29 kept
    mistereo.write_xyz(p, xyz, np.array([[200.0, 150.0, 100.0]], np.float32), ref_truncate=True)
    assert open(p).read() == "1.000000 2.000000 3.000000 200.000000 150.000000 100.000\n"
    # four 16-bit bands: 13 characters each, the fourth band disappears (39 = 10*4-1 kept)
    mistereo.write_xyz(p, xyz, np.array([[12345.0, 23456.0, 34567.0, 45678.0]], np.float32), ref_truncate=True)
    assert open(p).read() == "1.000000 2.000000 3.000000 12345.000000 23456.000000 34567.000000 \n"
    # values below 10 fit: 9 characters per band, nothing is cut
    mistereo.write_xyz(p, xyz, np.array([[1.0, 2.5, 9.0]], np.float32), ref_truncate=True)
    assert open(p).read() == "1.000000 2.000000 3.000000 1.000000 2.500000 9.000000 \n"
    mistereo.write_xyz(p, xyz, np.array([[12345.0, 23456.0, 34567.0, 45678.0]], np.float32))
    assert open(p).read() == "1.000000 2.000000 3.000000 12345.000000 23456.000000 34567.000000 45678.000000 \n"


def test_rasterize_leaves():
    leaves = [(0, 0, 4, 4, 3, 4, 0, 0, 0, 0, 0.9), (2, 2, 4, 4, -1, 0, 0, 0, 0, 0, 0.8)]
    ras = mistereo.rasterize_leaves(leaves, 8, 8, fill=-9.0)
    assert ras[0, 0, 0] == 5.0 and ras[1, 0, 0] == 3 and ras[2, 0, 0] == 4      # sqrtf(9+16), tx, ty
    assert ras[0, 3, 3] == 1.0 and ras[1, 3, 3] == -1                          # the later leaf overwrites
    assert ras[0, 7, 7] == -9.0                                                # untouched


@pytest.mark.parametrize("h,world", [(8192, 8), (100, 3), (5, 8), (16384, 2)])
def test_row_strips_cover_the_frame_once(h, world):
    strips = partition.row_strips(h, world)
    assert len(strips) == world and sum(n for _, n in strips) == h
    row = 0
    for r0, n in strips:
        assert r0 == row
        row += n
    lo, n = partition.strip_input_rows(*strips[-1], h, 21)
    assert lo + n <= h and lo >= 0
    assert partition.frames_for_rank(16, 3, 8) == [3, 11]


def _gloo_worker(rank, world, port, h, w, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(h * w, dtype=torch.float32).reshape(h, w)      # stands for a disparity raster
        strips = partition.row_strips(h, world)
        r0, n = strips[rank]
        out = partition.gather_rows(full[r0:r0 + n].clone(), strips, rank, world)
        if rank == 0:
            q.put(bool(torch.equal(out, full)))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, 37, 16, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _pipeline_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bufs = [torch.zeros((2, 8, 8)) for _ in range(2)]
        pipe = bench.GatherPipeline(rank, world, bufs)
        ok = True
        for step in range(5):                       # the bench loop: reuse, compute, submit
            b = step % 2
            pipe.reuse(b)
            bufs[b].fill_(100 * step + rank)
            pipe.submit(b)
        pipe.drain()
        if rank == 0:
            # the last two steps' gathers are still in the buffers
            for step in (3, 4):
                for r in range(world):
                    ok = ok and bool((pipe.gathered[step % 2][r] == 100 * step + r).all())
            q.put(ok)
    finally:
        dist.destroy_process_group()


def test_bench_gather_pipeline_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _strip_job_worker(rank, world, port, h, w, window, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r = window // 2
        state = {"step": 0}
        job_box = []

        def compute(disp, peak, local_row0, nrows):
            # stands for the library call: every output row gets a value made of its GLOBAL row index and
            # the step, computed from the rank's own strip geometry as the real call would
            job = job_box[0]
            assert disp.shape[0] >= local_row0 + nrows
            rows = torch.arange(job.r0, job.r0 + nrows, dtype=torch.float32)[:, None]
            disp[local_row0:local_row0 + nrows] = rows * 10 + state["step"]
            peak[local_row0:local_row0 + nrows] = -rows - 1000 * state["step"]

        job = bench.StripJob(rank, world, h, w, window, "cpu", compute)
        job_box.append(job)
        # the rank holds its strip plus the window halo, clipped to the frame
        assert job.lo == max(0, job.r0 - r) and job.lo + job.nin == min(h, job.r0 + job.n + r)
        ok = True
        for i in range(5):                           # the bench loop
            state["step"] = i
            job.step(i)
        job.drain()
        if rank == 0:
            rows = torch.arange(h, dtype=torch.float32)[:, None].expand(h, w)
            for i in (3, 4):                         # both buffers still hold their last step
                d, p = job.result(i)
                ok = ok and bool(torch.equal(d, rows * 10 + i)) and bool(torch.equal(p, -rows - 1000 * i))
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 37), (3, 64)])
def test_bench_strip_job_gloo(world, h):
    """bench.py --partition strips with a stub in place of the library call: strip ownership, halo rows,
    double buffering and the point-to-point gather into rank 0's full rasters (uneven strips)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_strip_job_worker, args=(r, world, port, h, 16, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


OPENCV_YAML = """%YAML:1.0
A: !!opencv-matrix
   rows: 3
   cols: 3
   dt: d
   data: [ 1.0000000000000000e+04, 0., 4.0965000000000000e+03, 0.,
       1.0000250000000000e+04, 3.1205000000000000e+03, 0., 0., 1. ]
R: !!opencv-matrix
   rows: 3
   cols: 3
   dt: f
   data: [ 9.99847710e-01, -1.74524058e-02, 0., 1.74524058e-02,
       9.99847710e-01, 0., 0., 0., 1. ]
T: !!opencv-matrix
   rows: 3
   cols: 1
   dt: d
   data: [ -9.0012345678901234e+02, .5, -3000. ]
"""


def test_camera_yaml_reader(tmp_path):
    """OpenCV FileStorage YAML as image3.cpp:1014-1030 reads it (A, R, T; doubles narrowed to float)."""
    import mistereo
    path = os.path.join(tmp_path, "left.tif.yaml")
    with open(path, "w") as f:
        f.write(OPENCV_YAML)
    cam = mistereo.load_camera_yaml(path)
    A = np.array(cam.A[:], np.float32).reshape(3, 3)
    assert np.array_equal(A, np.array([[1e4, 0, 4096.5], [0, 10000.25, 3120.5], [0, 0, 1]], np.float32))
    R = np.array(cam.R[:], np.float32).reshape(3, 3)
    assert R[0, 0] == np.float32(9.99847710e-01) and R[0, 1] == np.float32(-1.74524058e-02) and R[2, 2] == 1
    assert np.array_equal(np.array(cam.T[:], np.float32),
                          np.array([-9.0012345678901234e+02, 0.5, -3000.0]).astype(np.float32))
    with pytest.raises(mistereo.MisError):
        mistereo.load_camera_yaml(os.path.join(tmp_path, "missing.yaml"))
    bad = os.path.join(tmp_path, "bad.yaml")
    with open(bad, "w") as f:
        f.write(OPENCV_YAML.replace("T: !!opencv-matrix", "Q: !!opencv-matrix"))
    with pytest.raises(mistereo.MisError):
        mistereo.load_camera_yaml(bad)


def _read_tiff(path):
    """Minimal classic-TIFF reader for the tags the writer emits."""
    import struct
    raw = open(path, "rb").read()
    assert raw[:4] == b"II*\x00"
    (ifd,) = struct.unpack_from("<I", raw, 4)
    (n,) = struct.unpack_from("<H", raw, ifd)
    size = {3: 2, 4: 4}
    tags = {}
    for i in range(n):
        tid, typ, cnt, val = struct.unpack_from("<HHII", raw, ifd + 2 + 12 * i)
        if size[typ] * cnt <= 4:
            vals = list(struct.unpack_from("<%d%s" % (cnt, "H" if typ == 3 else "I"), raw, ifd + 2 + 12 * i + 8))
        else:
            vals = list(struct.unpack_from("<%d%s" % (cnt, "H" if typ == 3 else "I"), raw, val))
        tags[tid] = vals
    return raw, tags


def test_tiff_writer_band_interleaved_float32(tmp_path):
    """3-band Float32 TIFF with INTERLEAVE=BAND (image3.cpp:872-893)."""
    import mistereo
    rng = np.random.default_rng(4)
    planes = rng.normal(size=(3, 37, 53)).astype(np.float32)
    path = os.path.join(tmp_path, "disp.tif")
    mistereo.write_tiff_f32(path, planes)
    raw, tags = _read_tiff(path)
    assert tags[256] == [53] and tags[257] == [37]
    assert tags[258] == [32, 32, 32] and tags[339] == [3, 3, 3]       # Float32 samples
    assert tags[277] == [3] and tags[284] == [2] and tags[259] == [1]  # 3 samples, planar, uncompressed
    assert tags[279] == [37 * 53 * 4] * 3
    for b in range(3):
        got = np.frombuffer(raw, np.float32, 37 * 53, tags[273][b]).reshape(37, 53)
        assert np.array_equal(got, planes[b])
    # a single band is readable by an independent TIFF reader
    from PIL import Image
    one = os.path.join(tmp_path, "one.tif")
    mistereo.write_tiff_f32(one, planes[0])
    assert np.array_equal(np.asarray(Image.open(one)), planes[0])


def test_bench_starts_its_own_rank_processes():
    """`python bench.py --gpus 2` without a launcher's environment starts the two rank processes itself (a child
    torch.distributed.run), relays rank 0's line and returns the ranks' exit status.  --rehearse keeps the GPU out of
    it: gloo, CPU tensors, a stand-in for the library call, the measured run's own gather objects."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["value"] is None
    assert line["frames_gather_ok"] is True
    assert line["strips"]["n_gpus"] == 2 and line["strips"]["gathered_equals_full_frame_bits"] is True
    assert sum(line["strips"]["rows_per_rank"]) == 96
    # a launcher's environment that disagrees with --gpus is an error, not a silent single-rank run
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env2, timeout=120)
    assert p.returncode == 2
