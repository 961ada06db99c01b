#!/usr/bin/env python3
"""Dense-NCC throughput benchmark (BASELINE.json metric) for libmistereo on MI355X.

One step = one full pass of the dense windowed-NCC hot path over one synthetic stereo pair that
is already resident in HBM: frame-border fill, window-statistics tables, main correlation kernel
(argmax), refinement (exact scores, subpixel).  Workload at N=1: BASELINE config[1] -- 8192x8192
8-bit pair, 21x21 window, +-256 px search.

With N>1 (one process per GPU, launched by torch.distributed.run) there are two partitions:
  --partition frames  (default)  every rank processes a pair of its own (weak scaling: frames are
                                 independent units, BASELINE config[3]'s batch of pairs);
  --partition strips             ONE pair for the whole job, sharded by row strips with the window
                                 halo (strong scaling): a rank holds and uploads only its strip plus
                                 halo rows and computes its rows.
In both, disparity and peak rasters are gathered to rank 0 over RCCL while the next step computes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--partition frames|strips]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Both forms work for N > 1: without a launcher's environment (no WORLD_SIZE) the first form starts the second as a
child process before it touches the GPU, relays the ranks' output and returns their worst exit status.

Prints ONE JSON line on rank 0.  At N=1 the line also carries, outside the headline value, the
FFT tile correlation (path_f) and the 16-bit dense NCC (ncc16) with their own rooflines and CPU
baselines (--extras off skips them).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle")]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
VECTOR_PEAK_TFLOPS = 157.3     # MI355X vector fp32 peak (packed fma)
ALG_BYTES_PER_PX_U8 = 10.0     # left + right pixel read once, disparity f32 + peak f32 written
ALG_BYTES_PER_PX_U16 = 12.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=8192, help="frame is size x size pixels")
    ap.add_argument("--window", type=int, default=21)
    ap.add_argument("--disp", type=int, default=256, help="search is +-disp pixels")
    ap.add_argument("--partition", choices=["frames", "strips"], default="frames",
                    help="with N>1: a pair per rank (weak scaling) or row strips of one pair (strong scaling)")
    ap.add_argument("--gather", choices=["on", "off"], default="on",
                    help="with N>1: gather disparity/peak rasters to rank 0 (overlapped)")
    ap.add_argument("--cpu-rows", type=int, default=8192,
                    help="rows of the frame the CPU oracle is timed on (0: skip the CPU baseline)")
    ap.add_argument("--extras", choices=["on", "off"], default="on",
                    help="at N=1: also measure the FFT tile correlation and the 16-bit dense NCC")
    ap.add_argument("--strips-leg", choices=["on", "off", "force"], default="on",
                    help="with N>1 in the frames partition: afterwards also time rank 0's pair sharded by row strips over the "
                         "same ranks (reported under \"strips\"); force: run that leg at N=1 too (one strip, no transfers)")
    ap.add_argument("--second-peak", choices=["on", "off"], default="on",
                    help="at N=1 with extras: also time the step with the second-peak output (off for counter passes that sum per step)")
    ap.add_argument("--cxx-gather", choices=["on", "off"], default="off",
                    help="strips leg: afterwards gather the last result once more through libmistereo_rccl.so (the C++ RCCL "
                         "gather of include/mi_stereo_rccl.h) and compare it with the torch.distributed one")
    ap.add_argument("--rehearse", action="store_true",
                    help="test only: no GPU and no library call -- the rank processes rendezvous over gloo and run the step loops of "
                         "both partitions with a stand-in for the correlation, to check the launcher, the gathers and the line's shape")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`bench.py --gpus N` (N > 1) started without a launcher's environment: start the N rank processes ourselves,
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD of this process, which has made no
    GPU call (torch is not even imported here), relay what the ranks print, and return the launcher's exit status
    (non-zero if any rank failed)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1" if args.rehearse else str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def rehearse(args, rank, world):
    """--rehearse: the rank processes of a multi-rank run without a GPU.  The same StripJob / GatherPipeline objects
    and the same loop order as the measured run, over gloo on CPU tensors, with a stand-in for the library call that
    writes values made of (rank, step, global row); rank 0 checks what it gathered and prints a line of the measured
    line's shape whose value is null.  Nothing here is a measurement."""
    import torch
    import torch.distributed as dist
    import partition
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    H = W = min(args.size, 96)
    nbuf = 2
    packed = [torch.zeros((2, H, W)) for _ in range(nbuf)]
    pipe = GatherPipeline(rank, world, packed) if world > 1 else None
    t0 = time.perf_counter()
    for i in range(args.warmup + args.steps):
        b = i % nbuf
        if pipe:
            pipe.reuse(b)
        packed[b][0].fill_(1000.0 * i + rank)
        packed[b][1].fill_(-1000.0 * i - rank)
        if pipe:
            pipe.submit(b)
    if pipe:
        pipe.drain()
        dist.barrier()
    dt = time.perf_counter() - t0
    last = args.warmup + args.steps - 1
    frames_ok = True
    if pipe and rank == 0 and last >= 0:
        for r in range(world):
            g = pipe.gathered[last % nbuf][r]
            frames_ok = frames_ok and bool((g[0] == 1000.0 * last + r).all()) and bool((g[1] == -1000.0 * last - r).all())
    # the strips partition on the same ranks
    state = {"step": 0}
    box = []

    def compute(disp, peak, local_row0, nrows):
        rows = torch.arange(box[0].r0, box[0].r0 + nrows, dtype=torch.float32)[:, None]
        disp[local_row0:local_row0 + nrows] = rows * 10 + state["step"]
        peak[local_row0:local_row0 + nrows] = -rows - 1000 * state["step"]

    job = StripJob(rank, world, H, W, args.window, "cpu", compute, gather=world > 1)
    box.append(job)
    for i in range(args.warmup + args.steps):
        state["step"] = i
        job.step(i)
    job.drain()
    if world > 1:
        dist.barrier()
    if rank == 0:
        rows = torch.arange(H, dtype=torch.float32)[:, None].expand(H, W)
        d, p = job.result(max(last, 0))
        strips_ok = bool(torch.equal(d, rows * 10 + max(last, 0))) and bool(torch.equal(p, -rows - 1000 * max(last, 0)))
        print(json.dumps({
            "metric": "NCC throughput", "value": None, "unit": "Gpx*disp/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "none: --rehearse runs no correlation (launcher, rendezvous and gathers only, gloo on the CPU)",
            "rehearsal": True, "config": {"workload": "rehearsal %dx%d" % (W, H), "partition": "frames", "gather_to_rank0": world > 1},
            "frames_gather_ok": frames_ok, "wall_s": round(dt, 3),
            "strips": {"partition": "strips", "scaling": "strong", "n_gpus": world,
                       "rows_per_rank": [n_ for _, n_ in partition.row_strips(H, world)], "halo_rows": args.window // 2,
                       "gathered_equals_full_frame_bits": strips_ok}}), flush=True)
        ok = frames_ok and strips_ok
    else:
        ok = True
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


STRIPS_LEG_LIMIT_S = 150      # the strips leg and the final barrier of a multi-rank run must come back within this


def load_traffic_note(kernel_name, field="bytes"):
    """HBM bytes per launch (or another field) of a kernel from a committed rocprofv3 --pmc run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(kernel_name)
        return entry.get(field) if isinstance(entry, dict) else (entry if field == "bytes" else None)
    except Exception:
        return None


def issue_floor(W, H, window, nd, kernel_s):
    """Time the argmax kernels' own instruction mix needs at full vector-issue rate.

    Windows up to 21 (matrix-core kernel): a scored wavefront row is 96 accumulator registers per lane, each
    one v_mul_f32, one v_fmac_f32, one v_and_or_b32 and half a v_max3_f32 (1.05 + 1.05 + 1.8 + 0.9 ns of SIMD
    time, prices measured with tools/ubench/valu_rates on this part at 8 wavefronts per SIMD, kept in
    profiles/), plus the issue slots of 18 MFMAs (8 cycles each); a warm-up row is the 18 MFMAs alone
    (32 cycles each at the 2.1 GHz the kernel holds).  Wavefronts: 64-column blocks x ranges of 96 right
    columns x bands of rows (the library's rule), 1024 SIMDs.
    Windows beyond 21 x 21: one more v_fma_f32 per register and 4 products per tile instead of 3."""
    r = window // 2
    rows = H - 2 * r
    blocks = -(-(W - 2 * r) // 64)
    ranges = (((nd + 30) >> 5) + 2 + 2) // 3
    band = 256
    while band > 32 and blocks * ranges * -(-rows // band) < 4 * 2048:
        band //= 2
    bands = -(-rows // band)
    # windows beyond 21 x 21: one more v_fma_f32 per register (the mean correction's integer part), no spare taps
    ks = (2 * r + 1 + (2 if window <= 21 else 0) + 7) // 8
    per_reg = 1.05 + 1.05 + 1.8 + 0.9 + (0.0 if window <= 21 else 1.05)
    scored_ns = 96 * per_reg + 6 * ks * 8 / 2.1
    warm_ns = 6 * ks * 32 / 2.1
    floor_s = blocks * ranges * (rows * scored_ns + bands * 2 * r * warm_ns) * 1e-9 / 1024
    # The same mix priced at the kernel's own occupancy (tools/ubench/mfma_valu_overlap, two wavefronts per SIMD):
    # v_mul / v_fmac / v_fma take 2 clocks of the vector pipe and do not overlap with products in flight, v_and_or /
    # v_max3 and the 85 other vector instructions of a row (operand selects, statistics, key merge; counted in the
    # ISA) take 4; a product 32.  Clock: 2.25 GHz, what SQ_BUSY_CYCLES over the kernel's duration gives (profiles/).
    fast = 2 * 96 + (96 if window > 21 else 0)
    scored_clk = fast * 2 + (96 + 48 + 85) * 4 + 6 * ks * 32
    warm_clk = 40 * 4 + 6 * ks * 32
    pipes_s = blocks * ranges * (rows * scored_clk + bands * 2 * r * warm_clk) / 2.25e9 / 1024
    return {"ms": round(floor_s * 1e3, 3), "frac": round(floor_s / kernel_s, 3) if kernel_s > 0 else None,
            "note": "vector-issue floor of the matrix-core kernel's instruction mix (prices at 8 wavefronts per SIMD; "
                    "the kernel's 96 accumulators allow 2) / measured kernel time",
            "two_wavefronts": {"ms": round(pipes_s * 1e3, 3), "frac": round(pipes_s / kernel_s, 3) if kernel_s > 0 else None,
                               "note": "vector + matrix pipe time of the same mix at 2 wavefronts per SIMD, where multiplications "
                                       "and products do not overlap (microbenchmark prices, 2.25 GHz): a model, not a bound"}}


def mfma_rate(W, H, window, nd, kernel_s):
    """Matrix-core work of the argmax kernels: one 32x32x16 f16 product per 8 window taps (up to 21 x 21: the
    window's taps and two spare ones that carry the mean correction), tile and row: 2 * 16 * KS flop per
    pixel and disparity, against the dense f16 peak."""
    if kernel_s <= 0:
        return None
    r = window // 2
    ks = (2 * r + 1 + (2 if window <= 21 else 0) + 7) // 8
    flop = 2.0 * 16 * ks * (W - 2 * r) * (H - 2 * r) * nd
    return {"flop": flop, "achieved_tflops": round(flop / kernel_s / 1e12, 1), "peak_tflops": 2500.0,
            "frac": round(flop / kernel_s / 2.5e15, 4), "note": "the products are not the bound: the vector unit that turns "
            "window sums into scores and keys is"}


class GatherPipeline:
    """Double-buffered gather of per-rank result rasters to rank 0, overlapped with the next step
    (frames partition: every rank contributes a whole raster).

    submit(b) starts an asynchronous torch.distributed.gather of buffer b; reuse(b) must be called
    before buffer b is written again and waits for that gather.  Works with any backend (RCCL on
    the GPU box, gloo in the CPU tests)."""

    def __init__(self, rank, world, buffers, dst=0):
        import torch
        self.rank, self.world, self.dst = rank, world, dst
        self.buffers = buffers
        self.pending = [None] * len(buffers)
        self.gathered = None
        if rank == dst:
            self.gathered = [[torch.empty_like(b) for _ in range(world)] for b in buffers]

    def reuse(self, b):
        if self.pending[b] is not None:
            self.pending[b].wait()
            self.pending[b] = None

    def submit(self, b):
        import torch.distributed as dist
        self.pending[b] = dist.gather(self.buffers[b], self.gathered[b] if self.rank == self.dst else None,
                                      dst=self.dst, async_op=True)

    def drain(self):
        for b in range(len(self.buffers)):
            self.reuse(b)


class StripJob:
    """One pair sharded by row strips (strips partition).  Rank r owns output rows [r0, r0+n) and
    holds rows [lo, lo+nin) of the frame (its strip plus the window halo) as a raster of its own;
    rank 0 owns the full result rasters and receives every other rank's rows straight into them
    (grouped point-to-point transfers, see partition.StripGather), double buffered so that the
    transfers of one step overlap the next step's correlation.

    compute(disp, peak, local_row0, nrows) fills rows [local_row0, local_row0+nrows) of the given
    rasters (addressed as the rank's own raster; on rank 0 that addressing equals the frame's);
    before_compute / after_compute order the device streams around it (None on the CPU)."""

    def __init__(self, rank, world, H, W, window, device, compute, before_compute=None, after_compute=None,
                 nbuf=2, gather=True):
        import torch
        import partition
        self.rank, self.world, self.nbuf = rank, world, nbuf
        self.strips = partition.row_strips(H, world)
        self.r0, self.n = self.strips[rank]
        self.lo, self.nin = partition.strip_input_rows(self.r0, self.n, H, window)
        self.local_row0 = self.r0 - self.lo
        rows_out = H if rank == 0 else max(self.nin, 1)
        self.out = [[torch.empty((rows_out, W), dtype=torch.float32, device=device) for _ in range(2)]
                    for _ in range(nbuf)]
        self.gathers = [partition.StripGather(self.strips, rank, world, full=self.out[b] if rank == 0 else None)
                        for b in range(nbuf)] if (gather and world > 1) else None
        self.compute, self.before_compute, self.after_compute = compute, before_compute, after_compute

    def step(self, i):
        b = i % self.nbuf
        if self.gathers:
            self.gathers[b].wait()                  # the buffer's previous transfers have drained
        if self.before_compute:
            self.before_compute()
        if self.n > 0:
            self.compute(self.out[b][0], self.out[b][1], self.local_row0, self.n)
        if self.after_compute:
            self.after_compute()
        if self.gathers:
            self.gathers[b].submit([t[self.local_row0:self.local_row0 + self.n] for t in self.out[b]])

    def drain(self):
        if self.gathers:
            for g in self.gathers:
                g.wait()

    def result(self, i):
        """Rank 0: the full (disparity, peak) rasters of step i."""
        return self.out[i % self.nbuf]


def bench_cameras(W, H):
    """Two synthetic nadir cameras (f = 10000 px, height 3000 m, baseline 900 m, R = I); the second principal point
    carries the f B / H = 3000 px between the origins of the two crops."""
    import numpy as np
    import mistereo
    A = [[10000.0, 0.0, W / 2.0], [0.0, 10000.0, H / 2.0], [0.0, 0.0, 1.0]]
    A2 = [[10000.0, 0.0, W / 2.0 + 10000.0 * 900.0 / 3000.0], [0.0, 10000.0, H / 2.0], [0.0, 0.0, 1.0]]
    return (mistereo.make_camera(A, np.eye(3), [0.0, 0.0, -3000.0]),
            mistereo.make_camera(A2, np.eye(3), [-900.0, 0.0, -3000.0]))


def strips_leg(args, rank, world, dev, ctx, stream, H, W, D, seed, full_frame, full_frame_matched):
    """The strips partition timed on the ranks of a frames-partition run: the pair of `seed` (rank 0's) sharded by
    row strips with halo, every rank holding only its strip, rows sent to rank 0 as grouped point-to-point transfers.
    full_frame: rank 0's (disparity, peak) of the same pair computed as one frame; the gathered rasters must equal
    it bit for bit.  Runs on every rank; returns the report on rank 0."""
    import numpy as np
    import torch
    import torch.distributed as dist
    import partition
    nd = 2 * D + 1
    left_h, right_h = synth_pair(H, W, D, seed)
    r0, n = partition.row_strips(H, world)[rank]
    lo, nin = partition.strip_input_rows(r0, n, H, args.window)
    left = torch.from_numpy(np.ascontiguousarray(left_h[lo:lo + nin])).to(dev)
    right = torch.from_numpy(np.ascontiguousarray(right_h[lo:lo + nin])).to(dev)
    del left_h, right_h

    def lib_waits_for_torch():
        stream.wait_stream(torch.cuda.current_stream())

    def torch_waits_for_lib():
        ev = torch.cuda.Event()
        ev.record(stream)
        torch.cuda.current_stream().wait_event(ev)

    def compute(disp_t, peak_t, local_row0, nrows):
        with torch.cuda.stream(stream):
            ctx.bind_pair_torch(left, right)                 # every step rebuilds the strip's tables
            ctx.ncc_dense_torch(args.window, -D, D, disp_t, peak_t, row0=local_row0, nrows=nrows)

    job = StripJob(rank, world, H, W, args.window, dev, compute, lib_waits_for_torch, torch_waits_for_lib, nbuf=2, gather=True)

    def drain():
        job.drain()
        stream.synchronize()
        torch.cuda.synchronize()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        job.step(i)
    drain()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        job.step(i)
    drain()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # every rank triangulates the rows it computed, where they are (raster row k is frame row k + its first input row):
    # the XYZ work is sharded like the correlation; the matched counts add up to the whole frame's
    cam1, cam2 = bench_cameras(W, H)
    own_d, own_p = job.out[max(args.steps - 1, 0) % job.nbuf]
    xyz = torch.empty((own_d.shape[0], W, 3), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        t_tri = time.perf_counter()
        cnt = ctx.triangulate_disparity_rows_torch(cam1, cam2, own_d, own_p, 0.75, xyz, job.local_row0, job.n,
                                                   0 if rank == 0 else job.lo) if job.n > 0 else 0
        stream.synchronize()
        t_tri = time.perf_counter() - t_tri
    del xyz
    tot = torch.tensor([float(cnt), t_tri], dtype=torch.float64, device=dev)
    if world > 1:
        tsum = tot.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(tot, op=dist.ReduceOp.MAX)
        cnt_all, t_tri = int(tsum[0].item()), float(tot[1].item())
    else:
        cnt_all = int(cnt)
    cxx = None
    if args.cxx_gather == "on":
        # the same rows once more through the C++ gather (ncclSend / ncclRecv posted in one group on the library's stream)
        import mistereo
        idt = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt = torch.frombuffer(bytearray(mistereo.RcclGather.unique_id()), dtype=torch.uint8).to(dev)
        if world > 1:
            dist.broadcast(idt, 0)
        rg = mistereo.RcclGather(bytes(idt.cpu().numpy().tobytes()), rank, world, dev.index)
        full2 = [torch.zeros((H, W), dtype=torch.float32, device=dev) for _ in range(2)] if rank == 0 else None
        torch.cuda.synchronize()
        t_c = time.perf_counter()
        rg.gather_rows(H, W, [own_d, own_p], job.local_row0, full2, stream.cuda_stream)
        stream.synchronize()
        t_c = time.perf_counter() - t_c
        if rank == 0:
            gd0, gp0 = job.result(max(args.steps - 1, 0))
            cxx = {"equals_torch_gather_bits": bool(torch.equal(full2[0].view(torch.int32), gd0.view(torch.int32))
                                                    and torch.equal(full2[1].view(torch.int32), gp0.view(torch.int32))),
                   "ms": round(t_c * 1e3, 3)}
        rg.close()
    if rank != 0:
        return None
    gd, gp = job.result(max(args.steps - 1, 0))
    same = bool(torch.equal(gd.view(torch.int32), full_frame[0].view(torch.int32))
                and torch.equal(gp.view(torch.int32), full_frame[1].view(torch.int32)))
    return {"partition": "strips", "scaling": "strong", "n_gpus": world, "steps": args.steps,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4),
            "value": round(float(H) * W * nd * args.steps / dt / 1e9, 3), "unit": "Gpx*disp/s",
            "rows_per_rank": [n_ for _, n_ in partition.row_strips(H, world)],
            "halo_rows": args.window // 2, "gathered_equals_full_frame_bits": same,
            "matched_points_all_ranks": cnt_all, "matched_points_equal_full_frame": bool(cnt_all == full_frame_matched),
            "triangulation_ms_per_rank_strip": round(t_tri * 1e3, 3), "cxx_rccl_gather": cxx,
            "note": "rank 0's pair of the timed run, ONE pair sharded by row strips (each rank holds its strip plus halo rows only), "
                    "rows gathered into rank 0's rasters as grouped point-to-point transfers overlapped with the next step"}


def synth_pair(H, W, D, seed):
    """Synthetic pair of BASELINE config[1]'s kind: texture, piecewise-constant disparity, +-2 DN noise."""
    import numpy as np
    import synth
    left = synth.texture(H, W, seed)
    dl, dh = (-min(200, D - 56), min(200, D - 56)) if D > 60 else (-D // 2, D // 2)
    dmap = synth.planar_disparity(H, W, dl, dh, 64, seed + 1)
    right = synth.shift_right_image(left, dmap, fill_seed=seed + 1000)
    noise = (synth.splitmix64(seed + 2, H * W) % np.uint64(5)).astype(np.int16).reshape(H, W) - 2
    right = np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return left, right


def bench_path_f(ctx, stream, dev, left_h, steps=10, cpu=True):
    """BASELINE config[2]: FFT tile correlation (findOffset, image3.cpp:346-517) of 512 x 512 tiles over
    the frame, a different offset in [-100,100]^2 per tile, one batched call per step; plus config[0]'s
    recursion (calcDisparityMap, image3.cpp:676-865) on a 2048 x 2048 pair."""
    import numpy as np
    import torch
    import oracle
    import synth
    from concurrent.futures import ThreadPoolExecutor
    size, tile = left_h.shape[0], 512
    nt = size // tile
    n = nt * nt
    rnd = synth.splitmix64(0xC0FFEE + 4, 2 * n).reshape(n, 2)
    offs = (rnd % np.uint64(201)).astype(np.int64) - 100
    right = np.empty_like(left_h)
    pad = np.pad(left_h, 128, mode="reflect")
    rects = np.empty((n, 4), np.int32)
    for k in range(n):
        j, i = divmod(k, nt)
        dx, dy = int(offs[k, 0]), int(offs[k, 1])
        y0, x0 = j * tile + 128 + dy, i * tile + 128 + dx
        right[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile] = pad[y0:y0 + tile, x0:x0 + tile]
        rects[k] = (i * tile, j * tile, tile, tile)
    del pad
    tl, tr = torch.from_numpy(left_h).to(dev), torch.from_numpy(right).to(dev)
    d_rects = torch.from_numpy(rects).to(dev)
    d_out = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        ctx.bind_pair_torch(tl, tr)
        for _ in range(2):
            ctx.find_offsets_torch(d_rects, d_rects, tile, tile, d_out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            ctx.find_offsets_torch(d_rects, d_rects, tile, tile, d_out)
        e1.record(stream)
    stream.synchronize()
    ms = e0.elapsed_time(e1) / steps
    got = d_out.cpu().numpy()
    recovered = int(((got[:, 0] == offs[:, 0]) & (got[:, 1] == offs[:, 1])).sum())
    N = tile * tile
    flop_tile = 3 * 5 * N * np.log2(N) + 2 * 2 * 151 * N + 16 * N      # three transforms, two Gaussian passes, pointwise
    alg_tile = 2 * N * 1 + 16
    res = {
        "workload": "BASELINE config[2]: %d tiles of %dx%d over a %dx%d 8-bit pair, offsets in [-100,100]^2" % (n, tile, tile, size, size),
        "ms_per_batch": round(ms, 4), "tiles_per_s": round(n / ms * 1e3, 1), "gpx_per_s": round(n * N / ms / 1e6, 3),
        "offsets_recovered": "%d/%d" % (recovered, n),
        "roofline": {"flop_per_tile": float(flop_tile), "achieved_tflops": round(n * flop_tile / ms / 1e9, 3),
                     "vector_peak_tflops": VECTOR_PEAK_TFLOPS, "vector_frac": round(n * flop_tile / ms / 1e9 / VECTOR_PEAK_TFLOPS, 4),
                     "algorithmic_bytes_per_tile": alg_tile, "hbm_achieved_gbs": round(n * alg_tile / ms / 1e6, 2),
                     "hbm_frac": round(n * alg_tile / ms / 1e6 / HBM_PEAK_GBS, 5),
                     "traffic": load_traffic_note("xcorr_batch")},
    }
    if not cpu:
        return res
    # CPU: the oracle's findOffset.  All cores: tiles side by side, one thread each.  Reference-shaped: what
    # image3.cpp's structure gives -- one tile after another, allocation and set-up inside every call, one thread.
    cores = oracle.num_threads()
    ncpu = min(n, 2 * cores)

    def one(k):
        oracle.set_num_threads(1)
        r = oracle.find_offset(left_h, right, tuple(rects[k]), tuple(rects[k]), tile, tile, precision=0)
        return (r["dx"], r["dy"])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        cpu_offs = list(ex.map(one, range(ncpu)))
    tc = time.perf_counter() - t0
    res["cpu_baseline"] = {"value": round(ncpu / tc, 2), "unit": "tiles/s", "cores": cores, "kind": "port",
                           "sample": "%d of the same tiles, CPU oracle findOffset (C, float32), one tile per thread, %.2f s" % (ncpu, tc),
                           "offsets_equal_gpu": bool(all(tuple(int(v) for v in got[k, :2]) == cpu_offs[k] for k in range(ncpu)))}
    oracle.set_num_threads(1)
    t0 = time.perf_counter()
    for k in range(3):
        oracle.find_offset(left_h, right, tuple(rects[k]), tuple(rects[k]), tile, tile, precision=0)
    ts = time.perf_counter() - t0
    oracle.set_num_threads(cores)
    res["cpu_reference_shaped"] = {"value": round(3 / ts, 3), "unit": "tiles/s", "cores": 1, "kind": "port",
                                   "sample": "3 tiles one after another, buffers and set-up inside every call (the structure of image3.cpp:357-372), %.2f s" % ts}
    del tl, tr
    # config[0]: the coarse-to-fine recursion on a 2048 x 2048 pair with a known horizontal shift
    l2, r2, _ = synth.config_pair(1)
    ctx.set_stream(None)
    ctx.upload_pair(l2, r2)
    ctx.disparity_map(min_size=32)
    t0 = time.perf_counter()
    leaves, calls = ctx.disparity_map(min_size=32)
    tg = time.perf_counter() - t0
    t0 = time.perf_counter()
    oleaves, ocalls = oracle.disparity_map(l2, r2, min_size=32)
    to = time.perf_counter() - t0
    res["recursion_2048"] = {"workload": "BASELINE config[0] pair (2048x2048, shift 17): calcDisparityMap, min size 32, host rasters in, leaves out",
                             "findoffset_calls": calls, "leaves": len(leaves), "ms": round(tg * 1e3, 2),
                             "cpu_oracle_s": round(to, 2), "cpu_cores_blur_only": cores,
                             "leaves_equal_oracle": bool(calls == ocalls and [l[:10] for l in leaves] == [l[:10] for l in oleaves])}
    # a frame whose sides are no powers of two (8000 x 6000): the first level correlates the whole rasters (mixed-radix
    # transforms, segmented Gaussian passes), the levels below it take power-of-two tiles; the library call alone
    import ctypes as C
    import mistereo
    H3, W3 = 6000, 8000
    l3 = synth.texture(H3, W3, 7)
    r3 = synth.shift_right_image(l3, synth.planar_disparity(H3, W3, -20, 20, 16, 8), fill_seed=9)
    ctx.upload_pair(l3, r3)
    buf = (mistereo.Leaf * 65536)()
    n_, nt_ = C.c_int(0), C.c_int(0)
    ctx._lib.mis_disparity_map(ctx._h, W3, H3, 32, buf, len(buf), C.byref(n_), C.byref(nt_))
    t0 = time.perf_counter()
    rc3 = ctx._lib.mis_disparity_map(ctx._h, W3, H3, 32, buf, len(buf), C.byref(n_), C.byref(nt_))
    t3 = time.perf_counter() - t0
    res["recursion_8000x6000"] = {"workload": "8000x6000 8-bit synthetic pair, disparity plane -20..20: calcDisparityMap, min size 32, pair resident, leaves out",
                                  "findoffset_calls": nt_.value, "leaves": n_.value, "ms": round(t3 * 1e3, 2), "rc": rc3,
                                  "note": "GPU only; parity of general-size frames is covered by the GPU tests and tools/stress_driver.py"}
    ctx.set_stream(stream.cuda_stream)
    return res


def bench_ncc16(ctx, stream, dev, steps=3, cpu=True):
    """BASELINE config[4]'s kernel work on a band: 16384 x 1024, 16-bit, 31x31 window, +-512 search
    (1025 disparities), subpixel, every matched pixel triangulated."""
    import numpy as np
    import torch
    import mistereo
    import oracle
    import synth
    W, H, win, D = 16384, 1024, 31, 512
    nd = 2 * D + 1
    left = synth.texture(H, W, 0xC0FFEE + 5, np.uint16)
    rng = np.random.default_rng(5)
    left = np.clip(left.astype(np.int64) + rng.integers(-120, 121, left.shape), 0, 65535).astype(np.uint16)
    right = synth.shift_right_image_subpixel(left, 37.25)
    tl = torch.from_numpy(left.view(np.int16)).to(dev).view(torch.uint16)
    tr = torch.from_numpy(right.view(np.int16)).to(dev).view(torch.uint16)
    disp = torch.empty((H, W), dtype=torch.float32, device=dev)
    peak = torch.empty_like(disp)
    idisp = torch.empty((H, W), dtype=torch.int16, device=dev)
    xyz = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    A = [[10000.0, 0.0, W / 2.0], [0.0, 10000.0, H / 2.0], [0.0, 0.0, 1.0]]
    # second principal point offset by f B / H = 3000 px (the offset between the two crops): the 37.25 px shift is relief
    A2 = [[10000.0, 0.0, W / 2.0 + 3000.0], [0.0, 10000.0, H / 2.0], [0.0, 0.0, 1.0]]
    cam1 = mistereo.make_camera(A, np.eye(3), [0.0, 0.0, -3000.0])
    cam2 = mistereo.make_camera(A2, np.eye(3), [-900.0, 0.0, -3000.0])
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        ctx.bind_pair_torch(tl, tr)
        ctx.ncc_dense_torch(win, -D, D, disp, peak, idisp)
        ctx.triangulate_disparity_torch(cam1, cam2, disp, peak, 0.75, xyz, count=False)
        stream.synchronize()
        ctx.profile(True)
        ctx.profile_read()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(stream)
        for _ in range(steps):
            ctx.bind_pair_torch(tl, tr)                     # every step rebuilds the tables
            ctx.ncc_dense_torch(win, -D, D, disp, peak)
        e[1].record(stream)
        for _ in range(steps):
            ctx.triangulate_disparity_torch(cam1, cam2, disp, peak, 0.75, xyz, count=False)
        e[2].record(stream)
        for _ in range(steps):                              # the fractional column x - disp instead of the rounded one
            ctx.triangulate_disparity_torch(cam1, cam2, disp, peak, 0.75, xyz, count=False, subpixel=True)
        e3 = torch.cuda.Event(enable_timing=True)
        e3.record(stream)
        matched = int(ctx.triangulate_disparity_torch(cam1, cam2, disp, peak, 0.75, xyz))
    stream.synchronize()
    tables_ms, main_ms, refine_ms, calls = ctx.profile_read()
    ctx.profile(False)
    ms, tri_ms = e[0].elapsed_time(e[1]) / steps, e[1].elapsed_time(e[2]) / steps
    tri_sub_ms = e[2].elapsed_time(e3) / steps
    kernel_ms = main_ms / max(calls, 1)
    alg = ALG_BYTES_PER_PX_U16 * H * W
    res = {
        "workload": "BASELINE config[4] band: %dx%d 16-bit, %dx%d window, +-%d px (%d disparities), subpixel + XYZ" % (W, H, win, win, D, nd),
        "ms_per_step": round(ms, 3), "gpx_disp_per_s": round(H * W * nd / ms / 1e6, 2),
        "triangulation_ms": round(tri_ms, 3), "triangulation_subpixel_columns_ms": round(tri_sub_ms, 3), "matched_points": matched,
        "matched_points_per_s": round(matched / (ms + tri_ms) * 1e3, 1),
        "roofline": {"bound": "hbm", "kernel": "ncc16_argmax_kernel", "kernel_ms": round(kernel_ms, 3),
                     "tables_ms": round(tables_ms / max(calls, 1), 3), "refine_ms": round(refine_ms / max(calls, 1), 3),
                     "algorithmic_bytes": alg, "achieved": round(alg / kernel_ms / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(alg / kernel_ms / 1e6 / HBM_PEAK_GBS, 5), "kernel_gpx_disp_per_s": round(H * W * nd / kernel_ms / 1e6, 2),
                     "traffic": load_traffic_note("ncc16_argmax_kernel")},
    }
    # Vector-issue floor of the 16-bit kernel: the wave64 instructions it executes (SQ_INSTS_VALU of a committed counter
    # pass on this workload, profiles/) at 1.8 ns of SIMD time each, what the 64-bit integer, binary64 and dot
    # instructions that make up its loop cost (tools/ubench/valu_rates64, profiles/), over 1024 SIMDs.
    insts = load_traffic_note("ncc16_argmax_kernel", "insts_valu")
    if insts and (W, H, win, D) == (16384, 1024, 31, 512):
        floor_ms = insts * 1.8e-9 / 1024 * 1e3
        res["roofline"]["issue_floor"] = {"ms": round(floor_ms, 3), "frac": round(floor_ms / kernel_ms, 3), "insts_valu": insts,
                                          "note": "SQ_INSTS_VALU (committed counter pass, not measured in this run) x 1.8 ns / 1024 SIMDs / measured kernel time"}
    if not cpu:
        return res
    r = win // 2
    rows = min(32 * oracle.num_threads(), H - 2 * r)     # the oracle's threads each take 32-row strips
    cl, cr = np.ascontiguousarray(left[: rows + 2 * r]), np.ascontiguousarray(right[: rows + 2 * r])
    t0 = time.perf_counter()
    ora = oracle.ncc_dense(cl, cr, win, -D, D)
    tc = time.perf_counter() - t0
    # GPU rows of the crop in the crop's own coordinates (its first and last r rows are border rows)
    full = {"disp": np.full((rows + 2 * r, W), -32768, np.float32), "peak": np.zeros((rows + 2 * r, W), np.float32),
            "idisp": np.full((rows + 2 * r, W), -32768, np.int16)}
    for k, t in (("disp", disp), ("peak", peak), ("idisp", idisp)):
        full[k][r: r + rows] = t[r: r + rows].cpu().numpy()
    rep = oracle.parity_report(full["disp"], full["peak"], ora, cl, cr, win, -D, D, gpu_idisp=full["idisp"])
    res["cpu_baseline"] = {"value": round((rows + 2 * r) * W * nd / tc / 1e9, 4), "unit": "Gpx*disp/s", "cores": oracle.num_threads(),
                           "kind": "port", "sample": "first %d rows of the same band, CPU oracle (C, OpenMP, int64 sums), %.1f s" % (rows + 2 * r, tc),
                           "parity": rep}
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become the launcher (before torch is imported or any GPU call is made)
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher's WORLD_SIZE is %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if args.rehearse:
        sys.exit(rehearse(args, rank, world))
    import numpy as np
    import torch
    import torch.distributed as dist

    import mistereo
    import partition

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    H = W = args.size
    D = args.disp
    nd = 2 * D + 1
    strips_mode = args.partition == "strips" and world > 1
    # frames: a different pair per rank; strips: the same pair on every rank (each keeps its strip only)
    seed = 0xC0FFEE + 2 + (0 if strips_mode else 7919 * rank)
    left_h, right_h = synth_pair(H, W, D, seed)

    ctx = mistereo.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    do_gather = world > 1 and args.gather == "on"
    nbuf = 2

    def lib_waits_for_torch():
        stream.wait_stream(torch.cuda.current_stream())

    def torch_waits_for_lib():
        ev = torch.cuda.Event()
        ev.record(stream)
        torch.cuda.current_stream().wait_event(ev)

    if strips_mode:
        r0, n = partition.row_strips(H, world)[rank]
        lo, nin = partition.strip_input_rows(r0, n, H, args.window)
        left = torch.from_numpy(np.ascontiguousarray(left_h[lo:lo + nin])).to(dev)      # the strip and its halo rows only
        right = torch.from_numpy(np.ascontiguousarray(right_h[lo:lo + nin])).to(dev)
        if rank != 0:
            del left_h, right_h

        def compute(disp_t, peak_t, local_row0, nrows):
            with torch.cuda.stream(stream):
                ctx.bind_pair_torch(left, right)             # re-binding drops the per-pair tables: every step redoes all the work
                ctx.ncc_dense_torch(args.window, -D, D, disp_t, peak_t, row0=local_row0, nrows=nrows)

        job = StripJob(rank, world, H, W, args.window, dev, compute, lib_waits_for_torch, torch_waits_for_lib,
                       nbuf=nbuf, gather=do_gather)
        step = job.step

        def drain():
            job.drain()
            stream.synchronize()
            torch.cuda.synchronize()

        def last_result(steps):
            return job.result(max(steps - 1, 0))
    else:
        left = torch.from_numpy(left_h).to(dev)
        right = torch.from_numpy(right_h).to(dev)
        disp = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        peak = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        packed = [torch.empty((2, H, W), dtype=torch.float32, device=dev) for _ in range(nbuf)] if do_gather else None
        pipe = GatherPipeline(rank, world, packed) if do_gather else None

        def step(i):
            b = i % nbuf
            if pipe:
                pipe.reuse(b)              # the buffer's previous gather has drained (a wait on torch's current stream) ...
                lib_waits_for_torch()      # ... and the library's stream must honour it too
            with torch.cuda.stream(stream):
                # re-binding invalidates the context's per-pair tables: every step redoes all the work
                ctx.bind_pair_torch(left, right)
                if do_gather:
                    ctx.ncc_dense_torch(args.window, -D, D, packed[b][0], packed[b][1])
                else:
                    ctx.ncc_dense_torch(args.window, -D, D, disp[b], peak[b])
            if do_gather:
                torch_waits_for_lib()
                pipe.submit(b)

        def drain():
            if pipe:
                pipe.drain()
            stream.synchronize()
            torch.cuda.synchronize()

        def last_result(steps):
            b = max(steps - 1, 0) % nbuf
            return (packed[b][0], packed[b][1]) if do_gather else (disp[b], peak[b])

    for i in range(args.warmup):
        step(i)
    drain()
    ctx.profile(True)
    ctx.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pre_ms, main_ms, refine_ms, calls = ctx.profile_read()
    ctx.profile(False)

    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    result = None
    if rank == 0:
        dp, pk = last_result(args.steps)
        # every matched pixel (peak >= 0.75, the reference's accept threshold image3.cpp:751) triangulated to XYZ
        # (calculateXYZ, image3.cpp:148-264) with two synthetic nadir cameras (f = 10000 px, height 3000 m,
        # baseline 900 m, R = I), timed on its own, outside the timed region.  The second camera's principal point
        # carries the f B / H = 3000 px between the origins of the two crops, so that the pair's disparities of
        # +-200 px are +-200 m of relief around z = 0 (a crop pair whose disparities straddle zero has such an offset).
        cam1, cam2 = bench_cameras(W, H)
        cam2_same_pp = mistereo.make_camera([[10000.0, 0.0, W / 2.0], [0.0, 10000.0, H / 2.0], [0.0, 0.0, 1.0]], np.eye(3),
                                            [-900.0, 0.0, -3000.0])
        xyz = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        n_dev = torch.zeros((1,), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            # the raster size is passed in: in the strips partition rank 0's bound raster is its strip only
            ctx.triangulate_disparity_noalloc_torch(cam1, cam2, dp, pk, 0.75, xyz, n_dev)      # warm-up
            stream.synchronize()
            # the same with one principal point for both cameras: the rays are then nearly parallel and the
            # reference's search (image3.cpp:220-246) runs hundreds of steps per point
            t_par = time.perf_counter()
            ctx.triangulate_disparity_noalloc_torch(cam1, cam2_same_pp, dp, pk, 0.75, xyz, n_dev)
            stream.synchronize()
            t_par = time.perf_counter() - t_par
            t_tri = time.perf_counter()
            ctx.triangulate_disparity_noalloc_torch(cam1, cam2, dp, pk, 0.75, xyz, n_dev)
            stream.synchronize()
            t_tri = time.perf_counter() - t_tri
        matched = int(n_dev.item())
        del xyz

        frames_per_step = 1 if strips_mode else world
        work = float(H) * W * nd * frames_per_step          # px*disp per step, all ranks
        value = work * args.steps / dt / 1e9
        # the dominant kernel's launch covers this rank's rows
        rows_rank0 = partition.row_strips(H, world)[0][1] if strips_mode else H
        kernel_s = (main_ms / max(calls, 1)) / 1e3
        alg_bytes = ALG_BYTES_PER_PX_U8 * rows_rank0 * W
        achieved = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        kname = "ncc_mfma_argmax_kernel" if os.environ.get("MISTEREO_NCC_MFMA", "1") != "0" else "ncc_argmax_kernel"
        wl = ("BASELINE config[1]" if (W, args.window, D) == (8192, 21, 256) else
              "BASELINE config[3] frame size (one of its 16 pairs)" if (W, args.window, D) == (16384, 21, 256) else "non-baseline size")
        result = {
            "metric": "NCC throughput", "value": round(value, 3), "unit": "Gpx*disp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strips_mode else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d 8-bit synthetic aerial pair, %dx%d NCC window, +-%d px search (%d disparities)%s"
                                   % (wl, W, H, args.window, args.window, D, nd,
                                      ", ONE pair sharded by row strips with halo" if strips_mode else " per GPU"),
                       "window": args.window, "dmin": -D, "dmax": D,
                       "partition": "strips" if strips_mode else "frames",
                       "gather_to_rank0": bool(do_gather)},
            "matched_points_per_s": round(matched * (1 if strips_mode else world) / (dt / max(args.steps, 1) + t_tri), 1),
            "matched_points": {"per_frame": matched, "peak_threshold": 0.75, "triangulation_ms": round(t_tri * 1e3, 3),
                               "triangulation_ms_nearly_parallel_rays": round(t_par * 1e3, 3),
                               "note": "correspondences with peak >= 0.75 triangulated to XYZ; rate = points / (correlation step + triangulation); "
                                       "cameras: f 10000 px, height 3000 m, baseline 900 m, the second principal point offset by f B / H so that "
                                       "the disparities are relief around z = 0; nearly_parallel_rays: the same with one principal point for both"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic_note(kname),
                         "traffic_fetch_doubled": (2 * (load_traffic_note(kname, "fetch_bytes_raw") or 0) + (load_traffic_note(kname, "write_bytes") or 0)) or None,
                         "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/ (not measured in this run); "
                                           "traffic = FETCH_SIZE + WRITE_SIZE as reported, traffic_fetch_doubled = 2 x FETCH_SIZE + WRITE_SIZE "
                                           "(on gfx950 FETCH_SIZE counts half the bytes of 16-byte-per-lane streaming reads, which is how these kernels stage their rows)",
                         "kernel": kname, "kernel_ms": round(kernel_s * 1e3, 4),
                         "tables_ms": round(pre_ms / max(calls, 1), 4),
                         "refine_ms": round(refine_ms / max(calls, 1), 4),
                         "algorithmic_bytes": alg_bytes,
                         "valu_busy_frac_pmc": load_traffic_note(kname, "valu_busy"),
                         "issue_floor": issue_floor(W, rows_rank0, args.window, nd, kernel_s) if not strips_mode else None,
                         "mfma": mfma_rate(W, rows_rank0, args.window, nd, kernel_s),
                         "note": "sliding-window correlation: compulsory HBM traffic is 10 B/px whatever the search "
                                 "range; the window sums slide on the matrix cores (two launches: interior and edge "
                                 "ranges of right columns, timed together), the kernel is bound by the vector "
                                 "instructions that turn them into scores and keys, not by HBM"},
        }
        if world == 1 and args.cpu_rows > 0:
            import oracle
            r = args.window // 2
            rows = min(args.cpu_rows, H - 2 * r)
            crop_l = np.ascontiguousarray(left_h[: rows + 2 * r])
            crop_r = np.ascontiguousarray(right_h[: rows + 2 * r])
            t1 = time.perf_counter()
            ora = oracle.ncc_dense(crop_l, crop_r, args.window, -D, D)
            tc = time.perf_counter() - t1
            cpu_work = float(rows + 2 * r) * W * nd
            # cross-check the GPU rows against the oracle while we have both: one more (untimed) call that also
            # writes the integer disparity
            idisp = torch.empty((H, W), dtype=torch.int16, device=dev)
            with torch.cuda.stream(stream):
                ctx.ncc_dense_torch(args.window, -D, D, dp, pk, idisp)
            stream.synchronize()
            hh = rows + 2 * r
            gd, gp, gi = dp[:hh].cpu().numpy(), pk[:hh].cpu().numpy(), idisp[:hh].cpu().numpy()
            if hh < H:      # the crop's last r rows are border rows of the crop only
                gd[-r:] = -32768.0; gp[-r:] = 0.0; gi[-r:] = -32768
            rep = oracle.parity_report(gd, gp, ora, crop_l, crop_r, args.window, -D, D, gpu_idisp=gi)
            result["cpu_baseline"] = {
                "value": round(cpu_work / tc / 1e9, 4), "unit": "Gpx*disp/s", "cores": oracle.num_threads(),
                "kind": "port",
                "sample": "first %d rows of the same pair (%.2f Gpx*disp), CPU oracle (C, OpenMP), %.1f s"
                          % (rows + 2 * r, cpu_work / 1e9, tc),
                "parity": rep,
            }
            del ora, gd, gp, gi, idisp
        if world == 1 and args.extras == "on" and args.second_peak == "on":
            # the same step with the second-peak output (best score further than one disparity from the winner)
            pk2 = torch.empty((H, W), dtype=torch.float32, device=dev)
            with torch.cuda.stream(stream):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                for i in range(4):
                    if i == 1:
                        e[0].record(stream)
                    ctx.bind_pair_torch(left, right)
                    ctx.ncc_dense_torch(args.window, -D, D, dp, pk, None, pk2)
                e[1].record(stream)
            stream.synchronize()
            ms2 = e[0].elapsed_time(e[1]) / 3
            result["second_peak"] = {"ms_per_step": round(ms2, 3), "vs_plain_step": round(ms2 / (dt / max(args.steps, 1) * 1e3), 3),
                                     "gpx_disp_per_s": round(float(H) * W * nd / ms2 / 1e6, 1),
                                     "note": "same pair and search with the peak2 raster requested; parity of peak2 is covered by the GPU tests"}
            del pk2
        if world == 1 and args.extras == "on" and args.second_peak == "on" and args.window <= 21:
            # the same pair and search under a 31 x 31 window: from 23 x 23 on only the window sums slide on the matrix
            # cores and the mean correction's integer part is one more vector instruction per score
            d31, p31 = torch.empty_like(dp), torch.empty_like(pk)      # (dp / pk stay the plain step's result)
            with torch.cuda.stream(stream):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
                for i in range(4):
                    if i == 1:
                        e[0].record(stream)
                    ctx.bind_pair_torch(left, right)
                    ctx.ncc_dense_torch(31, -D, D, d31, p31)
                e[1].record(stream)
            stream.synchronize()
            del d31, p31
            ms31 = e[0].elapsed_time(e[1]) / 3
            result["window_31"] = {"ms_per_step": round(ms31, 3), "vs_plain_step": round(ms31 / (dt / max(args.steps, 1) * 1e3), 3),
                                   "gpx_disp_per_s": round(float(H) * W * nd / ms31 / 1e6, 1),
                                   "note": "same pair and search, 31x31 window (matrix-core kernels; parity covered by the GPU tests)"}
        if world == 1 and args.extras == "on":
            result["path_f"] = bench_path_f(ctx, stream, dev, left_h, cpu=args.cpu_rows > 0)
            result["ncc16"] = bench_ncc16(ctx, stream, dev, cpu=args.cpu_rows > 0)
    # The other partition on the same ranks (outside the timed region above).  It needs every rank; a rank that
    # failed or hangs in it must not cost the line above: a watchdog prints the line without this leg and ends the
    # process when the leg (or the final barrier) does not come back.
    import threading
    lock = threading.Lock()
    printed = [False]
    headline = json.dumps(result) if rank == 0 else None      # the line as it stands before the leg: what the watchdog prints

    def emit(line):
        with lock:
            if rank == 0 and not printed[0]:
                printed[0] = True
                print(line, flush=True)

    def give_up():
        # the leg or the final barrier hangs: the measured line is still printed (without the leg), but the run FAILS
        if rank == 0:
            late = json.loads(headline)
            late["strips"] = {"error": "did not finish within %d s" % STRIPS_LEG_LIMIT_S}
            emit(json.dumps(late))
        print("bench.py: rank %d gave up waiting for the strips leg or the final barrier after %d s" % (rank, STRIPS_LEG_LIMIT_S),
              file=sys.stderr, flush=True)
        os._exit(3)

    want_leg = (not strips_mode) and args.gpus == world and (
        (world > 1 and args.strips_leg == "on" and args.gather == "on") or args.strips_leg == "force")
    dog = None
    if want_leg or world > 1:
        dog = threading.Timer(STRIPS_LEG_LIMIT_S, give_up)
        dog.daemon = True
        dog.start()
    if want_leg:
        try:
            leg = strips_leg(args, rank, world, dev, ctx, stream, H, W, D, 0xC0FFEE + 2,
                             last_result(args.steps) if rank == 0 else None, matched if rank == 0 else None)
        except Exception as e:      # reported, never fatal for the measured line
            leg = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            result["strips"] = leg
    emit(json.dumps(result) if rank == 0 else None)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if dog:
        dog.cancel()
    ctx.close()


if __name__ == "__main__":
    main()
