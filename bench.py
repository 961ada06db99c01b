#!/usr/bin/env python3
"""Dense-NCC throughput benchmark (BASELINE.json metric) for libmistereo on MI355X.

One step = one full pass of the dense windowed-NCC hot path over one synthetic stereo pair that
is already resident in HBM: frame-border fill, window-statistics tables, main correlation kernel
(argmax + subpixel).  Workload at N=1: BASELINE config[1] -- 8192x8192 8-bit pair, 21x21 window,
+-256 px search.  With N>1 every rank processes a pair of its own (weak scaling: frames are
independent units) and the disparity/peak rasters are gathered to rank 0 over RCCL while the next
step computes.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle")]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
ALG_BYTES_PER_PX_U8 = 10.0     # left + right pixel read once, disparity f32 + peak f32 written


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=8192, help="frame is size x size pixels")
    ap.add_argument("--window", type=int, default=21)
    ap.add_argument("--disp", type=int, default=256, help="search is +-disp pixels")
    ap.add_argument("--gather", choices=["on", "off"], default="on",
                    help="with N>1: gather disparity/peak rasters to rank 0 (overlapped)")
    ap.add_argument("--cpu-rows", type=int, default=8192,
                    help="rows of the frame the CPU oracle is timed on (0: skip the CPU baseline)")
    return ap.parse_args()


def load_traffic_note(kernel_name, field="bytes"):
    """HBM bytes per launch (or another field) of the dominant kernel from a committed rocprofv3 --pmc run."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            entry = json.load(f).get(kernel_name)
        return entry.get(field) if isinstance(entry, dict) else (entry if field == "bytes" else None)
    except Exception:
        return None


def issue_floor(W, H, window, nd, kernel_s):
    """Time the argmax kernel's own instruction mix needs at full vector-issue rate, for the 21x21
    window: 509 ns of SIMD time per strip row (32 columns x 64 disparities; per-instruction prices
    measured with tools/ubench/valu_rates on this part: dot2c 94, DPP maxima 93, sliding sums 89,
    packed fmas 86, keys 58, rest 90), 1024 SIMDs, 256- or 128-row bands with 2r warm-up rows."""
    if window != 21:
        return None
    r = window // 2
    strips = -(-(W - 2 * r) // 32)
    rows = H - 2 * r
    chunks = nd // 64
    band = 256 if strips * -(-rows // 256) * max(chunks, 1) >= 6 * 5120 else 128      # the library's rule
    bands = -(-rows // band)
    strip_rows = strips * chunks * (rows + bands * 2 * r * 0.37)      # warm-up rows run the column sums only (37 % of a row)
    floor_s = strip_rows * 509e-9 / 1024
    return {"ms": round(floor_s * 1e3, 3), "frac": round(floor_s / kernel_s, 3) if kernel_s > 0 else None,
            "note": "vector-issue floor of the kernel's instruction mix / measured kernel time"}


class GatherPipeline:
    """Double-buffered gather of per-rank result rasters to rank 0, overlapped with the next step.

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


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import mistereo
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
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
    # synthetic pair of BASELINE config[1]'s kind, a different seed per rank
    seed = 0xC0FFEE + 2 + 7919 * rank
    left_h = synth.texture(H, W, seed)
    dmap = synth.planar_disparity(H, W, -min(200, D - 56) if D > 60 else -D // 2, min(200, D - 56) if D > 60 else D // 2,
                                  64, seed + 1)
    right_h = synth.shift_right_image(left_h, dmap, fill_seed=seed + 1000)
    noise = (synth.splitmix64(seed + 2, H * W) % np.uint64(5)).astype(np.int16).reshape(H, W) - 2
    right_h = np.clip(right_h.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    del noise, dmap

    left = torch.from_numpy(left_h).to(dev)
    right = torch.from_numpy(right_h).to(dev)
    nbuf = 2
    disp = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    peak = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    do_gather = world > 1 and args.gather == "on"
    packed = [torch.empty((2, H, W), dtype=torch.float32, device=dev) for _ in range(nbuf)] if do_gather else None
    pipe = GatherPipeline(rank, world, packed) if do_gather else None

    ctx = mistereo.Context(local_rank)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    def step(i):
        b = i % nbuf
        if pipe:
            pipe.reuse(b)              # the buffer's previous gather has drained (a wait on torch's current stream) ...
            stream.wait_stream(torch.cuda.current_stream())   # ... and the library's stream must honour it too
        with torch.cuda.stream(stream):
            # re-binding invalidates the context's per-pair tables: every step redoes all the work
            ctx.bind_pair_torch(left, right)
            if do_gather:
                ctx.ncc_dense_torch(args.window, -D, D, packed[b][0], packed[b][1])
            else:
                ctx.ncc_dense_torch(args.window, -D, D, disp[b], peak[b])
        if do_gather:
            ev = torch.cuda.Event()
            ev.record(stream)
            torch.cuda.current_stream().wait_event(ev)
            pipe.submit(b)

    def drain():
        if pipe:
            pipe.drain()
        stream.synchronize()
        torch.cuda.synchronize()

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

    # matched points (peak >= 0.75, the reference's accept threshold image3.cpp:751), outside the timed region
    last = (args.steps - 1) % nbuf if args.steps > 0 else 0
    pk = packed[last][1] if do_gather else peak[last]
    dp = packed[last][0] if do_gather else disp[last]
    # every matched pixel triangulated to XYZ (calculateXYZ, image3.cpp:148-264) with two synthetic nadir
    # cameras (f = 10000 px, height 3000 m, baseline 900 m, R = I), timed on its own
    A = [[10000.0, 0.0, W / 2.0], [0.0, 10000.0, H / 2.0], [0.0, 0.0, 1.0]]
    cam1 = mistereo.make_camera(A, np.eye(3), [0.0, 0.0, -3000.0])
    cam2 = mistereo.make_camera(A, np.eye(3), [-900.0, 0.0, -3000.0])
    xyz = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    with torch.cuda.stream(stream):
        ctx.triangulate_disparity_torch(cam1, cam2, dp, pk, 0.75, xyz)      # warm-up
        stream.synchronize()
        t_tri = time.perf_counter()
        matched = int(ctx.triangulate_disparity_torch(cam1, cam2, dp, pk, 0.75, xyz))   # reading the count synchronises
        t_tri = time.perf_counter() - t_tri
    del xyz

    result = None
    if rank == 0:
        work = float(H) * W * nd * world            # px*disp per step, all ranks
        value = work * args.steps / dt / 1e9
        kernel_s = (main_ms / max(calls, 1)) / 1e3
        alg_bytes = ALG_BYTES_PER_PX_U8 * H * W
        achieved = alg_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        kname = "ncc_argmax_kernel"
        result = {
            "metric": "NCC throughput", "value": round(value, 3), "unit": "Gpx*disp/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d 8-bit synthetic aerial pair, %dx%d NCC window, +-%d px search (%d disparities) per GPU"
                                   % ("BASELINE config[1]" if (W, args.window, D) == (8192, 21, 256) else
                                      "BASELINE config[3] frame size (one of its 16 pairs)" if (W, args.window, D) == (16384, 21, 256) else
                                      "non-baseline size", W, H, args.window, args.window, D, nd),
                       "window": args.window, "dmin": -D, "dmax": D,
                       "gather_to_rank0": bool(do_gather)},
            "matched_points_per_s": round(matched * world / (dt / max(args.steps, 1) + t_tri), 1),
            "matched_points": {"per_frame": matched, "peak_threshold": 0.75, "triangulation_ms": round(t_tri * 1e3, 3),
                               "note": "correspondences with peak >= 0.75 triangulated to XYZ; rate = points / (correlation step + triangulation)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": load_traffic_note(kname),
                         "kernel": kname, "kernel_ms": round(kernel_s * 1e3, 4),
                         "tables_ms": round(pre_ms / max(calls, 1), 4),
                         "refine_ms": round(refine_ms / max(calls, 1), 4),
                         "algorithmic_bytes": alg_bytes,
                         "valu_busy_frac_pmc": load_traffic_note(kname, "valu_busy"),
                         "issue_floor": issue_floor(W, H, args.window, nd, kernel_s),
                         "note": "sliding-window correlation: compulsory HBM traffic is 10 B/px whatever the search "
                                 "range, the kernel is bound by vector ALU / LDS issue, not by HBM"},
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
            # cross-check the GPU rows against the oracle while we have both
            gi = disp[last][r: r + rows].round().to(torch.int32).cpu().numpy() if not do_gather else None
            agree = None
            if gi is not None:
                oi = ora["idisp"][r: r + rows].astype(np.int32)
                valid = oi != -32768
                gd = disp[last][r: r + rows].cpu().numpy()
                agree = float((np.abs(gd - ora["disp"][r: r + rows])[valid] <= 0.01).mean())
            result["cpu_baseline"] = {
                "value": round(cpu_work / tc / 1e9, 4), "unit": "Gpx*disp/s", "cores": oracle.num_threads(),
                "kind": "port",
                "sample": "first %d rows of the same pair (%.2f Gpx*disp), CPU oracle (C, OpenMP), %.1f s"
                          % (rows + 2 * r, cpu_work / 1e9, tc),
                "gpu_vs_oracle_disp_within_0.01px": agree,
            }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
