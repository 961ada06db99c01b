#!/usr/bin/env python3
"""BASELINE config[2]: batched 512x512 FFT tile correlation over an 8192x8192 pair (256 tiles),
per-tile integer offsets drawn in [-100,100]^2.  Prints one JSON line (tiles/s, Gpx/s)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "image-processing_amd"), os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mistereo  # noqa: E402
import synth  # noqa: E402


def main():
    size, tile, steps = 8192, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = 0xC0FFEE + 3
    left = synth.texture(size, size, seed)
    nt = size // tile
    r = synth.splitmix64(seed + 1, 2 * nt * nt).reshape(nt * nt, 2)
    offs = (r % np.uint64(201)).astype(np.int64) - 100
    right = np.empty_like(left)
    pad = np.pad(left, 128, mode="reflect")
    rects1, rects2 = [], []
    k = 0
    for j in range(nt):
        for i in range(nt):
            dx, dy = int(offs[k, 0]), int(offs[k, 1]); k += 1
            # image-2 tile content = image-1 content displaced by (-dx,-dy): peak expected at (dx,dy)
            y0, x0 = j * tile + 128 + dy, i * tile + 128 + dx
            right[j * tile:(j + 1) * tile, i * tile:(i + 1) * tile] = pad[y0:y0 + tile, x0:x0 + tile]
            rects1.append((i * tile, j * tile, tile, tile)); rects2.append((i * tile, j * tile, tile, tile))
    ctx = mistereo.Context(0)
    ctx.upload_pair(left, right)
    got = ctx.find_offsets(rects1, rects2, tile, tile)          # warm-up + correctness
    hit = float(np.mean((got["dx"] == offs[:, 0]) & (got["dy"] == offs[:, 1])))
    t0 = time.perf_counter()
    for _ in range(steps):
        got = ctx.find_offsets(rects1, rects2, tile, tile)
    dt = (time.perf_counter() - t0) / steps
    n = len(rects1)
    # CPU side by side: the C oracle's findOffset restatement on a few of the same tiles
    import oracle
    ncpu = 6
    t1 = time.perf_counter()
    same = 0
    for k in range(ncpu):
        o = oracle.find_offset(left, right, rects1[k], rects2[k], tile, tile, precision=0)
        same += int((o["dx"], o["dy"]) == (int(got["dx"][k]), int(got["dy"][k])))
    tc = (time.perf_counter() - t1) / ncpu
    cpu = {"value": round(1.0 / tc, 2), "unit": "tiles/s", "cores": oracle.num_threads(), "kind": "port",
           "sample": "%d of the same tiles, %.0f ms each; offsets equal to the GPU's on %d" % (ncpu, tc * 1e3, same)}
    print(json.dumps({"metric": "FFT tile correlation", "value": round(n / dt, 1), "unit": "tiles/s",
                      "ms_per_step": round(dt * 1e3, 3), "tiles": n, "tile": tile,
                      "gpx_per_s": round(n * tile * tile / dt / 1e9, 3),
                      "offsets_recovered": hit,
                      "algorithmic_GBps": round(n * (2 * tile * tile + 16) / dt / 1e9, 2),
                      "cpu_baseline": cpu,
                      "config": {"workload": "BASELINE config[2]: 8192x8192 pair, 256 tiles of 512x512, offsets in [-100,100]^2"}}))


if __name__ == "__main__":
    main()
