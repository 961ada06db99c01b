"""Work partition for several GPUs (one process per GPU, torch.distributed over RCCL).

The reference is a single process (its only concurrency is FFTW's thread pool,
image3.cpp:974-975); tiles and rows are independent, so the only exchange is a gather of result
rasters to rank 0.  Two partitions are used:
  * frames over ranks (batch of pairs: every rank owns whole frames), and
  * row strips of one frame: rank r computes output rows [row0, row0+nrows) with
    mis_ncc_dense_rows_device; the correlation window reaches r = window//2 rows beyond the strip.
    A rank either binds the whole frame, or only its strip plus those halo rows as a raster of its
    own (strip_input_rows): windows never leave that raster, so the rows come out bit for bit as in
    the full frame and no halo is exchanged between ranks.
Gather: every rank's rows go to rank 0 as point-to-point transfers posted together (one send per
rank and raster, received straight into the rows of the full raster): on a fully connected xGMI
node each peer has its own link into rank 0, so the transfers run side by side; there is no ring
and no padding to a common strip height.
"""


def row_strips(height, world):
    """Contiguous output-row ownership [(row0, nrows)] for ranks 0..world-1 (remainder rows go to
    the first ranks; empty strips are allowed when world > height)."""
    base, rem = divmod(height, world)
    out, row = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((row, n))
        row += n
    return out


def strip_input_rows(row0, nrows, height, window):
    """Rows of the frame a strip's computation reads: its own rows plus the window halo."""
    r = window // 2
    if nrows == 0:
        return (row0, 0)
    lo, hi = max(0, row0 - r), min(height, row0 + nrows + r)
    return (lo, hi - lo)


def frames_for_rank(nframes, rank, world):
    """Frame indices owned by `rank` (round robin)."""
    return list(range(rank, nframes, world))


def gather_rows(local_rows, strips, rank, world, dst=0, group=None):
    """Gather the per-rank row strips of a raster to rank `dst` and stack them in row order.
    local_rows: tensor [nrows_r, W, ...] holding this rank's output rows.  Returns the full raster
    on `dst`, None elsewhere.  Strips may differ in height: they are padded to the tallest one for
    the collective and cropped afterwards."""
    import torch
    import torch.distributed as dist

    hmax = max(n for _, n in strips)
    pad_shape = (hmax,) + tuple(local_rows.shape[1:])
    buf = torch.zeros(pad_shape, dtype=local_rows.dtype, device=local_rows.device)
    buf[: local_rows.shape[0]] = local_rows
    if world == 1:
        return buf[: strips[0][1]].clone()
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([outs[r][: strips[r][1]] for r in range(world)], dim=0)


class StripGather:
    """Gather of row strips into full rasters on rank `dst`, as grouped point-to-point transfers.

    full:   on `dst`, a list of tensors [H, W, ...] (one per raster, e.g. disparity and peak); rank
            `dst` writes its own rows into them itself.
    submit(local) on the other ranks sends local[k] (this rank's rows of raster k, contiguous); on `dst`
    it posts the matching receives into full[k][row0:row0+nrows].  wait() blocks until the transfers
    of the last submit have completed (on the device: the current stream is ordered behind them).
    Works with RCCL ("nccl" backend: the sends and receives of one submit are one group) and gloo."""

    def __init__(self, strips, rank, world, full=None, dst=0, group=None):
        self.strips, self.rank, self.world, self.dst, self.group = strips, rank, world, dst, group
        self.full = full
        self.pending = []

    def submit(self, local=None):
        import torch.distributed as dist
        if self.world == 1:
            return
        ops = []
        if self.rank == self.dst:
            for src in range(self.world):
                r0, n = self.strips[src]
                if src == self.dst or n == 0:
                    continue
                for t in self.full:
                    ops.append(dist.P2POp(dist.irecv, t[r0:r0 + n], src, group=self.group))
        else:
            if self.strips[self.rank][1] > 0:
                for t in local:
                    ops.append(dist.P2POp(dist.isend, t, self.dst, group=self.group))
        if ops:
            self.pending = dist.batch_isend_irecv(ops)

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
