"""Splitting one frame over the GPUs of a node: tile rows per rank, bands gathered to rank 0.

The reference hands 8x8 tiles to CPU workers from one shared counter (threads.cpp:158-169); every
pixel is written once and tiles do not depend on each other.  Across GPUs the same independence
is used at a coarser grain: the scene is replicated, rank r renders a contiguous range of
8-pixel tile rows (``band_rows``), and the finished bands are sent to rank 0 with one grouped
send/recv (``gather_bands``; RCCL over xGMI with the "nccl" backend, gloo on CPU for tests).
No other collective is needed.
"""


def band_rows(height, world, rank):
    """Rows [y0, y1) of rank's band: contiguous tile rows, the first (rows % world) ranks get one more."""
    if height % 8:
        raise ValueError("height must be a multiple of 8")
    rows = height // 8
    base, rem = divmod(rows, world)
    start = rank * base + min(rank, rem)
    return start * 8, (start + base + (1 if rank < rem else 0)) * 8


def gather_bands(fb, world, rank, dist):
    """Rank 0 receives every other rank's band straight into its own frame tensor ``fb`` (H x W);
    the other ranks send the band they rendered.  One batch of point-to-point operations."""
    if world <= 1:
        return
    height = fb.shape[0]
    ops = []
    if rank == 0:
        for r in range(1, world):
            a, b = band_rows(height, world, r)
            if b > a:
                ops.append(dist.P2POp(dist.irecv, fb[a:b], r))
    else:
        a, b = band_rows(height, world, rank)
        if b > a:
            ops.append(dist.P2POp(dist.isend, fb[a:b], 0))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
