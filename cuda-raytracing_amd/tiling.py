"""Frame tiling across ranks: stripe bookkeeping and the per-frame gather.  The same code drives
bench.py on N GPUs (backend nccl = RCCL, device tensors) and the world_size-2 gloo test on CPU."""
import numpy as np


def stripe_rows(height, stripe, rank, world):
    """Rows rank `rank` owns: stripes s with s % world == rank (mirror of rt_stripe_rows)."""
    n = 0
    s = rank
    while s * stripe < height:
        n += min(stripe, height - s * stripe)
        s += world
    return n


def frame_rows_of(height, stripe, rank, world):
    """Frame row index of every local row of `rank`, in local order."""
    rows = []
    s = rank
    while s * stripe < height:
        rows.extend(range(s * stripe, min((s + 1) * stripe, height)))
        s += world
    return np.asarray(rows, np.int64)


def unstripe_host(gathered, height, stripe, world):
    """gathered[world, max_rows, pitch] (rank-major, padded) -> frame[height, pitch]; host mirror of rt_unstripe."""
    out = np.zeros((height, gathered.shape[2]), gathered.dtype)
    for r in range(world):
        fr = frame_rows_of(height, stripe, r, world)
        out[fr] = gathered[r, :len(fr)]
    return out


# ---- buffer layout when F frames travel in one gather --------------------------------------------------
# local buffer of a rank:   [F][max_rows][pitch]           (frame f starts f * max_rows rows in)
# gathered buffer on dst:   [world][F][max_rows][pitch]    (rank r starts r * F * max_rows rows in)

def batch_local_ptrs(local_base, F, max_rows, pitch):
    """Device pointers of the F per-frame stripe buffers inside one rank's local buffer."""
    return [local_base + f * max_rows * pitch for f in range(F)]


def batch_unstripe_args(gathered_base, f, F, max_rows, pitch):
    """(d_gathered, rank_stride) to pass to rt_unstripe for frame f of a gathered batch."""
    return gathered_base + f * max_rows * pitch, F * max_rows * pitch


def gather_stripes(local, gathered, rank, dst=0):
    """One frame's exchange step: every rank's padded local stripe buffer to rank `dst`.
    local: [max_rows, pitch] uint8 tensor; gathered: [world, max_rows, pitch] on dst, None elsewhere."""
    import torch.distributed as dist
    if rank == dst:
        dist.gather(local, list(gathered.unbind(0)), dst=dst)
    else:
        dist.gather(local, None, dst=dst)


class StripePipeline:
    """Double-buffered frame loop for N ranks: while group i's stripes are being gathered (on the
    collective's own stream) and un-striped (on `side_stream`), group i+1 is already rendering.
    render_fn(b) renders this rank's stripes into local buffer b; unstripe_fn(b) (rank `dst` only) turns gathered
    buffer b into frames -- it is called with `side_stream` current, so it must launch on torch's current stream.
    side_stream = None (CPU backends, tests): waits and un-stripes inline.
    Works with any torch.distributed backend (nccl on GPUs, gloo in the CPU tests)."""

    def __init__(self, rank, world, local, gathered, render_fn, unstripe_fn, dst=0, side_stream=None):
        self.rank, self.world, self.dst = rank, world, dst
        self.local, self.gathered = local, gathered
        self.render_fn, self.unstripe_fn = render_fn, unstripe_fn
        self.side = side_stream
        self.pending = [None, None]
        self.side_busy = [False, False]   # buffer b was handed to the side stream and not yet waited for
        self.frames_done = 0

    def _finish(self, b):
        """Order 'wait for gather b, then un-stripe it' -- on the side stream if there is one, so that the stream
        that renders never waits for a collective or spends time copying rows."""
        work = self.pending[b]
        if work is None:
            return
        if self.side is None:
            work.wait()
            if self.rank == self.dst:
                self.unstripe_fn(b)
        else:
            import torch
            with torch.cuda.stream(self.side):
                work.wait()               # stream-ordered for nccl: only the side stream waits for the gather
                if self.rank == self.dst:
                    self.unstripe_fn(b)
            self.side_busy[b] = True
        self.pending[b] = None
        self.frames_done += 1

    def release(self, b):
        """Call before buffers b are written again: everything that still reads them is ordered before what the
        current stream does next."""
        self._finish(b)
        if self.side is not None and self.side_busy[b]:
            import torch
            torch.cuda.current_stream().wait_stream(self.side)
            self.side_busy = [False, False]

    def step(self, i):
        import torch.distributed as dist
        b = i & 1
        self.release(b)                   # buffers b were last used by group i-2
        self.render_fn(b)
        if self.rank == self.dst:
            self.pending[b] = dist.gather(self.local[b], list(self.gathered[b].unbind(0)), dst=self.dst, async_op=True)
        else:
            self.pending[b] = dist.gather(self.local[b], None, dst=self.dst, async_op=True)
        self._finish(b ^ 1)               # group i-1: its gather overlapped this group's render

    def drain(self):
        self.release(0)
        self.release(1)
