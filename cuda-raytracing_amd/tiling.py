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


def frames_per_rank(F, world):
    """F frame slots dealt to `world` assembling ranks in contiguous blocks, as even as possible, earlier ranks first:
    (counts, offsets)."""
    counts = [F // world + (1 if d < F % world else 0) for d in range(world)]
    offsets = [sum(counts[:d]) for d in range(world)]
    return counts, offsets


def rotating_plan(count, world):
    """Plan of one rotating-root exchange of a group of `count` frames: (slots, counts, offsets, real).
    The exchange always moves slots = max(count, world) frame slots (slots past `count` carry stale rows), so that no
    rank ever sends or receives an empty message whatever the group size; rank d assembles slots
    [offsets[d], offsets[d] + counts[d]), of which the first real[d] hold frames of the group."""
    slots = max(count, world)
    counts, offsets = frames_per_rank(slots, world)
    real = [min(max(count - offsets[d], 0), counts[d]) for d in range(world)]
    return slots, counts, offsets, real


def exchange_to_root(local, gathered, rank, dst=0):
    """Start the gather of a group to one root: local [F * max_rows, pitch] from every rank into
    gathered [world, F * max_rows, pitch] on `dst` (None elsewhere).  Returns the async work handle."""
    import torch.distributed as dist
    if rank == dst:
        return dist.gather(local, list(gathered.unbind(0)), dst=dst, async_op=True)
    return dist.gather(local, None, dst=dst, async_op=True)


def exchange_rotating(local, received, count, world, max_rows):
    """Start the gather of a group with a rotating root, fused into ONE all-to-all: the stripes of frame f go to the
    rank that assembles f (rotating_plan), so every rank receives 1/world of the pixels and every xGMI link carries
    the same load in both directions instead of seven links converging on one root.
    local: [>= slots * max_rows, pitch] (frame-major, so the block for destination d is contiguous);
    received: [>= world * counts[me] * max_rows, pitch] (filled source-major).  Returns the async work handle."""
    import torch.distributed as dist
    slots, counts, _, _ = rotating_plan(count, world)
    me = dist.get_rank()
    return dist.all_to_all_single(received[:world * counts[me] * max_rows], local[:slots * max_rows],
                                  output_split_sizes=[counts[me] * max_rows] * world,
                                  input_split_sizes=[c * max_rows for c in counts], async_op=True)


class StripePipeline:
    """Double-buffered frame loop for N ranks: while group i's stripes are being exchanged (on the
    collective's own stream) and un-striped (on `side_stream`), group i+1 is already rendering.
    render_fn(b) renders this rank's stripes into local buffer b; exchange_fn(b) starts the collective on buffer b
    and returns its work handle; unstripe_fn(b) (only on ranks for which `assembles` is true) turns received
    buffer b into frames -- it is called with `side_stream` current, so it must launch on torch's current stream.
    side_stream = None (CPU backends, tests): waits and un-stripes inline.
    compute_streams = (s0, s1): buffer b's render launch and collective are issued with s_b current (render_fn(b) must
    launch on s_b), so consecutive render kernels sit on different streams and the tail of one overlaps the start of
    the next; None = everything on the caller's current stream.
    Works with any torch.distributed backend (nccl on GPUs, gloo in the CPU tests)."""

    def __init__(self, render_fn, exchange_fn, unstripe_fn, assembles=True, side_stream=None, compute_streams=None):
        self.render_fn, self.exchange_fn, self.unstripe_fn = render_fn, exchange_fn, unstripe_fn
        self.assembles = assembles
        self.side = side_stream
        self.compute = compute_streams
        self.pending = [None, None]
        self.side_busy = [False, False]   # buffer b was handed to the side stream and not yet waited for
        self.frames_done = 0

    def _finish(self, b):
        """Order 'wait for exchange b, then un-stripe it' -- on the side stream if there is one, so that the stream
        that renders never waits for a collective or spends time copying rows."""
        work = self.pending[b]
        if work is None:
            return
        if self.side is None:
            work.wait()
            if self.assembles:
                self.unstripe_fn(b)
        else:
            import torch
            with torch.cuda.stream(self.side):
                work.wait()               # stream-ordered for nccl: only the side stream waits for the collective
                if self.assembles:
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

    def _issue(self, b):
        self.release(b)                   # buffers b were last used by group i-2
        self.render_fn(b)
        self.pending[b] = self.exchange_fn(b)

    def step(self, i):
        b = i & 1
        if self.compute is None:
            self._issue(b)
        else:
            import torch
            with torch.cuda.stream(self.compute[b]):
                self._issue(b)
        self._finish(b ^ 1)               # group i-1: its exchange overlapped this group's render

    def drain(self):
        self.release(0)
        self.release(1)
