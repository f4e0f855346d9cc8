"""Frame tiling across ranks: stripe bookkeeping, the exchange step and the double-buffered frame loop.

The data path of the product is RCCL through the C-ABI (include/rt_hip.h: rt_gather / rt_all_to_all on an RtComm, or the
one-call rt_render_tiled); `RcclExchange` binds it.  `TorchExchange` moves the same buffers with torch.distributed on
host tensors (gloo): it exists for the world_size-2 CPU tests and for bench.py's rehearsal mode, never for numbers.
Both expose the same two calls, so `StripePipeline` and bench.py's loop are the code under test in either case."""
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


def owner_of(rank, frame_index, world, rotate=True):
    """Whose stripes `rank` renders for the frame with index `frame_index` in its group: with rotation every rank plays every
    owner in turn (mirror of rt_render_stripes_batch_rotating), so the ranks' shares of a group are equal."""
    return (rank + frame_index) % world if rotate else rank


def unstripe_host(gathered, height, stripe, world, frame_index=None):
    """gathered[world, max_rows, pitch] (rank-major, padded) -> frame[height, pitch]; host mirror of rt_unstripe, and of
    rt_unstripe_batch_rotating when frame_index (the frame's index in its group) is given."""
    out = np.zeros((height, gathered.shape[2]), gathered.dtype)
    for r in range(world):
        fr = frame_rows_of(height, stripe, r if frame_index is None else owner_of(r, frame_index, world), world)
        out[fr] = gathered[r, :len(fr)]
    return out


# ---- buffer layout when F frames travel in one exchange ------------------------------------------------
# local buffer of a rank:   [F][max_rows][pitch]           (frame f starts f * max_rows rows in)
# gathered buffer on dst:   [world][F][max_rows][pitch]    (rank r starts r * F * max_rows rows in)

def batch_local_ptrs(local_base, F, max_rows, pitch):
    """Device pointers of the F per-frame stripe buffers inside one rank's local buffer."""
    return [local_base + f * max_rows * pitch for f in range(F)]


def batch_unstripe_args(gathered_base, f, F, max_rows, pitch):
    """(d_gathered, rank_stride) to pass to rt_unstripe for frame f of a gathered batch."""
    return gathered_base + f * max_rows * pitch, F * max_rows * pitch


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


def sub_groups(count, world, parts=4):
    """A group of `count` frames that is rendered between two synchronises ON ITS OWN (the driver's `--steps 20`) as a pipeline of
    sub-groups: [(first frame, frames)] -- at most `parts` of them, as even as possible, none smaller than the rank count (a
    rotating-root exchange moves at least one frame slot per rank; smaller sub-groups would carry stale slots).  The exchange of
    sub-group k then runs beside the render of k + 1, and only the last sub-group's exchange and un-stripe pass are left exposed
    behind the render.  20 frames: 8 ranks -> 10 + 10, 4 or 2 ranks -> 4 x 5."""
    n = max(1, min(parts, count // max(world, 1)))
    out, first = [], 0
    for j in range(n):
        c = count // n + (1 if j < count % n else 0)
        out.append((first, c))
        first += c
    return out


# ---- the exchange step ----------------------------------------------------------------------------------
# to_root(local, gathered, root):   every rank's `local` [rows, pitch] to rank `root`, whose `gathered`
#                                   [world, rows, pitch] receives rank r's block at index r (None elsewhere)
# rotating(local, received, count, max_rows): the gathers of a group of `count` frames with a rotating root, fused into
#                                   ONE all-to-all: local is frame-major [>= slots * max_rows, pitch], so the block for
#                                   destination d is contiguous; received [>= world * counts[me] * max_rows, pitch] is
#                                   filled source-major (what rt_unstripe_batch takes as it is)

class RcclExchange:
    """RCCL over xGMI through the C-ABI.  Tensors are torch CUDA tensors (torch is plumbing: device memory and the
    current stream); the collectives are enqueued on torch's current stream and are asynchronous."""
    on_device = True

    def __init__(self, comm):
        self.comm = comm
        self.rank, self.world = comm.rank, comm.num_ranks
        self._plans = {}

    @staticmethod
    def _stream():
        import torch
        return torch.cuda.current_stream().cuda_stream

    def to_root(self, local, gathered, root=0):
        nbytes = local.numel() * local.element_size()
        self.comm.gather(local.data_ptr(), nbytes, gathered.data_ptr() if self.rank == root else None, root, self._stream())

    def rotating(self, local, received, count, max_rows):
        row = local.shape[1] * local.element_size()
        key = (count, max_rows, row)
        plan = self._plans.get(key)
        if plan is None:                                        # (built once per group size: this runs once per group of frames)
            _, counts, offsets, _ = rotating_plan(count, self.world)
            me = self.rank
            plan = self.comm.all_to_all_plan([c * max_rows * row for c in counts], [o * max_rows * row for o in offsets],
                                             [counts[me] * max_rows * row] * self.world,
                                             [r * counts[me] * max_rows * row for r in range(self.world)])
            self._plans[key] = plan
        self.comm.all_to_all_planned(local.data_ptr(), received.data_ptr(), plan, self._stream())


class TorchExchange:
    """The same two calls over torch.distributed: with host tensors (gloo) in the CPU tests and bench.py's rehearsal, with
    device tensors (backend nccl = RCCL driven by PyTorch) as bench.py's announced fallback when the C-ABI communicator
    cannot be created.  The collectives are ordered against torch's current stream."""

    def __init__(self, rank, world, on_device=False):
        self.rank, self.world, self.on_device = rank, world, on_device

    def to_root(self, local, gathered, root=0):
        import torch.distributed as dist
        dist.gather(local, list(gathered.unbind(0)) if self.rank == root else None, dst=root)

    def rotating(self, local, received, count, max_rows):
        import torch.distributed as dist
        slots, counts, _, _ = rotating_plan(count, self.world)
        me = self.rank
        dist.all_to_all_single(received[:self.world * counts[me] * max_rows], local[:slots * max_rows],
                               output_split_sizes=[counts[me] * max_rows] * self.world,
                               input_split_sizes=[c * max_rows for c in counts])


class StripePipeline:
    """Double-buffered frame loop for N ranks.  Group i uses buffer set b = i & 1:

        compute stream of b:  [wait: set b free]  render_fn(b)                      -> event rendered[b]
        comm stream:          [wait: rendered[b]] exchange_fn(b); unstripe_fn(b)    -> event free[b]

    so the exchange and un-stripe pass of group i overlap the render of group i + 1 (which runs on the other compute
    stream), and nothing touches a buffer of set b again before everything that read it has finished: the render of group
    i + 2 waits for free[b], and exchanges / un-stripe passes are ordered among themselves by the one comm stream.
    All three callbacks must enqueue their work on torch's CURRENT stream and return without waiting.
    unstripe_fn runs only on ranks for which `assembles` is true.
    comm_stream = None: everything inline and synchronous on the caller's thread (host backends: tests, rehearsal)."""

    def __init__(self, render_fn, exchange_fn, unstripe_fn, assembles=True, compute_streams=None, comm_stream=None):
        self.render_fn, self.exchange_fn, self.unstripe_fn = render_fn, exchange_fn, unstripe_fn
        self.assembles = assembles
        self.compute = compute_streams
        self.comm = comm_stream
        self.frames_done = 0
        self.rendered = self.free = None
        self.used = [False, False]
        self.timing = False                                     # stage_timing(True): every group records its stage boundaries
        self._marks = []                                        # per timed group: five events (streams) or five clock readings (inline)
        self._pool = []                                         # events created ahead of the timed region (stage_timing)
        if comm_stream is not None:
            import torch
            self.rendered = [torch.cuda.Event(), torch.cuda.Event()]
            self.free = [torch.cuda.Event(), torch.cuda.Event()]

    def stage_timing(self, on, expect_groups=0):
        """Per-group stage times for stage_times(): with streams, timing events at the five boundaries of a group (before the
        wait for its buffer set, render start, render end = exchange may start, exchange end, un-stripe end); inline, the
        host clock at the same places.  Nothing waits for the events.  expect_groups: that many groups' events are created
        HERE, before the timed region, so that the region itself only records them (event creation is a runtime call of
        several microseconds; a group of one short launch is not much longer)."""
        self.timing = bool(on)
        if on:
            self._marks = []
            self._pool = []
            if self.comm is not None and expect_groups > 0:
                import torch
                self._pool = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(expect_groups)]

    def step(self, i):
        b = i & 1
        if self.comm is None:
            import time
            t = [time.perf_counter()] * 2                       # (no buffer wait inline: everything is synchronous)
            self.render_fn(b)
            t.append(time.perf_counter())
            self.exchange_fn(b)
            t.append(time.perf_counter())
            if self.assembles:
                self.unstripe_fn(b)
            t.append(time.perf_counter())
            if self.timing:
                self._marks.append(t)
        else:
            import torch
            cs = self.compute[b] if self.compute is not None else torch.cuda.current_stream()
            ev = (self._pool.pop() if self._pool else [torch.cuda.Event(enable_timing=True) for _ in range(5)]) if self.timing else None
            if ev:
                ev[0].record(cs)                                # completes when the stream has nothing left but this group
            if self.used[b]:
                cs.wait_event(self.free[b])            # group i - 2 has left buffer set b
            with torch.cuda.stream(cs):
                if ev:
                    ev[1].record(cs)
                self.render_fn(b)
                self.rendered[b].record(cs)
                if ev:
                    ev[2].record(cs)
            self.comm.wait_event(self.rendered[b])
            with torch.cuda.stream(self.comm):
                self.exchange_fn(b)
                if ev:
                    ev[3].record(self.comm)
                if self.assembles:
                    self.unstripe_fn(b)
                self.free[b].record(self.comm)
                if ev:
                    ev[4].record(self.comm)
            self.used[b] = True
            if ev:
                self._marks.append(ev)
        self.frames_done += 1

    def drain(self):
        """Host wait for everything issued so far."""
        if self.comm is not None:
            self.comm.synchronize()

    def stage_times(self, groups_per_region=0):
        """Mean milliseconds per timed group of each stage, after drain():
        wait_for_buffer (the compute stream idle until group i - 2 had left the buffer set), render, exchange (from the end
        of the render: includes any wait for the comm stream to finish the previous group), unstripe, and group_span (first
        boundary to last: wait + render + exchange + unstripe = span by construction)."""
        if not self._marks:
            return None
        if self.comm is None:
            d = [[(m[k + 1] - m[k]) * 1e3 for k in range(4)] for m in self._marks]
        else:
            self.drain()
            for m in self._marks:
                m[4].synchronize()
            d = [[m[k].elapsed_time(m[k + 1]) for k in range(4)] for m in self._marks]
        n = float(len(d))
        mean = [sum(row[k] for row in d) / n for k in range(4)]
        out = {"groups_timed": len(d), "wait_for_buffer_ms": round(mean[0], 4), "render_ms_per_group": round(mean[1], 4),
               "exchange_ms_per_group": round(mean[2], 4), "unstripe_ms_per_group": round(mean[3], 4),
               "group_span_ms": round(sum(mean), 4)}
        # How much of group k's exchange + un-stripe pass ran INSIDE the render of group k + 1 (same timed region: groups_per_region
        # consecutive groups between two synchronises): the interval [render end of k, un-stripe end of k] against [render start,
        # render end] of k + 1.  1.0 = the exchange is hidden, 0 = it ran on its own.
        if groups_per_region > 1:
            fr, lens = [], []
            for k in range(len(self._marks) - 1):
                if (k + 1) % groups_per_region == 0:
                    continue                                    # (the next group belongs to the next region: a synchronise lies between)
                a, b = self._marks[k], self._marks[k + 1]
                if self.comm is None:
                    ex_end, r0, r1 = (a[4] - a[2]) * 1e3, (b[1] - a[2]) * 1e3, (b[2] - a[2]) * 1e3
                else:
                    ex_end, r0, r1 = a[2].elapsed_time(a[4]), a[2].elapsed_time(b[1]), a[2].elapsed_time(b[2])
                if ex_end > 0:
                    fr.append(max(0.0, min(ex_end, r1) - max(0.0, r0)) / ex_end)
                    lens.append(ex_end)
            if fr:
                out["exchange_inside_next_render_frac"] = round(sum(fr) / len(fr), 4)
                out["exchange_plus_unstripe_ms"] = round(sum(lens) / len(lens), 4)
        return out
