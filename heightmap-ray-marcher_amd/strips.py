"""Multi-GPU row-strip tiling of ONE frame (BASELINE config C4).

Pixels are independent (main/hmap.cpp:978), so the framebuffer shards with no
exchange during the march.  Each rank holds the whole heightmap and renders a
cyclic set of row bands (band b goes to rank b % world): sky rows cost ~0 steps
and terrain rows thousands, so contiguous H/world strips would be badly
unbalanced.  The only collective is the final gather of the RGBA8 strips to rank
0 (torch.distributed: RCCL over xGMI with backend "nccl", gloo in CPU tests).

A second variant needs no collective at all: every rank copies its strip to ITS OWN pinned host
memory over its own PCIe link (`render_strip_to_host`; what hmrm_render_multi does inside one
process) -- the gather funnels (world-1)/world of the frame through rank 0's links, this does not.

The renderer is passed in as a callable so that the same plan/gather/reassemble
code runs with the HIP path on GPUs and with any row renderer in CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class BandPlan:
    height: int        # framebuffer rows
    width: int
    band_rows: int
    world: int

    @property
    def total_bands(self) -> int:
        return (self.height + self.band_rows - 1) // self.band_rows

    @property
    def bands_per_rank(self) -> int:
        """Every rank's strip is padded to this many bands so that the gather is regular."""
        return (self.total_bands + self.world - 1) // self.world

    @property
    def strip_rows(self) -> int:
        return self.bands_per_rank * self.band_rows

    def bands_of(self, rank: int):
        return list(range(rank, self.total_bands, self.world))

    def rows_of(self, rank: int):
        """Global row indices held by `rank`, in strip order (only rows < height)."""
        rows = []
        for b in self.bands_of(rank):
            rows.extend(range(b * self.band_rows, min((b + 1) * self.band_rows, self.height)))
        return rows


def orbit_frames_of_rank(rank: int, world: int, count: int, orbit_frames: int = 64):
    """Frame sharding of a recording (BASELINE config C5): frame k of the orbit belongs to GPU
    k mod world.  -> the first `count` frames rank `rank` renders, in order: rank, rank + world, ...
    (wrapping around the orbit after `orbit_frames`, so that a bench can time more frames than the
    orbit has).  The C ABI's hmrm_orbit_frame_owner is the same rule seen from the frame."""
    return [k % orbit_frames for k in range(rank, rank + world * count, world)]


def reassemble_numpy(plan: BandPlan, strips) -> np.ndarray:
    """strips[rank] = (strip_rows, width, 4) uint8 -> (height, width, 4)."""
    out = np.zeros((plan.height, plan.width, 4), dtype=np.uint8)
    for rank in range(plan.world):
        for k, b in enumerate(plan.bands_of(rank)):
            g0 = b * plan.band_rows
            g1 = min(g0 + plan.band_rows, plan.height)
            out[g0:g1] = np.asarray(strips[rank])[k * plan.band_rows:k * plan.band_rows + (g1 - g0)]
    return out


def reassemble_torch(plan: BandPlan, gathered):
    """gathered: tensor (world, strip_rows, width, 4) on any device -> (height, width, 4).
    Band b sits at gathered[b % world, (b // world)*band_rows : ...]: one permute + crop."""
    w, bpr, br = plan.world, plan.bands_per_rank, plan.band_rows
    g = gathered.view(w, bpr, br, plan.width, 4).permute(1, 0, 2, 3, 4).reshape(bpr * w * br, plan.width, 4)
    return g[:plan.height].contiguous()


def render_frame_distributed(plan: BandPlan, rank: int, render_rows, dist, strip, block=None):
    """One frame over `plan.world` ranks.

    render_rows(strip, band_rows, band_index, band_count) fills `strip` (a
    (strip_rows, width, 4) uint8 torch tensor on this rank's device) with this
    rank's bands.  The strips are then gathered into `block`
    ((world, strip_rows, width, 4), rank 0 only) and rank 0 returns the
    reassembled (height, width, 4) frame; other ranks return None.
    """
    render_rows(strip, plan.band_rows, rank, plan.world)
    if plan.world == 1 and block is None:
        return reassemble_torch(plan, strip.unsqueeze(0))
    dist.gather(strip, gather_list=list(block.unbind(0)) if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    return reassemble_torch(plan, block)


def render_strip_to_host(plan: BandPlan, rank: int, render_rows, strip, host_strip, sync=None):
    """The no-collective variant: this rank's bands into `strip` (device), then one copy of the strip
    into `host_strip` (pinned host tensor of the same shape) over this rank's own link.  `sync()`
    (e.g. torch.cuda.synchronize) is called afterwards when given.  The frame then exists as
    `world` host strips; `strip_rows_match` / `reassemble_numpy` say where each row lives."""
    render_rows(strip, plan.band_rows, rank, plan.world)
    host_strip.copy_(strip, non_blocking=True)
    if sync is not None:
        sync()
    return host_strip


def strip_rows_match(plan: BandPlan, rank: int, host_strip, frame) -> bool:
    """Does `host_strip` (this rank's strip) hold exactly its rows of `frame` ((height, width, 4))?"""
    hs, fr = np.asarray(host_strip), np.asarray(frame)
    for k, b in enumerate(plan.bands_of(rank)):
        g0 = b * plan.band_rows
        g1 = min(g0 + plan.band_rows, plan.height)
        if not np.array_equal(hs[k * plan.band_rows:k * plan.band_rows + (g1 - g0)], fr[g0:g1]):
            return False
    return True


# ---------------------------------------------------------------------------------------------------------------------
# A SEQUENCE of frames over the ranks, pipelined.  render_frame_distributed is render -> gather ->
# return: the gather of 7/8 of a frame into one GPU is longer than the kernel of an eighth of it (C4: 7 strips of 16.6 MB
# against ~0.07 ms of kernel), and nothing overlaps.  Here
#   * frame k + 1 renders while frame k's strips are on the wire: `depth` strip buffers per rank, the gather on its own
#     stream behind an event of the render stream, a buffer reused only when its gather is done;
#   * a frame is rendered and gathered in `chunks` interleaved band sets (chunk c of rank r = the bands of the virtual
#     rank r + world * c among world * chunks), so that the root starts receiving when the first chunk is rendered, not the
#     last;
#   * `rotate_root`: frame k is gathered to rank k mod world.  Into ONE GPU a C4 frame needs 116 MB / (7 links in) per
#     frame whatever is overlapped; with the root rotating every link carries one strip per direction per `world` frames
#     and the sequence is bound by the kernels (DESIGN.md section 7 has the arithmetic).  Each frame still exists whole on
#     one GPU -- its root -- which is where a consumer of a frame sequence (an encoder per GPU, each GPU's own PCIe link)
#     wants it.
# The code is the same for RCCL on GPUs (streams and events given) and for gloo in the CPU tests (streams None: the
# asynchronous gather's work handle alone orders things).
class StripPipeline:
    def __init__(self, plan: BandPlan, rank: int, dist, torch, device, depth: int = 2, chunks: int = 1,
                 rotate_root: bool = False, render_stream=None, comm_stream=None, keep_log: bool = False):
        assert depth >= 1 and chunks >= 1
        self.plan, self.rank, self.dist, self.torch = plan, rank, dist, torch
        self.depth, self.chunks, self.rotate_root = depth, chunks, rotate_root
        self.render_stream, self.comm_stream = render_stream, comm_stream
        # chunk c of this rank = virtual rank (rank + world * c) of a plan over world * chunks virtual ranks
        self.vplan = BandPlan(height=plan.height, width=plan.width, band_rows=plan.band_rows, world=plan.world * chunks)
        rows = self.vplan.strip_rows
        self.strips = [torch.zeros((chunks, rows, plan.width, 4), dtype=torch.uint8, device=device) for _ in range(depth)]
        # (every rank may be a root when the root rotates)
        self.blocks = [torch.zeros((self.vplan.world, rows, plan.width, 4), dtype=torch.uint8, device=device)
                       if (rotate_root or rank == 0) else None for _ in range(depth)]
        self.inflight = [None] * depth  # per slot: (frame index, root, [work handles], event or None)
        # per slot, GPUs only: the reassembly of the slot's last frame (a copy out of `blocks[slot]`, enqueued on the stream
        # collect() ran on) -- the next gather into that block waits for it, whichever stream the caller renders on
        self.block_read = [None] * depth
        self.log = [] if keep_log else None  # (what, frame, chunk): the order things were issued in (tests only: it grows)

    def _note(self, what, k, c):
        if self.log is not None:
            self.log.append((what, k, c))

    def root_of(self, k: int) -> int:
        return k % self.plan.world if self.rotate_root else 0

    def _drain_slot(self, slot):
        fl = self.inflight[slot]
        if fl is None:
            return None
        k, root, works, ev = fl
        for w in works:
            w.wait()  # (RCCL: makes the current stream wait; gloo: blocks until the bytes are there)
        if ev is not None:
            ev.synchronize()
        self.inflight[slot] = None
        self._note("drained", k, -1)
        return k, root

    def submit(self, k: int, render_rows):
        """Start frame k: render_rows(strip_chunk, band_rows, band_index, band_count) per chunk, each chunk's gather behind
        it.  Returns at once (asynchronous on GPUs); the frame is collected with `collect(k)`."""
        torch = self.torch
        slot = k % self.depth
        # the slot's buffers are free once frame k - depth has been collected: its gather has finished (strips) and its
        # frame has been copied out of the block (reassembly)
        assert self.inflight[slot] is None, "submit(k): frame k - depth has not been collected yet"
        root = self.root_of(k)
        world, works = self.plan.world, []
        for c in range(self.chunks):
            chunk = self.strips[slot][c]
            render_rows(chunk, self.plan.band_rows, self.rank + world * c, self.vplan.world)
            self._note("rendered", k, c)
            dst = list(self.blocks[slot][c * world:(c + 1) * world].unbind(0)) if self.rank == root else None
            if self.comm_stream is not None:
                ready = torch.cuda.Event()
                ready.record(self.render_stream)
                self.comm_stream.wait_event(ready)
                if self.block_read[slot] is not None:
                    self.comm_stream.wait_event(self.block_read[slot])
                with torch.cuda.stream(self.comm_stream):
                    works.append(self.dist.gather(chunk, gather_list=dst, dst=root, async_op=True))
            elif self.dist is not None and world > 1:
                works.append(self.dist.gather(chunk, gather_list=dst, dst=root, async_op=True))
            elif dst is not None:
                dst[0].copy_(chunk)
            self._note("gather issued", k, c)
        ev = None
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            with torch.cuda.stream(self.comm_stream):
                for w in works:
                    w.wait()
                ev.record(self.comm_stream)
            works = []
        self.inflight[slot] = (k, root, works, ev)

    def collect(self, k: int):
        """Wait for frame k.  On its root: the reassembled (height, width, 4) frame; elsewhere None."""
        slot = k % self.depth
        fl = self.inflight[slot]
        assert fl is not None and fl[0] == k, "collect(k): frame k is not in flight (collected already, or overwritten by frame k + depth)"
        _, root = self._drain_slot(slot)
        if self.rank != root:
            return None
        frame = reassemble_torch(self.vplan, self.blocks[slot])
        # The slot's block is written again by frame k + depth: the frame handed out must not be a view of it.  (With one band
        # per virtual rank the reassembly's permute is the identity and its reshape a view, not a copy.)
        if frame.untyped_storage().data_ptr() == self.blocks[slot].untyped_storage().data_ptr():
            frame = frame.clone()
        if self.comm_stream is not None:
            done = self.torch.cuda.Event()
            done.record()  # (the current stream: where the reassembly's copy was enqueued)
            self.block_read[slot] = done
        return frame

    def run(self, frames, render_rows_of, on_frame=None):
        """frames: iterable of frame indices; render_rows_of(k) -> the render_rows callable of frame k.  Keeps `depth`
        frames in flight; on_frame(k, frame_or_None) is called for every frame in order."""
        pending = []
        for k in frames:
            if len(pending) == self.depth:
                j = pending.pop(0)
                f = self.collect(j)
                if on_frame:
                    on_frame(j, f)
            self.submit(k, render_rows_of(k))
            pending.append(k)
        for j in pending:
            f = self.collect(j)
            if on_frame:
                on_frame(j, f)
