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
