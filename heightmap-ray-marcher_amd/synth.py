"""Synthetic scenes for BASELINE.json's configs.

BASELINE.json fixes only map size, resolution, projection and step_dist; the map
content, grid_width and camera are pinned here so that runs are comparable.
Everything is integer arithmetic on a seeded hash, so the maps are identical on
every machine.
"""
from __future__ import annotations

import hashlib
import os
from dataclasses import dataclass

import numpy as np

from . import lib as _lib

SEED = 0x9E3779B9


def _hash2(ix: np.ndarray, iy: np.ndarray, seed: int) -> np.ndarray:
    """32-bit integer hash of lattice coordinates (uint32 in, uint32 out)."""
    with np.errstate(over="ignore"):
        h = (ix.astype(np.uint32) * np.uint32(0x85EBCA6B)) ^ (iy.astype(np.uint32) * np.uint32(0xC2B2AE35))
        h = h ^ np.uint32(seed & 0xFFFFFFFF)
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x7FEB352D)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x846CA68B)
        h ^= h >> np.uint32(16)
    return h


def _noise_rows(size: int, seed: int, octaves: int, total: int, r0: int, r1: int) -> np.ndarray:
    xs = np.broadcast_to(np.arange(size, dtype=np.int64)[None, :], (r1 - r0, size))
    ys = np.broadcast_to(np.arange(r0, r1, dtype=np.int64)[:, None], (r1 - r0, size))
    acc = np.zeros((r1 - r0, size), dtype=np.int64)
    for o in range(octaves):
        spacing = max(size >> (o + 2), 1)
        cx, cy = xs // spacing, ys // spacing
        fx = ((xs % spacing) * 256) // spacing  # 0..255
        fy = ((ys % spacing) * 256) // spacing
        s = (seed + 0x632BE5AB * o) & 0xFFFFFFFF
        v00 = (_hash2(cx, cy, s) >> np.uint32(16)).astype(np.int64)
        v10 = (_hash2(cx + 1, cy, s) >> np.uint32(16)).astype(np.int64)
        v01 = (_hash2(cx, cy + 1, s) >> np.uint32(16)).astype(np.int64)
        v11 = (_hash2(cx + 1, cy + 1, s) >> np.uint32(16)).astype(np.int64)
        top = v00 * (256 - fx) + v10 * fx
        bot = v01 * (256 - fx) + v11 * fx
        val = (top * (256 - fy) + bot * fy) >> 16  # 0..65535
        acc += val >> o
    return ((acc * 255) // total).astype(np.uint8)


def _row_blocks(size: int, px_per_block: int = 1 << 20):
    rows = max(1, min(size, px_per_block // size))  # bounded temporaries (~1 M pixels per block and thread)
    return [(r0, min(size, r0 + rows)) for r0 in range(0, size, rows)]


def _map_blocks(fn, size: int, out: np.ndarray):
    """out[r0:r1] = fn(r0, r1) for every row block, on a few threads (numpy releases the GIL in its loops; the
    arithmetic per block is the same whatever the thread count, so the maps are identical everywhere)."""
    blocks = _row_blocks(size)
    workers = max(1, min(len(blocks), (os.cpu_count() or 1), 16))
    if workers == 1:
        for r0, r1 in blocks:
            out[r0:r1] = fn(r0, r1)
        return out
    from concurrent.futures import ThreadPoolExecutor

    def run(b):
        out[b[0]:b[1]] = fn(b[0], b[1])
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(run, blocks))
    return out


def value_noise_u8(size: int, seed: int = SEED, octaves: int = 6) -> np.ndarray:
    """size x size uint8: `octaves` octaves of bilinearly interpolated lattice noise,
    lattice spacing size/4, size/8, ...; amplitude halves per octave. Integer only."""
    total = sum(65535 >> o for o in range(octaves))
    return _map_blocks(lambda r0, r1: _noise_rows(size, seed, octaves, total, r0, r1), size,
                       np.empty((size, size), dtype=np.uint8))


def _colour_from_heights(v: np.ndarray, seed: int):
    size_y, size_x = v.shape
    xs = np.arange(size_x, dtype=np.int64)[None, :]
    ys = np.arange(size_y, dtype=np.int64)[:, None]
    tex = (_hash2(xs, ys, seed ^ 0x5BD1E995) & np.uint32(0x1F)).astype(np.uint8)
    vi = v.astype(np.int32)
    r = np.clip(vi * 2 - 96, 0, 255).astype(np.uint8) ^ tex
    g = np.clip(64 + (vi * 3) // 4, 0, 255).astype(np.uint8) ^ tex
    b = np.clip(160 - vi, 0, 255).astype(np.uint8) ^ tex
    a = np.full_like(v, 255)
    return np.ascontiguousarray(np.stack([r, g, b, a], axis=2))


def synth_maps(size: int, seed: int = SEED):
    """(height_rgb HxWx3, color_rgba HxWx4) for a size x size map: grey heightmap,
    height-ramp colour map textured by a second hash; alpha 255 everywhere."""
    v = value_noise_u8(size, seed)
    return np.ascontiguousarray(np.repeat(v[:, :, None], 3, axis=2)), _colour_from_heights(v, seed)


# ---- content the traversal does not like (every earlier number sat on one smooth terrain) ----
# The exact-leap traversal jumps as far as the maximum of a window of cells ahead allows, so its speed depends on the
# map: these seeded maps are built to keep window maxima high (pixels stay bit-exact whatever the content; only the
# time changes).  All integer arithmetic, identical on every machine.
CONTENT_KINDS = ("smooth", "white", "spikes", "needles", "canyon")


def content_heights_u8(size: int, kind: str, seed: int = SEED) -> np.ndarray:
    """size x size uint8 grey heights:
      smooth   the 6-octave value noise of synth_maps (every BASELINE workload);
      white    white noise: every cell an independent hash byte (a 4-cell window's maximum is already ~240);
      spikes   the smooth terrain with ONE cell of 255 per 256 x 256-cell block, at a hashed place (every coarse
               window holds a spike: only windows of <= 128 cells can clear a ray below the spike height);
      needles  a plateau at 40 with a needle of 255 in one cell of 64 (hashed; every 8 x 8 block has one on average);
      canyon   the smooth terrain raised to at least 180 except inside a corridor of |x - y| < 48 cells along the
               map's diagonal, where it is the smooth terrain / 8: a camera at low altitude looking down the corridor
               travels below the maxima of every window wider than the corridor."""
    if kind == "smooth":
        return value_noise_u8(size, seed)
    xs = np.arange(size, dtype=np.int64)[None, :]
    ys = np.arange(size, dtype=np.int64)[:, None]
    if kind == "white":
        out = np.empty((size, size), dtype=np.uint8)
        return _map_blocks(lambda r0, r1: (_hash2(np.broadcast_to(xs, (r1 - r0, size)), np.broadcast_to(ys[r0:r1], (r1 - r0, size)),
                                                  seed ^ 0x1B873593) >> np.uint32(24)).astype(np.uint8), size, out)
    if kind == "needles":
        out = np.empty((size, size), dtype=np.uint8)

        def rows(r0, r1):
            h = _hash2(np.broadcast_to(xs, (r1 - r0, size)), np.broadcast_to(ys[r0:r1], (r1 - r0, size)), seed ^ 0x2545F491)
            return np.where((h >> np.uint32(26)) == 0, np.uint8(255), np.uint8(40)).astype(np.uint8)
        return _map_blocks(rows, size, out)
    v = value_noise_u8(size, seed)
    if kind == "spikes":
        nb = (size + 255) // 256
        bx = np.arange(nb, dtype=np.int64)[None, :]
        by = np.arange(nb, dtype=np.int64)[:, None]
        h = _hash2(np.broadcast_to(bx, (nb, nb)), np.broadcast_to(by, (nb, nb)), seed ^ 0x68E31DA4)
        px = np.minimum(bx * 256 + (h & np.uint32(255)).astype(np.int64), size - 1)
        py = np.minimum(by * 256 + ((h >> np.uint32(8)) & np.uint32(255)).astype(np.int64), size - 1)
        v = v.copy()
        v[py.ravel(), px.ravel()] = 255
        return v
    if kind == "canyon":
        inside = np.abs(xs - ys) < 48
        return np.where(inside, v // 8, np.maximum(v, 180)).astype(np.uint8)
    raise ValueError(f"unknown content kind {kind!r} (one of {CONTENT_KINDS})")


def content_maps(size: int, kind: str, seed: int = SEED):
    """(height_rgb, color_rgba) like synth_maps for one of CONTENT_KINDS."""
    if kind == "smooth":
        return synth_maps(size, seed)
    v = content_heights_u8(size, kind, seed)
    return np.ascontiguousarray(np.repeat(v[:, :, None], 3, axis=2)), _colour_from_heights(v, seed)


def maps_sha256(height_rgb: np.ndarray, color_rgba: np.ndarray) -> str:
    h = hashlib.sha256()
    h.update(height_rgb.tobytes())
    h.update(color_rgba.tobytes())
    return h.hexdigest()


@dataclass
class Workload:
    name: str
    map_size: int
    width: int
    height: int
    projection: int
    step_dist: float
    hfov_deg: float
    ortho_width: float = 0.1
    content: str = "smooth"      # one of CONTENT_KINDS
    vang_deg: float = 115.0
    cam_z_over_size: float = 0.25  # camera height as a fraction of the map size (max_height is size / 16)
    grid_width: float = 1.0
    # step_dist and the camera position are given in CELLS and scaled by grid_width, so that a workload at another
    # grid width sees the same cells (the comparison VERDICT r03 asks for: gw 0.05 / 0.3 / 3.0 against gw 1)
    heights_scale_with_grid: bool = True

    def scene_params(self) -> _lib.SceneParams:
        s = float(self.map_size) * (self.grid_width if self.heights_scale_with_grid else 1.0)
        return _lib.SceneParams.make(min_height=0.0, max_height=s / 16.0, grid_width=self.grid_width)

    def maps(self):
        return content_maps(self.map_size, self.content)

    def camera(self, frame: int = 0, frames: int = 1) -> _lib.Camera:
        """Static pose of SURVEY §8(d): pos (-S/8, S/8, S/4), hang -45, vang 115.  With
        frames > 1 the camera orbits the map centre at radius 0.9*S (config C5);
        frame 0 of the orbit is the static pose up to rounding of R."""
        s = float(self.map_size) * self.grid_width
        base = _lib.Camera.make(width=self.width, height=self.height, projection=self.projection,
                                hfov=_lib.degrees_to_rads(self.hfov_deg), hang=_lib.degrees_to_rads(-45.0),
                                vang=_lib.degrees_to_rads(self.vang_deg), pos=(-s / 8.0, s / 8.0, s * self.cam_z_over_size),
                                ortho_width=self.ortho_width * self.grid_width, step_dist=self.step_dist * self.grid_width,
                                bg=(0, 0, 0))
        if frames > 1:
            # one definition of the sweep: the library's (hmrm_orbit_camera)
            return _lib.orbit_camera(base, s / 2.0, -s / 2.0, 0.9 * s, _lib.degrees_to_rads(-45.0), frame, frames)
        return base


# BASELINE.json configs[0..4], made concrete as in BASELINE.md.
WORKLOADS = {
    "C1": Workload("C1", 256, 320, 240, _lib.PERSPECTIVE, 1.0, 90.0),
    "C2": Workload("C2", 1024, 1920, 1080, _lib.PERSPECTIVE, 0.5, 90.0),
    "C3": Workload("C3", 4096, 3840, 2160, _lib.SPHERICAL, 0.25, 180.0),
    # the metric line "3840x2160, 4096^2 heightmap" at the north_star's headline step_dist 0.5
    "C3h": Workload("C3h", 4096, 3840, 2160, _lib.SPHERICAL, 0.5, 180.0),
    "C4": Workload("C4", 8192, 7680, 4320, _lib.ORTHOGRAPHIC, 0.5, 90.0, ortho_width=1.7 * 8192 / 7680),
    "C5": Workload("C5", 4096, 3840, 2160, _lib.PERSPECTIVE, 0.5, 90.0),
}


def content_workload(base: str, kind: str) -> Workload:
    """BASELINE workload `base` over one of the CONTENT_KINDS maps.  `canyon` also moves the camera: low (half the
    box height, below the raised terrain) and nearly level, looking down the corridor from outside the box --
    a camera INSIDE the box renders sky only (AABB.cpp:33-44 rejects d < 0), so "below max_height" is done from
    outside."""
    from dataclasses import replace
    w = WORKLOADS[base]
    if kind == "canyon":
        return replace(w, name=f"{base}/{kind}", content=kind, vang_deg=93.0, cam_z_over_size=1.0 / 32.0)
    return replace(w, name=f"{base}/{kind}", content=kind)


def grid_workload(base: str, grid_width: float) -> Workload:
    """BASELINE workload `base` at another grid_width, seeing the same cells (positions, heights and step in cells)."""
    from dataclasses import replace
    return replace(WORKLOADS[base], name=f"{base}/gw{grid_width:g}", grid_width=grid_width)


# The reference's own operating point (sample_config.txt:5-7: grid_width 0.01, step_dist 0.05 = 5 cells per step;
# hmap.cpp:65,68 defaults are 0.05 / 0.25, also 5 cells) on the C5 frame: 4096^2 map, 3840x2160, perspective.
# Here step_dist is given in cells (x grid_width, see Workload).
WORKLOADS["REFDEF"] = Workload("REFDEF", 4096, 3840, 2160, _lib.PERSPECTIVE, 5.0, 90.0, grid_width=0.01)


def config_text(w: Workload, heightmap_path: str, colormap_path: str, output_path: str | None = None) -> str:
    """The workload as a reference-format config file (sample_config.txt layout, with
    `cycle 1` and the additive `projection` key)."""
    cam = w.camera()
    par = w.scene_params()
    lines = [
        f"resolution {w.width} {w.height}",
        f"hfov {w.hfov_deg:g}",
        "hang -45",
        f"vang {w.vang_deg:g}",
        f"pos {cam.pos[0]:.17g} {cam.pos[1]:.17g} {cam.pos[2]:.17g}",
        "min_height 0.0",
        f"max_height {par.max_height:.17g}",
        f"grid_width {par.grid_width:.17g}",
        f"ortho_width {cam.ortho_width:.17g}",
        f"step_dist {cam.step_dist:.17g}",
        "bg_color 0 0 0",
        "cycle 1",
        f"projection {('perspective', 'spherical', 'orthographic')[w.projection - 1]}",
        f"heightmap {heightmap_path}",
        f"colormap {colormap_path}",
    ]
    if output_path:
        lines.append(f"output {output_path}")
    return "\n".join(lines) + "\n"
