"""Synthetic scenes for BASELINE.json's configs.

BASELINE.json fixes only map size, resolution, projection and step_dist; the map
content, grid_width and camera are pinned here so that runs are comparable.
Everything is integer arithmetic on a seeded hash, so the maps are identical on
every machine.
"""
from __future__ import annotations

import hashlib
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


def value_noise_u8(size: int, seed: int = SEED, octaves: int = 6) -> np.ndarray:
    """size x size uint8: `octaves` octaves of bilinearly interpolated lattice noise,
    lattice spacing size/4, size/8, ...; amplitude halves per octave. Integer only."""
    out = np.empty((size, size), dtype=np.uint8)
    total = sum(65535 >> o for o in range(octaves))
    rows_per_block = max(1, min(size, (1 << 22) // size))  # bounded temporaries (~4 M pixels per block)
    xs_row = np.arange(size, dtype=np.int64)[None, :]
    for r0 in range(0, size, rows_per_block):
        r1 = min(size, r0 + rows_per_block)
        ys = np.arange(r0, r1, dtype=np.int64)[:, None]
        xs = np.broadcast_to(xs_row, (r1 - r0, size))
        ys = np.broadcast_to(ys, (r1 - r0, size))
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
        out[r0:r1] = ((acc * 255) // total).astype(np.uint8)
    return out


def synth_maps(size: int, seed: int = SEED):
    """(height_rgb HxWx3, color_rgba HxWx4) for a size x size map: grey heightmap,
    height-ramp colour map textured by a second hash; alpha 255 everywhere."""
    v = value_noise_u8(size, seed)
    height_rgb = np.repeat(v[:, :, None], 3, axis=2)
    xs = np.arange(size, dtype=np.int64)[None, :]
    ys = np.arange(size, dtype=np.int64)[:, None]
    tex = (_hash2(xs, ys, seed ^ 0x5BD1E995) & np.uint32(0x1F)).astype(np.uint8)
    vi = v.astype(np.int32)
    r = np.clip(vi * 2 - 96, 0, 255).astype(np.uint8) ^ tex
    g = np.clip(64 + (vi * 3) // 4, 0, 255).astype(np.uint8) ^ tex
    b = np.clip(160 - vi, 0, 255).astype(np.uint8) ^ tex
    a = np.full_like(v, 255)
    color_rgba = np.stack([r, g, b, a], axis=2)
    return np.ascontiguousarray(height_rgb), np.ascontiguousarray(color_rgba)


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

    def scene_params(self) -> _lib.SceneParams:
        s = float(self.map_size)
        return _lib.SceneParams.make(min_height=0.0, max_height=s / 16.0, grid_width=1.0)

    def camera(self, frame: int = 0, frames: int = 1) -> _lib.Camera:
        """Static pose of SURVEY §8(d): pos (-S/8, S/8, S/4), hang -45, vang 115.  With
        frames > 1 the camera orbits the map centre at radius 0.9*S (config C5);
        frame 0 of the orbit is the static pose up to rounding of R."""
        s = float(self.map_size)
        base = _lib.Camera.make(width=self.width, height=self.height, projection=self.projection,
                                hfov=_lib.degrees_to_rads(self.hfov_deg), hang=_lib.degrees_to_rads(-45.0),
                                vang=_lib.degrees_to_rads(115.0), pos=(-s / 8.0, s / 8.0, s / 4.0),
                                ortho_width=self.ortho_width, step_dist=self.step_dist, bg=(0, 0, 0))
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


def config_text(w: Workload, heightmap_path: str, colormap_path: str, output_path: str | None = None) -> str:
    """The workload as a reference-format config file (sample_config.txt layout, with
    `cycle 1` and the additive `projection` key)."""
    cam = w.camera()
    s = float(w.map_size)
    lines = [
        f"resolution {w.width} {w.height}",
        f"hfov {w.hfov_deg:g}",
        "hang -45",
        "vang 115",
        f"pos {cam.pos[0]:.17g} {cam.pos[1]:.17g} {cam.pos[2]:.17g}",
        "min_height 0.0",
        f"max_height {s / 16.0:.17g}",
        "grid_width 1.0",
        f"ortho_width {w.ortho_width:.17g}",
        f"step_dist {w.step_dist:.17g}",
        "bg_color 0 0 0",
        "cycle 1",
        f"projection {('perspective', 'spherical', 'orthographic')[w.projection - 1]}",
        f"heightmap {heightmap_path}",
        f"colormap {colormap_path}",
    ]
    if output_path:
        lines.append(f"output {output_path}")
    return "\n".join(lines) + "\n"
