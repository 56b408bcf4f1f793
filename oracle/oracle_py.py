"""ctypes binding of the CPU oracle (oracle/hmrm_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OracleCfg(C.Structure):
    _fields_ = [("screen_width", C.c_int32), ("screen_height", C.c_int32), ("image_plane", C.c_int32),
                ("heightmap_width", C.c_int32), ("heightmap_height", C.c_int32),
                ("hfov", C.c_double), ("hang", C.c_double), ("vang", C.c_double),
                ("cam_pos", C.c_double * 3),
                ("min_height", C.c_double), ("max_height", C.c_double),
                ("grid_width", C.c_double), ("step_dist", C.c_double), ("ortho_width", C.c_double),
                ("bg_r", C.c_uint8), ("bg_g", C.c_uint8), ("bg_b", C.c_uint8), ("sampling", C.c_uint8),
                ("step_cap", C.c_int64)]


def build(force: bool = False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref) with make."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "hmrm_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


_libs = {}


def load(opt: str = "O2"):
    if opt not in _libs:
        build()
        name = "liboracle.so" if opt == "O2" else "liboracle_O0.so"
        lib = C.CDLL(os.path.join(_HERE, "_build", name))
        lib.oracle_update_heightmap.restype = None
        lib.oracle_update_heightmap.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                                C.c_double, C.c_double, C.c_void_p]
        lib.oracle_render.restype = C.c_int64
        lib.oracle_render.argtypes = [C.POINTER(OracleCfg), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        lib.oracle_probe_ray.restype = None
        lib.oracle_probe_ray.argtypes = [C.POINTER(OracleCfg), C.c_int, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.oracle_frame_record.restype = None
        lib.oracle_frame_record.argtypes = [C.POINTER(OracleCfg), C.c_void_p]
        lib.oracle_degrees_to_rads.restype = C.c_double
        lib.oracle_degrees_to_rads.argtypes = [C.c_double]
        lib.oracle_max_threads.restype = C.c_int
        lib.oracle_set_glm_variant.restype = None
        lib.oracle_set_glm_variant.argtypes = [C.c_int]
        _libs[opt] = lib
    return _libs[opt]


DEFAULT_STEP_CAP = 1 << 26


def make_cfg(cam, params, map_w, map_h, step_cap=DEFAULT_STEP_CAP) -> OracleCfg:
    """From the product's Camera / SceneParams ctypes structs (plain field copies)."""
    c = OracleCfg()
    c.screen_width, c.screen_height, c.image_plane = cam.width, cam.height, cam.projection
    c.heightmap_width, c.heightmap_height = map_w, map_h
    c.hfov, c.hang, c.vang = cam.hfov, cam.hang, cam.vang
    c.cam_pos[0], c.cam_pos[1], c.cam_pos[2] = cam.pos[0], cam.pos[1], cam.pos[2]
    c.min_height, c.max_height = params.min_height, params.max_height
    c.grid_width, c.step_dist, c.ortho_width = params.grid_width, cam.step_dist, cam.ortho_width
    c.bg_r, c.bg_g, c.bg_b = cam.bg_r, cam.bg_g, cam.bg_b
    c.sampling = cam.sampling
    c.step_cap = step_cap
    return c


def update_heightmap(height_rgb: np.ndarray, params, opt="O2") -> np.ndarray:
    rgb = np.ascontiguousarray(height_rgb, dtype=np.uint8)
    out = np.empty(rgb.shape[:2], dtype=np.float64)
    load(opt).oracle_update_heightmap(rgb.ctypes.data, rgb.shape[0] * rgb.shape[1], params.lum_r, params.lum_g,
                                      params.lum_b, params.min_height, params.max_height, out.ctypes.data)
    return out


def render(cfg: OracleCfg, heights: np.ndarray, color_rgba: np.ndarray, per_pixel=False, rows=None,
           row_stride=1, threads=0, opt="O2", framebuf=None):
    """Returns (framebuf HxWx4, total_steps, capped, steps_per_pixel|None, entry_d|None)."""
    W, H = cfg.screen_width, cfg.screen_height
    heights = np.ascontiguousarray(heights, dtype=np.float64)
    cmap = np.ascontiguousarray(color_rgba, dtype=np.uint8)
    fb = np.zeros((H, W, 4), dtype=np.uint8) if framebuf is None else framebuf
    steps = np.zeros((H, W), dtype=np.int64) if per_pixel else None
    entry = np.zeros((H, W), dtype=np.float64) if per_pixel else None
    r0, r1 = rows if rows else (0, H)
    capped = C.c_int64(0)
    total = load(opt).oracle_render(C.byref(cfg), heights.ctypes.data, cmap.ctypes.data, fb.ctypes.data,
                                    steps.ctypes.data if per_pixel else None,
                                    entry.ctypes.data if per_pixel else None,
                                    r0, r1, row_stride, threads, C.byref(capped))
    return fb, int(total), int(capped.value), steps, entry


def probe_ray(cfg: OracleCfg, px: int, py: int, opt="O2"):
    pos, dirv, d = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
    load(opt).oracle_probe_ray(C.byref(cfg), px, py, pos, dirv, C.byref(d))
    return np.array(pos[:]), np.array(dirv[:]), d.value


def frame_record(cfg: OracleCfg, opt="O2") -> np.ndarray:
    out = np.zeros(26, dtype=np.float64)
    load(opt).oracle_frame_record(C.byref(cfg), out.ctypes.data)
    return out


def max_threads() -> int:
    return int(load().oracle_max_threads())


def set_glm_variant(v: int, opt: str = "O2"):
    """Sensitivity switch for the unpinned glm formulas (hmrm_oracle.c g_glm_variant); 0 = the assumed ones."""
    load(opt).oracle_set_glm_variant(int(v))
