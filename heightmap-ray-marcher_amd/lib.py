"""ctypes binding of libhmrm.so (C ABI: include/hmrm.h).

This module is plumbing only: it loads the in-tree shared library built from
csrc/ and exposes thin wrappers.  There is NO Python or CPU fallback for the hot
path -- if the library is missing, import fails loudly; if no GPU is usable, the
render calls raise HmrmError(HMRM_E_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhmrm.so")

HMRM_OK = 0
HMRM_E_ARG, HMRM_E_IO, HMRM_E_IMAGE, HMRM_E_CONFIG, HMRM_E_DEVICE, HMRM_E_NOTERM = -1, -2, -3, -4, -5, -6
PERSPECTIVE, SPHERICAL, ORTHOGRAPHIC = 1, 2, 3
NEAREST, BILINEAR, NEAREST_F32 = 0, 1, 2
_PROJ_NAMES = {"perspective": 1, "spherical": 2, "orthographic": 3}


NO_PROBE = 1  # HMRM_NO_PROBE (hmrm.h)


class HmrmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hmrm error {code}: {msg}")
        self.code = code
        self.message = msg


class SceneParams(C.Structure):
    """hmrm_scene_params -- defaults are main/hmap.cpp:38-47,65."""
    _fields_ = [("min_height", C.c_double), ("max_height", C.c_double),
                ("lum_r", C.c_double), ("lum_g", C.c_double), ("lum_b", C.c_double),
                ("grid_width", C.c_double)]

    @classmethod
    def make(cls, min_height=0.0, max_height=10.0, lum=(0.299, 0.587, 0.114), grid_width=0.05):
        return cls(min_height, max_height, lum[0], lum[1], lum[2], grid_width)


class Camera(C.Structure):
    """hmrm_camera -- angles in radians; defaults are main/hmap.cpp:31-35,68,75-85,98,107,110-112."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("projection", C.c_int32),
                ("bg_r", C.c_uint8), ("bg_g", C.c_uint8), ("bg_b", C.c_uint8), ("sampling", C.c_uint8),
                ("hfov", C.c_double), ("hang", C.c_double), ("vang", C.c_double),
                ("pos", C.c_double * 3), ("ortho_width", C.c_double), ("step_dist", C.c_double)]

    @classmethod
    def make(cls, width=800, height=600, projection=PERSPECTIVE, hfov=np.pi / 2.0, hang=-np.pi / 4.0,
             vang=np.pi / 2.0, pos=(-5.0, 5.0, 0.0), ortho_width=0.1, step_dist=0.25, bg=(0, 0, 0), sampling=0):
        if isinstance(projection, str):
            projection = _PROJ_NAMES[projection]
        c = cls()
        c.width, c.height, c.projection = int(width), int(height), int(projection)
        c.bg_r, c.bg_g, c.bg_b = (int(v) & 255 for v in bg)
        c.hfov, c.hang, c.vang = float(hfov), float(hang), float(vang)
        c.pos[0], c.pos[1], c.pos[2] = (float(v) for v in pos)
        c.ortho_width, c.step_dist = float(ortho_width), float(step_dist)
        c.sampling = int(sampling)
        return c


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("steps", C.c_uint64), ("hits", C.c_uint64), ("capped", C.c_uint64),
                ("leap_attempts", C.c_uint64), ("leaps", C.c_uint64), ("groups", C.c_uint64),
                ("leaped_steps", C.c_uint64)]


def degrees_to_rads(deg: float) -> float:
    """DegreesToRads, main/hmap.cpp:131-133 (same two operations, same order)."""
    return (float(deg) / 180.0) * float(np.pi)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C heightmap-ray-marcher_amd/csrc` (there is no fallback path)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and must be
    # the first to load it, otherwise torch later finds "No HIP GPUs" next to the system
    # runtime this library would pull in.  Loading torch first makes libhmrm.so resolve the
    # same, already-loaded runtime (same SONAME).  The C++ CLI uses the system runtime alone.
    if os.environ.get("HMRM_NO_TORCH_PRELOAD", "") == "":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    u8p, dp, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    vp, i32 = C.c_void_p, C.c_int32
    sig = {
        "hmrm_abi_version": (C.c_int, []),
        "hmrm_last_error": (C.c_char_p, []),
        "hmrm_device_count": (C.c_int, []),
        "hmrm_set_device": (C.c_int, [C.c_int]),
        "hmrm_scene_create": (C.c_int, [vp, vp, i32, i32, C.POINTER(SceneParams), C.POINTER(vp)]),
        "hmrm_scene_update": (C.c_int, [vp, C.POINTER(SceneParams)]),
        "hmrm_scene_destroy": (None, [vp]),
        "hmrm_scene_read_heights": (C.c_int, [vp, vp]),
        "hmrm_render": (C.c_int, [vp, C.POINTER(Camera), vp, C.c_size_t]),
        "hmrm_render_cycle": (C.c_int, [vp, C.POINTER(Camera), vp, C.c_size_t, i32, i32]),
        "hmrm_render_multi": (C.c_int, [C.POINTER(vp), i32, C.POINTER(Camera), vp, C.c_size_t]),
        "hmrm_render_rows_device": (C.c_int, [vp, C.POINTER(Camera), vp, C.c_size_t, i32, i32, i32, i32, i32, vp]),
        "hmrm_scene_take_capped": (C.c_int, [vp, vp, C.POINTER(C.c_uint64)]),
        "hmrm_debug_reload_env": (C.c_int, [vp]),
        "hmrm_debug_kernel_choice": (C.c_int, [vp]),
        "hmrm_debug_read_records": (C.c_int, [vp, vp, vp]),
        "hmrm_debug_pick_kernel": (C.c_int, [i32, i32, i32, i32, i32, C.POINTER(i32)]),
        "hmrm_debug_calibrate": (C.c_int, [vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "hmrm_debug_mip_layout": (C.c_int, [i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "hmrm_band_local_rows": (i32, [i32, i32, i32, i32]),
        "hmrm_render_stats": (C.c_int, [vp, C.POINTER(Camera), vp, C.c_size_t, C.POINTER(Stats), vp, vp]),
        "hmrm_debug_ray": (C.c_int, [vp, C.POINTER(Camera), i32, i32, dp, dp, dp]),
        "hmrm_debug_frame": (C.c_int, [C.POINTER(Camera), C.POINTER(SceneParams), i32, i32, vp, vp]),
        "hmrm_debug_plan_order": (C.c_int, [vp, i32, i32, C.POINTER(i32), C.POINTER(i32), vp]),
        "hmrm_debug_rcp_error": (C.c_int, [i32, C.c_uint64, C.c_uint64, i32, i32, dp, C.POINTER(C.c_uint64)]),
        "hmrm_last_kernel_ms": (C.c_double, []),
        "hmrm_bench_kernel_ms": (C.c_double, [vp, C.POINTER(Camera), i32]),
        "hmrm_orbit_camera": (None, [C.POINTER(Camera), C.c_double, C.c_double, C.c_double, C.c_double, i32, i32,
                                     C.POINTER(Camera)]),
        "hmrm_record_orbit": (C.c_int, [vp, C.POINTER(Camera), C.c_double, C.c_double, C.c_double, C.c_double, i32,
                                        C.c_char_p, C.c_longlong, i32, i32]),
        "hmrm_config_record_mode": (i32, [vp]),
        "hmrm_config_devices": (i32, [vp]),
        "hmrm_record_orbit_multi": (C.c_int, [C.POINTER(vp), i32, C.POINTER(Camera), C.c_double, C.c_double, C.c_double,
                                              C.c_double, i32, C.c_char_p, C.c_longlong, i32, i32]),
        "hmrm_orbit_frame_owner": (i32, [i32, i32]),
        "hmrm_render_begin": (C.c_int, [vp, C.POINTER(Camera), C.POINTER(i32)]),
        "hmrm_render_begin_flags": (C.c_int, [vp, C.POINTER(Camera), C.c_uint32, C.POINTER(i32)]),
        "hmrm_render_wait": (C.c_int, [vp, i32, C.POINTER(u8p), C.POINTER(C.c_size_t)]),
        "hmrm_render_release": (None, [vp, i32]),
        "hmrm_render_device_begin": (C.c_int, [vp, C.POINTER(Camera), vp, C.c_size_t, C.POINTER(i32)]),
        "hmrm_render_device_begin_flags": (C.c_int, [vp, C.POINTER(Camera), vp, C.c_size_t, C.c_uint32, C.POINTER(i32)]),
        "hmrm_render_device_wait": (C.c_int, [vp, i32]),
        "hmrm_config_create": (vp, []),
        "hmrm_config_destroy": (None, [vp]),
        "hmrm_config_consume_file": (C.c_int, [vp, C.c_char_p]),
        "hmrm_config_consume_string": (C.c_int, [vp, C.c_char_p]),
        "hmrm_config_log": (C.c_char_p, [vp]),
        "hmrm_config_warnings": (C.c_char_p, [vp]),
        "hmrm_config_get_camera": (None, [vp, C.POINTER(Camera)]),
        "hmrm_config_get_scene_params": (None, [vp, C.POINTER(SceneParams)]),
        "hmrm_config_cycle": (i32, [vp]),
        "hmrm_config_recording_frame_count": (i32, [vp]),
        "hmrm_config_heightmap_path": (C.c_char_p, [vp]),
        "hmrm_config_colormap_path": (C.c_char_p, [vp]),
        "hmrm_config_output_path": (C.c_char_p, [vp]),
        "hmrm_config_height_rgb": (u8p, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "hmrm_config_color_rgba": (u8p, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "hmrm_config_take_heightmap_dirty": (C.c_int, [vp]),
        "hmrm_config_create_scene": (C.c_int, [vp, C.POINTER(vp)]),
        "hmrm_image_load": (C.c_int, [C.c_char_p, i32, C.POINTER(u8p), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "hmrm_image_load_memory": (C.c_int, [vp, C.c_size_t, i32, C.POINTER(u8p), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "hmrm_image_free": (None, [vp]),
        "hmrm_write_png": (C.c_int, [C.c_char_p, i32, i32, i32, vp, C.c_size_t]),
        "hmrm_write_png_memory": (C.c_int, [i32, i32, i32, vp, C.c_size_t, C.POINTER(u8p), C.POINTER(C.c_size_t)]),
        "hmrm_write_ppm": (C.c_int, [C.c_char_p, i32, i32, i32, vp, C.c_size_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = the ABI lost a symbol: fail loudly
        fn.restype = res
        fn.argtypes = args
    del u32p
    return lib, tuple(sig.keys())


lib, EXPORTED_SYMBOLS = _load()


# Device + launch sources: a PMC summary (profiles/traffic.json) is only valid for the kernel it was
# collected on, so it carries this hash and bench.py refuses one that does not match the tree.
KERNEL_SOURCES = ("render_fast.hip", "leap_common.hpp", "leap_diag.hpp", "render.hip", "device_common.hpp", "frame.hpp",
                  "render.hpp", "api.cpp", "launch_order.cpp", "launch_order.hpp", "camera.cpp", "Makefile")


def kernel_src_sha() -> str:
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def last_error() -> str:
    return (lib.hmrm_last_error() or b"").decode("utf-8", "replace")


def _check(rc: int, allow=()):
    if rc != HMRM_OK and rc not in allow:
        raise HmrmError(rc, last_error())
    return rc


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = lib.hmrm_device_count()
    if n < 0:
        raise HmrmError(n, last_error())
    return n


def set_device(i: int):
    _check(lib.hmrm_set_device(int(i)))


# The library reads its environment knobs once per scene.  Tests and tools flip them
# on live scenes, so the Python wrappers re-read them when one changed since the scene last looked.
_ENV_KNOBS = ("HMRM_KERNEL", "HMRM_STEP_CAP", "HMRM_TILE_ORDER", "HMRM_DIAG_ITERS", "HMRM_MIN_LEVEL", "HMRM_FINEST_PAUSE", "HMRM_TILE_SEGMENTS", "HMRM_ORDER_VERBOSE", "HMRM_TRY_GROUP")


def _env_snapshot():
    return tuple(os.environ.get(k) for k in _ENV_KNOBS)


class Scene:
    """Device-resident height + colour maps (replaces the globals filled at hmap.cpp:314-353)."""

    def __init__(self, height_rgb: np.ndarray, color_rgba: np.ndarray, params: SceneParams):
        h_rgb = np.ascontiguousarray(height_rgb, dtype=np.uint8)
        c_rgba = np.ascontiguousarray(color_rgba, dtype=np.uint8)
        if h_rgb.ndim != 3 or h_rgb.shape[2] != 3:
            raise ValueError("height_rgb must be HxWx3 uint8")
        if c_rgba.shape != (h_rgb.shape[0], h_rgb.shape[1], 4):
            # hmap.cpp:503-515
            raise ValueError(f"heightmap dimensions ({h_rgb.shape[1]}x{h_rgb.shape[0]}) must match colormap "
                             f"dimensions ({c_rgba.shape[1]}x{c_rgba.shape[0]})")
        self.map_h, self.map_w = int(h_rgb.shape[0]), int(h_rgb.shape[1])
        self.params = params
        self._h = C.c_void_p()
        self._env = _env_snapshot()
        _check(lib.hmrm_scene_create(_ptr(h_rgb), _ptr(c_rgba), self.map_w, self.map_h,
                                     C.byref(params), C.byref(self._h)))

    def _sync_env(self):
        now = _env_snapshot()
        if now != getattr(self, "_env", None):
            _check(lib.hmrm_debug_reload_env(self._h))
            self._env = now

    @classmethod
    def _adopt(cls, handle, map_w, map_h, params):
        s = cls.__new__(cls)
        s._h, s.map_w, s.map_h, s.params = handle, map_w, map_h, params
        s._env = _env_snapshot()
        return s

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # (lib is None during interpreter shutdown)
            lib.hmrm_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def update(self, params: SceneParams):
        _check(lib.hmrm_scene_update(self._h, C.byref(params)))
        self.params = params

    def read_heights(self) -> np.ndarray:
        out = np.empty((self.map_h, self.map_w), dtype=np.float64)
        _check(lib.hmrm_scene_read_heights(self._h, _ptr(out)))
        return out

    def render(self, cam: Camera) -> np.ndarray:
        """One full frame (hmap.cpp:978-1058 at `cycle 1`) -> HxWx4 uint8."""
        self._sync_env()
        fb = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
        _check(lib.hmrm_render(self._h, C.byref(cam), _ptr(fb), cam.width * 4))
        return fb

    def render_cycle(self, cam: Camera, framebuf: np.ndarray, cycle: int, cycle_period: int):
        """Progressive refresh (hmap.cpp:976-983): rewrites pixels p = cycle (mod cycle_period) in place."""
        assert framebuf.dtype == np.uint8 and framebuf.shape == (cam.height, cam.width, 4) and framebuf.flags.c_contiguous
        self._sync_env()
        _check(lib.hmrm_render_cycle(self._h, C.byref(cam), _ptr(framebuf), cam.width * 4, cycle, cycle_period))

    def render_stats(self, cam: Camera, per_pixel=False, allow_capped=False):
        self._sync_env()
        fb = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
        st = Stats()
        steps = np.empty((cam.height, cam.width), dtype=np.uint32) if per_pixel else None
        entry = np.empty((cam.height, cam.width), dtype=np.float64) if per_pixel else None
        _check(lib.hmrm_render_stats(self._h, C.byref(cam), _ptr(fb), cam.width * 4, C.byref(st),
                                     _ptr(steps) if per_pixel else None, _ptr(entry) if per_pixel else None),
               allow=(HMRM_E_NOTERM,) if allow_capped else ())
        return fb, st, steps, entry

    def render_rows_device(self, cam: Camera, d_ptr: int, stride_bytes: int, row_begin=0, row_end=0,
                           band_rows=0, band_index=0, band_count=1, stream: int = 0):
        self._sync_env()
        _check(lib.hmrm_render_rows_device(self._h, C.byref(cam), C.c_void_p(d_ptr), stride_bytes,
                                           row_begin, row_end, band_rows, band_index, band_count,
                                           C.c_void_p(stream)))

    def render_begin(self, cam: Camera, no_probe: bool = False) -> int:
        """Enqueue a frame (kernel + copy into a pinned frame of the scene's ring) -> ticket.  no_probe = HMRM_NO_PROBE."""
        self._sync_env()
        t = C.c_int32()
        _check(lib.hmrm_render_begin_flags(self._h, C.byref(cam), NO_PROBE if no_probe else 0, C.byref(t)))
        return int(t.value)

    def render_wait(self, ticket: int, shape, allow_capped=False, copy=True) -> np.ndarray:
        """Wait for the frame of `ticket` -> HxWx4 uint8 (a copy, or with copy=False a view of the
        ring's pinned memory that is valid until render_release)."""
        p = C.POINTER(C.c_uint8)()
        stride = C.c_size_t()
        _check(lib.hmrm_render_wait(self._h, ticket, C.byref(p), C.byref(stride)),
               allow=(HMRM_E_NOTERM,) if allow_capped else ())
        h, w = shape
        assert stride.value == w * 4
        a = np.ctypeslib.as_array(p, shape=(h, w, 4))
        return a.copy() if copy else a

    def render_release(self, ticket: int):
        lib.hmrm_render_release(self._h, ticket)

    def render_device_begin(self, cam: Camera, d_ptr: int, stride_bytes: int, no_probe: bool = False) -> int:
        """Launch a frame into device memory on the next of the scene's launch streams -> ticket.  no_probe = HMRM_NO_PROBE."""
        self._sync_env()
        t = C.c_int32()
        _check(lib.hmrm_render_device_begin_flags(self._h, C.byref(cam), C.c_void_p(d_ptr), stride_bytes, NO_PROBE if no_probe else 0, C.byref(t)))
        return int(t.value)

    def render_device_wait(self, ticket: int, allow_capped=False):
        _check(lib.hmrm_render_device_wait(self._h, ticket), allow=(HMRM_E_NOTERM,) if allow_capped else ())

    def take_capped(self, stream: int = 0, allow_capped=False) -> int:
        """Rays of the launches enqueued on `stream` that reached the step cap since the last call
        (waits for the stream); raises HMRM_E_NOTERM for a non-zero count unless allowed."""
        n = C.c_uint64()
        _check(lib.hmrm_scene_take_capped(self._h, C.c_void_p(stream), C.byref(n)),
               allow=(HMRM_E_NOTERM,) if allow_capped else ())
        return int(n.value)

    def debug_ray(self, cam: Camera, px: int, py: int):
        pos, dirv, d = (C.c_double * 3)(), (C.c_double * 3)(), C.c_double()
        _check(lib.hmrm_debug_ray(self._h, C.byref(cam), px, py, pos, dirv, C.byref(d)))
        return np.array(pos[:]), np.array(dirv[:]), d.value

    def read_records(self):
        """(records, thr): the window records as a structured array of shape (ceil(map_h / 4), ceil(map_w / 4)) with fields
        max2 / xs / ys, and the threshold table (map_h x map_w doubles) they were built from."""
        rw, rh = (self.map_w + 3) // 4, (self.map_h + 3) // 4
        dt = np.dtype([("max2", np.float32), ("spare0", np.uint32), ("xs", np.uint8, 8), ("ys", np.uint8, 8), ("spare1", np.uint32, 2)])
        assert dt.itemsize == 32
        recs = np.empty((rh, rw), dtype=dt)
        thr = np.empty((self.map_h, self.map_w), dtype=np.float64)
        _check(lib.hmrm_debug_read_records(self._h, recs.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p)))
        return recs, thr

    def kernel_choice(self) -> int:
        """0 production kernel (leaps), 1 plain groups, 2 literal loop, 3 groups + window records (1 / 3: forced, or chosen by the scene's probe)."""
        return int(lib.hmrm_debug_kernel_choice(self._h))

    def bench_kernel_ms(self, cam: Camera, iters: int) -> float:
        self._sync_env()
        ms = lib.hmrm_bench_kernel_ms(self._h, C.byref(cam), int(iters))
        if ms < 0:
            raise HmrmError(HMRM_E_DEVICE, last_error())
        return ms


def debug_frame(cam: Camera, params: SceneParams, map_w: int, map_h: int):
    """Host-only per-frame record (no GPU): dict of the DevFrame fields + spherical tables."""
    out = np.empty(25, dtype=np.float64)
    tables = np.empty(2 * cam.width + 2 * cam.height, dtype=np.float64)
    _check(lib.hmrm_debug_frame(C.byref(cam), C.byref(params), map_w, map_h, _ptr(out), _ptr(tables)))
    W, H = cam.width, cam.height
    rec = {"cam": out[0:3], "upper_left": out[3:6], "plane_right": out[6:9], "plane_down": out[9:12],
           "look": out[12:15], "c0": out[15:18], "c1": out[18:21], "nudge": out[21], "step_dist": out[22],
           "grid_pow2": int(out[23]), "inv_grid_width": out[24]}
    if cam.projection == SPHERICAL:
        rec.update(col_cos_ha=tables[0:W], col_sin_ha=tables[W:2 * W],
                   row_sin_va=tables[2 * W:2 * W + H], row_cos_va=tables[2 * W + H:])
    return rec


def mip_layout(map_w: int, map_h: int):
    """Pyramid layout hmrm_scene_create would choose (no GPU needed) -> (row pitch, log2 plane pitch, levels,
    production kernel usable: its 32-bit look-up offsets cover every plane)."""
    row, shift, levels = C.c_int32(), C.c_int32(), C.c_int32()
    rc = lib.hmrm_debug_mip_layout(int(map_w), int(map_h), C.byref(row), C.byref(shift), C.byref(levels))
    if rc < 0:
        raise HmrmError(rc, last_error())
    return row.value, shift.value, levels.value, bool(rc)


def pick_kernel(forced=0, use_other=False, records_ok=True, verdict=False, verdict_with_records=False):
    """launch_order.hpp pick_fast_kernel (no GPU) -> (kernel: 0 plain groups, 1 production, 2 records; what the scene remembers
    as the probe's alternative afterwards)."""
    after = C.c_int32()
    k = lib.hmrm_debug_pick_kernel(int(forced), int(use_other), int(records_ok), int(verdict), int(verdict_with_records), C.byref(after))
    return int(k), bool(after.value)


def calibrate(records: np.ndarray, rot: int, may_probe=True, scene_already_probed=False, can_measure=None):
    """hmrm_debug_calibrate: records (launches, tile_rows, 2) uint64 -> dict of the per-launch decisions and the outcome."""
    rec = np.ascontiguousarray(records, dtype=np.uint64)
    n, rows = int(rec.shape[0]), int(rec.shape[1])
    used, meas, grp = (np.zeros(n, dtype=np.int32) for _ in range(3))
    cm = None if can_measure is None else np.ascontiguousarray(can_measure, dtype=np.uint8)
    nt, best, verdict, at = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib.hmrm_debug_calibrate(_ptr(rec), n, rows, int(rot), int(bool(may_probe)), int(bool(scene_already_probed)),
                                    _ptr(cm) if cm is not None else None, _ptr(used), _ptr(meas), _ptr(grp), C.byref(nt), C.byref(best),
                                    C.byref(verdict), C.byref(at)))
    return {"settled_at": int(at.value), "trial": used.tolist(), "measured": meas.tolist(), "group": grp.tolist(), "n_trials": int(nt.value),
            "best": int(best.value), "scene_use_group": bool(verdict.value)}


def plan_order(records: np.ndarray, rot: int):
    """The launch-order plan for measured records ((tile_rows, 2) uint64: start, longest wave) -> (pieces, permutation)."""
    rec = np.ascontiguousarray(records, dtype=np.uint64)
    n_rows = rec.shape[0]
    b, c = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    perm = np.empty(n_rows, dtype=np.int32)
    n = lib.hmrm_debug_plan_order(_ptr(rec), n_rows, rot, b, c, _ptr(perm))
    if n < 0:
        raise HmrmError(n, last_error())
    return [(b[k], c[k]) for k in range(n)], perm


def rcp_error(mode: int, count: int, seed: int = 0, exp_lo: int = 0, exp_hi: int = 0):
    """Largest relative error of v_rcp_f64 over a sample (hmrm_debug_rcp_error) -> (max_rel_err, hist[64])."""
    m = C.c_double()
    hist = (C.c_uint64 * 64)()
    _check(lib.hmrm_debug_rcp_error(mode, count, seed, exp_lo, exp_hi, C.byref(m), hist))
    return m.value, np.array(hist[:], dtype=np.uint64)


def orbit_camera(base: Camera, centre_x: float, centre_y: float, radius: float, hang0: float,
                 frame: int, frames: int) -> Camera:
    """Frame `frame` of an orbit sweep (BASELINE config C5); see hmrm_orbit_camera."""
    out = Camera()
    lib.hmrm_orbit_camera(C.byref(base), centre_x, centre_y, radius, hang0, frame, frames, C.byref(out))
    return out


def record_orbit(scene: "Scene", base: Camera, centre_x, centre_y, radius, hang0, frames, directory, rec_id,
                 encoder_threads=0, verbose=False):
    _check(lib.hmrm_record_orbit(scene._h, C.byref(base), centre_x, centre_y, radius, hang0, frames,
                                 os.fsencode(directory), rec_id, encoder_threads, int(verbose)),
           allow=(HMRM_E_NOTERM,))


def record_orbit_multi(scenes, base: Camera, centre_x, centre_y, radius, hang0, frames, directory, rec_id,
                       encoder_threads=0, verbose=False):
    """The sweep sharded over several scenes (one per GPU): frame k on scenes[k mod len(scenes)]."""
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    _check(lib.hmrm_record_orbit_multi(arr, len(scenes), C.byref(base), centre_x, centre_y, radius, hang0, frames,
                                       os.fsencode(directory), rec_id, encoder_threads, int(verbose)),
           allow=(HMRM_E_NOTERM,))


def render_multi(scenes, cam: Camera, allow_capped=False) -> np.ndarray:
    """One frame over several scenes (one per GPU): scene i renders the cyclic 16-row bands i, i+n, ..."""
    for s in scenes:
        s._sync_env()
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    fb = np.empty((cam.height, cam.width, 4), dtype=np.uint8)
    _check(lib.hmrm_render_multi(arr, len(scenes), C.byref(cam), _ptr(fb), cam.width * 4),
           allow=(HMRM_E_NOTERM,) if allow_capped else ())
    return fb


def orbit_frame_owner(frame: int, n_devices: int) -> int:
    return int(lib.hmrm_orbit_frame_owner(frame, n_devices))


def band_local_rows(height, band_rows, band_index, band_count) -> int:
    return int(lib.hmrm_band_local_rows(height, band_rows, band_index, band_count))


class Config:
    """The reference's config grammar (ConsumeConfigStream, hmap.cpp:309-520)."""

    def __init__(self):
        self._h = C.c_void_p(lib.hmrm_config_create())

    def close(self):
        if getattr(self, "_h", None) and lib is not None:
            lib.hmrm_config_destroy(self._h)
            self._h = None

    __del__ = close

    def consume_string(self, text: str):
        _check(lib.hmrm_config_consume_string(self._h, text.encode()))
        return self

    def consume_file(self, path: str):
        _check(lib.hmrm_config_consume_file(self._h, os.fsencode(path)))
        return self

    @property
    def log(self) -> str:
        return lib.hmrm_config_log(self._h).decode()

    @property
    def warnings(self) -> str:
        return lib.hmrm_config_warnings(self._h).decode()

    def camera(self) -> Camera:
        c = Camera()
        lib.hmrm_config_get_camera(self._h, C.byref(c))
        return c

    def scene_params(self) -> SceneParams:
        p = SceneParams()
        lib.hmrm_config_get_scene_params(self._h, C.byref(p))
        return p

    @property
    def cycle(self) -> int:
        return lib.hmrm_config_cycle(self._h)

    @property
    def recording_frame_count(self) -> int:
        return lib.hmrm_config_recording_frame_count(self._h)

    @property
    def heightmap_path(self) -> str:
        return lib.hmrm_config_heightmap_path(self._h).decode()

    @property
    def colormap_path(self) -> str:
        return lib.hmrm_config_colormap_path(self._h).decode()

    @property
    def record_mode(self) -> int:
        return lib.hmrm_config_record_mode(self._h)

    @property
    def devices(self) -> int:
        return lib.hmrm_config_devices(self._h)

    @property
    def output_path(self) -> str:
        return lib.hmrm_config_output_path(self._h).decode()

    def take_heightmap_dirty(self) -> bool:
        return bool(lib.hmrm_config_take_heightmap_dirty(self._h))

    def _map(self, fn, comp):
        w, h = C.c_int32(), C.c_int32()
        p = fn(self._h, C.byref(w), C.byref(h))
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, comp)).copy()

    def height_rgb(self):
        return self._map(lib.hmrm_config_height_rgb, 3)

    def color_rgba(self):
        return self._map(lib.hmrm_config_color_rgba, 4)

    def create_scene(self) -> Scene:
        h = C.c_void_p()
        _check(lib.hmrm_config_create_scene(self._h, C.byref(h)))
        rgb = self.height_rgb()
        return Scene._adopt(h, rgb.shape[1], rgb.shape[0], self.scene_params())


def image_load(path: str, req_comp: int):
    """stbi_load(path,&w,&h,&n,req_comp) replacement -> (HxWxC uint8, channels_in_file)."""
    out = C.POINTER(C.c_uint8)()
    w, h, n = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib.hmrm_image_load(os.fsencode(path), req_comp, C.byref(out), C.byref(w), C.byref(h), C.byref(n)))
    comp = req_comp if req_comp else n.value
    arr = np.ctypeslib.as_array(out, shape=(h.value, w.value, comp)).copy()
    lib.hmrm_image_free(out)
    return arr, n.value


def image_load_memory(data: bytes, req_comp: int):
    out = C.POINTER(C.c_uint8)()
    w, h, n = C.c_int32(), C.c_int32(), C.c_int32()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    _check(lib.hmrm_image_load_memory(buf, len(data), req_comp, C.byref(out), C.byref(w), C.byref(h), C.byref(n)))
    comp = req_comp if req_comp else n.value
    arr = np.ctypeslib.as_array(out, shape=(h.value, w.value, comp)).copy()
    lib.hmrm_image_free(out)
    return arr, n.value


def png_encode(img: np.ndarray) -> bytes:
    """stbi_write_png_to_mem replacement (same bytes as stb_image_write v1.16)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, comp = img.shape
    out = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    _check(lib.hmrm_write_png_memory(w, h, comp, _ptr(img), w * comp, C.byref(out), C.byref(n)))
    data = bytes(np.ctypeslib.as_array(out, shape=(n.value,)))
    lib.hmrm_image_free(out)
    return data


def write_png(path: str, img: np.ndarray):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, comp = img.shape
    _check(lib.hmrm_write_png(os.fsencode(path), w, h, comp, _ptr(img), w * comp))


def write_ppm(path: str, img: np.ndarray):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, comp = img.shape
    _check(lib.hmrm_write_ppm(os.fsencode(path), w, h, comp, _ptr(img), w * comp))
