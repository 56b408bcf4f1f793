"""heightmap-ray-marcher_amd -- MI355X-native (gfx950) heightmap ray marcher.

Host-side mirror of the reference's interface for the hot path only
(config text in -> RGBA8 framebuffer / PNG out) over the C ABI of libhmrm.so
(include/hmrm.h).  Import with importlib (the directory name has hyphens):

    hmrm = importlib.import_module("heightmap-ray-marcher_amd")
"""
from .lib import (  # noqa: F401
    Camera, Config, HmrmError, Scene, SceneParams, Stats,
    PERSPECTIVE, SPHERICAL, ORTHOGRAPHIC, NEAREST, BILINEAR, NEAREST_F32,
    HMRM_OK, HMRM_E_ARG, HMRM_E_IO, HMRM_E_IMAGE, HMRM_E_CONFIG, HMRM_E_DEVICE, HMRM_E_NOTERM,
    EXPORTED_SYMBOLS, LIB_PATH,
    band_local_rows, debug_frame, degrees_to_rads, device_count, image_load, image_load_memory, kernel_src_sha,
    last_error, mip_layout, calibrate, pick_kernel, plan_order, rcp_error, orbit_camera, orbit_frame_owner, png_encode, record_orbit, record_orbit_multi, render_multi, set_device, write_png, write_ppm,
)
from . import synth  # noqa: F401
